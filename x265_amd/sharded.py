"""The two workloads whose units split over GPUs inside ONE picture or ONE lookahead window (strong scaling): an
exhaustive search of a 4K picture against several references, split by (reference, CTU-row band), and the lowres MV
fields of an 8K lookahead window, dealt out field by field. One process per GPU; a rank uploads only the rows / frames
its units read; the only exchange is the gather of the result records (x265_amd.sharding.gather_records).

Both classes do their set-up in __init__ (untimed: synthetic clip, predictors, job lists, uploads), run one pass in
step(), which leaves the rank's contribution to the gather in a device buffer of `records_bytes` bytes (the same on
every rank: padded to the largest share) at `records_ptr` -- the caller's own buffer if it passed one (a torch tensor's
data_ptr() for the RCCL gather), else one the engine allocated.
"""
import numpy as np

from . import sharding, workload
from .engine import FULL, HEX, JOB_DTYPE, RESULT_DTYPE
from .synth import lowres_dims, make_clip, pad_plane


class EsaPicture:
    """--me full --merange 92 --subme 5 over every square 64/32/16/8 PU of one picture against `nrefs` references.
    Units: (reference, CTU row); a rank owns a contiguous run of them (sharding.row_units) and uploads, per run of rows
    of one reference, the source rows of the band and the reference rows the band's search windows reach (band_jobs).
    One launch over all of the rank's bands: the scan per CTU (x265amd_esa_ctu_batch_pairs_dev; a band starts on a CTU
    row, so its PUs form whole CTUs), its records picked into the rank's record order on the device; ctu=False: one
    launch per PU size (x265amd_search_batch_pairs_dev)."""

    SIZES = (64, 32, 16, 8)

    def __init__(self, e, rank, world, width=3840, height=2160, merange=92, nrefs=3, subme=5, qp=32,
                 seed=99, margin=256, sizes=SIZES, records_ptr=None, ctu=True):
        self.e, self.rank, self.world = e, rank, world
        self.W, self.H, self.merange, self.nrefs, self.subme, self.qp, self.M = width, height, merange, nrefs, subme, qp, margin
        self.sizes = sizes
        # frames: two past, the picture, one future (a B picture with two L0 references and one L1 reference)
        clip = make_clip(width, height, nrefs + 1, seed=seed)
        cur_i = min(2, nrefs)
        order = [i for i in (cur_i - 1, cur_i - 2, cur_i + 1, cur_i + 2, cur_i - 3) if 0 <= i <= nrefs][:nrefs]
        self.cur, self.refs = clip[cur_i], [clip[i] for i in order]
        lw, lh, wcu, hcu = lowres_dims(width, height)
        # predictors: the lowres MV field of (picture, reference), computed on this GPU from the whole frames
        full = []
        for f in [self.cur] + self.refs:
            buf, _, stride = pad_plane(f, 64)
            p = e.upload_plane(buf, width, height, 64, 64, stride)
            full.append(e.lowres_init(p))
            e.free_plane(p)
        la_mvs, _, _ = e.lookahead_fields([(full[0][0], full[1 + r]) for r in range(nrefs)], wcu, hcu, 12, HEX, 1, 16)
        for l in full:
            for p in l:
                e.free_plane(p)
        self.jobs = [workload.square_pu_jobs(width, height, la_mvs[r], wcu, hcu, merange, sizes=sizes) for r in range(nrefs)]
        self.nrows = (height + 63) // 64

        # this rank's pieces, their cropped planes and job lists
        self.pieces = sharding.row_units(nrefs, self.nrows, rank, world)
        cbuf, _, stride = pad_plane(self.cur, margin)
        self.planes, pairs, jl = [], [], []
        for k, (r, lo, hi) in enumerate(self.pieces):
            idx, top, bottom = sharding.band_jobs(self.jobs[r], lo, hi)
            rbuf, _, _ = pad_plane(self.refs[r], margin)
            src, hb, halo, y0 = sharding.crop_band(cbuf, stride, margin, height, lo, hi, top, bottom)
            ref, _, _, _ = sharding.crop_band(rbuf, stride, margin, height, lo, hi, top, bottom)
            ps = e.upload_plane(src, width, hb, margin, halo, stride)
            pr = e.upload_plane(ref, width, hb, margin, halo, stride)
            self.planes += [ps, pr]
            pairs.append((ps, pr))
            j = self.jobs[r][idx].copy()
            j["pu_y"] -= y0
            j["reserved"] = k
            jl.append(j)
        mine = np.concatenate(jl) if jl else np.zeros(0, JOB_DTYPE)
        self.my_jobs = mine[np.argsort(-mine["w"].astype(np.int64), kind="stable")]
        self.index = self.rank_index(rank)
        self.n = len(mine)
        self.classes = []
        for s in sizes:
            sel = np.flatnonzero(self.my_jobs["w"] == s)
            if len(sel):
                self.classes.append((int(sel[0]), len(sel), s))
        # validated once through the host entry point (bounds of every job against its band's planes), then resident
        self.npairs = len(pairs)
        self.d_pairs = e.pairs_upload(pairs) if pairs else None
        self.d_jobs = e.device_alloc(max(self.n, 1) * JOB_DTYPE.itemsize)
        if self.n:
            e.upload(self.d_jobs, self.my_jobs.view(np.uint8))
        self.d_res = e.device_alloc(max(self.n, 1) * RESULT_DTYPE.itemsize)
        # what the host encoder reads back: MV and cost, 8 of a record's 16 bytes; padded to the largest rank's count
        self.n_pad = max(len(self.rank_index(q)) for q in range(world))
        self.records_bytes = max(self.n_pad, 1) * 8
        self.own_records = None if records_ptr else e.device_alloc(self.records_bytes)
        self.records_ptr = records_ptr or self.own_records
        self.pairs = pairs
        # the scan per CTU: the same jobs in CTU order, and where each of the rank's records is found in that list
        self.ctu = bool(ctu) and self.n > 0
        self.d_ctu_jobs = self.d_ctu_res = self.d_ctu_index = None
        if self.ctu:
            cj, index, self.nctu = workload.esa_ctu_jobs(self.my_jobs)
            slot_of = np.zeros(self.n, np.int32)
            slot_of[index[index >= 0]] = np.flatnonzero(index >= 0)
            self.d_ctu_jobs = e.device_alloc(len(cj) * JOB_DTYPE.itemsize)
            e.upload(self.d_ctu_jobs, cj.view(np.uint8))
            self.d_ctu_res = e.device_alloc(len(cj) * RESULT_DTYPE.itemsize)
            self.d_ctu_index = e.device_alloc(self.n * 4)
            e.upload(self.d_ctu_index, slot_of.view(np.uint8))
            self.ctu_slot_of = slot_of

    def validate(self):
        """The rank's whole job list once through the entry point that checks every job against its planes."""
        if self.n:
            return self.e.search_batch_pairs(self.pairs, self.my_jobs, self.qp, FULL, self.subme)
        return np.zeros(0, RESULT_DTYPE)

    def step(self):
        if self.ctu:
            self.e.esa_ctu_batch_pairs_dev(self.d_pairs, self.npairs, self.d_ctu_jobs, self.nctu, self.d_ctu_res, self.qp, self.subme)
            self.e.pack_mv_cost_indexed(self.d_ctu_res, self.d_ctu_index, self.n, self.records_ptr)
            self.e.sync()
            return
        for (first, cnt, s) in self.classes:
            self.e.search_batch_pairs_dev(self.d_pairs, self.npairs, self.d_jobs + first * JOB_DTYPE.itemsize, cnt,
                                          self.d_res + first * RESULT_DTYPE.itemsize, self.qp, FULL, self.subme, max_pu_dim=s)
        self.e.pack_mv_cost(self.d_res, self.n, self.records_ptr)
        self.e.sync()

    def records(self):
        """The rank's packed records as a host array (uint8)."""
        return self.e.download(self.records_ptr, self.records_bytes, np.uint8)

    def results(self):
        if not self.n:
            return np.zeros(0, RESULT_DTYPE)
        if self.ctu:
            return self.e.download(self.d_ctu_res, 85 * self.nctu, RESULT_DTYPE)[self.ctu_slot_of]
        return self.e.download(self.d_res, self.n, RESULT_DTYPE)

    def pixel_sads(self):
        return workload.pixel_sads(self.my_jobs, self.results())

    def measured_pixel_sads(self):
        """The SADs the launches measure. Per-size launches: every PU's own range (= pixel_sads of the full-pel stage);
        the scan per CTU: the 8x8 level over the box around the ranges of a CTU's PUs, once for all four depths."""
        if not self.ctu:
            return self.pixel_sads()
        j = self.my_jobs
        key = (j["reserved"].astype(np.int64) << 32) | ((j["pu_y"].astype(np.int64) // 64) << 16) | (j["pu_x"].astype(np.int64) // 64)
        total = 0
        order = np.argsort(key, kind="stable")
        ks, first = np.unique(key[order], return_index=True)
        for a, b in zip(first, list(first[1:]) + [len(order)]):
            g = j[order[a:b]]
            nx = int(g["mvmax"][:, 0].max()) - int(g["mvmin"][:, 0].min()) + 1
            ny = int(g["mvmax"][:, 1].max()) - int(g["mvmin"][:, 1].min()) + 1
            total += nx * ny * 4096
        return total

    def algorithmic_bytes(self):
        return workload.algorithmic_bytes(self.my_jobs)

    def rank_index(self, q):
        return sharding.band_record_index(self.jobs, self.nrows, q, self.world)

    def assemble(self, gathered):
        """Per-rank uint8 arrays of records -> (mv, cost) per reference in the order of the unsharded job lists."""
        return sharding.assemble_band_records(self.jobs, self.nrows, self.world, gathered)

    def close(self):
        for p in self.planes:
            self.e.free_plane(p)
        for d in (self.d_pairs, self.d_jobs, self.d_res, self.own_records, self.d_ctu_jobs, self.d_ctu_res, self.d_ctu_index):
            if d:
                self.e.device_free(d)


def window_fields(nframes, max_distance):
    """Every (frame, reference) search a lookahead window can ask for: all ordered pairs at most max_distance apart
    (slicetype.cpp:1150-1154 runs one search per (b, list, distance); bframes + 1 is the largest distance)."""
    return [(b, r) for b in range(nframes) for r in range(nframes) if b != r and abs(b - r) <= max_distance]


class LookaheadWindow:
    """The lowres MV fields of one lookahead window (HEX, subme 1, merange 16), dealt out in turn over the ranks. A rank
    uploads only the frames its fields name."""

    def __init__(self, e, rank, world, width=7680, height=4320, nframes=10, max_distance=9, qp=12, seed=21, records_ptr=None):
        self.e, self.rank, self.world = e, rank, world
        self.W, self.H, self.qp = width, height, qp
        self.all_fields = window_fields(nframes, max_distance)
        self.mine = sharding.round_robin(len(self.all_fields), rank, world)
        self.clip = make_clip(width, height, nframes, seed=seed)
        lw, lh, self.wcu, self.hcu = lowres_dims(width, height)
        self.ncu = self.wcu * self.hcu
        self.low = {}
        for f in sorted({f for u in self.mine for f in self.all_fields[u]}):
            buf, _, stride = pad_plane(self.clip[f], 64)
            p = e.upload_plane(buf, width, height, 64, 64, stride)
            self.low[f] = e.lowres_init(p)
            e.free_plane(p)
        self.n = len(self.mine)
        self.n_pad = sharding.padded_count(len(self.all_fields), world)
        fields = [(self.low[b][0], self.low[r]) for (b, r) in (self.all_fields[u] for u in self.mine)]
        self.d_fields = e.fields_upload(fields, self.wcu, self.hcu, HEX, 16) if fields else None
        self.records_bytes = self.n_pad * self.ncu * 4
        self.own_records = None if records_ptr else e.device_alloc(self.records_bytes)
        self.records_ptr = records_ptr or self.own_records           # the MV fields (int16 x, y per CU)
        self.d_costs = e.device_alloc(self.n_pad * self.ncu * 4)
        self.d_stats = e.device_alloc(max(self.n, 1) * self.ncu * RESULT_DTYPE.itemsize)

    def step(self, stats=False):
        if self.n:
            self.e.lookahead_fields_dev(self.d_fields, self.n, self.wcu, self.hcu, self.qp, HEX, 1, 16, self.records_ptr,
                                        self.d_costs, self.d_stats if stats else None)
        self.e.sync()

    def records(self):
        """The rank's MV fields as a host array (uint8)."""
        return self.e.download(self.records_ptr, self.records_bytes, np.uint8)

    def pixel_sads(self):
        if not self.n:
            return 0
        self.step(stats=True)
        st = self.e.download(self.d_stats, self.n * self.ncu, RESULT_DTYPE)
        return int((st["sad_calls"].astype(np.int64) + st["subpel_calls"]).sum()) * 64

    def algorithmic_bytes(self):
        """Per field: the source lowres plane, the four reference planes (whole, half-pel x, y, xy), MVs and costs out."""
        lw, lh = self.wcu * 8, self.hcu * 8
        return self.n * (5 * lw * lh + self.ncu * 8)

    def assemble(self, gathered):
        """Per-rank uint8 arrays of records -> MV fields in the order of all_fields."""
        out = np.zeros((len(self.all_fields), self.ncu, 2), np.int16)
        for q, t in enumerate(gathered):
            units = sharding.round_robin(len(self.all_fields), q, self.world)
            out[units] = np.asarray(t, np.uint8).view(np.int16).reshape(-1, self.ncu, 2)[:len(units)]
        return out

    def close(self):
        for l in self.low.values():
            for p in l:
                self.e.free_plane(p)
        for d in (self.d_fields, self.d_costs, self.d_stats, self.own_records):
            if d:
                self.e.device_free(d)
