"""GPU: x265amd_search_batch (HIP) against the CPU oracle and the reference-generated
golden vectors. Bit-exact bar: identical MV, cost, and evaluation counts."""
import numpy as np
import pytest

from x265_amd.engine import METHODS, PARTITIONS, MotionEstimate, make_jobs, EngineError
import golden_util as G
from util import clip_planes, random_jobs, oracle_search, MARGIN, margin_for

pytestmark = pytest.mark.gpu
GPU_METHODS = {k: v for k, v in METHODS.items()}


def _engine(request, depth):
    return request.getfixturevalue("engine8" if depth == 8 else "engine10")


@pytest.mark.parametrize("depth", [8, 10])
def test_golden_vectors(request, depth):
    e = _engine(request, depth)
    planes = {}
    for case, (jobs, exp) in G.me_golden().items():
        d, kind, mname, subme = case
        if d != depth:
            continue
        m = margin_for(METHODS[mname])
        if (kind, m) not in planes:
            fenc, ref, org, stride = clip_planes(G.ME_W, G.ME_H, depth, kind, margin=m)
            planes[(kind, m)] = (e.upload_plane(fenc, G.ME_W, G.ME_H, m, m, stride),
                                 e.upload_plane(ref, G.ME_W, G.ME_H, m, m, stride))
        pf, pr = planes[(kind, m)]
        r = e.search_batch(pf, pr, jobs, 20 + 6 * (depth - 8) + subme, METHODS[mname], subme)
        assert np.array_equal(r["mv"], exp[:, :2]), case
        assert np.array_equal(r["cost"], exp[:, 2]), case
    for pf, pr in planes.values():
        e.free_plane(pf)
        e.free_plane(pr)


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("mname", sorted(GPU_METHODS))
def test_random_jobs_vs_oracle(request, oracle, depth, mname):
    e = _engine(request, depth)
    method = METHODS[mname]
    W, H = 320, 192
    m = margin_for(method)
    for kind in ("texture", "noise", "static"):
        fenc, ref, org, stride = clip_planes(W, H, depth, kind, margin=m)
        pf = e.upload_plane(fenc, W, H, m, m, stride)
        pr = e.upload_plane(ref, W, H, m, m, stride)
        for subme in range(8):
            rng = np.random.default_rng(100 + method * 10 + subme)
            jobs = random_jobs(rng, 96 if method != 4 else 24, W, H, method)
            qp = int(rng.integers(10, 45))
            r = e.search_batch(pf, pr, jobs, qp, method, subme)
            mv, cost, sc, qc = oracle_search(oracle, depth, fenc, [ref], org, stride, jobs, qp, method, subme)
            assert np.array_equal(r["mv"], mv), (kind, subme)
            assert np.array_equal(r["cost"], cost), (kind, subme)
            assert np.array_equal(r["sad_calls"], sc) and np.array_equal(r["subpel_calls"], qc), (kind, subme)
        e.free_plane(pf)
        e.free_plane(pr)


def test_every_partition_every_method(engine8, oracle):
    W, H = 256, 160
    m = margin_for(2)
    fenc, ref, org, stride = clip_planes(W, H, 8, "texture", seed=33, margin=m)
    pf = engine8.upload_plane(fenc, W, H, m, m, stride)
    pr = engine8.upload_plane(ref, W, H, m, m, stride)
    for mname, method in GPU_METHODS.items():
        for part in PARTITIONS[1:]:
            rng = np.random.default_rng(hash((method,) + part) & 0xffff)
            jobs = random_jobs(rng, 6 if method != 4 else 2, W, H, method, parts=[part])
            r = engine8.search_batch(pf, pr, jobs, 26, method, 5)
            mv, cost, _, _ = oracle_search(oracle, 8, fenc, [ref], org, stride, jobs, 26, method, 5)
            assert np.array_equal(r["mv"], mv) and np.array_equal(r["cost"], cost), (mname, part)


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("size", [8, 16, 32, 64])
@pytest.mark.parametrize("mname", ["star", "hex", "dia", "umh"])
def test_small_pu_batches_vs_oracle(request, oracle, mname, size, depth):
    """Batches whose largest PU is a square of 8, 16, 32 or 64 samples take the staged route on 8-bit planes (the search
    up to the full-pel / predictor choice -- for STAR on 8-bit 8x8 / 16x16 PUs the fixed-shape kernel with a list of the
    PUs that leave the common case -- then the sub-pel rounds of the square PUs of that size in their own kernel: four 8x8 or
    16x16 PUs per wave, a wave per 32x32 PU, four waves per 64x64 PU): every subme level, batch sizes that are not
    multiples of four, other partitions mixed in (finished by the one-wave search), and PUs whose full-pel cost is 0
    (no refinement, motion.cpp:1073-1076). 10-bit samples take the same refinement kernels on 16-bit planes."""
    engine8 = _engine(request, depth)
    method = METHODS[mname]
    others = {8: [(8, 4), (4, 8)], 16: [(16, 8), (8, 16), (8, 8), (12, 16), (16, 4)],
              32: [(32, 16), (16, 32), (16, 16), (32, 24), (8, 32)], 64: [(64, 32), (32, 64), (64, 48), (16, 64), (32, 32)]}[size]
    W, H = 320, 192
    m = margin_for(method)
    for kind in ("texture", "noise"):
        fenc, ref, org, stride = clip_planes(W, H, depth, kind, margin=m)
        pf = engine8.upload_plane(fenc, W, H, m, m, stride)
        pr = engine8.upload_plane(ref, W, H, m, m, stride)
        for subme in range(8):
            rng = np.random.default_rng(900 + method * 10 + subme)
            n = 61 + subme
            jobs = random_jobs(rng, n, W, H, method, parts=[(size, size)] * 6 + others)
            qp = int(rng.integers(10, 45)) + 6 * (depth - 8)
            r = engine8.search_batch(pf, pr, jobs, qp, method, subme)
            mv, cost, sc, qc = oracle_search(oracle, depth, fenc, [ref], org, stride, jobs, qp, method, subme)
            assert np.array_equal(r["mv"], mv), (kind, subme)
            assert np.array_equal(r["cost"], cost), (kind, subme)
            assert np.array_equal(r["sad_calls"], sc) and np.array_equal(r["subpel_calls"], qc), (kind, subme)
        if kind == "texture":
            # a picture against itself: zero-predictor PUs end at cost 0 and are left as they are
            jobs = random_jobs(np.random.default_rng(5), 37, W, H, method, parts=[(size, size)])
            r = engine8.search_batch(pf, pf, jobs, 30, method, 5)
            mv, cost, sc, qc = oracle_search(oracle, depth, fenc, [fenc], org, stride, jobs, 30, method, 5)
            assert np.array_equal(r["mv"], mv) and np.array_equal(r["cost"], cost)
            assert np.array_equal(r["sad_calls"], sc) and np.array_equal(r["subpel_calls"], qc)
            assert (cost == 0).any()
        engine8.free_plane(pf)
        engine8.free_plane(pr)


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("mname", ["star", "hex", "dia", "umh"])
def test_small_other_partitions_packed_vs_oracle(request, oracle, mname, depth):
    """The small other partitions (8x4, 4x8 | 16x4, 4x16, 8x8 beside 16x16 | 16x8, 8x16) have kernels that put 8, 4 or 2
    PUs into one wave (star_tiny_kernel for the integer stage of star / hex / dia launches, refine_tiny_kernel for the
    sub-pel rounds of every method): lists long enough for many waves with the shapes of a class mixed inside a wave, with
    waves whose last groups are empty, with squares and larger partitions in between, and on noise, where most PUs are
    handed on to the resumed search (motion.cpp:562-621, :307-547, :1071-1141)."""
    e = _engine(request, depth)
    method = METHODS[mname]
    W, H = 320, 192
    m = margin_for(method)
    small = [(8, 4), (4, 8), (16, 4), (4, 16), (8, 8), (16, 8), (8, 16)]
    for kind, parts, n, subme in (("texture", small, 333, 5), ("noise", small + [(16, 16), (12, 16)], 257, 2),
                                  ("texture", [(8, 4)] * 7 + [(16, 16)], 150, 7), ("static", small, 130, 1),
                                  ("texture", small + [(32, 32), (32, 16), (64, 64)], 201, 3)):
        fenc, ref, org, stride = clip_planes(W, H, depth, kind, margin=m)
        pf = e.upload_plane(fenc, W, H, m, m, stride)
        pr = e.upload_plane(ref, W, H, m, m, stride)
        rng = np.random.default_rng(4000 + method * 10 + subme + n)
        jobs = random_jobs(rng, n, W, H, method, parts=parts)
        qp = int(rng.integers(10, 45)) + 6 * (depth - 8)
        r = e.search_batch(pf, pr, jobs, qp, method, subme)
        mv, cost, sc, qc = oracle_search(oracle, depth, fenc, [ref], org, stride, jobs, qp, method, subme)
        assert np.array_equal(r["mv"], mv), (kind, n)
        assert np.array_equal(r["cost"], cost), (kind, n)
        assert np.array_equal(r["sad_calls"], sc) and np.array_equal(r["subpel_calls"], qc), (kind, n)
        e.free_plane(pf)
        e.free_plane(pr)


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("mname", ["star", "hex", "dia"])
def test_larger_other_partitions_fixed_shape_vs_oracle(request, oracle, mname, depth):
    """The other partitions from 16x12 up have a fixed-shape integer stage of a wave per PU (star_rect_kernel; 16x12 and
    12x16 inside star_tiny_kernel): every one of the fourteen shapes -- strips of quad rows per lane for the power-of-two
    widths, scattered quad rows for 12, 24 and 48 samples, idle lanes for the 48 pairs of 16x12 -- mixed with squares, on
    texture (most PUs end in the kernel), noise (most are handed on to the resumed search) and a static clip
    (motion.cpp:562-621, :307-547)."""
    e = _engine(request, depth)
    method = METHODS[mname]
    W, H = 320, 192
    m = margin_for(method)
    s16 = [(16, 12), (12, 16)]
    s32 = [(32, 16), (16, 32), (32, 8), (8, 32), (32, 24), (24, 32)]
    s64 = [(64, 32), (32, 64), (64, 16), (16, 64), (64, 48), (48, 64)]
    for kind, parts, n, subme in (("texture", s16 + [(16, 16), (16, 8)], 140, 5), ("noise", s16 + s32 + [(32, 32)], 120, 2),
                                  ("texture", s16 + s32 + s64 + [(64, 64), (32, 32), (8, 8)], 170, 6), ("noise", s64 + s32, 60, 1),
                                  ("static", s16 + s32 + s64, 90, 3)):
        fenc, ref, org, stride = clip_planes(W, H, depth, kind, margin=m)
        pf = e.upload_plane(fenc, W, H, m, m, stride)
        pr = e.upload_plane(ref, W, H, m, m, stride)
        rng = np.random.default_rng(5000 + method * 10 + subme + n)
        jobs = random_jobs(rng, n, W, H, method, parts=parts)
        qp = int(rng.integers(10, 45)) + 6 * (depth - 8)
        r = e.search_batch(pf, pr, jobs, qp, method, subme)
        mv, cost, sc, qc = oracle_search(oracle, depth, fenc, [ref], org, stride, jobs, qp, method, subme)
        assert np.array_equal(r["mv"], mv), (kind, n)
        assert np.array_equal(r["cost"], cost), (kind, n)
        assert np.array_equal(r["sad_calls"], sc) and np.array_equal(r["subpel_calls"], qc), (kind, n)
        e.free_plane(pf)
        e.free_plane(pr)


@pytest.mark.parametrize("mname", ["star", "hex", "umh"])
def test_pu_larger_than_the_launch_is_refused(engine8, oracle, mname):
    """x265amd_search_batch_dev with max_pu_dim below some PUs of a device-resident list: those PUs get cost -1 (whatever
    kernel their shape would take in a launch made for them) and every other PU its record."""
    from x265_amd.engine import RESULT_DTYPE
    method = METHODS[mname]
    W, H = 320, 192
    m = margin_for(method)
    fenc, ref, org, stride = clip_planes(W, H, 8, "texture", margin=m)
    pf = engine8.upload_plane(fenc, W, H, m, m, stride)
    pr = engine8.upload_plane(ref, W, H, m, m, stride)
    for dim, parts in ((16, [(16, 16), (16, 8), (8, 16), (8, 8), (16, 12), (12, 16), (8, 4), (32, 16), (16, 32), (32, 32), (24, 32), (32, 8), (64, 32)]),
                       (8, [(8, 8), (8, 4), (4, 8), (16, 8), (8, 16), (16, 16), (16, 4), (4, 16), (16, 12)]),
                       (32, [(32, 32), (32, 16), (16, 16), (8, 8), (24, 32), (64, 32), (32, 64), (64, 16), (48, 64), (64, 64)])):
        jobs = random_jobs(np.random.default_rng(700 + dim + method), 150, W, H, method, parts=parts)
        d_jobs, d_res = engine8.device_alloc(jobs.nbytes), engine8.device_alloc(len(jobs) * RESULT_DTYPE.itemsize)
        engine8.upload(d_jobs, jobs)
        engine8.search_batch_dev(pf, pr, d_jobs, len(jobs), d_res, 30, method, 5, max_pu_dim=dim)
        got = engine8.download(d_res, (len(jobs),), RESULT_DTYPE)
        engine8.device_free(d_jobs)
        engine8.device_free(d_res)
        big = (jobs["w"] > dim) | (jobs["h"] > dim)
        assert big.any() and (~big).any()
        assert (got["cost"][big] == -1).all(), (dim, jobs[big][got["cost"][big] != -1][["w", "h"]])
        mv, cost, sc, qc = oracle_search(oracle, 8, fenc, [ref], org, stride, jobs[~big], 30, method, 5)
        assert np.array_equal(got["mv"][~big], mv) and np.array_equal(got["cost"][~big], cost), dim
        assert np.array_equal(got["sad_calls"][~big], sc) and np.array_equal(got["subpel_calls"][~big], qc), dim
    engine8.free_plane(pf)
    engine8.free_plane(pr)


@pytest.mark.parametrize("depth", [8, 10])
def test_pairs_in_one_launch_vs_oracle(request, oracle, depth):
    """x265amd_search_batch_pairs: jobs of three (source, reference) pairs of different sizes and strides in one launch,
    interleaved job by job (so the four 8x8 PUs one refinement wave takes belong to different pictures), every method,
    mixed-size and 8x8-only batches."""
    e = _engine(request, depth)
    geo = ((320, 192, "texture", 7), (256, 160, "noise", 8), (192, 128, "texture", 9))
    for mname, method in sorted(GPU_METHODS.items()):
        m = margin_for(method)
        host, pairs = [], []
        for (W, H, kind, seed) in geo:
            fenc, ref, org, stride = clip_planes(W, H, depth, kind, seed=seed, margin=m)
            host.append((fenc, ref, org, stride, W, H))
            pairs.append((e.upload_plane(fenc, W, H, m, m, stride), e.upload_plane(ref, W, H, m, m, stride)))
        for parts, n in ((None, 90), ([(8, 8)] * 5 + [(4, 8)], 75)):
            rng = np.random.default_rng(300 + method)
            per = [random_jobs(rng, n // 3 if method != 4 else 8, W, H, method, parts=parts) for (W, H, _, _) in geo]
            jobs = make_jobs(sum(len(j) for j in per))
            order = np.argsort(np.concatenate([np.arange(len(j)) * 3 + k for k, j in enumerate(per)]), kind="stable")
            allj = np.concatenate(per)
            which = np.concatenate([np.full(len(j), k) for k, j in enumerate(per)])
            jobs[:] = allj[order]
            jobs["reserved"] = which[order]
            r = e.search_batch_pairs(pairs, jobs, 28 + 6 * (depth - 8), method, 5)
            for k, (fenc, ref, org, stride, W, H) in enumerate(host):
                sel = jobs["reserved"] == k
                mv, cost, sc, qc = oracle_search(oracle, depth, fenc, [ref], org, stride, jobs[sel], 28 + 6 * (depth - 8), method, 5)
                assert np.array_equal(r["mv"][sel], mv) and np.array_equal(r["cost"][sel], cost), (mname, k)
                assert np.array_equal(r["sad_calls"][sel], sc) and np.array_equal(r["subpel_calls"][sel], qc), (mname, k)
        bad = jobs.copy()
        bad["reserved"][0] = 3
        with pytest.raises(EngineError):
            e.search_batch_pairs(pairs, bad, 28, method, 5)        # pair index beyond the table
        for pf, pr in pairs:
            e.free_plane(pf)
            e.free_plane(pr)


def test_edge_cases(engine8, oracle):
    """Degenerate ranges, zero-cost blocks (sub-pel skipped, motion.cpp:1073), picture corners."""
    W, H = 128, 96
    m = margin_for(2)
    fenc, _, org, stride = clip_planes(W, H, 8, "static", margin=m)
    pf = engine8.upload_plane(fenc, W, H, m, m, stride)
    jobs = make_jobs(6)
    for i, (px, py, w, h, mn, mx, mvp) in enumerate([
            (0, 0, 8, 8, (0, 0), (0, 0), (0, 0)),                 # single-point range
            (W - 64, H - 64, 64, 64, (-8, -8), (8, 8), (3, -5)),  # bottom-right corner CTU
            (0, 0, 64, 64, (-57, -57), (57, 57), (0, 0)),         # window reaches into the margins
            (64, 32, 16, 16, (-3, 2), (5, 2), (40, 40)),          # predictor outside the range (gets clipped)
            (32, 32, 4, 8, (-16, -16), (16, 16), (-7, 9)),
            (32, 32, 12, 16, (-16, -16), (16, 16), (1, 1))]):
        j = jobs[i]
        j["pu_x"], j["pu_y"], j["w"], j["h"] = px, py, w, h
        j["mvmin"], j["mvmax"], j["qmvp"], j["merange"] = mn, mx, mvp, 16
        j["numc"] = 2
        j["mvc"][0] = (0, 0)
        j["mvc"][1] = (400, -400)
    for method in GPU_METHODS.values():
        # identical source and reference: the best cost is 0 wherever (0,0) is reachable
        r = engine8.search_batch(pf, pf, jobs, 30, method, 5)
        mv, cost, sc, qc = oracle_search(oracle, 8, fenc, [fenc], org, stride, jobs, 30, method, 5)
        assert np.array_equal(r["mv"], mv) and np.array_equal(r["cost"], cost)
        assert np.array_equal(r["sad_calls"], sc) and np.array_equal(r["subpel_calls"], qc)
    assert len(engine8.search_batch(pf, pf, make_jobs(0), 30, 1, 5)) == 0   # empty batch


def test_device_resident_jobs_are_checkable(engine8):
    """x265amd_check_jobs_dev / _pairs_dev: the host entry points' checks for job lists the *_dev entry points run as
    they are -- a good list passes, a list with two jobs that must not run is refused with the first one named."""
    W, H, m = 128, 128, margin_for(3)
    fenc, _, org, stride = clip_planes(W, H, 8, "static", margin=m)
    pf = engine8.upload_plane(fenc, W, H, m, m, stride)
    jobs = random_jobs(np.random.default_rng(3), 200, W, H, 3)
    d = engine8.device_alloc(jobs.nbytes)
    engine8.upload(d, jobs.view(np.uint8))
    engine8.check_jobs_dev(pf, pf, d, len(jobs), 3)
    assert engine8.check_jobs_pairs_dev([(pf, pf)], d, len(jobs), 3) == (0, -1)
    bad = jobs.copy()
    bad["mvmax"][57] = (300, 4)         # window leaves the padded plane
    bad["w"][150] = 20                  # not a partition
    engine8.upload(d, bad.view(np.uint8))
    with pytest.raises(EngineError, match="first is job 57: search window"):
        engine8.check_jobs_dev(pf, pf, d, len(jobs), 3)
    assert engine8.check_jobs_pairs_dev([(pf, pf)], d, len(jobs), 3) == (2, 57)
    bad = jobs.copy()
    bad["reserved"][9] = 1              # names a pair the table does not have
    engine8.upload(d, bad.view(np.uint8))
    assert engine8.check_jobs_pairs_dev([(pf, pf)], d, len(jobs), 3) == (1, 9)
    engine8.check_jobs_dev(pf, pf, d, len(jobs), 3)      # one pair of planes for all jobs: the field is not read
    engine8.device_free(d)
    engine8.free_plane(pf)


def test_clock_probe(engine8):
    """The measurement aid behind the issue-rate bounds: a plausible shader clock, and the argument checks."""
    engine8.clock_probe_start(2.0)
    mhz = engine8.clock_probe_read()
    assert 300.0 < mhz < 4000.0, mhz
    with pytest.raises(EngineError):
        engine8.clock_probe_start(0.0)


def test_argument_errors(engine8):
    W, H = 64, 64
    fenc, _, org, stride = clip_planes(W, H, 8, "static", margin=32)
    pf = engine8.upload_plane(fenc, W, H, 32, 32, stride)
    j = make_jobs(1)
    j["w"], j["h"] = 16, 16
    j["mvmin"][0], j["mvmax"][0] = (-16, -16), (16, 16)
    j["merange"] = 16
    engine8.search_batch(pf, pf, j, 20, 1, 5)
    bad = j.copy(); bad["w"] = 20
    with pytest.raises(EngineError):
        engine8.search_batch(pf, pf, bad, 20, 1, 5)            # not a partition
    bad = j.copy(); bad["mvmax"][0] = (200, 16)
    with pytest.raises(EngineError):
        engine8.search_batch(pf, pf, bad, 20, 1, 5)            # window leaves the padded plane
    with pytest.raises(EngineError):
        engine8.search_batch(pf, pf, j, 20, 9, 5)              # unknown method
    with pytest.raises(EngineError):
        engine8.search_batch(pf, pf, j, 99, 1, 5)              # qp out of range


def test_motion_estimate_class_mirror(engine8, oracle):
    """The per-call class reads like the reference call site (TEncSearch.cpp:4266-4303)."""
    W, H = 192, 128
    fenc, ref, org, stride = clip_planes(W, H, 8, "texture")
    pf = engine8.upload_plane(fenc, W, H, MARGIN, MARGIN, stride)
    pr = engine8.upload_plane(ref, W, H, MARGIN, MARGIN, stride)
    me = MotionEstimate(engine8)
    me.setSourcePlane(pf)
    me.setQP(28)
    me.setSearchMethod(METHODS["star"])
    me.setSubpelRefine(5)
    me.setSourcePU(64, 32, 32, 32)
    mvc = [(4, -8), (0, 0), (12, 9)]
    cost, mv = me.motionEstimate(pr, (-57, -57), (57, 57), (5, 3), 3, mvc, 57)
    ctx = oracle.make_ctx(8, fenc, org, stride, [ref], org, stride, False, 28, 3, 5)
    assert oracle.me_search(ctx, 32 * stride + 64, 32, 32, (-57, -57), (57, 57), (5, 3), mvc, 57) == (mv, cost)
    for q in ((0, 0), (2, 0), (0, 3), (1, 2), (-5, 7)):
        for cmp in (0, 1):
            assert me.subpelCompare(pr, q, cmp) == oracle.subpel_compare(ctx, 32 * stride + 64, 32, 32, q, cmp)


def test_cost_tables_vs_reference(engine8, engine10):
    """x265amd_cost_table returns the table the kernels read (device copy): equal to the reference's BitCost::setQP table
    for every QP (bitcost.cpp:34-83; golden tables made by the reference build with its own -ffast-math flags)."""
    gold = G.cost_tables()
    assert sorted(gold) == list(range(70))
    for qp, tab in gold.items():
        assert np.array_equal(engine8.cost_table(qp), tab), qp
    for qp in (0, 24, 51, 69):
        assert np.array_equal(engine10.cost_table(qp), gold[qp]), qp
    with pytest.raises(EngineError):
        engine8.cost_table(70)


def test_full_size_properties(engine8):
    """1080p, every 16x16 PU: size-independent properties instead of a CPU replay.
    (a) searching a frame against itself finds MV 0 with cost 0 everywhere;
    (b) against a copy shifted by (+3, -2) pixels an exhaustive search recovers exactly that shift with SAD 0
        (interior PUs), so bcost == mvcost of the shift."""
    from x265_amd.synth import make_clip, pad_plane
    W, H = 1920, 1080
    img = make_clip(W, H, 1, seed=5, kind="noise")[0]
    buf, org, stride = pad_plane(img, 80)
    pf = engine8.upload_plane(buf, W, H, 80, 80, stride)
    shifted = np.roll(img, (-2, 3), axis=(0, 1))      # ref(x + 3, y - 2) == cur(x, y)
    bufs, _, _ = pad_plane(shifted, 80)
    ps = engine8.upload_plane(bufs, W, H, 80, 80, stride)
    xs, ys = np.meshgrid(np.arange(16, W - 32, 16), np.arange(16, H - 32, 16))
    jobs = make_jobs(xs.size)
    jobs["pu_x"], jobs["pu_y"] = xs.ravel(), ys.ravel()
    jobs["w"] = 16; jobs["h"] = 16
    jobs["mvmin"] = (-4, -4); jobs["mvmax"] = (4, 4); jobs["merange"] = 4
    r = engine8.search_batch(pf, pf, jobs, 30, METHODS["hex"], 5)
    assert (r["cost"] == 0).all() and (r["mv"] == 0).all()
    r = engine8.search_batch(pf, ps, jobs, 30, METHODS["full"], 5)
    assert (r["mv"] == (12, -8)).all()
    t = G.cost_tables()[30].astype(np.int64)      # the reference's own BitCost table
    assert (r["cost"] == t[32767 + 12] + t[32767 - 8]).all()
    assert (r["sad_calls"] == 81).all()   # the 9x9 raster; the zero predictor itself is measured through subpelCompare


@pytest.mark.parametrize("depth,kind", [(8, "texture"), (8, "static"), (10, "texture"), (10, "static"), (12, "noise")])
def test_exhaustive_search_windows(engine8, engine10, oracle, depth, kind):
    """X265_FULL_SEARCH through the LDS-window kernels (8-bit: packed quad-SAD, 4 columns x 8 rows per lane; deeper
    samples: v_sad_u16, 2 columns x 8 rows): ranges that are not multiples of the lane patch, asymmetric and clipped
    ranges, every partition height class, and the same jobs again with the smallest window budget so that the range is
    scanned in several x / y strips. A static clip makes every position of a range cost the same SAD, so the winner is
    decided by the table costs and the raster order alone; 12-bit noise gives the largest sums."""
    from x265_amd.engine import Engine
    W, H = 256, 160
    m = 110
    fenc, ref, org, stride = clip_planes(W, H, depth, kind, seed=21, margin=m)
    own = Engine(0, depth) if depth == 12 else None
    engine8 = {8: engine8, 10: engine10, 12: own}[depth]      # the engine of this depth
    pf = engine8.upload_plane(fenc, W, H, m, m, stride)
    pr = engine8.upload_plane(ref, W, H, m, m, stride)
    rng = np.random.default_rng(77)
    parts = [(8, 8), (16, 16), (32, 32), (64, 64), (12, 16), (16, 4), (4, 16), (24, 32), (64, 16), (48, 64), (8, 32)]
    jobs = make_jobs(len(parts) * 2)
    for i in range(len(jobs)):
        w, h = parts[i % len(parts)]
        j = jobs[i]
        j["pu_x"], j["pu_y"], j["w"], j["h"] = int(rng.integers(0, (W - w) // 4 + 1)) * 4, int(rng.integers(0, (H - h) // 4 + 1)) * 4, w, h
        rx0, rx1, ry0, ry1 = (int(v) for v in rng.integers(1, 45, 4))
        qmvp = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
        cx, cy = qmvp[0] >> 2, qmvp[1] >> 2
        j["mvmin"], j["mvmax"] = (cx - rx0, cy - ry0), (cx + rx1, cy + ry1)
        j["qmvp"] = qmvp
        j["numc"] = 2
        j["mvc"][0] = (int(rng.integers(-60, 61)), int(rng.integers(-60, 61)))
        j["mvc"][1] = (0, 4)
        j["merange"] = 45
    mv, cost, sc, qc = oracle_search(oracle, depth, fenc, [ref], org, stride, jobs, 27, METHODS["full"], 5)
    for budget in (65536, 8192 if depth == 8 else 16384):
        engine8.set_esa_window_bytes(budget)
        r = engine8.search_batch(pf, pr, jobs, 27, METHODS["full"], 5)
        assert np.array_equal(r["mv"], mv) and np.array_equal(r["cost"], cost), budget
        assert np.array_equal(r["sad_calls"], sc) and np.array_equal(r["subpel_calls"], qc), budget
    engine8.set_esa_window_bytes(36864)      # what the library takes for most launches when none is set
    with pytest.raises(EngineError):
        engine8.set_esa_window_bytes(100)
    engine8.free_plane(pf)
    engine8.free_plane(pr)
    if own:
        own.close()


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("mname", ["star", "hex"])
@pytest.mark.parametrize("psize", [64, 16])
def test_children_refined_with_their_parent(request, oracle, mname, depth, psize):
    """A launch for 64x64 PUs whose list has every 64x64 PU followed by its four 32x32 children: where the five start
    their refinement from one MV the children's candidates are measured with the parent's (the sums of the quadrants),
    as long as every decision keeps them together; otherwise, and for children in any other place of the list, each is
    refined on its own. Coherent motion (the children mostly follow), predictors of their own per child (they mostly
    do not), orphans and strays, every subme level. The same lists of 16x16 PUs and 8x8 children (each PU on its own
    there) must give the oracle's results as well."""
    e = _engine(request, depth)
    method = METHODS[mname]
    W, H = 320, 192
    m = margin_for(method)
    fenc, ref, org, stride = clip_planes(W, H, depth, "texture", margin=m)
    pf = e.upload_plane(fenc, W, H, m, m, stride)
    pr = e.upload_plane(ref, W, H, m, m, stride)
    shared = 0
    for subme in range(8):
        rng = np.random.default_rng(4000 + subme)
        jl = []
        half = psize // 2
        for cy in range(H // psize if psize == 64 else 5):
            for cx in range(W // psize if psize == 64 else 9):
                coherent = (cx + cy + subme) % 3 != 0
                fam = random_jobs(rng, 5, W, H, method, parts=[(psize, psize)], meranges=(16, 24))
                fam["pu_x"], fam["pu_y"] = cx * psize + 32 * (psize == 16), cy * psize + 48 * (psize == 16)
                for q in range(1, 5):
                    fam[q]["w"] = fam[q]["h"] = half
                    fam[q]["pu_x"] += half * ((q - 1) & 1)
                    fam[q]["pu_y"] += half * ((q - 1) >> 1)
                    if coherent:
                        for k in ("qmvp", "numc", "mvc", "merange"):
                            fam[q][k] = fam[0][k]
                for j in fam:           # the range of each PU around its own predictor
                    fx, fy = int(j["qmvp"][0]) >> 2, int(j["qmvp"][1]) >> 2
                    mr = int(j["merange"])
                    j["mvmin"] = (max(fx - mr, -int(j["pu_x"]) - 60), max(fy - mr, -int(j["pu_y"]) - 60))
                    j["mvmax"] = (min(fx + mr, W - int(j["pu_x"]) - int(j["w"]) + 60), min(fy + mr, H - int(j["pu_y"]) - int(j["h"]) + 60))
                if (cx + cy) % 4 == 3:
                    fam = fam[[0, 2, 1, 3, 4]]      # children out of order: nobody is taken along
                jl.append(fam)
        strays = [(32, 32), (64, 64), (16, 16), (64, 32)] if psize == 64 else [(8, 8), (16, 16), (16, 8), (8, 4)]
        jl.append(random_jobs(rng, 9, W, H, method, parts=strays))
        jobs = np.concatenate(jl)
        qp = int(rng.integers(10, 45)) + 6 * (depth - 8)
        r = e.search_batch(pf, pr, jobs, qp, method, subme)
        mv, cost, sc, qc = oracle_search(oracle, depth, fenc, [ref], org, stride, jobs, qp, method, subme)
        assert np.array_equal(r["mv"], mv), subme
        assert np.array_equal(r["cost"], cost), subme
        assert np.array_equal(r["sad_calls"], sc) and np.array_equal(r["subpel_calls"], qc), subme
        par = np.flatnonzero((jobs["w"][:-9] == psize) & (jobs["h"][:-9] == psize))
        shared += sum(int((mv[p + 1:p + 5] == mv[p]).all()) for p in par if p + 4 < len(jobs))
    assert shared > 0       # families that ended on one MV did occur
    e.free_plane(pf)
    e.free_plane(pr)
