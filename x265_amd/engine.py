"""Host-side mirror of the reference's ME interface over the C ABI (ctypes).

``Engine`` wraps one ``x265amd_ctx`` (include/x265_amd_me.h); ``MotionEstimate``
mirrors ``class MotionEstimate`` of the reference (source/encoder/motion.h:35-120):
same method names, argument meaning and order, one PU per call.  The batch entry
points are what a real caller uses; the per-call class exists so tests read like
the reference's call sites (TEncSearch.cpp:4266-4303, slicetype.cpp:1415-1468).

There is no CPU fallback: if the HIP library is missing or no GPU is present the
constructor raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# X265AMD_LIB names another build of the same library (timing / experiment builds under tools/); never a fallback
LIB_PATH = os.environ.get("X265AMD_LIB") or os.path.join(HERE, "libx265_amd_me.so")

DIA, HEX, UMH, STAR, FULL = 0, 1, 2, 3, 4
SQUARE_ONLY = 0x100      # x265amd_search_batch*_dev: max_pu_dim | SQUARE_ONLY (the list holds only squares of that size)
METHODS = {"dia": DIA, "hex": HEX, "umh": UMH, "star": STAR, "full": FULL}

PARTITIONS = [(4, 4), (8, 8), (8, 4), (4, 8), (16, 16), (16, 8), (8, 16), (16, 12), (12, 16), (16, 4), (4, 16),
              (32, 32), (32, 16), (16, 32), (32, 24), (24, 32), (32, 8), (8, 32),
              (64, 64), (64, 32), (32, 64), (64, 48), (48, 64), (64, 16), (16, 64)]

JOB_DTYPE = np.dtype([("pu_x", "<i2"), ("pu_y", "<i2"), ("w", "u1"), ("h", "u1"), ("numc", "u1"), ("reserved", "u1"),
                      ("mvmin", "<i2", (2,)), ("mvmax", "<i2", (2,)), ("qmvp", "<i2", (2,)),
                      ("mvc", "<i2", (4, 2)), ("merange", "<i4")], align=True)
RESULT_DTYPE = np.dtype([("mv", "<i2", (2,)), ("cost", "<i4"), ("sad_calls", "<u4"), ("subpel_calls", "<u4")], align=True)
assert JOB_DTYPE.itemsize == 40 and RESULT_DTYPE.itemsize == 16
# x265amd_rk_ctu / x265amd_rk_result
RK_CTU_DTYPE = np.dtype([("ctu_x", "<i2"), ("ctu_y", "<i2"), ("valid_w", "<i2"), ("valid_h", "<i2"),
                         ("mvmin", "<i2", (2,)), ("mvmax", "<i2", (2,)), ("pmv", "<i2", (2, 2)), ("mvp", "<i2", (3, 2)),
                         ("have_mvd", "u1"), ("reserved", "u1", (3,))], align=True)
RK_MV_DTYPE = np.dtype([("mv", "<i2", (2,)), ("cost", "<i4")], align=True)
RK_RESULT_DTYPE = np.dtype([("cime", RK_MV_DTYPE), ("rime", RK_MV_DTYPE, (85,)), ("fme", RK_MV_DTYPE, (85,))], align=True)
assert RK_CTU_DTYPE.itemsize == 40 and RK_RESULT_DTYPE.itemsize == 171 * 8


BIPRED_JOB_DTYPE = np.dtype([("pu_x", "<i2"), ("pu_y", "<i2"), ("w", "u1"), ("h", "u1"), ("reserved", "u1", (2,)),
                             ("mv0", "<i2", (2,)), ("mv1", "<i2", (2,))], align=True)
MERGE_JOB_DTYPE = np.dtype([("pu_x", "<i2"), ("pu_y", "<i2"), ("size", "u1"), ("dir", "u1"), ("ref0", "u1"), ("ref1", "u1"),
                            ("mv0", "<i2", (2,)), ("mv1", "<i2", (2,))], align=True)
assert MERGE_JOB_DTYPE.itemsize == 16
MC_JOB_DTYPE = np.dtype([("pu_x", "<i2"), ("pu_y", "<i2"), ("w", "u1"), ("h", "u1"), ("dir", "u1"), ("flags", "u1"),
                         ("ref0", "u1"), ("ref1", "u1"), ("reserved", "u1", (2,)), ("mv0", "<i2", (2,)), ("mv1", "<i2", (2,))], align=True)
assert MC_JOB_DTYPE.itemsize == 20
MC_WEIGHT_DTYPE = np.dtype([("w", "<i2", (3,)), ("offset", "<i2", (3,)), ("log2denom", "u1", (3,)), ("reserved", "u1")], align=True)
assert MC_WEIGHT_DTYPE.itemsize == 16
BIPRED_RESULT_DTYPE = np.dtype([("satd", "<i4"), ("satd_zero", "<i4")], align=True)
assert BIPRED_JOB_DTYPE.itemsize == 16 and BIPRED_RESULT_DTYPE.itemsize == 8


class Plane(C.Structure):
    """x265amd_plane: a padded picture plane in device memory."""
    _fields_ = [("origin", C.c_void_p), ("stride", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("margin_x", C.c_int32), ("margin_y", C.c_int32)]


class Picture(C.Structure):
    """x265amd_picture: the three planes of a 4:2:0 picture in device memory."""
    _fields_ = [("y", Plane), ("cb", Plane), ("cr", Plane)]


class Pair(C.Structure):
    """x265amd_pair: one (source, reference) pair of a multi-pair search launch."""
    _fields_ = [("fenc", Plane), ("ref", Plane)]


class Field(C.Structure):
    _fields_ = [("fenc", Plane), ("ref", Plane * 4)]


class FrameCost(C.Structure):
    """x265amd_frame_cost: one estimateFrameCost evaluation; pointers are device addresses."""
    _fields_ = [("fenc", Plane), ("ref0", Plane * 4), ("ref1", Plane * 4), ("d0", C.c_int32), ("d1", C.c_int32),
                ("intra_calculated", C.c_int32), ("reserved", C.c_int32),
                ("mvs0", C.c_void_p), ("mv_costs0", C.c_void_p), ("mvs1", C.c_void_p), ("mv_costs1", C.c_void_p),
                ("inv_qscale", C.c_void_p), ("intra_cost", C.c_void_p), ("lowres_costs", C.c_void_p),
                ("row_satds_intra", C.c_void_p), ("row_satds", C.c_void_p), ("totals", C.c_void_p)]


class Weight(C.Structure):
    """x265amd_weight: outcome of weightsAnalyse."""
    _fields_ = [("weighted", C.c_int32), ("scale", C.c_int32), ("denom", C.c_int32), ("offset", C.c_int32),
                ("orig_score", C.c_uint32), ("min_score", C.c_uint32)]


ABI_VERSION = 2


class EngineError(RuntimeError):
    pass


def load_library():
    if not os.path.exists(LIB_PATH):
        raise EngineError("HIP engine library %s is missing: build it with __graft_entry__.build()" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    # this binding is written against X265AMD_ABI_VERSION 2 of include/x265_amd_me.h: a library with other argument
    # lists must not be called through it
    if not hasattr(lib, "x265amd_abi_version") or lib.x265amd_abi_version() != ABI_VERSION:
        raise EngineError("HIP engine library %s has another ABI version than this binding (%d): rebuild it with "
                          "__graft_entry__.build()" % (LIB_PATH, ABI_VERSION))
    return lib


_FN_TYPES = {
    "pixelcmp": C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t),
    "pixelcmp_x3": C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_void_p),
    "pixelcmp_x4": C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_void_p),
    "filter_pp": C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int),
    "filter_hps": C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int),
    "filter_sp": C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int),
    "pixelavg_pp": C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int),
    "downscale": C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_ssize_t,
                             C.c_int, C.c_int),
}


class Primitives(C.Structure):
    """x265amd_primitives: mirror of the ME slice of struct EncoderPrimitives (primitives.h:209-288)."""
    _fields_ = [("sad", _FN_TYPES["pixelcmp"] * 25), ("sad_x3", _FN_TYPES["pixelcmp_x3"] * 25),
                ("sad_x4", _FN_TYPES["pixelcmp_x4"] * 25), ("satd", _FN_TYPES["pixelcmp"] * 25),
                ("sa8d_inter", _FN_TYPES["pixelcmp"] * 25), ("luma_hpp", _FN_TYPES["filter_pp"] * 25),
                ("luma_hps", _FN_TYPES["filter_hps"] * 25), ("luma_vpp", _FN_TYPES["filter_pp"] * 25),
                ("luma_vsp", _FN_TYPES["filter_sp"] * 25), ("pixelavg_pp", _FN_TYPES["pixelavg_pp"] * 25),
                ("frame_init_lowres_core", _FN_TYPES["downscale"])]


def partition_from_sizes(w, h):
    return PARTITIONS.index((w, h))


class Engine:
    def __init__(self, device=0, bit_depth=8, high_priority=False):
        self.L = L = load_library()
        L.x265amd_last_error.restype = C.c_char_p
        L.x265amd_stream.restype = C.c_void_p
        self.depth = bit_depth
        self.dtype = np.uint8 if bit_depth <= 8 else np.uint16
        self.ctx = C.c_void_p()
        rc = L.x265amd_create_priority(device, bit_depth, int(high_priority), C.byref(self.ctx))
        if rc:
            raise EngineError("x265amd_create(device=%d) failed with %d (no MI355X visible?)" % (device, rc))

    def close(self):
        if self.ctx:
            self.L.x265amd_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def _chk(self, rc):
        if rc:
            raise EngineError("x265amd error %d: %s" % (rc, self.L.x265amd_last_error(self.ctx).decode()))

    # ---- planes
    def upload_plane(self, buf, width, height, margin_x, margin_y, stride):
        buf = np.ascontiguousarray(buf, self.dtype)
        assert buf.size >= (height + 2 * margin_y) * stride
        p = Plane()
        self._chk(self.L.x265amd_plane_upload(self.ctx, C.c_void_p(buf.ctypes.data), width, height, margin_x, margin_y,
                                              stride, C.byref(p)))
        return p

    def upload_image(self, img, margin=80):
        """Pad ``img`` (H x W) by edge replication and upload it."""
        from .synth import pad_plane
        buf, _, stride = pad_plane(np.asarray(img, self.dtype), margin)
        return self.upload_plane(buf, img.shape[1], img.shape[0], margin, margin, stride)

    def free_plane(self, p):
        self._chk(self.L.x265amd_plane_free(self.ctx, C.byref(p)))

    def download_plane(self, p):
        out = np.zeros(((p.height + 2 * p.margin_y), p.stride), self.dtype)
        self._chk(self.L.x265amd_plane_download(self.ctx, C.byref(p), C.c_void_p(out.ctypes.data)))
        return out

    def lowres_init(self, src):
        out = (Plane * 4)()
        self._chk(self.L.x265amd_lowres_init(self.ctx, C.byref(src), out))
        return [out[i] for i in range(4)]

    # ---- cost table
    def cost_table(self, qp):
        t = np.zeros(65535, np.uint16)
        self._chk(self.L.x265amd_cost_table(self.ctx, qp, C.c_void_p(t.ctypes.data)))
        return t

    # ---- search
    def search_batch(self, fenc, ref, jobs, qp, method, subme, is_lowres=False):
        jobs = np.ascontiguousarray(jobs, JOB_DTYPE)
        res = np.zeros(len(jobs), RESULT_DTYPE)
        if is_lowres:
            refs = (Plane * 4)(*ref)
        else:
            refs = (Plane * 1)(ref if isinstance(ref, Plane) else ref[0])
        self._chk(self.L.x265amd_search_batch(self.ctx, C.byref(fenc), refs, int(is_lowres), qp, method, subme,
                                              C.c_void_p(jobs.ctypes.data), len(jobs), C.c_void_p(res.ctypes.data)))
        return res

    def esa_ctu_batch(self, fenc, ref, jobs, nctu, qp, subme):
        """X265_FULL_SEARCH of whole CTUs through the scan that shares the 8x8 SADs between the quadtree depths:
        jobs = 85 * nctu records in the order workload.esa_ctu_jobs makes."""
        jobs = np.ascontiguousarray(jobs, JOB_DTYPE)
        assert len(jobs) == 85 * nctu
        res = np.zeros(len(jobs), RESULT_DTYPE)
        self._chk(self.L.x265amd_esa_ctu_batch(self.ctx, C.byref(fenc), C.byref(ref), qp, subme,
                                               C.c_void_p(jobs.ctypes.data), nctu, C.c_void_p(res.ctypes.data)))
        return res

    def esa_ctu_batch_dev(self, fenc, ref, d_jobs, nctu, d_results, qp, subme):
        self._chk(self.L.x265amd_esa_ctu_batch_dev(self.ctx, C.byref(fenc), C.byref(ref), qp, subme,
                                                   C.c_void_p(d_jobs), nctu, C.c_void_p(d_results)))

    def subpel_compare_batch(self, fenc, ref, jobs, cmp):
        jobs = np.ascontiguousarray(jobs, JOB_DTYPE)
        res = np.zeros(len(jobs), RESULT_DTYPE)
        self._chk(self.L.x265amd_subpel_compare_batch(self.ctx, C.byref(fenc), C.byref(ref), cmp,
                                                      C.c_void_p(jobs.ctypes.data), len(jobs), C.c_void_p(res.ctypes.data)))
        return res

    def lookahead_fields(self, fields, wcu, hcu, qp, method=HEX, subme=1, merange=16, want_stats=False):
        """fields: list of (fenc_plane0, [4 ref planes]). Returns (mvs[n, ncu, 2], costs[n, ncu], stats|None)."""
        n = len(fields)
        arr = (Field * max(1, n))()
        for i, (fe, refs) in enumerate(fields):
            arr[i].fenc = fe
            for k in range(4):
                arr[i].ref[k] = refs[k]
        mvs = np.zeros((n, wcu * hcu, 2), np.int16)
        costs = np.zeros((n, wcu * hcu), np.int32)
        stats = np.zeros((n, wcu * hcu), RESULT_DTYPE) if want_stats else None
        self._chk(self.L.x265amd_lookahead_fields(self.ctx, arr, n, wcu, hcu, qp, method, subme, merange,
                                                  C.c_void_p(mvs.ctypes.data), C.c_void_p(costs.ctypes.data),
                                                  C.c_void_p(stats.ctypes.data) if want_stats else None))
        return mvs, costs, stats

    def set_qp(self, qp):
        """BitCost::setQP: builds the table of `qp` on the device (the only step of a lookahead call that allocates)."""
        self._chk(self.L.x265amd_set_qp(self.ctx, qp))

    def lookahead_fields_static(self, fields, wcu, hcu, qp, method=HEX, subme=1, merange=16, want_stats=False):
        """lookahead_fields through the entry point that allocates nothing (static device scratch array); the
        cost table of `qp` must exist (set_qp)."""
        n = len(fields)
        arr = (Field * max(1, n))()
        for i, (fe, refs) in enumerate(fields):
            arr[i].fenc = fe
            for k in range(4):
                arr[i].ref[k] = refs[k]
        mvs = np.zeros((n, wcu * hcu, 2), np.int16)
        costs = np.zeros((n, wcu * hcu), np.int32)
        stats = np.zeros((n, wcu * hcu), RESULT_DTYPE) if want_stats else None
        self._chk(self.L.x265amd_lookahead_fields_static(self.ctx, arr, n, wcu, hcu, qp, method, subme, merange,
                                                         C.c_void_p(mvs.ctypes.data), C.c_void_p(costs.ctypes.data),
                                                         C.c_void_p(stats.ctypes.data) if want_stats else None))
        return mvs, costs, stats

    # ---- device-resident variants (pointers are raw device addresses, e.g. torch tensor.data_ptr())
    def stream(self):
        return self.L.x265amd_stream(self.ctx)

    def marked_count(self):
        """PUs of the last STAR launch that the fixed-shape kernels did not finish (x265amd_marked_count)."""
        n = C.c_int(0)
        self._chk(self.L.x265amd_marked_count(self.ctx, C.byref(n)))
        return n.value

    def sync(self):
        self._chk(self.L.x265amd_sync(self.ctx))

    def clock_probe_start(self, ms):
        """Queues the shader-clock probe on this engine's stream (give it an engine of its own next to the work)."""
        self._chk(self.L.x265amd_clock_probe_start(self.ctx, C.c_double(ms)))

    def clock_probe_read(self):
        """Average shader clock in MHz over the probe's time."""
        mhz = C.c_double(0)
        self._chk(self.L.x265amd_clock_probe_read(self.ctx, C.byref(mhz)))
        return mhz.value

    def set_esa_window_bytes(self, nbytes):
        """LDS one PU of an exhaustive (FULL) search may use for its reference window; tuning only."""
        self._chk(self.L.x265amd_set_esa_window_bytes(self.ctx, int(nbytes)))

    def search_batch_dev(self, fenc, ref, d_jobs, njobs, d_results, qp, method, subme, is_lowres=False, max_pu_dim=0):
        refs = (Plane * 4)(*ref) if is_lowres else (Plane * 1)(ref if isinstance(ref, Plane) else ref[0])
        self._chk(self.L.x265amd_search_batch_dev(self.ctx, C.byref(fenc), refs, int(is_lowres), qp, method, subme,
                                                  C.c_void_p(d_jobs), njobs, C.c_void_p(d_results), max_pu_dim))

    def check_jobs_dev(self, fenc, ref, d_jobs, njobs, method, is_lowres=False):
        """The host entry points' checks for a device-resident job list (which search_batch_dev runs as it is):
        raises EngineError naming the first job that must not run; returns nothing when all may."""
        refs = (Plane * 4)(*ref) if is_lowres else (Plane * 1)(ref if isinstance(ref, Plane) else ref[0])
        self._chk(self.L.x265amd_check_jobs_dev(self.ctx, C.byref(fenc), refs, int(is_lowres), method,
                                                C.c_void_p(d_jobs), njobs, None, None))

    def check_jobs_pairs_dev(self, pairs, d_jobs, njobs, method):
        """check_jobs_dev for a list that search_batch_pairs_dev runs: `pairs` is the list pairs_upload was given.
        Returns (number of jobs that must not run, index of the first or -1) instead of raising."""
        arr = (Pair * len(pairs))()
        for i, (f, r) in enumerate(pairs):
            arr[i].fenc, arr[i].ref = f, r
        n_bad, first = C.c_int(0), C.c_int(-1)
        rc = self.L.x265amd_check_jobs_pairs_dev(self.ctx, arr, len(pairs), method, C.c_void_p(d_jobs), njobs,
                                                 C.byref(n_bad), C.byref(first))
        if rc and not n_bad.value:
            self._chk(rc)
        return n_bad.value, first.value

    def pairs_upload(self, pairs):
        """pairs: list of (source Plane, reference Plane) -> device table for search_batch_pairs_dev."""
        arr = (Pair * len(pairs))()
        for i, (f, r) in enumerate(pairs):
            arr[i].fenc, arr[i].ref = f, r
        d = C.c_void_p()
        self._chk(self.L.x265amd_pairs_upload(self.ctx, arr, len(pairs), C.byref(d)))
        return d

    def search_batch_pairs(self, pairs, jobs, qp, method, subme):
        """Searches over several (source, reference) pairs in one launch; jobs["reserved"] names each job's pair."""
        jobs = np.ascontiguousarray(jobs, JOB_DTYPE)
        res = np.zeros(len(jobs), RESULT_DTYPE)
        arr = (Pair * max(1, len(pairs)))()
        for i, (f, r) in enumerate(pairs):
            arr[i].fenc, arr[i].ref = f, r
        self._chk(self.L.x265amd_search_batch_pairs(self.ctx, arr, len(pairs), qp, method, subme,
                                                    C.c_void_p(jobs.ctypes.data), len(jobs), C.c_void_p(res.ctypes.data)))
        return res

    def search_batch_pairs_dev(self, d_pairs, npairs, d_jobs, njobs, d_results, qp, method, subme, max_pu_dim=0):
        self._chk(self.L.x265amd_search_batch_pairs_dev(self.ctx, d_pairs, npairs, qp, method, subme,
                                                        C.c_void_p(d_jobs), njobs, C.c_void_p(d_results), max_pu_dim))

    def search_batch_pairs_family_dev(self, d_pairs, npairs, d_parents, nparents, d_parent_results, d_children, nchildren,
                                      d_child_results, qp, subme, method=STAR):
        """STAR (or HEX / DIA) search of a list of 16x16 PUs and the list of their 8x8 children (4 k + q = quadrant q of
        parent k) as one launch."""
        if method == STAR:
            self._chk(self.L.x265amd_search_batch_pairs_family_dev(self.ctx, d_pairs, npairs, qp, subme,
                                                                   C.c_void_p(d_parents), nparents, C.c_void_p(d_parent_results),
                                                                   C.c_void_p(d_children), nchildren, C.c_void_p(d_child_results)))
        else:
            self._chk(self.L.x265amd_search_batch_pairs_family_method_dev(self.ctx, d_pairs, npairs, qp, method, subme,
                                                                          C.c_void_p(d_parents), nparents, C.c_void_p(d_parent_results),
                                                                          C.c_void_p(d_children), nchildren, C.c_void_p(d_child_results)))

    def fields_upload(self, fields, wcu, hcu, method=HEX, merange=16):
        """Checks the planes against the CU grid / method / merange and puts the descriptors on the device."""
        n = len(fields)
        arr = (Field * n)()
        for i, (fe, refs) in enumerate(fields):
            arr[i].fenc = fe
            for k in range(4):
                arr[i].ref[k] = refs[k]
        d = C.c_void_p()
        self._chk(self.L.x265amd_fields_upload(self.ctx, arr, n, wcu, hcu, method, merange, C.byref(d)))
        return d

    def esa_ctu_batch_pairs_dev(self, d_pairs, npairs, d_jobs, nctu, d_results, qp, subme):
        self._chk(self.L.x265amd_esa_ctu_batch_pairs_dev(self.ctx, d_pairs, npairs, qp, subme,
                                                         C.c_void_p(d_jobs), nctu, C.c_void_p(d_results)))

    def pack_mv_cost_indexed(self, d_results, d_index, n, d_out):
        """out[i] = the MV + cost half of record d_index[i] (int32 indices on the device)."""
        self._chk(self.L.x265amd_pack_mv_cost_indexed(self.ctx, C.c_void_p(d_results), C.c_void_p(d_index), n, C.c_void_p(d_out)))

    def pack_mv_cost(self, d_results, n, d_out):
        """The MV + cost halves of n result records, packed (device pointers, asynchronous on the engine's stream)."""
        self._chk(self.L.x265amd_pack_mv_cost(self.ctx, C.c_void_p(d_results), n, C.c_void_p(d_out)))

    def device_free(self, d):
        self._chk(self.L.x265amd_device_free(self.ctx, C.c_void_p(d) if isinstance(d, int) else d))

    def lookahead_fields_dev(self, d_fields, nfields, wcu, hcu, qp, method, subme, merange, d_mvs, d_costs, d_stats=None):
        d_fields = C.c_void_p(d_fields) if isinstance(d_fields, int) else d_fields
        self._chk(self.L.x265amd_lookahead_fields_dev(self.ctx, d_fields, nfields, wcu, hcu, qp, method, subme, merange,
                                                      C.c_void_p(d_mvs), C.c_void_p(d_costs),
                                                      C.c_void_p(d_stats) if d_stats else None))

    # ---- lookahead frame cost
    def device_alloc(self, nbytes):
        d = C.c_void_p()
        self._chk(self.L.x265amd_device_alloc(self.ctx, C.c_size_t(nbytes), C.byref(d)))
        return d.value

    def mem_info(self):
        """(free, total) bytes of the context's GPU."""
        f, t = C.c_size_t(), C.c_size_t()
        self._chk(self.L.x265amd_device_mem_info(self.ctx, C.byref(f), C.byref(t)))
        return f.value, t.value

    def upload(self, d_dst, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(self.L.x265amd_upload(self.ctx, C.c_void_p(d_dst), C.c_void_p(arr.ctypes.data), C.c_size_t(arr.nbytes)))

    def download(self, d_src, shape, dtype):
        out = np.zeros(shape, dtype)
        self._chk(self.L.x265amd_download(self.ctx, C.c_void_p(d_src), C.c_void_p(out.ctypes.data), C.c_size_t(out.nbytes)))
        return out

    def frame_costs_upload(self, descs, wcu, hcu):
        arr = (FrameCost * len(descs))(*descs)
        d = C.c_void_p()
        self._chk(self.L.x265amd_frame_costs_upload(self.ctx, arr, len(descs), wcu, hcu, C.byref(d)))
        return d

    def lookahead_frame_costs_dev(self, d_frames, nframes, wcu, hcu):
        self._chk(self.L.x265amd_lookahead_frame_costs_dev(self.ctx, d_frames, nframes, wcu, hcu))

    def estimate_frame_costs(self, frames, wcu, hcu, qp, method=HEX, subme=1, merange=16):
        """CostEstimate::estimateFrameCost (slicetype.cpp:1132-1221) for a batch of (b, p0, p1) triples in three
        launches: the MV fields of every list that has none yet, then the intra / bidirectional / choice kernels.
        frames: dicts with fenc (4 lowres planes of frame b), ref0 / ref1 (4 planes of frames p0 / p1 or None),
        d0, d1, and optionally intra_cost (bIntraCalculated), inv_qscale, mvs0 / mv_costs0, mvs1 / mv_costs1 (lists
        searched by an earlier call). Returns dicts of host arrays named like the reference's Lowres members."""
        n, ncu = len(frames), wcu * hcu
        todo = [(i, l) for i, f in enumerate(frames) for l in (0, 1)
                if f["d%d" % l] > 0 and f.get("mvs%d" % l) is None]
        d_mvs = self.device_alloc(max(1, 2 * n) * ncu * 4)
        d_mvc = self.device_alloc(max(1, 2 * n) * ncu * 4)
        sizes = (("intra_cost", ncu * 4), ("lowres_costs", ncu * 2), ("row_satds_intra", hcu * 4), ("row_satds", hcu * 4),
                 ("totals", 40), ("inv_qscale", ncu * 4))
        d_out = {k: self.device_alloc(n * sz) for k, sz in sizes}
        owned = [d_mvs, d_mvc] + list(d_out.values())
        try:
            for i, f in enumerate(frames):
                for l in (0, 1):
                    if f["d%d" % l] > 0 and f.get("mvs%d" % l) is not None:
                        self.upload(d_mvs + (2 * i + l) * ncu * 4, np.ascontiguousarray(f["mvs%d" % l], np.int16))
                        self.upload(d_mvc + (2 * i + l) * ncu * 4, np.ascontiguousarray(f["mv_costs%d" % l], np.int32))
                if f.get("intra_cost") is not None:
                    self.upload(d_out["intra_cost"] + i * ncu * 4, np.ascontiguousarray(f["intra_cost"], np.int32))
                if f.get("inv_qscale") is not None:
                    self.upload(d_out["inv_qscale"] + i * ncu * 4, np.ascontiguousarray(f["inv_qscale"], np.int32))
            # the searches: one launch per run of fields whose outputs are adjacent
            k = 0
            while k < len(todo):
                m = k + 1
                while m < len(todo) and 2 * todo[m][0] + todo[m][1] == 2 * todo[m - 1][0] + todo[m - 1][1] + 1:
                    m += 1
                d_f = self.fields_upload([(frames[i]["fenc"][0], frames[i]["ref%d" % l]) for i, l in todo[k:m]], wcu, hcu, method, merange)
                owned.append(d_f)
                slot = 2 * todo[k][0] + todo[k][1]
                self.lookahead_fields_dev(d_f, m - k, wcu, hcu, qp, method, subme, merange,
                                          d_mvs + slot * ncu * 4, d_mvc + slot * ncu * 4)
                k = m
            descs = []
            for i, f in enumerate(frames):
                d = FrameCost()
                d.fenc = f["fenc"][0]
                for l in (0, 1):
                    if f["d%d" % l] > 0:
                        for q in range(4):
                            getattr(d, "ref%d" % l)[q] = f["ref%d" % l][q]
                        setattr(d, "mvs%d" % l, d_mvs + (2 * i + l) * ncu * 4)
                        setattr(d, "mv_costs%d" % l, d_mvc + (2 * i + l) * ncu * 4)
                d.d0, d.d1 = f["d0"], f["d1"]
                d.intra_calculated = int(f.get("intra_cost") is not None)
                d.inv_qscale = d_out["inv_qscale"] + i * ncu * 4 if f.get("inv_qscale") is not None else None
                for key, sz in sizes[:5]:
                    setattr(d, key, d_out[key] + i * sz)
                descs.append(d)
            d_desc = self.frame_costs_upload(descs, wcu, hcu)
            owned.append(d_desc)
            self.lookahead_frame_costs_dev(d_desc, n, wcu, hcu)
            mvs = self.download(d_mvs, (n, 2, ncu, 2), np.int16)
            mvc = self.download(d_mvc, (n, 2, ncu), np.int32)
            res = []
            for i in range(n):
                r = dict(mvs0=mvs[i, 0], mv_costs0=mvc[i, 0], mvs1=mvs[i, 1], mv_costs1=mvc[i, 1])
                r["intra_cost"] = self.download(d_out["intra_cost"] + i * ncu * 4, ncu, np.int32)
                r["lowres_costs"] = self.download(d_out["lowres_costs"] + i * ncu * 2, ncu, np.uint16)
                r["row_satds_intra"] = self.download(d_out["row_satds_intra"] + i * hcu * 4, hcu, np.int32)
                r["row_satds"] = self.download(d_out["row_satds"] + i * hcu * 4, hcu, np.int32)
                r["totals"] = self.download(d_out["totals"] + i * 40, 5, np.int64)
                res.append(r)
            return res
        finally:
            self.sync()
            for d in owned:
                self.device_free(d)

    # ---- explicit weighted prediction
    def weight_buffer(self, src, w0, rnd, shift, offset, out=None):
        out = out if out is not None else Plane()
        self._chk(self.L.x265amd_weight_buffer(self.ctx, C.byref(src), w0, rnd, shift, offset, C.byref(out)))
        return out

    def apply_weight(self, recon, input_weight, log2_denom, input_offset, out=None):
        out = out if out is not None else Plane()
        self._chk(self.L.x265amd_apply_weight(self.ctx, C.byref(recon), input_weight, log2_denom, input_offset, C.byref(out)))
        return out

    def lookahead_weight_cost(self, fenc, src, d_intra_cost, wcu, hcu, wp=None):
        cost = C.c_uint32(0)
        s, d, o = wp if wp else (0, 0, 0)
        self._chk(self.L.x265amd_lookahead_weight_cost(self.ctx, C.byref(fenc), C.byref(src), C.c_void_p(d_intra_cost), wcu, hcu,
                                                       int(wp is not None), s, d, o, C.byref(cost)))
        return cost.value

    def weights_analyse(self, fenc, ref4, d_intra_cost, wcu, hcu, stats, reuse=None):
        """Returns (Weight, [4 weighted planes] or None); `reuse`: four planes to write into instead of new ones."""
        st = (C.c_uint64 * 4)(*[int(v) for v in stats])
        out = Weight()
        planes = (Plane * 4)(*reuse) if reuse else (Plane * 4)()
        self._chk(self.L.x265amd_weights_analyse(self.ctx, C.byref(fenc), (Plane * 4)(*ref4), C.c_void_p(d_intra_cost), wcu, hcu,
                                                 st, C.byref(out), planes))
        return out, ([planes[i] for i in range(4)] if out.weighted else None)

    # ---- bi-prediction estimate
    def bipred_batch(self, fenc, ref0, ref1, jobs):
        jobs = np.ascontiguousarray(jobs, BIPRED_JOB_DTYPE)
        res = np.zeros(len(jobs), BIPRED_RESULT_DTYPE)
        self._chk(self.L.x265amd_bipred_batch(self.ctx, C.byref(fenc), C.byref(ref0), C.byref(ref1), C.c_void_p(jobs.ctypes.data),
                                              len(jobs), C.c_void_p(res.ctypes.data)))
        return res

    def bipred_batch_dev(self, fenc, ref0, ref1, d_jobs, njobs, d_results):
        self._chk(self.L.x265amd_bipred_batch_dev(self.ctx, C.byref(fenc), C.byref(ref0), C.byref(ref1), C.c_void_p(d_jobs), njobs,
                                                  C.c_void_p(d_results)))

    def bipred_list_create(self, fenc, ref0, ref1, jobs):
        jobs = np.ascontiguousarray(jobs, BIPRED_JOB_DTYPE)
        h = C.c_void_p()
        self._chk(self.L.x265amd_bipred_list_create(self.ctx, C.byref(fenc), C.byref(ref0), C.byref(ref1), C.c_void_p(jobs.ctypes.data),
                                                    len(jobs), C.byref(h)))
        return h

    def bipred_list_free(self, h):
        self._chk(self.L.x265amd_bipred_list_free(self.ctx, h))

    def bipred_batch_list(self, fenc, ref0, ref1, blist, d_out, d_results0=None, d_results1=None):
        """Launches a list asynchronously; d_results0 / d_results1: device search results that supply the MVs."""
        self._chk(self.L.x265amd_bipred_batch_list(self.ctx, C.byref(fenc), C.byref(ref0), C.byref(ref1), blist,
                                                   C.c_void_p(d_results0), C.c_void_p(d_results1), C.c_void_p(d_out)))

    # ---- merge-candidate estimate
    def merge_sa8d_batch(self, fenc, refs, jobs):
        jobs = np.ascontiguousarray(jobs, MERGE_JOB_DTYPE)
        out = np.zeros(len(jobs), np.int32)
        self._chk(self.L.x265amd_merge_sa8d_batch(self.ctx, C.byref(fenc), (Plane * len(refs))(*refs), len(refs),
                                                  C.c_void_p(jobs.ctypes.data), len(jobs), C.c_void_p(out.ctypes.data)))
        return out

    # ---- motion compensation
    def upload_picture(self, planes):
        """planes: [(padded buffer, origin, stride, width, height, margin)] for Y, Cb, Cr -> Picture."""
        pic = Picture()
        for name, (buf, _, stride, w, h, m) in zip(("y", "cb", "cr"), planes):
            setattr(pic, name, self.upload_plane(buf, w, h, m, m, stride))
        return pic

    def free_picture(self, pic):
        for p in (pic.y, pic.cb, pic.cr):
            self.free_plane(p)

    @staticmethod
    def _weights(weights, nrefs):
        if weights is None:
            return None, None
        w = np.ascontiguousarray(weights, MC_WEIGHT_DTYPE)
        assert len(w) == nrefs
        return w, C.c_void_p(w.ctypes.data)

    def mc_batch(self, refs, pred, jobs, weights=None):
        """TComPrediction::motionCompensation for MC_JOB_DTYPE records: refs = list of Picture, pred = output Picture,
        weights = one MC_WEIGHT_DTYPE record per reference (for jobs with flags bit 2)."""
        jobs = np.ascontiguousarray(jobs, MC_JOB_DTYPE)
        keep, wp = self._weights(weights, len(refs))
        self._chk(self.L.x265amd_mc_batch(self.ctx, (Picture * len(refs))(*refs), wp, len(refs), C.byref(pred),
                                          C.c_void_p(jobs.ctypes.data), len(jobs)))

    def pictures_upload(self, refs, weights=None):
        d = C.c_void_p()
        keep, wp = self._weights(weights, len(refs))
        self._chk(self.L.x265amd_pictures_upload(self.ctx, (Picture * len(refs))(*refs), wp, len(refs), C.byref(d)))
        return d.value

    def mc_list_create(self, refs, pred, jobs, weights=None):
        """Checks the jobs against the pictures, numbers their tiles and uploads them; returns the list handle."""
        jobs = np.ascontiguousarray(jobs, MC_JOB_DTYPE)
        h = C.c_void_p()
        keep, wp = self._weights(weights, len(refs))
        self._chk(self.L.x265amd_mc_list_create(self.ctx, (Picture * len(refs))(*refs), wp, len(refs), C.byref(pred),
                                                C.c_void_p(jobs.ctypes.data), len(jobs), C.byref(h)))
        return h

    def mc_list_free(self, h):
        self._chk(self.L.x265amd_mc_list_free(self.ctx, h))

    def mc_batch_dev(self, d_ref_planes, pred, mc_list, d_results0=None, d_results1=None):
        """Launches a list; d_results0 / d_results1: device arrays of search results whose MVs replace the jobs' list-0 /
        list-1 MVs (x265amd_mc_batch_dev_mv)."""
        self._chk(self.L.x265amd_mc_batch_dev_mv(self.ctx, C.c_void_p(d_ref_planes), C.byref(pred), mc_list,
                                                 C.c_void_p(d_results0), C.c_void_p(d_results1)))

    # ---- Rockchip path
    def rk_ds_plane(self, orig):
        out = Plane()
        self._chk(self.L.x265amd_rk_ds_plane(self.ctx, C.byref(orig), C.byref(out)))
        return out

    def rk_ctu_batch(self, fenc, recon, ds, qp, ctus):
        ctus = np.ascontiguousarray(ctus, RK_CTU_DTYPE)
        res = np.zeros(len(ctus), RK_RESULT_DTYPE)
        self._chk(self.L.x265amd_rk_ctu_batch(self.ctx, C.byref(fenc), C.byref(recon), C.byref(ds), qp,
                                              C.c_void_p(ctus.ctypes.data), len(ctus), C.c_void_p(res.ctypes.data)))
        return res

    def rk_ctu_batch_dev(self, fenc, recon, ds, qp, d_ctus, nctus, d_results):
        self._chk(self.L.x265amd_rk_ctu_batch_dev(self.ctx, C.byref(fenc), C.byref(recon), C.byref(ds), qp,
                                                  C.c_void_p(d_ctus), nctus, C.c_void_p(d_results)))

    def download_ds_plane(self, p):
        """int16 plane of x265amd_rk_ds_plane incl. margins, as (rows, stride)."""
        n = (p.height + 2 * p.margin_y) * p.stride
        out = np.zeros(n, np.int16)
        base = p.origin - (p.margin_y * p.stride + p.margin_x) * 2
        self._chk(self.L.x265amd_download(self.ctx, C.c_void_p(base), C.c_void_p(out.ctypes.data), C.c_size_t(n * 2)))
        return out.reshape(-1, p.stride)

    def primitives(self):
        p = Primitives()
        self._chk(self.L.x265amd_setup_primitives(self.ctx, C.byref(p)))
        return p


def make_jobs(n):
    return np.zeros(n, JOB_DTYPE)


class MotionEstimate:
    """Per-call mirror of the reference class (source/encoder/motion.h:35-120)."""

    COST_MAX = 1 << 28

    def __init__(self, engine):
        self.e = engine
        self.searchMethod = STAR      # motion.cpp:102
        self.subpelRefine = 5         # motion.cpp:103
        self.qp = 12
        self.fencplane = None
        self.pu = None

    def setSearchMethod(self, i):
        self.searchMethod = i

    def setSubpelRefine(self, i):
        self.subpelRefine = i

    def setQP(self, qp):
        self.qp = qp

    def setSourcePlane(self, plane):
        """plane: a device ``Plane`` (the reference passes pixel* + stride, motion.h:76)."""
        self.fencplane = plane

    def setSourcePU(self, pu_x, pu_y, pwidth, pheight):
        """The reference passes offset = pu_y * stride + pu_x (motion.cpp:116)."""
        assert (pwidth, pheight) in PARTITIONS[1:], "not a searchable luma partition"
        self.pu = (pu_x, pu_y, pwidth, pheight)

    def motionEstimate(self, ref, mvmin, mvmax, qmvp, numCandidates, mvc, merange, is_lowres=False):
        """Returns (cost, outQMv) -- motion.cpp:549."""
        j = make_jobs(1)
        j["pu_x"], j["pu_y"], j["w"], j["h"] = self.pu
        j["mvmin"][0] = mvmin
        j["mvmax"][0] = mvmax
        j["qmvp"][0] = qmvp
        j["numc"] = numCandidates
        for i in range(numCandidates):
            j["mvc"][0][i] = mvc[i]
        j["merange"] = merange
        r = self.e.search_batch(self.fencplane, ref, j, self.qp, self.searchMethod, self.subpelRefine, is_lowres)
        return int(r["cost"][0]), (int(r["mv"][0][0]), int(r["mv"][0][1]))

    def subpelCompare(self, ref, qmv, cmp):
        """cmp: 0 = sad, 1 = satd (the reference passes the function pointer, motion.cpp:1651)."""
        j = make_jobs(1)
        j["pu_x"], j["pu_y"], j["w"], j["h"] = self.pu
        j["qmvp"][0] = qmv
        return int(self.e.subpel_compare_batch(self.fencplane, ref, j, cmp)["cost"][0])
