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
LIB_PATH = os.path.join(HERE, "libx265_amd_me.so")

DIA, HEX, UMH, STAR, FULL = 0, 1, 2, 3, 4
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


class Plane(C.Structure):
    """x265amd_plane: a padded picture plane in device memory."""
    _fields_ = [("origin", C.c_void_p), ("stride", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("margin_x", C.c_int32), ("margin_y", C.c_int32)]


class Field(C.Structure):
    _fields_ = [("fenc", Plane), ("ref", Plane * 4)]


class EngineError(RuntimeError):
    pass


def load_library():
    if not os.path.exists(LIB_PATH):
        raise EngineError("HIP engine library %s is missing: build it with __graft_entry__.build()" % LIB_PATH)
    return C.CDLL(LIB_PATH)


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
    def __init__(self, device=0, bit_depth=8):
        self.L = L = load_library()
        L.x265amd_last_error.restype = C.c_char_p
        L.x265amd_stream.restype = C.c_void_p
        self.depth = bit_depth
        self.dtype = np.uint8 if bit_depth <= 8 else np.uint16
        self.ctx = C.c_void_p()
        rc = L.x265amd_create(device, bit_depth, C.byref(self.ctx))
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

    # ---- device-resident variants (pointers are raw device addresses, e.g. torch tensor.data_ptr())
    def stream(self):
        return self.L.x265amd_stream(self.ctx)

    def sync(self):
        self._chk(self.L.x265amd_sync(self.ctx))

    def set_esa_window_bytes(self, nbytes):
        """LDS one PU of an exhaustive (FULL) search may use for its reference window; tuning only."""
        self._chk(self.L.x265amd_set_esa_window_bytes(self.ctx, int(nbytes)))

    def search_batch_dev(self, fenc, ref, d_jobs, njobs, d_results, qp, method, subme, is_lowres=False, max_pu_dim=0):
        refs = (Plane * 4)(*ref) if is_lowres else (Plane * 1)(ref if isinstance(ref, Plane) else ref[0])
        self._chk(self.L.x265amd_search_batch_dev(self.ctx, C.byref(fenc), refs, int(is_lowres), qp, method, subme,
                                                  C.c_void_p(d_jobs), njobs, C.c_void_p(d_results), max_pu_dim))

    def fields_upload(self, fields):
        n = len(fields)
        arr = (Field * n)()
        for i, (fe, refs) in enumerate(fields):
            arr[i].fenc = fe
            for k in range(4):
                arr[i].ref[k] = refs[k]
        d = C.c_void_p()
        self._chk(self.L.x265amd_fields_upload(self.ctx, arr, n, C.byref(d)))
        return d

    def device_free(self, d):
        self._chk(self.L.x265amd_device_free(self.ctx, d))

    def lookahead_fields_dev(self, d_fields, nfields, wcu, hcu, qp, method, subme, merange, d_mvs, d_costs, d_stats=None):
        self._chk(self.L.x265amd_lookahead_fields_dev(self.ctx, d_fields, nfields, wcu, hcu, qp, method, subme, merange,
                                                      C.c_void_p(d_mvs), C.c_void_p(d_costs),
                                                      C.c_void_p(d_stats) if d_stats else None))

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
