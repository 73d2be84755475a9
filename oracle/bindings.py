"""TEST INFRASTRUCTURE ONLY: ctypes bindings of the two CPU checkers.

* ``Oracle``  -- oracle/libme_oracle.so, this repository's plain-C restatement
  (me_oracle.c); travels to the GPU box.
* ``Ref``     -- oracle/_ref/libx265ref{8,10}.so, the reference's own sources
  compiled by oracle/Makefile; exists only where /root/reference does.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

PARTITIONS = [(4, 4), (8, 8), (8, 4), (4, 8), (16, 16), (16, 8), (8, 16), (16, 12), (12, 16), (16, 4), (4, 16),
              (32, 32), (32, 16), (16, 32), (32, 24), (24, 32), (32, 8), (8, 32),
              (64, 64), (64, 32), (32, 64), (64, 48), (48, 64), (64, 16), (16, 64)]
METHODS = {"dia": 0, "hex": 1, "umh": 2, "star": 3, "full": 4}


def dtype_for(depth):
    return np.uint8 if depth <= 8 else np.uint16


def _p(arr, offset_elems=0):
    """void* to element ``offset_elems`` of a numpy array (flat index)."""
    return C.c_void_p(arr.ctypes.data + offset_elems * arr.itemsize)


class MV(C.Structure):
    _fields_ = [("x", C.c_int16), ("y", C.c_int16)]


class MECtx(C.Structure):
    _fields_ = [("depth", C.c_int), ("fenc_plane", C.c_void_p), ("fenc_stride", C.c_ssize_t),
                ("ref_planes", C.c_void_p * 4), ("ref_stride", C.c_ssize_t), ("is_lowres", C.c_int),
                ("cost", C.c_void_p), ("method", C.c_int), ("subme", C.c_int)]


class MEStats(C.Structure):
    _fields_ = [("sad_pixels", C.c_uint64), ("subpel_pixels", C.c_uint64),
                ("sad_calls", C.c_uint32), ("subpel_calls", C.c_uint32)]


RK_CTU_DTYPE = np.dtype([("ctu_x", "<i2"), ("ctu_y", "<i2"), ("valid_w", "<i2"), ("valid_h", "<i2"),
                         ("mvmin", "<i2", (2,)), ("mvmax", "<i2", (2,)), ("pmv", "<i2", (2, 2)), ("mvp", "<i2", (3, 2)),
                         ("have_mvd", "u1"), ("reserved", "u1", (3,))], align=True)
RK_MV_DTYPE = np.dtype([("mv", "<i2", (2,)), ("cost", "<i4")], align=True)
RK_RESULT_DTYPE = np.dtype([("cime", RK_MV_DTYPE), ("rime", RK_MV_DTYPE, (85,)), ("fme", RK_MV_DTYPE, (85,))], align=True)
assert RK_CTU_DTYPE.itemsize == 40 and RK_RESULT_DTYPE.itemsize == 171 * 8


BIPRED_JOB_DTYPE = np.dtype([("pu_x", "<i2"), ("pu_y", "<i2"), ("w", "u1"), ("h", "u1"), ("reserved", "u1", (2,)),
                             ("mv0", "<i2", (2,)), ("mv1", "<i2", (2,))], align=True)
assert BIPRED_JOB_DTYPE.itemsize == 16


MERGE_JOB_DTYPE = np.dtype([("pu_x", "<i2"), ("pu_y", "<i2"), ("size", "u1"), ("dir", "u1"), ("ref0", "u1"), ("ref1", "u1"),
                            ("mv0", "<i2", (2,)), ("mv1", "<i2", (2,))], align=True)
assert MERGE_JOB_DTYPE.itemsize == 16


MC_JOB_DTYPE = np.dtype([("pu_x", "<i2"), ("pu_y", "<i2"), ("w", "u1"), ("h", "u1"), ("dir", "u1"), ("flags", "u1"),
                         ("ref0", "u1"), ("ref1", "u1"), ("reserved", "u1", (2,)), ("mv0", "<i2", (2,)), ("mv1", "<i2", (2,))], align=True)
assert MC_JOB_DTYPE.itemsize == 20
MC_WEIGHT_DTYPE = np.dtype([("w", "<i2", (3,)), ("offset", "<i2", (3,)), ("log2denom", "u1", (3,)), ("reserved", "u1")], align=True)
assert MC_WEIGHT_DTYPE.itemsize == 16


def build_oracle():
    import subprocess
    subprocess.check_call(["make", "-C", HERE, "libme_oracle.so"], stdout=subprocess.DEVNULL)


class Oracle:
    """Plain-C restatement (oracle/me_oracle.c)."""

    def __init__(self):
        path = os.path.join(HERE, "libme_oracle.so")
        if not os.path.exists(path):
            build_oracle()
        L = self.L = C.CDLL(path)
        L.orc_me_search.restype = C.c_int
        L.orc_me_search.argtypes = [C.POINTER(MECtx), C.c_ssize_t, C.c_int, C.c_int, MV, MV, MV, C.c_int,
                                    C.c_void_p, C.c_int, C.POINTER(MV), C.c_void_p]
        L.orc_subpel_compare.argtypes = [C.POINTER(MECtx), C.c_ssize_t, C.c_int, C.c_int, MV, C.c_int]
        L.orc_lookahead_field.argtypes = [C.POINTER(MECtx), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        for n in ("orc_sad", "orc_satd", "orc_sa8d_inter"):
            getattr(L, n).argtypes = [C.c_int, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int]
        self._tables = {}

    def cost_table(self, qp):
        if qp not in self._tables:
            t = np.zeros(65535, np.uint16)
            self.L.orc_cost_table(C.c_int(qp), _p(t))
            self._tables[qp] = t
        return self._tables[qp]

    def sad(self, depth, a, ao, sa, b, bo, sb, w, h):
        return self.L.orc_sad(depth, _p(a, ao), sa, _p(b, bo), sb, w, h)

    def satd(self, depth, a, ao, sa, b, bo, sb, w, h):
        return self.L.orc_satd(depth, _p(a, ao), sa, _p(b, bo), sb, w, h)

    def sa8d_inter(self, depth, a, ao, sa, b, bo, sb, w, h):
        return self.L.orc_sa8d_inter(depth, _p(a, ao), sa, _p(b, bo), sb, w, h)

    def sad_x3(self, depth, fenc, r, offs, rs, w, h):
        res = (C.c_int32 * 3)()
        self.L.orc_sad_x3(depth, _p(fenc), _p(r, offs[0]), _p(r, offs[1]), _p(r, offs[2]), C.c_ssize_t(rs), w, h, res)
        return list(res)

    def sad_x4(self, depth, fenc, r, offs, rs, w, h):
        res = (C.c_int32 * 4)()
        self.L.orc_sad_x4(depth, _p(fenc), _p(r, offs[0]), _p(r, offs[1]), _p(r, offs[2]), _p(r, offs[3]),
                          C.c_ssize_t(rs), w, h, res)
        return list(res)

    def pixelavg(self, depth, a, ao, sa, b, bo, sb, w, h):
        dst = np.zeros((h, w), dtype_for(depth))
        self.L.orc_pixelavg_pp(depth, _p(dst), C.c_ssize_t(w), _p(a, ao), C.c_ssize_t(sa), _p(b, bo), C.c_ssize_t(sb), w, h)
        return dst

    def luma_hpp(self, depth, src, so, ss, ci, w, h):
        dst = np.zeros((h, w), dtype_for(depth))
        self.L.orc_luma_hpp(depth, _p(src, so), C.c_ssize_t(ss), _p(dst), C.c_ssize_t(w), ci, w, h)
        return dst

    def luma_vpp(self, depth, src, so, ss, ci, w, h):
        dst = np.zeros((h, w), dtype_for(depth))
        self.L.orc_luma_vpp(depth, _p(src, so), C.c_ssize_t(ss), _p(dst), C.c_ssize_t(w), ci, w, h)
        return dst

    def luma_hps(self, depth, src, so, ss, ci, ext, w, h):
        dst = np.zeros((h + (7 if ext else 0), w), np.int16)
        self.L.orc_luma_hps(depth, _p(src, so), C.c_ssize_t(ss), _p(dst), C.c_ssize_t(w), ci, ext, w, h)
        return dst

    def luma_vsp(self, depth, src16, so, ss, ci, w, h):
        dst = np.zeros((h, w), dtype_for(depth))
        self.L.orc_luma_vsp(depth, _p(src16, so), C.c_ssize_t(ss), _p(dst), C.c_ssize_t(w), ci, w, h)
        return dst

    def lowres_core(self, depth, src, so, ss, w, h):
        outs = [np.zeros((h, w), dtype_for(depth)) for _ in range(4)]
        self.L.orc_lowres_core(depth, _p(src, so), _p(outs[0]), _p(outs[1]), _p(outs[2]), _p(outs[3]),
                               C.c_ssize_t(ss), C.c_ssize_t(w), w, h)
        return outs

    def make_ctx(self, depth, fenc, fenc_origin, fenc_stride, refs, ref_origin, ref_stride, is_lowres, qp, method, subme):
        ctx = MECtx()
        ctx.depth = depth
        ctx.fenc_plane = fenc.ctypes.data + fenc_origin * fenc.itemsize
        ctx.fenc_stride = fenc_stride
        for i in range(4):
            r = refs[i] if i < len(refs) else refs[0]
            ctx.ref_planes[i] = r.ctypes.data + ref_origin * r.itemsize
        ctx.ref_stride = ref_stride
        ctx.is_lowres = int(is_lowres)
        ctx.cost = self.cost_table(qp).ctypes.data
        ctx.method = method
        ctx.subme = subme
        ctx._keep = (fenc, refs)
        return ctx

    def me_search(self, ctx, off, w, h, mvmin, mvmax, qmvp, mvc, merange, stats=None):
        out = MV()
        cand = (MV * max(1, len(mvc)))(*[MV(*m) for m in mvc])
        cost = self.L.orc_me_search(C.byref(ctx), off, w, h, MV(*mvmin), MV(*mvmax), MV(*qmvp), len(mvc),
                                    C.cast(cand, C.c_void_p), merange, C.byref(out),
                                    C.cast(C.pointer(stats), C.c_void_p) if stats is not None else None)
        return (out.x, out.y), cost

    def me_search_batch(self, ctx, jobs):
        """jobs: numpy array with the engine's job layout (40-byte records); returns 16-byte result records."""
        res = np.zeros(len(jobs), np.dtype([("mv", "<i2", (2,)), ("cost", "<i4"), ("sad_calls", "<u4"), ("subpel_calls", "<u4")]))
        jobs = np.ascontiguousarray(jobs)
        assert jobs.dtype.itemsize == 40
        self.L.orc_me_search_batch(C.byref(ctx), _p(jobs), len(jobs), _p(res))
        return res

    def rk_downsample(self, depth, src, origin, width, height, stride, margin4):
        """4x4-sum plane of the picture, edge-replicated by margin4 samples (reference.cpp:143-178)."""
        ds = np.zeros((height // 4, width // 4), np.int16)
        self.L.orc_rk_downsample(depth, _p(src, origin), width, height, C.c_ssize_t(stride), _p(ds), C.c_ssize_t(width // 4))
        padded = np.pad(ds, margin4, mode="edge")
        dstride = padded.shape[1]
        return np.ascontiguousarray(padded).reshape(-1), margin4 * dstride + margin4, dstride

    def rk_ctu_batch(self, ctx, ds, ds_origin, ds_stride, pic_w, pic_h, jobs):
        jobs = np.ascontiguousarray(jobs, RK_CTU_DTYPE)
        out = np.zeros(len(jobs), RK_RESULT_DTYPE)
        self.L.orc_rk_ctu_batch(C.byref(ctx), _p(ds, ds_origin), C.c_ssize_t(ds_stride), pic_w, pic_h, _p(jobs), len(jobs), _p(out))
        return out

    def subpel_compare(self, ctx, off, w, h, qmv, cmp):
        return self.L.orc_subpel_compare(C.byref(ctx), off, w, h, MV(*qmv), cmp)

    def lookahead_field(self, ctx, wcu, hcu, merange, stats=None):
        mvs = np.zeros((hcu * wcu, 2), np.int16)
        costs = np.zeros(hcu * wcu, np.int32)
        self.L.orc_lookahead_field(C.byref(ctx), wcu, hcu, merange, _p(mvs), _p(costs),
                                   C.cast(C.pointer(stats), C.c_void_p) if stats is not None else None)
        return mvs, costs

    def lookahead_frame_cost(self, depth, fenc, ref0, ref1, origin, stride, wcu, hcu, d0, d1, mvs0=None, mvcosts0=None,
                             mvs1=None, mvcosts1=None, intra_cost=None, inv_qscale=None):
        """Post-search part of estimateCUCost + the sums of estimateFrameCost. fenc: lowres full-pel plane of frame b;
        ref0 / ref1: the four hpel planes of frames p0 / p1 (None when the list is unused). intra_cost=None computes
        it (bIntraCalculated false). Returns a dict of the reference's output arrays."""
        ncu = wcu * hcu
        planes = lambda r: (C.c_void_p * 4)(*[p.ctypes.data + origin * p.itemsize for p in r]) if r is not None else None
        calc = intra_cost is None
        out = dict(intra_cost=np.zeros(ncu, np.int32) if calc else np.ascontiguousarray(intra_cost, np.int32).copy(),
                   lowres_costs=np.zeros(ncu, np.uint16), row_satds_intra=np.zeros(hcu, np.int32),
                   row_satds=np.zeros(hcu, np.int32), totals=np.zeros(5, np.int64))
        opt = lambda a, dt: _p(np.ascontiguousarray(a, dt)) if a is not None else None
        keep = [opt(inv_qscale, np.int32), opt(mvs0, np.int16), opt(mvcosts0, np.int32), opt(mvs1, np.int16), opt(mvcosts1, np.int32)]
        self.L.orc_lookahead_frame_cost(depth, _p(fenc, origin), planes(ref0), planes(ref1), C.c_ssize_t(stride), wcu, hcu, d0, d1,
                                        int(not calc), keep[0], keep[1], keep[2], keep[3], keep[4], _p(out["intra_cost"]),
                                        _p(out["lowres_costs"]), _p(out["row_satds_intra"]), _p(out["row_satds"]), _p(out["totals"]))
        return out

    def weight_pp(self, depth, src, so, dst, do, ss, ds, width, height, w0, rnd, shift, offset):
        self.L.orc_weight_pp(depth, _p(src, so), _p(dst, do), C.c_ssize_t(ss), C.c_ssize_t(ds), width, height, w0, rnd, shift, offset)

    def apply_weight(self, depth, src, origin, stride, width, height, mx, my, weight, denom, offset):
        dst = np.zeros_like(src)
        self.L.orc_apply_weight(depth, _p(src, origin), _p(dst, origin), C.c_ssize_t(stride), width, height, mx, my, weight, denom, offset)
        return dst

    def weight_cost_luma(self, depth, fenc, src, origin, stride, wcu, hcu, intra_cost, wp=None):
        ic = np.ascontiguousarray(intra_cost, np.int32)
        self.L.orc_weight_cost_luma.restype = C.c_uint32
        s, d, o = wp if wp else (0, 0, 0)
        return self.L.orc_weight_cost_luma(depth, _p(fenc, origin), _p(src, origin), C.c_ssize_t(stride), wcu, hcu, _p(ic),
                                           int(wp is not None), s, d, o)

    def weights_analyse(self, depth, fenc, ref, origin, stride, wcu, hcu, intra_cost, stats):
        """Returns (weighted, (scale, denom, offset), (origscore, minscore))."""
        ic = np.ascontiguousarray(intra_cost, np.int32)
        st = np.ascontiguousarray(stats, np.uint64)
        out = np.zeros(3, np.int32)
        sc = np.zeros(2, np.uint32)
        w = self.L.orc_weights_analyse(depth, _p(fenc, origin), _p(ref, origin), C.c_ssize_t(stride), wcu, hcu, _p(ic), _p(st), _p(out), _p(sc))
        return bool(w), tuple(int(v) for v in out), tuple(int(v) for v in sc)

    def bipred_batch(self, depth, fenc, ref0, ref1, origin, stride, jobs):
        """jobs: BIPRED_JOB_DTYPE records; returns int32 [n, 2] = (satd of the averaged predictions, of the zero-MV pair)."""
        jobs = np.ascontiguousarray(jobs, BIPRED_JOB_DTYPE)
        out = np.zeros((len(jobs), 2), np.int32)
        self.L.orc_bipred_batch(depth, _p(fenc, origin), C.c_ssize_t(stride), _p(ref0, origin), _p(ref1, origin), C.c_ssize_t(stride),
                                _p(jobs), len(jobs), _p(out))
        return out

    def merge_sa8d_batch(self, depth, fenc, refs, origin, stride, jobs):
        """jobs: MERGE_JOB_DTYPE records; refs: list of padded reference buffers (same geometry). Returns int32 [n]."""
        jobs = np.ascontiguousarray(jobs, MERGE_JOB_DTYPE)
        out = np.zeros(len(jobs), np.int32)
        ptrs = (C.c_void_p * len(refs))(*[r.ctypes.data + origin * r.itemsize for r in refs])
        strides = (C.c_ssize_t * len(refs))(*[stride] * len(refs))
        self.L.orc_merge_sa8d_batch(depth, _p(fenc, origin), C.c_ssize_t(stride), ptrs, strides, _p(jobs), len(jobs), _p(out))
        return out

    def mc_batch(self, depth, refs, pred, jobs, weights=None):
        """Motion compensation of MC_JOB_DTYPE records. refs: list of pictures, a picture = [(buf, origin, stride)] * 3 for
        Y, Cb, Cr (both chroma planes share a geometry); pred: one such picture, written in place; weights: one
        MC_WEIGHT_DTYPE record per reference picture, for jobs with flags bit 2."""
        jobs = np.ascontiguousarray(jobs, MC_JOB_DTYPE)
        wts = None if weights is None else np.ascontiguousarray(weights, MC_WEIGHT_DTYPE)
        planes = (C.c_void_p * (3 * len(refs)))(*[b.ctypes.data + o * b.itemsize for pic in refs for (b, o, _) in pic])
        strides = (C.c_ssize_t * (2 * len(refs)))(*[pic[k][2] for pic in refs for k in (0, 1)])
        pp = (C.c_void_p * 3)(*[b.ctypes.data + o * b.itemsize for (b, o, _) in pred])
        ps = (C.c_ssize_t * 2)(pred[0][2], pred[1][2])
        self.L.orc_mc_batch(depth, planes, strides, None if wts is None else _p(wts), pp, ps, _p(jobs), len(jobs))


def ref_available(depth=8):
    return os.path.exists(os.path.join(HERE, "_ref", "libx265ref%d.so" % (8 if depth <= 8 else 10)))


class Ref:
    """The reference's own code (oracle/_ref), one instance per bit depth."""

    def __init__(self, depth):
        self.depth = depth
        L = self.L = C.CDLL(os.path.join(HERE, "_ref", "libx265ref%d.so" % (8 if depth <= 8 else 10)))
        L.ref_setup(depth)
        L.ref_me_create.restype = C.c_void_p
        self.h = C.c_void_p(L.ref_me_create())

    def part(self, w, h):
        return PARTITIONS.index((w, h))

    def cost_table(self, qp):
        t = np.zeros(65535, np.uint16)
        self.L.ref_cost_table(qp, _p(t))
        return t

    def _cmp(self, name, a, ao, sa, b, bo, sb, w, h):
        return getattr(self.L, name)(self.part(w, h), _p(a, ao), C.c_ssize_t(sa), _p(b, bo), C.c_ssize_t(sb))

    def sad(self, a, ao, sa, b, bo, sb, w, h):
        return self._cmp("ref_sad", a, ao, sa, b, bo, sb, w, h)

    def satd(self, a, ao, sa, b, bo, sb, w, h):
        return self._cmp("ref_satd", a, ao, sa, b, bo, sb, w, h)

    def sa8d_inter(self, a, ao, sa, b, bo, sb, w, h):
        return self._cmp("ref_sa8d_inter", a, ao, sa, b, bo, sb, w, h)

    def sad_x3(self, fenc, r, offs, rs, w, h):
        res = (C.c_int32 * 3)()
        self.L.ref_sad_x3(self.part(w, h), _p(fenc), _p(r, offs[0]), _p(r, offs[1]), _p(r, offs[2]), C.c_ssize_t(rs), res)
        return list(res)

    def sad_x4(self, fenc, r, offs, rs, w, h):
        res = (C.c_int32 * 4)()
        self.L.ref_sad_x4(self.part(w, h), _p(fenc), _p(r, offs[0]), _p(r, offs[1]), _p(r, offs[2]), _p(r, offs[3]),
                          C.c_ssize_t(rs), res)
        return list(res)

    def pixelavg(self, a, ao, sa, b, bo, sb, w, h):
        dst = np.zeros((h, w), dtype_for(self.depth))
        self.L.ref_pixelavg_pp(self.part(w, h), _p(dst), C.c_ssize_t(w), _p(a, ao), C.c_ssize_t(sa), _p(b, bo), C.c_ssize_t(sb))
        return dst

    def luma_hpp(self, src, so, ss, ci, w, h):
        dst = np.zeros((h, w), dtype_for(self.depth))
        self.L.ref_luma_hpp(self.part(w, h), _p(src, so), C.c_ssize_t(ss), _p(dst), C.c_ssize_t(w), ci)
        return dst

    def luma_vpp(self, src, so, ss, ci, w, h):
        dst = np.zeros((h, w), dtype_for(self.depth))
        self.L.ref_luma_vpp(self.part(w, h), _p(src, so), C.c_ssize_t(ss), _p(dst), C.c_ssize_t(w), ci)
        return dst

    def luma_hps(self, src, so, ss, ci, ext, w, h):
        dst = np.zeros((h + (7 if ext else 0), w), np.int16)
        self.L.ref_luma_hps(self.part(w, h), _p(src, so), C.c_ssize_t(ss), _p(dst), C.c_ssize_t(w), ci, ext)
        return dst

    def luma_vsp(self, src16, so, ss, ci, w, h):
        dst = np.zeros((h, w), dtype_for(self.depth))
        self.L.ref_luma_vsp(self.part(w, h), _p(src16, so), C.c_ssize_t(ss), _p(dst), C.c_ssize_t(w), ci)
        return dst

    def lowres_core(self, src, so, ss, w, h):
        outs = [np.zeros((h, w), dtype_for(self.depth)) for _ in range(4)]
        self.L.ref_lowres_core(_p(src, so), _p(outs[0]), _p(outs[1]), _p(outs[2]), _p(outs[3]),
                               C.c_ssize_t(ss), C.c_ssize_t(w), w, h)
        return outs

    def me_config(self, fenc, fenc_origin, fenc_stride, qp, method, subme):
        self._keep = fenc
        self.L.ref_me_config(self.h, _p(fenc, fenc_origin), C.c_ssize_t(fenc_stride), qp, method, subme)

    def me_search(self, off, w, h, refs, ref_origin, ref_stride, is_lowres, mvmin, mvmax, qmvp, mvc, merange):
        planes = (C.c_void_p * 4)(*[refs[i if i < len(refs) else 0].ctypes.data + ref_origin * refs[0].itemsize
                                    for i in range(4)])
        i16 = lambda v: (C.c_int16 * 2)(*v)
        cand = (C.c_int16 * max(2, 2 * len(mvc)))(*[c for m in mvc for c in m])
        out = (C.c_int16 * 2)()
        cost = self.L.ref_me_search(self.h, off, w, h, planes, ref_stride, int(is_lowres), i16(mvmin), i16(mvmax),
                                    i16(qmvp), len(mvc), cand, merange, out)
        return (out[0], out[1]), cost

    def me_search_batch(self, plane, origin, stride, jobs):
        """All jobs (40-byte records) against one full-pel reference plane; me_config first. Returns 16-byte records."""
        res = np.zeros(len(jobs), np.dtype([("mv", "<i2", (2,)), ("cost", "<i4"), ("sad_calls", "<u4"), ("subpel_calls", "<u4")]))
        jobs = np.ascontiguousarray(jobs)
        assert jobs.dtype.itemsize == 40
        self.L.ref_me_search_batch(self.h, _p(plane, origin), stride, _p(jobs), len(jobs), _p(res))
        return res

    def subpel_compare(self, off, w, h, plane, origin, stride, qmv, cmp):
        return self.L.ref_me_subpel_compare(self.h, off, w, h, _p(plane, origin), stride, qmv[0], qmv[1], cmp)

    def rk_ds_plane(self, src, origin, width, height, stride, margin4):
        """MotionReference::downSampleAndSum + extendEdge run by the reference itself."""
        dstride = width // 4 + 2 * margin4
        buf = np.zeros((height // 4 + 2 * margin4) * dstride, np.int16)
        org = margin4 * dstride + margin4
        self.L.ref_rk_ds_plane(_p(src, origin), width, height, stride, _p(buf, org), dstride, margin4, margin4)
        return buf, org, dstride

    def rk_ctu(self, fenc, forigin, fstride, qp, recon, rorigin, rstride, ds, ds_origin, ds_stride, pic_w, pic_h, job,
               ref_idx=0):
        """All CUs of one CTU through the reference's RK overload, in the encoder's order
        (the isSavePmv CU first). Returns a record shaped like RK_RESULT_DTYPE."""
        out = np.zeros(1, RK_RESULT_DTYPE)[0]
        out["rime"]["cost"] = -1
        out["fme"]["cost"] = -1
        nb = np.zeros((10, 2), np.int16)
        nb[0], nb[1] = job["pmv"][0], job["pmv"][1]
        nb[7:10] = job["mvp"]
        cx, cy = int(job["ctu_x"]), int(job["ctu_y"])
        size_of = lambda i: 8 if i < 64 else 16 if i < 80 else 32 if i < 84 else 64
        def offs(i):
            if i < 64: return (i & 7) * 8, (i >> 3) * 8
            if i < 80: return ((i - 64) & 3) * 16, ((i - 64) >> 2) * 16
            if i < 84: return ((i - 80) & 1) * 32, ((i - 80) >> 1) * 32
            return 0, 0
        inside = [i for i in range(85) if cx + offs(i)[0] + size_of(i) <= pic_w and cy + offs(i)[1] + size_of(i) <= pic_h]
        # the save CU: the largest CU at the CTU origin that fits (TEncSearch.cpp:4007-4061)
        save = max((i for i in inside if offs(i) == (0, 0)), key=size_of)
        order = [save] + [i for i in sorted(inside, reverse=True) if i != save]
        res = (C.c_int32 * 12)()
        i16 = lambda v: (C.c_int16 * 2)(int(v[0]), int(v[1]))
        for n, i in enumerate(order):
            ox, oy = offs(i)
            size = size_of(i)
            depth = {64: 0, 32: 1, 16: 2, 8: 3}[size]
            self.L.ref_rk_me(self.h, _p(fenc, forigin), C.c_ssize_t(fstride), qp, _p(recon, rorigin), rstride, _p(ds, ds_origin),
                             (cy + oy) * fstride + cx + ox, size, i16(job["mvmin"]), i16(job["mvmax"]),
                             _p(nb), 2, int(job["have_mvd"]), int(n == 0), i, depth,
                             C.c_ssize_t(cx // 4 + (cy // 4) * ds_stride), ds_stride, int(job["valid_w"]), int(job["valid_h"]),
                             ref_idx, res)
            out["fme"][i]["mv"] = (res[6], res[7]); out["fme"][i]["cost"] = res[8]
            assert (res[0], res[1], res[2]) == (res[6], res[7], res[8])
            out["rime"][i]["mv"] = (res[3], res[4]); out["rime"][i]["cost"] = res[5]
            if n == 0:
                out["cime"]["mv"] = (res[9], res[10]); out["cime"]["cost"] = res[11]
        return out

    def cmodel_cime(self, cur, ref, pic_w, pic_h, qp, ctu_addr, sw_w, sw_h, neigh):
        """The hardware C-model's coarse search for one CTU (CoarseIntMotionEstimation::CIME). `cur` / `ref` are
        unpadded 8-bit pictures (2-D arrays); `neigh` the 28 neighbour MVs. Returns (mv, cost, mvp[3])."""
        assert self.depth == 8
        cur = np.ascontiguousarray(cur, np.uint8)
        ref = np.ascontiguousarray(ref, np.uint8)
        nb = np.ascontiguousarray(neigh, np.int16).reshape(28, 2)
        out = np.zeros(8, np.int16)
        cost = self.L.ref_cmodel_cime(_p(cur), cur.shape[1], _p(ref), ref.shape[1], pic_w, pic_h, qp, ctu_addr, sw_w, sw_h,
                                      _p(nb), _p(out))
        return (int(out[0]), int(out[1])), cost, out[2:].reshape(3, 2).copy()

    def cmodel_rime_fme(self, cur, ref, pic_w, pic_h, qp, ctu_addr, neigh, cime_mv):
        """The hardware C-model's refine + fractional stages for the 85 CUs of one CTU
        (RefineIntMotionEstimation::RimeAndFme). Returns (rime[85] {mv, cost}, fme[85] {mv, cost}, window
        centres pmv[3], cost predictors mvp[3])."""
        assert self.depth == 8
        cur = np.ascontiguousarray(cur, np.uint8)
        ref = np.ascontiguousarray(ref, np.uint8)
        nb = np.ascontiguousarray(neigh, np.int16).reshape(28, 2)
        cm = np.ascontiguousarray(cime_mv, np.int16)
        out = np.zeros(85 * 6 + 12, np.int32)
        self.L.ref_cmodel_rime_fme(_p(cur), cur.shape[1], _p(ref), ref.shape[1], pic_w, pic_h, qp, ctu_addr, _p(nb), _p(cm), _p(out))
        cu = out[:510].reshape(85, 6)
        return cu[:, 0:3].copy(), cu[:, 3:6].copy(), out[510:516].reshape(3, 2).copy(), out[516:522].reshape(3, 2).copy()

    def lookahead_field(self, fenc, refs, origin, stride, wcu, hcu, qp, method, subme, merange):
        planes = (C.c_void_p * 4)(*[r.ctypes.data + origin * r.itemsize for r in refs])
        mvs = np.zeros((hcu * wcu, 2), np.int16)
        costs = np.zeros(hcu * wcu, np.int32)
        self.L.ref_lookahead_field(_p(fenc, origin), planes, stride, wcu, hcu, qp, method, subme, merange, _p(mvs), _p(costs))
        return mvs, costs

    def estimate_frame(self, fenc4, ref0, ref1, origin, stride, wcu, hcu, d0, d1, intra_cost=None, inv_qscale=None,
                       mvs0=None, mvcosts0=None, mvs1=None, mvcosts1=None):
        """The reference's own EstimateRow::estimateCUCost over a whole frame (serial processRow order) and the row sums
        of estimateFrameCost. A list whose mvs are given is not searched again (bDoSearch false)."""
        ncu = wcu * hcu
        planes = lambda r: (C.c_void_p * 4)(*[p.ctypes.data + origin * p.itemsize for p in r]) if r is not None else None
        calc = intra_cost is None
        own = lambda a, dt, shape: np.zeros(shape, dt) if a is None else np.ascontiguousarray(a, dt).copy()
        out = dict(mvs0=own(mvs0, np.int16, (ncu, 2)), mvcosts0=own(mvcosts0, np.int32, ncu),
                   mvs1=own(mvs1, np.int16, (ncu, 2)), mvcosts1=own(mvcosts1, np.int32, ncu),
                   intra_cost=own(intra_cost, np.int32, ncu), lowres_costs=np.zeros(ncu, np.uint16),
                   row_satds_intra=np.zeros(hcu, np.int32), row_satds=np.zeros(hcu, np.int32), totals=np.zeros(5, np.int64))
        q = np.ascontiguousarray(inv_qscale, np.int32) if inv_qscale is not None else None
        self.L.ref_estimate_frame(planes(fenc4), planes(ref0), planes(ref1), stride, wcu, hcu, d0, d1,
                                  int(d0 > 0 and mvs0 is None), int(d1 > 0 and mvs1 is None), int(not calc),
                                  _p(q) if q is not None else None,
                                  _p(out["mvs0"]), _p(out["mvcosts0"]), _p(out["mvs1"]), _p(out["mvcosts1"]),
                                  _p(out["intra_cost"]), _p(out["lowres_costs"]), _p(out["row_satds_intra"]),
                                  _p(out["row_satds"]), _p(out["totals"]))
        return out

    def weight_pp(self, src, so, dst, do, ss, ds, width, height, w0, rnd, shift, offset):
        self.L.ref_weight_pp(_p(src, so), _p(dst, do), C.c_ssize_t(ss), C.c_ssize_t(ds), width, height, w0, rnd, shift, offset)

    def apply_weight(self, buf, width, height, mx, my, stride, weight, denom, offset):
        """MotionReference::init(pic, wp) + applyWeight row by row, by the reference itself; the weighted padded plane."""
        out = np.zeros_like(buf)
        assert self.L.ref_apply_weight(_p(buf), width, height, mx, my, stride, weight, denom, offset, _p(out)) == 0
        return out

    def weight_cost_luma(self, fenc, src, origin, stride, wcu, hcu, intra_cost, wp=None):
        ic = np.ascontiguousarray(intra_cost, np.int32)
        scratch = np.zeros(stride * hcu * 8 + 64, fenc.dtype)
        self.L.ref_weight_cost_luma.restype = C.c_uint32
        s, d, o = wp if wp else (0, 0, 0)
        return self.L.ref_weight_cost_luma(_p(fenc, origin), _p(src, origin), stride, wcu, hcu, _p(ic), int(wp is not None), s, d, o,
                                           _p(scratch))

    def weights_analyse(self, fenc4, ref4, origin, stride, padded_lines, wcu, hcu, intra_cost, stats, delta_index=1):
        """CostEstimate::weightsAnalyse by the reference itself. ref4: the four padded lowres buffers of the reference
        (flat arrays whose pixel (0,0) is at `origin`). Returns (weighted, (scale, denom, offset), weighted buffers, delta)."""
        ic = np.ascontiguousarray(intra_cost, np.int32)
        st = np.ascontiguousarray(stats, np.uint64)
        pl = lambda r: (C.c_void_p * 4)(*[p.ctypes.data + origin * p.itemsize for p in r])
        bufs = (C.c_void_p * 4)(*[p.ctypes.data for p in ref4])
        wbuf = np.zeros(4 * stride * padded_lines, ref4[0].dtype)
        out = np.zeros(4, np.int32)
        delta = C.c_double(0)
        self.L.ref_weights_analyse(pl(fenc4), pl(ref4), bufs, stride, padded_lines, origin, wcu, hcu, _p(ic), _p(st), delta_index,
                                   _p(wbuf), _p(out), C.byref(delta))
        return bool(out[0]), (int(out[1]), int(out[2]), int(out[3])), wbuf.reshape(4, -1), delta.value

    def bipred_batch(self, fenc, ref0, ref1, origin, stride, jobs):
        out = np.zeros((len(jobs), 2), np.int32)
        for i, j in enumerate(jobs):
            off = origin + int(j["pu_y"]) * stride + int(j["pu_x"])
            mv0 = (C.c_int16 * 2)(int(j["mv0"][0]), int(j["mv0"][1]))
            mv1 = (C.c_int16 * 2)(int(j["mv1"][0]), int(j["mv1"][1]))
            r = (C.c_int32 * 2)()
            self.L.ref_bipred_cost(_p(fenc, off), C.c_ssize_t(stride), _p(ref0, off), _p(ref1, off), C.c_ssize_t(stride),
                                   int(j["w"]), int(j["h"]), mv0, mv1, r)
            out[i] = (r[0], r[1])
        return out

    def merge_sa8d_batch(self, fenc, refs, origin, stride, jobs):
        out = np.zeros(len(jobs), np.int32)
        for i, j in enumerate(jobs):
            off = origin + int(j["pu_y"]) * stride + int(j["pu_x"])
            mv0 = (C.c_int16 * 2)(int(j["mv0"][0]), int(j["mv0"][1]))
            mv1 = (C.c_int16 * 2)(int(j["mv1"][0]), int(j["mv1"][1]))
            out[i] = self.L.ref_merge_sa8d(_p(fenc, off), C.c_ssize_t(stride), _p(refs[int(j["ref0"])], off), C.c_ssize_t(stride),
                                           _p(refs[int(j["ref1"])], off), C.c_ssize_t(stride), int(j["size"]), int(j["dir"]), mv0, mv1)
        return out

    def mc_batch(self, refs, pred, jobs, weights=None):
        """TComPrediction::motionCompensation's primitive calls per PU (ref_mc_pu); arguments as Oracle.mc_batch."""
        wts = None if weights is None else np.ascontiguousarray(weights, MC_WEIGHT_DTYPE)
        wp = lambda r: None if wts is None else _p(wts, r)
        def at(pic, px, py):
            ptrs = [pic[0][0].ctypes.data + (pic[0][1] + py * pic[0][2] + px) * pic[0][0].itemsize]
            for k in (1, 2):
                ptrs.append(pic[k][0].ctypes.data + (pic[k][1] + (py >> 1) * pic[k][2] + (px >> 1)) * pic[k][0].itemsize)
            return (C.c_void_p * 3)(*ptrs), (C.c_ssize_t * 2)(pic[0][2], pic[1][2])
        for j in jobs:
            px, py = int(j["pu_x"]), int(j["pu_y"])
            p0, s0 = at(refs[int(j["ref0"])], px, py)
            p1, s1 = at(refs[int(j["ref1"])], px, py)
            pd, sd = at(pred, px, py)
            mv0 = (C.c_int16 * 2)(int(j["mv0"][0]), int(j["mv0"][1]))
            mv1 = (C.c_int16 * 2)(int(j["mv1"][0]), int(j["mv1"][1]))
            self.L.ref_mc_pu(p0, s0, p1, s1, pd, sd, int(j["w"]), int(j["h"]), int(j["dir"]), int(j["flags"]), mv0, mv1,
                             wp(int(j["ref0"])), wp(int(j["ref1"])))
