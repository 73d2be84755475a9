"""Host-side job builders that reproduce what the reference's callers hand to
motionEstimate (the engine itself never decides predictors or ranges).

* ``square_pu_jobs``: every square PU of the CU quadtree depths 64/32/16/8 that lies
  inside the picture, predictor taken from a lookahead MV field, search range built
  the way xSetSearchRange does (TEncSearch.cpp:4805-4822): rounded predictor +-merange,
  clipped so the window stays inside the padded reference.
* ``algorithmic_bytes``: the compulsory bytes of one job with no inter-job reuse:
  bpp * [w*h + (w + 2Rx + 7)(h + 2Ry + 7)] + 8  (source block + reference window with
  the 8-tap apron + the result).
"""
import numpy as np

from .engine import JOB_DTYPE


def square_pu_jobs(width, height, lowres_mvs, wcu, hcu, merange, sizes=(64, 32, 16, 8), clip_margin=100):
    """lowres_mvs: (ncu, 2) int16 qpel MVs of the half-resolution field (8x8 lowres CU = 16x16 luma)."""
    mv_full = lowres_mvs.astype(np.int32).reshape(hcu, wcu, 2) * 2      # half-res qpel -> full-res qpel
    out = []
    for s in sizes:
        nx, ny = width // s, height // s
        xs, ys = np.meshgrid(np.arange(nx) * s, np.arange(ny) * s)
        xs, ys = xs.ravel(), ys.ravel()
        n = xs.size
        jobs = np.zeros(n, JOB_DTYPE)
        jobs["pu_x"], jobs["pu_y"], jobs["w"], jobs["h"] = xs, ys, s, s
        # predictor: lowres MV of the CU under the PU centre; candidates: left and above neighbours
        cx = np.minimum((xs + s // 2) // 16, wcu - 1)
        cy = np.minimum((ys + s // 2) // 16, hcu - 1)
        mvp = mv_full[cy, cx]
        left = mv_full[cy, np.maximum(cx - 1, 0)]
        above = mv_full[np.maximum(cy - 1, 0), cx]
        jobs["qmvp"] = mvp
        jobs["numc"] = 2
        jobs["mvc"][:, 0] = left
        jobs["mvc"][:, 1] = above
        fx, fy = mvp[:, 0] >> 2, mvp[:, 1] >> 2
        jobs["mvmin"][:, 0] = np.maximum(fx - merange, -xs - clip_margin)
        jobs["mvmin"][:, 1] = np.maximum(fy - merange, -ys - clip_margin)
        jobs["mvmax"][:, 0] = np.minimum(fx + merange, width - xs - s + clip_margin)
        jobs["mvmax"][:, 1] = np.minimum(fy + merange, height - ys - s + clip_margin)
        jobs["merange"] = merange
        out.append(jobs)
    return np.concatenate(out)


def family_order(jobs, parent=64):
    """Quadtree order of a job list of the square PUs of two adjacent sizes of one picture (64x64 with 32x32, or 16x16 with
    8x8): every parent followed by its four children in raster order (the library then refines them together with it
    where all five start from one MV); the children without a parent (a picture whose height is no multiple of the
    parent's size) at the end. Returns the permutation."""
    child = parent // 2
    pos = {(int(j["pu_x"]), int(j["pu_y"]), int(j["w"])): k for k, j in enumerate(jobs)}
    fam, used = [], set()
    for k, j in enumerate(jobs):
        if j["w"] != parent:
            continue
        kids = [pos.get((int(j["pu_x"]) + child * (q & 1), int(j["pu_y"]) + child * (q >> 1), child)) for q in range(4)]
        fam.append(k)
        if all(c is not None for c in kids):
            fam += kids
            used.update(kids)
    fam += [k for k, j in enumerate(jobs) if j["w"] == child and k not in used]
    fam = np.array(fam)
    assert len(fam) == len(jobs) and len(set(fam.tolist())) == len(jobs)
    return fam


def class_job_list(host_jobs, sizes, families=False):
    """One job list over several frame pairs for a launch of x265amd_search_batch_pairs_dev: the PUs of the given square
    sizes of every pair's list (host_jobs[i], as square_pu_jobs makes it), pair-major, job.reserved naming the pair;
    `families`: every pair's part in quadtree order (family_order). Returns (jobs, index) with index[k] = (pair, position
    in host_jobs[pair]) of job k."""
    parts, index = [], []
    for i, jobs in enumerate(host_jobs):
        idx = np.flatnonzero(np.isin(jobs["w"], sizes) & (jobs["w"] == jobs["h"]))
        j = jobs[idx].copy()
        j["reserved"] = i
        if families:
            order = family_order(j, max(sizes))
            j, idx = j[order], idx[order]
        parts.append(j)
        index.append(np.stack([np.full(len(idx), i), idx], axis=1))
    return np.concatenate(parts), np.concatenate(index)


def family_lists(host_jobs, parent=16):
    """The two job lists of x265amd_search_batch_pairs_family_dev over several frame pairs: the parents (square PUs of
    `parent` samples, pair-major, job.reserved naming the pair) and their children (half the size) with PU 4 k + q = quadrant q
    (raster order) of parent k; a parent whose children are not all in the list gets four placeholders' worth of nothing --
    its slots are filled with the list's first children that have no parent (so the lists stay dense), and what is left of
    those follows at the end. Returns (parents, parent index, children, child index), index[k] = (pair, position in
    host_jobs[pair]) of job k."""
    child = parent // 2
    pj, pi, fam_c, fam_i, orphans, orph_i = [], [], [], [], [], []
    for i, jobs in enumerate(host_jobs):
        sq = jobs["w"] == jobs["h"]
        pidx = np.flatnonzero(sq & (jobs["w"] == parent))
        cidx = np.flatnonzero(sq & (jobs["w"] == child))
        pos = {(int(jobs["pu_x"][k]), int(jobs["pu_y"][k])): int(k) for k in cidx}
        used = set()
        kids_of = []
        for k in pidx:
            x, y = int(jobs["pu_x"][k]), int(jobs["pu_y"][k])
            kids = [pos.get((x + child * (q & 1), y + child * (q >> 1))) for q in range(4)]
            kids_of.append(kids if all(c is not None for c in kids) else None)
            if kids_of[-1]:
                used.update(kids)
        free = [int(k) for k in cidx if int(k) not in used]
        for kids in kids_of:
            if kids is None:        # keep the 4 k + q correspondence of the parents behind: any four children without a parent
                kids, free = free[:4], free[4:]
                assert len(kids) == 4, "a parent without children and no spare children to fill its slots"
            fam_i.append(np.stack([np.full(4, i), np.array(kids)], axis=1))
            c = jobs[kids].copy()
            c["reserved"] = i
            fam_c.append(c)
        p = jobs[pidx].copy()
        p["reserved"] = i
        pj.append(p)
        pi.append(np.stack([np.full(len(pidx), i), pidx], axis=1))
        if free:
            o = jobs[free].copy()
            o["reserved"] = i
            orphans.append(o)
            orph_i.append(np.stack([np.full(len(free), i), np.array(free)], axis=1))
    return (np.concatenate(pj), np.concatenate(pi), np.concatenate(fam_c + orphans), np.concatenate(fam_i + orph_i))


def ctu_slot(p):
    """(x offset, y offset, size) of slot p of the 85 square PUs of a 64x64 CTU as x265amd_esa_ctu_batch orders them:
    0-63 the 8x8 PUs in Morton order, 64-79 the 16x16 PUs in Morton order, 80-83 the 32x32 PUs, 84 the 64x64 PU."""
    if p >= 84:
        k, s = 0, 64
    elif p >= 80:
        k, s = p - 80, 32
    elif p >= 64:
        k, s = p - 64, 16
    else:
        k, s = p, 8
    mx = (k & 1) | ((k >> 1) & 2) | ((k >> 2) & 4)
    my = ((k >> 1) & 1) | ((k >> 2) & 2) | ((k >> 3) & 4)
    return mx * s, my * s, s


def esa_ctu_jobs(jobs, width=0, height=0):
    """The job list of x265amd_esa_ctu_batch for a picture: `jobs` holds square PUs of 64 / 32 / 16 / 8 samples (as
    square_pu_jobs makes them); every 64x64 CTU that holds at least one of them gets its 85 slots, size-major (all
    CTUs' 8x8 slots, then the 16x16, 32x32, 64x64 ones), a slot without a PU in `jobs` with w = h = 0.
    Jobs of several (source, reference) pairs (job.reserved, x265amd_esa_ctu_batch_pairs_dev) form CTUs per pair.
    Returns (list, index, nctu) with index[k] = position in `jobs` of slot k, or -1."""
    pos = {(int(j["reserved"]), int(j["pu_x"]), int(j["pu_y"]), int(j["w"])): k for k, j in enumerate(jobs) if j["w"] == j["h"]}
    ctus = sorted({(g, y // 64, x // 64) for (g, x, y, s) in pos})
    n = len(ctus)
    first = (0, 64 * n, 80 * n, 84 * n)
    out = np.zeros(85 * n, JOB_DTYPE)
    index = np.full(85 * n, -1, np.int64)
    for c, (g, cy, cx) in enumerate(ctus):
        for p in range(85):
            ox, oy, s = ctu_slot(p)
            lvl = 0 if p < 64 else 1 if p < 80 else 2 if p < 84 else 3
            slot = first[lvl] + c * (64, 16, 4, 1)[lvl] + (p - (0, 64, 80, 84)[lvl])
            k = pos.get((g, cx * 64 + ox, cy * 64 + oy, s))
            if k is None:
                out[slot]["pu_x"], out[slot]["pu_y"], out[slot]["reserved"] = cx * 64 + ox, cy * 64 + oy, g
            else:
                out[slot] = jobs[k]
                index[slot] = k
    return out, index, n


def algorithmic_bytes(jobs, bytes_per_pixel=1):
    w = jobs["w"].astype(np.int64)
    h = jobs["h"].astype(np.int64)
    rx = (jobs["mvmax"][:, 0].astype(np.int64) - jobs["mvmin"][:, 0]) + 1
    ry = (jobs["mvmax"][:, 1].astype(np.int64) - jobs["mvmin"][:, 1]) + 1
    return int((bytes_per_pixel * (w * h + (w + rx - 1 + 7) * (h + ry - 1 + 7)) + 8).sum())


def pixel_sads(jobs, results):
    """Pixel-SADs actually evaluated: (full-pel + sub-pel candidates visited) x PU area."""
    area = jobs["w"].astype(np.int64) * jobs["h"].astype(np.int64)
    return int(((results["sad_calls"].astype(np.int64) + results["subpel_calls"].astype(np.int64)) * area).sum())


# ---------------------------------------------------------------- Rockchip path

def rk_merange(pic_w, pic_h, me_range_x=384, me_range_y=320):
    """Effective RK search range for a picture size (TEncSearch.cpp:3717-3742)."""
    mx = min(384, me_range_x // 4 * 4)
    my = min(320, me_range_y // 4 * 4)
    if pic_w < mx + 60 or pic_w <= 352:
        mx = min(pic_w, mx) // 4 * 2
    if pic_h < my + 60 or pic_h <= 288:
        my = min(pic_h, my) // 4 * 2
    return mx, my


def _tdiv(a, b):
    """C integer division (truncation toward zero)."""
    q = abs(a) // abs(b)
    return q if (a < 0) == (b < 0) else -q


def rk_search_range(pic_w, pic_h, cu_x, cu_y, merange_x, merange_y, search_range=208, ctu=64):
    """xSetSearchRange of the RK build with mvp = (0, 0) (TEncSearch.cpp:4188-4215, clipMv
    TComDataCU.cpp:3386-3410); note mvmax is derived from the already updated mvmin."""
    def clip(v, lo, hi):
        return min(hi, max(lo, v))
    xmax, xmin = (pic_w + search_range - cu_x) << 2, (-ctu - search_range - cu_x) << 2
    ymax, ymin = (pic_h + search_range - cu_y) << 2, (-ctu - search_range - cu_y) << 2
    dx, dy = (merange_x // 2) << 2, (merange_y // 2) << 2
    mnx, mny = clip(-dx, xmin, xmax) >> 2, clip(-dy, ymin, ymax) >> 2
    mxx, mxy = clip(dx - 4, xmin, xmax) >> 2, clip(dy - 4, ymin, ymax) >> 2
    mnx, mny, mxx, mxy = _tdiv(mnx, 4), _tdiv(mny, 4), _tdiv(mxx, 4), _tdiv(mxy, 4)
    mnx = -min(abs(mnx), mxx) * 4
    mny = -min(abs(mny), mxy) * 4
    mxx = (min(abs(mnx), mxx) - 1) * 4
    mxy = (min(abs(mny), mxy) - 1) * 4
    return (mnx, mny), (mxx, mxy)


def rk_cu_offsets():
    """(x offset, y offset, size) of the 85 CUs of a CTU in offsIdx order (OffsFromCtu64, hardwareC/inter.cpp:23-38):
    0-63 the 8x8 CUs in raster order, 64-79 16x16, 80-83 32x32, 84 the 64x64 CU."""
    out = []
    for size, n in ((8, 8), (16, 4), (32, 2), (64, 1)):
        out += [((i % n) * size, (i // n) * size, size) for i in range(n * n)]
    return out


def rk_ctu_jobs(pic_w, pic_h, pmvs, mvps, have_mvd=True, me_range=(384, 320), dtype=None):
    """One job per 64x64 CTU in raster order. pmvs: (nctu, 2, 2) even full-pel neighbour PMVs
    (RIME window centres 1 and 2); mvps: (nctu, 3, 2) cost predictors (TEncSearch.cpp:3788-3805)."""
    from .engine import RK_CTU_DTYPE
    ncx, ncy = (pic_w + 63) // 64, (pic_h + 63) // 64
    jobs = np.zeros(ncx * ncy, dtype or RK_CTU_DTYPE)
    mx, my = rk_merange(pic_w, pic_h, *me_range)
    for cy in range(ncy):
        for cx in range(ncx):
            j = jobs[cy * ncx + cx]
            j["ctu_x"], j["ctu_y"] = cx * 64, cy * 64
            j["valid_w"] = min(64, pic_w - cx * 64)
            j["valid_h"] = min(64, pic_h - cy * 64)
            j["mvmin"], j["mvmax"] = rk_search_range(pic_w, pic_h, cx * 64, cy * 64, mx, my)
            j["pmv"] = pmvs[cy * ncx + cx]
            j["mvp"] = mvps[cy * ncx + cx]
            j["have_mvd"] = int(have_mvd)
    return jobs


def shape_jobs(width, height, w, h, la, wcu, hcu, merange, cu):
    """Every PU of shape w x h of a picture's grid of `cu` x `cu` CUs (both PUs of a 2NxN / Nx2N split; the first PU of an
    asymmetric one), predictor and two candidates from the lookahead MV field `la` (the CU under the PU's centre, its left
    and upper neighbours), range `merange` around the predictor clipped 100 samples past the picture."""
    from .engine import JOB_DTYPE
    mv_full = la.astype(np.int32).reshape(hcu, wcu, 2) * 2
    xs, ys = np.meshgrid(np.arange(0, width - cu + 1, cu), np.arange(0, height - cu + 1, cu))
    xs, ys = xs.ravel(), ys.ravel()
    offs = [(0, 0)] if (w, h) == (cu, cu) else ([(0, 0), (0, h)] if w == cu and 2 * h == cu else [(0, 0), (w, 0)] if h == cu and 2 * w == cu else [(0, 0)])
    parts = []
    for ox, oy in offs:
        jobs = np.zeros(xs.size, JOB_DTYPE)
        jobs["pu_x"], jobs["pu_y"], jobs["w"], jobs["h"] = xs + ox, ys + oy, w, h
        cx = np.minimum((xs + ox + w // 2) // 16, wcu - 1)
        cy = np.minimum((ys + oy + h // 2) // 16, hcu - 1)
        mvp = mv_full[cy, cx]
        jobs["qmvp"] = mvp
        jobs["numc"] = 2
        jobs["mvc"][:, 0] = mv_full[cy, np.maximum(cx - 1, 0)]
        jobs["mvc"][:, 1] = mv_full[np.maximum(cy - 1, 0), cx]
        fx, fy = mvp[:, 0] >> 2, mvp[:, 1] >> 2
        jobs["mvmin"][:, 0] = np.maximum(fx - merange, -(xs + ox) - 100)
        jobs["mvmin"][:, 1] = np.maximum(fy - merange, -(ys + oy) - 100)
        jobs["mvmax"][:, 0] = np.minimum(fx + merange, width - (xs + ox) - w + 100)
        jobs["mvmax"][:, 1] = np.minimum(fy + merange, height - (ys + oy) - h + 100)
        jobs["merange"] = merange
        parts.append(jobs)
    return np.concatenate(parts)
