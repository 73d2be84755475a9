"""Shared helpers for the parity tests: seeded job generators and oracle drivers."""
import numpy as np

from x265_amd.engine import JOB_DTYPE, PARTITIONS
from x265_amd.synth import make_clip, pad_plane, lowres_dims

MARGIN = 96
UMH_MARGIN = 320   # UMH reads far outside its window (pix_base / CROSS quirks of the reference)


def margin_for(method):
    return UMH_MARGIN if method == 2 else MARGIN


def clip_planes(width, height, depth, kind="texture", seed=7, frames=(2, 0), margin=MARGIN):
    """(fenc buffer, ref buffer, origin, stride): two frames of a synthetic clip, padded."""
    clip = make_clip(width, height, max(frames) + 1, seed=seed, depth=depth, kind=kind)
    fenc, org, stride = pad_plane(clip[frames[0]], margin)
    ref, _, _ = pad_plane(clip[frames[1]], margin)
    return fenc, ref, org, stride


def random_jobs(rng, n, width, height, method, parts=None, meranges=(4, 16, 24, 57), slack=60, max_numc=3):
    """Jobs shaped like the full-encode caller builds them: a window of +-merange around the
    rounded predictor, clipped to the picture plus `slack` pixels (TEncSearch.cpp:4805-4822)."""
    parts = parts or PARTITIONS[1:]
    jobs = np.zeros(n, JOB_DTYPE)
    for i in range(n):
        w, h = parts[int(rng.integers(0, len(parts)))]
        px = int(rng.integers(0, (width - w) // 4 + 1)) * 4
        py = int(rng.integers(0, (height - h) // 4 + 1)) * 4
        merange = int(rng.choice(meranges)) if method != 4 else int(rng.choice([3, 6, 9]))
        qmvp = (int(rng.integers(-60, 61)), int(rng.integers(-60, 61)))
        if i % 5 == 0:
            qmvp = (0, 0)
        elif i % 5 == 1:
            qmvp = (qmvp[0] & ~3, qmvp[1] & ~3)
        cx, cy = qmvp[0] >> 2, qmvp[1] >> 2
        j = jobs[i]
        j["pu_x"], j["pu_y"], j["w"], j["h"] = px, py, w, h
        j["mvmin"] = (max(cx - merange, -px - slack), max(cy - merange, -py - slack))
        j["mvmax"] = (min(cx + merange, width - px - w + slack), min(cy + merange, height - py - h + slack))
        j["qmvp"] = qmvp
        numc = int(rng.integers(0, max_numc + 1))
        j["numc"] = numc
        for k in range(numc):
            j["mvc"][k] = (int(rng.integers(-80, 81)), int(rng.integers(-80, 81)))
        j["merange"] = merange
    return jobs


def oracle_search(orc, depth, fenc, ref_planes, org, stride, jobs, qp, method, subme, is_lowres=False):
    """Run the CPU oracle over a job array; returns (mv[n,2], cost[n], sad_calls[n], subpel_calls[n])."""
    from oracle.bindings import MEStats
    ctx = orc.make_ctx(depth, fenc, org, stride, ref_planes, org, stride, is_lowres, qp, method, subme)
    n = len(jobs)
    mv = np.zeros((n, 2), np.int16)
    cost = np.zeros(n, np.int32)
    sc = np.zeros(n, np.uint32)
    qc = np.zeros(n, np.uint32)
    for i, j in enumerate(jobs):
        st = MEStats()
        off = int(j["pu_y"]) * stride + int(j["pu_x"])
        mvc = [tuple(int(v) for v in j["mvc"][k]) for k in range(int(j["numc"]))]
        m, c = orc.me_search(ctx, off, int(j["w"]), int(j["h"]), tuple(int(v) for v in j["mvmin"]),
                             tuple(int(v) for v in j["mvmax"]), tuple(int(v) for v in j["qmvp"]), mvc,
                             int(j["merange"]), st)
        mv[i] = m
        cost[i] = c
        sc[i] = st.sad_calls
        qc[i] = st.subpel_calls
    return mv, cost, sc, qc


def host_lowres_planes(orc, depth, img, margin=64):
    """Four padded lowres planes of `img` made with the oracle's downscaler (lowres.cpp:155-166)."""
    h, w = img.shape
    lw, lh, wcu, hcu = lowres_dims(w, h)
    src, so, ss = pad_plane(img, 24)   # the 2x filter of the CU-rounded size reads up to 18 px past the picture
    outs = orc.lowres_core(depth, src, so, ss, lw, lh)
    padded = [pad_plane(o, margin, stride_align=32) for o in outs]
    return [p[0] for p in padded], padded[0][1], padded[0][2], wcu, hcu, lw, lh


MC_W, MC_H, MC_MARGIN = 320, 192, 80


def mc_pictures(depth, seed, nrefs=2):
    """Reference pictures for the motion-compensation tests: a picture = [(buffer, origin, stride)] * 3 for Y, Cb, Cr,
    noise so that every filter tap and the clipping at both ends of the range matter."""
    from x265_amd.synth import make_chroma
    y = make_clip(MC_W, MC_H, nrefs, seed=seed, depth=depth, kind="noise")
    rng = np.random.default_rng(seed + 1)
    pics = []
    for i in range(nrefs):
        dt = y[i].dtype
        cb = rng.integers(0, 1 << depth, size=(MC_H // 2, MC_W // 2)).astype(dt)
        cr = rng.integers(0, 1 << depth, size=(MC_H // 2, MC_W // 2)).astype(dt)
        pics.append([pad_plane(y[i], MC_MARGIN), pad_plane(cb, MC_MARGIN // 2), pad_plane(cr, MC_MARGIN // 2)])
    return pics


def mc_blank_pred(depth, fill=0x5A):
    """An unpadded prediction picture filled with a sentinel (so writes outside a PU show up)."""
    dt = np.uint8 if depth <= 8 else np.uint16
    return [(np.full(MC_W * MC_H, fill, dt), 0, MC_W), (np.full(MC_W * MC_H // 4, fill, dt), 0, MC_W // 2),
            (np.full(MC_W * MC_H // 4, fill, dt), 0, MC_W // 2)]


def mc_weights(rng, dtype, nrefs):
    """Explicit weights per reference picture and component: denominators 0..7, weights on both sides of 1.0 (a few
    negative), offsets over the whole signed 8-bit range."""
    wts = np.zeros(nrefs, dtype)
    for r in wts:
        den = rng.integers(0, 8, 3)
        den[2] = den[1]
        r["log2denom"] = den
        r["w"] = [int(np.clip(round((1 << int(d)) * rng.uniform(0.4, 1.8)), -128, 127)) for d in den]
        if rng.random() < 0.25:
            r["w"][int(rng.integers(0, 3))] *= -1
        r["offset"] = rng.integers(-128, 128, 3)
    return wts


def mc_job_batches(rng, dtype, nbatches, nrefs=2, weighted=False):
    """Batches of non-overlapping PUs (one per 64x64 cell, at a random multiple of 4 inside it): every partition, the three
    directions, luma-only / chroma-only / both, and all 64 eighth-pel chroma fraction pairs many times over."""
    parts = PARTITIONS[1:]
    batches, k = [], 0
    for b in range(nbatches):
        jobs = np.zeros((MC_W // 64) * (MC_H // 64), dtype)
        for c, j in enumerate(jobs):
            w, h = parts[k % len(parts)]
            cx, cy = c % (MC_W // 64), c // (MC_W // 64)
            j["pu_x"] = cx * 64 + int(rng.integers(0, (64 - w) // 4 + 1)) * 4
            j["pu_y"] = cy * 64 + int(rng.integers(0, (64 - h) // 4 + 1)) * 4
            j["w"], j["h"] = w, h
            j["dir"] = 1 + k % 3
            j["flags"] = (3, 3, 3, 1, 2)[(k // 3) % 5] | (4 if weighted and k % 4 else 0)
            j["ref0"], j["ref1"] = k % nrefs, (k // nrefs) % nrefs
            mv0, mv1 = rng.integers(-150, 151, 2), rng.integers(-150, 151, 2)
            # walk the 8 x 8 fraction pairs of list 0; list 1 runs through them at another pace
            mv0 = ((mv0[0] & ~7) | (k & 7), (mv0[1] & ~7) | ((k >> 3) & 7))
            mv1 = ((mv1[0] & ~7) | ((k // 5) & 7), (mv1[1] & ~7) | ((k // 7) & 7))
            if k % 11 == 0:
                mv0, mv1 = (mv0[0] & ~7, mv0[1] & ~7), (0, 0)
            j["mv0"], j["mv1"] = mv0, mv1
            k += 1
        batches.append(jobs)
    return batches

