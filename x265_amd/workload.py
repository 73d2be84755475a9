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
