#!/usr/bin/env python3
"""How often the PUs of a quadtree end on the same MV as their parent: one frame pair of the headline workload (1080p,
STAR merange 57), once with the refinement (subme 5) and once without (subme 0: the whole-sample MVs the refinement
starts from). A parent whose four children start from its own MV could take its half-pel candidates' distortions as
the sums of theirs.

    python tools/ctu_mv_agreement.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from x265_amd import workload
    from x265_amd.engine import Engine, METHODS
    from x265_amd.synth import make_clip, pad_plane, lowres_dims
    W, H, M = bench.W, bench.H, bench.MARGIN
    e = Engine(0, 8)
    clip = make_clip(W, H, 2, seed=1234)
    planes, low = [], []
    for f in clip:
        buf, org, stride = pad_plane(f, M)
        p = e.upload_plane(buf, W, H, M, M, stride)
        planes.append(p)
        low.append(e.lowres_init(p))
    lw, lh, wcu, hcu = lowres_dims(W, H)
    la_mvs, _, _ = e.lookahead_fields([(low[1][0], low[0])], wcu, hcu, bench.LA_QP, 1, 1, bench.LA_MERANGE)
    jobs = workload.square_pu_jobs(W, H, la_mvs[0], wcu, hcu, bench.MERANGE)
    for subme in (0, 5):
        res = e.search_batch(planes[1], planes[0], jobs, bench.QP, METHODS["star"], subme)
        mv = {}
        for s in (64, 32, 16, 8):
            sel = jobs["w"] == s
            nx, ny = W // s, H // s
            mv[s] = res["mv"][sel].reshape(ny, nx, 2)
        print("subme %d" % subme)
        for s in (64, 32, 16):
            c = mv[s // 2]
            ny, nx = mv[s].shape[:2]
            c = c[:2 * ny, :2 * nx].reshape(ny, 2, nx, 2, 2)
            same = (c == mv[s][:, None, :, None, :]).all(axis=(1, 3, 4))
            print("    %2dx%-2d PUs whose four children end on the parent's MV: %5.1f %%" % (s, s, 100.0 * same.mean()))


if __name__ == "__main__":
    main()
