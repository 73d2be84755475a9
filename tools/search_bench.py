#!/usr/bin/env python3
"""Per-PU-size timing of the pattern-search kernel on one 1080p frame pair: every square PU of one size in one
launch, with the LDS reservation (max_pu_dim) of that size and of larger classes, so the cost of sharing a
launch with bigger PUs is visible. HIP events on the engine's stream.

    python tools/search_bench.py [--method star --subme 5 --reps 10]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--method", default="star")
    ap.add_argument("--subme", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--lib", default=None, help="another build of the engine library, e.g. a -DX265AMD_STAGE_STOP=n timing build")
    args = ap.parse_args()
    import torch
    from x265_amd.engine import Engine, METHODS, RESULT_DTYPE
    from x265_amd.synth import make_clip, pad_plane, lowres_dims
    from x265_amd import workload
    from x265_amd.measure import _timed
    dev = torch.device("cuda", 0)
    e0 = Engine(0, 8)                       # the product build: makes the predictors (lookahead field)
    if args.lib:
        import x265_amd.engine as eng
        eng.LIB_PATH = os.path.abspath(args.lib)
    e = Engine(0, 8)
    W, H, M = args.width, args.height, 128
    clip = make_clip(W, H, 2, seed=1234)
    planes = []
    for f in clip:
        buf, org, stride = pad_plane(f, M)
        planes.append(e.upload_plane(buf, W, H, M, M, stride))
    low = [e0.lowres_init(p) for p in planes]
    lw, lh, wcu, hcu = lowres_dims(W, H)
    la_mvs, _, _ = e0.lookahead_fields([(low[1][0], low[0])], wcu, hcu, 12)
    for s in (8, 16, 32, 64):
        jobs = workload.square_pu_jobs(W, H, la_mvs[0], wcu, hcu, 57, sizes=(s,))
        n = len(jobs)
        d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(n, -1).copy()).to(dev)
        d_res = torch.zeros((n, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        out = []
        for dim in (8, 16, 32, 64):
            if dim < s:
                continue
            ms = _timed(torch, e, dev, lambda: e.search_batch_dev(planes[1], planes[0], d_jobs.data_ptr(), n, d_res.data_ptr(), 32,
                                                                  METHODS[args.method], args.subme, max_pu_dim=dim), args.reps)
            out.append("dim%d %.3f ms (%.1f ns/PU)" % (dim, ms, ms * 1e6 / n))
        res = d_res.cpu().numpy().view(RESULT_DTYPE).reshape(-1)
        px = workload.pixel_sads(jobs, res)
        print("%2dx%-2d %6d PUs  %s   sad_calls/PU %.1f subpel/PU %.1f" % (s, s, n, "  ".join(out), res["sad_calls"].mean(), res["subpel_calls"].mean()))


if __name__ == "__main__":
    main()
