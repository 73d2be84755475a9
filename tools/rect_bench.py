#!/usr/bin/env python3
"""Pattern search + sub-pel refinement of every PU of one partition shape of one picture against one reference, a launch
per shape (2NxN / Nx2N and the asymmetric pairs beside the squares): ns per PU and per sample, so that the shapes the
fixed-shape kernels take can be compared with those the one-wave search takes.

    python tools/rect_bench.py [--width 1920 --height 1080 --method star --subme 5 --pairs 4]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--method", default="star")
    ap.add_argument("--subme", type=int, default=5)
    ap.add_argument("--merange", type=int, default=57)
    ap.add_argument("--check", type=int, default=0, help="PUs per shape to compare with the CPU oracle")
    ap.add_argument("--shapes", default="", help="only these shapes, e.g. 16x8,8x16")
    ap.add_argument("--pairs", type=int, default=1, help="launch the picture's PUs this many times over in one launch (throughput instead of latency)")
    args = ap.parse_args()
    import torch
    from x265_amd.engine import Engine, METHODS, RESULT_DTYPE, HEX
    from x265_amd import measure, workload
    from x265_amd.synth import make_clip, pad_plane, lowres_dims
    dev = torch.device("cuda", 0)
    e = Engine(0, 8)
    W, H, M = args.width, args.height, 128
    clip = make_clip(W, H, 2, seed=1234)
    planes, host = [], []
    for f in clip:
        buf, org, stride = pad_plane(f, M)
        planes.append(e.upload_plane(buf, W, H, M, M, stride))
        host.append(buf)
    low = [e.lowres_init(p) for p in planes]
    lw, lh, wcu, hcu = lowres_dims(W, H)
    la, _, _ = e.lookahead_fields([(low[1][0], low[0])], wcu, hcu, 12, HEX, 1, 16)
    d_pairs = e.pairs_upload([(planes[1], planes[0])] * args.pairs) if args.pairs > 1 else None
    shapes = [(64, 64, 64), (64, 32, 64), (32, 64, 64), (64, 16, 64), (64, 48, 64), (32, 32, 32), (32, 16, 32), (16, 32, 32), (32, 8, 32), (32, 24, 32),
              (16, 16, 16), (16, 8, 16), (8, 16, 16), (16, 4, 16), (16, 12, 16), (8, 8, 8), (8, 4, 8), (4, 8, 8)]
    if args.shapes:
        want = {tuple(int(v) for v in t.split("x")) for t in args.shapes.split(",")}
        shapes = [t for t in shapes if (t[0], t[1]) in want]
    for (w, h, cu) in shapes:
        jobs = workload.shape_jobs(W, H, w, h, la[0], wcu, hcu, args.merange, cu)
        if args.pairs > 1:
            reps = []
            for i in range(args.pairs):
                r = jobs.copy()
                r["reserved"] = i
                reps.append(r)
            jobs = np.concatenate(reps)
        n = len(jobs)
        d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(n, -1).copy()).to(dev)
        d_res = torch.zeros((n, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        dim = max(w, h)
        if args.pairs > 1:
            ms = measure._timed(torch, e, dev, lambda: e.search_batch_pairs_dev(d_pairs, args.pairs, d_jobs.data_ptr(), n, d_res.data_ptr(), 32,
                                                                                METHODS[args.method], args.subme, max_pu_dim=dim), 5)
        else:
            ms = measure._timed(torch, e, dev, lambda: e.search_batch_dev(planes[1], planes[0], d_jobs.data_ptr(), n, d_res.data_ptr(), 32,
                                                                          METHODS[args.method], args.subme, max_pu_dim=dim), 5)
        res = d_res.cpu().numpy().view(RESULT_DTYPE).reshape(-1)
        line = "%2dx%-2d %7d PUs %8.3f ms %8.1f ns/PU %7.3f ns/sample" % (w, h, n, ms, ms * 1e6 / n, ms * 1e6 / (n * w * h))
        if args.check:
            from oracle.bindings import Oracle   # checker only
            from util import oracle_search
            pick = np.linspace(0, n - 1, args.check).astype(int)
            mv, cost, sc, qc = oracle_search(Oracle(), 8, host[1], [host[0]], org, stride, jobs[pick], 32, METHODS[args.method], args.subme)
            ok = (np.array_equal(res["mv"][pick], mv) and np.array_equal(res["cost"][pick], cost)
                  and np.array_equal(res["sad_calls"][pick], sc) and np.array_equal(res["subpel_calls"][pick], qc))
            line += "  oracle %s" % ("OK" if ok else "MISMATCH")
        print(line, flush=True)


if __name__ == "__main__":
    main()
