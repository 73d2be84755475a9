#!/usr/bin/env python3
"""The two estimates after the per-list searches, per PU size on one 1080p picture: the bi-prediction estimate
(x265amd_bipred_batch_dev: device-resident jobs, HIP events) and the merge-candidate estimate (x265amd_merge_sa8d_batch:
host job / result arrays, so its time includes the transfers; wall clock). Random quarter-pel MVs.

    python tools/bipred_bench.py [--width 1920 --height 1080 --depth 8 --reps 10]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    import torch
    from x265_amd.engine import Engine, BIPRED_JOB_DTYPE, BIPRED_RESULT_DTYPE, MERGE_JOB_DTYPE
    from x265_amd.synth import make_clip, pad_plane
    from x265_amd.measure import _timed
    dev = torch.device("cuda", 0)
    e = Engine(0, args.depth)
    W, H, M = args.width, args.height, 96
    bpp = 1 if args.depth <= 8 else 2
    clip = make_clip(W, H, 3, seed=4, depth=args.depth)
    planes = []
    for f in clip:
        buf, org, stride = pad_plane(f, M)
        planes.append(e.upload_plane(buf, W, H, M, M, stride))
    rng = np.random.default_rng(9)
    for size in (8, 16, 32, 64):
        xs, ys = np.meshgrid(np.arange(0, W - size + 1, size), np.arange(0, H - size + 1, size))
        n = xs.size
        jobs = np.zeros(n, BIPRED_JOB_DTYPE)
        jobs["pu_x"], jobs["pu_y"], jobs["w"], jobs["h"] = xs.ravel(), ys.ravel(), size, size
        jobs["mv0"], jobs["mv1"] = rng.integers(-64, 65, (n, 2)), rng.integers(-64, 65, (n, 2))
        d_jobs = e.device_alloc(jobs.nbytes)
        e.upload(d_jobs, jobs)
        d_res = e.device_alloc(n * BIPRED_RESULT_DTYPE.itemsize)
        ms = _timed(torch, e, dev, lambda: e.bipred_batch_dev(planes[1], planes[0], planes[2], d_jobs, n, d_res), args.reps)
        nbytes = n * (bpp * (size * size + 2 * (size + 7) ** 2) + 8)
        print("%dx%d %d-bit bi-prediction estimate %2dx%-2d: %6d PUs %.3f ms  %.2f ns per PU  %.0f GB/s algorithmic"
              % (W, H, args.depth, size, size, n, ms, ms * 1e6 / n, nbytes / ms / 1e6))
        e.device_free(d_jobs)
        e.device_free(d_res)
        e.bipred_batch(planes[1], planes[0], planes[2], jobs)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            e.bipred_batch(planes[1], planes[0], planes[2], jobs)
        msh = (time.perf_counter() - t0) * 1e3 / args.reps
        print("%dx%d %d-bit bi-prediction estimate %2dx%-2d, host job / result arrays (64 tiles per wave): %.3f ms incl. transfers  %.2f ns per PU"
              % (W, H, args.depth, size, size, msh, msh * 1e6 / n))
        mj = np.zeros(n, MERGE_JOB_DTYPE)
        mj["pu_x"], mj["pu_y"], mj["size"] = xs.ravel(), ys.ravel(), size
        mj["dir"], mj["ref0"], mj["ref1"] = 1 + np.arange(n) % 3, 0, 1
        mj["mv0"], mj["mv1"] = jobs["mv0"], jobs["mv1"]
        refs = [planes[0], planes[2]]
        e.merge_sa8d_batch(planes[1], refs, mj)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            e.merge_sa8d_batch(planes[1], refs, mj)
        ms = (time.perf_counter() - t0) * 1e3 / args.reps
        print("%dx%d %d-bit merge-candidate estimate %2dx%-2d: %6d candidates %.3f ms incl. transfers  %.2f ns per candidate"
              % (W, H, args.depth, size, size, n, ms, ms * 1e6 / n))


if __name__ == "__main__":
    main()
