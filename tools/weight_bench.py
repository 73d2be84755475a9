#!/usr/bin/env python3
"""Streaming kernels of the weighted-prediction path against the HBM roofline: weight_pp over a whole padded
buffer (x265amd_weight_buffer) and applyWeight (weight the picture + extend its borders) on one plane.
Bytes = every sample read once and written once. HIP events on the engine's stream; the output plane is
allocated once and reused, as the reference reuses its weight buffer.

    python tools/weight_bench.py [--width 3840 --height 2160 --depth 8 --reps 20]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch
    from x265_amd.engine import Engine
    from x265_amd.synth import make_clip, pad_plane
    dev = torch.device("cuda", 0)
    e = Engine(0, args.depth)
    M = 304
    img = make_clip(args.width, args.height, 1, seed=2, depth=args.depth)[0]
    buf, org, stride = pad_plane(img, M)
    p = e.upload_plane(buf, args.width, args.height, M, M, stride)
    bpp = 1 if args.depth <= 8 else 2
    total = (args.height + 2 * M) * stride * bpp
    stream = torch.cuda.ExternalStream(e.stream(), device=dev)
    for name, fn, nbytes in (("weight_buffer", lambda o: e.weight_buffer(p, 50, 1 << (5 + 14 - args.depth), 6 + 14 - args.depth, 7, o), 2 * total),
                             ("apply_weight", lambda o: e.apply_weight(p, 50, 6, 7, o), args.width * args.height * bpp + total)):
        out = fn(None)
        e.sync()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        for _ in range(args.reps):
            fn(out)
        ev1.record(stream)
        e.sync()
        ms = ev0.elapsed_time(ev1) / args.reps
        e.free_plane(out)
        print("%dx%d %d-bit %-14s %.3f ms  %.0f GB/s (%.1f MB read + written)"
              % (args.width, args.height, args.depth, name, ms, nbytes / ms / 1e6, nbytes / 1e6))


if __name__ == "__main__":
    main()
