#!/usr/bin/env python3
"""The headline's searches CTU row by CTU row (measure.ctu_row_wavefront): ms per 32 frame pairs and per row step.

    python tools/rows_bench.py [--pairs 32]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=32)
    ap.add_argument("--method", default="star")
    args = ap.parse_args()
    import torch
    from x265_amd.engine import Engine
    from x265_amd import measure
    dev = torch.device("cuda", 0)
    e, e2 = Engine(0, 8), Engine(0, 8)
    wf = measure.ctu_row_wavefront(e, e2, dev, pairs=args.pairs, method=args.method, reps=5)
    print("%d frame pairs, %s: %d CTU rows, %.3f ms in all, %.3f ms per row step, %d PUs, %.1f Gpixel-SAD/s"
          % (args.pairs, args.method, wf["rows"], wf["ms"], wf["ms"] / wf["rows"], wf["pus"], wf["pixel_sads"] / wf["ms"] / 1e6))


if __name__ == "__main__":
    main()
