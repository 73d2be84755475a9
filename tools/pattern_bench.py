#!/usr/bin/env python3
"""Pattern search + sub-pel refinement of every square 64/32/16/8 PU of one picture against one reference, one launch
per PU size on one stream (x265_amd.measure.pattern_search_picture): BASELINE.json configs[3] (4K Main10, STAR, subme 5)
and the same picture at 8 bits.

    python tools/pattern_bench.py [--width 3840 --height 2160 --method star --subme 5]
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
    ap.add_argument("--method", default="star")
    ap.add_argument("--subme", type=int, default=5)
    args = ap.parse_args()
    import torch
    from x265_amd.engine import Engine
    from x265_amd import measure
    dev = torch.device("cuda", 0)
    for depth, qp in ((10, 38), (8, 32)):
        e = Engine(0, depth)
        m = measure.pattern_search_picture(e, dev, args.width, args.height, depth=depth, method=args.method, subme=args.subme, qp=qp)
        print("%2d-bit %dx%d %s subme %d: %d PU searches, %.3f ms per picture-reference, %.1f Gpixel-SAD/s" %
              (depth, args.width, args.height, args.method, args.subme, m["pus"], m["ms"], m["pixel_sads"] / m["ms"] / 1e6))
        if args.method in ("star", "hex", "dia"):
            e2 = Engine(0, depth)
            f = measure.pattern_search_picture_families(e, e2, dev, args.width, args.height, depth=depth, method=args.method, subme=args.subme, qp=qp)
            same = all(np.array_equal(f["results"][k], m["results"][k]) for k in ("mv", "cost", "sad_calls", "subpel_calls"))
            print("       the same as two launches side by side (16x16 with their 8x8 children; 64x64 with their 32x32 children): "
                  "%.3f ms per picture-reference, %.1f Gpixel-SAD/s, records %s" %
                  (f["ms"], f["pixel_sads"] / f["ms"] / 1e6, "equal" if same else "DIFFER"))
            e2.close()
        e.close()


if __name__ == "__main__":
    main()
