#!/usr/bin/env python3
"""The headline's STAR searches on the three kinds of synthetic clip, class by class on one stream: time, rate and the share
of PUs the fixed-shape kernels hand on (marked).   python tools/clip_kinds.py [--pairs 8] [--kinds texture,noise,static]"""
import argparse
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=8)
ap.add_argument("--kinds", default="texture,noise,static")
ap.add_argument("--seed", type=int, default=1234)
args = ap.parse_args()
import torch
from x265_amd.engine import Engine
from x265_amd import measure
dev = torch.device("cuda", 0)
e = Engine(0, 8)
for kind in args.kinds.split(","):
    out, _ = measure.pattern_search_clip(e, dev, kind, pairs=args.pairs, seed=args.seed)
    ms = sum(v["ms"] for v in out.values())
    px = sum(v["pixel_sads"] for v in out.values())
    print("%-8s %2d pairs: %.3f ms, %.0f Gpixel-SAD/s" % (kind, args.pairs, ms, px / ms / 1e6), flush=True)
    for k, v in out.items():
        print("    %-6s %8d PUs  %7.3f ms  marked %.4f  %7.1f ns/PU  %.0f Gpixel-SAD/s" % (k, v["pus"], v["ms"], v["marked_frac"], v["ms"] * 1e6 / v["pus"], v["pixel_sads"] / v["ms"] / 1e6), flush=True)
