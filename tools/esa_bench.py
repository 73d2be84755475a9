#!/usr/bin/env python3
"""Exhaustive-search (--me full) throughput per PU size on one GPU: BASELINE.json configs[2]
(4K 8-bit, merange 92) by default. Prints one line per PU size with the launch time measured
by HIP events on the engine's stream, T pixel-SAD/s and the fraction of the byte-SAD issue
peak (tools/sad_rate.hip measures that peak: ~150 T pixel-SAD/s on MI355X).

    python tools/esa_bench.py [--width 3840 --height 2160 --merange 92 --sizes 64,32,16,8]
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
    ap.add_argument("--merange", type=int, default=92)
    ap.add_argument("--sizes", default="64,32,16,8")
    ap.add_argument("--subme", type=int, default=5)
    ap.add_argument("--depth", type=int, default=8, help="bits per sample; above 8 the peak is v_sad_u16's, half the byte-SAD one")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--window", type=int, default=0, help="LDS window bytes (0 = library default)")
    ap.add_argument("--check", type=int, default=0, help="PUs per size to compare with the CPU oracle")
    ap.add_argument("--lib", default="", help="engine library to load instead of the in-tree build (experiments)")
    ap.add_argument("--ctu", action="store_true", help="also time the CTU scan (all four sizes in one launch, 8x8 SADs shared) "
                                                       "and compare its records with the per-size launches'")
    ap.add_argument("--spread", type=int, default=12, help="the stand-in lookahead field's MVs lie in +-spread (lowres quarter-pels)")
    args = ap.parse_args()

    import torch
    if args.lib:
        import x265_amd.engine as eng
        eng.LIB_PATH = os.path.abspath(args.lib)
    from x265_amd.engine import Engine, METHODS
    from x265_amd import measure

    dev = torch.device("cuda", 0)
    e = Engine(0, args.depth)
    peak = measure.SAD_ISSUE_PEAK if args.depth <= 8 else measure.SAD_ISSUE_PEAK / 2
    if args.window:
        e.set_esa_window_bytes(args.window)
    rows, (cur, ref, org, stride) = measure.esa_per_size(e, dev, args.width, args.height, args.merange,
                                                         [int(v) for v in args.sizes.split(",")], args.subme, args.reps)
    total_px, total_ms = 0, 0.0
    for r in rows:
        total_px += r["pixel_sads"]
        total_ms += r["ms"]
        line = "%2dx%-2d  %6d PUs  %8.3f ms  %7.2f T pixel-SAD/s  %5.1f %% of SAD issue peak" % (
            r["size"], r["size"], r["pus"], r["ms"], r["pixel_sads"] / r["ms"] / 1e9,
            100.0 * r["pixel_sads"] / (r["ms"] * 1e-3) / peak)
        if args.check:
            from oracle.bindings import Oracle   # checker only
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from util import oracle_search
            pick = np.linspace(0, r["pus"] - 1, args.check).astype(int)
            mv, cost, sc, qc = oracle_search(Oracle(), args.depth, cur, [ref], org, stride, r["jobs"][pick], 32, METHODS["full"], args.subme)
            res = r["results"]
            ok = (np.array_equal(res["mv"][pick], mv) and np.array_equal(res["cost"][pick], cost)
                  and np.array_equal(res["sad_calls"][pick], sc))
            line += "  oracle %s" % ("OK" if ok else "MISMATCH")
        print(line, flush=True)
    print("all     %8.3f ms  %7.2f T pixel-SAD/s  %5.1f %% of SAD issue peak" % (
        total_ms, total_px / total_ms / 1e9, 100.0 * total_px / (total_ms * 1e-3) / peak))
    if args.ctu:
        for spread in sorted({args.spread, 0}):
            r = measure.esa_ctu_picture(e, dev, args.width, args.height, args.merange, args.subme, args.reps, spread=spread, compare=True)
            # the SADs the scan measures are the 8x8 level's; the count every PU is credited with is four times that
            print("CTU scan, field +-%-2d  %6d PUs in %d CTUs  %8.3f ms  %7.2f T pixel-SAD/s credited  records %s the per-size launches'" % (
                spread, r["pus"], r["ctus"], r["ms"], r["pixel_sads"] / r["ms"] / 1e9, "equal" if r["equal"] else "DIFFER FROM"), flush=True)


if __name__ == "__main__":
    main()
