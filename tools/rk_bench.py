#!/usr/bin/env python3
"""Rockchip path (rd >= 5: CIME + RIME + FME) throughput on one GPU: every CTU of a picture against
one reference, default RK search range (meRangeX/Y 384/320). Prints the time per picture measured
by HIP events on the engine's stream, CTUs/s and pixel-SADs/s (CIME positions x 256 samples +
RIME 576 x CU area + FME 49 x CU area, summed over the 85 CUs of a CTU).

    python tools/rk_bench.py [--width 1920 --height 1080 --reps 5 --check 4]
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
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--check", type=int, default=0, help="CTUs to compare with the CPU oracle")
    ap.add_argument("--depth", type=int, default=8, help="bits per sample (8 or 10)")
    args = ap.parse_args()

    import torch
    from x265_amd.engine import Engine
    from x265_amd import measure

    e = Engine(0, args.depth)
    m = measure.rk_picture(e, torch.device("cuda", 0), args.width, args.height, args.reps, qp=30 + 6 * (args.depth - 8))
    line = "%dx%d  %d CTUs  %.3f ms per picture-reference  %.1f kCTU/s  %.2f T pixel-SAD/s (CIME %.2f G, RIME+FME %.2f G per picture)" % (
        args.width, args.height, m["ctus"], m["ms"], m["ctus"] / m["ms"], m["pixel_sads"] / m["ms"] / 1e9,
        m["cime_pixel_sads"] / 1e9, m["cu_pixel_sads"] / 1e9)
    if args.check:
        from oracle.bindings import Oracle   # checker only
        orc = Oracle()
        cur, rorig, recon, org, stride = m["planes"]
        ds, dorg, dstride = orc.rk_downsample(args.depth, rorig, org, args.width, args.height, stride, m["margin"] // 4)
        ctx = orc.make_ctx(args.depth, cur, org, stride, [recon], org, stride, False, m["qp"], 0, 0)
        pick = np.linspace(0, m["ctus"] - 1, args.check).astype(int)
        t0 = time.perf_counter()
        exp = orc.rk_ctu_batch(ctx, ds, dorg, dstride, args.width, args.height, m["jobs"][pick])
        dt = time.perf_counter() - t0
        res = m["results"]
        ok = all(np.array_equal(res[pick][k]["mv"], exp[k]["mv"]) and np.array_equal(res[pick][k]["cost"], exp[k]["cost"])
                 for k in ("cime", "rime", "fme"))
        line += "  oracle %s (%.1f ms per CTU on one host core)" % ("OK" if ok else "MISMATCH", dt / len(pick) * 1e3)
    print(line)


if __name__ == "__main__":
    main()
