#!/usr/bin/env python3
"""Lookahead frame-cost throughput on one GPU: estimateFrameCost of a batch of B-frame triples (two lowres MV
fields each, then the intra / bidirectional / choice kernels), device resident. Prints the time of the search
launch and of the cost launches (HIP events on the engine's stream) and, with --check, compares the first triple
with the CPU oracle.

    python tools/la_bench.py [--width 1920 --height 1080 --frames 16 --reps 3 --check]
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
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()

    import torch
    from x265_amd.engine import Engine
    from x265_amd import measure

    e = Engine(0, 8)
    m = measure.lookahead_costs(e, torch.device("cuda", 0), args.width, args.height, args.frames, args.reps)
    tot = m["ms_search"] + m["ms_costs"]
    line = "%dx%d  %d triples (%d fields of %d CUs)  search %.3f ms  intra+bidir+choice %.3f ms  -> %.1f frame costs/s, %.2f M CU/s" % (
        args.width, args.height, m["frames"], m["fields"], m["cus"], m["ms_search"], m["ms_costs"],
        m["frames"] / tot * 1e3, m["fields"] * m["cus"] / m["ms_search"] / 1e3)
    if args.check:
        from oracle.bindings import Oracle   # checker only
        orc = Oracle()
        org, stride, wcu, hcu = m["geometry"]
        h = m["host_planes"]
        r = m["first"]
        t0 = time.perf_counter()
        ok = True
        for l, ref in ((0, h[0]), (1, h[2])):
            ctx = orc.make_ctx(8, h[1][0], org, stride, ref, org, stride, True, 12, 1, 1)
            mv, cost = orc.lookahead_field(ctx, wcu, hcu, 16)
            ok &= np.array_equal(mv, r["mvs"][l]) and np.array_equal(cost, r["mv_costs"][l])
        exp = orc.lookahead_frame_cost(8, h[1][0], h[0], h[2], org, stride, wcu, hcu, 1, 1, r["mvs"][0], r["mv_costs"][0],
                                       r["mvs"][1], r["mv_costs"][1])
        dt = time.perf_counter() - t0
        ok &= all(np.array_equal(exp[k], r[k]) for k in ("intra_cost", "lowres_costs", "row_satds_intra", "row_satds", "totals"))
        line += "  oracle %s (%.0f ms per triple on one host core)" % ("OK" if ok else "MISMATCH", dt * 1e3)
    print(line)


if __name__ == "__main__":
    main()
