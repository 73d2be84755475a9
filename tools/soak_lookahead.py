#!/usr/bin/env python3
"""Randomised soak of the lookahead MV fields (hexagon search, the band kernel) against the CPU oracle: random picture
sizes, content kinds, global motions, depths, QPs and sub-pel levels; MVs, costs and visit counts compared.

    python tools/soak_lookahead.py [--fields 120 --seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fields", type=int, default=120)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    from x265_amd.engine import Engine
    from x265_amd.synth import make_clip
    from oracle.bindings import Oracle, MEStats   # checker only
    from util import host_lowres_planes
    orc = Oracle()
    rng = np.random.default_rng(args.seed)
    eng = {8: Engine(0, 8), 10: Engine(0, 10)}
    bad = cus = 0
    t0 = time.time()
    for n in range(args.fields):
        depth = int(rng.choice([8, 10]))
        e = eng[depth]
        w, h = int(rng.integers(2, 40)) * 16, int(rng.integers(2, 24)) * 16
        kind = str(rng.choice(["texture", "noise"]))
        dx, dy = (int(v) for v in rng.integers(-40, 41, 2))
        big = make_clip(w + 96, h + 96, 2, seed=int(rng.integers(1 << 30)), depth=depth, kind=kind)
        ref = np.ascontiguousarray(big[0][48:48 + h, 48:48 + w])
        cur = np.ascontiguousarray(big[int(rng.integers(0, 2))][48 + dy:48 + dy + h, 48 + dx:48 + dx + w])
        subme, qp, merange = int(rng.integers(0, 8)), int(rng.integers(0, 52)), int(rng.choice([8, 16, 32]))
        margin = int(rng.choice([12, 13, 16, 64]))
        pa, pb = host_lowres_planes(orc, depth, cur, 64), host_lowres_planes(orc, depth, ref, 64)
        wcu, hcu, lw, lh = pa[3], pa[4], pa[5], pa[6]
        ctx = orc.make_ctx(depth, pa[0][0], pa[1], pa[2], pb[0], pa[1], pa[2], True, qp, 1, subme)
        st = MEStats()
        mv, cost = orc.lookahead_field(ctx, wcu, hcu, merange, st)
        na, nb = host_lowres_planes(orc, depth, cur, 16), host_lowres_planes(orc, depth, ref, margin)
        fenc = e.upload_plane(na[0][0], lw, lh, 16, 16, na[2])
        refs = [e.upload_plane(nb[0][k], lw, lh, margin, margin, nb[2]) for k in range(4)]
        mvs, costs, stats = e.lookahead_fields([(fenc, refs)], wcu, hcu, qp, 1, subme, merange, want_stats=True)
        ok = (np.array_equal(mvs[0], mv) and np.array_equal(costs[0], cost)
              and int(stats["sad_calls"].sum()) == st.sad_calls and int(stats["subpel_calls"].sum()) == st.subpel_calls)
        for p in [fenc] + refs:
            e.free_plane(p)
        cus += wcu * hcu
        if not ok:
            bad += 1
            print("MISMATCH field %d: %dx%d depth %d %s motion (%d, %d) subme %d qp %d merange %d margin %d"
                  % (n, w, h, depth, kind, dx, dy, subme, qp, merange, margin), flush=True)
        if n % 20 == 19:
            print("%d fields, %d CUs, %d mismatching fields, %.0f s" % (n + 1, cus, bad, time.time() - t0), flush=True)
    print("lookahead soak: %d fields, %d CU searches, %d mismatching fields" % (args.fields, cus, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
