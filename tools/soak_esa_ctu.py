#!/usr/bin/env python3
"""Randomised soak of the exhaustive scan per CTU (x265amd_esa_ctu_batch) against the per-PU scans of the same jobs
(every record) and against the CPU oracle (a sample): picture sizes that cut CTUs, predictors spread by up to 24 samples,
ranges of their own per PU (down to a single position), PUs left out at random (empty slots), 8- and 10-bit samples.

    python tools/soak_esa_ctu.py [--rounds 40 --seed 1]
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
    ap.add_argument("--rounds", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    from x265_amd.engine import Engine, METHODS
    from x265_amd import workload
    from oracle.bindings import Oracle   # checker only
    from util import clip_planes, oracle_search
    orc = Oracle()
    engines = {8: Engine(0, 8), 10: Engine(0, 10)}
    bad = total = checked = 0
    t0 = time.time()
    for rnd in range(args.rounds):
        rng = np.random.default_rng([args.seed, rnd])
        depth = int(rng.choice([8, 8, 10]))
        e = engines[depth]
        W, H = int(rng.integers(8, 41)) * 8, int(rng.integers(8, 33)) * 8
        kind = str(rng.choice(["texture", "noise", "static"]))
        merange = int(rng.choice([1, 3, 8, 16, 33, 57, 64]))
        spread = int(rng.choice([0, 1, 4, 12, 24]))
        m = merange + spread + 40
        fenc, ref, org, stride = clip_planes(W, H, depth, kind, seed=int(rng.integers(1, 1000)), margin=m)
        pf = e.upload_plane(fenc, W, H, m, m, stride)
        pr = e.upload_plane(ref, W, H, m, m, stride)
        wcu, hcu = (W // 2 + 7) // 8, (H // 2 + 7) // 8
        la = (rng.integers(-spread, spread + 1, (wcu * hcu, 2)) * 2).astype(np.int16)
        jobs = workload.square_pu_jobs(W, H, la, wcu, hcu, merange, clip_margin=m - 12)
        if rng.random() < 0.6:      # ranges and predictors of their own
            for j in jobs:
                cx, cy = int(j["qmvp"][0]) >> 2, int(j["qmvp"][1]) >> 2
                ext = rng.integers(0, merange + 1, 4)
                j["mvmin"] = (max(cx - ext[0], j["mvmin"][0]), max(cy - ext[1], j["mvmin"][1]))
                j["mvmax"] = (min(cx + ext[2], j["mvmax"][0]), min(cy + ext[3], j["mvmax"][1]))
                j["qmvp"] = (int(j["qmvp"][0]) + int(rng.integers(-5, 6)), int(j["qmvp"][1]) + int(rng.integers(-5, 6)))
        if rng.random() < 0.5:      # some PUs are not searched
            jobs = jobs[rng.random(len(jobs)) > rng.choice([0.05, 0.5, 0.9])]
        if not len(jobs):
            continue
        cj, index, nctu = workload.esa_ctu_jobs(jobs)
        qp, subme = int(rng.integers(4, 46)) + 6 * (depth - 8), int(rng.integers(0, 8))
        res = e.esa_ctu_batch(pf, pr, cj, nctu, qp, subme)
        per = e.search_batch(pf, pr, jobs, qp, METHODS["full"], subme)
        have = index >= 0
        ok = res[have] == per[index[have]]
        total += int(have.sum())
        if not ok.all() or not (res["cost"][~have] == -1).all():
            bad += int((~ok).sum())
            i = int(np.flatnonzero(~ok)[0]) if not ok.all() else -1
            print("MISMATCH round %d depth %d %s %dx%d merange %d spread %d qp %d subme %d: slot %d %s got %s exp %s" %
                  (rnd, depth, kind, W, H, merange, spread, qp, subme, i, cj[have][i] if i >= 0 else "", res[have][i] if i >= 0 else "",
                   per[index[have]][i] if i >= 0 else ""))
        pick = rng.choice(len(jobs), min(12, len(jobs)), replace=False)
        mv, cost, sc, qc = oracle_search(orc, depth, fenc, [ref], org, stride, jobs[pick], qp, METHODS["full"], subme)
        o_ok = (per["mv"][pick] == mv).all(axis=1) & (per["cost"][pick] == cost) & (per["sad_calls"][pick] == sc) & (per["subpel_calls"][pick] == qc)
        checked += len(pick)
        if not o_ok.all():
            bad += int((~o_ok).sum())
            print("ORACLE MISMATCH round %d" % rnd)
        e.free_plane(pf)
        e.free_plane(pr)
        if rnd % 10 == 9:
            print("round %d: %d PUs compared, %d through the oracle, %d mismatches, %.0f s" % (rnd + 1, total, checked, bad, time.time() - t0), flush=True)
    for e in engines.values():
        e.close()
    print("SOAK %s: %d PUs compared with the per-PU scans, %d of them with the oracle, %d mismatches" % ("OK" if not bad else "FAILED", total, checked, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
