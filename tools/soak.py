#!/usr/bin/env python3
"""Randomised soak of the search paths against the CPU oracle, beyond what the test suite samples: for every depth,
method, launch size class and sub-pel level a fresh batch of random jobs per round (square PUs of the class mixed with
other partitions), MVs, costs and visit counts compared. Prints the first mismatch of a batch and a summary.

    python tools/soak.py [--rounds 4 --jobs 160 --seed 1]
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
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--jobs", type=int, default=160)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    from x265_amd.engine import Engine, METHODS
    from oracle.bindings import Oracle   # checker only
    from util import clip_planes, random_jobs, oracle_search, margin_for
    orc = Oracle()
    others = {8: [(8, 4), (4, 8)], 16: [(16, 8), (8, 16), (8, 8), (12, 16), (16, 4)],
              32: [(32, 16), (16, 32), (16, 16), (32, 24), (8, 32)], 64: [(64, 32), (32, 64), (64, 48), (16, 64), (32, 32)]}
    W, H = 320, 192
    bad = total = 0
    t0 = time.time()
    for depth in (8, 10):
        e = Engine(0, depth)
        for mname in ("star", "hex", "dia", "umh", "full"):
            method = METHODS[mname]
            m = margin_for(method)
            for kind in ("texture", "noise"):
                fenc, ref, org, stride = clip_planes(W, H, depth, kind, margin=m)
                pf = e.upload_plane(fenc, W, H, m, m, stride)
                pr = e.upload_plane(ref, W, H, m, m, stride)
                for size in (8, 16, 32, 64):
                    for rnd in range(args.rounds):
                        rng = np.random.default_rng([args.seed, depth, method, size, rnd, kind == "noise"])
                        subme = int(rng.integers(0, 8))
                        n = args.jobs if method != 4 else max(8, args.jobs // 8)
                        jobs = random_jobs(rng, n + int(rng.integers(0, 4)), W, H, method, parts=[(size, size)] * 6 + others[size], max_numc=4)
                        qp = int(rng.integers(10, 45)) + 6 * (depth - 8)
                        r = e.search_batch(pf, pr, jobs, qp, method, subme)
                        mv, cost, sc, qc = oracle_search(orc, depth, fenc, [ref], org, stride, jobs, qp, method, subme)
                        ok = (r["mv"] == mv).all(axis=1) & (r["cost"] == cost) & (r["sad_calls"] == sc) & (r["subpel_calls"] == qc)
                        total += len(jobs)
                        if not ok.all():
                            i = int(np.flatnonzero(~ok)[0])
                            bad += int((~ok).sum())
                            print("MISMATCH depth %d %s %s size %d subme %d qp %d round %d: job %d %s got %s %d exp %s %d" %
                                  (depth, mname, kind, size, subme, qp, rnd, i, jobs[i], r["mv"][i], r["cost"][i], mv[i], cost[i]))
                e.free_plane(pf)
                e.free_plane(pr)
            print("depth %d %-5s done: %d searches so far, %d mismatches, %.0f s" % (depth, mname, total, bad, time.time() - t0), flush=True)
        e.close()
    print("SOAK %s: %d searches, %d mismatches" % ("OK" if not bad else "FAILED", total, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
