#!/usr/bin/env python3
"""Randomised soak of the launches bench.py's step is made of -- the quadtree-ordered 64x64 + 32x32 list
(x265amd_search_batch_pairs_dev with X265AMD_SQUARE_ONLY) and the 16x16 list with its 8x8 children
(x265amd_search_batch_pairs_family_dev) -- against the CPU oracle, every PU of every picture: MVs, costs, visit counts.
Per round a fresh clip pair (texture with a random shift, or noise), sample depth 8 or 10, a random sub-pel level, QP
and range, and a predictor field that is coherent (one MV, a few CUs differing: parents keep their children), smooth
noise around one MV, or random per CU (children mostly leave their parents).

    python tools/soak_families.py [--rounds 24 --seed 1]
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
    ap.add_argument("--rounds", type=int, default=24)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    from x265_amd import workload
    from x265_amd.engine import Engine, METHODS, RESULT_DTYPE, SQUARE_ONLY
    from x265_amd.synth import make_clip, pad_plane, lowres_dims
    from oracle.bindings import Oracle   # checker only
    orc = Oracle()
    W, H, M = 384, 248, 128              # 6 x 3.9 CTUs: the last CTU row is cut, 8x8 PUs without a parent exist
    lw, lh, wcu, hcu = lowres_dims(W, H)
    STAR = METHODS["star"]
    bad = total = kept = 0
    t0 = time.time()
    engines = {8: Engine(0, 8), 10: Engine(0, 10)}
    for rnd in range(args.rounds):
        rng = np.random.default_rng([args.seed, rnd])
        depth = 8 if rnd % 3 else 10
        e = engines[depth]
        kind = ("texture", "noise", "static")[int(rng.integers(0, 3))] if rnd % 4 else "texture"
        clip = make_clip(W, H, 3, seed=int(rng.integers(1, 1 << 30)), depth=depth, kind=kind)
        pairs_f = [(1, 0), (2, 1)]
        pads = [pad_plane(f, M) for f in clip]
        planes = [e.upload_plane(p[0], W, H, M, M, p[2]) for p in pads]
        org, stride = pads[0][1], pads[0][2]
        subme = int(rng.integers(0, 8))
        qp = int(rng.integers(12, 45)) + 6 * (depth - 8)
        merange = int(rng.choice([8, 16, 57]))
        field = ("coherent", "smooth", "random")[rnd % 3]
        host_jobs = []
        for _ in pairs_f:
            g = rng.integers(-60, 61, 2)
            if field == "coherent":
                mv = np.tile(g, (wcu * hcu, 1))
                odd = rng.random(wcu * hcu) < 0.05
                mv[odd] += rng.integers(-12, 13, (int(odd.sum()), 2))
            elif field == "smooth":
                mv = g + rng.integers(-2, 3, (wcu * hcu, 2))
            else:
                mv = rng.integers(-60, 61, (wcu * hcu, 2))
            jobs = workload.square_pu_jobs(W, H, (mv // 2).astype(np.int16), wcu, hcu, merange)
            # (the field's MVs are doubled half-resolution MVs, i.e. even: an odd offset for the whole picture brings
            # quarter-pel predictors in without taking parents and children apart)
            odd_ = rng.integers(0, 2, 2).astype(jobs["qmvp"].dtype)
            jobs["qmvp"] += odd_
            jobs["mvc"][:, :2] += odd_
            host_jobs.append(jobs)
        d_pairs = e.pairs_upload([(planes[a], planes[b]) for a, b in pairs_f])
        res = [np.zeros(len(j), RESULT_DTYPE) for j in host_jobs]
        for dim in (64, 16):
            if dim == 64:
                cj, ci = workload.class_job_list(host_jobs, (64, 32), families=True)
                npar = 0
            else:
                pj, pi, kj, ki = workload.family_lists(host_jobs, 16)
                cj, ci, npar = np.concatenate([pj, kj]), np.concatenate([pi, ki]), len(pj)
            d_jobs, d_res = e.device_alloc(cj.nbytes), e.device_alloc(len(cj) * RESULT_DTYPE.itemsize)
            e.upload(d_jobs, cj)
            if dim == 64:
                e.search_batch_pairs_dev(d_pairs, len(pairs_f), d_jobs, len(cj), d_res, qp, STAR, subme, max_pu_dim=64 | SQUARE_ONLY)
            else:
                e.search_batch_pairs_family_dev(d_pairs, len(pairs_f), d_jobs, npar, d_res, d_jobs + 40 * npar, len(cj) - npar,
                                                d_res + 16 * npar, qp, subme)
            got = e.download(d_res, (len(cj),), RESULT_DTYPE)
            for i in range(len(pairs_f)):
                sel = ci[:, 0] == i
                res[i][ci[sel, 1]] = got[sel]
            e.device_free(d_jobs)
            e.device_free(d_res)
        for i, (a, b) in enumerate(pairs_f):
            ctx = orc.make_ctx(depth, pads[a][0], org, stride, [pads[b][0]], org, stride, False, qp, STAR, subme)
            exp = orc.me_search_batch(ctx, host_jobs[i])
            ok = ((res[i]["mv"] == exp["mv"]).all(axis=1) & (res[i]["cost"] == exp["cost"]) &
                  (res[i]["sad_calls"] == exp["sad_calls"]) & (res[i]["subpel_calls"] == exp["subpel_calls"]))
            total += len(ok)
            # children that ended where their parent did (what the shared refinement is for)
            j = host_jobs[i]
            if not ok.all():
                k = int(np.flatnonzero(~ok)[0])
                bad += int((~ok).sum())
                print("MISMATCH round %d depth %d %s %s subme %d qp %d merange %d pair %d: job %d %s got %s %d exp %s %d"
                      % (rnd, depth, kind, field, subme, qp, merange, i, k, j[k], res[i]["mv"][k], res[i]["cost"][k], exp["mv"][k], exp["cost"][k]))
        e.device_free(d_pairs)
        for p in planes:
            e.free_plane(p)
        print("round %2d depth %2d %-7s %-8s subme %d merange %2d: %d searches so far, %d mismatches, %.0f s"
              % (rnd, depth, kind, field, subme, merange, total, bad, time.time() - t0), flush=True)
    for e in engines.values():
        e.close()
    print("SOAK %s: %d searches, %d mismatches" % ("OK" if not bad else "FAILED", total, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
