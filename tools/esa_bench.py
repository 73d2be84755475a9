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

SAD_PEAK = 150e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--merange", type=int, default=92)
    ap.add_argument("--sizes", default="64,32,16,8")
    ap.add_argument("--subme", type=int, default=5)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--window", type=int, default=0, help="LDS window bytes (0 = library default)")
    ap.add_argument("--check", type=int, default=0, help="PUs per size to compare with the CPU oracle")
    ap.add_argument("--lib", default="", help="engine library to load instead of the in-tree build (experiments)")
    args = ap.parse_args()

    import torch
    if args.lib:
        import x265_amd.engine as eng
        eng.LIB_PATH = os.path.abspath(args.lib)
    from x265_amd.engine import Engine, METHODS, RESULT_DTYPE
    from x265_amd.synth import make_clip, pad_plane
    from x265_amd import workload

    W, H, M = args.width, args.height, 128
    dev = torch.device("cuda", 0)
    e = Engine(0, 8)
    if args.window:
        e.set_esa_window_bytes(args.window)
    clip = make_clip(W, H, 2, seed=99)
    cur, org, stride = pad_plane(clip[1], M)
    ref, _, _ = pad_plane(clip[0], M)
    pc = e.upload_plane(cur, W, H, M, M, stride)
    pr = e.upload_plane(ref, W, H, M, M, stride)
    rng = np.random.default_rng(5)
    wcu, hcu = (W // 2 + 7) // 8, (H // 2 + 7) // 8
    la = rng.integers(-12, 13, (wcu * hcu, 2)).astype(np.int16)      # stand-in lookahead field
    stream = torch.cuda.ExternalStream(e.stream(), device=dev)
    total_px, total_ms = 0, 0.0
    for s in [int(v) for v in args.sizes.split(",")]:
        jobs = workload.square_pu_jobs(W, H, la, wcu, hcu, args.merange, sizes=(s,))
        n = len(jobs)
        d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(n, -1).copy()).to(dev)
        d_res = torch.zeros((n, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()

        def launch():
            e.search_batch_dev(pc, pr, d_jobs.data_ptr(), n, d_res.data_ptr(), 32, METHODS["full"], args.subme, max_pu_dim=s)
        launch()
        e.sync()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        for _ in range(args.reps):
            launch()
        ev1.record(stream)
        e.sync()
        ms = ev0.elapsed_time(ev1) / args.reps
        res = d_res.cpu().numpy().view(RESULT_DTYPE).reshape(-1)
        px = workload.pixel_sads(jobs, res)
        total_px += px
        total_ms += ms
        line = "%2dx%-2d  %6d PUs  %8.3f ms  %7.2f T pixel-SAD/s  %5.1f %% of SAD issue peak" % (
            s, s, n, ms, px / ms / 1e9, 100.0 * px / (ms * 1e-3) / SAD_PEAK)
        if args.check:
            from oracle.bindings import Oracle   # checker only
            from tests.util import oracle_search
            orc = Oracle()
            pick = np.linspace(0, n - 1, args.check).astype(int)
            mv, cost, sc, qc = oracle_search(orc, 8, cur, [ref], org, stride, jobs[pick], 32, METHODS["full"], args.subme)
            ok = (np.array_equal(res["mv"][pick], mv) and np.array_equal(res["cost"][pick], cost)
                  and np.array_equal(res["sad_calls"][pick], sc))
            line += "  oracle %s" % ("OK" if ok else "MISMATCH")
        print(line, flush=True)
    print("all     %8.3f ms  %7.2f T pixel-SAD/s  %5.1f %% of SAD issue peak" % (
        total_ms, total_px / total_ms / 1e9, 100.0 * total_px / (total_ms * 1e-3) / SAD_PEAK))


if __name__ == "__main__":
    main()
