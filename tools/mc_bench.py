#!/usr/bin/env python3
"""Motion compensation (x265amd_mc_batch_dev) of whole pictures against the HBM roofline: every PU of a frame cut
into one PU size, luma + 4:2:0 chroma, uni- or bi-predicted with random quarter-pel MVs (all phases), device-resident
jobs. Algorithmic bytes per PU = its reference footprints (8-tap luma, 4-tap chroma, per list) + the prediction
samples written. HIP events on the engine's stream; one batch is checked against the oracle first.

    python tools/mc_bench.py [--width 1920 --height 1080 --depth 8 --reps 20]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--lib", default=None, help="another build of the engine library, e.g. one with -DX265AMD_MC_WAVES=n")
    args = ap.parse_args()
    import torch
    if args.lib:
        import x265_amd.engine as eng
        eng.LIB_PATH = os.path.abspath(args.lib)
    from x265_amd.engine import Engine, MC_JOB_DTYPE, Picture
    from x265_amd.synth import make_clip, make_chroma, pad_plane
    from x265_amd.measure import mixed_partition
    dev = torch.device("cuda", 0)
    e = Engine(0, args.depth)
    W, H, M = args.width, args.height, 304
    bpp = 1 if args.depth <= 8 else 2
    ys = make_clip(W, H, 2, seed=3, depth=args.depth)
    cbs, crs = make_chroma(W, H, 2, seed=3, depth=args.depth)
    host, dpics = [], []
    for i in range(2):
        planes = [pad_plane(ys[i], M), pad_plane(cbs[i], M // 2), pad_plane(crs[i], M // 2)]
        host.append(planes)
        dims = ((W, H, M), (W // 2, H // 2, M // 2), (W // 2, H // 2, M // 2))
        dpics.append(e.upload_picture([(b, o, s, w, h, m) for (b, o, s), (w, h, m) in zip(planes, dims)]))
    dt = np.uint8 if args.depth <= 8 else np.uint16
    blank = [(np.zeros(W * H, dt), 0, W, W, H, 0), (np.zeros(W * H // 4, dt), 0, W // 2, W // 2, H // 2, 0),
             (np.zeros(W * H // 4, dt), 0, W // 2, W // 2, H // 2, 0)]
    dpred = e.upload_picture(blank)
    d_planes = e.pictures_upload(dpics)
    stream = torch.cuda.ExternalStream(e.stream(), device=dev)
    rng = np.random.default_rng(5)
    for size in (8, 16, 32, 64, 0):
        for dir_ in (1, 3):
            if size:
                xs, yy = np.meshgrid(np.arange(0, W - size + 1, size), np.arange(0, H - size + 1, size))
                jobs = np.zeros(xs.size, MC_JOB_DTYPE)
                jobs["pu_x"], jobs["pu_y"], jobs["w"], jobs["h"] = xs.ravel(), yy.ravel(), size, size
            else:
                jobs = mixed_partition(rng, W, H, MC_JOB_DTYPE)
            jobs["dir"], jobs["flags"], jobs["ref0"], jobs["ref1"] = dir_, 3, 0, 1
            jobs["mv0"] = rng.integers(-64, 65, (len(jobs), 2))
            jobs["mv1"] = rng.integers(-64, 65, (len(jobs), 2))
            lst = e.mc_list_create(dpics, dpred, jobs)  # unweighted
            e.mc_batch_dev(d_planes, dpred, lst)
            e.sync()
            if not args.no_check and size in (8, 64):
                from oracle.bindings import Oracle   # checker only
                n = min(len(jobs), 400)
                exp = [(np.zeros(W * H, dt), 0, W), (np.zeros(W * H // 4, dt), 0, W // 2), (np.zeros(W * H // 4, dt), 0, W // 2)]
                Oracle().mc_batch(args.depth, [[pl[:3] for pl in p] for p in host], exp, jobs[:n])
                got = [e.download_plane(p).reshape(-1) for p in (dpred.y, dpred.cb, dpred.cr)]
                rows = int(jobs["pu_y"][n - 1])          # rows above the n-th PU's row are complete in both
                assert np.array_equal(got[0][:rows * W], exp[0][0][:rows * W]) and \
                    np.array_equal(got[1][:rows // 2 * (W // 2)], exp[1][0][:rows // 2 * (W // 2)]) and \
                    np.array_equal(got[2][:rows // 2 * (W // 2)], exp[2][0][:rows // 2 * (W // 2)]), "mismatch vs oracle"
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(stream)
            for _ in range(args.reps):
                e.mc_batch_dev(d_planes, dpred, lst)
            ev1.record(stream)
            e.sync()
            ms = ev0.elapsed_time(ev1) / args.reps
            lists = 2 if dir_ == 3 else 1
            jw, jh = jobs["w"].astype(np.int64), jobs["h"].astype(np.int64)
            nbytes = int((bpp * (lists * ((jw + 7) * (jh + 7) + 2 * (jw // 2 + 3) * (jh // 2 + 3)) + jw * jh * 3 // 2)).sum())
            print("%dx%d %d-bit MC %-6s %s: %6d PUs %.3f ms  %.0f GB/s algorithmic (%.1f MB), %.2f ns per PU, %.1f Gsample/s"
                  % (W, H, args.depth, "%dx%d" % (size, size) if size else "mixed", "bi " if dir_ == 3 else "uni", len(jobs), ms,
                     nbytes / ms / 1e6, nbytes / 1e6, ms * 1e6 / len(jobs), float((jw * jh).sum()) * 1.5 / ms / 1e6))
            e.mc_list_free(lst)


if __name__ == "__main__":
    main()
