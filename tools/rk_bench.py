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

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--check", type=int, default=0, help="CTUs to compare with the CPU oracle")
    args = ap.parse_args()

    import torch
    from x265_amd.engine import Engine, RK_CTU_DTYPE, RK_RESULT_DTYPE
    from x265_amd.synth import make_clip, pad_plane
    from x265_amd import workload

    W, H, M = args.width, args.height, 304
    dev = torch.device("cuda", 0)
    e = Engine(0, 8)
    clip = make_clip(W, H, 3, seed=5)
    cur, org, stride = pad_plane(clip[2], M)
    rorig, _, _ = pad_plane(clip[0], M)
    noise = np.random.default_rng(5).integers(-2, 3, clip[0].shape)
    recon, _, _ = pad_plane(np.clip(clip[0].astype(int) + noise, 0, 255).astype(np.uint8), M)
    pc = e.upload_plane(cur, W, H, M, M, stride)
    po = e.upload_plane(rorig, W, H, M, M, stride)
    pr = e.upload_plane(recon, W, H, M, M, stride)
    pds = e.rk_ds_plane(po)
    rng = np.random.default_rng(6)
    nctu = ((W + 63) // 64) * ((H + 63) // 64)
    pmvs = (rng.integers(-12, 13, (nctu, 2, 2)) * 2).astype(np.int16)
    mvps = (rng.integers(-12, 13, (nctu, 3, 2)) * 2).astype(np.int16)
    jobs = workload.rk_ctu_jobs(W, H, pmvs, mvps, have_mvd=True)
    d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(nctu, -1).copy()).to(dev)
    d_res = torch.zeros((nctu, RK_RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    stream = torch.cuda.ExternalStream(e.stream(), device=dev)

    def launch():
        e.rk_ctu_batch_dev(pc, pr, pds, 30, d_jobs.data_ptr(), nctu, d_res.data_ptr())
    launch()
    e.sync()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(stream)
    for _ in range(args.reps):
        launch()
    ev1.record(stream)
    e.sync()
    ms = ev0.elapsed_time(ev1) / args.reps
    res = d_res.cpu().numpy().view(RK_RESULT_DTYPE).reshape(-1)
    nx = (jobs["mvmax"][:, 0].astype(np.int64) // 4) - (-(-jobs["mvmin"][:, 0].astype(np.int64) // 4)) + 1
    ny = (jobs["mvmax"][:, 1].astype(np.int64) // 4) - (-(-jobs["mvmin"][:, 1].astype(np.int64) // 4)) + 1
    cime_px = int((nx * ny * (jobs["valid_w"].astype(np.int64) // 4) * (jobs["valid_h"].astype(np.int64) // 4)).sum())
    valid = res["rime"]["cost"] >= 0
    area = np.array([64] * 64 + [256] * 16 + [1024] * 4 + [4096], np.int64)
    cu_px = int((valid * area[None, :]).sum())
    px = cime_px + cu_px * (576 + 49)
    line = "%dx%d  %d CTUs  %.3f ms per picture-reference  %.1f kCTU/s  %.2f T pixel-SAD/s (CIME %.2f G, RIME+FME %.2f G per picture)" % (
        W, H, nctu, ms, nctu / ms, px / ms / 1e9, cime_px / 1e9, cu_px * 625 / 1e9)
    if args.check:
        from oracle.bindings import Oracle   # checker only
        orc = Oracle()
        ds, dorg, dstride = orc.rk_downsample(8, rorig, org, W, H, stride, M // 4)
        ctx = orc.make_ctx(8, cur, org, stride, [recon], org, stride, False, 30, 0, 0)
        pick = np.linspace(0, nctu - 1, args.check).astype(int)
        import time
        t0 = time.perf_counter()
        exp = orc.rk_ctu_batch(ctx, ds, dorg, dstride, W, H, jobs[pick])
        dt = time.perf_counter() - t0
        ok = all(np.array_equal(res[pick][k]["mv"], exp[k]["mv"]) and np.array_equal(res[pick][k]["cost"], exp[k]["cost"])
                 for k in ("cime", "rime", "fme"))
        line += "  oracle %s (%.1f ms per CTU on one host core)" % ("OK" if ok else "MISMATCH", dt / len(pick) * 1e3)
    print(line)


if __name__ == "__main__":
    main()
