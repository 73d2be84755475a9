#!/usr/bin/env python3
"""Debug: where the waves of one exhaustive-search launch ran and for how long. Needs the trace
build of the engine: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DX265AMD_TRACE
-o tools/libx265_amd_me_trace.so x265_amd/csrc/me_engine.hip. Prints wave lifetimes, the span of the launch and how many waves each SIMD ran."""
import os
import sys
import collections

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    import torch
    import x265_amd.engine as eng
    eng.LIB_PATH = os.path.join(ROOT, "tools", "libx265_amd_me_trace.so")
    from x265_amd.engine import Engine, METHODS, RESULT_DTYPE
    from x265_amd.synth import make_clip, pad_plane
    from x265_amd import workload
    W, H, M = 3840, 2160, 128
    dev = torch.device("cuda", 0)
    e = Engine(0, 8)
    if len(sys.argv) > 2:
        e.set_esa_window_bytes(int(sys.argv[2]))
    clip = make_clip(W, H, 2, seed=99)
    cur, org, stride = pad_plane(clip[1], M)
    ref, _, _ = pad_plane(clip[0], M)
    pc = e.upload_plane(cur, W, H, M, M, stride)
    pr = e.upload_plane(ref, W, H, M, M, stride)
    rng = np.random.default_rng(5)
    wcu, hcu = (W // 2 + 7) // 8, (H // 2 + 7) // 8
    la = rng.integers(-12, 13, (wcu * hcu, 2)).astype(np.int16)
    jobs = workload.square_pu_jobs(W, H, la, wcu, hcu, 92, sizes=(size,))
    n = len(jobs)
    d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(n, -1).copy()).to(dev)
    d_res = torch.zeros(n * 16 + n * 4 * 32, dtype=torch.uint8, device=dev)
    for _ in range(2):
        e.search_batch_dev(pc, pr, d_jobs.data_ptr(), n, d_res.data_ptr(), 32, METHODS["full"], 5, max_pu_dim=size)
        e.sync()
    t = d_res.cpu().numpy()[n * 16:].view(np.uint64).reshape(n, 4, 4)
    start, end, hw, scan_end = t[..., 0].astype(np.int64), t[..., 1].astype(np.int64), t[..., 2], t[..., 3].astype(np.int64)
    t0 = start.min()
    print("launch span %.1f us; wave lifetime mean %.1f us (min %.1f max %.1f); scan part mean %.1f us" % (
        (end.max() - t0) / 100.0, (end - start).mean() / 100.0, (end - start).min() / 100.0, (end - start).max() / 100.0,
        (scan_end - start).mean() / 100.0))
    hwid = (hw & 0xffffffff).astype(np.int64)
    xcc = (hw >> 32).astype(np.int64) & 0xf
    simd = (hwid >> 4) & 3
    cu = (hwid >> 8) & 0xf
    sh = (hwid >> 12) & 1
    se = (hwid >> 13) & 7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    per_cu = collections.Counter(key[:, 0].tolist())
    print("CUs used %d; workgroups per CU min %d max %d" % (len(per_cu), min(per_cu.values()), max(per_cu.values())))
    # concurrency per CU: sweep
    busy = []
    for k in list(per_cu)[:8]:
        sel = key[:, 0] == k
        ev = sorted([(s, 1) for s in start[sel, 0]] + [(x, -1) for x in end[sel].max(axis=1)])
        cur, last, acc = 0, ev[0][0], collections.Counter()
        for tt, d in ev:
            acc[cur] += tt - last
            last = tt
            cur += d
        busy.append({c: round(v / 100.0, 1) for c, v in sorted(acc.items())})
    print("time (us) at each workgroup concurrency, first CUs:", busy)
    wg_start = start.min(axis=1)
    order = np.argsort(wg_start)
    print("first 12 workgroup start times (us):", [round((wg_start[i] - t0) / 100.0, 1) for i in order[:12]])
    print("last 12 workgroup end times (us):", [round((end[i].max() - t0) / 100.0, 1) for i in np.argsort(end.max(axis=1))[-12:]])


if __name__ == "__main__":
    main()
