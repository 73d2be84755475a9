#!/usr/bin/env python3
"""Latency of tiny launches (a handful of PUs) of the search entry point, per method: what one launch chain costs when
there is next to nothing to do. HIP events on the engine's stream; run under rocprofv3 --kernel-trace for per-kernel times."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from x265_amd.engine import Engine, METHODS, RESULT_DTYPE, make_jobs
from x265_amd.synth import make_clip, pad_plane
from x265_amd.measure import _timed
dev = torch.device("cuda", 0)
e = Engine(0, 8)
W, H, M = 416, 240, 96
clip = make_clip(W, H, 2, seed=1)
cur, org, stride = pad_plane(clip[1], M)
ref, _, _ = pad_plane(clip[0], M)
pc, pr = e.upload_plane(cur, W, H, M, M, stride), e.upload_plane(ref, W, H, M, M, stride)
for size in (8, 16, 64):
    for n in (4, 1024):
        jobs = make_jobs(n)
        jobs["pu_x"], jobs["pu_y"], jobs["w"], jobs["h"] = (np.arange(n) % 16) * size % (W - 64), 64, size, size
        jobs["mvmin"], jobs["mvmax"], jobs["merange"] = (-16, -16), (16, 16), 16
        d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(n, -1).copy()).to(dev)
        d_res = torch.zeros((n, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        for m in ("hex", "star"):
            ms = _timed(torch, e, dev, lambda: e.search_batch_dev(pc, pr, d_jobs.data_ptr(), n, d_res.data_ptr(), 30, METHODS[m], 5, max_pu_dim=size), 20)
            print("size %2d  %5d PUs  %-4s  %.3f ms per launch chain" % (size, n, m, ms), flush=True)
