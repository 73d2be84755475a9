#!/usr/bin/env python3
"""One launch over a mixed list -- every square 64/32/16/8 PU of a 4K picture in quadtree order of size, max_pu_dim 64, nothing
said about its contents -- beside the four per-size launches of the same PUs (X265AMD_SQUARE_ONLY); records compared.

    python tools/mixed_bench.py [--method star --depth 8]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--method", default="star")
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--subme", type=int, default=5)
    args = ap.parse_args()
    import torch
    from x265_amd.engine import Engine, METHODS, RESULT_DTYPE
    from x265_amd import measure
    dev = torch.device("cuda", 0)
    e = Engine(0, args.depth)
    per = measure.pattern_search_picture(e, dev, depth=args.depth, method=args.method, subme=args.subme)
    jobs, want = per["jobs"], per["results"]
    # the same planes again (pattern_search_picture frees its own)
    from x265_amd.synth import make_clip, pad_plane
    clip = make_clip(3840, 2160, 2, seed=31, depth=args.depth)
    planes = []
    for f in clip:
        buf, org, stride = pad_plane(f, 128)
        planes.append(e.upload_plane(buf, 3840, 2160, 128, 128, stride))
    n = len(jobs)
    d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(n, -1).copy()).to(dev)
    d_res = torch.zeros((n, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ms = measure._timed(torch, e, dev, lambda: e.search_batch_dev(planes[1], planes[0], d_jobs.data_ptr(), n, d_res.data_ptr(), 38,
                                                                  METHODS[args.method], args.subme, max_pu_dim=64), 5)
    res = d_res.cpu().numpy().view(RESULT_DTYPE).reshape(-1)
    same = all(np.array_equal(res[k], want[k]) for k in ("mv", "cost", "sad_calls", "subpel_calls"))
    print("%2d-bit 3840x2160 %s subme %d, %d PUs: one mixed launch %.3f ms, four per-size launches %.3f ms, records %s"
          % (args.depth, args.method, args.subme, n, ms, per["ms"], "equal" if same else "DIFFER"))
    sys.exit(0 if same else 1)


if __name__ == "__main__":
    main()
