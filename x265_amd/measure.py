"""Timed runs of the secondary paths (exhaustive search, Rockchip CIME/RIME/FME) on one GPU, shared by
bench.py and the tools/ scripts. Inputs are synthetic and resident in HBM; times come from HIP events
recorded on the engine's own stream."""
import numpy as np

from . import workload
from .engine import METHODS, RESULT_DTYPE, RK_RESULT_DTYPE
from .synth import make_clip, pad_plane

SAD_ISSUE_PEAK = 150e12   # pixel-SADs/s of v_sad_u8 / v_qsad_pk_u16_u8 on all 1024 SIMDs (tools/sad_rate.hip, measured)


def _timed(torch, e, dev, launch, reps):
    stream = torch.cuda.ExternalStream(e.stream(), device=dev)
    launch()
    e.sync()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(stream)
    for _ in range(reps):
        launch()
    ev1.record(stream)
    e.sync()
    return ev0.elapsed_time(ev1) / reps


def esa_per_size(e, dev, width=3840, height=2160, merange=92, sizes=(64, 32, 16, 8), subme=5, reps=3, seed=99, qp=32):
    """--me full over every square PU of one picture against one reference, one launch per PU size.
    Returns (rows, host planes); rows = [{size, pus, ms, pixel_sads, jobs, results}]."""
    import torch
    M = 128
    clip = make_clip(width, height, 2, seed=seed)
    cur, org, stride = pad_plane(clip[1], M)
    ref, _, _ = pad_plane(clip[0], M)
    pc = e.upload_plane(cur, width, height, M, M, stride)
    pr = e.upload_plane(ref, width, height, M, M, stride)
    rng = np.random.default_rng(5)
    wcu, hcu = (width // 2 + 7) // 8, (height // 2 + 7) // 8
    la = rng.integers(-12, 13, (wcu * hcu, 2)).astype(np.int16)      # stand-in lookahead field
    rows = []
    for s in sizes:
        jobs = workload.square_pu_jobs(width, height, la, wcu, hcu, merange, sizes=(s,))
        n = len(jobs)
        d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(n, -1).copy()).to(dev)
        d_res = torch.zeros((n, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ms = _timed(torch, e, dev, lambda: e.search_batch_dev(pc, pr, d_jobs.data_ptr(), n, d_res.data_ptr(), qp,
                                                              METHODS["full"], subme, max_pu_dim=s), reps)
        res = d_res.cpu().numpy().view(RESULT_DTYPE).reshape(-1)
        rows.append({"size": s, "pus": n, "ms": ms, "pixel_sads": workload.pixel_sads(jobs, res), "jobs": jobs, "results": res})
    e.free_plane(pc)
    e.free_plane(pr)
    return rows, (cur, ref, org, stride)


def rk_picture(e, dev, width=1920, height=1080, reps=5, seed=5, qp=30):
    """CIME + RIME + FME of every CTU of one picture against one reference, default RK search range.
    Returns a dict with ms, ctus, pixel_sads and the inputs / results for checking."""
    import torch
    M = 304
    clip = make_clip(width, height, 3, seed=seed)
    cur, org, stride = pad_plane(clip[2], M)
    rorig, _, _ = pad_plane(clip[0], M)
    noise = np.random.default_rng(seed).integers(-2, 3, clip[0].shape)
    recon, _, _ = pad_plane(np.clip(clip[0].astype(int) + noise, 0, 255).astype(np.uint8), M)
    pc = e.upload_plane(cur, width, height, M, M, stride)
    po = e.upload_plane(rorig, width, height, M, M, stride)
    pr = e.upload_plane(recon, width, height, M, M, stride)
    pds = e.rk_ds_plane(po)
    rng = np.random.default_rng(seed + 1)
    nctu = ((width + 63) // 64) * ((height + 63) // 64)
    pmvs = (rng.integers(-12, 13, (nctu, 2, 2)) * 2).astype(np.int16)
    mvps = (rng.integers(-12, 13, (nctu, 3, 2)) * 2).astype(np.int16)
    jobs = workload.rk_ctu_jobs(width, height, pmvs, mvps, have_mvd=True)
    d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(nctu, -1).copy()).to(dev)
    d_res = torch.zeros((nctu, RK_RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ms = _timed(torch, e, dev, lambda: e.rk_ctu_batch_dev(pc, pr, pds, qp, d_jobs.data_ptr(), nctu, d_res.data_ptr()), reps)
    res = d_res.cpu().numpy().view(RK_RESULT_DTYPE).reshape(-1)
    tdiv = lambda v: np.sign(v) * (np.abs(v) // 4)      # C division toward zero (motion.cpp:1280-1282)
    nx = tdiv(jobs["mvmax"][:, 0].astype(np.int64)) - tdiv(jobs["mvmin"][:, 0].astype(np.int64)) + 1
    ny = tdiv(jobs["mvmax"][:, 1].astype(np.int64)) - tdiv(jobs["mvmin"][:, 1].astype(np.int64)) + 1
    cime_px = int((nx * ny * (jobs["valid_w"].astype(np.int64) // 4) * (jobs["valid_h"].astype(np.int64) // 4)).sum())
    area = np.array([64] * 64 + [256] * 16 + [1024] * 4 + [4096], np.int64)
    cu_px = int(((res["rime"]["cost"] >= 0) * area[None, :]).sum())
    for p in (pc, po, pr):
        e.free_plane(p)
    return {"ms": ms, "ctus": nctu, "cime_pixel_sads": cime_px, "cu_pixel_sads": cu_px * (576 + 49),
            "pixel_sads": cime_px + cu_px * (576 + 49), "jobs": jobs, "results": res,
            "planes": (cur, rorig, recon, org, stride), "margin": M, "qp": qp}
