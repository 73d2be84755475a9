"""Timed runs of the secondary paths (exhaustive search, Rockchip CIME/RIME/FME) on one GPU, shared by
bench.py and the tools/ scripts. Inputs are synthetic and resident in HBM; times come from HIP events
recorded on the engine's own stream."""
import numpy as np

from . import workload
from .engine import METHODS, RESULT_DTYPE, RK_RESULT_DTYPE
from .synth import make_clip, pad_plane

SAD_ISSUE_PEAK = 150e12   # pixel-SADs/s of v_sad_u8 / v_qsad_pk_u16_u8 on all 1024 SIMDs (tools/sad_rate.hip, measured);
                          # v_sad_u16 measures two samples where those measure four: half of it for 16-bit samples


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
    clip = make_clip(width, height, 2, seed=seed, depth=e.depth)
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


def esa_ctu_picture(e, dev, width=3840, height=2160, merange=92, subme=5, reps=3, seed=99, qp=32, spread=12, compare=False):
    """--me full over every square PU of one picture against one reference through the CTU scan (x265amd_esa_ctu_batch_dev:
    the 8x8 SADs measured once per CTU and summed up the quadtree), the same picture, stand-in lookahead field
    (`spread`: its MVs lie in +-spread lowres quarter-pels) and jobs as esa_per_size.
    Returns {ms, pus, ctus, pixel_sads, jobs, results[, equal]}; compare: the per-size launches' records are compared too."""
    import torch
    M = 128
    clip = make_clip(width, height, 2, seed=seed, depth=e.depth)
    cur, org, stride = pad_plane(clip[1], M)
    ref, _, _ = pad_plane(clip[0], M)
    pc = e.upload_plane(cur, width, height, M, M, stride)
    pr = e.upload_plane(ref, width, height, M, M, stride)
    rng = np.random.default_rng(5)
    wcu, hcu = (width // 2 + 7) // 8, (height // 2 + 7) // 8
    la = rng.integers(-spread, spread + 1, (wcu * hcu, 2)).astype(np.int16)
    jobs = workload.square_pu_jobs(width, height, la, wcu, hcu, merange)
    cj, index, nctu = workload.esa_ctu_jobs(jobs, width, height)
    n = len(cj)
    d_jobs = torch.from_numpy(cj.view(np.uint8).reshape(n, -1).copy()).to(dev)
    d_res = torch.zeros((n, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ms = _timed(torch, e, dev, lambda: e.esa_ctu_batch_dev(pc, pr, d_jobs.data_ptr(), nctu, d_res.data_ptr(), qp, subme), reps)
    res = d_res.cpu().numpy().view(RESULT_DTYPE).reshape(-1)
    have = index >= 0
    out = {"ms": ms, "pus": int(have.sum()), "ctus": nctu, "pixel_sads": workload.pixel_sads(cj[have], res[have]),
           "jobs": cj[have], "results": res[have], "planes": (cur, ref, org, stride)}
    if compare:
        m = len(jobs)
        dj = torch.from_numpy(jobs.view(np.uint8).reshape(m, -1).copy()).to(dev)
        dr = torch.zeros((m, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        first = 0
        for s in (64, 32, 16, 8):
            cnt = int((jobs["w"] == s).sum())
            e.search_batch_dev(pc, pr, dj.data_ptr() + 40 * first, cnt, dr.data_ptr() + 16 * first, qp, METHODS["full"], subme, max_pu_dim=s)
            first += cnt
        e.sync()
        per = dr.cpu().numpy().view(RESULT_DTYPE).reshape(-1)
        out["equal"] = bool(np.array_equal(res[have], per[index[have]]))
    e.free_plane(pc)
    e.free_plane(pr)
    return out


def rk_picture(e, dev, width=1920, height=1080, reps=5, seed=5, qp=30):
    """CIME + RIME + FME of every CTU of one picture against one reference, default RK search range.
    Returns a dict with ms, ctus, pixel_sads and the inputs / results for checking."""
    import torch
    M = 304
    clip = make_clip(width, height, 3, seed=seed, depth=e.depth)
    cur, org, stride = pad_plane(clip[2], M)
    rorig, _, _ = pad_plane(clip[0], M)
    noise = np.random.default_rng(seed).integers(-2, 3, clip[0].shape)
    recon, _, _ = pad_plane(np.clip(clip[0].astype(int) + noise, 0, (1 << e.depth) - 1).astype(clip[0].dtype), M)
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


def lookahead_costs(e, dev, width=1920, height=1080, nframes=16, reps=3, seed=21, qp=12):
    """estimateFrameCost of `nframes` B-frame triples (b - 1, b, b + 1) of one clip: 2 * nframes MV fields (HEX,
    subme 1, merange 16), then the intra / bidirectional / choice kernels, everything device resident.
    Returns the per-stage times (ms per batch) and the first triple's results with its host planes for checking."""
    import ctypes as C
    import torch
    from .engine import FrameCost, HEX
    from .synth import lowres_dims
    M = 64
    clip = make_clip(width, height, nframes + 2, seed=seed)
    low = []
    for f in clip:
        buf, org, stride = pad_plane(f, M)
        p = e.upload_plane(buf, width, height, M, M, stride)
        low.append(e.lowres_init(p))
        e.free_plane(p)
    lw, lh, wcu, hcu = lowres_dims(width, height)
    ncu = wcu * hcu
    n = nframes
    fields = []
    for b in range(1, n + 1):
        fields += [(low[b][0], low[b - 1]), (low[b][0], low[b + 1])]
    d_fields = e.fields_upload(fields, wcu, hcu, HEX, 16)
    d_mvs = torch.zeros((n, 2, ncu, 2), dtype=torch.int16, device=dev)
    d_mvc = torch.zeros((n, 2, ncu), dtype=torch.int32, device=dev)
    out = {"intra_cost": torch.zeros((n, ncu), dtype=torch.int32, device=dev),
           "lowres_costs": torch.zeros((n, ncu), dtype=torch.int16, device=dev),
           "row_satds_intra": torch.zeros((n, hcu), dtype=torch.int32, device=dev),
           "row_satds": torch.zeros((n, hcu), dtype=torch.int32, device=dev),
           "totals": torch.zeros((n, 5), dtype=torch.int64, device=dev)}
    descs = []
    for i in range(n):
        d = FrameCost()
        d.fenc = low[i + 1][0]
        for q in range(4):
            d.ref0[q] = low[i][q]
            d.ref1[q] = low[i + 2][q]
        d.d0 = d.d1 = 1
        d.mvs0, d.mvs1 = d_mvs[i, 0].data_ptr(), d_mvs[i, 1].data_ptr()
        d.mv_costs0, d.mv_costs1 = d_mvc[i, 0].data_ptr(), d_mvc[i, 1].data_ptr()
        for k, t in out.items():
            setattr(d, k, t[i].data_ptr())
        descs.append(d)
    d_desc = e.frame_costs_upload(descs, wcu, hcu)
    torch.cuda.synchronize()
    search = lambda: e.lookahead_fields_dev(d_fields, 2 * n, wcu, hcu, qp, HEX, 1, 16, d_mvs.data_ptr(), d_mvc.data_ptr())
    costs = lambda: e.lookahead_frame_costs_dev(d_desc, n, wcu, hcu)
    ms_search = _timed(torch, e, dev, search, reps)
    ms_costs = _timed(torch, e, dev, costs, reps)
    res = {k: t[0].cpu().numpy() for k, t in out.items()}
    res["lowres_costs"] = res["lowres_costs"].view(np.uint16)
    res["mvs"] = d_mvs[0].cpu().numpy()
    res["mv_costs"] = d_mvc[0].cpu().numpy()
    host = [[e.download_plane(p).reshape(-1) for p in low[k]] for k in range(3)]
    geo = (low[0][0].margin_y * low[0][0].stride + low[0][0].margin_x, low[0][0].stride, wcu, hcu)
    e.device_free(d_fields)
    e.device_free(d_desc)
    for l in low:
        for p in l:
            e.free_plane(p)
    return {"frames": n, "fields": 2 * n, "cus": ncu, "ms_search": ms_search, "ms_costs": ms_costs, "first": res,
            "host_planes": host, "geometry": geo}


def pattern_search_picture(e, dev, width=3840, height=2160, depth=10, method="star", subme=5, merange=57, reps=5, seed=31, qp=38):
    """Pattern search + sub-pel refinement of every square 64/32/16/8 PU of one picture against one reference in
    one launch per PU size (as the headline does), for an engine of `depth` bits. Predictors come from
    the lookahead field of the same pair. Returns ms per picture-reference, the pixel-SAD count and jobs / results."""
    import torch
    from .engine import HEX, SQUARE_ONLY
    from .synth import lowres_dims
    M = 128
    clip = make_clip(width, height, 2, seed=seed, depth=depth)
    planes = []
    for f in clip:
        buf, org, stride = pad_plane(f, M)
        planes.append(e.upload_plane(buf, width, height, M, M, stride))
    low = [e.lowres_init(p) for p in planes]
    lw, lh, wcu, hcu = lowres_dims(width, height)
    la_mvs, _, _ = e.lookahead_fields([(low[1][0], low[0])], wcu, hcu, 12 + 6 * (depth - 8), HEX, 1, 16)
    jobs = workload.square_pu_jobs(width, height, la_mvs[0], wcu, hcu, merange)      # sorted 64, 32, 16, 8
    n = len(jobs)
    classes = []
    for s_ in (64, 32, 16, 8):
        idx = np.flatnonzero(jobs["w"] == s_)
        classes.append((int(idx[0]), len(idx), s_))
    d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(n, -1).copy()).to(dev)
    d_res = torch.zeros((n, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def launch():
        for (first, cnt, dim) in classes:
            e.search_batch_dev(planes[1], planes[0], d_jobs.data_ptr() + first * 40, cnt, d_res.data_ptr() + first * 16, qp,
                               METHODS[method], subme, max_pu_dim=dim | SQUARE_ONLY)      # (a list per size: nothing but its squares)
    ms = _timed(torch, e, dev, launch, reps)
    res = d_res.cpu().numpy().view(RESULT_DTYPE).reshape(-1)
    for l in low:
        for p in l:
            e.free_plane(p)
    for p in planes:
        e.free_plane(p)
    return {"ms": ms, "pus": n, "pixel_sads": workload.pixel_sads(jobs, res), "jobs": jobs, "results": res}


def pattern_search_picture_families(e, e2, dev, width=3840, height=2160, depth=10, method="star", subme=5, merange=57, reps=5, seed=31, qp=38):
    """The searches of pattern_search_picture launched the way the headline launches them: the 16x16 PUs with the list of
    their 8x8 children as one family launch on engine `e`, the 64x64 PUs each followed by their 32x32 children on engine
    `e2` beside it (two contexts, so the two launches share the chip). Times a picture from the first launch to the end
    of both (host clock around `reps` pictures, every picture waited for). Returns ms, pixel-SADs, jobs / results."""
    import time
    import torch
    from .engine import HEX, SQUARE_ONLY
    from .synth import lowres_dims
    M = 128
    clip = make_clip(width, height, 2, seed=seed, depth=depth)
    planes = []
    for f in clip:
        buf, org, stride = pad_plane(f, M)
        planes.append(e.upload_plane(buf, width, height, M, M, stride))
    low = [e.lowres_init(p) for p in planes]
    lw, lh, wcu, hcu = lowres_dims(width, height)
    la_mvs, _, _ = e.lookahead_fields([(low[1][0], low[0])], wcu, hcu, 12 + 6 * (depth - 8), HEX, 1, 16)
    host_jobs = [workload.square_pu_jobs(width, height, la_mvs[0], wcu, hcu, merange)]
    d_pairs = e.pairs_upload([(planes[1], planes[0])])
    bj, bi = workload.class_job_list(host_jobs, (64, 32), families=True)
    pj, pi, kj, ki = workload.family_lists(host_jobs, 16)
    sj, si, npar = np.concatenate([pj, kj]), np.concatenate([pi, ki]), len(pj)
    up = lambda j: torch.from_numpy(j.view(np.uint8).reshape(len(j), -1).copy()).to(dev)
    d_bj, d_sj = up(bj), up(sj)
    d_br = torch.zeros((len(bj), RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    d_sr = torch.zeros((len(sj), RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def picture():
        e2.search_batch_pairs_dev(d_pairs, 1, d_bj.data_ptr(), len(bj), d_br.data_ptr(), qp, METHODS[method], subme, max_pu_dim=64 | SQUARE_ONLY)
        e.search_batch_pairs_family_dev(d_pairs, 1, d_sj.data_ptr(), npar, d_sr.data_ptr(), d_sj.data_ptr() + 40 * npar, len(sj) - npar,
                                        d_sr.data_ptr() + 16 * npar, qp, subme, METHODS[method])
        e.sync()
        e2.sync()
    picture()
    t0 = time.perf_counter()
    for _ in range(reps):
        picture()
    ms = (time.perf_counter() - t0) / reps * 1e3
    res = np.zeros(len(host_jobs[0]), RESULT_DTYPE)
    res[bi[:, 1]] = d_br.cpu().numpy().view(RESULT_DTYPE).reshape(-1)
    res[si[:, 1]] = d_sr.cpu().numpy().view(RESULT_DTYPE).reshape(-1)
    e.device_free(d_pairs)
    for l in low:
        for p in l:
            e.free_plane(p)
    for p in planes:
        e.free_plane(p)
    return {"ms": ms, "pus": len(res), "pixel_sads": workload.pixel_sads(host_jobs[0], res), "jobs": host_jobs[0], "results": res,
            "planes": clip}


def mixed_partition(rng, W, H, dtype):
    """A frame's final partitioning as mode decision leaves it: every 64x64 CTU split at random down the quadtree, leaf CUs
    as 2Nx2N, 2NxN, Nx2N or an asymmetric pair."""
    pus = []

    def leaf(x, y, s):
        m = int(rng.integers(0, 6)) if s > 8 else int(rng.integers(0, 3))
        if m == 0 or m > 4:
            pus.append((x, y, s, s))
        elif m == 1:
            pus.extend([(x, y, s, s // 2), (x, y + s // 2, s, s // 2)])
        elif m == 2:
            pus.extend([(x, y, s // 2, s), (x + s // 2, y, s // 2, s)])
        elif m == 3:
            pus.extend([(x, y, s, s // 4), (x, y + s // 4, s, 3 * s // 4)])
        else:
            pus.extend([(x, y, 3 * s // 4, s), (x + 3 * s // 4, y, s // 4, s)])

    def cu(x, y, s):
        if s > 8 and rng.random() < 0.55:
            for dy in (0, s // 2):
                for dx in (0, s // 2):
                    cu(x + dx, y + dy, s // 2)
        else:
            leaf(x, y, s)

    for y in range(0, H - 63, 64):
        for x in range(0, W - 63, 64):
            cu(x, y, 64)
    jobs = np.zeros(len(pus), dtype)
    a = np.array(pus)
    jobs["pu_x"], jobs["pu_y"], jobs["w"], jobs["h"] = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    return jobs


def mc_picture(e, dev, width=3840, height=2160, reps=10, seed=5):
    """TComPrediction::motionCompensation of every PU of one picture (a mixed quadtree partitioning, luma + 4:2:0 chroma,
    random quarter-pel MVs) from device-resident pictures and job lists; uni- and bi-predicted."""
    import torch
    from .engine import MC_JOB_DTYPE
    from .synth import make_chroma
    W, H, M = width, height, 304
    bpp = 1 if e.depth <= 8 else 2
    ys = make_clip(W, H, 2, seed=seed, depth=e.depth)
    cbs, crs = make_chroma(W, H, 2, seed=seed, depth=e.depth)
    dims = ((W, H, M), (W // 2, H // 2, M // 2), (W // 2, H // 2, M // 2))
    dpics = []
    for i in range(2):
        planes = [pad_plane(ys[i], M), pad_plane(cbs[i], M // 2), pad_plane(crs[i], M // 2)]
        dpics.append(e.upload_picture([(b, o, s, w, h, m) for (b, o, s), (w, h, m) in zip(planes, dims)]))
    dpred = e.upload_picture([(np.zeros(w * h, e.dtype), 0, w, w, h, 0) for (w, h, _) in dims])
    d_planes = e.pictures_upload(dpics)
    rng = np.random.default_rng(seed)
    out = {}
    for name, dir_ in (("uni", 1), ("bi", 3)):
        jobs = mixed_partition(rng, W, H, MC_JOB_DTYPE)
        jobs["dir"], jobs["flags"], jobs["ref0"], jobs["ref1"] = dir_, 3, 0, 1
        jobs["mv0"] = rng.integers(-64, 65, (len(jobs), 2))
        jobs["mv1"] = rng.integers(-64, 65, (len(jobs), 2))
        lst = e.mc_list_create(dpics, dpred, jobs)
        ms = _timed(torch, e, dev, lambda: e.mc_batch_dev(d_planes, dpred, lst), reps)
        e.mc_list_free(lst)
        jw, jh = jobs["w"].astype(np.int64), jobs["h"].astype(np.int64)
        lists = 2 if dir_ == 3 else 1
        nbytes = int((bpp * (lists * ((jw + 7) * (jh + 7) + 2 * (jw // 2 + 3) * (jh // 2 + 3)) + jw * jh * 3 // 2)).sum())
        out[name] = {"ms": ms, "pus": len(jobs), "algorithmic_bytes": nbytes, "samples": int((jw * jh).sum()) * 3 // 2}
    e.device_free(d_planes)
    e.free_picture(dpred)
    for p_ in dpics:
        e.free_picture(p_)
    return out


def bipred_picture(e, dev, width=3840, height=2160, reps=10, seed=6):
    """The bi-prediction estimate of every PU of one picture (mixed quadtree partitioning, random quarter-pel MVs) as a
    device-resident list (x265amd_bipred_batch_list)."""
    import torch
    from .engine import BIPRED_JOB_DTYPE, BIPRED_RESULT_DTYPE
    W, H, M = width, height, 96
    bpp = 1 if e.depth <= 8 else 2
    clip = make_clip(W, H, 3, seed=seed, depth=e.depth)
    planes = []
    for f in clip:
        buf, _, stride = pad_plane(f, M)
        planes.append(e.upload_plane(buf, W, H, M, M, stride))
    rng = np.random.default_rng(seed)
    jobs = mixed_partition(rng, W, H, BIPRED_JOB_DTYPE)
    jobs["mv0"] = rng.integers(-64, 65, (len(jobs), 2))
    jobs["mv1"] = rng.integers(-64, 65, (len(jobs), 2))
    lst = e.bipred_list_create(planes[1], planes[0], planes[2], jobs)
    d_out = e.device_alloc(len(jobs) * BIPRED_RESULT_DTYPE.itemsize)
    ms = _timed(torch, e, dev, lambda: e.bipred_batch_list(planes[1], planes[0], planes[2], lst, d_out), reps)
    e.bipred_list_free(lst)
    e.device_free(d_out)
    for p_ in planes:
        e.free_plane(p_)
    jw, jh = jobs["w"].astype(np.int64), jobs["h"].astype(np.int64)
    return {"ms": ms, "pus": len(jobs), "algorithmic_bytes": int((bpp * (jw * jh + 2 * (jw + 7) * (jh + 7)) + 8).sum())}



def pattern_search_clip(e, dev, kind, pairs=8, width=1920, height=1080, method="star", subme=5, merange=57, reps=3, seed=1234, qp=32):
    """The headline's searches on another kind of clip (synth.make_clip: "noise" = no two frames alike, the worst case of
    a pattern search; "static" = every frame the same, the best case; "texture" = the headline's translated texture):
    every square 64/32/16/8 PU of `pairs` frame pairs, predictors from the lookahead field of each pair, one launch per
    PU class over all pairs (the 64x64 PUs with their 32x32 children behind them, the 16x16 PUs with their 8x8 children), the
    classes one after the other on one stream. Per class: ms, pixel-SADs, and the share of its PUs the fixed-shape kernels did not finish (marked).
    Returns the classes and, for checking, the jobs and results of pair 0."""
    import torch
    from .engine import HEX, SQUARE_ONLY
    from .synth import lowres_dims
    M = 128
    clip = make_clip(width, height, pairs + 1, seed=seed, kind=kind)
    planes = []
    for f in clip:
        buf, org, stride = pad_plane(f, M)
        planes.append(e.upload_plane(buf, width, height, M, M, stride))
    low = [e.lowres_init(p) for p in planes]
    lw, lh, wcu, hcu = lowres_dims(width, height)
    la_mvs, _, _ = e.lookahead_fields([(low[i + 1][0], low[i]) for i in range(pairs)], wcu, hcu, 12, HEX, 1, 16)
    host_jobs = [workload.square_pu_jobs(width, height, la_mvs[i], wcu, hcu, merange) for i in range(pairs)]
    d_pairs = e.pairs_upload([(planes[i + 1], planes[i]) for i in range(pairs)])
    out, keep = {}, []
    for name, sizes, dim in (("64+32", (64, 32), 64), ("16+8", (16, 8), 16)):
        if dim == 64:
            cj, index = workload.class_job_list(host_jobs, sizes, families=True)
            nparents = 0
        else:
            pj, pi, kj, ki = workload.family_lists(host_jobs, 16)
            cj, index, nparents = np.concatenate([pj, kj]), np.concatenate([pi, ki]), len(pj)
        n = len(cj)
        d_jobs = torch.from_numpy(cj.view(np.uint8).reshape(n, -1).copy()).to(dev)
        d_res = torch.zeros((n, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        if dim == 64:
            launch = lambda: e.search_batch_pairs_dev(d_pairs, pairs, d_jobs.data_ptr(), n, d_res.data_ptr(), qp, METHODS[method], subme,
                                                      max_pu_dim=dim | SQUARE_ONLY)
        else:
            launch = lambda: e.search_batch_pairs_family_dev(d_pairs, pairs, d_jobs.data_ptr(), nparents, d_res.data_ptr(),
                                                             d_jobs.data_ptr() + 40 * nparents, n - nparents,
                                                             d_res.data_ptr() + 16 * nparents, qp, subme, METHODS[method])
        ms = _timed(torch, e, dev, launch, reps)
        marked = e.marked_count()
        res = d_res.cpu().numpy().view(RESULT_DTYPE).reshape(-1)
        out[name] = {"pus": n, "ms": ms, "marked_frac": marked / (nparents or n), "pixel_sads": workload.pixel_sads(cj, res)}
        sel = index[:, 0] == 0
        keep.append((cj[sel], res[sel]))
    e.device_free(d_pairs)
    for l in low:
        for p in l:
            e.free_plane(p)
    for p in planes:
        e.free_plane(p)
    return out, (clip[0], clip[1], np.concatenate([k[0] for k in keep]), np.concatenate([k[1] for k in keep]))


def other_partitions_clip(e, dev, pairs=8, width=1920, height=1080, method="star", subme=5, merange=57, reps=3, seed=1234, qp=32):
    """The 2NxN and Nx2N partitions of every CU size of `pairs` frame pairs of the headline's clip: per CU size one launch
    over all pairs holding both shapes mixed (64x32 + 32x64 | 32x16 + 16x32 | 16x8 + 8x16 | 8x4 + 4x8), predictors from
    the lookahead field of each pair. The two small classes run in the kernels that pack several PUs into a wave, the two
    large ones a wave per PU. Returns per class PUs, ms and pixel-SADs and, for checking, the jobs and results of pair 0."""
    import torch
    from .engine import HEX
    from .synth import lowres_dims
    M = 128
    clip = make_clip(width, height, pairs + 1, seed=seed)
    planes = []
    for f in clip:
        buf, org, stride = pad_plane(f, M)
        planes.append(e.upload_plane(buf, width, height, M, M, stride))
    low = [e.lowres_init(p) for p in planes]
    lw, lh, wcu, hcu = lowres_dims(width, height)
    la_mvs, _, _ = e.lookahead_fields([(low[i + 1][0], low[i]) for i in range(pairs)], wcu, hcu, 12, HEX, 1, 16)
    d_pairs = e.pairs_upload([(planes[i + 1], planes[i]) for i in range(pairs)])
    out, keep = {}, []
    for cu in (64, 32, 16, 8):
        per_pair = []
        for i in range(pairs):
            # the CUs of a picture in raster order, both splits of a CU side by side as a caller deciding between them has them
            a = workload.shape_jobs(width, height, cu, cu // 2, la_mvs[i], wcu, hcu, merange, cu)
            b = workload.shape_jobs(width, height, cu // 2, cu, la_mvs[i], wcu, hcu, merange, cu)
            n2 = len(a) // 2
            mixed = np.stack([a[:n2], a[n2:], b[:n2], b[n2:]], axis=1).reshape(-1)
            mixed["reserved"] = i
            per_pair.append(mixed)
        cj = np.concatenate(per_pair)
        n = len(cj)
        d_jobs = torch.from_numpy(cj.view(np.uint8).reshape(n, -1).copy()).to(dev)
        d_res = torch.zeros((n, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ms = _timed(torch, e, dev, lambda: e.search_batch_pairs_dev(d_pairs, pairs, d_jobs.data_ptr(), n, d_res.data_ptr(), qp, METHODS[method], subme,
                                                                    max_pu_dim=cu), reps)
        res = d_res.cpu().numpy().view(RESULT_DTYPE).reshape(-1)
        out["%dx%d+%dx%d" % (cu, cu // 2, cu // 2, cu)] = {"pus": n, "ms": ms, "pixel_sads": workload.pixel_sads(cj, res)}
        keep.append((per_pair[0], res[:len(per_pair[0])]))
    e.device_free(d_pairs)
    for l in low:
        for p in l:
            e.free_plane(p)
    for p in planes:
        e.free_plane(p)
    return out, (clip[0], clip[1], np.concatenate([k[0] for k in keep]), np.concatenate([k[1] for k in keep]))


def ctu_row_wavefront(e, e2, dev, pairs=32, width=1920, height=1080, method="star", subme=5, merange=57, reps=3, seed=1234, qp=32):
    """The headline's searches in the shape an encoder can batch them: a CTU row needs the MVs of the row above before its
    predictors are known (predInterSearch takes them from the neighbours' final MVs, TEncSearch.cpp:4268-4292), so the
    legal batch is one CTU row of every frame pair in flight. Per row two launches side by side, as the headline's (the
    64x64 PUs with their 32x32 children on `e`, the 16x16 PUs with the list of their 8x8 children on `e2`), both waited
    for before the next row starts; the predictors still come from the lookahead field (the dependency is kept as a
    barrier, not as data). Returns ms for all rows of all pairs, the PU and pixel-SAD counts and the jobs / results of pair 0."""
    import torch
    from .engine import HEX, SQUARE_ONLY
    from .synth import lowres_dims
    M = 128
    clip = make_clip(width, height, pairs + 1, seed=seed)
    planes = []
    for f in clip:
        buf, org, stride = pad_plane(f, M)
        planes.append(e.upload_plane(buf, width, height, M, M, stride))
    low = [e.lowres_init(p) for p in planes]
    lw, lh, wcu, hcu = lowres_dims(width, height)
    la_mvs, _, _ = e.lookahead_fields([(low[i + 1][0], low[i]) for i in range(pairs)], wcu, hcu, 12, HEX, 1, 16)
    host_jobs = [workload.square_pu_jobs(width, height, la_mvs[i], wcu, hcu, merange) for i in range(pairs)]
    d_pairs = e.pairs_upload([(planes[i + 1], planes[i]) for i in range(pairs)])
    rows = []
    nrows = (height + 63) // 64
    for r in range(nrows):
        sel = [np.flatnonzero(j["pu_y"] // 64 == r) for j in host_jobs]
        row_jobs = [j[s_] for j, s_ in zip(host_jobs, sel)]
        big, big_index = workload.class_job_list(row_jobs, (64, 32), families=True)
        pj, pi, kj, ki = workload.family_lists(row_jobs, 16)
        fam, fam_index, npar = np.concatenate([pj, kj]), np.concatenate([pi, ki]), len(pj)
        dev_lists = []
        for cj in (big, fam):
            n = len(cj)
            dev_lists.append((torch.from_numpy(cj.view(np.uint8).reshape(n, -1).copy()).to(dev),
                              torch.zeros((n, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev), n))
        rows.append((dev_lists, npar, big, big_index, fam, fam_index, sel))
    torch.cuda.synchronize()

    def step():
        for (dev_lists, npar, *_rest) in rows:
            (bj, br, bn), (fj, fr, fn) = dev_lists
            if bn:
                e.search_batch_pairs_dev(d_pairs, pairs, bj.data_ptr(), bn, br.data_ptr(), qp, METHODS[method], subme, max_pu_dim=64 | SQUARE_ONLY)
            e2.search_batch_pairs_family_dev(d_pairs, pairs, fj.data_ptr(), npar, fr.data_ptr(), fj.data_ptr() + 40 * npar, fn - npar,
                                             fr.data_ptr() + 16 * npar, qp, subme, METHODS[method])
            e.sync()
            e2.sync()

    import time
    step()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    ms = (time.perf_counter() - t0) / reps * 1e3
    pus = sads = 0
    keep_j, keep_r = [], []
    for (dev_lists, npar, big, big_index, fam, fam_index, sel) in rows:
        for (dj, dr, n), cj, index in zip(dev_lists, (big, fam), (big_index, fam_index)):
            if not n:
                continue
            res = dr.cpu().numpy().view(RESULT_DTYPE).reshape(-1)
            pus += n
            sads += workload.pixel_sads(cj, res)
            first = index[:, 0] == 0
            keep_j.append(cj[first])
            keep_r.append(res[first])
    e.device_free(d_pairs)
    for l in low:
        for p in l:
            e.free_plane(p)
    for p in planes:
        e.free_plane(p)
    return {"ms": ms, "rows": nrows, "pus": pus, "pixel_sads": sads, "frames": (clip[0], clip[1]),
            "jobs": np.concatenate(keep_j), "results": np.concatenate(keep_r)}
