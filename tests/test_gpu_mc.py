"""GPU: motion compensation into a prediction picture, luma + 4:2:0 chroma (x265amd_mc_batch), against the digests recorded
from the reference's primitives and against the oracle."""
import ctypes as C

import numpy as np
import pytest

from x265_amd.engine import MC_JOB_DTYPE, MC_WEIGHT_DTYPE
import golden_util as G
from util import MC_W, MC_H, MC_MARGIN, mc_pictures, mc_blank_pred, mc_job_batches, mc_weights

pytestmark = pytest.mark.gpu


def upload_pictures(e, pics, margin=MC_MARGIN):
    dims = ((MC_W, MC_H, margin), (MC_W // 2, MC_H // 2, margin // 2), (MC_W // 2, MC_H // 2, margin // 2))
    return [e.upload_picture([(b, o, s, w, h, m) for (b, o, s), (w, h, m) in zip(pic, dims)]) for pic in pics]


def download_picture(e, pic):
    return [e.download_plane(p).reshape(-1) for p in (pic.y, pic.cb, pic.cr)]


def run_batch(e, drefs, jobs, depth, weights=None):
    """One batch into a fresh sentinel-filled prediction picture; returns its three planes."""
    dpred = upload_pictures(e, [mc_blank_pred(depth)], margin=0)[0]
    e.mc_batch(drefs, dpred, jobs, weights)
    out = download_picture(e, dpred)
    e.free_picture(dpred)
    return out


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("depth", [8, 10])
def test_mc_golden(request, depth, weighted):
    e = request.getfixturevalue("engine8" if depth == 8 else "engine10")
    refs, batches, wts = G.mc_case(depth, weighted)
    drefs = upload_pictures(e, refs)
    exp = G.mc_golden()["%d%s" % (depth, "w" if weighted else "")]
    for b, jobs in enumerate(batches):
        got = run_batch(e, drefs, jobs, depth, wts)
        for k in range(3):
            assert np.array_equal(G.sha_bytes(got[k]), exp[b, k]), "batch %d plane %d" % (b, k)
    for p in drefs:
        e.free_picture(p)


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("depth", [8, 10])
def test_mc_vs_oracle(request, oracle, depth, weighted):
    """Other seeds than the golden case, sample-by-sample against the oracle (names the first differing job), samples outside
    the PUs untouched; then the device-resident entry point, the empty batch and rejected jobs."""
    e = request.getfixturevalue("engine8" if depth == 8 else "engine10")
    refs = mc_pictures(depth, 41)
    drefs = upload_pictures(e, refs)
    rng = np.random.default_rng(42)
    batches = mc_job_batches(rng, MC_JOB_DTYPE, 30, weighted=weighted)
    wts = mc_weights(rng, MC_WEIGHT_DTYPE, 2) if weighted else None
    for b, jobs in enumerate(batches):
        exp = mc_blank_pred(depth)
        oracle.mc_batch(depth, refs, exp, jobs, wts)
        got = run_batch(e, drefs, jobs, depth, wts)
        for k in range(3):
            if not np.array_equal(got[k], exp[k][0]):
                w = MC_W if k == 0 else MC_W // 2
                idx = int(np.flatnonzero(got[k] != exp[k][0])[0])
                y, x = divmod(idx, w)
                sc = 1 if k == 0 else 2
                hit = [dict(zip(j.dtype.names, j.tolist())) for j in jobs
                       if j["pu_x"] <= x * sc < j["pu_x"] + j["w"] and j["pu_y"] <= y * sc < j["pu_y"] + j["h"]]
                raise AssertionError("batch %d plane %d first difference at (%d, %d): got %d, expected %d; job %s"
                                     % (b, k, x, y, got[k][idx], exp[k][0][idx], hit))
    # device-resident jobs, asynchronous
    jobs = batches[3]
    d_planes = e.pictures_upload(drefs, wts)
    dpred = upload_pictures(e, [mc_blank_pred(depth)], margin=0)[0]
    lst = e.mc_list_create(drefs, dpred, jobs, wts)
    e.mc_batch_dev(d_planes, dpred, lst)
    e._chk(e.L.x265amd_sync(e.ctx))
    exp = mc_blank_pred(depth)
    oracle.mc_batch(depth, refs, exp, jobs, wts)
    for k, got in enumerate(download_picture(e, dpred)):
        assert np.array_equal(got, exp[k][0])
    e.mc_list_free(lst)
    # empty batch leaves the picture alone
    e.mc_batch(drefs, dpred, jobs[:0])
    empty = e.mc_list_create(drefs, dpred, jobs[:0])
    e.mc_batch_dev(d_planes, dpred, empty)
    e.mc_list_free(empty)
    with pytest.raises(Exception):
        e.mc_list_create(drefs, dpred, np.zeros(1, MC_JOB_DTYPE))
    if not weighted:      # a weighted job without weights
        bad = jobs[:1].copy()
        bad["flags"] = 7
        with pytest.raises(Exception):
            e.mc_batch(drefs, dpred, bad)
    for bad_field, bad_value in (("w", 20), ("dir", 0), ("flags", 0), ("ref1", 7), ("pu_x", 2), ("mv0", (8 * (MC_W + MC_MARGIN), 0)),
                                 ("pu_y", MC_H)):
        bad = jobs[:1].copy()
        bad["dir"] = 3
        bad[bad_field] = bad_value
        with pytest.raises(Exception):
            e.mc_batch(drefs, dpred, bad, wts)
    e._chk(e.L.x265amd_device_free(e.ctx, C.c_void_p(d_planes)))
    e.free_picture(dpred)
    for p in drefs:
        e.free_picture(p)


def test_mc_full_size_properties(engine8):
    """3840x2160, every 32x32 PU of the picture in one launch. Whole-sample MVs (multiples of 8 quarter-pels, so chroma is
    whole-sample too) must reproduce the displaced reference exactly, uni-predicted; bi-predicting from one picture with
    equal MVs must give the same samples (addAvg of two equal 14-bit copies); and weights of exactly 1.0 with offset 0
    must change nothing, whatever the denominator."""
    from x265_amd.synth import make_clip, make_chroma, pad_plane
    from x265_amd.engine import MC_WEIGHT_DTYPE
    e = engine8
    W, H, M = 3840, 2160, 96
    y = make_clip(W, H, 1, seed=77)[0]
    cb, cr = (c[0] for c in make_chroma(W, H, 1, seed=77))
    dims = ((W, H, M), (W // 2, H // 2, M // 2), (W // 2, H // 2, M // 2))
    host = [pad_plane(y, M), pad_plane(cb, M // 2), pad_plane(cr, M // 2)]
    dref = e.upload_picture([(b, o, s, w, h, m) for (b, o, s), (w, h, m) in zip(host, dims)])
    dpred = e.upload_picture([(np.zeros(w * h, np.uint8), 0, w, w, h, 0) for (w, h, _) in dims])
    xs, ys = np.meshgrid(np.arange(0, W - 31, 32), np.arange(0, H - 31, 32))
    jobs = np.zeros(xs.size, MC_JOB_DTYPE)
    jobs["pu_x"], jobs["pu_y"], jobs["w"], jobs["h"] = xs.ravel(), ys.ravel(), 32, 32
    rng = np.random.default_rng(78)
    mv = rng.integers(-6, 7, (len(jobs), 2)) * 8
    jobs["mv0"], jobs["mv1"], jobs["flags"] = mv, mv, 3
    padded = [np.pad(p, m, mode="edge") for p, m in ((y, M), (cb, M // 2), (cr, M // 2))]
    exp = [np.zeros((H, W), np.uint8), np.zeros((H // 2, W // 2), np.uint8), np.zeros((H // 2, W // 2), np.uint8)]
    for j in jobs:
        px, py, dx, dy = int(j["pu_x"]), int(j["pu_y"]), int(j["mv0"][0]) // 4, int(j["mv0"][1]) // 4
        exp[0][py:py + 32, px:px + 32] = padded[0][M + py + dy:M + py + dy + 32, M + px + dx:M + px + dx + 32]
        for k in (1, 2):
            exp[k][py // 2:py // 2 + 16, px // 2:px // 2 + 16] = \
                padded[k][M // 2 + (py + dy) // 2:M // 2 + (py + dy) // 2 + 16, M // 2 + (px + dx) // 2:M // 2 + (px + dx) // 2 + 16]
    wts = np.zeros(1, MC_WEIGHT_DTYPE)
    wts["log2denom"], wts["w"] = (5, 3, 3), (32, 8, 8)
    for dir_, extra, w_ in ((1, 0, None), (3, 0, None), (1, 4, wts), (3, 4, wts)):
        jobs["dir"], jobs["flags"] = dir_, 3 | extra
        e.mc_batch([dref], dpred, jobs, w_)
        for k, p in enumerate((dpred.y, dpred.cb, dpred.cr)):
            got = e.download_plane(p)
            rows = (H // 32 * 32) >> (k > 0)      # the PUs cover 2144 of the 2160 luma rows
            assert np.array_equal(got[:rows], exp[k][:rows]), "dir %d flags %d plane %d" % (dir_, 3 | extra, k)
    e.free_picture(dpred)
    e.free_picture(dref)


def test_search_then_mc_on_device(engine8, oracle):
    """motionEstimate -> motionCompensation without a host round trip: two device-resident searches (two references) feed a
    bi-predicted MC launch through their result arrays; the prediction equals the oracle's for the downloaded MVs."""
    from x265_amd.engine import JOB_DTYPE, RESULT_DTYPE, STAR
    from x265_amd.synth import make_clip, make_chroma, pad_plane
    e = engine8
    W, H, M = 320, 192, 96
    ys = make_clip(W, H, 3, seed=91)
    cbs, crs = make_chroma(W, H, 3, seed=91)
    dims = ((W, H, M), (W // 2, H // 2, M // 2), (W // 2, H // 2, M // 2))
    host = [[pad_plane(ys[i], M), pad_plane(cbs[i], M // 2), pad_plane(crs[i], M // 2)] for i in range(3)]
    pics = [e.upload_picture([(b, o, s, w, h, m) for (b, o, s), (w, h, m) in zip(p, dims)]) for p in host]
    xs, yy = np.meshgrid(np.arange(0, W, 16), np.arange(0, H, 16))
    jobs = np.zeros(xs.size, JOB_DTYPE)
    jobs["pu_x"], jobs["pu_y"], jobs["w"], jobs["h"] = xs.ravel(), yy.ravel(), 16, 16
    jobs["mvmin"], jobs["mvmax"], jobs["merange"] = (-24, -24), (24, 24), 24
    d_jobs = e.device_alloc(jobs.nbytes)
    e.upload(d_jobs, jobs)
    d_res = [e.device_alloc(len(jobs) * RESULT_DTYPE.itemsize) for _ in range(2)]
    for k, ref in enumerate((0, 2)):        # frame 1 searched in frames 0 and 2
        e.search_batch_dev(pics[1].y, pics[ref].y, d_jobs, len(jobs), d_res[k], 30, STAR, 5)
    mj = np.zeros(len(jobs), MC_JOB_DTYPE)
    for f in ("pu_x", "pu_y", "w", "h"):
        mj[f] = jobs[f]
    mj["dir"], mj["flags"], mj["ref0"], mj["ref1"] = 3, 3, 0, 1
    refs = [pics[0], pics[2]]
    dpred = e.upload_picture([(np.zeros(w * h, np.uint8), 0, w, w, h, 0) for (w, h, _) in dims])
    lst = e.mc_list_create(refs, dpred, mj)
    d_planes = e.pictures_upload(refs)
    e.mc_batch_dev(d_planes, dpred, lst, d_res[0], d_res[1])
    e.sync()
    res = []
    for d in d_res:
        r = np.zeros(len(jobs), RESULT_DTYPE)
        e._chk(e.L.x265amd_download(e.ctx, C.c_void_p(d), C.c_void_p(r.ctypes.data), C.c_size_t(r.nbytes)))
        res.append(r)
    assert (res[0]["mv"] != 0).any() and (res[1]["mv"] != 0).any()
    mj["mv0"], mj["mv1"] = res[0]["mv"], res[1]["mv"]
    exp = [(np.zeros(w * h, np.uint8), 0, w) for (w, h, _) in dims]
    oracle.mc_batch(8, [host[0], host[2]], exp, mj)
    for got, (ex, _, _) in zip((dpred.y, dpred.cb, dpred.cr), exp):
        assert np.array_equal(e.download_plane(got).reshape(-1), ex)
    # the inter-mode distortion of xComputeCostInter (compress.cpp:209-212): sa8d of each source CU against the prediction
    # picture, taken through the merge entry point with the picture as reference and MV 0 (no margins needed)
    from x265_amd.engine import MERGE_JOB_DTYPE
    cj = np.zeros(len(jobs), MERGE_JOB_DTYPE)
    cj["pu_x"], cj["pu_y"], cj["size"], cj["dir"] = jobs["pu_x"], jobs["pu_y"], 16, 1
    got = e.merge_sa8d_batch(pics[1].y, [dpred.y], cj)
    fe, fo, fs = host[1][0]
    want = np.array([oracle.sa8d_inter(8, fe, fo + int(j["pu_y"]) * fs + int(j["pu_x"]), fs, exp[0][0], int(j["pu_y"]) * W + int(j["pu_x"]), W, 16, 16)
                     for j in cj], np.int32)
    assert np.array_equal(got, want) and got.min() >= 0 and got.max() > 0
    # the bi-prediction estimate of the same PUs, MVs again straight from the two result arrays
    from x265_amd.engine import BIPRED_JOB_DTYPE, BIPRED_RESULT_DTYPE
    bj = np.zeros(len(jobs), BIPRED_JOB_DTYPE)
    for f in ("pu_x", "pu_y", "w", "h"):
        bj[f] = jobs[f]
    bl = e.bipred_list_create(pics[1].y, pics[0].y, pics[2].y, bj)
    d_bi = e.device_alloc(len(bj) * BIPRED_RESULT_DTYPE.itemsize)
    e.bipred_batch_list(pics[1].y, pics[0].y, pics[2].y, bl, d_bi, d_res[0], d_res[1])
    e.bipred_batch_list(pics[1].y, pics[0].y, pics[2].y, bl, d_bi, d_res[0], d_res[1])      # a list can be launched again
    got_bi = e.download(d_bi, len(bj), BIPRED_RESULT_DTYPE)
    bj["mv0"], bj["mv1"] = res[0]["mv"], res[1]["mv"]
    exp_bi = oracle.bipred_batch(8, host[1][0][0], host[0][0][0], host[2][0][0], host[1][0][1], host[1][0][2], bj)
    assert np.array_equal(got_bi["satd"], exp_bi[:, 0]) and np.array_equal(got_bi["satd_zero"], exp_bi[:, 1])
    e.bipred_list_free(bl)
    e.device_free(d_bi)
    e.mc_list_free(lst)
    for d in [d_jobs, d_planes] + d_res:
        e.device_free(d)
    for p in pics + [dpred]:
        e.free_picture(p)


def test_mc_12bit_vs_oracle(oracle):
    """12-bit samples (headroom 2: the lossy end of the 14-bit intermediates), weighted and unweighted PUs mixed."""
    from x265_amd.engine import Engine
    e = Engine(0, 12)
    try:
        refs = mc_pictures(12, 51)
        drefs = upload_pictures(e, refs)
        rng = np.random.default_rng(52)
        wts = mc_weights(rng, MC_WEIGHT_DTYPE, 2)
        for jobs in mc_job_batches(rng, MC_JOB_DTYPE, 10, weighted=True):
            exp = mc_blank_pred(12)
            oracle.mc_batch(12, refs, exp, jobs, wts)
            got = run_batch(e, drefs, jobs, 12, wts)
            for k in range(3):
                assert np.array_equal(got[k], exp[k][0]), "plane %d" % k
        for p in drefs:
            e.free_picture(p)
    finally:
        e.close()

