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
