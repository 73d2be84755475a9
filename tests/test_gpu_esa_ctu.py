"""X265_FULL_SEARCH of whole CTUs (x265amd_esa_ctu_batch: one scan measures the 8x8 blocks of a CTU and sums them up the
quadtree for the 85 square PUs) against the CPU oracle and against the per-PU scans (motion.cpp:1021-1056 for both)."""
import numpy as np
import pytest

from util import clip_planes, oracle_search
from x265_amd import workload
from x265_amd.engine import EngineError, METHODS, make_jobs

pytestmark = pytest.mark.gpu


def _picture_jobs(W, H, rng, merange, spread, margin, own_ranges=False):
    """Square PUs of all four sizes of a W x H picture, predictors from a random field (`spread`: full-pel samples the
    predictors of neighbouring PUs differ by at most); own_ranges: every PU gets range extents of its own."""
    wcu, hcu = (W // 2 + 7) // 8, (H // 2 + 7) // 8
    la = (rng.integers(-spread, spread + 1, (wcu * hcu, 2)) * 2).astype(np.int16)
    jobs = workload.square_pu_jobs(W, H, la, wcu, hcu, merange, clip_margin=margin - 12)
    if own_ranges:
        for j in jobs:
            cx, cy = int(j["qmvp"][0]) >> 2, int(j["qmvp"][1]) >> 2
            ext = rng.integers(1, merange + 1, 4)
            j["mvmin"] = (max(cx - ext[0], j["mvmin"][0]), max(cy - ext[1], j["mvmin"][1]))
            j["mvmax"] = (min(cx + ext[2], j["mvmax"][0]), min(cy + ext[3], j["mvmax"][1]))
            j["qmvp"] = (int(j["qmvp"][0]) + int(rng.integers(-3, 4)), int(j["qmvp"][1]) + int(rng.integers(-3, 4)))
    return jobs


@pytest.mark.parametrize("depth", [8, 10, 12])
@pytest.mark.parametrize("kind,W,H,merange,spread,own", [
    ("texture", 256, 192, 20, 3, False),        # whole CTUs, predictors a few samples apart
    ("texture", 192, 160, 33, 0, False),        # the last CTU row is cut: 64x64 and lower 32x32 slots are empty
    ("static", 128, 128, 13, 2, True),          # every position the same SAD: table costs and raster order decide
    ("noise", 192, 128, 57, 6, True),           # ranges of their own per PU inside a wide box: several strips each way
])
def test_ctu_scan_vs_oracle_and_per_pu_scans(request, oracle, kind, W, H, merange, spread, own, depth):
    """8-bit samples: packed quad-SADs, 4 x 8 displacements per lane and task; deeper samples: v_sad_u16, 2 x 8."""
    from x265_amd.engine import Engine
    own_engine = Engine(0, depth) if depth == 12 else None
    engine8 = own_engine or request.getfixturevalue("engine8" if depth == 8 else "engine10")      # the engine of this depth
    m = 96
    fenc, ref, org, stride = clip_planes(W, H, depth, kind, seed=5, margin=m)
    pf = engine8.upload_plane(fenc, W, H, m, m, stride)
    pr = engine8.upload_plane(ref, W, H, m, m, stride)
    rng = np.random.default_rng(W + H + merange)
    jobs = _picture_jobs(W, H, rng, merange, spread, m, own)
    cj, index, nctu = workload.esa_ctu_jobs(jobs, W, H)
    assert nctu == ((W + 63) // 64) * ((H + 63) // 64) and (index >= 0).sum() == len(jobs)
    # (the last QP is the largest one: the largest lambda * bits entries, 27 050 -- two of them still add up below 2^16)
    for qp, subme in ((30 + 6 * (depth - 8), 5), (12 + 6 * (depth - 8), 2), (69, 1)):
        res = engine8.esa_ctu_batch(pf, pr, cj, nctu, qp, subme)
        have = index >= 0
        assert (res["cost"][~have] == -1).all()
        # the per-PU scans of the same jobs (themselves checked against the oracle elsewhere): every record equal
        per = engine8.search_batch(pf, pr, jobs, qp, METHODS["full"], subme)
        got = res[have]
        exp = per[index[have]]
        for k in ("mv", "cost", "sad_calls", "subpel_calls"):
            assert np.array_equal(got[k], exp[k]), (k, qp)
        # and the oracle itself on a sample of every size
        pick = np.concatenate([np.flatnonzero(have & (cj["w"] == s))[:: max(1, (have & (cj["w"] == s)).sum() // 6)] for s in (8, 16, 32, 64)])
        mv, cost, sc, qc = oracle_search(oracle, depth, fenc, [ref], org, stride, cj[pick], qp, METHODS["full"], subme)
        assert np.array_equal(res["mv"][pick], mv) and np.array_equal(res["cost"][pick], cost)
        assert np.array_equal(res["sad_calls"][pick], sc) and np.array_equal(res["subpel_calls"][pick], qc)
    engine8.free_plane(pf)
    engine8.free_plane(pr)
    if own_engine:
        own_engine.close()


def test_ctu_list_checks(engine8):
    W, H, m = 128, 64, 96
    fenc, ref, org, stride = clip_planes(W, H, 8, "texture", seed=6, margin=m)
    pf = engine8.upload_plane(fenc, W, H, m, m, stride)
    pr = engine8.upload_plane(ref, W, H, m, m, stride)
    jobs = _picture_jobs(W, H, np.random.default_rng(1), 16, 2, m)
    cj, index, nctu = workload.esa_ctu_jobs(jobs, W, H)
    bad = cj.copy()
    bad[3]["pu_x"] += 8                 # not the block of its slot
    with pytest.raises(EngineError):
        engine8.esa_ctu_batch(pf, pr, bad, nctu, 30, 5)
    bad = cj.copy()
    bad[64 * nctu]["w"] = 8             # a 16x16 slot holding another size
    with pytest.raises(EngineError):
        engine8.esa_ctu_batch(pf, pr, bad, nctu, 30, 5)
    bad = cj.copy()
    bad[5]["mvmax"] = (500, 0)          # window off the padded plane
    with pytest.raises(EngineError):
        engine8.esa_ctu_batch(pf, pr, bad, nctu, 30, 5)
    # an empty list and a list of empty slots are fine
    engine8.esa_ctu_batch(pf, pr, make_jobs(0), 0, 30, 5)
    empty = make_jobs(85)
    assert (engine8.esa_ctu_batch(pf, pr, empty, 1, 30, 5)["cost"] == -1).all()
    engine8.free_plane(pf)
    engine8.free_plane(pr)
