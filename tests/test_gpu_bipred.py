"""GPU: bi-prediction estimate (x265amd_bipred_batch) against the reference-generated records and the oracle."""
import numpy as np
import pytest

from x265_amd.engine import BIPRED_JOB_DTYPE, PARTITIONS
from x265_amd.synth import make_clip, pad_plane
import golden_util as G

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth", [8, 10])
def test_bipred_golden(request, depth):
    e = request.getfixturevalue("engine8" if depth == 8 else "engine10")
    f, r0, r1, org, stride, jobs = G.bipred_case(depth)
    pf, p0, p1 = (e.upload_plane(b, G.ME_W, G.ME_H, 96, 96, stride) for b in (f, r0, r1))
    res = e.bipred_batch(pf, p0, p1, jobs)
    exp = G.bipred_golden()[depth]
    assert np.array_equal(res["satd"], exp[:, 0]) and np.array_equal(res["satd_zero"], exp[:, 1])
    for p in (pf, p0, p1):
        e.free_plane(p)


def test_bipred_unaligned_and_edges_vs_oracle(engine8, oracle):
    """PUs at odd positions, MVs reaching into the margins, empty batch, rejected jobs."""
    W, H, M = 160, 96, 80
    clip = make_clip(W, H, 3, seed=12, depth=8, kind="noise")
    f, org, stride = pad_plane(clip[1], M)
    r0, _, _ = pad_plane(clip[0], M)
    r1, _, _ = pad_plane(clip[2], M)
    pf, p0, p1 = (engine8.upload_plane(b, W, H, M, M, stride) for b in (f, r0, r1))
    rng = np.random.default_rng(6)
    jobs = np.zeros(400, BIPRED_JOB_DTYPE)
    for j in jobs:
        w, h = PARTITIONS[1:][int(rng.integers(0, 24))]
        j["pu_x"], j["pu_y"], j["w"], j["h"] = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1)), w, h
        j["mv0"], j["mv1"] = rng.integers(-200, 201, 2), rng.integers(-200, 201, 2)
    res = engine8.bipred_batch(pf, p0, p1, jobs)
    exp = oracle.bipred_batch(8, f, r0, r1, org, stride, jobs)
    assert np.array_equal(res["satd"], exp[:, 0]) and np.array_equal(res["satd_zero"], exp[:, 1])
    assert len(engine8.bipred_batch(pf, p0, p1, jobs[:0])) == 0
    bad = jobs[:1].copy()
    bad["mv0"] = (4 * (M + W), 0)
    with pytest.raises(Exception):
        engine8.bipred_batch(pf, p0, p1, bad)
    bad = jobs[:1].copy()
    bad["w"] = 20
    with pytest.raises(Exception):
        engine8.bipred_batch(pf, p0, p1, bad)


@pytest.mark.parametrize("depth", [8, 10])
def test_merge_sa8d_golden(request, depth):
    e = request.getfixturevalue("engine8" if depth == 8 else "engine10")
    f, refs, org, stride, jobs = G.merge_case(depth)
    pf = e.upload_plane(f, G.ME_W, G.ME_H, 96, 96, stride)
    pr = [e.upload_plane(r, G.ME_W, G.ME_H, 96, 96, stride) for r in refs]
    assert np.array_equal(e.merge_sa8d_batch(pf, pr, jobs), G.merge_golden()[depth])
    for p in [pf] + pr:
        e.free_plane(p)


def test_merge_sa8d_fractions_vs_oracle(engine8, oracle):
    """All 256 fraction pairs of bi-predicted candidates on noise, unaligned CU positions, rejected jobs."""
    from x265_amd.engine import MERGE_JOB_DTYPE
    W, H, M = 160, 96, 80
    clip = make_clip(W, H, 3, seed=14, depth=8, kind="noise")
    f, org, stride = pad_plane(clip[1], M)
    refs = [pad_plane(clip[0], M)[0], pad_plane(clip[2], M)[0]]
    pf = engine8.upload_plane(f, W, H, M, M, stride)
    pr = [engine8.upload_plane(r, W, H, M, M, stride) for r in refs]
    rng = np.random.default_rng(16)
    jobs = np.zeros(256 + 60, MERGE_JOB_DTYPE)
    for i, j in enumerate(jobs):
        s = (8, 16, 32, 64)[i % 4]
        j["pu_x"], j["pu_y"], j["size"] = int(rng.integers(0, W - s + 1)), int(rng.integers(0, H - s + 1)), s
        j["dir"], j["ref0"], j["ref1"] = (3 if i < 256 else 1 + i % 2), i % 2, (i // 2) % 2
        j["mv0"], j["mv1"] = rng.integers(-150, 151, 2), rng.integers(-150, 151, 2)
        if i < 256:
            j["mv0"] = ((j["mv0"][0] & ~3) | (i & 3), (j["mv0"][1] & ~3) | ((i >> 2) & 3))
            j["mv1"] = ((j["mv1"][0] & ~3) | ((i >> 4) & 3), (j["mv1"][1] & ~3) | ((i >> 6) & 3))
    assert np.array_equal(engine8.merge_sa8d_batch(pf, pr, jobs), oracle.merge_sa8d_batch(8, f, refs, org, stride, jobs))
    for field, val in (("size", 12), ("dir", 0), ("ref1", 2)):
        bad = jobs[:1].copy()
        bad[field] = val
        with pytest.raises(Exception):
            engine8.merge_sa8d_batch(pf, pr, bad)
