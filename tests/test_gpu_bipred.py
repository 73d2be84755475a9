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
