"""GPU: lookahead MV fields (x265amd_lookahead_fields, x265amd_lowres_init) against the
oracle and the reference-generated golden fields."""
import numpy as np
import pytest

from x265_amd.synth import make_clip, pad_plane
from x265_amd.engine import make_jobs
import golden_util as G
from util import host_lowres_planes

pytestmark = pytest.mark.gpu


def _device_lowres(e, img, margin=64):
    buf, org, stride = pad_plane(img, margin)
    p = e.upload_plane(buf, img.shape[1], img.shape[0], margin, margin, stride)
    return e.lowres_init(p)


@pytest.mark.parametrize("depth", [8, 10])
def test_lowres_init_matches_host(request, oracle, depth):
    e = request.getfixturevalue("engine8" if depth == 8 else "engine10")
    for (w, h) in ((416, 240), (200, 120), (1920, 1080)):
        img = make_clip(w, h, 1, seed=13, depth=depth)[0]
        lp = _device_lowres(e, img)
        hp, horg, hstride, wcu, hcu, lw, lh = host_lowres_planes(oracle, depth, img, 64)
        for k in range(4):
            assert lp[k].stride == hstride and (lp[k].width, lp[k].height) == (lw, lh)
            assert np.array_equal(e.download_plane(lp[k]).reshape(-1), hp[k]), (w, h, k)


@pytest.mark.parametrize("depth", [8, 10])
def test_golden_fields(request, depth):
    e = request.getfixturevalue("engine8" if depth == 8 else "engine10")
    g = G.lookahead_golden()
    clip = make_clip(G.LA_W, G.LA_H, 4, seed=3, depth=depth)
    la, lb = _device_lowres(e, clip[3]), _device_lowres(e, clip[1])
    wcu, hcu = 26, 15
    for (method, subme) in ((1, 1), (3, 2), (0, 0), (2, 5)):
        mvs, costs, _ = e.lookahead_fields([(la[0], lb), (la[0], lb)], wcu, hcu, 12 + 6 * (depth - 8), method, subme, 16)
        for i in range(2):
            assert np.array_equal(mvs[i], g["%d_%d_%d_mvs" % (depth, method, subme)]), (method, subme)
            assert np.array_equal(costs[i], g["%d_%d_%d_costs" % (depth, method, subme)]), (method, subme)


def test_field_shapes_vs_oracle(engine8, oracle):
    """Ragged and tiny CU grids: 1xN, Nx1, 2x2, odd sizes."""
    from oracle.bindings import MEStats
    for (w, h) in ((16, 16), (16, 144), (144, 16), (32, 32), (88, 56), (300, 200)):
        clip = make_clip(w, h, 3, seed=w + h, depth=8)
        pa = host_lowres_planes(oracle, 8, clip[2], 64)
        pb = host_lowres_planes(oracle, 8, clip[0], 64)
        la, lb = _device_lowres(engine8, clip[2]), _device_lowres(engine8, clip[0])
        wcu, hcu = pa[3], pa[4]
        ctx = oracle.make_ctx(8, pa[0][0], pa[1], pa[2], pb[0], pa[1], pa[2], True, 12, 1, 1)
        st = MEStats()
        mv, cost = oracle.lookahead_field(ctx, wcu, hcu, 16, st)
        mvs, costs, stats = engine8.lookahead_fields([(la[0], lb)], wcu, hcu, 12, 1, 1, 16, want_stats=True)
        assert np.array_equal(mvs[0], mv) and np.array_equal(costs[0], cost), (w, h)
        assert int(stats["sad_calls"].sum()) == st.sad_calls and int(stats["subpel_calls"].sum()) == st.subpel_calls


def test_lowres_search_batch_vs_oracle(engine8, oracle):
    """Independent 8x8 lowres jobs through the batch entry point (is_lowres = 1)."""
    from util import oracle_search
    clip = make_clip(416, 240, 3, seed=17, depth=8)
    pa = host_lowres_planes(oracle, 8, clip[2], 64)
    pb = host_lowres_planes(oracle, 8, clip[0], 64)
    la, lb = _device_lowres(engine8, clip[2]), _device_lowres(engine8, clip[0])
    rng = np.random.default_rng(4)
    n = 200
    jobs = make_jobs(n)
    jobs["pu_x"] = rng.integers(0, 25, n) * 8
    jobs["pu_y"] = rng.integers(0, 14, n) * 8
    jobs["w"] = 8; jobs["h"] = 8
    jobs["mvmin"] = (-24, -24); jobs["mvmax"] = (24, 24); jobs["merange"] = 16
    jobs["qmvp"] = rng.integers(-30, 31, (n, 2))
    jobs["numc"] = 2
    jobs["mvc"][:, :2] = rng.integers(-40, 41, (n, 2, 2))
    for method, subme in ((1, 1), (3, 5), (0, 2)):
        r = engine8.search_batch(la[0], lb, jobs, 12, method, subme, is_lowres=True)
        mv, cost, sc, qc = oracle_search(oracle, 8, pa[0][0], pb[0], pa[1], pa[2], jobs, 12, method, subme, is_lowres=True)
        assert np.array_equal(r["mv"], mv) and np.array_equal(r["cost"], cost)
        assert np.array_equal(r["sad_calls"], sc) and np.array_equal(r["subpel_calls"], qc)
