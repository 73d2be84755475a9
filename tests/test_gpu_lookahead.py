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
    wcu, hcu = 26, 15
    for (method, subme) in ((1, 1), (3, 2), (0, 0), (2, 5)):
        # UMH measures far outside its range (the entry point demands 32 + 8 * 14 + 30 samples of margin for it)
        margin = 192 if method == 2 else 64
        la, lb = _device_lowres(e, clip[3], margin), _device_lowres(e, clip[1], margin)
        mvs, costs, _ = e.lookahead_fields([(la[0], lb), (la[0], lb)], wcu, hcu, 12 + 6 * (depth - 8), method, subme, 16)
        for i in range(2):
            assert np.array_equal(mvs[i], g["%d_%d_%d_mvs" % (depth, method, subme)]), (method, subme)
            assert np.array_equal(costs[i], g["%d_%d_%d_costs" % (depth, method, subme)]), (method, subme)


def test_field_geometry_is_checked(engine8):
    """Planes whose margins cannot hold what the search method reads are refused before anything is launched."""
    from x265_amd.engine import EngineError, Plane
    clip = make_clip(G.LA_W, G.LA_H, 2, seed=3, depth=8)
    la, lb = _device_lowres(engine8, clip[1], 16), _device_lowres(engine8, clip[0], 16)
    engine8.lookahead_fields([(la[0], lb)], 26, 15, 12, 1, 1, 16)                 # HEX: 12 samples are enough
    with pytest.raises(EngineError):
        engine8.lookahead_fields([(la[0], lb)], 26, 15, 12, 2, 1, 16)             # UMH reads far outside its range
    with pytest.raises(EngineError):
        engine8.fields_upload([(la[0], lb)], 26, 15, 2, 16)
    with pytest.raises(EngineError):
        engine8.lookahead_fields([(la[0], lb)], 27, 15, 12, 1, 1, 16)             # grid wider than the planes
    small = _device_lowres(engine8, clip[0], 8)
    with pytest.raises(EngineError):
        engine8.lookahead_fields([(la[0], small)], 26, 15, 12, 1, 1, 16)          # margins below 12 samples
    mixed = [lb[0], lb[1], small[2], lb[3]]
    with pytest.raises(EngineError):
        engine8.lookahead_fields([(la[0], mixed)], 26, 15, 12, 1, 1, 16)          # one half-pel plane with another stride
    with pytest.raises(EngineError):
        engine8.lookahead_fields([(Plane(), lb)], 26, 15, 12, 1, 1, 16)           # null source plane
    # the device-resident form runs only on what an upload was checked for
    d = engine8.fields_upload([(la[0], lb), (la[0], lb)], 26, 15, 1, 16)
    ncu = 26 * 15
    mv, cost = engine8.device_alloc(2 * ncu * 4), engine8.device_alloc(2 * ncu * 4)
    engine8.lookahead_fields_dev(d, 2, 26, 15, 12, 1, 1, 16, mv, cost)
    engine8.sync()
    got = engine8.download(mv, (2, ncu, 2), np.int16)
    want = engine8.lookahead_fields([(la[0], lb)], 26, 15, 12, 1, 1, 16)[0]
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[0])
    engine8.lookahead_fields_dev(d.value + 80, 1, 26, 15, 12, 1, 1, 8, mv, cost)     # a run of its fields, a smaller range
    for bad in (dict(n=3), dict(wcu=25), dict(merange=17), dict(method=2), dict(ptr=d.value + 40), dict(ptr=mv)):
        with pytest.raises(EngineError):
            engine8.lookahead_fields_dev(bad.get("ptr", d), bad.get("n", 2), bad.get("wcu", 26), 15, 12, bad.get("method", 1), 1,
                                         bad.get("merange", 16), mv, cost)
    engine8.sync()
    engine8.device_free(d)
    with pytest.raises(EngineError):
        engine8.lookahead_fields_dev(d, 2, 26, 15, 12, 1, 1, 16, mv, cost)          # freed
    engine8.device_free(mv)
    engine8.device_free(cost)


@pytest.mark.parametrize("rows", [2, 4])
def test_field_shapes_vs_oracle(engine8, oracle, monkeypatch, rows):
    """Ragged and tiny CU grids: 1xN, Nx1, 2x2, odd sizes; through both forms of the band kernel (the library picks by
    launch size; X265AMD_LOOKAHEAD_ROWS forces the CU rows per wave)."""
    from oracle.bindings import MEStats
    monkeypatch.setenv("X265AMD_LOOKAHEAD_ROWS", str(rows))
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


def test_frame_cost_golden(engine8):
    """estimateFrameCost on the device (searches + intra + bidirectional candidates + choice + sums) against the
    records of the reference's own estimateCUCost, all cases in one batch."""
    g = G.frame_cost_golden()
    cache, frames, names = {}, [], []
    for name, kind, seed, b, d0, d1, aq in G.frame_cost_cases():
        if kind not in cache:
            clip = make_clip(G.LA_W, G.LA_H, 5, seed=seed, depth=8, kind=kind)
            cache[kind] = [_device_lowres(engine8, f) for f in clip]
        L = cache[kind]
        frames.append(dict(fenc=L[b], ref0=L[b - d0] if d0 else None, ref1=L[b + d1] if d1 else None, d0=d0, d1=d1,
                           inv_qscale=G.frame_cost_inv_qscale(26 * 15) if aq else None))
        names.append((name, d0, d1))
    res = engine8.estimate_frame_costs(frames, 26, 15, 12)
    for (name, d0, d1), r in zip(names, res):
        keys = ["intra_cost", "lowres_costs", "row_satds_intra", "totals"]
        if d0:
            keys += ["row_satds", "mvs0", "mvcosts0"]
        if d1:
            keys += ["mvs1", "mvcosts1"]
        for k in keys:
            assert np.array_equal(r[k.replace("mvcosts", "mv_costs")], g[name + "_" + k]), (name, k)


def test_frame_cost_shapes_vs_oracle(engine8, oracle):
    """Ragged and tiny CU grids, first call and repeat call (intra costs and MV fields handed back in)."""
    for (w, h, kind) in ((16, 16, "texture"), (16, 144, "noise"), (144, 16, "texture"), (88, 56, "noise"), (300, 200, "texture")):
        clip = make_clip(w, h, 4, seed=w + h, depth=8, kind=kind)
        P = [host_lowres_planes(oracle, 8, f, 64) for f in clip]
        L = [_device_lowres(engine8, f) for f in clip]
        org, stride, wcu, hcu = P[0][1], P[0][2], P[0][3], P[0][4]
        q = np.random.default_rng(w).integers(64, 700, wcu * hcu).astype(np.int32)
        triples = ((1, 0, 0), (2, 2, 0), (1, 1, 2))
        first = engine8.estimate_frame_costs([dict(fenc=L[b], ref0=L[b - d0] if d0 else None, ref1=L[b + d1] if d1 else None,
                                                   d0=d0, d1=d1, inv_qscale=q) for b, d0, d1 in triples], wcu, hcu, 12)
        again = engine8.estimate_frame_costs([dict(fenc=L[b], ref0=L[b - d0] if d0 else None, ref1=L[b + d1] if d1 else None,
                                                   d0=d0, d1=d1, inv_qscale=q, intra_cost=r["intra_cost"],
                                                   mvs0=r["mvs0"] if d0 else None, mv_costs0=r["mv_costs0"],
                                                   mvs1=r["mvs1"] if d1 else None, mv_costs1=r["mv_costs1"])
                                              for (b, d0, d1), r in zip(triples, first)], wcu, hcu, 12)
        for (b, d0, d1), r, r2 in zip(triples, first, again):
            exp = oracle.lookahead_frame_cost(8, P[b][0][0], P[b - d0][0] if d0 else None, P[b + d1][0] if d1 else None,
                                              org, stride, wcu, hcu, d0, d1, r["mvs0"], r["mv_costs0"], r["mvs1"], r["mv_costs1"],
                                              None, q)
            for k in ("intra_cost", "lowres_costs", "row_satds_intra", "totals") + (("row_satds",) if d0 else ()):
                assert np.array_equal(r[k], exp[k]), (w, h, b, d0, d1, k)
            for k in ("lowres_costs", "totals") + (("row_satds",) if d0 else ()):
                got, want = r2[k], exp[k].copy()
                if k == "totals":
                    want[2] = want[3] = 0     # the intra sums belong to the call that computed the intra costs
                assert np.array_equal(got, want), (w, h, b, d0, d1, k, "repeat")


def test_frame_cost_bidir_10bit(engine10, oracle):
    """>8-bit pictures: the bidirectional / choice part with intra costs supplied by the caller."""
    w, h = 200, 120
    clip = make_clip(w, h, 4, seed=9, depth=10)
    P = [host_lowres_planes(oracle, 10, f, 64) for f in clip]
    L = [_device_lowres(engine10, f) for f in clip]
    org, stride, wcu, hcu = P[0][1], P[0][2], P[0][3], P[0][4]
    ic = np.random.default_rng(2).integers(200, 3000, wcu * hcu).astype(np.int32)
    for (b, d0, d1) in ((2, 2, 0), (1, 1, 2)):
        r = engine10.estimate_frame_costs([dict(fenc=L[b], ref0=L[b - d0], ref1=L[b + d1] if d1 else None, d0=d0, d1=d1,
                                                intra_cost=ic)], wcu, hcu, 24)[0]
        exp = oracle.lookahead_frame_cost(10, P[b][0][0], P[b - d0][0], P[b + d1][0] if d1 else None, org, stride, wcu, hcu,
                                          d0, d1, r["mvs0"], r["mv_costs0"], r["mvs1"], r["mv_costs1"], ic, None)
        for k in ("lowres_costs", "row_satds", "totals"):
            assert np.array_equal(r[k], exp[k]), (b, d0, d1, k)
    with pytest.raises(Exception):
        engine10.estimate_frame_costs([dict(fenc=L[1], ref0=None, ref1=None, d0=0, d1=0)], wcu, hcu, 24)


def test_both_lookahead_kernels_agree(engine8, oracle, monkeypatch):
    """The workgroup-per-field kernel and the row-parallel kernel give the same fields (the library picks by batch
    size; X265AMD_LOOKAHEAD_KERNEL forces one), for a batch larger than the number of rows a launch starts with."""
    clip = make_clip(416, 240, 5, seed=23, depth=8)
    L = [_device_lowres(engine8, f) for f in clip]
    fields = [(L[i][0], L[j]) for i in range(5) for j in range(5) if i != j] * 3        # 60 fields
    out = {}
    for which in ("field", "rows"):
        monkeypatch.setenv("X265AMD_LOOKAHEAD_KERNEL", which)
        out[which] = engine8.lookahead_fields(fields, 26, 15, 12, 1, 1, 16, want_stats=True)
    monkeypatch.delenv("X265AMD_LOOKAHEAD_KERNEL")
    for a, b in zip(out["field"], out["rows"]):
        assert np.array_equal(a, b)
    pa = host_lowres_planes(oracle, 8, clip[0], 64)
    pb = host_lowres_planes(oracle, 8, clip[1], 64)
    ctx = oracle.make_ctx(8, pa[0][0], pa[1], pa[2], pb[0], pa[1], pa[2], True, 12, 1, 1)
    mv, cost = oracle.lookahead_field(ctx, 26, 15, 16)
    assert np.array_equal(out["rows"][0][0], mv) and np.array_equal(out["rows"][1][0], cost)


@pytest.mark.parametrize("depth", [8, 10])
def test_fields_at_the_margin_limit(request, oracle, depth):
    """Reference planes with exactly the 12 margin samples the entry point asks for, and motion that drags the border
    CUs' searches to the end of their range on every side: the fields equal the oracle's (which reads wide planes)."""
    from util import lowres_dims
    e = request.getfixturevalue("engine8" if depth == 8 else "engine10")
    w, h = 224, 128
    big = make_clip(w + 96, h + 96, 1, seed=31, depth=depth)[0]
    ref = np.ascontiguousarray(big[48:48 + h, 48:48 + w])
    for (dx, dy) in ((22, 18), (-22, -18), (20, -14), (-6, 24)):
        cur = np.ascontiguousarray(big[48 + dy:48 + dy + h, 48 + dx:48 + dx + w])
        pa = host_lowres_planes(oracle, depth, cur, 64)
        pb = host_lowres_planes(oracle, depth, ref, 64)
        wcu, hcu, lw, lh = pa[3], pa[4], pa[5], pa[6]
        ctx = oracle.make_ctx(depth, pa[0][0], pa[1], pa[2], pb[0], pa[1], pa[2], True, 12, 1, 1)
        mv, cost = oracle.lookahead_field(ctx, wcu, hcu, 16)
        assert np.abs(mv.astype(np.int32)).max() >= 4 * 8        # some search did reach the end of its range
        na = host_lowres_planes(oracle, depth, cur, 16)
        nb = host_lowres_planes(oracle, depth, ref, 12)
        fenc = e.upload_plane(na[0][0], lw, lh, 16, 16, na[2])
        refs = [e.upload_plane(nb[0][k], lw, lh, 12, 12, nb[2]) for k in range(4)]
        mvs, costs, _ = e.lookahead_fields([(fenc, refs)], wcu, hcu, 12, 1, 1, 16)
        assert np.array_equal(mvs[0], mv) and np.array_equal(costs[0], cost), (dx, dy)
