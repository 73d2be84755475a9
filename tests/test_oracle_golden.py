"""CPU: the plain-C oracle (oracle/me_oracle.c) against the golden vectors that
were generated from the reference's own code (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from oracle.bindings import PARTITIONS, METHODS
from x265_amd.synth import make_clip
import golden_util as G
from util import clip_planes, oracle_search, host_lowres_planes, margin_for


def test_cost_tables(oracle):
    for qp, tab in G.cost_tables().items():
        assert np.array_equal(oracle.cost_table(qp), tab), "BitCost table differs at qp %d" % qp


@pytest.mark.parametrize("depth", [8, 10])
def test_primitives(oracle, depth):
    g = G.prim_golden()[str(depth)]
    a, b = G.prim_inputs(depth)
    assert [G.sha(a), G.sha(b)] == g["input_sha"], "synthetic input generator drifted"
    for pi, (w, h) in enumerate(PARTITIONS):
        rows = g["parts"]["%dx%d" % (w, h)]
        for (ao, bo), e in zip(G.prim_offsets(pi), rows):
            fenc = np.ascontiguousarray(a.reshape(160, 256)[:64, :64]).reshape(-1)
            offs = [bo + k * 3 + k * 256 for k in range(4)]
            so = bo + 10 * 256 + 10
            assert oracle.sad(depth, a, ao, 256, b, bo, 251, w, h) == e["sad"]
            assert oracle.satd(depth, a, ao, 256, b, bo, 251, w, h) == e["satd"]
            assert oracle.sa8d_inter(depth, a, ao, 256, b, bo, 251, w, h) == e["sa8d"]
            assert oracle.sad_x3(depth, fenc, b, offs, 251, w, h) == e["x3"]
            assert oracle.sad_x4(depth, fenc, b, offs, 251, w, h) == e["x4"]
            assert G.sha(oracle.pixelavg(depth, a, ao, 256, b, bo, 251, w, h)) == e["avg"]
            for ci in (1, 2, 3):
                assert G.sha(oracle.luma_hpp(depth, b, so, 256, ci, w, h)) == e["hpp%d" % ci]
                assert G.sha(oracle.luma_vpp(depth, b, so, 256, ci, w, h)) == e["vpp%d" % ci]
                assert G.sha(oracle.luma_hps(depth, b, so, 256, ci, 0, w, h)) == e["hps%d" % ci]
                im = oracle.luma_hps(depth, b, so, 256, ci, 1, w, h)
                assert G.sha(im) == e["hpsx%d" % ci]
                assert G.sha(oracle.luma_vsp(depth, im.reshape(-1), 3 * w, w, ci, w, h)) == e["vsp%d" % ci]
    assert [G.sha(o) for o in oracle.lowres_core(depth, a, 0, 256, 100, 70)] == g["lowres"]


def test_motion_search(oracle):
    shas = G.input_shas()
    for case, (jobs, exp) in G.me_golden().items():
        depth, kind, mname, subme = case
        m = margin_for(METHODS[mname])
        fenc, ref, org, stride = clip_planes(G.ME_W, G.ME_H, depth, kind, margin=m)
        assert [G.sha(fenc), G.sha(ref)] == shas["%d_%s_%d" % (depth, kind, m)], "synthetic clip generator drifted"
        qp = 20 + 6 * (depth - 8) + subme
        mv, cost, _, _ = oracle_search(oracle, depth, fenc, [ref], org, stride, jobs, qp, METHODS[mname], subme)
        assert np.array_equal(mv, exp[:, :2]), case
        assert np.array_equal(cost, exp[:, 2]), case


@pytest.mark.parametrize("depth", [8, 10])
def test_lookahead_field(oracle, depth):
    g = G.lookahead_golden()
    clip = make_clip(G.LA_W, G.LA_H, 4, seed=3, depth=depth)
    assert [G.sha(clip[3]), G.sha(clip[1])] == G.input_shas()["la_%d" % depth]
    pa = host_lowres_planes(oracle, depth, clip[3], 64)
    pb = host_lowres_planes(oracle, depth, clip[1], 64)
    for (method, subme) in ((1, 1), (3, 2), (0, 0), (2, 5)):
        ctx = oracle.make_ctx(depth, pa[0][0], pa[1], pa[2], pb[0], pa[1], pa[2], True, 12 + 6 * (depth - 8), method, subme)
        mvs, costs = oracle.lookahead_field(ctx, pa[3], pa[4], 16)
        assert np.array_equal(mvs, g["%d_%d_%d_mvs" % (depth, method, subme)])
        assert np.array_equal(costs, g["%d_%d_%d_costs" % (depth, method, subme)])


def test_rk_ctu_records(oracle):
    """CIME / RIME / FME records of every CTU of a 416x240 frame pair (incl. the partial bottom row)."""
    from test_gpu_rk import rk_case, M
    g = np.load(G.GOLD + "/rk_golden.npz")
    W, H = 416, 240
    cur, rorig, recon, org, stride, jobs = rk_case(8, W, H, 5)
    assert [G.sha(cur), G.sha(recon)] == G.input_shas()["rk_8"]
    ds, dorg, dstride = oracle.rk_downsample(8, rorig, org, W, H, stride, M // 4)
    ctx = oracle.make_ctx(8, cur, org, stride, [recon], org, stride, False, 30, 0, 0)
    got = oracle.rk_ctu_batch(ctx, ds, dorg, dstride, W, H, jobs)
    for k in ("cime", "rime", "fme"):
        assert np.array_equal(got[k]["mv"], g[k + "_mv"]) and np.array_equal(got[k]["cost"], g[k + "_cost"]), k


def test_rk_matches_cmodel_records(oracle):
    """Oracle CIME / RIME / FME == the hardware C-model's records (made by tests/golden/make_golden.py)."""
    from test_gpu_rk import cmodel_case, M
    from x265_amd.synth import pad_plane
    g = np.load(G.GOLD + "/cmodel_golden.npz")
    W, H, qp, clip, recon_pic, sw, neigh, jobs = cmodel_case()
    cur, org, stride = pad_plane(clip[1], M)
    rorig, _, _ = pad_plane(clip[0], M)
    recon, _, _ = pad_plane(recon_pic, M)
    assert [G.sha(cur), G.sha(rorig), G.sha(recon)] == G.input_shas()["cmodel_8"]
    jobs["mvp"] = g["mvp"]
    jobs["pmv"] = g["pmv"][:, 1:3]
    ds, dorg, dstride = oracle.rk_downsample(8, rorig, org, W, H, stride, M // 4)
    ctx = oracle.make_ctx(8, cur, org, stride, [recon], org, stride, False, qp, 0, 0)
    got = oracle.rk_ctu_batch(ctx, ds, dorg, dstride, W, H, jobs)
    for k in ("cime", "rime", "fme"):
        assert np.array_equal(got[k]["mv"], g[k + "_mv"]) and np.array_equal(got[k]["cost"], g[k + "_cost"]), k


def test_lookahead_frame_cost(oracle):
    """Intra estimate, bidirectional candidates, list choice and the row / frame sums against the records of the
    reference's own estimateCUCost; the MV fields the oracle searches itself must equal the reference's too."""
    g = G.frame_cost_golden()
    shas = G.input_shas()
    cache = {}
    for name, kind, seed, b, d0, d1, aq in G.frame_cost_cases():
        if kind not in cache:
            clip = make_clip(G.LA_W, G.LA_H, 5, seed=seed, depth=8, kind=kind)
            assert [G.sha(f) for f in clip] == shas["fc_%s" % kind], "synthetic clip generator drifted"
            cache[kind] = [host_lowres_planes(oracle, 8, f, 64) for f in clip]
        P = cache[kind]
        org, stride, wcu, hcu = P[0][1], P[0][2], P[0][3], P[0][4]
        f, r0, r1 = P[b][0], (P[b - d0][0] if d0 else None), (P[b + d1][0] if d1 else None)
        fields = {}
        for l, (d, r) in enumerate(((d0, r0), (d1, r1))):
            if d:
                ctx = oracle.make_ctx(8, f[0], org, stride, r, org, stride, True, 12, 1, 1)
                mv, cost = oracle.lookahead_field(ctx, wcu, hcu, 16)
                assert np.array_equal(mv, g["%s_mvs%d" % (name, l)]) and np.array_equal(cost, g["%s_mvcosts%d" % (name, l)])
                fields[l] = (mv, cost)
        out = oracle.lookahead_frame_cost(8, f[0], r0, r1, org, stride, wcu, hcu, d0, d1,
                                          *(fields.get(0, (None, None)) + fields.get(1, (None, None))),
                                          inv_qscale=G.frame_cost_inv_qscale(wcu * hcu) if aq else None)
        for k in ("intra_cost", "lowres_costs", "row_satds_intra", "totals") + (("row_satds",) if d0 else ()):
            assert np.array_equal(out[k], g[name + "_" + k]), (name, k)


@pytest.mark.parametrize("depth", [8, 10])
def test_weighted_prediction(oracle, depth):
    """applyWeight, weightCostLuma and weightsAnalyse against the reference's own records."""
    from x265_amd.synth import pad_plane
    g = G.wp_golden()
    img = make_clip(200, 136, 1, seed=5, depth=depth)[0]
    buf, org, stride = pad_plane(img, 48, 32)
    assert G.sha(buf) == G.input_shas()["wp_apply_%d" % depth]
    for (w, d, o) in G.WP_WEIGHTS:
        got = oracle.apply_weight(depth, buf, org, stride, 200, 136, 48, 32, w, d, o)
        assert np.array_equal(G.sha_bytes(got), g["%d_apply_%d_%d_%d" % (depth, w, d, o)]), (w, d, o)
    for fade, off in G.WP_FADES:
        f, r = G.wp_fade_frames(depth, fade, off)
        PF, PR = host_lowres_planes(oracle, depth, f, 64), host_lowres_planes(oracle, depth, r, 64)
        org, stride, wcu, hcu = PF[1], PF[2], PF[3], PF[4]
        ic = G.wp_intra_costs(wcu * hcu)
        key = "%d_fade_%g_%d" % (depth, fade, off)
        assert oracle.weight_cost_luma(depth, PF[0][0], PR[0][0], org, stride, wcu, hcu, ic) == g[key + "_cost_plain"][0]
        assert oracle.weight_cost_luma(depth, PF[0][0], PR[0][0], org, stride, wcu, hcu, ic, (50, 6, 7)) == g[key + "_cost_wp"][0]
        weighted, w, _ = oracle.weights_analyse(depth, PF[0][0], PR[0][0], org, stride, wcu, hcu, ic, G.wp_stats(f) + G.wp_stats(r))
        assert [int(weighted)] + (list(w) if weighted else [0, 0, 0]) == list(g[key + "_decision"]), key
        if weighted:
            corr, padded = 14 - depth, PF[6] + 128
            for i in range(4):
                out = np.zeros_like(PR[0][i])
                oracle.weight_pp(depth, PR[0][i], 0, out, 0, stride, stride, stride, padded, w[0], 1 << (w[1] - 1 + corr), w[1] + corr,
                                 w[2] << (depth - 8))
                assert np.array_equal(G.sha_bytes(out[:stride * padded]), g[key + "_planes"][i]), (key, i)


@pytest.mark.parametrize("depth", [8, 10])
def test_bipred_cost(oracle, depth):
    f, r0, r1, org, stride, jobs = G.bipred_case(depth)
    assert [G.sha(f), G.sha(r0), G.sha(r1), G.sha(jobs)] == G.input_shas()["bipred_%d" % depth]
    assert np.array_equal(oracle.bipred_batch(depth, f, r0, r1, org, stride, jobs), G.bipred_golden()[depth])


@pytest.mark.parametrize("depth", [8, 10])
def test_merge_sa8d(oracle, depth):
    f, refs, org, stride, jobs = G.merge_case(depth)
    assert [G.sha(f)] + [G.sha(r) for r in refs] + [G.sha(jobs)] == G.input_shas()["merge_%d" % depth]
    assert np.array_equal(oracle.merge_sa8d_batch(depth, f, refs, org, stride, jobs), G.merge_golden()[depth])


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("depth", [8, 10])
def test_motion_compensation(oracle, depth, weighted):
    """Luma + chroma prediction pictures of the oracle vs the digests recorded from the reference's primitives."""
    from util import mc_blank_pred
    refs, batches, wts = G.mc_case(depth, weighted)
    key = "%d%s" % (depth, "w" if weighted else "")
    assert [G.sha(pl[0]) for pic in refs for pl in pic] + [G.sha(np.concatenate(batches))] + ([G.sha(wts)] if weighted else []) \
        == G.input_shas()["mc_" + key]
    exp = G.mc_golden()[key]
    for b, jobs in enumerate(batches):
        pred = mc_blank_pred(depth)
        oracle.mc_batch(depth, refs, pred, jobs, wts)
        for k in range(3):
            assert np.array_equal(G.sha_bytes(pred[k][0]), exp[b, k]), "batch %d plane %d" % (b, k)
