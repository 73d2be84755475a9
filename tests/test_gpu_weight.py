"""GPU: explicit weighted prediction (x265amd_apply_weight, x265amd_weight_buffer, x265amd_lookahead_weight_cost,
x265amd_weights_analyse) against the reference-generated records and the oracle."""
import numpy as np
import pytest

from x265_amd.synth import make_clip, pad_plane
import golden_util as G
from util import host_lowres_planes

pytestmark = pytest.mark.gpu


def _device_lowres(e, img, margin=64):
    buf, org, stride = pad_plane(img, margin)
    p = e.upload_plane(buf, img.shape[1], img.shape[0], margin, margin, stride)
    return e.lowres_init(p)


@pytest.mark.parametrize("depth", [8, 10])
def test_apply_weight_golden(request, oracle, depth):
    e = request.getfixturevalue("engine8" if depth == 8 else "engine10")
    g = G.wp_golden()
    img = make_clip(200, 136, 1, seed=5, depth=depth)[0]
    buf, org, stride = pad_plane(img, 48, 32)
    p = e.upload_plane(buf, 200, 136, 48, 32, stride)
    for (w, d, o) in G.WP_WEIGHTS:
        q = e.apply_weight(p, w, d, o)
        assert np.array_equal(G.sha_bytes(e.download_plane(q).reshape(-1)), g["%d_apply_%d_%d_%d" % (depth, w, d, o)]), (w, d, o)
        e.free_plane(q)
    # odd geometry: width not a multiple of 16, last CTU row partial
    img = make_clip(180, 100, 1, seed=9, depth=depth)[0]
    buf, org, stride = pad_plane(img, 40, 24)
    p2 = e.upload_plane(buf, 180, 100, 40, 24, stride)
    for (w, d, o) in ((77, 6, -9), (5, 3, 40), (0, 0, 17)):
        q = e.apply_weight(p2, w, d, o)
        assert np.array_equal(e.download_plane(q).reshape(-1), oracle.apply_weight(depth, buf, org, stride, 180, 100, 40, 24, w, d, o))
        e.free_plane(q)
    with pytest.raises(Exception):
        e.apply_weight(p2, 64, -1, 0)


@pytest.mark.parametrize("depth", [8, 10])
def test_weights_analyse_golden(request, oracle, depth):
    e = request.getfixturevalue("engine8" if depth == 8 else "engine10")
    g = G.wp_golden()
    for fade, off in G.WP_FADES:
        f, r = G.wp_fade_frames(depth, fade, off)
        LF, LR = _device_lowres(e, f), _device_lowres(e, r)
        wcu, hcu = 26, 15
        ic = G.wp_intra_costs(wcu * hcu)
        d_ic = e.device_alloc(ic.nbytes)
        e.upload(d_ic, ic)
        key = "%d_fade_%g_%d" % (depth, fade, off)
        assert e.lookahead_weight_cost(LF[0], LR[0], d_ic, wcu, hcu) == g[key + "_cost_plain"][0]
        assert e.lookahead_weight_cost(LF[0], LR[0], d_ic, wcu, hcu, (50, 6, 7)) == g[key + "_cost_wp"][0]
        w, planes = e.weights_analyse(LF[0], LR, d_ic, wcu, hcu, G.wp_stats(f) + G.wp_stats(r))
        dec = [int(w.weighted)] + ([w.scale, w.denom, w.offset] if w.weighted else [0, 0, 0])
        assert dec == list(g[key + "_decision"]), key
        if w.weighted:
            for i in range(4):
                assert np.array_equal(G.sha_bytes(e.download_plane(planes[i]).reshape(-1)), g[key + "_planes"][i]), (key, i)
                e.free_plane(planes[i])
        e.device_free(d_ic)


def test_weighted_reference_feeds_the_search(engine8, oracle):
    """A faded frame searched against its weighted reference: the lookahead field on the weighted planes equals the
    oracle's on the oracle-weighted planes, and beats the unweighted search."""
    f, r = G.wp_fade_frames(8, 0.6, -5)
    LF, LR = _device_lowres(engine8, f), _device_lowres(engine8, r)
    PF, PR = host_lowres_planes(oracle, 8, f, 64), host_lowres_planes(oracle, 8, r, 64)
    org, stride, wcu, hcu = PF[1], PF[2], PF[3], PF[4]
    ic = G.wp_intra_costs(wcu * hcu)
    d_ic = engine8.device_alloc(ic.nbytes)
    engine8.upload(d_ic, ic)
    w, planes = engine8.weights_analyse(LF[0], LR, d_ic, wcu, hcu, G.wp_stats(f) + G.wp_stats(r))
    assert w.weighted
    host_w = []
    for i in range(4):
        out = np.zeros_like(PR[0][i])
        oracle.weight_pp(8, PR[0][i], 0, out, 0, stride, stride, stride, PF[6] + 128, w.scale, 1 << (w.denom - 1 + 6), w.denom + 6, w.offset)
        host_w.append(out)
    ctx = oracle.make_ctx(8, PF[0][0], org, stride, host_w, org, stride, True, 12, 1, 1)
    mv, cost = oracle.lookahead_field(ctx, wcu, hcu, 16)
    mvs, costs, _ = engine8.lookahead_fields([(LF[0], planes)], wcu, hcu, 12)
    assert np.array_equal(mvs[0], mv) and np.array_equal(costs[0], cost)
    _, plain, _ = engine8.lookahead_fields([(LF[0], LR)], wcu, hcu, 12)
    assert costs[0].sum() < plain[0].sum()
    engine8.device_free(d_ic)
