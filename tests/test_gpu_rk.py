"""GPU: the Rockchip CIME / RIME / FME path (x265amd_rk_ctu_batch, x265amd_rk_ds_plane)
against the oracle and the reference-generated golden records."""
import numpy as np
import pytest

from x265_amd import workload
from x265_amd.engine import RK_CTU_DTYPE
from x265_amd.synth import make_clip, pad_plane
import golden_util as G

pytestmark = pytest.mark.gpu
M = 304   # the reference's luma margin (TComPicYuv.cpp:85-91)


def rk_case(depth, W, H, seed):
    """(cur, orig ref, recon ref, origin, stride, jobs) -- recon = orig + small noise, as after coding."""
    clip = make_clip(W, H, 3, seed=seed, depth=depth)
    maxv = (1 << depth) - 1
    cur, org, stride = pad_plane(clip[2], M)
    rorig, _, _ = pad_plane(clip[0], M)
    noise = np.random.default_rng(seed).integers(-2, 3, clip[0].shape)
    recon, _, _ = pad_plane(np.clip(clip[0].astype(int) + noise, 0, maxv).astype(clip[0].dtype), M)
    rng = np.random.default_rng(seed + 1)
    nctu = ((W + 63) // 64) * ((H + 63) // 64)
    pmvs = (rng.integers(-12, 13, (nctu, 2, 2)) * 2).astype(np.int16)
    mvps = (rng.integers(-12, 13, (nctu, 3, 2)) * 2).astype(np.int16)
    jobs = workload.rk_ctu_jobs(W, H, pmvs, mvps, have_mvd=True)
    jobs["have_mvd"][::5] = 0
    return cur, rorig, recon, org, stride, jobs


def cmodel_case():
    """Inputs of the C-model twin checks: (W, H, qp, clip, recon picture, search-window size of the coarse stage,
    per-CTU 28 neighbour MVs, jobs without predictors). The coarse window is range / 4 + 16 samples of the 4x4-sum
    plane (TEncCu.cpp:404-406) and the MV range follows from it (hardwareC/inter.cpp:683-685)."""
    W, H, meX, meY, qp = 320, 192, 96, 64, 31
    clip = make_clip(W, H, 2, seed=17)
    noise = np.random.default_rng(3).integers(-2, 3, clip[0].shape)
    recon_pic = np.clip(clip[0].astype(int) + noise, 0, 255).astype(np.uint8)
    sw_w, sw_h = meX // 4 + 16, meY // 4 + 16
    lo_x, lo_y = -(sw_w - 16) // 2, -(sw_h - 16) // 2
    wctu, hctu = (W + 63) // 64, (H + 63) // 64
    neigh = (np.random.default_rng(23).integers(-16, 17, (wctu * hctu, 28, 2)) * 2).astype(np.int16)
    jobs = np.zeros(wctu * hctu, RK_CTU_DTYPE)
    for a in range(wctu * hctu):
        cx, cy = (a % wctu) * 64, (a // wctu) * 64
        jobs[a]["ctu_x"], jobs[a]["ctu_y"] = cx, cy
        jobs[a]["valid_w"], jobs[a]["valid_h"] = min(64, W - cx), min(64, H - cy)
        jobs[a]["mvmin"], jobs[a]["mvmax"] = (4 * lo_x, 4 * lo_y), (4 * (-1 - lo_x), 4 * (-1 - lo_y))
        jobs[a]["have_mvd"] = 1
    return W, H, qp, clip, recon_pic, (sw_w, sw_h), neigh, jobs


def same(a, b):
    return all(np.array_equal(a[k]["mv"], b[k]["mv"]) and np.array_equal(a[k]["cost"], b[k]["cost"])
               for k in ("cime", "rime", "fme"))


@pytest.mark.parametrize("size", [(416, 240), (200, 136), (128, 64), (204, 140)])
def test_rk_vs_oracle(engine8, oracle, size):
    W, H = size
    cur, rorig, recon, org, stride, jobs = rk_case(8, W, H, 5)
    pc = engine8.upload_plane(cur, W, H, M, M, stride)
    po = engine8.upload_plane(rorig, W, H, M, M, stride)
    pr = engine8.upload_plane(recon, W, H, M, M, stride)
    pds = engine8.rk_ds_plane(po)
    ds, dorg, dstride = oracle.rk_downsample(8, rorig, org, W, H, stride, M // 4)
    assert pds.stride == dstride
    assert np.array_equal(engine8.download_ds_plane(pds).reshape(-1), ds)
    ctx = oracle.make_ctx(8, cur, org, stride, [recon], org, stride, False, 30, 0, 0)
    exp = oracle.rk_ctu_batch(ctx, ds, dorg, dstride, W, H, jobs)
    got = engine8.rk_ctu_batch(pc, pr, pds, 30, jobs)
    assert same(got, exp)
    # the int16 plane is released through the same call as every other plane, on an 8-bit context too
    engine8.free_plane(pds)
    assert not pds.origin
    from x265_amd.engine import EngineError
    with pytest.raises(EngineError):
        engine8.free_plane(pds)
    for p in (pc, po, pr):
        engine8.free_plane(p)


def test_rk_golden(engine8):
    g = np.load(G.GOLD + "/rk_golden.npz")
    W, H = 416, 240
    cur, rorig, recon, org, stride, jobs = rk_case(8, W, H, 5)
    assert [G.sha(cur), G.sha(recon)] == G.input_shas()["rk_8"]
    pc = engine8.upload_plane(cur, W, H, M, M, stride)
    po = engine8.upload_plane(rorig, W, H, M, M, stride)
    pr = engine8.upload_plane(recon, W, H, M, M, stride)
    got = engine8.rk_ctu_batch(pc, pr, engine8.rk_ds_plane(po), 30, jobs)
    for k in ("cime", "rime", "fme"):
        assert np.array_equal(got[k]["mv"], g[k + "_mv"]) and np.array_equal(got[k]["cost"], g[k + "_cost"])


@pytest.mark.parametrize("case", [(10, 192, 128, "clip"), (10, 204, 140, "clip"), (10, 416, 240, "clip"), (10, 128, 128, "extreme")])
def test_rk_10bit_vs_oracle(engine10, oracle, case):
    """The RK overload compiles for 16-bit pixels (HIGH_BIT_DEPTH ME-only build); same arithmetic. RIME runs for the
    whole CTU from 16-bit block surfaces (partial CTUs included); "extreme" is a black picture against a white
    reference, where every 8x8 SAD is 64 * 1023, the largest value such a surface entry can take, and every position
    ties. Deeper samples are refused."""
    from x265_amd.engine import Engine, EngineError
    depth, W, H, kind = case
    e = engine10
    cur, rorig, recon, org, stride, jobs = rk_case(depth, W, H, 9)
    if kind == "extreme":
        cur = np.zeros_like(cur)
        recon = np.full_like(recon, 1023)
    pc = e.upload_plane(cur, W, H, M, M, stride)
    po = e.upload_plane(rorig, W, H, M, M, stride)
    pr = e.upload_plane(recon, W, H, M, M, stride)
    pds = e.rk_ds_plane(po)
    qp = 36 + 6 * (depth - 10)
    ds, dorg, dstride = oracle.rk_downsample(depth, rorig, org, W, H, stride, M // 4)
    ctx = oracle.make_ctx(depth, cur, org, stride, [recon], org, stride, False, qp, 0, 0)
    exp = oracle.rk_ctu_batch(ctx, ds, dorg, dstride, W, H, jobs)
    got = e.rk_ctu_batch(pc, pr, pds, qp, jobs)
    assert same(got, exp)
    if kind == "extreme":
        assert (got["rime"]["cost"][:, :64] >= 64 * 1023).all()
    for p in (pc, po, pr, pds):
        e.free_plane(p)
    if kind == "extreme":
        e12 = Engine(0, 12)
        planes = [e12.upload_plane(b, W, H, M, M, stride) for b in (cur, recon)]
        pds12 = e12.rk_ds_plane(planes[1])
        with pytest.raises(EngineError):
            e12.rk_ctu_batch(planes[0], planes[1], pds12, 40, jobs)
        e12.close()


def test_rk_argument_errors(engine8):
    from x265_amd.engine import EngineError
    W, H = 128, 64
    cur, rorig, recon, org, stride, jobs = rk_case(8, W, H, 5)
    pc = engine8.upload_plane(cur, W, H, M, M, stride)
    pds = engine8.rk_ds_plane(pc)
    bad = jobs.copy(); bad["ctu_x"][0] = 8
    with pytest.raises(EngineError):
        engine8.rk_ctu_batch(pc, pc, pds, 30, bad)
    bad = jobs.copy(); bad["mvmax"][0] = (2000, 0)
    with pytest.raises(EngineError):
        engine8.rk_ctu_batch(pc, pc, pds, 30, bad)
    assert len(engine8.rk_ctu_batch(pc, pc, pds, 30, jobs[:0])) == 0


def test_rk_matches_cmodel_golden(engine8):
    """All three stages against records made by the reference's hardware C-model
    (CoarseIntMotionEstimation::CIME, RefineIntMotionEstimation::RimeAndFme), which works on its own
    prefetched, edge-replicated search windows."""
    g = np.load(G.GOLD + "/cmodel_golden.npz")
    W, H, qp, clip, recon_pic, sw, neigh, jobs = cmodel_case()
    cur, org, stride = pad_plane(clip[1], M)
    rorig, _, _ = pad_plane(clip[0], M)
    recon, _, _ = pad_plane(recon_pic, M)
    assert [G.sha(cur), G.sha(rorig), G.sha(recon)] == G.input_shas()["cmodel_8"]
    jobs["mvp"] = g["mvp"]
    jobs["pmv"] = g["pmv"][:, 1:3]
    pc = engine8.upload_plane(cur, W, H, M, M, stride)
    po = engine8.upload_plane(rorig, W, H, M, M, stride)
    pr = engine8.upload_plane(recon, W, H, M, M, stride)
    got = engine8.rk_ctu_batch(pc, pr, engine8.rk_ds_plane(po), qp, jobs)
    for k in ("cime", "rime", "fme"):
        assert np.array_equal(got[k]["mv"], g[k + "_mv"]) and np.array_equal(got[k]["cost"], g[k + "_cost"]), k
