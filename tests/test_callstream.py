"""The reference ENCODER's own call stream (tests/golden/callstream.npz, recorded by tests/golden/make_callstream.py
from two real encodes whose bitstream MD5s equal the recorded ones) replayed through the oracle (CPU, here) and
through the HIP engine (`-m gpu`): what the encoder handed to its motion estimation, and what it got back.

  * Rockchip overload (TEncSearch.cpp:3821 -> motion.cpp:1250): every CTU of three P-frames of the 416x240 / 30-frame
    encode with the neighbour PMVs, cost predictors, range and QP the encoder derived; CIME / RIME / FME of every CU.
  * lookahead (slicetype.cpp:1468): all 66 lowres MV fields the two encodes searched, P and B directions.
  * the host-side rules that build those arguments (x265_amd.workload) against the recorded arguments.
"""
import hashlib

import numpy as np
import pytest

from x265_amd import workload
from x265_amd.engine import RK_CTU_DTYPE, make_jobs
from x265_amd.synth import baseline_clip, pad_plane
import golden_util as G
from util import host_lowres_planes

W, H, M = 416, 240, 304      # M: the reference's luma margin (TComPicYuv.cpp:85-91)


@pytest.fixture(scope="module")
def stream():
    z = np.load(G.GOLD + "/callstream.npz")
    clip = baseline_clip(W, H, 30, 1234)
    assert hashlib.sha256(np.stack(clip).tobytes()).digest() == z["clip_sha256"].tobytes(), \
        "the synthetic clip differs from the one the call stream was recorded on (numpy generator drift)"
    return z, clip


def rk_frames(z, clip):
    """(poc, qp, source buffer, original reference buffer, reconstructed reference buffer, origin, stride, jobs, expected)"""
    for poc in z["rk_pocs"]:
        k = "rk_%d_" % poc
        assert tuple(z[k + "margin"]) == (M, M)
        cur, org, stride = pad_plane(clip[poc], M)
        rorig, _, _ = pad_plane(clip[int(z[k + "ref_poc"])], M)
        recon, _, _ = pad_plane(z[k + "recon"], M)
        jobs = np.ascontiguousarray(z[k + "ctus"]).view(RK_CTU_DTYPE).reshape(-1)
        yield int(poc), int(z[k + "qp"]), cur, rorig, recon, org, stride, jobs, {s: z[k + s] for s in ("cime", "rime", "fme")}


def check_rk(got, exp, tagp):
    assert np.array_equal(got["cime"]["mv"], exp["cime"][:, :2]) and np.array_equal(got["cime"]["cost"], exp["cime"][:, 2]), tagp
    for s in ("rime", "fme"):
        assert np.array_equal(got[s]["mv"], exp[s][:, :, :2]), (tagp, s)
        assert np.array_equal(got[s]["cost"], exp[s][:, :, 2]), (tagp, s)


def test_recorded_encodes(stream):
    z, _ = stream
    assert z["c0_bitstream_md5"].tobytes().hex() == "f2fe9f37c8c7e18d4f4610df4a25c48f"
    assert z["lb_bitstream_md5"].tobytes().hex() == "435ffbe83cfe979fec8e147e5fb9b070"
    assert len(z["la_c0_fields"]) + len(z["la_lb_fields"]) == 66
    # the B-frame run searched both directions
    f = z["la_lb_fields"]
    assert (f[:, 1] > f[:, 0]).any() and (f[:, 1] < f[:, 0]).any()


def test_rk_arguments_follow_host_rules(stream):
    """What TEncSearch::predInterSearch derives per CTU equals what x265_amd.workload derives: effective range
    (TEncSearch.cpp:3717-3742), search window (xSetSearchRange :4188-4215, every CU of the CTU), valid CTU size
    (:3811-3818); neighbour PMVs are whole-sample, even values (:4151-4165)."""
    z, _ = stream
    mx, my = workload.rk_merange(W, H)
    offs = workload.rk_cu_offsets()
    for poc in z["rk_pocs"]:
        k = "rk_%d_" % poc
        jobs = np.ascontiguousarray(z[k + "ctus"]).view(RK_CTU_DTYPE).reshape(-1)
        mine = workload.rk_ctu_jobs(W, H, jobs["pmv"], jobs["mvp"])
        for name in ("ctu_x", "ctu_y", "valid_w", "valid_h", "mvmin", "mvmax", "have_mvd"):
            assert np.array_equal(mine[name], jobs[name]), (poc, name)
        assert (jobs["pmv"] % 2 == 0).all() and (jobs["mvp"] % 2 == 0).all()
        called = z[k + "fme"][:, :, 2] >= 0
        for a in range(len(jobs)):
            for i in np.flatnonzero(called[a]):
                ox, oy, size = offs[i]
                mn, mxv = workload.rk_search_range(W, H, int(jobs["ctu_x"][a]) + ox, int(jobs["ctu_y"][a]) + oy, mx, my)
                assert tuple(z[k + "cu_ranges"][a, i]) == mn + mxv, (poc, a, i)
            # the encoder searches exactly the CUs that lie inside the picture (TEncCu::isOutPic, TEncCu.cpp:475-493)
            inside = np.array([jobs["ctu_x"][a] + ox + s <= W and jobs["ctu_y"][a] + oy + s <= H for ox, oy, s in offs])
            assert np.array_equal(called[a], inside), (poc, a)


def test_rk_call_stream_vs_oracle(stream, oracle):
    z, clip = stream
    for poc, qp, cur, rorig, recon, org, stride, jobs, exp in rk_frames(z, clip):
        ds, dorg, dstride = oracle.rk_downsample(8, rorig, org, W, H, stride, M // 4)
        ctx = oracle.make_ctx(8, cur, org, stride, [recon], org, stride, False, qp, 0, 0)
        check_rk(oracle.rk_ctu_batch(ctx, ds, dorg, dstride, W, H, jobs), exp, poc)


def lookahead_fields_of(z, run):
    p = [int(v) for v in z["la_%s_params" % run]]
    method, subme, qp, merange, wcu, hcu = p[:6]
    assert (method, subme, merange, wcu, hcu) == (1, 1, 16, 26, 15)       # slicetype.h:65-72: always HEX, subme 1, merange 16
    return qp, wcu, hcu, [tuple(int(v) for v in f) for f in z["la_%s_fields" % run]]


@pytest.mark.parametrize("run", ["c0", "lb"])
def test_lookahead_call_stream_vs_oracle(stream, oracle, run):
    """Fields: the whole CU-driver chain (lowres planes, neighbour predictors, bounds, search) from the original
    frames; calls: every recorded call again as an independent search with the predictor / candidates / bounds the
    encoder passed."""
    from util import oracle_search
    z, clip = stream
    qp, wcu, hcu, fields = lookahead_fields_of(z, run)
    low = {}
    for n, (b, r) in enumerate(fields):
        for f in (b, r):
            if f not in low:
                low[f] = host_lowres_planes(oracle, 8, clip[f], 64)
        pa, pb = low[b], low[r]
        ctx = oracle.make_ctx(8, pa[0][0], pa[1], pa[2], pb[0], pa[1], pa[2], True, qp, 1, 1)
        mv, cost = oracle.lookahead_field(ctx, wcu, hcu, 16)
        assert np.array_equal(mv, z["la_%s_mvs" % run][n]) and np.array_equal(cost, z["la_%s_costs" % run][n]), (run, b, r)
        if n % 6 == 0:
            jobs = recorded_lookahead_jobs(z, run, n, wcu, hcu)
            emv, ecost, _, _ = oracle_search(oracle, 8, pa[0][0], pb[0], pa[1], pa[2], jobs, qp, 1, 1, is_lowres=True)
            assert np.array_equal(emv, z["la_%s_mvs" % run][n]) and np.array_equal(ecost, z["la_%s_costs" % run][n]), (run, b, r)


def recorded_lookahead_jobs(z, run, n, wcu, hcu):
    """The calls of field n as independent jobs, arguments exactly as the encoder passed them."""
    jobs = make_jobs(wcu * hcu)
    cu = np.arange(wcu * hcu)
    jobs["pu_x"], jobs["pu_y"], jobs["w"], jobs["h"] = (cu % wcu) * 8, (cu // wcu) * 8, 8, 8
    rng = z["la_%s_ranges" % run][n]
    jobs["mvmin"], jobs["mvmax"] = rng[:, :2], rng[:, 2:]
    jobs["qmvp"] = z["la_%s_qmvp" % run][n]
    jobs["numc"] = z["la_%s_numc" % run][n]
    jobs["mvc"] = z["la_%s_mvc" % run][n]
    jobs["merange"] = 16
    return jobs


# ---------------------------------------------------------------- the HIP engine

@pytest.mark.gpu
def test_rk_call_stream_vs_engine(stream, engine8):
    z, clip = stream
    for poc, qp, cur, rorig, recon, org, stride, jobs, exp in rk_frames(z, clip):
        pc = engine8.upload_plane(cur, W, H, M, M, stride)
        po = engine8.upload_plane(rorig, W, H, M, M, stride)
        pr = engine8.upload_plane(recon, W, H, M, M, stride)
        pds = engine8.rk_ds_plane(po)
        check_rk(engine8.rk_ctu_batch(pc, pr, pds, qp, jobs), exp, poc)
        for p in (pc, po, pr, pds):
            engine8.free_plane(p)


@pytest.mark.gpu
@pytest.mark.parametrize("run", ["c0", "lb"])
def test_lookahead_call_stream_vs_engine(stream, engine8, run):
    z, clip = stream
    qp, wcu, hcu, fields = lookahead_fields_of(z, run)
    low = {}
    for f in sorted({f for pair in fields for f in pair}):
        buf, org, stride = pad_plane(clip[f], M)
        p = engine8.upload_plane(buf, W, H, M, M, stride)
        low[f] = engine8.lowres_init(p)          # Lowres::init on the device, margins as the encoder's
        engine8.free_plane(p)
    mvs, costs, _ = engine8.lookahead_fields([(low[b][0], low[r]) for b, r in fields], wcu, hcu, qp, 1, 1, 16)
    assert np.array_equal(mvs, z["la_%s_mvs" % run]) and np.array_equal(costs, z["la_%s_costs" % run])
    # every recorded call of a few fields once more as independent jobs with the encoder's own arguments
    for n in range(0, len(fields), 5):
        b, r = fields[n]
        res = engine8.search_batch(low[b][0], low[r], recorded_lookahead_jobs(z, run, n, wcu, hcu), qp, 1, 1, is_lowres=True)
        assert np.array_equal(res["mv"], z["la_%s_mvs" % run][n]) and np.array_equal(res["cost"], z["la_%s_costs" % run][n]), (b, r)
    for planes in low.values():
        for p in planes:
            engine8.free_plane(p)
