"""Generate the golden vectors under tests/golden/ from the REFERENCE's own code.

Runs only where oracle/_ref/libx265ref{8,10}.so exist (built by oracle/Makefile from
/root/reference). Inputs are seeded synthetic planes (x265_amd.synth) whose SHA-256
is stored next to the expected outputs, so a drifting generator is detected instead
of silently invalidating the vectors.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle.bindings import Ref, PARTITIONS, METHODS  # noqa: E402
from util import clip_planes, random_jobs, host_lowres_planes, margin_for  # noqa: E402
from oracle.bindings import Oracle  # noqa: E402
from x265_amd.synth import make_clip, pad_plane  # noqa: E402

ME_W, ME_H = 192, 128
LA_W, LA_H = 416, 240
QPS = list(range(70))     # every QP BitCost::setQP accepts (bitcost.cpp:34-59, MAX_QP 69)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def me_cases():
    """(depth, kind, method name, subme) combinations stored in me_golden.npz."""
    cases = []
    for depth in (8, 10):
        for mname in ("dia", "hex", "umh", "star", "full"):
            for subme in (0, 1, 2, 3, 5, 7):
                cases.append((depth, "texture" if subme % 2 else "noise", mname, subme))
    return cases


def me_jobs_for(case):
    depth, kind, mname, subme = case
    seed = 1000 + depth * 100 + METHODS[mname] * 10 + subme
    rng = np.random.default_rng(seed)
    n = 60 if mname != "full" else 16
    return random_jobs(rng, n, ME_W, ME_H, METHODS[mname])


def prim_inputs(depth):
    rng = np.random.default_rng(42 + depth)
    maxv = (1 << depth) - 1
    dt = np.uint8 if depth <= 8 else np.uint16
    a = rng.integers(0, maxv + 1, size=(160, 256)).astype(dt)
    b = rng.integers(0, maxv + 1, size=(160, 256)).astype(dt)
    a[:70, :70] = 0
    b[:70, :70] = maxv          # extreme differences (saturating SAD / Hadamard range)
    a[0:70:2, 0:70:2] = maxv
    return a.reshape(-1), b.reshape(-1)


def prim_offsets(part_index):
    """(fenc offset, ref offset) pairs used per partition: aligned and unaligned."""
    return [(0, 0), (5 * 256 + 3 + part_index, 9 * 256 + 7 + 2 * part_index), (80 * 256 + 64, 75 * 256 + 33)]


def main():
    out = {}
    make_cost_tables(None)

    # ---- primitives
    prim = {}
    for depth in (8, 10):
        ref = Ref(depth)
        a, b = prim_inputs(depth)
        d = {"input_sha": [sha(a), sha(b)], "parts": {}}
        for pi, (w, h) in enumerate(PARTITIONS):
            rows = []
            for (ao, bo) in prim_offsets(pi):
                fenc = np.ascontiguousarray(a.reshape(160, 256)[:64, :64]).reshape(-1)
                offs = [bo + k * 3 + k * 256 for k in range(4)]
                so = bo + 10 * 256 + 10
                e = {"sad": ref.sad(a, ao, 256, b, bo, 251, w, h), "satd": ref.satd(a, ao, 256, b, bo, 251, w, h),
                     "sa8d": ref.sa8d_inter(a, ao, 256, b, bo, 251, w, h),
                     "x3": ref.sad_x3(fenc, b, offs, 251, w, h), "x4": ref.sad_x4(fenc, b, offs, 251, w, h),
                     "avg": sha(ref.pixelavg(a, ao, 256, b, bo, 251, w, h))}
                for ci in (1, 2, 3):
                    e["hpp%d" % ci] = sha(ref.luma_hpp(b, so, 256, ci, w, h))
                    e["vpp%d" % ci] = sha(ref.luma_vpp(b, so, 256, ci, w, h))
                    e["hps%d" % ci] = sha(ref.luma_hps(b, so, 256, ci, 0, w, h))
                    im = ref.luma_hps(b, so, 256, ci, 1, w, h)
                    e["hpsx%d" % ci] = sha(im)
                    e["vsp%d" % ci] = sha(ref.luma_vsp(im.reshape(-1), 3 * w, w, ci, w, h))
                rows.append(e)
            d["parts"]["%dx%d" % (w, h)] = rows
        d["lowres"] = [sha(o) for o in ref.lowres_core(a, 0, 256, 100, 70)]
        prim[str(depth)] = d
    json.dump(prim, open(os.path.join(HERE, "prim_golden.json"), "w"), indent=0, sort_keys=True)

    # ---- motion search
    me = {}
    shas = {}
    for case in me_cases():
        depth, kind, mname, subme = case
        ref = Ref(depth)
        fenc, rpl, org, stride = clip_planes(ME_W, ME_H, depth, kind, margin=margin_for(METHODS[mname]))
        shas["%d_%s_%d" % (depth, kind, margin_for(METHODS[mname]))] = [sha(fenc), sha(rpl)]
        jobs = me_jobs_for(case)
        qp = 20 + 6 * (depth - 8) + subme
        ref.me_config(fenc, org, stride, qp, METHODS[mname], subme)
        exp = np.zeros((len(jobs), 3), np.int32)
        for i, j in enumerate(jobs):
            off = int(j["pu_y"]) * stride + int(j["pu_x"])
            mvc = [tuple(int(v) for v in j["mvc"][k]) for k in range(int(j["numc"]))]
            mv, cost = ref.me_search(off, int(j["w"]), int(j["h"]), [rpl], org, stride, False,
                                     tuple(int(v) for v in j["mvmin"]), tuple(int(v) for v in j["mvmax"]),
                                     tuple(int(v) for v in j["qmvp"]), mvc, int(j["merange"]))
            exp[i] = (mv[0], mv[1], cost)
        key = "%d_%s_%s_%d" % case
        me[key + "_jobs"] = jobs.view(np.uint8).reshape(len(jobs), -1)
        me[key + "_exp"] = exp
    np.savez_compressed(os.path.join(HERE, "me_golden.npz"), **me)

    # ---- lookahead fields (config-1 geometry: 416x240 -> 26x15 lowres CUs)
    la = {}
    orc = Oracle()
    for depth in (8, 10):
        ref = Ref(depth)
        clip = make_clip(LA_W, LA_H, 4, seed=3, depth=depth)
        shas["la_%d" % depth] = [sha(clip[3]), sha(clip[1])]
        # the lowres planes come from the reference's own downscaler
        pa = host_lowres_planes(ref_as_orc(ref), depth, clip[3], 64)
        pb = host_lowres_planes(ref_as_orc(ref), depth, clip[1], 64)
        planes_a, org, stride, wcu, hcu = pa[0], pa[1], pa[2], pa[3], pa[4]
        for (method, subme) in ((1, 1), (3, 2), (0, 0), (2, 5)):
            mvs, costs = ref.lookahead_field(planes_a[0], pb[0], org, stride, wcu, hcu, 12 + 6 * (depth - 8), method, subme, 16)
            la["%d_%d_%d_mvs" % (depth, method, subme)] = mvs
            la["%d_%d_%d_costs" % (depth, method, subme)] = costs
    np.savez_compressed(os.path.join(HERE, "lookahead_golden.npz"), **la)

    # ---- Rockchip CIME / RIME / FME (8-bit, as the full RK encoder is 8-bit only)
    from test_gpu_rk import rk_case, M
    ref = Ref(8)
    W, H = 416, 240
    cur, rorig, recon, org, stride, jobs = rk_case(8, W, H, 5)
    shas["rk_8"] = [sha(cur), sha(recon)]
    ds, dorg, dstride = ref.rk_ds_plane(rorig, org, W, H, stride, M // 4)
    recs = [ref.rk_ctu(cur, org, stride, 30, recon, org, stride, ds, dorg, dstride, W, H, j) for j in jobs]
    rk = {}
    for k in ("cime", "rime", "fme"):
        rk[k + "_mv"] = np.stack([r[k]["mv"] for r in recs])
        rk[k + "_cost"] = np.stack([r[k]["cost"] for r in recs])
    np.savez_compressed(os.path.join(HERE, "rk_golden.npz"), **rk)
    make_cmodel(shas)
    make_frame_cost(shas)
    make_wp(shas)
    make_bipred(shas)
    make_merge(shas)
    make_mc(shas)
    json.dump(shas, open(os.path.join(HERE, "input_sha.json"), "w"), indent=0, sort_keys=True)
    print("golden vectors written to", HERE)


def make_cost_tables(shas):
    """BitCost::setQP tables (bitcost.cpp:34-83) of the reference build. The tables are symmetric in the MVD and grow
    with |mvd|, so only the non-negative half is stored, as run lengths: entry v of `steps` is the first |mvd| whose
    cost is at least v (golden_util.cost_tables expands them again)."""
    ref8 = Ref(8)
    out = {}
    for qp in QPS:
        t = ref8.cost_table(qp)                      # 65535 entries, mvd -32767 .. 32767
        half = t[32767:].astype(np.int64)
        assert np.array_equal(t[:32768][::-1], t[32767:]) and (np.diff(half) >= 0).all()
        out["qp%d" % qp] = np.searchsorted(half, np.arange(int(half[-1]) + 2), side="left").astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "cost_tables.npz"), **out)


class ref_as_orc:
    """Gives host_lowres_planes the reference's own downscaler behind the oracle's call shape."""

    def __init__(self, ref):
        self.ref = ref

    def lowres_core(self, depth, src, so, ss, w, h):
        return self.ref.lowres_core(src, so, ss, w, h)


FC_KEYS = ("mvs0", "mvcosts0", "mvs1", "mvcosts1", "intra_cost", "lowres_costs", "row_satds_intra", "row_satds", "totals")


def frame_cost_cases():
    """(name, clip kind, seed, b, d0, d1, with AQ factors) of the lookahead frame-cost vectors (416x240, 8-bit)."""
    out = []
    for kind, seed in (("texture", 3), ("noise", 11)):
        for (b, d0, d1) in ((2, 0, 0), (2, 2, 0), (2, 1, 2), (1, 1, 3)):
            for aq in (0, 1):
                out.append(("%s_%d_%d_%d_%d" % (kind, b, d0, d1, aq), kind, seed, b, d0, d1, aq))
    return out


def frame_cost_inv_qscale(ncu):
    return np.random.default_rng(77).integers(96, 520, ncu).astype(np.int32)


def make_frame_cost(shas):
    """The reference's own EstimateRow::estimateCUCost run over whole frames (slicetype.cpp:1401-1582, in the
    serial order of processRow) and the row sums of estimateFrameCost (:1193-1204)."""
    ref = Ref(8)
    out = {}
    for name, kind, seed, b, d0, d1, aq in frame_cost_cases():
        clip = make_clip(LA_W, LA_H, 5, seed=seed, depth=8, kind=kind)
        shas["fc_%s" % kind] = [sha(f) for f in clip]
        P = [host_lowres_planes(ref_as_orc(ref), 8, f, 64) for f in clip]
        org, stride, wcu, hcu = P[0][1], P[0][2], P[0][3], P[0][4]
        r = ref.estimate_frame(P[b][0], P[b - d0][0] if d0 else None, P[b + d1][0] if d1 else None, org, stride, wcu, hcu,
                               d0, d1, inv_qscale=frame_cost_inv_qscale(wcu * hcu) if aq else None)
        for k in FC_KEYS:
            out[name + "_" + k] = r[k]
    np.savez_compressed(os.path.join(HERE, "framecost_golden.npz"), **out)


def wp_stats(img):
    """wp_sum / wp_ssd of a plane as the rate control computes them (ratecontrol.cpp:44-56, :185-192)."""
    p = img.astype(np.uint64)
    n, s, q = p.size, int(p.sum()), int((p * p).sum())
    return q - (s * s + n // 2) // n, s


WP_WEIGHTS = ((64, 6, 0), (50, 6, 7), (127, 7, -20), (33, 5, 100), (1, 0, 3), (90, 6, -128))
WP_FADES = ((1.0, 0), (0.8, 10), (0.6, -5), (1.3, 0), (1.0, 30))


def wp_fade_frames(depth, fade, off):
    """(faded frame b, reference frame p0) of the weighted-prediction vectors."""
    clip = make_clip(LA_W, LA_H, 2, seed=3, depth=depth)
    maxv = (1 << depth) - 1
    f = np.clip(clip[1].astype(np.float64) * fade + off * (1 << (depth - 8)), 0, maxv).astype(clip[1].dtype)
    return f, clip[0]


def wp_intra_costs(ncu):
    return np.random.default_rng(123).integers(300, 4000, ncu).astype(np.int32)


def make_wp(shas):
    """MotionReference::applyWeight, CostEstimate::weightCostLuma and weightsAnalyse run by the reference itself."""
    out = {}
    for depth in (8, 10):
        ref = Ref(depth)
        img = make_clip(200, 136, 1, seed=5, depth=depth)[0]
        buf, org, stride = pad_plane(img, 48, 32)
        shas["wp_apply_%d" % depth] = sha(buf)
        for (w, d, o) in WP_WEIGHTS:
            out["%d_apply_%d_%d_%d" % (depth, w, d, o)] = np.frombuffer(bytes.fromhex(sha(ref.apply_weight(buf, 200, 136, 48, 32, stride, w, d, o))), np.uint8)
        for fade, off in WP_FADES:
            f, r = wp_fade_frames(depth, fade, off)
            PF = host_lowres_planes(ref_as_orc(ref), depth, f, 64)
            PR = host_lowres_planes(ref_as_orc(ref), depth, r, 64)
            org, stride, wcu, hcu = PF[1], PF[2], PF[3], PF[4]
            ic = wp_intra_costs(wcu * hcu)
            padded = PF[6] + 128
            key = "%d_fade_%g_%d" % (depth, fade, off)
            weighted, w, wbuf, _ = ref.weights_analyse(PF[0], PR[0], org, stride, padded, wcu, hcu, ic, wp_stats(f) + wp_stats(r))
            out[key + "_decision"] = np.array([int(weighted)] + (list(w) if weighted else [0, 0, 0]), np.int32)
            out[key + "_cost_plain"] = np.array([ref.weight_cost_luma(PF[0][0], PR[0][0], org, stride, wcu, hcu, ic)], np.uint32)
            out[key + "_cost_wp"] = np.array([ref.weight_cost_luma(PF[0][0], PR[0][0], org, stride, wcu, hcu, ic, (50, 6, 7))], np.uint32)
            if weighted:
                out[key + "_planes"] = np.stack([np.frombuffer(bytes.fromhex(sha(wbuf[i])), np.uint8) for i in range(4)])
    np.savez_compressed(os.path.join(HERE, "wp_golden.npz"), **out)


def bipred_case(depth):
    """(fenc, ref0, ref1 padded buffers, origin, stride, jobs) of the bi-prediction vectors."""
    from oracle.bindings import BIPRED_JOB_DTYPE
    clip = make_clip(ME_W, ME_H, 3, seed=8, depth=depth)
    f, org, stride = pad_plane(clip[1], 96)
    r0, _, _ = pad_plane(clip[0], 96)
    r1, _, _ = pad_plane(clip[2], 96)
    rng = np.random.default_rng(3)
    parts = PARTITIONS[1:]
    jobs = np.zeros(12 * len(parts), BIPRED_JOB_DTYPE)
    for i in range(len(jobs)):
        w, h = parts[i % len(parts)]
        j = jobs[i]
        j["pu_x"], j["pu_y"] = int(rng.integers(0, (ME_W - w) // 4 + 1)) * 4, int(rng.integers(0, (ME_H - h) // 4 + 1)) * 4
        j["w"], j["h"] = w, h
        j["mv0"], j["mv1"] = rng.integers(-120, 121, 2), rng.integers(-120, 121, 2)
        if i % 7 == 0:
            j["mv0"] = (j["mv0"] // 4) * 4
        if i % 11 == 0:
            j["mv1"] = 0
        if i % 13 == 0:
            j["mv0"][1] = (j["mv0"][1] // 4) * 4
    return f, r0, r1, org, stride, jobs


def make_bipred(shas):
    out = {}
    for depth in (8, 10):
        f, r0, r1, org, stride, jobs = bipred_case(depth)
        shas["bipred_%d" % depth] = [sha(f), sha(r0), sha(r1), sha(jobs)]
        out["%d" % depth] = Ref(depth).bipred_batch(f, r0, r1, org, stride, jobs)
    np.savez_compressed(os.path.join(HERE, "bipred_golden.npz"), **out)


def merge_case(depth):
    """(fenc, [3 refs] padded buffers, origin, stride, jobs) of the merge-candidate vectors."""
    from oracle.bindings import MERGE_JOB_DTYPE
    clip = make_clip(ME_W, ME_H, 4, seed=8, depth=depth)
    f, org, stride = pad_plane(clip[1], 96)
    refs = [pad_plane(clip[k], 96)[0] for k in (0, 2, 3)]
    rng = np.random.default_rng(3)
    jobs = np.zeros(240, MERGE_JOB_DTYPE)
    for i, j in enumerate(jobs):
        s = (8, 16, 32, 64)[i % 4]
        j["pu_x"], j["pu_y"], j["size"] = int(rng.integers(0, (ME_W - s) // 8 + 1)) * 8, int(rng.integers(0, (ME_H - s) // 8 + 1)) * 8, s
        j["dir"], j["ref0"], j["ref1"] = 1 + i % 3, int(rng.integers(0, 3)), int(rng.integers(0, 3))
        j["mv0"], j["mv1"] = rng.integers(-120, 121, 2), rng.integers(-120, 121, 2)
        if i % 5 == 0:
            j["mv0"] = (j["mv0"] // 4) * 4
        if i % 7 == 0:
            j["mv1"][0] = (j["mv1"][0] // 4) * 4
        if i % 11 == 0:
            j["mv1"][1] = (j["mv1"][1] // 4) * 4
    return f, refs, org, stride, jobs


def make_merge(shas):
    out = {}
    for depth in (8, 10):
        f, refs, org, stride, jobs = merge_case(depth)
        shas["merge_%d" % depth] = [sha(f)] + [sha(r) for r in refs] + [sha(jobs)]
        out["%d" % depth] = Ref(depth).merge_sa8d_batch(f, refs, org, stride, jobs)
    np.savez_compressed(os.path.join(HERE, "merge_golden.npz"), **out)


def mc_case(depth, weighted=False):
    """(reference pictures, job batches, weights) of the motion-compensation vectors: pictures as [(buffer, origin,
    stride)] * 3; the weighted set mixes explicit weighted prediction (flags bit 2) into three PUs of four."""
    from oracle.bindings import MC_JOB_DTYPE, MC_WEIGHT_DTYPE
    from util import mc_pictures, mc_job_batches, mc_weights
    if not weighted:
        return mc_pictures(depth, 31, nrefs=3), mc_job_batches(np.random.default_rng(32), MC_JOB_DTYPE, 14, nrefs=3), None
    rng = np.random.default_rng(34)
    return (mc_pictures(depth, 33, nrefs=3), mc_job_batches(rng, MC_JOB_DTYPE, 14, nrefs=3, weighted=True),
            mc_weights(rng, MC_WEIGHT_DTYPE, 3))


def make_mc(shas):
    """SHA-256 of the three planes of the prediction picture after each batch, computed by the reference's primitives in
    TComPrediction::motionCompensation's order (Ref.mc_batch)."""
    from util import mc_blank_pred
    out = {}
    for depth in (8, 10):
        for weighted in (False, True):
            refs, batches, wts = mc_case(depth, weighted)
            key = "%d%s" % (depth, "w" if weighted else "")
            shas["mc_" + key] = [sha(pl[0]) for pic in refs for pl in pic] + [sha(np.concatenate(batches))] + ([sha(wts)] if weighted else [])
            digests = []
            for jobs in batches:
                pred = mc_blank_pred(depth)
                Ref(depth).mc_batch(refs, pred, jobs, wts)
                digests.append([np.frombuffer(bytes.fromhex(sha(pl[0])), np.uint8) for pl in pred])
            out[key] = np.array(digests, np.uint8)
    np.savez_compressed(os.path.join(HERE, "mc_golden.npz"), **out)


def make_cmodel(shas):
    """Records of the reference's hardware C-model: coarse stage (CoarseIntMotionEstimation::CIME,
    hardwareC/inter.cpp:679-734), refine + fractional stages for the 85 CUs of each CTU
    (RefineIntMotionEstimation::RimeAndFme, :2521-2624), and the window centres / cost predictors the model
    derived from the neighbour MVs (setPmv :882-989, setMvp :755-842, :991-1080)."""
    from test_gpu_rk import cmodel_case, M
    ref = Ref(8)
    W, H, qp, clip, recon_pic, (sw_w, sw_h), neigh, jobs = cmodel_case()
    cur, org, stride = pad_plane(clip[1], M)
    rorig, _, _ = pad_plane(clip[0], M)
    recon, _, _ = pad_plane(recon_pic, M)
    shas["cmodel_8"] = [sha(cur), sha(rorig), sha(recon)]
    out = {k: [] for k in ("cime_mv", "cime_cost", "rime_mv", "rime_cost", "fme_mv", "fme_cost", "pmv", "mvp")}
    for a in range(len(jobs)):
        mv, cost, mvp = ref.cmodel_cime(clip[1], clip[0], W, H, qp, a, sw_w, sw_h, neigh[a])
        rime, fme, pmv, mvp_r = ref.cmodel_rime_fme(clip[1], recon_pic, W, H, qp, a, neigh[a], mv)
        assert np.array_equal(mvp, mvp_r)
        out["cime_mv"].append(mv); out["cime_cost"].append(cost)
        out["rime_mv"].append(rime[:, :2]); out["rime_cost"].append(rime[:, 2])
        out["fme_mv"].append(fme[:, :2]); out["fme_cost"].append(fme[:, 2])
        out["pmv"].append(pmv); out["mvp"].append(mvp)
    np.savez_compressed(os.path.join(HERE, "cmodel_golden.npz"),
                        **{k: np.array(v, np.int32 if k.endswith("cost") else np.int16) for k, v in out.items()})


if __name__ == "__main__":
    if sys.argv[1:] in (["costtables"], ["cmodel"], ["framecost"], ["wp"], ["bipred"], ["merge"], ["mc"]):
        # only (re)generate the C-model / frame-cost records, keeping the other files as they are
        shas_path = os.path.join(HERE, "input_sha.json")
        shas_ = json.load(open(shas_path))
        {"costtables": make_cost_tables, "cmodel": make_cmodel, "framecost": make_frame_cost, "wp": make_wp, "bipred": make_bipred, "merge": make_merge, "mc": make_mc}[sys.argv[1]](shas_)
        json.dump(shas_, open(shas_path, "w"), indent=0, sort_keys=True)
    else:
        main()
