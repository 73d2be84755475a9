"""GPU: the BASELINE.json configurations at their full picture sizes.

  configs[1]  1080p 8-bit, --me star --merange 57           -> lookahead field + full-encode pattern search
  configs[2]  4K 8-bit, --me full, merange 92, 3 references   -> exhaustive-search kernel
  configs[3]  4K Main10, --subme 5                             -> 16-bit interpolation + SATD refinement
  configs[4]  8K 8-bit lookahead                               -> one lowres MV field of a 7680x4320 frame

A CPU replay of every PU at these sizes would take minutes, so each case checks (a) a seeded sample of
the PUs against the oracle and (b) a size-independent property over all of them.
"""
import numpy as np
import pytest

from x265_amd.engine import METHODS, make_jobs
from x265_amd.synth import make_clip, pad_plane
from x265_amd import workload
from util import oracle_search, host_lowres_planes

pytestmark = pytest.mark.gpu


def _sample(jobs, n, seed):
    idx = np.sort(np.random.default_rng(seed).choice(len(jobs), n, replace=False))
    return idx, jobs[idx]


def test_1080p_star_merange57_and_lookahead(engine8, oracle):
    """The benchmark's own workload: the 120x68 HEX lookahead field of a 1080p frame pair (whole field against the
    oracle), then STAR / merange 57 / subme 5 over all 42 900 square PUs with predictors from that field; a seeded sample
    of every PU size plus the PUs on the four picture borders against the oracle."""
    from oracle.bindings import MEStats
    W, H, M, MR, QP = 1920, 1080, 128, 57, 32
    clip = make_clip(W, H, 2, seed=1234)
    cur, org, stride = pad_plane(clip[1], M)
    ref, _, _ = pad_plane(clip[0], M)
    pc = engine8.upload_plane(cur, W, H, M, M, stride)
    pr = engine8.upload_plane(ref, W, H, M, M, stride)
    lc, lr = engine8.lowres_init(pc), engine8.lowres_init(pr)
    wcu, hcu = 120, 68
    mvs, costs, stats = engine8.lookahead_fields([(lc[0], lr)], wcu, hcu, 12, METHODS["hex"], 1, 16, want_stats=True)
    hc = engine8.download_plane(lc[0]).reshape(-1)
    hr = [engine8.download_plane(p).reshape(-1) for p in lr]
    lorg = lc[0].margin_y * lc[0].stride + lc[0].margin_x
    lctx = oracle.make_ctx(8, hc, lorg, lc[0].stride, hr, lorg, lc[0].stride, True, 12, METHODS["hex"], 1)
    st = MEStats()
    emv, ecost = oracle.lookahead_field(lctx, wcu, hcu, 16, st)
    assert np.array_equal(mvs[0], emv) and np.array_equal(costs[0], ecost)
    assert int(stats["sad_calls"].sum()) == st.sad_calls and int(stats["subpel_calls"].sum()) == st.subpel_calls
    jobs = workload.square_pu_jobs(W, H, mvs[0], wcu, hcu, MR)
    assert len(jobs) == 42900
    res = engine8.search_batch(pc, pr, jobs, QP, METHODS["star"], 5)
    pick = []
    rng = np.random.default_rng(57)
    for s in (64, 32, 16, 8):
        of_size = np.flatnonzero(jobs["w"] == s)
        border = of_size[(jobs["pu_x"][of_size] == 0) | (jobs["pu_y"][of_size] == 0) |
                         (jobs["pu_x"][of_size] + s >= W - (W % s)) | (jobs["pu_y"][of_size] + s >= H - (H % s))]
        pick += list(rng.choice(of_size, 45, replace=False)) + list(rng.choice(border, 15, replace=False))
    idx = np.unique(np.array(pick))
    assert len(idx) >= 200
    mv, cost, sc, qc = oracle_search(oracle, 8, cur, [ref], org, stride, jobs[idx], QP, METHODS["star"], 5)
    assert np.array_equal(res["mv"][idx], mv) and np.array_equal(res["cost"][idx], cost)
    assert np.array_equal(res["sad_calls"][idx], sc) and np.array_equal(res["subpel_calls"][idx], qc)
    # every result lies inside its range (quarter-pel) and was refined at least once unless its cost is 0
    q = res["mv"].astype(np.int32)
    assert (q >= jobs["mvmin"].astype(np.int32) * 4 - 3).all() and (q <= jobs["mvmax"].astype(np.int32) * 4 + 3).all()
    assert ((res["subpel_calls"] >= 1 + 9 + 8) | (res["cost"] == 0)).all()
    for p in [pc, pr] + lc + lr:
        engine8.free_plane(p)


def test_4k_exhaustive_search_three_refs(engine8, oracle):
    W, H, M, MR = 3840, 2160, 128, 92
    clip = make_clip(W, H, 4, seed=41)
    cur, org, stride = pad_plane(clip[3], M)
    pc = engine8.upload_plane(cur, W, H, M, M, stride)
    rng = np.random.default_rng(9)
    wcu, hcu = (W // 2 + 7) // 8, (H // 2 + 7) // 8
    la = rng.integers(-16, 17, (wcu * hcu, 2)).astype(np.int16)
    jobs64 = workload.square_pu_jobs(W, H, la, wcu, hcu, MR, sizes=(64,))
    jobs32 = workload.square_pu_jobs(W, H, la, wcu, hcu, MR, sizes=(32,))
    jobs16 = workload.square_pu_jobs(W, H, la, wcu, hcu, MR, sizes=(16,))
    jobs8 = workload.square_pu_jobs(W, H, la, wcu, hcu, MR, sizes=(8,))
    for r in range(3):   # 3 reference pictures, one search batch each
        ref, _, _ = pad_plane(clip[r], M)
        pr = engine8.upload_plane(ref, W, H, M, M, stride)
        # the first reference with all four PU sizes (the 8x8 and 16x16 scans have routines of their own, and a 185 x 185
        # range is scanned in strips at every size), the others with two
        for jobs, n in ((jobs64, 6), (jobs16, 24)) + (((jobs32, 12), (jobs8, 24)) if r == 0 else ()):
            res = engine8.search_batch(pc, pr, jobs, 32, METHODS["full"], 5)
            idx, sub = _sample(jobs, n, 100 + r)
            mv, cost, sc, qc = oracle_search(oracle, 8, cur, [ref], org, stride, sub, 32, METHODS["full"], 5)
            assert np.array_equal(res["mv"][idx], mv) and np.array_equal(res["cost"][idx], cost), r
            assert np.array_equal(res["sad_calls"][idx], sc) and np.array_equal(res["subpel_calls"][idx], qc), r
            # every PU visited its whole range once, plus at most two start-up points
            rngx = jobs["mvmax"][:, 0].astype(np.int64) - jobs["mvmin"][:, 0] + 1
            rngy = jobs["mvmax"][:, 1].astype(np.int64) - jobs["mvmin"][:, 1] + 1
            extra = res["sad_calls"].astype(np.int64) - rngx * rngy
            assert (extra >= 0).all() and (extra <= 2).all()
        engine8.free_plane(pr)
    # property: against a copy of the source shifted by (+5, -7) the scan finds exactly that vector with SAD 0
    img = make_clip(W, H, 1, seed=42, kind="noise")[0]
    buf, _, _ = pad_plane(img, M)
    sh, _, _ = pad_plane(np.roll(img, (-7, 5), axis=(0, 1)), M)
    pn, ps = engine8.upload_plane(buf, W, H, M, M, stride), engine8.upload_plane(sh, W, H, M, M, stride)
    xs, ys = np.meshgrid(np.arange(128, W - 192, 64), np.arange(128, H - 192, 64))
    jobs = make_jobs(xs.size)
    jobs["pu_x"], jobs["pu_y"], jobs["w"], jobs["h"] = xs.ravel(), ys.ravel(), 64, 64
    jobs["mvmin"] = (-MR, -MR); jobs["mvmax"] = (MR, MR); jobs["merange"] = MR
    res = engine8.search_batch(pn, ps, jobs, 30, METHODS["full"], 0)
    import golden_util as G
    t = G.cost_tables()[30].astype(np.int64)      # the reference's own BitCost table
    assert (res["mv"] == (20, -28)).all()
    assert (res["cost"] == ((t[32767 + 20] + t[32767 - 28]) & 0xffff)).all()
    for p in (pc, pn, ps):
        engine8.free_plane(p)


def test_4k_main10_subpel_refine(engine10, oracle):
    W, H, M = 3840, 2160, 96
    clip = make_clip(W, H, 2, seed=43, depth=10)
    cur, org, stride = pad_plane(clip[1], M)
    ref, _, _ = pad_plane(clip[0], M)
    pc = engine10.upload_plane(cur, W, H, M, M, stride)
    pr = engine10.upload_plane(ref, W, H, M, M, stride)
    rng = np.random.default_rng(10)
    wcu, hcu = (W // 2 + 7) // 8, (H // 2 + 7) // 8
    la = rng.integers(-20, 21, (wcu * hcu, 2)).astype(np.int16)
    jobs = workload.square_pu_jobs(W, H, la, wcu, hcu, 57, clip_margin=70)
    for mname in ("star", "hex"):
        res = engine10.search_batch(pc, pr, jobs, 38, METHODS[mname], 5)
        idx, sub = _sample(jobs, 160, 7)
        mv, cost, sc, qc = oracle_search(oracle, 10, cur, [ref], org, stride, sub, 38, METHODS[mname], 5)
        assert np.array_equal(res["mv"][idx], mv) and np.array_equal(res["cost"][idx], cost), mname
        assert np.array_equal(res["sad_calls"][idx], sc) and np.array_equal(res["subpel_calls"][idx], qc), mname
        # subme 5 measures the start-up predictor(s), then one half-pel and one quarter-pel round of
        # 8 points with the centre re-measured once (motion.cpp:53-63, :1100-1141) unless the full-pel cost is 0
        assert ((res["subpel_calls"] >= 1 + 9 + 8) | (res["cost"] == 0)).all()
    # a picture searched against itself: every PU keeps MV 0 at cost 0 and skips the refinement
    res = engine10.search_batch(pc, pc, jobs, 38, METHODS["star"], 5)
    zero = (jobs["qmvp"] == 0).all(axis=1)
    assert (res["cost"][zero] == 0).all() and (res["mv"][zero] == 0).all()
    engine10.free_plane(pc)
    engine10.free_plane(pr)


def test_8k_lookahead_field(engine8, oracle):
    from oracle.bindings import MEStats
    W, H = 7680, 4320
    clip = make_clip(W, H, 2, seed=44)
    pa = host_lowres_planes(oracle, 8, clip[1], 64)
    pb = host_lowres_planes(oracle, 8, clip[0], 64)
    wcu, hcu = pa[3], pa[4]
    assert (wcu, hcu) == (480, 270)
    planes = []
    for img in (clip[1], clip[0]):
        buf, org, stride = pad_plane(img, 64)
        p = engine8.upload_plane(buf, W, H, 64, 64, stride)
        planes.append(engine8.lowres_init(p))
        engine8.free_plane(p)
    la, lb = planes
    for k in range(4):
        assert np.array_equal(engine8.download_plane(lb[k]).reshape(-1), pb[0][k])
    ctx = oracle.make_ctx(8, pa[0][0], pa[1], pa[2], pb[0], pa[1], pa[2], True, 12, 1, 1)
    st = MEStats()
    mv, cost = oracle.lookahead_field(ctx, wcu, hcu, 16, st)
    mvs, costs, stats = engine8.lookahead_fields([(la[0], lb)], wcu, hcu, 12, 1, 1, 16, want_stats=True)
    assert np.array_equal(mvs[0], mv) and np.array_equal(costs[0], cost)
    assert int(stats["sad_calls"].sum()) == st.sad_calls and int(stats["subpel_calls"].sum()) == st.subpel_calls


@pytest.mark.parametrize("mname", ["star", "hex", "dia"])
def test_1080p_headline_launch_lists(engine8, oracle, mname):
    """The launches of the benchmark's step as bench.py builds them (workload.class_job_list): per PU class one job list
    over several frame pairs -- the 64x64 PUs each followed by their four 32x32 children (x265amd_search_batch_pairs_dev with the
    statement that the list holds only squares), the 16x16 PUs and the list of their 8x8 children
    (x265amd_search_batch_pairs_family_dev). Two frame pairs: one of steady
    motion, where nearly every search ends with its first three pattern rounds, and one across the clip's wrap-around
    (a jump of 29 samples the lookahead cannot follow), where many PUs are handed on to the resumed search. A sample
    of more than 200 PUs, marked ones among them, against the oracle. The same lists with the hexagon and the diamond
    search, whose first ring + square / first round the same kernels measure (x265amd_search_batch_pairs_family_method_dev)."""
    from x265_amd.engine import RESULT_DTYPE, SQUARE_ONLY
    method = METHODS[mname]
    W, H, M, MR, QP = 1920, 1080, 128, 57, 32
    clip = make_clip(W, H, 12, seed=1234)
    frames = [(1, 0), (11, 10)]
    up = {}
    for f in {f for pr_ in frames for f in pr_}:
        buf, org, stride = pad_plane(clip[f], M)
        up[f] = (engine8.upload_plane(buf, W, H, M, M, stride), buf)
    low = {f: engine8.lowres_init(up[f][0]) for f in up}
    wcu, hcu = 120, 68
    mvs, _, _ = engine8.lookahead_fields([(low[a][0], low[b]) for a, b in frames], wcu, hcu, 12, METHODS["hex"], 1, 16)
    host_jobs = [workload.square_pu_jobs(W, H, mvs[i], wcu, hcu, MR) for i in range(2)]
    d_pairs = engine8.pairs_upload([(up[a][0], up[b][0]) for a, b in frames])
    res = [np.zeros(len(host_jobs[0]), RESULT_DTYPE) for _ in range(2)]
    marked = {}
    for dim in (64, 16):
        if dim == 64:
            cj, ci = workload.class_job_list(host_jobs, (64, 32), families=True)
            # quadtree order: every 64x64 PU with all four children inside the picture is followed by them
            k = int(np.flatnonzero(cj["w"] == 64)[0])
            assert list(cj["w"][k:k + 5]) == [64, 32, 32, 32, 32] and (cj["pu_x"][k + 1:k + 5] - cj["pu_x"][k]).tolist() == [0, 32, 0, 32]
            npar = 0
        else:
            # the 16x16 PUs and the list of their 8x8 children: 4 k + q = quadrant q of parent k
            pj, pi, kj, ki = workload.family_lists(host_jobs, 16)
            k = len(pj) - 7
            assert (kj["pu_x"][4 * k:4 * k + 4] - pj["pu_x"][k]).tolist() == [0, 8, 0, 8] and (kj["reserved"][4 * k:4 * k + 4] == pj["reserved"][k]).all()
            cj, ci, npar = np.concatenate([pj, kj]), np.concatenate([pi, ki]), len(pj)
        d_jobs, d_res = engine8.device_alloc(cj.nbytes), engine8.device_alloc(len(cj) * RESULT_DTYPE.itemsize)
        engine8.upload(d_jobs, cj)
        if dim == 64:
            engine8.search_batch_pairs_dev(d_pairs, 2, d_jobs, len(cj), d_res, QP, method, 5, max_pu_dim=dim | SQUARE_ONLY)
        else:
            engine8.search_batch_pairs_family_dev(d_pairs, 2, d_jobs, npar, d_res, d_jobs + 40 * npar, len(cj) - npar,
                                                  d_res + 16 * npar, QP, 5, method)
        marked[dim] = engine8.marked_count()
        got = engine8.download(d_res, (len(cj),), RESULT_DTYPE)
        for i in range(2):
            sel = ci[:, 0] == i
            res[i][ci[sel, 1]] = got[sel]
        if dim == 16:
            # the same two lists as launches of their own: identical records
            for off, n_, d_ in ((0, npar, 16), (npar, len(cj) - npar, 8)):
                engine8.search_batch_pairs_dev(d_pairs, 2, d_jobs + 40 * off, n_, d_res + 16 * off, QP, method, 5,
                                               max_pu_dim=d_ | SQUARE_ONLY)
            again = engine8.download(d_res, (len(cj),), RESULT_DTYPE)
            assert np.array_equal(again, got)
        engine8.device_free(d_jobs)
        engine8.device_free(d_res)
    assert all(m > 0 for m in marked.values()), marked          # the wrap-around pair hands PUs of every class on
    total = 0
    for i, (a, b) in enumerate(frames):
        jobs = host_jobs[i]
        long_ = np.flatnonzero(res[i]["sad_calls"] > {"star": 22, "hex": 16, "dia": 6}[mname])      # went on past what the fixed-shape kernels measure
        rng = np.random.default_rng(5 + i)
        pick = [rng.choice(np.flatnonzero(jobs["w"] == s), 20, replace=False) for s in (64, 32, 16, 8)]
        for s in (64, 32, 16, 8):
            of = long_[jobs["w"][long_] == s]
            if len(of):
                pick.append(rng.choice(of, min(12, len(of)), replace=False))
        idx = np.unique(np.concatenate(pick))
        total += len(idx)
        org = M * (W + 2 * M) + M
        mv, cost, sc, qc = oracle_search(oracle, 8, up[a][1], [up[b][1]], org, W + 2 * M, jobs[idx], QP, method, 5)
        assert np.array_equal(res[i]["mv"][idx], mv) and np.array_equal(res[i]["cost"][idx], cost), i
        assert np.array_equal(res[i]["sad_calls"][idx], sc) and np.array_equal(res[i]["subpel_calls"][idx], qc), i
    assert total >= 200
    engine8.device_free(d_pairs)
    for f in up:
        for p in [up[f][0]] + low[f]:
            engine8.free_plane(p)
