"""Record what the REFERENCE ENCODER itself hands to its motion estimation during real encodes, and what it gets back.

Runs only in the build container (needs /root/reference and oracle/_ref/x265_trace, the reference's own encoder and CLI
compiled by oracle/Makefile with its ME entry points routed through oracle/enc_trace.cpp). Two encodes of the
416x240 synthetic clip whose bitstream MD5s are on record:

  c0   30 frames, --preset medium --me hex --threads 1                 (rd 6: the Rockchip CIME / RIME / FME overload)
  lb   8 frames, the same + --bframes 3 --rc-lookahead 10 --b-adapt 2  (lookahead with B-frames)

The bitstream MD5 of each run must equal the recorded one, so the calls in the fixture are the calls of exactly
those encodes. Written to tests/golden/callstream.npz:

  rk_*   for a few P-frames of c0: the reconstructed reference picture the searches read, and per 64x64 CTU the
         arguments of the RK overload (TEncSearch.cpp:3821: range, neighbour PMVs, cost predictors, valid size, QP) with
         the CIME / RIME / FME results of every CU it was called for (outQMv / cost, g_leftPMV, g_RimeMv, g_FmeMv)
  la_*   every lowres MV field either run's lookahead searched (slicetype.cpp:1468): per 8x8 CU the predictor and
         candidates it was given and the MV / cost it returned

    python tests/golden/make_callstream.py
"""
import collections
import hashlib
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from x265_amd.synth import baseline_clip, write_yuv420  # noqa: E402
from x265_amd.engine import RK_CTU_DTYPE  # noqa: E402

W, H, SEED = 416, 240, 1234
RUNS = {
    # name: (frames, extra options, recorded bitstream MD5, POCs whose RK calls are kept)
    "c0": (30, [], "f2fe9f37c8c7e18d4f4610df4a25c48f", (1, 2, 17)),
    "lb": (8, ["--bframes", "3", "--rc-lookahead", "10", "--b-adapt", "2"], "435ffbe83cfe979fec8e147e5fb9b070", ()),
}
CLIP_MD5 = {30: "47910afbe49bc72bcd09ad47f04a4dba"}
TRACER = os.path.join(ROOT, "oracle", "_ref", "x265_trace")


def parse_trace(path):
    """Records of oracle/enc_trace.cpp, in file order."""
    d = open(path, "rb").read()
    o = 0
    recs = []

    def i32(n=1):
        nonlocal o
        v = struct.unpack_from("<%di" % n, d, o)
        o += 4 * n
        return list(v) if n > 1 else v[0]

    def u64():
        nonlocal o
        v = struct.unpack_from("<Q", d, o)[0]
        o += 8
        return v

    def mv(n=1):
        nonlocal o
        v = np.frombuffer(d, "<i2", 2 * n, o).reshape(n, 2).copy()
        o += 4 * n
        return v if n > 1 else v[0]

    while o < len(d):
        t = d[o:o + 4].decode()
        o += 4
        if t == "FRAM":
            poc, stype, w, h = i32(4)
            recs.append(dict(t=t, poc=poc, slice_type=stype, w=w, h=h))
        elif t == "MREF":
            cur, rid, w, h, mx, my, wp = i32(7)
            orig = np.frombuffer(d, np.uint8, w * h, o).reshape(h, w).copy()
            o += w * h
            recon = np.frombuffer(d, np.uint8, w * h, o).reshape(h, w).copy()
            o += w * h
            recs.append(dict(t=t, cur=cur, id=rid, margin=(mx, my), wp=wp, orig=orig, recon=recon))
        elif t == "RKME":
            cur, rid, x, y, bw, bh, qp = i32(7)
            mn, mx, nb = mv(), mv(), mv(10)
            ncand, have_mvd, save, offs, depth, boff_ds, stride_ds, vw, vh, refidx = i32(10)
            out, cost = mv(), i32()
            g = i32(9)
            recs.append(dict(t=t, cur=cur, ref=rid, x=x, y=y, bw=bw, bh=bh, qp=qp, mvmin=mn, mvmax=mx, neigh=nb, ncand=ncand,
                             have_mvd=have_mvd, save=save, offs=offs, depth=depth, boff_ds=boff_ds, stride_ds=stride_ds,
                             vw=vw, vh=vh, refidx=refidx, out=out, cost=cost, cime=g[0:3], rime=g[3:6], fme=g[6:9]))
        elif t == "LOWR":
            poc, p = i32(), u64()
            w, lines, stride, mx, my = i32(5)
            recs.append(dict(t=t, poc=poc, plane=p, w=w, lines=lines, stride=stride, margin=(mx, my)))
        elif t == "CLME":
            f, r = u64(), u64()
            lowres, x, y, bw, bh, method, subme, qp = i32(8)
            mn, mx, q = mv(), mv(), mv()
            numc, mvc, merange = i32(), mv(4), i32()
            out, cost = mv(), i32()
            recs.append(dict(t=t, fenc=f, ref=r, lowres=lowres, x=x, y=y, bw=bw, bh=bh, method=method, subme=subme, qp=qp,
                             mvmin=mn, mvmax=mx, qmvp=q, numc=numc, mvc=mvc, merange=merange, out=out, cost=cost))
        else:
            raise SystemExit("bad tag %r at %d" % (t, o))
    return recs


def run_encoder(name, clip, tmp):
    frames, extra, md5, pocs = RUNS[name]
    yuv = os.path.join(tmp, "%s.yuv" % name)
    write_yuv420(yuv, clip[:frames])
    if frames in CLIP_MD5:
        assert hashlib.md5(open(yuv, "rb").read()).hexdigest() == CLIP_MD5[frames], "clip generator drifted"
    trace, hevc = os.path.join(tmp, name + ".trace"), os.path.join(tmp, name + ".hevc")
    env = dict(os.environ, X265AMD_TRACE=trace, X265AMD_TRACE_RK_POCS=",".join(str(p) for p in pocs) if pocs else "-1")
    cmd = [TRACER, "--input", yuv, "--input-res", "%dx%d" % (W, H), "--fps", "30", "-f", str(frames), "--preset", "medium",
           "--me", "hex", "--threads", "1", "--no-progress"] + extra + ["-o", hevc]
    subprocess.run(cmd, env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    got = hashlib.md5(open(hevc, "rb").read()).hexdigest()
    assert got == md5, "%s: bitstream MD5 %s differs from the recorded %s" % (name, got, md5)
    return parse_trace(trace)


def rk_fixture(recs, clip, out):
    """Per kept P-frame: reference planes, one x265amd_rk_ctu per CTU from the call that set the CTU's PMVs
    (isSavePmv), and the three stage results of every CU the encoder asked for."""
    pocs = RUNS["c0"][3]
    refs = {r["id"]: r for r in recs if r["t"] == "MREF"}
    ncx, ncy = (W + 63) // 64, (H + 63) // 64
    out["rk_pocs"] = np.array(pocs, np.int32)
    for poc in pocs:
        calls = [r for r in recs if r["t"] == "RKME" and r["cur"] == poc]
        rid = calls[0]["ref"]
        assert all(c["ref"] == rid and c["refidx"] == 0 for c in calls), "one reference per P-frame expected"
        ref = refs[rid]
        ref_poc = [i for i, f in enumerate(clip) if np.array_equal(f, ref["orig"])]
        assert len(ref_poc) == 1 and ref["wp"] == 0
        ctus = np.zeros(ncx * ncy, RK_CTU_DTYPE)
        cime = np.zeros((ncx * ncy, 3), np.int32)
        rime = np.full((ncx * ncy, 85, 3), -1, np.int32)
        fme = np.full((ncx * ncy, 85, 3), -1, np.int32)
        rime[:, :, :2] = 0
        fme[:, :, :2] = 0
        ranges = np.zeros((ncx * ncy, 85, 4), np.int16)      # mvmin / mvmax handed in with every CU call
        qps = set()
        seen = set()
        for c in calls:
            a = (c["y"] // 64) * ncx + c["x"] // 64
            if c["save"]:
                assert a not in seen
                seen.add(a)
                j = ctus[a]
                j["ctu_x"], j["ctu_y"] = (c["x"] // 64) * 64, (c["y"] // 64) * 64
                j["valid_w"], j["valid_h"] = c["vw"], c["vh"]
                j["mvmin"], j["mvmax"] = c["mvmin"], c["mvmax"]
                # motion.cpp:1305-1312: RIME window centres 1 and 2 = the de-duplicated neighbours, the second
                # repeating the first when there is only one
                assert 1 <= c["ncand"] <= 2
                j["pmv"][0] = c["neigh"][0]
                j["pmv"][1] = c["neigh"][1] if c["ncand"] > 1 else c["neigh"][0]
                j["mvp"] = c["neigh"][7:10]
                j["have_mvd"] = c["have_mvd"]
                cime[a] = c["cime"]
                out["rk_%d_ncand_%d" % (poc, a)] = np.int32(c["ncand"])
            else:
                assert a in seen, "a CU was searched before its CTU's PMVs were set"
                # the statics the overload keeps per CTU (TEncSearch.cpp:3698, motion.cpp:1269): later CUs see the same
                # neighbour array and the CIME result of the CTU's first call
                assert np.array_equal(c["neigh"][7:10], ctus[a]["mvp"]) and list(c["cime"]) == list(cime[a])
            k = c["offs"]
            assert rime[a, k, 2] == -1, "CU searched twice"
            rime[a, k] = c["rime"]
            fme[a, k] = c["fme"]
            ranges[a, k] = list(c["mvmin"]) + list(c["mvmax"])
            assert list(c["out"]) == c["fme"][:2] and c["cost"] == c["fme"][2]
            qps.add(c["qp"])
        assert len(seen) == ncx * ncy and len(qps) == 1
        key = "rk_%d_" % poc
        out[key + "ref_poc"] = np.int32(ref_poc[0])
        out[key + "recon"] = ref["recon"]
        out[key + "margin"] = np.array(ref["margin"], np.int32)
        out[key + "qp"] = np.int32(qps.pop())
        out[key + "ctus"] = ctus.view(np.uint8).reshape(len(ctus), -1)
        out[key + "cime"] = cime
        out[key + "rime"] = rime
        out[key + "fme"] = fme
        out[key + "cu_ranges"] = ranges


def lookahead_fixture(name, recs, out):
    """Every lowres field the lookahead searched: (frame searched, reference frame) -> per-CU inputs and results."""
    poc_of = {}
    geom = None
    fields = collections.OrderedDict()
    for r in recs:
        if r["t"] == "LOWR":
            poc_of[r["plane"]] = r["poc"]
            geom = (r["w"], r["lines"], r["stride"]) + tuple(r["margin"])
        elif r["t"] == "CLME":
            assert r["lowres"] == 1 and (r["bw"], r["bh"]) == (8, 8)
            key = (poc_of[r["fenc"]], poc_of[r["ref"]])
            fields.setdefault(key, []).append(r)
    wcu, hcu = geom[0] // 8, geom[1] // 8
    ncu = wcu * hcu
    keys, mvs, costs, qmvp, numc, mvc, rng = [], [], [], [], [], [], []
    consts = set()
    for key, calls in fields.items():
        # a pair can be searched again after Lowres::init re-armed the frame; every pass must agree
        assert len(calls) % ncu == 0, (key, len(calls))
        f_mv = np.zeros((ncu, 2), np.int16); f_cost = np.zeros(ncu, np.int32)
        f_q = np.zeros((ncu, 2), np.int16); f_n = np.zeros(ncu, np.uint8); f_c = np.zeros((ncu, 4, 2), np.int16)
        f_r = np.zeros((ncu, 4), np.int16)
        for p in range(len(calls) // ncu):
            order = []
            for c in calls[p * ncu:(p + 1) * ncu]:
                i = (c["y"] // 8) * wcu + c["x"] // 8
                order.append(i)
                vals = (c["out"], c["cost"], c["qmvp"], c["numc"], c["mvc"], list(c["mvmin"]) + list(c["mvmax"]))
                if p:
                    assert np.array_equal(f_mv[i], vals[0]) and f_cost[i] == vals[1] and np.array_equal(f_q[i], vals[2])
                f_mv[i], f_cost[i], f_q[i], f_n[i], f_c[i], f_r[i] = vals
                consts.add((c["method"], c["subme"], c["qp"], c["merange"]))
            # rows run bottom-up, CUs right to left, a row following the one below by two CUs (slicetype.cpp:1342-1387);
            # two threads may interleave rows, so what is checked is that every CU came after the neighbours it reads
            pos = {cu: n for n, cu in enumerate(order)}
            assert sorted(order) == list(range(ncu))
            for cu in range(ncu):
                x, y = cu % wcu, cu // wcu
                for dx, dy in ((1, 0), (0, 1), (-1, 1), (1, 1)):
                    if 0 <= x + dx < wcu and y + dy < hcu:
                        assert pos[cu + dx + dy * wcu] < pos[cu], "a CU was searched before a neighbour it takes its predictor from"
        keys.append(key); mvs.append(f_mv); costs.append(f_cost); qmvp.append(f_q); numc.append(f_n); mvc.append(f_c); rng.append(f_r)
    assert len(consts) == 1
    out["la_%s_fields" % name] = np.array(keys, np.int32)
    out["la_%s_mvs" % name] = np.array(mvs)
    out["la_%s_costs" % name] = np.array(costs)
    out["la_%s_qmvp" % name] = np.array(qmvp)
    out["la_%s_numc" % name] = np.array(numc)
    out["la_%s_mvc" % name] = np.array(mvc)
    out["la_%s_ranges" % name] = np.array(rng)
    out["la_%s_params" % name] = np.array(list(consts.pop()) + [wcu, hcu] + list(geom), np.int32)   # method, subme, qp, merange, wcu, hcu, w, lines, stride, margins


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "encoder"], stdout=subprocess.DEVNULL)
    clip = baseline_clip(W, H, 30, SEED)
    out = {"clip_sha256": np.frombuffer(hashlib.sha256(np.stack(clip).tobytes()).digest(), np.uint8)}
    with tempfile.TemporaryDirectory() as tmp:
        for name in RUNS:
            recs = run_encoder(name, clip, tmp)
            out["%s_bitstream_md5" % name] = np.frombuffer(bytes.fromhex(RUNS[name][2]), np.uint8)
            out["%s_slice_types" % name] = np.array([(r["poc"], r["slice_type"]) for r in recs if r["t"] == "FRAM"], np.int32)
            if name == "c0":
                rk_fixture(recs, clip, out)
            lookahead_fixture(name, recs, out)
    np.savez_compressed(os.path.join(HERE, "callstream.npz"), **out)
    print("callstream.npz:", {k: getattr(v, "shape", ()) for k, v in out.items() if not k.startswith("rk_") or k.endswith("ctus")})


if __name__ == "__main__":
    main()
