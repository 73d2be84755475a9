#!/usr/bin/env python3
"""Motion-estimation throughput on MI355X: BASELINE.json configs[1].

Workload (one step = one batch of synthetic 1080p 8-bit frame pairs per GPU):
  * full-encode ME: every square PU of the 64/32/16/8 quadtree depths of each frame
    (42 900 PUs per frame), STAR search (--me star), merange 57, subme 5, predictors
    taken from the lookahead field of the same frame pair;
  * lookahead ME: the lowres MV field of each pair (8160 8x8 CUs, HEX, subme 1,
    merange 16) on its own stream.
Inputs (planes, job lists) are resident in HBM before the timed region. Results are
checked against the CPU oracle before anything is reported.

metric value = pixel-SADs evaluated per second over all ranks: (full-pel + sub-pel
candidates the search visited) x PU area, counted by the kernel itself per PU and equal
to the oracle's count for the same jobs.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The step runs four launch chains side by side, and every search context forks a second stream for the few PUs whose
# searches run long: give the HIP runtime a hardware queue per stream (its default of 4 makes streams share queues, and
# chains that share a queue run one after the other). Read when the runtime starts, so set before torch is imported.
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("X265AMD_BENCH_HW_QUEUES", "8")

W, H, MARGIN = 1920, 1080, 128
QP, MERANGE, SUBME_DEFAULT = 32, 57, 5
LA_QP, LA_MERANGE = 12, 16
HBM_PEAK_GBS = 8000.0
SIMDS = 1024            # 256 CUs x 4 SIMDs


_T0 = time.perf_counter()


def progress(msg, rank=0):
    """One line on stderr per phase (stdout carries the JSON line only): a long run stays visibly alive."""
    if rank == 0:
        print("[bench %6.1f s] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def host_threads(args):
    """Threads of the many-core CPU leg: the box's CPU share, not the machine's core count."""
    if args.cpu_threads > 0:
        return args.cpu_threads
    n = len(os.sched_getaffinity(0))
    env = os.environ.get("OMP_NUM_THREADS", "")
    return max(1, min(n, int(env))) if env.isdigit() else n


def measured_encode(frames=30):
    """BASELINE configs[0] encoded for real: the reference's own encoder + CLI (oracle/_ref/x265_served, compiled in the
    build container from where the reference's sources lie) run twice on this box -- once with its motion estimation
    answered by the engine through the C ABI (oracle/enc_serve.cpp: one x265amd_rk_ctu_batch per CTU, one
    x265amd_lookahead_fields per lowres field, reference planes and lowres planes made on the device), once on the CPU
    alone -- bitstream MD5 and the fps of the encoder's own summary line (x265.cpp:761) for both."""
    import hashlib
    import re
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "x265_served")
    if not os.path.exists(exe):
        return {"value": None, "measured": False, "why": "oracle/_ref/x265_served was not built (needs the reference's sources)"}
    from x265_amd.synth import baseline_clip, write_yuv420
    recorded = "f2fe9f37c8c7e18d4f4610df4a25c48f"
    out = {"unit": "fps", "measured": True,
           "config": "416x240 synthetic 8-bit 4:2:0, %d frames, --preset medium --me hex --threads 1 (rd 6: the Rockchip "
                     "CIME + RIME + FME overload per CU, lookahead HEX field per P-frame)" % frames,
           "recorded_md5": recorded}
    with tempfile.TemporaryDirectory() as tmp:
        yuv = os.path.join(tmp, "c0.yuv")
        write_yuv420(yuv, baseline_clip(416, 240, frames, 1234))
        for mode in ("gpu", "cpu"):
            progress("measured encode: %d frames, motion estimation on the %s" % (frames, "engine" if mode == "gpu" else "host"))
            hevc, stats = os.path.join(tmp, mode + ".hevc"), os.path.join(tmp, mode + ".json")
            env = dict(os.environ, X265AMD_SERVE_STATS=stats)
            env.pop("X265AMD_SERVE", None)
            if mode == "gpu":
                env["X265AMD_SERVE"] = "gpu"
            t0 = time.perf_counter()
            p = subprocess.run([exe, "--input", yuv, "--input-res", "416x240", "--fps", "30", "-f", str(frames), "--preset", "medium",
                                "--me", "hex", "--threads", "1", "--no-progress", "-o", hevc],
                               env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
            wall = time.perf_counter() - t0
            m = re.search(r"encoded (\d+) frames in ([0-9.]+)s \(([0-9.]+) fps\)", p.stdout.decode(errors="replace"))
            if p.returncode != 0 or not m:
                return {"value": None, "measured": False, "why": "%s run failed (rc %d)" % (mode, p.returncode)}
            st = json.load(open(stats))
            md5 = hashlib.md5(open(hevc, "rb").read()).hexdigest()
            out[mode] = {"fps": float(m.group(3)), "encoder_seconds": float(m.group(2)), "process_wall_seconds": wall,
                         "bitstream_md5": md5, "md5_is_the_recorded_one": md5 == recorded,
                         "seconds_inside_motionEstimate": st["rk_seconds"] + st["lookahead_seconds"],
                         "seconds_uploading_planes": st["upload_seconds"],
                         "motionEstimate_calls": st["rk_calls"] + st["lookahead_calls"], "calls_not_served": st["fallback_calls"],
                         "engine_launches": {"rk_ctu": st["rk_ctu_launches"], "lookahead_field": st["lookahead_field_launches"]}}
    out["value"] = out["gpu"]["fps"]
    out["cpu_only_fps"] = out["cpu"]["fps"]
    out["note"] = ("the encoder's mode decision needs a CTU's MVs before it derives the next CTU's predictors, so the engine is "
                   "called once per CTU (85 CUs x 3 stages per launch) and synchronously: this measures dispatch latency, not "
                   "throughput. What stays on the host is everything else of the encoder, including the hardware C-model's own "
                   "re-computation of CIME / RIME / FME for its self-check (hardwareC.cpp:67-155), which dominates both runs")
    return out



def headline_roofline(vb, pus, launch_ms, alg_gbs, alg_bytes, traffic, compulsory, kernel):
    """The roofline entry of the timed launch class. The bound that holds is VALU issue: `achieved` counts the ESSENTIAL
    wave64 VALU instructions of the class's searches (profiles/valu.json says how they are counted) per second of the
    launch, `peak` is what the chip's 1024 SIMDs issue at the measured shader clock (one wave64 VALU instruction per 4
    cycles), so frac <= busy_frac <= 1 by construction; busy_frac is the same with the instructions the kernels really
    issued (SQ counters). HBM is far from binding: `hbm` is the measured traffic of the launch (FETCH_SIZE doubled +
    WRITE_SIZE, separate --pmc passes) over its duration; `hbm_per_job` keeps the per-job algorithmic byte formula, which
    prices every PU's window as if no two PUs shared a byte and is therefore not traffic."""
    launch_s = launch_ms * 1e-3
    hbm = {"achieved": traffic / launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": traffic / launch_s / 1e9 / HBM_PEAK_GBS, "bytes_per_launch": traffic} if traffic else None
    per_job = {"algorithmic_gbs": alg_gbs, "algorithmic_bytes_per_launch": alg_bytes,
               "formula": "per PU: source block + search window incl. 8-tap apron + result record, no reuse between PUs assumed; "
                          "neighbouring PUs share their windows through L2, so this is not HBM traffic and may exceed the HBM peak"}
    if vb is None:
        # no instruction profile for this method / level: only the measured traffic can be priced
        return {"bound": "hbm", "achieved": hbm["achieved"] if hbm else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": hbm["frac"] if hbm else None, "traffic": traffic, "kernel": kernel, "launch_ms": launch_ms,
                "hbm_per_job": per_job, "compulsory": compulsory,
                "note": "no VALU instruction profile for this method: HBM traffic of the star / subme-5 launch is reported as is"}
    hz = vb["shader_clock_mhz"] * 1e6
    peak = SIMDS * hz / 4
    ess = vb["essential_wave_instr_per_pu"]["value"] * pus / launch_s
    return {"bound": "valu", "achieved": ess / 1e9, "peak": peak / 1e9,
            "unit": "G wave-instr/s (wave64 VALU instructions; a SIMD issues one per 4 cycles, %d SIMDs at the measured shader clock)" % SIMDS,
            "frac": ess / peak, "busy_frac": vb["busy_frac"], "step_busy_frac": vb["step_busy_frac"],
            "traffic": traffic, "kernel": kernel, "launch_ms": launch_ms,
            "valu": vb, "hbm": hbm, "hbm_per_job": per_job, "compulsory": compulsory,
            "note": "achieved = essential VALU wave instructions per PU x PUs of the launch / launch duration (HIP events on the "
                    "launch's stream while the other classes run beside it); busy_frac = the instructions actually issued, same "
                    "denominator; step_busy_frac = all kernels of a step over the step. The launch reads %s of HBM peak. The "
                    "instruction counts come from a counter run of a 2-pair step and are scaled by PUs: that assumes a PU's "
                    "instructions do not depend on how many pairs a step holds (same kernels, same lists per pair; the share of "
                    "handed-on PUs differs from pair to pair by a few per cent of the PUs)"
                    % ("%.1f %%" % (100 * hbm["frac"]) if hbm else "an unmeasured share")}



def valu_bound(profile, kernels, pus, launch_s, step_pairs=0, step_s=0.0, clock_mhz=None):
    """VALU-issue accounting of one launch class from the SQ counters kept in profiles/valu.json (per frame pair):
    wave instructions issued per PU and the fraction of the launch during which the chip's 1024 SIMDs would be busy
    issuing them back to back (a wave64 VALU instruction occupies its SIMD for 4 cycles). The launch shares the chip
    with the other classes' launches, so the figure that says how close the step is to the VALU issue bound is
    step_busy_frac: the same for all kernels of a step over the step's duration."""
    if not profile:
        return None
    k = [profile["kernels_per_pair"][n] for n in kernels if n in profile["kernels_per_pair"]]
    if not k:
        return None
    valu, salu, lds = (sum(x[c] for x in k) for c in ("valu", "salu", "lds"))
    per_pair_pus = profile["pus_per_pair"]
    npairs = pus / per_pair_pus
    all_valu = profile.get("valu_wave_instr_per_pair_all_kernels")
    hz = clock_mhz * 1e6 if clock_mhz else profile["clock_hz"]
    return {"issued_wave_instr_per_pu": {"valu": valu / per_pair_pus, "salu": salu / per_pair_pus, "lds": lds / per_pair_pus},
            "shader_clock_mhz": hz / 1e6,
            "shader_clock_how": "measured during three extra steps by a probe wave (x265amd_clock_probe_*)" if clock_mhz
                                else "nominal (no probe)",
            "busy_frac": valu * npairs * 4 / (profile["simds"] * hz * launch_s),
            "step_busy_frac": all_valu * step_pairs * 4 / (profile["simds"] * hz * step_s) if all_valu and step_s else None,
            "step_valu_wave_instr_per_frame_pair": all_valu,
            "essential_wave_instr_per_pu": profile["essential_wave_instr_per_pu"],
            "source": profile["source"]}


def split_roofline(workload_name, value_gpx, alg_gbs, alg_bytes, world, measured_gpx=None):
    """Roofline entry of the split workloads (whole step: all launches + the gather, not one kernel). The exhaustive
    search is bound by the issue rate of the byte-SAD instructions; the lookahead window by the dependency chain of its
    wavefront (its HBM and VALU shares are both small), for which the HBM entry prices the compulsory bytes only."""
    from x265_amd.measure import SAD_ISSUE_PEAK
    per_job = {"algorithmic_gbs": alg_gbs, "algorithmic_bytes_per_step": alg_bytes,
               "formula": "compulsory bytes with no reuse between units; not measured traffic"}
    if workload_name == "esa4k":
        peak = SAD_ISSUE_PEAK * world / 1e9
        measured = value_gpx if measured_gpx is None else measured_gpx
        return {"bound": "valu", "achieved": measured, "peak": peak, "unit": "Gpixel-SAD/s (v_sad_u8 / v_qsad_pk_u16_u8 issue rate)",
                "frac": measured / peak, "credited": value_gpx, "traffic": None, "hbm_per_job": per_job,
                "note": "whole step (all launches + gather); achieved = the SADs the launches measure (the scan per CTU measures the "
                        "8x8 level once over the box around a CTU's ranges; `credited` = `value` counts every PU's own range at "
                        "its own size); 10^3 - 10^4 pixel-SADs per byte of window: HBM does not bind"}
    return {"bound": "hbm", "achieved": alg_gbs, "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": alg_gbs / (HBM_PEAK_GBS * world),
            "traffic": None, "hbm_per_job": per_job,
            "note": "whole step; the launch is bound by the dependency chain of the lookahead wavefront (width + 2 x height CU "
                    "searches per field), neither by HBM nor by VALU issue: achieved prices every plane of a field once"}



def run_split_workload(args, torch, dist, dev, cdev, rank, local_rank, world):
    """--workload esa4k / la8k: one picture's (reference, CTU-row band) units, or one lookahead window's fields, split
    over the ranks; every step ends with the gather of the MV records to rank 0. Strong scaling: the total work is
    fixed, `value` = the pixel-SADs of the whole picture / window per second."""
    from x265_amd.engine import Engine
    from x265_amd import sharded, sharding, workload
    from x265_amd.measure import SAD_ISSUE_PEAK
    e = Engine(local_rank, 8)
    if args.workload == "esa4k":
        wl = sharded.EsaPicture(e, rank, world)
        wl.validate()
        name = ("3840x2160 8-bit, --me full --merange 92 --subme 5, every square 64/32/16/8 PU of one picture against 3 references; "
                "%d (reference, CTU row) units split contiguously over %d GPU(s), each rank holds only its bands' rows + halo; "
                "one launch per rank: the scan per CTU over all of its bands (x265amd_esa_ctu_batch_pairs_dev)"
                % (wl.nrefs * wl.nrows, world))
    else:
        wl = sharded.LookaheadWindow(e, rank, world)
        name = ("7680x4320 8-bit lookahead window of 10 frames (bframes 8): the %d lowres MV fields (HEX, subme 1, merange 16, "
                "%d CUs each) of every (frame, reference) at most 9 apart, dealt out in turn over %d GPU(s)"
                % (len(wl.all_fields), wl.ncu, world))

    # the rank's records in a tensor of its own, so that RCCL gathers straight from where the kernels wrote them
    rec = torch.zeros(wl.records_bytes, dtype=torch.uint8, device=dev)
    wl.records_ptr = rec.data_ptr()

    def step():
        wl.step()
        if world > 1:
            sharding.gather_records(rec.to(cdev), dist, dst=0)

    progress("set-up done: %s; warm-up" % args.workload, rank)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    progress("timed steps", rank)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    measured = float(wl.measured_pixel_sads()) if hasattr(wl, "measured_pixel_sads") else 0.0
    tot = torch.tensor([float(wl.pixel_sads()), float(wl.algorithmic_bytes()), measured], dtype=torch.float64, device=cdev)
    t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    px, alg = float(tot[0].item()), float(tot[1].item())
    # parity: the gathered records against the unsharded run of the same units on rank 0's GPU (N > 1), and a sample
    # against the CPU oracle (rank 0)
    progress("%d timed steps: %.3f ms per step; parity against the oracle" % (args.steps, elapsed / args.steps * 1e3), rank)
    gathered = sharding.gather_records(rec.to(cdev), dist, dst=0) if world > 1 else [rec]
    if rank == 0:
        from oracle.bindings import Oracle   # checker / baseline only
        from x265_amd.synth import pad_plane
        orc = Oracle()
        rng = np.random.default_rng(7)
        cpu_s, cpu_px = 0.0, 0
        if args.workload == "esa4k":
            mv, cost = wl.assemble([g.cpu().numpy() for g in gathered])
            assert all((c >= 0).all() for c in cost), "a job of the picture was searched by no rank"
            cur, org, stride = pad_plane(wl.cur, 128)
            for r in range(wl.nrefs if args.cpu_pairs else 0):
                ref, _, _ = pad_plane(wl.refs[r], 128)
                ctx = orc.make_ctx(8, cur, org, stride, [ref], org, stride, False, wl.qp, 4, wl.subme)
                jobs = wl.jobs[r]
                pick = np.concatenate([rng.choice(np.flatnonzero(jobs["w"] == s), n, replace=False)
                                       for s, n in ((64, 64), (32, 192), (16, 768), (8, 3072))])
                c0 = time.perf_counter()
                exp = orc.me_search_batch(ctx, jobs[pick])
                cpu_s += time.perf_counter() - c0
                if not (np.array_equal(exp["mv"], mv[r][pick]) and np.array_equal(exp["cost"], cost[r][pick])):
                    print("PARITY FAILURE: gathered records of reference %d differ from the oracle" % r, file=sys.stderr)
                    sys.exit(2)
                cpu_px += workload.pixel_sads(jobs[pick], exp)
            sample = ("12288 PUs of the picture (64 64x64, 192 32x32, 768 16x16, 3072 8x8 per reference, seeded) through "
                      "oracle/me_oracle.c -O3, one thread")
        else:
            fields = wl.assemble([g.cpu().numpy() for g in gathered])
            from oracle.bindings import MEStats
            from x265_amd.synth import lowres_dims

            def host_lowres_planes(img):
                """Four padded lowres planes made with the oracle's downscaler (lowres.cpp:155-166): (planes, origin, stride)."""
                lw, lh, _, _ = lowres_dims(img.shape[1], img.shape[0])
                src, so, ss = pad_plane(img, 24)
                padded = [pad_plane(o, 64, stride_align=32) for o in orc.lowres_core(8, src, so, ss, lw, lh)]
                return [p[0] for p in padded], padded[0][1], padded[0][2]
            low = {}
            for u in (range(len(wl.all_fields)) if args.cpu_pairs else ()):
                b, r = wl.all_fields[int(u)]
                for f in (b, r):
                    if f not in low:
                        low[f] = host_lowres_planes(wl.clip[f])
                pa, pb = low[b], low[r]
                ctx = orc.make_ctx(8, pa[0][0], pa[1], pa[2], pb[0], pa[1], pa[2], True, wl.qp, 1, 1)
                st = MEStats()
                c0 = time.perf_counter()
                emv, _ = orc.lookahead_field(ctx, wl.wcu, wl.hcu, 16, stats=st)
                cpu_s += time.perf_counter() - c0
                if not np.array_equal(emv, fields[int(u)]):
                    print("PARITY FAILURE: gathered field %d differs from the oracle" % int(u), file=sys.stderr)
                    sys.exit(2)
                cpu_px += (st.sad_calls + st.subpel_calls) * 64
            sample = "all fields of the window through oracle/me_oracle.c -O3, one thread (lowres planes made beforehand)"
        value = px * args.steps / elapsed / 1e9
        out = {"metric": "ME Gpixel-SAD/s", "value": value, "unit": "Gpixel-SAD/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": name, "parallelism": "units split over %d GPU(s), gather of MV records to rank 0" % world},
               "roofline": split_roofline(args.workload, value, alg * args.steps / elapsed / 1e9, alg, world,
                                          float(tot[2].item()) * args.steps / elapsed / 1e9 if float(tot[2].item()) else None),
               "cpu_baseline": {"value": cpu_px / cpu_s / 1e9, "unit": "Gpixel-SAD/s", "cores": 1, "kind": "port",
                                "sample": sample + ", %.1f s; compared with the gathered GPU records" % cpu_s} if cpu_s else None}
        print(json.dumps(out))
    wl.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=32, help="frame pairs per GPU per step")
    ap.add_argument("--cpu-pairs", type=int, default=24,
                    help="frame pairs whose PU searches are re-run on one host core (parity gate + CPU baseline; 0 = skip)")
    ap.add_argument("--ref-pairs", type=int, default=8,
                    help="frame pairs re-run through the reference build (oracle/_ref) when it is present: the CPU baseline")
    ap.add_argument("--workload", choices=("me1080p", "esa4k", "la8k"), default="me1080p",
                    help="me1080p = the headline (BASELINE configs[1]; weak scaling: 32 frame pairs per GPU); esa4k = one 4K picture "
                         "searched exhaustively against 3 references, (reference, CTU-row band) units split over the GPUs "
                         "(configs[2], strong scaling); la8k = the lowres MV fields of one 8K lookahead window dealt out over the "
                         "GPUs (configs[4], strong scaling)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads of the many-core CPU baseline leg (0 = the box's CPU share: OMP_NUM_THREADS, else the "
                         "affinity mask)")
    ap.add_argument("--method", default="star")
    ap.add_argument("--subme", type=int, default=SUBME_DEFAULT, help="sub-pel refinement level (experiments only)")
    ap.add_argument("--launch", choices=("families", "pairs4", "pairs3", "split", "split3", "single", "per-size"), default="families",
                    help="families = pairs3 with the list of the 64x64 + 32x32 launch in quadtree order: every 64x64 PU followed "
                         "by its four 32x32 children, which the library then refines together with it where they start from "
                         "the same MV (32x32 PUs without a parent at the end of a pair's part); search launches: pairs4 = per step ONE launch per PU size (8x8 | 16x16 | 32x32, 64x64) over all frame "
                         "pairs (x265amd_search_batch_pairs_dev), three streams, the two large sizes sharing one; pairs3 = "
                         "the 64x64 and 32x32 PUs in one launch; the others launch per frame pair: split3 = the "
                         "same three classes, split = 16/8 and 64/32 on two streams (LDS per PU sized per class), single = one "
                         "launch, per-size = four launches on one stream")
    ap.add_argument("--prio", choices=("small", "big", "la", "flat"), default="flat",
                    help="which search stream gets the device's highest priority: small = the 16x16 + 8x8 class; big = the 64x64 + "
                         "32x32 class; la = the lookahead's stream; flat = none (measured on the family launches: flat 3.55 ms, "
                         "small 3.82, big 3.79 -- a favoured stream holds the others' dependent kernels back)")
    ap.add_argument("--s32", choices=("big", "mid", "la", "own"), default="la",
                    help="pairs4: the stream of the 32x32 launch: behind the 64x64 launch, behind the 16x16 launch, behind the "
                         "lookahead field, or one of its own")
    ap.add_argument("--sync-steps", choices=("yes", "no"), default="yes",
                    help="wait for every stream at the end of each step (yes), or let the launches of consecutive steps queue "
                         "behind each other and wait once at the end of the timed region (no: one GPU, pairs3 / pairs4 only; "
                         "measured 3 % slower: the next step's first kernels take room from this step's last ones)")
    ap.add_argument("--big-streams", type=int, default=1,
                    help="streams the launches of the 64/32 PU class rotate over (split / split3): a 64x64 PU is one long wave, "
                         "so a launch of one frame's 480 + 1980 large PUs is bounded by that wave's latency, and launches of "
                         "different frame pairs overlap when they sit on different streams")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="rehearsal of the N > 1 control flow on a box with one GPU: every rank uses cuda:0 and the "
                         "collectives go over gloo through host memory (RCCL wants one GPU per rank); not a measurement")
    ap.add_argument("--only", default="",
                    help="experiments only (the result is then no measurement of the workload): comma-separated launch classes "
                         "of the multi-pair step to run, of 8, 16, 64 (32 with pairs4) and la; the others are left out")
    ap.add_argument("--no-clock-probe", action="store_true",
                    help="skip the three extra (untimed) steps during which the shader clock is measured: profiler passes, "
                         "whose per-kernel sums should hold the timed steps only")
    ap.add_argument("--no-other-paths", action="store_true",
                    help="skip the exhaustive-search (4K) and Rockchip-path (1080p) measurements reported beside the headline")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from x265_amd.engine import Engine, METHODS, RESULT_DTYPE, HEX, SQUARE_ONLY
    from x265_amd.synth import make_clip, pad_plane, lowres_dims
    from x265_amd import workload, sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.rehearse_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if args.rehearse_one_gpu else dev      # where collectives take their tensors
    if world > 1:
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    method = METHODS[args.method]
    SUBME = args.subme
    if args.workload != "me1080p":
        run_split_workload(args, torch, dist, dev, cdev, rank, local_rank, world)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---------------- setup (untimed): planes, lookahead fields, job lists, all on the device
    es = Engine(local_rank, 8, high_priority=args.prio == "small")      # search stream (all launches, or the 16/8 class)
    el = Engine(local_rank, 8, high_priority=args.prio == "la")      # lookahead stream
    multi = args.launch in ("pairs3", "pairs4", "families")
    split = args.launch in ("split", "split3", "pairs3", "pairs4", "families")
    eb = Engine(local_rank, 8, high_priority=args.prio == "big") if split else es   # stream of the 64/32 class
    ebs = [eb] + [Engine(local_rank, 8) for _ in range(args.big_streams - 1)] if split else [es]
    em = Engine(local_rank, 8) if args.launch in ("split3", "pairs3", "pairs4") else es   # stream of the 16x16 class
    e32 = {"big": eb, "mid": em, "la": el, "own": None}[args.s32] or Engine(local_rank, 8)     # stream of the 32x32 class (pairs4)
    B = args.pairs
    clip = make_clip(W, H, B + 1, seed=1234 + rank)
    planes, lowres = [], []
    for f in clip:
        buf, org, stride = pad_plane(f, MARGIN)
        p = es.upload_plane(buf, W, H, MARGIN, MARGIN, stride)
        planes.append(p)
        lowres.append(es.lowres_init(p))
    lw, lh, wcu, hcu = lowres_dims(W, H)
    ncu = wcu * hcu
    fields = [(lowres[i + 1][0], lowres[i]) for i in range(B)]
    la_mvs, la_costs, la_stats = el.lookahead_fields(fields, wcu, hcu, LA_QP, HEX, 1, LA_MERANGE, want_stats=True)
    la_pixel_sads = int((la_stats["sad_calls"].astype(np.int64) + la_stats["subpel_calls"]).sum()) * 64

    njobs = len(workload.square_pu_jobs(W, H, la_mvs[0], wcu, hcu, MERANGE))
    host_jobs, d_jobs, d_res = [], [], []
    # one result buffer for all launches of a step (each writes its own slice), and the 8 of a record's 16 bytes the host
    # encoder reads back (MV, cost) as the persistent buffer a rank contributes to the gather
    all_res = torch.zeros((njobs * B, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    d_mvcost = torch.zeros((njobs * B, 8), dtype=torch.uint8, device=dev)
    segs = None     # per launch of a frame: (first job, count, largest PU size, engine); timed launch first
    for i in range(B):
        jobs = workload.square_pu_jobs(W, H, la_mvs[i], wcu, hcu, MERANGE)   # sorted 64, 32, 16, 8
        host_jobs.append(jobs)
        if segs is None:
            def seg(sel, dim, eng):
                idx = np.flatnonzero(sel)
                return (int(idx[0]), len(idx), dim, eng)
            if args.launch == "split":
                segs = [seg(jobs["w"] <= 16, 16, es), seg(jobs["w"] >= 32, 64, eb)]
            elif args.launch == "families":
                segs = [seg(jobs["w"] <= 16, 16, es), seg(jobs["w"] >= 32, 64, eb)]
            elif args.launch in ("split3", "pairs3"):
                segs = [seg(jobs["w"] == 8, 8, es), seg(jobs["w"] == 16, 16, em), seg(jobs["w"] >= 32, 64, eb)]
            elif args.launch == "pairs4":
                segs = [seg(jobs["w"] == 8, 8, es), seg(jobs["w"] == 16, 16, em), seg(jobs["w"] == 32, 32, e32),
                        seg(jobs["w"] == 64, 64, eb)]
            elif args.launch == "single":
                segs = [seg(jobs["w"] > 0, 64, es)]
            else:
                segs = [seg(jobs["w"] == s_, s_, es) for s_ in (8, 16, 32, 64)]
        d_jobs.append(torch.from_numpy(jobs.view(np.uint8).reshape(len(jobs), -1).copy()).to(dev))
        d_res.append(all_res[i * njobs:(i + 1) * njobs])
    if multi:
        # one job list per class over all frame pairs: job.reserved names the pair; the 64x64 PUs of all pairs come
        # before the 32x32 ones (longest waves first), the other classes are pair-major (neighbouring PUs share L2)
        d_pairs = es.pairs_upload([(planes[i + 1], planes[i]) for i in range(B)])
        cls_jobs, cls_res, cls_index = [], [], []
        res_first = 0
        for (first, cnt, dim, eng) in segs:
            sizes = sorted(set(int(w_) for w_ in host_jobs[0]["w"][first:first + cnt]))
            # workload.class_job_list: pair-major, job.reserved = pair; the 64 + 32 class of `families` in quadtree order
            if dim == 16 and args.launch == "families":
                # the 16x16 PUs and, as a list of its own, their 8x8 children (4 k + q = quadrant q of parent k): one launch
                pj_, pi_, kj_, ki_ = workload.family_lists(host_jobs, 16)
                cj, ci = np.concatenate([pj_, kj_]), np.concatenate([pi_, ki_])
                fam_parents = len(pj_)
            else:
                cj, ci = workload.class_job_list(host_jobs, sizes, families=args.launch == "families")
            assert len(cj) == cnt * B
            if dim == 64 and args.launch != "families":
                order = np.argsort(-cj["w"].astype(np.int64), kind="stable")
                cj, ci = cj[order], ci[order]
            cls_jobs.append(torch.from_numpy(cj.view(np.uint8).reshape(len(cj), -1).copy()).to(dev))
            cls_res.append(all_res[res_first:res_first + len(cj)])
            res_first += len(cj)
            cls_index.append(ci)
    d_fields = el.fields_upload(fields, wcu, hcu, HEX, LA_MERANGE)
    d_la_mvs = torch.zeros((B, ncu, 2), dtype=torch.int16, device=dev)
    d_la_costs = torch.zeros((B, ncu), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    # the roofline kernel is the first launch class (all of a frame, or its 16/8 PUs); its launches are
    # the only work on es's stream in split / single mode, so events on that stream bracket exactly them
    timed_first, timed_cnt = segs[0][0], segs[0][1]
    s_search = torch.cuda.ExternalStream(es.stream(), device=dev)
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    # The MV / cost records go to the rank that feeds the host encoder: copied out of the kernels' result buffers into
    # send buffers of their own, so the gather of step k runs while step k + 1 computes; the copies wait for the gather
    # before (it still reads the send buffers) and every gather is over before the timed region ends.
    d_la_out = torch.zeros_like(d_la_mvs)
    pending, recv = [], {}

    def gathers_wait():
        for (work, _keep) in pending:
            if work is not None:
                work.wait()
        pending.clear()

    def hand_over():
        gathers_wait()
        d_mvcost.copy_(all_res[:, :8])
        d_la_out.copy_(d_la_mvs)
        for name, t in (("mv", d_mvcost.view(-1)), ("la", d_la_out.view(torch.uint8).view(-1))):
            src = t.to(cdev)
            bufs, work = sharding.gather_records(src, dist, dst=0, bufs=recv.get(name), async_op=True)
            recv[name] = bufs
            pending.append((work, src))

    only = set(args.only.split(",")) if args.only else None

    def step_multi(k=None, gather=True):
        if only is None or "la" in only:
            el.lookahead_fields_dev(d_fields, B, wcu, hcu, LA_QP, HEX, 1, LA_MERANGE,
                                    d_la_mvs.data_ptr(), d_la_costs.data_ptr())
        for c in reversed(range(len(segs))):       # the class with the longest waves first; the timed 8x8 class last
            (first, cnt, dim, eng) = segs[c]
            if c == 0 and k is not None:
                ev0[k].record(s_search)
            if only is not None and str(dim) not in only:
                if c == 0 and k is not None:
                    ev1[k].record(s_search)
                continue
            if dim == 16 and args.launch == "families":
                eng.search_batch_pairs_family_dev(d_pairs, B, cls_jobs[c].data_ptr(), fam_parents, cls_res[c].data_ptr(),
                                                  cls_jobs[c].data_ptr() + 40 * fam_parents, cls_jobs[c].shape[0] - fam_parents,
                                                  cls_res[c].data_ptr() + 16 * fam_parents, QP, SUBME)
            else:
                eng.search_batch_pairs_dev(d_pairs, B, cls_jobs[c].data_ptr(), cls_jobs[c].shape[0], cls_res[c].data_ptr(),
                                           QP, method, SUBME, max_pu_dim=dim | SQUARE_ONLY)      # every list holds squares only
            if c == 0 and k is not None:
                ev1[k].record(s_search)
        if not sync_steps:
            return      # the next step's launches queue behind this one's on every stream; the timed region ends with a sync
        es.sync()
        eb.sync()
        em.sync()
        el.sync()
        e32.sync()
        if world > 1 and gather:
            hand_over()

    def step_pairwise(k=None, gather=True):
        el.lookahead_fields_dev(d_fields, B, wcu, hcu, LA_QP, HEX, 1, LA_MERANGE,
                                d_la_mvs.data_ptr(), d_la_costs.data_ptr())
        for (first, cnt, dim, eng) in segs[1:] if split else ():
            for i in range(B):
                (ebs[i % len(ebs)] if eng is eb else eng).search_batch_dev(
                    planes[i + 1], planes[i], d_jobs[i].data_ptr() + first * 40, cnt,
                    d_res[i].data_ptr() + first * 16, QP, method, SUBME, max_pu_dim=dim)
        if k is not None:
            ev0[k].record(s_search)
        for i in range(B):
            for (first, cnt, dim, eng) in segs if not split else segs[:1]:
                eng.search_batch_dev(planes[i + 1], planes[i], d_jobs[i].data_ptr() + first * 40, cnt,
                                     d_res[i].data_ptr() + first * 16, QP, method, SUBME, max_pu_dim=dim)
        if k is not None:
            ev1[k].record(s_search)
        es.sync()
        for e_ in ebs:
            e_.sync()
        em.sync()
        el.sync()
        if world > 1 and gather:
            hand_over()

    sync_steps = args.sync_steps == "yes" or world > 1 or not multi
    step = step_multi if multi else step_pairwise
    progress("set-up done (%d frame pairs, %d PU searches each); warm-up" % (B, njobs), rank)
    for _ in range(args.warmup):
        step()
    gathers_wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    gathers_wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    progress("%d timed steps: %.3f ms per step" % (args.steps, elapsed / args.steps * 1e3), rank)

    # the shader clock under this load: three more steps (untimed) with the probe wave beside them on a stream of its own
    clock_mhz = None
    if rank == 0 and multi and not args.no_clock_probe:
        try:
            # (rank 0 alone runs these steps: without the gather, which every rank would have to join)
            ep = Engine(local_rank, 8)
            step(gather=False)
            ep.clock_probe_start(2.5 * elapsed / args.steps * 1e3)
            for _ in range(3):
                step(gather=False)
            clock_mhz = ep.clock_probe_read()
            ep.close()
        except Exception as ex:     # the bench line must survive a failure here
            clock_mhz = None
            print("clock probe failed: %r" % (ex,), file=sys.stderr)
    torch.cuda.synchronize()

    # ---------------- accounting
    if multi:
        res = [np.zeros(njobs, RESULT_DTYPE) for _ in range(B)]
        for c in range(len(segs)):
            got = cls_res[c].cpu().numpy().view(RESULT_DTYPE).reshape(-1)
            for i in range(B):
                sel = cls_index[c][:, 0] == i
                res[i][cls_index[c][sel, 1]] = got[sel]
    else:
        res = [r.cpu().numpy().view(RESULT_DTYPE).reshape(-1) for r in d_res]
    px_search = sum(workload.pixel_sads(host_jobs[i], res[i]) for i in range(B))
    px_step = px_search + la_pixel_sads
    total_px = torch.tensor([float(px_step) * args.steps], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(total_px, op=dist.ReduceOp.SUM)
    value = float(total_px.item()) / elapsed / 1e9

    nlaunch = 1 if args.launch != "per-size" else len(segs)
    per_step = 1 if multi else B * nlaunch            # timed launches per step
    launch_ms = sum(ev0[k].elapsed_time(ev1[k]) for k in range(args.steps)) / (args.steps * per_step)
    tsel = slice(timed_first, timed_first + timed_cnt) if args.launch != "per-size" else slice(0, njobs)
    alg_bytes = sum(workload.algorithmic_bytes(j[tsel]) for j in host_jobs) / per_step
    achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath)).get("search_kernel_hbm_bytes_per_launch", {}).get(
            "pairs3" if args.launch == "pairs4" else args.launch)     # (the 8x8 launch of pairs4 is that of pairs3)
        if traffic is not None and multi:
            traffic *= B      # kept per frame pair (the counters were collected on a 2-pair step)
    # the floor of a step's HBM bytes: every plane of the frame pairs read once (padded as uploaded), job records in,
    # result records and the lookahead fields out
    plane_bytes = sum(p.stride * (p.height + 2 * p.margin_y) for p in planes)
    low_bytes = sum(q.stride * (q.height + 2 * q.margin_y) for pl in lowres for q in pl)
    comp_bytes = plane_bytes + low_bytes + njobs * B * (40 + 16) + ncu * B * 8
    compulsory = {"bytes_per_step": comp_bytes, "achieved": comp_bytes / (elapsed / args.steps) / 1e9, "unit": "GB/s",
                  "frac": comp_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                  "what": "whole step: each full-resolution and lowres plane once, job records in, result records and MV fields out"}
    vpath = os.path.join(ROOT, "profiles", "valu.json")
    # (the instruction profile is that of the family launches)
    vprof = json.load(open(vpath)) if os.path.exists(vpath) and args.launch == "families" and method == METHODS["star"] and SUBME == 5 else None

    if rank == 0:
        # parity gate + CPU baseline: the oracle re-runs every PU search of the first frame pairs
        from oracle.bindings import Oracle   # checker / baseline only
        orc = Oracle()
        cpu = None
        ncpu = min(args.cpu_pairs, B)
        if ncpu > 0:
            cpu_s, cpu_px = 0.0, 0
            for i in range(ncpu):
                progress("parity + CPU baseline: frame pair %d of %d through the oracle" % (i + 1, ncpu))
                cur, org, stride = pad_plane(clip[i + 1], MARGIN)
                ref, _, _ = pad_plane(clip[i], MARGIN)
                ctx = orc.make_ctx(8, cur, org, stride, [ref], org, stride, False, QP, method, SUBME)
                c0 = time.perf_counter()
                exp = orc.me_search_batch(ctx, host_jobs[i])
                cpu_s += time.perf_counter() - c0
                got = res[i]
                if not (np.array_equal(got["mv"], exp["mv"]) and np.array_equal(got["cost"], exp["cost"])
                        and np.array_equal(got["sad_calls"], exp["sad_calls"])
                        and np.array_equal(got["subpel_calls"], exp["subpel_calls"])):
                    print("PARITY FAILURE: GPU results of frame pair %d differ from the oracle" % i, file=sys.stderr)
                    sys.exit(2)
                cpu_px += workload.pixel_sads(host_jobs[i], exp)
            cpu = {"value": cpu_px / cpu_s / 1e9, "unit": "Gpixel-SAD/s", "cores": 1, "kind": "port",
                   "sample": "all %d PU searches of %d of the %d frame pairs of one step (the lookahead fields are not "
                             "in the sample), oracle/me_oracle.c -O3 single thread, %.1f s; every result compared "
                             "with the GPU's" % (njobs * ncpu, ncpu, B, cpu_s)}
            # where the reference's own code was built beforehand (oracle/_ref), it is the baseline: the same
            # searches through its MotionEstimate::motionEstimate and C primitives, one thread, results compared too
            from oracle.bindings import ref_available
            nref = min(args.ref_pairs, ncpu)
            if nref > 0 and ref_available(8):
                from oracle.bindings import Ref
                rf = Ref(8)
                ref_s, ref_px = 0.0, 0
                for i in range(nref):
                    progress("parity + CPU baseline: frame pair %d of %d through the reference build" % (i + 1, nref))
                    cur, org, stride = pad_plane(clip[i + 1], MARGIN)
                    ref, _, _ = pad_plane(clip[i], MARGIN)
                    rf.me_config(cur, org, stride, QP, method, SUBME)
                    c0 = time.perf_counter()
                    exp = rf.me_search_batch(ref, org, stride, host_jobs[i])
                    ref_s += time.perf_counter() - c0
                    if not (np.array_equal(res[i]["mv"], exp["mv"]) and np.array_equal(res[i]["cost"], exp["cost"])):
                        print("PARITY FAILURE: GPU results of frame pair %d differ from the reference build" % i, file=sys.stderr)
                        sys.exit(2)
                    ref_px += workload.pixel_sads(host_jobs[i], res[i])
                cpu = {"value": ref_px / ref_s / 1e9, "unit": "Gpixel-SAD/s", "cores": 1, "kind": "reference",
                       "sample": "all %d PU searches of %d of the %d frame pairs of one step through the reference's own "
                                 "MotionEstimate::motionEstimate + C primitives (oracle/_ref, g++ -O2 -ffast-math as the "
                                 "reference builds, one thread, %.1f s); MVs and costs compared with the GPU's. The oracle "
                                 "port ran %d pairs in %.1f s (also compared)" % (njobs * nref, nref, B, ref_s, ncpu, cpu_s),
                       "port_value": cpu_px / cpu_s / 1e9}
        if cpu is not None:
            # the same searches on all of the box's host cores: one MotionEstimate per thread (as the reference keeps
            # one per worker, TEncSearch.h:91), frame pairs dealt out over the threads, results compared as well
            from concurrent.futures import ThreadPoolExecutor
            nthr = host_threads(args)
            use_ref = cpu["kind"] == "reference"
            padded = [pad_plane(f, MARGIN) for f in clip]

            def work(t):
                px, ok = 0, True
                rf_t = Ref(8) if use_ref else None
                for i in range(t, B, nthr):
                    (cur, org, stride), (ref, _, _) = padded[i + 1], padded[i]
                    if use_ref:
                        rf_t.me_config(cur, org, stride, QP, method, SUBME)
                        exp = rf_t.me_search_batch(ref, org, stride, host_jobs[i])
                    else:
                        exp = orc.me_search_batch(orc.make_ctx(8, cur, org, stride, [ref], org, stride, False, QP, method, SUBME),
                                                  host_jobs[i])
                    ok &= bool(np.array_equal(res[i]["mv"], exp["mv"]) and np.array_equal(res[i]["cost"], exp["cost"]))
                    px += workload.pixel_sads(host_jobs[i], res[i])
                return px, ok
            progress("CPU baseline on %d threads: all %d frame pairs" % (nthr, B))
            c0 = time.perf_counter()
            with ThreadPoolExecutor(nthr) as ex:
                outs = list(ex.map(work, range(nthr)))
            mt_s = time.perf_counter() - c0
            if not all(o[1] for o in outs):
                print("PARITY FAILURE: GPU results differ from the CPU baseline's (threaded leg)", file=sys.stderr)
                sys.exit(2)
            one = {k: cpu[k] for k in ("value", "cores", "sample")}
            cpu.update({"value": sum(o[0] for o in outs) / mt_s / 1e9, "cores": nthr, "one_core": one,
                        "sample": "all %d PU searches of all %d frame pairs of one step on %d threads (one MotionEstimate per "
                                  "thread, frame pairs dealt out in turn), %.1f s, %s; MVs and costs compared with the GPU's"
                                  % (njobs * B, B, nthr, mt_s, "the reference's own code (oracle/_ref)" if use_ref
                                     else "oracle/me_oracle.c -O3")})
        kernel = {"families": "the family launch over the %d 16x16 PUs and their 8x8 children (%d PUs in all) of all %d frame pairs of a "
                              "step, on the context's two streams: star_small_kernel<8> (start-up candidates and the first three "
                              "pattern rounds of four 8x8 PUs per wave, and of their 16x16 parent from the children's sums where the "
                              "five jobs ask the same), star_small_kernel<16> (the parents left over), star_resume_kernel (the PUs those handed on: one per "
                              "wave), refine_small_kernel<16> (sub-pel rounds, two PUs per wave, each taking its four children along "
                              "while they decide alike) and refine_small_kernel<8> (the remaining 8x8 PUs, four per wave); the launch "
                              "of the 64x64 PUs with their 32x32 children and the lookahead fields run beside it on streams of their "
                              "own" % (int((host_jobs[0]["w"] == 16).sum()) * B, timed_cnt * B, B),
                  "pairs4": "the launch over the %d 8x8 PUs of all %d frame pairs of a step: star_small_kernel<8> (start-up and the "
                            "common case of the integer stage, four PUs per wave), search_kernel<uint8_t,false> (the PUs the first "
                            "kernel marked: one PU per wave) and refine_small_kernel<8> (sub-pel rounds, four PUs per wave); the "
                            "16x16 launch and the 64x64 + 32x32 launches run concurrently on two more streams" % (timed_cnt * B, B),
                  "pairs3": "the launch over the %d 8x8 PUs of all %d frame pairs of a step: star_small_kernel<8> (start-up and the "
                            "common case of the integer stage, four PUs per wave), search_kernel<uint8_t,false> (the PUs the first "
                            "kernel marked: one PU per wave) and refine_small_kernel<8> (sub-pel rounds, four PUs per wave); the "
                            "16x16 and the 64/32 launches run concurrently on two more streams" % (timed_cnt * B, B),
                  "split": "search_kernel<uint8_t,false>, the launch over a frame's %d 16x16 + 8x8 PUs (its 64/32 PUs run "
                           "concurrently on a second stream)" % timed_cnt,
                  "split3": "search_kernel<uint8_t,false>, the launch over a frame's %d 8x8 PUs (its 16x16 and its 64/32 PUs "
                            "run concurrently on two more streams)" % timed_cnt,
                  "single": "search_kernel<uint8_t,false>, one launch per frame (%d PUs)" % timed_cnt,
                  "per-size": "search_kernel<uint8_t,false>, average of the 4 per-size launches of a frame"}[args.launch]
        vb = valu_bound(vprof, vprof["timed_kernels"], timed_cnt * B, launch_ms * 1e-3, B, elapsed / args.steps, clock_mhz) if vprof else None
        out = {
            "metric": "ME Gpixel-SAD/s", "value": value, "unit": "Gpixel-SAD/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "1080p 8-bit, --me %s --merange %d --subme %d full-encode ME over all 64/32/16/8 "
                                   "square PUs + lookahead HEX field, %d frame pairs per GPU per step"
                                   % (args.method, MERANGE, SUBME, B),
                       "pu_searches_per_step_per_gpu": njobs * B, "lookahead_cus_per_step_per_gpu": ncu * B,
                       "qp": QP, "launch": args.launch,
                       "steps_synchronised": bool(sync_steps),
                       "job_independence": "the PU jobs of a step are searched as mutually independent: predictors and candidates come "
                                           "from the lookahead field, not from neighbours' final MVs as in predInterSearch "
                                           "(TEncSearch.cpp:4268-4292), where a legal batch is (reference x CTU row x frame) wide at "
                                           "most -- the figure is the engine's ceiling, not an encode-shaped rate (encode_fps is that)",
                       "parallelism": "frame pairs sharded over %d GPU(s), gather of MV records" % world},
            "roofline": headline_roofline(vb, timed_cnt * B, launch_ms, achieved, alg_bytes, traffic, compulsory, kernel),
            "cpu_baseline": cpu,
            "fps_equiv": B * args.steps * world / elapsed,
            "encode_fps": measured_encode() if not args.no_other_paths else None,
        }
        if not args.no_other_paths:
            # the two other search paths of the reference, measured the same way (HIP events, inputs resident);
            # reported beside the headline, never part of `value`
            try:
                from x265_amd import measure
                progress("other paths: exhaustive search 4K")
                rows, _ = measure.esa_per_size(es, dev)
                ms = sum(r["ms"] for r in rows)
                px = sum(r["pixel_sads"] for r in rows)
                progress("other paths: exhaustive search 4K, one scan per CTU")
                ctu = measure.esa_ctu_picture(es, dev, compare=True)
                ctu0 = measure.esa_ctu_picture(es, dev, spread=0)
                # parity gate: PUs of every size through the oracle, and every record against the per-size launches'
                cjobs, cres = ctu["jobs"], ctu["results"]
                pick = np.concatenate([np.random.default_rng(4).choice(np.flatnonzero(cjobs["w"] == s_), 3, replace=False) for s_ in (64, 32, 16, 8)])
                cur_, ref_, org_, st_ = ctu["planes"]
                exp = orc.me_search_batch(orc.make_ctx(8, cur_, org_, st_, [ref_], org_, st_, False, 32, METHODS["full"], 5), cjobs[pick])
                if not ctu["equal"] or not all(np.array_equal(exp[k_], cres[pick][k_]) for k_ in ("mv", "cost", "sad_calls", "subpel_calls")):
                    print("PARITY FAILURE: exhaustive search per CTU: GPU results differ from the oracle or the per-size scans", file=sys.stderr)
                    sys.exit(2)
                progress("other paths: Rockchip path, lookahead frame costs")
                rk = measure.rk_picture(es, dev)
                la = {}
                for (lw_, lh_, n_) in ((1920, 1080, 16), (7680, 4320, 4)):
                    m = measure.lookahead_costs(el, dev, lw_, lh_, n_)
                    la["%dx%d" % (lw_, lh_)] = {
                        "b_frame_triples": m["frames"], "cus_per_field": m["cus"], "ms_search_of_%d_fields" % m["fields"]: m["ms_search"],
                        "ms_intra_bidir_choice": m["ms_costs"],
                        "frame_costs_per_s": m["frames"] / (m["ms_search"] + m["ms_costs"]) * 1e3,
                        "searched_cus_per_s": m["fields"] * m["cus"] / (m["ms_search"] * 1e-3)}
                progress("other paths: the headline's searches on a noise clip and on a static clip")
                kinds = {}
                for kind in ("noise", "static", "hex"):
                    mname_ = "hex" if kind == "hex" else "star"
                    cls, (f0, f1, kj, kr) = measure.pattern_search_clip(es, dev, "texture" if kind == "hex" else kind, pairs=8, method=mname_)
                    # parity gate: a seeded sample of pair 0's PUs of every size through the oracle
                    pick = np.concatenate([np.random.default_rng(3).choice(np.flatnonzero(kj["w"] == s_), n_, replace=False)
                                           for s_, n_ in ((64, 24), (32, 48), (16, 96), (8, 192))])
                    cur_, org_, st_ = pad_plane(f1, 128)
                    ref_, _, _ = pad_plane(f0, 128)
                    exp = orc.me_search_batch(orc.make_ctx(8, cur_, org_, st_, [ref_], org_, st_, False, 32, METHODS[mname_], 5), kj[pick])
                    if not all(np.array_equal(exp[k_], kr[pick][k_]) for k_ in ("mv", "cost", "sad_calls", "subpel_calls")):
                        print("PARITY FAILURE: %s clip: GPU results differ from the oracle" % kind, file=sys.stderr)
                        sys.exit(2)
                    ms_ = sum(v["ms"] for v in cls.values())
                    px_ = sum(v["pixel_sads"] for v in cls.values())
                    kinds[kind] = {"ms_per_8_frame_pairs": ms_, "value": px_ / ms_ / 1e6, "unit": "Gpixel-SAD/s",
                                   "pu_searches": sum(v["pus"] for v in cls.values()),
                                   "per_class": {k_: {"pus": v["pus"], "ms": v["ms"], "marked_frac": v["marked_frac"],
                                                      "ns_per_pu": v["ms"] * 1e6 / v["pus"],
                                                      "Gpixel_SAD_per_s": v["pixel_sads"] / v["ms"] / 1e6} for k_, v in cls.items()},
                                   "parity": "360 PUs of the first frame pair (24 / 48 / 96 / 192 of 64 / 32 / 16 / 8 samples, seeded) "
                                             "compared with the oracle: MVs, costs, visit counts"}
                progress("other paths: the 2NxN / Nx2N partitions of the headline's clip")
                rect_cls, (f0, f1, kj, kr) = measure.other_partitions_clip(es, dev, pairs=8)
                pick = np.random.default_rng(9).choice(len(kj), 256, replace=False)
                cur_, org_, st_ = pad_plane(f1, 128)
                ref_, _, _ = pad_plane(f0, 128)
                exp = orc.me_search_batch(orc.make_ctx(8, cur_, org_, st_, [ref_], org_, st_, False, 32, METHODS["star"], 5), kj[pick])
                if not all(np.array_equal(exp[k_], kr[pick][k_]) for k_ in ("mv", "cost", "sad_calls", "subpel_calls")):
                    print("PARITY FAILURE: 2NxN / Nx2N partitions: GPU results differ from the oracle", file=sys.stderr)
                    sys.exit(2)
                progress("other paths: the headline's searches CTU row by CTU row")
                e_row = Engine(local_rank, 8)
                wf = measure.ctu_row_wavefront(es, e_row, dev, pairs=32)
                e_row.close()
                pick = np.random.default_rng(10).choice(len(wf["jobs"]), 256, replace=False)
                cur_, org_, st_ = pad_plane(wf["frames"][1], 128)
                ref_, _, _ = pad_plane(wf["frames"][0], 128)
                exp = orc.me_search_batch(orc.make_ctx(8, cur_, org_, st_, [ref_], org_, st_, False, 32, METHODS["star"], 5), wf["jobs"][pick])
                if not all(np.array_equal(exp[k_], wf["results"][pick][k_]) for k_ in ("mv", "cost", "sad_calls", "subpel_calls")):
                    print("PARITY FAILURE: CTU-row launches: GPU results differ from the oracle", file=sys.stderr)
                    sys.exit(2)
                progress("other paths: motion compensation, bi-prediction")
                mc = measure.mc_picture(es, dev)
                bp = measure.bipred_picture(es, dev)
                progress("other paths: Main10")
                e10 = Engine(local_rank, 10)
                m10 = measure.pattern_search_picture(e10, dev)
                e10b = Engine(local_rank, 10)
                m10f = measure.pattern_search_picture_families(e10, e10b, dev)
                e10b.close()
                j10 = m10f["jobs"]
                pick = np.concatenate([np.random.default_rng(8).choice(np.flatnonzero(j10["w"] == s_), n_, replace=False)
                                       for s_, n_ in ((64, 6), (32, 12), (16, 24), (8, 48))])
                cur_, org_, st_ = pad_plane(m10f["planes"][1], 128)
                ref_, _, _ = pad_plane(m10f["planes"][0], 128)
                exp = orc.me_search_batch(orc.make_ctx(10, cur_, org_, st_, [ref_], org_, st_, False, 38, METHODS["star"], 5), j10[pick])
                if not all(np.array_equal(m10f["results"][k_], m10["results"][k_]) and np.array_equal(exp[k_], m10f["results"][pick][k_])
                           for k_ in ("mv", "cost", "sad_calls", "subpel_calls")):
                    print("PARITY FAILURE: Main10 4K: GPU records differ between the launch forms or from the oracle", file=sys.stderr)
                    sys.exit(2)
                rows10, _ = measure.esa_per_size(e10, dev)
                ms10 = sum(r["ms"] for r in rows10)
                px10 = sum(r["pixel_sads"] for r in rows10)
                ctu10 = measure.esa_ctu_picture(e10, dev, compare=True)
                cj10, cr10 = ctu10["jobs"], ctu10["results"]
                pick = np.concatenate([np.random.default_rng(6).choice(np.flatnonzero(cj10["w"] == s_), 2, replace=False) for s_ in (64, 32, 16, 8)])
                cur_, ref_, org_, st_ = ctu10["planes"]
                exp = orc.me_search_batch(orc.make_ctx(10, cur_, org_, st_, [ref_], org_, st_, False, 32, METHODS["full"], 5), cj10[pick])
                if not ctu10["equal"] or not all(np.array_equal(exp[k_], cr10[pick][k_]) for k_ in ("mv", "cost", "sad_calls", "subpel_calls")):
                    print("PARITY FAILURE: exhaustive search per CTU, 10-bit: GPU results differ from the oracle or the per-size scans", file=sys.stderr)
                    sys.exit(2)
                rk10 = measure.rk_picture(e10, dev, qp=42)
                e10.close()
                out["other_paths"] = {
                    "me1080p_noise": dict(kinds["noise"], workload="the headline's searches (1080p 8-bit, --me star --merange 57 --subme 5, every square "
                                          "64/32/16/8 PU, predictors from the lookahead field) on a clip of independent uniform noise frames: the "
                                          "pattern search's worst case, most PUs run on past the first three rounds (marked_frac) through raster "
                                          "and re-centred patterns; 8 frame pairs, the three launch classes one after the other on one stream"),
                    "me1080p_static": dict(kinds["static"], workload="the same on a clip whose frames are all alike: the best case, every search "
                                           "ends at its start"),
                    "me1080p_hex": dict(kinds["hex"], workload="the headline's PUs and textured clip searched with --me hex (the reference's default "
                                        "method, BASELINE configs[0]): first hexagon ring + closing square in the fixed-shape kernels, PUs whose ring "
                                        "moves (marked_frac) walk on one per wave; 8 frame pairs, classes one after the other on one stream"),
                    "me1080p_ctu_rows": {"workload": "the headline's searches (same clip, same 32 frame pairs, same PUs) in the shape an encode "
                                                     "can batch them: one CTU row of every frame pair per dependent step (predInterSearch needs "
                                                     "the row above's final MVs for its predictors, TEncSearch.cpp:4268-4292), two launches side "
                                                     "by side per row as in the headline, both waited for on the host before the next row; no "
                                                     "lookahead launch beside them",
                                         "ctu_rows": wf["rows"], "ms_per_32_frame_pairs": wf["ms"], "pu_searches": wf["pus"],
                                         "ms_per_row_step": wf["ms"] / wf["rows"],
                                         "value": wf["pixel_sads"] / wf["ms"] / 1e6, "unit": "Gpixel-SAD/s",
                                         "parity": "256 PUs of the first frame pair (seeded) compared with the oracle"},
                    "me1080p_rect": {"workload": "the 2NxN and Nx2N partitions of every CU of the headline's clip (1080p 8-bit, --me star "
                                                 "--merange 57 --subme 5, 8 frame pairs): one launch per CU size with both shapes mixed; "
                                                 "16x8 / 8x16 and 8x4 / 4x8 run two and eight PUs per wave (star_tiny_kernel, "
                                                 "refine_tiny_kernel), the larger shapes a wave per PU",
                                     "ms_per_8_frame_pairs": sum(v["ms"] for v in rect_cls.values()),
                                     "pu_searches": sum(v["pus"] for v in rect_cls.values()),
                                     "value": sum(v["pixel_sads"] for v in rect_cls.values()) / sum(v["ms"] for v in rect_cls.values()) / 1e6,
                                     "unit": "Gpixel-SAD/s",
                                     "per_class": {k_: {"pus": v["pus"], "ms": v["ms"], "ns_per_pu": v["ms"] * 1e6 / v["pus"],
                                                        "Gpixel_SAD_per_s": v["pixel_sads"] / v["ms"] / 1e6} for k_, v in rect_cls.items()},
                                     "parity": "256 PUs of the first frame pair (all shapes, seeded) compared with the oracle: MVs, costs, visit counts"},
                    "main10_4k": {"workload": "3840x2160 10-bit, --me star --merange 57 --subme 5, all 64/32/16/8 square PUs of one "
                                              "picture against one reference (BASELINE configs[3], per reference)",
                                  "ms_per_picture_reference": m10f["ms"], "pu_searches": m10f["pus"],
                                  "value": m10f["pixel_sads"] / m10f["ms"] / 1e6, "unit": "Gpixel-SAD/s",
                                  "launch": "two launches side by side, as the headline's: the 16x16 PUs with the list of their 8x8 "
                                            "children on one context, the 64x64 PUs each followed by their 32x32 children on another; "
                                            "host clock from the first launch to the end of both, every picture waited for",
                                  "one_launch_per_size": {"ms_per_picture_reference": m10["ms"], "value": m10["pixel_sads"] / m10["ms"] / 1e6},
                                  "parity": "records equal between the two launch forms; 90 PUs (6 / 12 / 24 / 48 per size, seeded) through the oracle"},
                    "lookahead_frame_cost": dict(la, workload="estimateFrameCost of B-frame triples (b-1, b, b+1): two lowres HEX "
                                                 "subme-1 merange-16 MV fields each, then 35-mode intra, bidirectional candidates, "
                                                 "list choice and row / frame sums (BASELINE configs[4] at 7680x4320)"),
                    "esa_4k": {"workload": "3840x2160 8-bit, --me full --merange 92 --subme 5, all 64/32/16/8 square PUs of one "
                                           "picture against one reference (BASELINE configs[2], per reference)",
                               "ms_per_picture_reference": ms, "value": px / ms / 1e6, "unit": "Gpixel-SAD/s",
                               "bound": "valu", "peak": measure.SAD_ISSUE_PEAK / 1e9,
                               "frac": px / (ms * 1e-3) / measure.SAD_ISSUE_PEAK,
                               "per_pu_size_ms": {str(r["size"]): r["ms"] for r in rows}},
                    "esa_4k_ctu": {"workload": "the same searches as esa_4k through x265amd_esa_ctu_batch_dev: one scan per 64x64 CTU measures its "
                                               "64 8x8 blocks over the bounding box of the PUs' ranges and sums them up the quadtree for all 85 PUs "
                                               "(the predictors of the stand-in field lie up to 6 samples apart; `smooth_field`: one predictor for all)",
                                   "ms_per_picture_reference": ctu["ms"], "value": ctu["pixel_sads"] / ctu["ms"] / 1e6,
                                   "unit": "Gpixel-SAD/s credited (every PU its own range; the scan measures a quarter of them)",
                                   "smooth_field": {"ms_per_picture_reference": ctu0["ms"], "value": ctu0["pixel_sads"] / ctu0["ms"] / 1e6},
                                   "bound": "valu", "peak": measure.SAD_ISSUE_PEAK / 1e9,
                                   "frac": ctu0["pixel_sads"] / 4 / (ctu0["ms"] * 1e-3) / measure.SAD_ISSUE_PEAK,
                                   "frac_note": "SADs the scan measures (the 8x8 level: a quarter of the credited count) against the byte-SAD "
                                                "issue peak, smooth field, finishing stages and sub-pel refinement inside the time",
                                   "parity": "every record equal to the per-size launches'; 12 PUs (3 per size, seeded) through the oracle"},
                    "esa_4k_main10": {"workload": "the same exhaustive search on 10-bit samples (v_sad_u16: two samples per lane "
                                                  "and instruction, half the byte-SAD peak)",
                                      "ms_per_picture_reference": ms10, "value": px10 / ms10 / 1e6, "unit": "Gpixel-SAD/s",
                                      "bound": "valu", "peak": measure.SAD_ISSUE_PEAK / 2e9,
                                      "frac": px10 / (ms10 * 1e-3) / (measure.SAD_ISSUE_PEAK / 2),
                                      "per_pu_size_ms": {str(r["size"]): r["ms"] for r in rows10}},
                    "esa_4k_main10_ctu": {"workload": "the 10-bit searches through the scan per CTU (v_sad_u16, 2 x 8 displacements per lane and task)",
                                          "ms_per_picture_reference": ctu10["ms"], "value": ctu10["pixel_sads"] / ctu10["ms"] / 1e6,
                                          "unit": "Gpixel-SAD/s credited",
                                          "parity": "every record equal to the per-size launches'; 8 PUs (2 per size, seeded) through the oracle"},
                    "mc_4k": {"workload": "3840x2160 8-bit TComPrediction::motionCompensation of every PU of a picture (mixed quadtree "
                                          "partitioning, luma + 4:2:0 chroma, random quarter-pel MVs), device-resident job list",
                              "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              **{k: {"ms_per_picture": v["ms"], "pus": v["pus"], "achieved": v["algorithmic_bytes"] / v["ms"] / 1e6,
                                     "frac": v["algorithmic_bytes"] / v["ms"] / 1e6 / HBM_PEAK_GBS,
                                     "Gsample_per_s": v["samples"] / v["ms"] / 1e6} for k, v in mc.items()}},
                    "bipred_4k": {"workload": "3840x2160 8-bit bi-prediction estimate of predInterSearch for every PU of a picture (mixed quadtree "
                                              "partitioning, random quarter-pel MVs), device-resident list",
                                  "ms_per_picture": bp["ms"], "pus": bp["pus"], "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "achieved": bp["algorithmic_bytes"] / bp["ms"] / 1e6,
                                  "frac": bp["algorithmic_bytes"] / bp["ms"] / 1e6 / HBM_PEAK_GBS},
                    "rk_1080p": {"workload": "1920x1080 8-bit, Rockchip rd>=5 path: CIME + RIME + FME of all 510 CTUs (85 CUs each) "
                                             "against one reference, default RK range",
                                 "ms_per_picture_reference": rk["ms"], "ctus_per_s": rk["ctus"] / (rk["ms"] * 1e-3),
                                 "value": rk["pixel_sads"] / rk["ms"] / 1e6, "unit": "Gpixel-SAD/s"},
                    "rk_1080p_main10": {"workload": "the same on 10-bit samples", "ms_per_picture_reference": rk10["ms"],
                                        "ctus_per_s": rk10["ctus"] / (rk10["ms"] * 1e-3),
                                        "value": rk10["pixel_sads"] / rk10["ms"] / 1e6, "unit": "Gpixel-SAD/s"}}
            except Exception as ex:   # the headline line must survive a failure here
                out["other_paths"] = {"error": repr(ex)}
        progress("done")
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
