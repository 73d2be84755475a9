#!/usr/bin/env python3
"""Summaries of the rocprofv3 passes of tools/profile_round.sh for profiles/:

    python tools/profile_summary.py gpurun_out r02

  profiles/<tag>_bench_kernel_trace.txt   per (kernel, grid): calls, average / median / min / max duration; per stream
                                          of a step (= launch class), the span from the start of its first kernel to
                                          the end of its last one (a launch class uses two streams: bench.py's HIP events bracket
                                          the 16x16 + 8x8 class's launch on its first)
  profiles/<tag>_search_sq_counters.txt   SQ counters per (kernel, grid), totals and per wave
  profiles/valu.json                      the same per frame pair, read by bench.py for roofline.valu
  profiles/traffic.json                   HBM bytes per frame pair of the 16x16 + 8x8 launch class (FETCH_SIZE doubled for the
                                          128-byte requests gfx950 counts as 64, plus WRITE_SIZE)
"""
import collections
import csv
import glob
import json
import os
import re
import sys

root, tag = sys.argv[1], sys.argv[2]
PROF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
PAIRS_PMC = 2            # the counter passes run bench.py --pairs 2 --steps 1 --warmup 0
PUS_8 = 32400            # 8x8 PUs of a 1080p picture
STEPS = 5


def one(pattern):
    f = sorted(glob.glob(os.path.join(root, pattern), recursive=True), key=os.path.getmtime)
    assert f, pattern
    return f[-1]


def short(name):
    m = re.match(r"(?:void )?(?:[A-Za-z0-9_]+::)*([A-Za-z0-9_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


def class_of(kernels):
    """name of a stream's launch class from the kernels it ran in a step"""
    names = " ".join(kernels)
    parts = []
    if "lookahead_band_kernel" in names:
        parts.append("lookahead")
    for s in ("8", "16"):
        if re.search(r"(star_small_kernel|refine_small_kernel)<%s[,>]" % s, names):
            parts.append("%sx%s" % (s, s))
    for s in ("32", "64"):
        if re.search(r"refine_big_kernel<%s[,>]" % s, names):
            parts.append("%sx%s" % (s, s))
    return " + ".join(parts) or "handed-on PUs only"


# ---- kernel trace
rows = list(csv.DictReader(open(one("%s_trace/**/*kernel_trace.csv" % tag))))
agg = collections.defaultdict(list)
ev = []
for r in rows:
    k, g, s, e = short(r["Kernel_Name"]), int(r["Grid_Size_X"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    agg[(k, g)].append((s, e))
    ev.append((s, e, k, g, r["Queue_Id"]))
ev.sort()
lines = ["rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --cpu-pairs 0 --no-other-paths --no-clock-probe",
         "(set-up launches included in the call counts; durations in microseconds)", "",
         "%-44s %9s %6s %9s %9s %9s %9s" % ("kernel", "grid", "calls", "avg", "median", "min", "max")]
for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    d = sorted((e - s) / 1e3 for s, e in v)
    lines.append("%-44s %9d %6d %9.1f %9.1f %9.1f %9.1f" % (k, g, len(d), sum(d) / len(d), d[len(d) // 2], d[0], d[-1]))
# the steps: each begins with the launch of the lookahead field; the last STEPS of them are the timed ones
starts = [e[0] for e in ev if e[2].startswith("lookahead_band_kernel")][-STEPS:]
bounds = starts + [max(e[1] for e in ev) + 1]
per_class = collections.defaultdict(list)
for i in range(STEPS):
    by_q = collections.defaultdict(list)
    for e in ev:
        if bounds[i] <= e[0] < bounds[i + 1] and not e[2].startswith("__amd_rocclr") and "elementwise" not in e[2]:
            by_q[e[4]].append(e)
    by_c = collections.defaultdict(list)
    for q, es in by_q.items():
        by_c[class_of([e[2] for e in es])] += es
    for cname, es in by_c.items():
        per_class[cname].append(sorted(es))
lines += ["", "launch classes of the %d timed steps (a context's two streams each): span = end of the class's last kernel - start of "
          "its first, per step; then the kernels in start order (of the first of these steps) with start, end and duration "
          "relative to the step's first kernel, averaged over the steps" % STEPS]
for cname, steps in sorted(per_class.items()):
    sp = [(max(e[1] for e in es) - min(e[0] for e in es)) / 1e6 for es in steps]
    ks = [sum(e[1] - e[0] for e in es) / 1e6 for es in steps]
    lines.append("%-18s span avg %.3f ms (min %.3f max %.3f); sum of its kernels' durations avg %.3f ms" %
                 (cname, sum(sp) / len(sp), min(sp), max(sp), sum(ks) / len(ks)))
    # the same launch sequence every step: match kernels by (name, grid, occurrence)
    def keyed(es):
        seen, out = collections.Counter(), {}
        for e in es:
            seen[(e[2], e[3])] += 1
            out[(e[2], e[3], seen[(e[2], e[3])])] = e
        return out
    ks_ = [keyed(es) for es in steps]
    for key, e0 in sorted(ks_[0].items(), key=lambda kv: kv[1][0]):
        have = [(k_[key], bounds[i]) for i, k_ in enumerate(ks_) if key in k_]
        n_ = len(have)
        lines.append("    %-44s grid %9d  %6.3f -> %6.3f  (%.3f ms)" % (
            key[0], key[1], sum(e[0] - b for e, b in have) / n_ / 1e6, sum(e[1] - b for e, b in have) / n_ / 1e6,
            sum(e[1] - e[0] for e, b in have) / n_ / 1e6))
period = [(starts[i + 1] - starts[i]) / 1e6 for i in range(STEPS - 1)]
lines.append("step period (start of one step's first kernel to the next one's) avg %.3f ms" % (sum(period) / len(period)))
open(os.path.join(PROF, "%s_bench_kernel_trace.txt" % tag), "w").write("\n".join(lines) + "\n")

# ---- SQ counters
rows = list(csv.DictReader(open(one("%s_sq/**/*counter_collection.csv" % tag))))
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
queue_of = {}
for r in rows:
    key = (short(r["Kernel_Name"]), int(r["Grid_Size"]), r["Queue_Id"])
    cnt[key][r["Counter_Name"]] += float(r["Counter_Value"])
lines = ["rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES "
         "SQ_WAIT_INST_ANY -- python3 bench.py --steps 1 --warmup 0 --pairs 2 --cpu-pairs 0 --no-other-paths --no-clock-probe",
         "(totals over the run: one step of 2 frame pairs; kernels of the set-up included under their own grids)", ""]
for (k, g, q), v in sorted(cnt.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    w = v.get("SQ_WAVES", 0)
    lines.append("%s  grid %d  queue %s" % (k, g, q))
    for c, x in sorted(v.items()):
        lines.append("   %-22s %12.4g   per wave %10.1f" % (c, x, x / w if w else float("nan")))
open(os.path.join(PROF, "%s_search_sq_counters.txt" % tag), "w").write("\n".join(lines) + "\n")


def per_pair(v):
    return {"valu": v["SQ_INSTS_VALU"] / PAIRS_PMC, "salu": v["SQ_INSTS_SALU"] / PAIRS_PMC, "lds": v["SQ_INSTS_LDS"] / PAIRS_PMC,
            "waves": v["SQ_WAVES"] / PAIRS_PMC, "wave_cycles": v["SQ_WAVE_CYCLES"] / PAIRS_PMC}


# A launch class runs on a context's two streams: the queues that ran a fixed-shape kernel of the small PUs belong to the
# 16x16 + 8x8 family launch (the timed class), those that ran one of the large PUs to the 64x64 + 32x32 launch. A context's
# second stream may hold nothing but the compact search of handed-on PUs and its list kernel: those go by their grids
# (the counter run's lists are shorter than the cap of 8192 waves only for the large PUs).
FAMILY, BIG = "16x16 + 8x8", "64x64 + 32x32"
q_class = {}
for (k, g, q) in cnt:
    if k.startswith(("star_small_kernel", "refine_small_kernel")):
        q_class[q] = FAMILY
    elif k.startswith(("star_big_kernel", "refine_big_kernel")):
        q_class.setdefault(q, BIG)
BIG_PUS = (480 + 1980) * PAIRS_PMC      # 64x64 + 32x32 PUs of the counter run's step


def class_of_launch(k, g, q):
    if q in q_class:
        return q_class[q]
    if k.startswith(("star_resume_kernel", "search_kernel")):
        return BIG if g < 8192 * 64 else FAMILY
    if k == "redo_list_kernel":
        return BIG if g <= 64 * ((BIG_PUS + 1023) // 1024) else FAMILY
    return "?"


kern, timed = {}, []
for (k, g, q), v in cnt.items():
    if k.startswith(("star_small_kernel", "refine_small_kernel", "star_big_kernel", "refine_big_kernel", "lookahead_band_kernel")):
        name = k
    elif k.startswith(("redo_list_kernel", "star_resume_kernel", "search_kernel")):
        name = "%s (%s)" % (k, class_of_launch(k, g, q))
    else:
        continue
    if name in kern:
        for c, x in per_pair(v).items():
            kern[name][c] += x
    else:
        kern[name] = per_pair(v)
    if (k.startswith(("star_small", "refine_small")) or name.endswith("(%s)" % FAMILY)) and name not in timed:
        timed.append(name)
PUS_16 = 8040            # 120 x 67 whole 16x16 PUs (the 240 8x8 PUs of the last half row have no parent)
# Essential wave64 VALU instructions of a STAR / subme 5 search of an SxS PU, counted from the algorithm and the
# instruction set alone: a full-pel candidate is S*S/256 v_sad_u8 wave instructions (64 lanes x 4 samples) + 1 for the MV
# cost and the comparison; an interpolated SATD candidate costs per sample 8/3 multiply-adds of the horizontal pass (a
# filtered plane serves the three candidates of a column of the square), 4 two-sample dot products of the vertical pass, 1
# difference, 2 packed butterflies of the 4x4 Hadamard transform and 1 absolute-value add: 10.7 lane instructions.
FULLPEL, SUBPEL, LANE_PER_SAMPLE = 21, 18, 8 / 3 + 4 + 1 + 2 + 1


def essential(S, fullpel=True, subpel=True):
    return (FULLPEL * (S * S / 256 + 1) if fullpel else 0) + (SUBPEL * S * S * LANE_PER_SAMPLE / 64 if subpel else 0)


ess_family = (PUS_16 * essential(16) + PUS_8 * essential(8, subpel=False)) / (PUS_16 + PUS_8)
json.dump({"source": "profiles/%s_search_sq_counters.txt (rocprofv3 --pmc SQ_*, bench.py --pairs 2 --steps 1 --warmup 0)" % tag,
           "simds": 1024, "clock_hz": 2.4e9, "pus_per_pair": PUS_16 + PUS_8, "timed_kernels": timed,
           "essential_wave_instr_per_pu": {
               "value": ess_family,
               "per_pu_size": {str(S): essential(S) for S in (8, 16, 32, 64)},
               "how": "per SxS PU: %d full-pel candidates x (S*S/256 v_sad_u8 wave instructions + 1) + %d interpolated SATD candidates x "
                      "S*S x %.1f lane instructions per sample / 64 lanes (horizontal 8-tap pass shared by three candidates 8/3, vertical "
                      "pass as two-sample dot products 4, difference 1, packed 4x4 Hadamard 2, absolute sum 1); no start-up "
                      "interpolation, staging, reduction or control flow. The timed class holds a 16x16 PU per four 8x8 PUs, and a "
                      "candidate's SATD of a 16x16 PU is the sum of its children's: value = (essential(16) + 4 x the full-pel part "
                      "of essential(8)) / 5 PUs" % (FULLPEL, SUBPEL, LANE_PER_SAMPLE)},
           "valu_wave_instr_per_pair_all_kernels": sum(v["valu"] for v in kern.values()),
           "kernels_per_pair": kern}, open(os.path.join(PROF, "valu.json"), "w"), indent=1)

# ---- HBM traffic of the timed launch class
tr = {}
for what, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    rows = list(csv.DictReader(open(one("%s_%s/**/*counter_collection.csv" % (tag, what)))))
    for r in rows:
        key = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
        if r["Counter_Name"] == counter:
            tr.setdefault(key, {}).setdefault(counter, 0.0)
            tr[key][counter] += float(r["Counter_Value"])
            if r["Queue_Id"] in q_class:
                tr[key]["class"] = q_class[r["Queue_Id"]]       # (queue ids of one run; the classes' grids differ anyway)
tpath = os.path.join(PROF, "traffic.json")
t = json.load(open(tpath))
detail = {}
total = 0.0
for (k, g), v in tr.items():
    if not k.startswith(("star_small", "refine_small", "star_big", "refine_big", "star_resume", "search_kernel", "lookahead_band", "redo_list")):
        continue
    b = (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 / PAIRS_PMC
    cls = FAMILY if k.startswith(("star_small", "refine_small")) else class_of_launch(k, g, None) if not k.startswith(("star_big", "refine_big", "lookahead")) else ""
    detail["%s grid %d" % (k, g)] = {"FETCH_SIZE_KB": v.get("FETCH_SIZE", 0) / PAIRS_PMC, "WRITE_SIZE_KB": v.get("WRITE_SIZE", 0) / PAIRS_PMC,
                                      "hbm_bytes_per_pair": b, "class": cls}
    if cls == FAMILY:
        total += b
t["search_kernel_hbm_bytes_per_launch"]["families"] = total
t["families_detail"] = {"unit": "per frame pair (the step of the counter run held 2 pairs; bench.py multiplies by its pairs per step)",
                        "correction": "HBM bytes = 2 x FETCH_SIZE KB (gfx950 counts a 128-byte request as 64 bytes) + WRITE_SIZE KB, "
                                      "separate --pmc passes (tools/profile_round.sh)",
                        "timed_class": "the family launch: star_small_kernel<16> and <8>, the list kernels and compact searches of "
                                       "their handed-on PUs, refine_small_kernel<16> (with the 8x8 children it takes along) and <8>",
                        "kernels": detail}
json.dump(t, open(tpath, "w"), indent=1)
print("\n".join(open(os.path.join(PROF, "%s_bench_kernel_trace.txt" % tag)).read().splitlines()))
