#!/usr/bin/env python3
"""Summaries of the rocprofv3 passes of tools/profile_round.sh for profiles/:

    python tools/profile_summary.py gpurun_out r02

  profiles/<tag>_bench_kernel_trace.txt   per (kernel, grid): calls, average / median / min / max duration; per stream
                                          of a step (= launch class), the span from the start of its first kernel to
                                          the end of its last one (what bench.py's HIP events bracket for the 8x8 class)
  profiles/<tag>_search_sq_counters.txt   SQ counters per (kernel, grid), totals and per wave
  profiles/valu.json                      the same per frame pair, read by bench.py for roofline.valu
  profiles/traffic.json                   HBM bytes per frame pair of the 8x8 launch class (FETCH_SIZE doubled for the
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
        if re.search(r"star_small_kernel<%s[,>]" % s, names):
            parts.append("%sx%s" % (s, s))
    for s in ("32", "64"):
        if re.search(r"refine_big_kernel<%s[,>]" % s, names):
            parts.append("%sx%s" % (s, s))
    return " + ".join(parts) or "other"


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
    for q, es in by_q.items():
        per_class[class_of([e[2] for e in es])].append(es)
lines += ["", "launch classes of the %d timed steps (one stream each): span = end of the class's last kernel - start of its first, "
          "per step; then the kernels in launch order with their average durations" % STEPS]
for cname, steps in sorted(per_class.items()):
    sp = [(max(e[1] for e in es) - min(e[0] for e in es)) / 1e6 for es in steps]
    ks = [sum(e[1] - e[0] for e in es) / 1e6 for es in steps]
    lines.append("%-18s span avg %.3f ms (min %.3f max %.3f); sum of its kernels' durations avg %.3f ms" %
                 (cname, sum(sp) / len(sp), min(sp), max(sp), sum(ks) / len(ks)))
    n = min(len(es) for es in steps)
    for j in range(n):
        lines.append("    %-44s grid %9d  avg %.3f ms" % (steps[0][j][2], steps[0][j][3], sum(es[j][1] - es[j][0] for es in steps) / len(steps) / 1e6))
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


# the stream of a size class is the one its fixed-shape kernels ran on; the one-wave search launches (marked PUs of the
# small classes, the integer stage of the large ones) are told apart by that stream
q_of = {}
for (k, g, q) in cnt:
    for s in ("8", "16"):
        if re.match(r"star_small_kernel<%s[,>]" % s, k):
            q_of[q] = q_of.get(q, []) + [s]
    for s in ("32", "64"):
        if re.match(r"refine_big_kernel<%s[,>]" % s, k):
            q_of[q] = q_of.get(q, []) + [s]
kern, timed = {}, []
for (k, g, q), v in cnt.items():
    if k.startswith(("star_small_kernel", "refine_small_kernel", "refine_big_kernel", "redo_list_kernel", "lookahead_band_kernel")):
        name = k + (" (%s)" % "/".join(q_of.get(q, ["?"])) if k == "redo_list_kernel" else "")
    elif k.startswith("search_kernel"):
        waves = v["SQ_WAVES"]
        sizes = q_of.get(q, ["?"])
        # the integer stage of the 32x32 / 64x64 PUs has a wave per PU; the launches over marked PUs have at most 8192 waves
        big = [s for s in sizes if s in ("32", "64") and abs(waves - {"32": 1980, "64": 480}[s] * PAIRS_PMC) < 1]
        both = "32" in sizes and "64" in sizes and abs(waves - (1980 + 480) * PAIRS_PMC) < 1    # one launch for parents and children
        name = "search_kernel (%s)" % ("integer stage of the 64x64 and 32x32 PUs" if both else
                                       "integer stage of the %sx%s PUs" % (big[0], big[0]) if big else
                                       "PUs star_small_kernel marked, %s" % "/".join(s for s in sizes if s in ("8", "16")))
    else:
        continue
    if name in kern:
        for c, x in per_pair(v).items():
            kern[name][c] += x
    else:
        kern[name] = per_pair(v)
    if "8" in q_of.get(q, []) and not k.startswith(("lookahead", "refine_big")) and name not in timed:
        timed.append(name)
json.dump({"source": "profiles/%s_search_sq_counters.txt (rocprofv3 --pmc SQ_*, bench.py --pairs 2 --steps 1 --warmup 0)" % tag,
           "simds": 1024, "clock_hz": 2.4e9, "pus_per_pair": PUS_8, "timed_kernels": timed,
           "essential_wave_instr_per_pu": {"value": 175, "how": "an 8x8 STAR / subme 5 search visits about 21 full-pel and 18 sub-pel "
                                           "candidates; at four PUs per wave (16 lanes x 4 pixels per PU) a SAD of 64 pixels is 1 "
                                           "v_sad_u8 + 4 cross-lane adds, an interpolated candidate about 8 multiply-adds per "
                                           "filter pass and a packed Hadamard of about 20: 150-200 wave instructions per PU "
                                           "before any control flow"},
           "valu_wave_instr_per_pair_all_kernels": sum(v["valu"] for v in kern.values()),
           "kernels_per_pair": kern}, open(os.path.join(PROF, "valu.json"), "w"), indent=1)

# ---- HBM traffic of the 8x8 launch class
tr = {}
for what, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    rows = list(csv.DictReader(open(one("%s_%s/**/*counter_collection.csv" % (tag, what)))))
    for r in rows:
        key = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
        if r["Counter_Name"] == counter:
            tr.setdefault(key, {}).setdefault(counter, 0.0)
            tr[key][counter] += float(r["Counter_Value"])
tpath = os.path.join(PROF, "traffic.json")
t = json.load(open(tpath))
detail = {}
total = 0.0
for (k, g), v in tr.items():
    if not k.startswith(("star_small", "refine_small", "refine_big", "search_kernel", "lookahead_band", "redo_list")):
        continue
    b = (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 / PAIRS_PMC
    detail["%s grid %d" % (k, g)] = {"FETCH_SIZE_KB": v.get("FETCH_SIZE", 0) / PAIRS_PMC, "WRITE_SIZE_KB": v.get("WRITE_SIZE", 0) / PAIRS_PMC,
                                      "hbm_bytes_per_pair": b}
    if re.match(r"(star_small_kernel|refine_small_kernel)<8[,>]", k) or k == "redo_list_kernel" and g == (PUS_8 * PAIRS_PMC + 2047) // 2048 * 256:
        total += b
# the one-wave search over the marked 8x8 PUs shares its grid with the 16x16 class's: half of the pair is charged
red = tr.get(("search_kernel<unsigned char, false>", 8192 * 64))
if red:
    total += (2 * red.get("FETCH_SIZE", 0) + red.get("WRITE_SIZE", 0)) * 1024 / PAIRS_PMC / 2
t["search_kernel_hbm_bytes_per_launch"]["pairs3"] = total
t["pairs3_detail"] = {"unit": "per frame pair (the step of the counter run held 2 pairs; bench.py multiplies by its pairs per step)",
                      "correction": "HBM bytes = 2 x FETCH_SIZE KB (gfx950 counts a 128-byte request as 64 bytes) + WRITE_SIZE KB, "
                                    "separate --pmc passes (tools/profile_round.sh)",
                      "timed_class": "star_small_kernel<8> + redo_list_kernel + half of the two search launches over marked PUs + "
                                     "refine_small_kernel<8>", "kernels": detail}
json.dump(t, open(tpath, "w"), indent=1)
print("\n".join(open(os.path.join(PROF, "%s_bench_kernel_trace.txt" % tag)).read().splitlines()))
