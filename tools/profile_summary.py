#!/usr/bin/env python3
"""Summaries of the rocprofv3 passes of tools/profile_round.sh for profiles/:

    python tools/profile_summary.py gpurun_out r02

  profiles/<tag>_bench_kernel_trace.txt   per (kernel, grid): calls, average / median / min / max duration; per launch
                                          class of a step, the span from the start of its first kernel to the end of its
                                          last one (what bench.py's HIP events bracket for the 8x8 class)
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


def one(pattern):
    f = glob.glob(os.path.join(root, pattern), recursive=True)
    assert f, pattern
    return f[0]


def short(name):
    m = re.match(r"(?:void )?(?:[A-Za-z0-9_]+::)*([A-Za-z0-9_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


# ---- kernel trace
rows = list(csv.DictReader(open(one("%s_trace/**/*kernel_trace.csv" % tag))))
agg = collections.defaultdict(list)
for r in rows:
    agg[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
lines = ["rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --cpu-pairs 0 --no-other-paths",
         "(set-up launches included in the call counts; durations in microseconds)", "",
         "%-44s %9s %6s %9s %9s %9s %9s" % ("kernel", "grid", "calls", "avg", "median", "min", "max")]
for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    d = sorted((e - s) / 1e3 for s, e in v)
    lines.append("%-44s %9d %6d %9.1f %9.1f %9.1f %9.1f" % (k, g, len(d), sum(d) / len(d), d[len(d) // 2], d[0], d[-1]))
# the launch classes of a step: grid = PUs * 64 threads (one PU per wave) or PUs * 16 (four 8x8 PUs per wave)
B = 32
REDO_GRID = 8192 * 64    # threads of the search launch that walks the list of marked PUs
classes = {"8x8": [("star_small_kernel<8>", PUS_8 * B * 16), ("redo_list_kernel", (PUS_8 * B + 2047) // 2048 * 256),
                   ("refine_small_kernel<8>", PUS_8 * B * 16)],
           "16x16": [("star_small_kernel<16>", 8040 * B * 64), ("redo_list_kernel", (8040 * B + 2047) // 2048 * 256),
                     ("refine_small_kernel<16>", 8040 * B * 64)],
           "64+32": [("search_kernel<unsigned char, false>", 2460 * B * 64)]}
lines += ["", "launch classes of the timed steps (last 5 calls of each kernel): span = end of the class's last kernel - start "
          "of its first, per step"]
spans = {}
for cname, ks in classes.items():
    seqs = [sorted(agg.get(k, []))[-5:] for k in ks]
    if not all(len(s) == 5 for s in seqs):
        lines.append("%-6s kernels not found: %s" % (cname, [k for k, s in zip(ks, seqs) if len(s) != 5]))
        continue
    sp = [(seqs[-1][i][1] - seqs[0][i][0]) / 1e6 for i in range(5)]
    ksum = [sum((s[i][1] - s[i][0]) for s in seqs) / 1e6 for i in range(5)]
    spans[cname] = sum(sp) / 5
    lines.append("%-6s span avg %.3f ms (min %.3f max %.3f); sum of its kernels' durations avg %.3f ms; kernels: %s"
                 % (cname, sum(sp) / 5, min(sp), max(sp), sum(ksum) / 5,
                    ", ".join("%s %.3f" % (k[0], sum((e - s) for s, e in q) / 5e6) for k, q in zip(ks, seqs))))
la = sorted(agg.get(("lookahead_band_kernel<unsigned char>", 0), []))
open(os.path.join(PROF, "%s_bench_kernel_trace.txt" % tag), "w").write("\n".join(lines) + "\n")

# ---- SQ counters
rows = list(csv.DictReader(open(one("%s_sq/**/*counter_collection.csv" % tag))))
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    cnt[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]] += float(r["Counter_Value"])
lines = ["rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES "
         "SQ_WAIT_INST_ANY -- python3 bench.py --steps 1 --warmup 0 --pairs 2 --cpu-pairs 0 --no-other-paths",
         "(totals over the run: one step of 2 frame pairs; kernels of the set-up included under their own grids)", ""]
for (k, g), v in sorted(cnt.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    w = v.get("SQ_WAVES", 0)
    lines.append("%s  grid %d" % (k, g))
    for c, x in sorted(v.items()):
        lines.append("   %-22s %12.4g   per wave %10.1f" % (c, x, x / w if w else float("nan")))
open(os.path.join(PROF, "%s_search_sq_counters.txt" % tag), "w").write("\n".join(lines) + "\n")


def per_pair(k, g):
    v = cnt.get((k, g))
    if not v:
        return None
    return {"valu": v["SQ_INSTS_VALU"] / PAIRS_PMC, "salu": v["SQ_INSTS_SALU"] / PAIRS_PMC, "lds": v["SQ_INSTS_LDS"] / PAIRS_PMC,
            "waves": v["SQ_WAVES"] / PAIRS_PMC, "wave_cycles": v["SQ_WAVE_CYCLES"] / PAIRS_PMC}


timed = {"star_small_kernel<8>": ("star_small_kernel<8>", PUS_8 * PAIRS_PMC * 16),
         "search_kernel (marked 8x8 PUs)": ("search_kernel<unsigned char, false>", PUS_8 * PAIRS_PMC * 64),
         "refine_small_kernel<8>": ("refine_small_kernel<8>", PUS_8 * PAIRS_PMC * 16)}
others = {"star_small_kernel<16>": ("star_small_kernel<16>", 8040 * PAIRS_PMC * 64),
          "search_kernel (marked 16x16 PUs)": ("search_kernel<unsigned char, false>", 8040 * PAIRS_PMC * 64),
          "refine_small_kernel<16>": ("refine_small_kernel<16>", 8040 * PAIRS_PMC * 64),
          "search_kernel (64x64 + 32x32 PUs)": ("search_kernel<unsigned char, false>", 2460 * PAIRS_PMC * 64)}
for (k, g) in cnt:
    if k.startswith("lookahead_band_kernel"):
        others["lookahead_band_kernel"] = (k, g)
kern = {n: per_pair(*kg) for n, kg in {**timed, **others}.items()}
kern = {n: v for n, v in kern.items() if v}
json.dump({"source": "profiles/%s_search_sq_counters.txt (rocprofv3 --pmc SQ_*, bench.py --pairs 2 --steps 1 --warmup 0)" % tag,
           "simds": 1024, "clock_hz": 2.4e9, "pus_per_pair": PUS_8, "timed_kernels": list(timed),
           "essential_wave_instr_per_pu": {"value": 175, "how": "an 8x8 STAR / subme 5 search visits about 21 full-pel and 18 sub-pel "
                                           "candidates; at four PUs per wave (16 lanes x 4 pixels per PU) a SAD of 64 pixels is 1 "
                                           "v_sad_u8 + 4 cross-lane adds, an interpolated candidate about 8 multiply-adds per "
                                           "filter pass and a packed Hadamard of about 20: 150-200 wave instructions per PU "
                                           "before any control flow"},
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
for n, kg in {**timed, **others}.items():
    v = tr.get(kg)
    if v:
        b = (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 / PAIRS_PMC
        detail[n] = {"FETCH_SIZE_KB": v.get("FETCH_SIZE", 0) / PAIRS_PMC, "WRITE_SIZE_KB": v.get("WRITE_SIZE", 0) / PAIRS_PMC,
                     "hbm_bytes_per_pair": b}
        if n in timed:
            total += b
t["search_kernel_hbm_bytes_per_launch"]["pairs3"] = total
t["pairs3_detail"] = {"unit": "per frame pair (the step of the counter run held 2 pairs; bench.py multiplies by its pairs per step)",
                      "correction": "HBM bytes = 2 x FETCH_SIZE KB (gfx950 counts a 128-byte request as 64 bytes) + WRITE_SIZE KB, "
                                    "separate --pmc passes (tools/profile_round.sh)", "kernels": detail}
json.dump(t, open(tpath, "w"), indent=1)
print("\n".join(open(os.path.join(PROF, "%s_bench_kernel_trace.txt" % tag)).read().splitlines()[:40]))
