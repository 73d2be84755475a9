"""Per-launch summary of the three rocprofv3 --pmc passes of tools/mc_pmc.sh:
python profiles/mc_pmc_summary.py <sq counter_collection.csv> <fetch csv> <write csv> > profiles/r01_mc_pmc.txt"""
import collections
import csv
import sys


def load(p):
    out = collections.OrderedDict()
    for r in csv.DictReader(open(p)):
        if "mc_kernel" in r["Kernel_Name"]:
            out.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return [out[k] for k in sorted(out)]


sq, fe, wr = (load(p) for p in sys.argv[1:4])
names = ["8x8 uni", "8x8 bi", "16x16 uni", "16x16 bi", "32x32 uni", "32x32 bi", "64x64 uni", "64x64 bi", "mixed uni", "mixed bi"]
print("mc_kernel<uint8_t>, 3840x2160, luma + chroma, every PU of the picture (tools/mc_pmc.sh: three separate rocprofv3 --pmc passes of")
print("tools/mc_bench.py --width 3840 --height 2160 --no-check --reps 1; second launch of each configuration).")
print("SQ_WAVE_CYCLES and SQ_WAIT_INST_ANY count quad-cycles. HBM bytes = 2 x FETCH_SIZE (gfx950 counts 128-byte requests as 64) + WRITE_SIZE.")
print()
for i, n in enumerate(names):
    v = sq[2 * i + 1]
    w = v["SQ_WAVES"]
    f, o = fe[2 * i + 1]["FETCH_SIZE"] * 2 * 1024, wr[2 * i + 1]["WRITE_SIZE"] * 1024
    print("%-10s waves %5d  VALU/wave %5.0f  SALU/wave %4.0f  VMEM reads/wave %3.0f  wave lifetime %6.0f cycles  waiting %6.0f cycles  "
          "HBM %6.1f MB (fetch %5.1f + write %4.1f)"
          % (n, w, v["SQ_INSTS_VALU"] / w, v["SQ_INSTS_SALU"] / w, v["SQ_INSTS_VMEM_RD"] / w, 4 * v["SQ_WAVE_CYCLES"] / w,
             4 * v["SQ_WAIT_INST_ANY"] / w, (f + o) / 1e6, f / 1e6, o / 1e6))
