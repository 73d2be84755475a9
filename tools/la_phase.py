#!/usr/bin/env python3
"""Where a CU step of lookahead_band_kernel spends its cycles: runs lookahead launches on a -DX265AMD_PHASE_PROFILE build
of the library (X265AMD_LIB names it) and prints, per launch, the average shader-clock cycles a wave spends in every
phase of a step.   usage: X265AMD_LIB=alt/phase.so python tools/la_phase.py > profiles/rNN_lookahead_phase_cycles.txt"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x265_amd.engine import Engine, HEX
from x265_amd.synth import make_clip, pad_plane, lowres_dims

PHASES = ["neighbours' MVs (poll the band below / LDS)", "predictor, bounds, source tile, clipping", "start-up candidates (<= 7, sub-pel SAD)",
          "hexagon: 6 points", "hexagon walk: 3 points per iteration", "closing square: 8 points", "sub-pel window staged in LDS",
          "half-pel round (SAD)", "quarter-pel round (SATD)", "publish, step bookkeeping, wave_sync"]
NW = 8192


def run(e, name, W, H, nframes, nfields):
    clip = make_clip(W, H, nframes, seed=77)
    low = []
    for f in clip:
        buf, org, stride = pad_plane(f, 64)
        p = e.upload_plane(buf, W, H, 64, 64, stride)
        low.append(e.lowres_init(p))
        e.free_plane(p)
    lw, lh, wcu, hcu = lowres_dims(W, H)
    pairs = [(i, j) for i in range(nframes) for j in range(nframes) if i != j][:nfields]
    fields = [(low[i][0], low[j]) for i, j in pairs]
    d_f = e.fields_upload(fields, wcu, hcu, HEX, 16)
    ncu = wcu * hcu
    d_mv, d_cost = e.device_alloc(len(fields) * ncu * 4), e.device_alloc(len(fields) * ncu * 4)
    d_ph = e.device_alloc(NW * 16 * 8)
    e.L.x265amd_debug_phase_buffer(C.c_void_p(None))
    e.lookahead_fields_dev(d_f, len(fields), wcu, hcu, 12, HEX, 1, 16, d_mv, d_cost)       # warm-up, no counters
    e.sync()
    e.upload(d_ph, np.zeros(NW * 16, np.int64))
    e.L.x265amd_debug_phase_buffer(C.c_void_p(d_ph))
    e.lookahead_fields_dev(d_f, len(fields), wcu, hcu, 12, HEX, 1, 16, d_mv, d_cost)
    e.sync()
    e.L.x265amd_debug_phase_buffer(C.c_void_p(None))
    ph = e.download(d_ph, (NW, 16), np.int64)
    ph = ph[ph[:, 10] > 0]
    steps, walks = int(ph[:, 10].sum()), int(ph[:, 11].sum())
    tot = ph[:, :10].sum()
    print("%s: %d fields of %d x %d CUs, %d waves with work, %d wave steps (four CU rows each), %.2f walk iterations per step"
          % (name, len(fields), wcu, hcu, len(ph), steps, walks / steps))
    print("  %-50s %10s %7s" % ("phase", "cycles", "share"))
    for i, n in enumerate(PHASES):
        print("  %-50s %10.0f %6.1f%%" % (n, ph[:, i].sum() / steps, 100.0 * ph[:, i].sum() / tot))
    print("  %-50s %10.0f" % ("a step", tot / steps))
    for d in (d_f, d_mv, d_cost, d_ph):
        e.device_free(d)


def main():
    e = Engine(0, 8)
    if not hasattr(e.L, "x265amd_debug_phase_buffer"):
        sys.exit("the library is no -DX265AMD_PHASE_PROFILE build (set X265AMD_LIB)")
    e.L.x265amd_debug_phase_buffer.restype = None
    print("lookahead_band_kernel<uint8_t, 4>, HEX subme 1 merange 16: shader-clock cycles (s_memtime) per wave and CU step, summed in "
          "registers between marks\nand stored once per wave; a wave's step holds the steps of its four CU rows, which diverge "
          "(a phase lasts as long as its slowest row)\n")
    os.environ["X265AMD_LOOKAHEAD_ROWS"] = "4"
    run(e, "1080p, chip mostly idle (8 fields, 136 waves)", 1920, 1080, 4, 8)
    run(e, "1080p, the headline's launch (32 fields, 544 waves)", 1920, 1080, 7, 32)
    run(e, "8K, 8 fields (544 waves)", 7680, 4320, 4, 8)
    run(e, "8K, chip full (90 fields, 6120 band tickets for 4096 places)", 7680, 4320, 10, 90)
    e.close()


if __name__ == "__main__":
    main()
