"""Real encodes by the REFERENCE encoder with its motion estimation answered by the HIP engine through the C ABI.

oracle/_ref/x265_served is the reference's own encoder + CLI (compiled from where its sources lie by oracle/Makefile,
target `served`) with the two MotionEstimate::motionEstimate overloads, MotionReference::init and Lowres::init routed
through oracle/enc_serve.cpp, which answers them with x265amd_rk_ctu_batch / x265amd_lookahead_fields /
x265amd_rk_ds_plane / x265amd_lowres_init of x265_amd/libx265_amd_me.so (TEncSearch.cpp:3821, slicetype.cpp:1468,
frameencoder.cpp:625-627, slicetype.cpp:93-110). The bitstreams must be the ones the CPU-only reference wrote: their MD5s
were recorded in the build container (BASELINE configs[0]; tests/golden/callstream.npz carries the same two).

  * gpu mode:   every call answered by the engine; bitstream MD5 of the 30-frame encode; no call fell through
  * check mode: every call answered by the engine AND by the reference's own function, the binary aborts on the first
                difference in MV, cost or the C-model globals (8 frames with B-frames: both lookahead directions)
"""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from x265_amd.synth import baseline_clip, write_yuv420
import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SERVED = os.path.join(ROOT, "oracle", "_ref", "x265_served")
W, H = 416, 240


def encode(tmp_path, name, frames, extra, mode):
    if not os.path.exists(SERVED):
        pytest.fail("oracle/_ref/x265_served is missing: build it in the container that has the reference "
                    "(make -C oracle served, or __graft_entry__.build())")
    yuv = str(tmp_path / "clip.yuv")
    write_yuv420(yuv, baseline_clip(W, H, 30, 1234)[:frames])
    hevc, stats = str(tmp_path / (name + ".hevc")), str(tmp_path / (name + ".json"))
    env = dict(os.environ, X265AMD_SERVE=mode, X265AMD_SERVE_STATS=stats)
    cmd = [SERVED, "--input", yuv, "--input-res", "%dx%d" % (W, H), "--fps", "30", "-f", str(frames), "--preset", "medium",
           "--me", "hex", "--threads", "1", "--no-progress"] + extra + ["-o", hevc]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=110)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-2000:]
    return hashlib.md5(open(hevc, "rb").read()).hexdigest(), json.load(open(stats))


@pytest.mark.gpu
def test_encode_30_frames_served_by_the_engine(tmp_path):
    z = np.load(G.GOLD + "/callstream.npz")
    md5, st = encode(tmp_path, "c0", 30, [], "gpu")
    assert md5 == z["c0_bitstream_md5"].tobytes().hex() == "f2fe9f37c8c7e18d4f4610df4a25c48f"
    assert st["mode"] == "gpu" and st["fallback_calls"] == 0
    # 7 x 4 CTUs of 29 P-frames, one launch each; one lowres field per P-frame
    assert st["rk_ctu_launches"] == 28 * 29 and st["lookahead_field_launches"] == 29
    assert st["rk_calls"] > 28 * 29 * 40 and st["lookahead_calls"] == 29 * 26 * 15


@pytest.mark.gpu
def test_encode_with_b_frames_checked_call_by_call(tmp_path):
    z = np.load(G.GOLD + "/callstream.npz")
    md5, st = encode(tmp_path, "lb", 8, ["--bframes", "3", "--rc-lookahead", "10", "--b-adapt", "2"], "check")
    assert md5 == z["lb_bitstream_md5"].tobytes().hex() == "435ffbe83cfe979fec8e147e5fb9b070"
    assert st["mode"] == "check" and st["rk_ctu_launches"] > 0 and st["lookahead_field_launches"] > 0
    # the lookahead of this run searches both directions; fields whose reference was weighted would fall through
    assert st["fallback_calls"] == 0
