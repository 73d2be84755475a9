"""GPU: the EncoderPrimitives mirror (x265amd_setup_primitives) checked the way the
reference's TestBench does (test/pixelharness.cpp, test/ipfilterharness.cpp): every
partition, sliding unaligned reference offsets, odd strides -- against the oracle and
the reference-generated golden values."""
import ctypes as C

import numpy as np
import pytest

from x265_amd.engine import PARTITIONS
import golden_util as G

pytestmark = pytest.mark.gpu


def _ptr(a, off=0):
    return C.c_void_p(a.ctypes.data + off * a.itemsize)


@pytest.mark.parametrize("depth", [8, 10])
def test_table_vs_golden_and_oracle(request, oracle, depth):
    e = request.getfixturevalue("engine8" if depth == 8 else "engine10")
    P = e.primitives()
    g = G.prim_golden()[str(depth)]
    a, b = G.prim_inputs(depth)
    dt = a.dtype
    for pi, (w, h) in enumerate(PARTITIONS):
        rows = g["parts"]["%dx%d" % (w, h)]
        for (ao, bo), ex in zip(G.prim_offsets(pi), rows):
            fenc = np.ascontiguousarray(a.reshape(160, 256)[:64, :64]).reshape(-1)
            offs = [bo + k * 3 + k * 256 for k in range(4)]
            so = bo + 10 * 256 + 10
            assert P.sad[pi](_ptr(a, ao), 256, _ptr(b, bo), 251) == ex["sad"]
            assert P.satd[pi](_ptr(a, ao), 256, _ptr(b, bo), 251) == ex["satd"]
            assert P.sa8d_inter[pi](_ptr(a, ao), 256, _ptr(b, bo), 251) == ex["sa8d"]
            r3 = (C.c_int32 * 3)()
            P.sad_x3[pi](_ptr(fenc), _ptr(b, offs[0]), _ptr(b, offs[1]), _ptr(b, offs[2]), 251, r3)
            assert list(r3) == ex["x3"]
            r4 = (C.c_int32 * 4)()
            P.sad_x4[pi](_ptr(fenc), _ptr(b, offs[0]), _ptr(b, offs[1]), _ptr(b, offs[2]), _ptr(b, offs[3]), 251, r4)
            assert list(r4) == ex["x4"]
            d = np.zeros((h, w), dt)
            P.pixelavg_pp[pi](_ptr(d), w, _ptr(a, ao), 256, _ptr(b, bo), 251, 32)
            assert G.sha(d) == ex["avg"]
            for ci in (1, 2, 3):
                d = np.zeros((h, w), dt)
                P.luma_hpp[pi](_ptr(b, so), 256, _ptr(d), w, ci)
                assert G.sha(d) == ex["hpp%d" % ci]
                d = np.zeros((h, w), dt)
                P.luma_vpp[pi](_ptr(b, so), 256, _ptr(d), w, ci)
                assert G.sha(d) == ex["vpp%d" % ci]
                d = np.zeros((h, w), np.int16)
                P.luma_hps[pi](_ptr(b, so), 256, _ptr(d), w, ci, 0)
                assert G.sha(d) == ex["hps%d" % ci]
                im = np.zeros((h + 7, w), np.int16)
                P.luma_hps[pi](_ptr(b, so), 256, _ptr(im), w, ci, 1)
                assert G.sha(im) == ex["hpsx%d" % ci]
                d = np.zeros((h, w), dt)
                P.luma_vsp[pi](_ptr(im, 3 * w), w, _ptr(d), w, ci)
                assert G.sha(d) == ex["vsp%d" % ci]
    outs = [np.zeros((70, 100), dt) for _ in range(4)]
    P.frame_init_lowres_core(_ptr(a), _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]), _ptr(outs[3]), 256, 100, 100, 70)
    assert [G.sha(o) for o in outs] == g["lowres"]


def test_sliding_offsets_like_pixelharness(engine8, oracle):
    """check_pixelcmp slides fref through the buffer (pixelharness.cpp:88-103); x3/x4 use stride
    FENC_STRIDE - 5 (:139-175)."""
    P = engine8.primitives()
    rng = np.random.default_rng(77)
    pbuf1 = rng.integers(0, 256, 64 * 64 * 32).astype(np.uint8)
    pbuf2 = rng.integers(0, 256, 64 * 64 * 32).astype(np.uint8)
    for pi, (w, h) in enumerate(PARTITIONS):
        j = 0
        for _ in range(8):
            assert P.sad[pi](_ptr(pbuf1), 64, _ptr(pbuf2, j), 64) == oracle.sad(8, pbuf1, 0, 64, pbuf2, j, 64, w, h)
            assert P.satd[pi](_ptr(pbuf1), 64, _ptr(pbuf2, j), 64) == oracle.satd(8, pbuf1, 0, 64, pbuf2, j, 64, w, h)
            offs = [j, j + 1, j + 2, j + 3]
            r4 = (C.c_int32 * 4)()
            P.sad_x4[pi](_ptr(pbuf1), _ptr(pbuf2, offs[0]), _ptr(pbuf2, offs[1]), _ptr(pbuf2, offs[2]), _ptr(pbuf2, offs[3]), 59, r4)
            assert list(r4) == oracle.sad_x4(8, pbuf1, pbuf2, offs, 59, w, h)
            j += 32 + 1
