"""GPU: x265amd_lookahead_fields_static, the entry point that allocates nothing (its descriptors, result arrays and
the row ticket of the lookahead kernel live in a static device array), against x265amd_lookahead_fields on the same
inputs: bit-exact MVs, costs and evaluation counts, two calls in a row, the second with a smaller input."""
import numpy as np
import pytest

from x265_amd.engine import Engine, EngineError
from x265_amd.synth import make_clip, pad_plane

pytestmark = pytest.mark.gpu


def _device_lowres(e, img, margin=64):
    buf, org, stride = pad_plane(img, margin)
    p = e.upload_plane(buf, img.shape[1], img.shape[0], margin, margin, stride)
    low = e.lowres_init(p)
    e.free_plane(p)
    return low


def _same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


def test_static_lookahead_matches_allocating_entry_point():
    e = Engine(0, 8)       # a context of its own: the test must hold alone in a fresh process and after all others
    try:
        clip = make_clip(416, 240, 4, seed=61, depth=8)
        L = [_device_lowres(e, f) for f in clip]
        big = [(L[i][0], L[j]) for i in range(4) for j in range(4) if i != j]          # 12 fields of 26 x 15 CUs
        small_clip = make_clip(200, 120, 2, seed=62, depth=8)
        S = [_device_lowres(e, f) for f in small_clip]
        small = [(S[1][0], S[0])]                                                      # one field of 13 x 8 CUs
        with pytest.raises(EngineError):
            e.lookahead_fields_static(big, 26, 15, 13)         # no table for qp 13 yet, and building one allocates
        want_big = e.lookahead_fields(big, 26, 15, 12, 1, 1, 16, want_stats=True)
        want_small = e.lookahead_fields(small, 13, 8, 12, 1, 1, 16, want_stats=True)
        want_star = e.lookahead_fields(small, 13, 8, 12, 3, 2, 16, want_stats=True)
        # two calls in a row on the shared array, the second smaller; then the first again (stale contents of both)
        got_big = e.lookahead_fields_static(big, 26, 15, 12, 1, 1, 16, want_stats=True)
        got_small = e.lookahead_fields_static(small, 13, 8, 12, 1, 1, 16, want_stats=True)
        assert _same(got_big, want_big)
        assert _same(got_small, want_small)
        assert _same(e.lookahead_fields_static(small, 13, 8, 12, 3, 2, 16, want_stats=True), want_star)
        assert _same(e.lookahead_fields_static(big, 26, 15, 12, 1, 1, 16, want_stats=True), want_big)
        mv, cost, none = e.lookahead_fields_static(small, 13, 8, 12, 1, 1, 16)
        assert none is None and np.array_equal(mv, want_small[0]) and np.array_equal(cost, want_small[1])
        # the defining property: the calls take no device memory. Everything is warm by now (kernels loaded, cost table
        # built); the free memory the driver reports must not move over a run of calls, larger after smaller included
        e.sync()
        free0 = e.mem_info()[0]
        for _ in range(3):
            e.lookahead_fields_static(small, 13, 8, 12, 1, 1, 16, want_stats=True)
            e.lookahead_fields_static(big, 26, 15, 12, 1, 1, 16, want_stats=True)
            e.lookahead_fields_static(small, 13, 8, 12, 3, 2, 16)
        assert e.mem_info()[0] == free0
        # a second context on the same GPU uses the same array (one call at a time)
        e2 = Engine(0, 8)
        e2.set_qp(12)
        assert _same(e2.lookahead_fields_static(small, 13, 8, 12, 1, 1, 16, want_stats=True), want_small)
        # ... and a call that finds the array in use is refused (X265AMD_EBUSY), whatever context it comes from
        import threading
        refused, done = [], []

        def hammer(eng, fields, w, h):
            for _ in range(40):
                try:
                    got = eng.lookahead_fields_static(fields, w, h, 12, 1, 1, 16, want_stats=True)
                    done.append(_same(got, want_big if fields is big else want_small))
                except EngineError as ex:
                    refused.append(str(ex))
        th = [threading.Thread(target=hammer, args=(e, big, 26, 15)), threading.Thread(target=hammer, args=(e2, small, 13, 8))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert all(done) and len(done) + len(refused) == 80      # every call either ran alone and is right, or was refused
        assert all("in use" in r or "using" in r for r in refused)
        e2.close()
        # limits: more fields than the entry point takes, and fields that do not fit the array
        with pytest.raises(EngineError):
            e.lookahead_fields_static(big * 22, 26, 15, 12)                            # 264 > 256 fields
        huge = make_clip(7680, 4320, 2, seed=63, depth=8)
        H = [_device_lowres(e, f) for f in huge]
        with pytest.raises(EngineError):
            e.lookahead_fields_static([(H[1][0], H[0])] * 9, 480, 270, 12)              # 9 x 1.04 MB > 8 MB
        one = e.lookahead_fields_static([(H[1][0], H[0])], 480, 270, 12, want_stats=True)
        assert _same(one, e.lookahead_fields([(H[1][0], H[0])], 480, 270, 12, want_stats=True))
        assert len(e.lookahead_fields_static([], 26, 15, 12)[0]) == 0
    finally:
        e.close()
