"""GPU: the two workloads that split inside one picture / one lookahead window (x265_amd.sharded), with the ranks of a
world played one after the other on the one GPU: the assembled records equal the unsharded run and the oracle."""
import numpy as np
import pytest

from x265_amd import sharded
from x265_amd.engine import FULL, HEX
from x265_amd.synth import pad_plane
from util import host_lowres_planes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,ctu", [(1, True), (3, True), (4, True), (3, False)])
def test_esa_bands_equal_the_unsharded_picture(engine8, oracle, world, ctu):
    """ctu: the bands' PUs through the scan per CTU (one launch over all of a rank's bands, records picked into the rank's
    record order on the device), or through one launch per PU size."""
    W, H, merange, subme, qp = 640, 384, 20, 5, 30
    gathered, first = [], None
    for rank in range(world):
        wl = sharded.EsaPicture(engine8, rank, world, width=W, height=H, merange=merange, subme=subme, qp=qp,
                                seed=17, margin=160, ctu=ctu)
        wl.validate()
        wl.step()
        assert wl.records_bytes == 8 * max(len(wl.rank_index(q)) for q in range(world))
        gathered.append(wl.records())
        if first is None:
            first = wl
        else:
            wl.close()
    mv, cost = first.assemble(gathered)
    cur, org, stride = pad_plane(first.cur, 160)
    pc = engine8.upload_plane(cur, W, H, 160, 160, stride)
    rng = np.random.default_rng(3)
    for r in range(first.nrefs):
        ref, _, _ = pad_plane(first.refs[r], 160)
        pr = engine8.upload_plane(ref, W, H, 160, 160, stride)
        res = engine8.search_batch(pc, pr, first.jobs[r], qp, FULL, subme)
        assert np.array_equal(res["mv"], mv[r]) and np.array_equal(res["cost"], cost[r]), (world, r)
        engine8.free_plane(pr)
        pick = rng.choice(len(first.jobs[r]), 40, replace=False)
        exp = oracle.me_search_batch(oracle.make_ctx(8, cur, org, stride, [ref], org, stride, False, qp, FULL, subme),
                                     first.jobs[r][pick])
        assert np.array_equal(exp["mv"], mv[r][pick]) and np.array_equal(exp["cost"], cost[r][pick]), (world, r)
    engine8.free_plane(pc)
    first.close()


@pytest.mark.parametrize("world", [1, 2, 5])
def test_lookahead_window_fields_dealt_out(engine8, oracle, world):
    W, H = 416, 240
    gathered, first = [], None
    for rank in range(world):
        wl = sharded.LookaheadWindow(engine8, rank, world, width=W, height=H, nframes=5, max_distance=3, seed=4)
        wl.step()
        gathered.append(wl.records())
        assert wl.pixel_sads() > 0 or wl.n == 0
        if first is None:
            first = wl
        else:
            wl.close()
    fields = first.assemble(gathered)
    assert len(first.all_fields) == 18 and fields.shape == (18, first.ncu, 2)
    low = {f: host_lowres_planes(oracle, 8, first.clip[f], 64) for f in range(5)}
    for n, (b, r) in enumerate(first.all_fields):
        pa, pb = low[b], low[r]
        ctx = oracle.make_ctx(8, pa[0][0], pa[1], pa[2], pb[0], pa[1], pa[2], True, first.qp, HEX, 1)
        mv, _ = oracle.lookahead_field(ctx, first.wcu, first.hcu, 16)
        assert np.array_equal(mv, fields[n]), (world, b, r)
    first.close()
