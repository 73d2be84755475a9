"""CPU: host-side logic -- synthetic clips, job builders, sharding, and the N>1 gather over gloo."""
import os
import subprocess
import sys

import numpy as np

from x265_amd import sharding, workload
from x265_amd.engine import JOB_DTYPE, RESULT_DTYPE
from x265_amd.synth import make_clip, pad_plane, lowres_dims

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_clip_is_deterministic_and_translating():
    a = make_clip(128, 96, 3, seed=9)
    b = make_clip(128, 96, 3, seed=9)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # frame i is the base texture moved by (3i, 2i): frame1[y, x] ~ frame0[y + 2, x + 3]
    d = np.abs(a[1][:-2, :-3].astype(int) - a[0][2:, 3:].astype(int)).mean()
    assert d < 4.0
    assert make_clip(64, 64, 1, depth=10)[0].max() > 255


def test_pad_plane_replicates_edges():
    img = np.arange(12, dtype=np.uint8).reshape(3, 4)
    buf, org, stride = pad_plane(img, 2, 3)
    p = buf.reshape(-1, stride)
    assert stride == 8 and org == 3 * 8 + 2
    assert (p[:3, :2] == 0).all() and (p[-1, -2:] == 11).all() and p[3, 2] == 0 and p[5, 5] == 11


def test_lowres_dims():
    assert lowres_dims(416, 240) == (208, 120, 26, 15)
    assert lowres_dims(1920, 1080) == (960, 544, 120, 68)
    assert lowres_dims(3840, 2160) == (1920, 1080, 240, 135)


def test_square_pu_jobs_cover_the_picture():
    wcu, hcu = 120, 68
    mvs = np.zeros((wcu * hcu, 2), np.int16)
    mvs[:, 0] = 6
    jobs = workload.square_pu_jobs(1920, 1080, mvs, wcu, hcu, 57)
    assert len(jobs) == 30 * 16 + 60 * 33 + 120 * 67 + 240 * 135
    assert (jobs["qmvp"][:, 0] == 12).all()
    # windows stay inside a 128-pixel margin incl. the 8-tap apron
    assert (jobs["pu_x"] + jobs["mvmin"][:, 0] - 8 >= -128).all()
    assert (jobs["pu_x"] + jobs["w"] + jobs["mvmax"][:, 0] + 8 <= 1920 + 128).all()
    res = np.zeros(len(jobs), RESULT_DTYPE)
    res["sad_calls"] = 2
    res["subpel_calls"] = 1
    assert workload.pixel_sads(jobs, res) == 3 * int((jobs["w"].astype(int) * jobs["h"]).sum())


def test_algorithmic_bytes_formula():
    j = np.zeros(1, JOB_DTYPE)
    j["w"], j["h"] = 16, 16
    j["mvmin"], j["mvmax"] = (-57, -57), (57, 57)
    assert workload.algorithmic_bytes(j) == 16 * 16 + (16 + 114 + 7) ** 2 + 8
    assert workload.algorithmic_bytes(j, 2) == 2 * (16 * 16 + (16 + 114 + 7) ** 2) + 8


def test_shard_range_partitions_everything():
    for n in (0, 1, 7, 8, 9, 17, 8160):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi = sharding.shard_range(n, r, world)
                got += list(range(lo, hi))
            assert got == list(range(n))


def _random_field_jobs(width, height, merange, seed, nrefs=3):
    lw, lh, wcu, hcu = lowres_dims(width, height)
    rng = np.random.default_rng(seed)
    return [workload.square_pu_jobs(width, height, rng.integers(-40, 41, (wcu * hcu, 2)).astype(np.int16), wcu, hcu, merange)
            for _ in range(nrefs)]


def test_band_units_cover_the_picture_and_windows_fit():
    """(reference, CTU row) units: every job of every reference belongs to exactly one rank; a band's cropped planes
    hold every row its jobs' search windows (plus the sub-pel apron) can read, at the rows the shifted jobs name."""
    W, H, M, merange = 1920, 1080, 256, 57
    jobs = _random_field_jobs(W, H, merange, 11)
    nrows = (H + 63) // 64
    img = np.random.default_rng(2).integers(0, 256, (H, W)).astype(np.uint8)
    buf, org, stride = pad_plane(img, M)
    rows = buf.reshape(-1, stride)
    for world in (1, 2, 3, 8):
        seen = [np.zeros(len(j), int) for j in jobs]
        for rank in range(world):
            pieces = sharding.row_units(len(jobs), nrows, rank, world)
            index = sharding.band_record_index(jobs, nrows, rank, world)
            for r in range(len(jobs)):
                np.add.at(seen[r], index[index[:, 0] == r, 1], 1)
            sizes = np.array([jobs[r]["w"][i] for r, i in index])
            assert (np.diff(sizes.astype(int)) <= 0).all()          # one launch per size, largest first
            for (r, lo, hi) in pieces:
                idx, top, bottom = sharding.band_jobs(jobs[r], lo, hi)
                crop, hb, halo, y0 = sharding.crop_band(buf, stride, M, H, lo, hi, top, bottom)
                assert y0 == lo * 64 and hb == min(hi * 64, H) - y0 and crop.size == (hb + 2 * halo) * stride
                j = jobs[r][idx]
                y = j["pu_y"].astype(int) - y0
                assert (y + j["mvmin"][:, 1] - 8 >= -halo).all() and (y + j["h"] + j["mvmax"][:, 1] + 8 <= hb + halo).all()
                assert np.array_equal(crop.reshape(-1, stride), rows[M + y0 - halo:M + y0 + hb + halo])
        assert all((s_ == 1).all() for s_ in seen), world
    # a band whose windows leave the padded plane is refused, not cropped short
    try:
        sharding.crop_band(buf, stride, M, H, 0, 2, -M - 1, 200)
        assert False
    except ValueError:
        pass
    assert sharding.round_robin(10, 1, 4) == [1, 5, 9] and sharding.padded_count(10, 4) == 3
    assert sorted(sum((sharding.round_robin(90, r, 8) for r in range(8)), [])) == list(range(90))


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from x265_amd import sharding
from x265_amd.engine import RESULT_DTYPE
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
n_units = 9
lo, hi = sharding.shard_range(n_units, rank, 2)
per = 5                                    # pad to the largest shard so gather shapes match
rec = np.zeros(per * 4, RESULT_DTYPE)
for u in range(lo, hi):
    rec["cost"][(u - lo) * 4:(u - lo + 1) * 4] = 1000 + u
    rec["mv"][(u - lo) * 4:(u - lo + 1) * 4] = (u, -u)
t = torch.from_numpy(rec.view(np.uint8).copy())
got = sharding.gather_records(t, dist, dst=0)
if rank == 0:
    allrec = np.concatenate([g.numpy().view(RESULT_DTYPE) for g in got])
    units = []
    for r in range(2):
        l, h = sharding.shard_range(n_units, r, 2)
        units += [int(allrec["cost"][r * per * 4 + k * 4]) - 1000 for k in range(h - l)]
    assert units == list(range(n_units)), units
    assert tuple(allrec["mv"][per * 4]) == (5, -5)
    print("GATHER_OK")
else:
    assert got is None
# the overlapped form: two gathers in flight into receive buffers that are used again
bufs = None
for step in range(3):
    a = t + step
    b = t * 0 + rank + step
    ga, wa = sharding.gather_records(a, dist, dst=0, bufs=bufs, async_op=True)
    gb, wb = sharding.gather_records(b, dist, dst=0, async_op=True)
    wa.wait(); wb.wait()
    if rank == 0:
        assert bufs is None or all(x is y for x, y in zip(ga, bufs))
        bufs = ga
        assert torch.equal(ga[0], a) and int(gb[1][0]) == 1 + step and int(gb[0][0]) == step
    else:
        assert ga is None and gb is None
if rank == 0:
    print("ASYNC_GATHER_OK")
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_gather_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK" in outs[0] and "ASYNC_GATHER_OK" in outs[0]


def test_tile_plan_properties():
    """The launch plan of the tile-per-lane kernels (host code inside the library, no GPU needed): every record's tiles are
    consecutive, records follow each other without overlap, aligned records start where the cross-lane steps need them,
    and each 64-tile chunk names the record that owns its first tile (or the padding in front of the next one)."""
    import ctypes as C
    from x265_amd.engine import load_library
    L = load_library()
    rng = np.random.default_rng(3)
    for align16 in (0, 1):
        for n in (0, 1, 2, 63, 64, 65, 1000):
            choices = np.array([4, 16, 64, 256]) if align16 else np.array([2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256])
            tiles = rng.choice(choices, n).astype(np.int32)
            first = np.zeros(n + 64, np.int32)
            chunk = np.zeros(int(tiles.sum()) // 64 + n // 4 + 2, np.int32)
            nch = C.c_int(-1)
            assert L.x265amd_tile_plan(C.c_void_p(tiles.ctypes.data), n, align16, C.c_void_p(first.ctypes.data),
                                       C.c_void_p(chunk.ctypes.data), C.byref(nch)) == 0
            total = int(first[n])
            assert (first[n:] == total).all() and nch.value == (total + 63) // 64
            ends = first[:n] + tiles
            assert (first[1:n] >= ends[:-1]).all() and (n == 0 or first[0] == 0) and (n == 0 or total == ends[-1])
            if align16:
                assert (first[:n] % np.minimum(tiles, 16) == 0).all() and (first[1:n] - ends[:-1] < 16).all()
            else:
                assert (first[1:n] == ends[:-1]).all()
            for k in range(nch.value):
                j = int(chunk[k])
                assert first[j] <= 64 * k < first[j + 1] if j + 1 < n else first[j] <= 64 * k < total
    bad = np.array([4, 0, 4], np.int32)
    assert L.x265amd_tile_plan(C.c_void_p(bad.ctypes.data), 3, 0, C.c_void_p(first.ctypes.data), C.c_void_p(chunk.ctypes.data), C.byref(nch)) == -1



_BAND_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
from x265_amd import sharding
from test_host_logic import _random_field_jobs
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
W, H = 640, 360
jobs = _random_field_jobs(W, H, 24, 5)          # every rank derives the same unsharded lists
nrows = (H + 63) // 64
index = sharding.band_record_index(jobs, nrows, rank, 2)
n_pad = max(len(sharding.band_record_index(jobs, nrows, q, 2)) for q in range(2))
rec = np.zeros((n_pad, 8), np.uint8)
mine = np.zeros(len(index), np.dtype([("mv", "<i2", (2,)), ("cost", "<i4")]))
mine["mv"][:, 0], mine["mv"][:, 1], mine["cost"] = index[:, 0], index[:, 1] % 1000, index[:, 0] * 100000 + index[:, 1]
rec[:len(index)] = mine.view(np.uint8).reshape(-1, 8)
got = sharding.gather_records(torch.from_numpy(rec.reshape(-1).copy()), dist, dst=0)
if rank == 0:
    mv, cost = sharding.assemble_band_records(jobs, nrows, 2, [g.numpy() for g in got])
    for r in range(len(jobs)):
        n = len(jobs[r])
        assert np.array_equal(cost[r], r * 100000 + np.arange(n)) and (mv[r][:, 0] == r).all()
        assert np.array_equal(mv[r][:, 1], np.arange(n) % 1000)
    print("BANDS_OK")
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_band_records_over_gloo(tmp_path):
    """World size 2: each rank reports the records of its (reference, CTU-row band) units, rank 0 puts the gathered
    records back into the order of the unsharded job lists."""
    script = tmp_path / "band_worker.py"
    script.write_text(_BAND_WORKER)
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "BANDS_OK" in outs[0]


def test_ctu_job_lists_for_the_exhaustive_scan():
    """workload.esa_ctu_jobs: 85 slots per 64x64 CTU, size-major; inside a CTU the 8x8 and 16x16 slots in Morton order
    (a 16x16 PU's four 8x8 blocks are consecutive, a 32x32 PU's sixteen likewise); slots the picture's edge cuts off are
    empty but still name their place; jobs of several pairs form CTUs per pair."""
    W, H = 192, 160                                 # the third CTU row is cut: 32 of its 64 rows
    wcu, hcu = (W // 2 + 7) // 8, (H // 2 + 7) // 8
    la = np.zeros((wcu * hcu, 2), np.int16)
    jobs = workload.square_pu_jobs(W, H, la, wcu, hcu, 16)
    cj, index, n = workload.esa_ctu_jobs(jobs)
    assert n == 9 and len(cj) == 85 * n and (index >= 0).sum() == len(jobs)
    have = index >= 0
    assert np.array_equal(cj[have], jobs[index[have]])
    assert (cj["w"][~have] == 0).all() and (cj["h"][~have] == 0).all()
    first = (0, 64 * n, 80 * n, 84 * n, 85 * n)
    for lvl, (s, per) in enumerate(((8, 64), (16, 16), (32, 4), (64, 1))):
        part = cj[first[lvl]:first[lvl + 1]].reshape(n, per)
        for c in range(n):
            ox0, oy0 = int(part[c, 0]["pu_x"]), int(part[c, 0]["pu_y"])
            assert ox0 % 64 == 0 and oy0 % 64 == 0
            for p in range(per):
                ox, oy, ss = workload.ctu_slot(p + (0, 64, 80, 84)[lvl])
                assert ss == s and int(part[c, p]["pu_x"]) == ox0 + ox and int(part[c, p]["pu_y"]) == oy0 + oy
                inside = ox0 + ox + s <= W and oy0 + oy + s <= H
                assert (part[c, p]["w"] == s) == inside
    # Morton order: the children of 16x16 slot k are the 8x8 slots 4 k .. 4 k + 3, of 32x32 slot k the 16x16 slots 4 k .. 4 k + 3
    for k in range(16):
        x, y, _ = workload.ctu_slot(64 + k)
        assert sorted(workload.ctu_slot(4 * k + q)[:2] for q in range(4)) == sorted((x + dx, y + dy) for dy in (0, 8) for dx in (0, 8))
    for k in range(4):
        x, y, _ = workload.ctu_slot(80 + k)
        assert sorted(workload.ctu_slot(64 + 4 * k + q)[:2] for q in range(4)) == sorted((x + dx, y + dy) for dy in (0, 16) for dx in (0, 16))
    # two pairs: CTUs per pair, every slot carrying its pair
    two = np.concatenate([jobs, jobs])
    two["reserved"][len(jobs):] = 1
    cj2, index2, n2 = workload.esa_ctu_jobs(two)
    assert n2 == 2 * n
    eight = cj2[:64 * n2].reshape(n2, 64)
    assert (eight["reserved"][:n] == 0).all() and (eight["reserved"][n:] == 1).all()


def test_partition_shape_job_lists():
    """workload.shape_jobs: both PUs of every 2NxN / Nx2N split of a picture's CU grid, inside the picture, ranges around
    the predictor of the lookahead CU under the PU's centre."""
    import numpy as np
    from x265_amd import workload
    W, H, wcu, hcu = 416, 240, 26, 15
    la = np.random.default_rng(2).integers(-20, 21, (wcu * hcu, 2)).astype(np.int16)
    for cu in (64, 32, 16, 8):
        ncu = (W // cu) * (H // cu)
        for (w, h) in ((cu, cu // 2), (cu // 2, cu)):
            jobs = workload.shape_jobs(W, H, w, h, la, wcu, hcu, 57, cu)
            assert len(jobs) == 2 * ncu and (jobs["w"] == w).all() and (jobs["h"] == h).all()
            assert (jobs["pu_x"] + w <= W).all() and (jobs["pu_y"] + h <= H).all()
            # the two PUs of a CU tile it
            first, second = jobs[:ncu], jobs[ncu:]
            assert ((second["pu_x"] - first["pu_x"] == (0 if w == cu else w)) & (second["pu_y"] - first["pu_y"] == (0 if h == cu else h))).all()
            fx, fy = jobs["qmvp"][:, 0] >> 2, jobs["qmvp"][:, 1] >> 2
            assert (jobs["mvmin"][:, 0] <= fx).all() and (fx <= jobs["mvmax"][:, 0]).all()
            assert (jobs["mvmin"][:, 1] <= fy).all() and (fy <= jobs["mvmax"][:, 1]).all()
            assert (jobs["mvmax"][:, 0] - jobs["mvmin"][:, 0] <= 2 * 57).all()
        assert len(workload.shape_jobs(W, H, cu, cu, la, wcu, hcu, 57, cu)) == ncu
