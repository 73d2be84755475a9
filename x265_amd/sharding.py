"""One process per GPU: shard independent ME units across ranks and gather the MV /
cost records back to the rank that feeds the host encoder.

The units (lowres fields; (frame, reference) job lists; CTU-row bands) never exchange
data while they run, so the only collective is the final gather of result records
(KB..MB, latency-bound on xGMI) -- RCCL when the tensors live on the GPU, gloo on CPU.
"""
import numpy as np


def shard_range(n_units, rank, world):
    """Contiguous, balanced [lo, hi) slice of n_units for `rank` (first n % world ranks get one more)."""
    q, r = divmod(n_units, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_records(local, dist=None, dst=0, bufs=None, async_op=False):
    """Gather equally-shaped record tensors to `dst`. `local` is a torch tensor (uint8 view of the
    records); returns the list of per-rank tensors on dst, None elsewhere. `bufs`: dst's receive tensors to use again
    (a list of world tensors like `local`). With async_op the call returns (tensors or None, work or None) at once and
    the caller waits on the work object before it reads the tensors or writes `local` again."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return ([local], None) if async_op else [local]
    import torch
    world = dist.get_world_size()
    if dist.get_rank() == dst:
        if bufs is None:
            bufs = [torch.empty_like(local) for _ in range(world)]
        work = dist.gather(local, bufs, dst=dst, async_op=async_op)
        return (bufs, work) if async_op else bufs
    work = dist.gather(local, None, dst=dst, async_op=async_op)
    return (None, work) if async_op else None


# ---------------------------------------------------------------- (reference, CTU-row band) units of one picture

def row_units(n_refs, n_ctu_rows, rank, world):
    """The (reference, CTU row) units of one picture in reference-major order, split contiguously over the ranks.
    Returns rank's pieces [(reference index, first CTU row, end CTU row)]: runs of rows of one reference. The split is
    legal because a search against one reference reads nothing another search writes (the reference runs them per
    reference list and per CTU row too: predInterSearch's loop over lists, and FrameEncoder's rows that wait only for
    reconstructed reference rows, frameencoder.cpp:1109-1113)."""
    lo, hi = shard_range(n_refs * n_ctu_rows, rank, world)
    pieces, u = [], lo
    while u < hi:
        r = u // n_ctu_rows
        end = min(hi, (r + 1) * n_ctu_rows)
        pieces.append((r, u - r * n_ctu_rows, end - r * n_ctu_rows))
        u = end
    return pieces


def band_jobs(jobs, row_lo, row_hi, ctu=64, apron=8):
    """The jobs whose PU lies in CTU rows [row_lo, row_hi) and the picture rows [top, bottom) of the reference those
    searches can read: every MV inside the job's own bounds plus the sub-pel filter apron. Returns (index, top, bottom);
    top may be negative and bottom beyond the picture (the replicated border)."""
    y = jobs["pu_y"].astype(np.int64)
    idx = np.flatnonzero((y >= row_lo * ctu) & (y < row_hi * ctu))
    if len(idx) == 0:
        return idx, row_lo * ctu, row_lo * ctu
    j = jobs[idx]
    top = int((y[idx] + j["mvmin"][:, 1]).min()) - apron
    bottom = int((y[idx] + j["h"].astype(np.int64) + j["mvmax"][:, 1]).max()) + apron
    return idx, top, bottom


def crop_band(buf, stride, margin_y, height, row_lo, row_hi, top, bottom, ctu=64):
    """Rows of a padded plane (margin_y replicated rows above and below `height` picture rows) that a band needs, as a
    padded plane of its own: picture rows [row_lo * ctu, min(row_hi * ctu, height)) with a halo of `halo` rows on both
    sides, halo = the larger of what [top, bottom) asks above and below. Returns (flat buffer, band height, halo, y0);
    a PU at picture row y is at row y - y0 of the band."""
    y0, y1 = row_lo * ctu, min(row_hi * ctu, height)
    halo = max(y0 - top, bottom - y1, 0)
    if y0 - halo < -margin_y or y1 + halo > height + margin_y:
        raise ValueError("band rows %d..%d with a halo of %d rows leave the padded plane (margin %d)" % (y0, y1, halo, margin_y))
    rows = buf.reshape(-1, stride)[margin_y + y0 - halo:margin_y + y1 + halo]
    return np.ascontiguousarray(rows).reshape(-1), y1 - y0, halo, y0


def band_record_index(jobs, n_ctu_rows, rank, world):
    """jobs: one job list per reference (unsharded). Returns the (reference, index into that reference's list) of every
    record `rank` produces, in its record order: its pieces in turn, then stably sorted by PU size, largest first (one
    launch per size)."""
    il, wl = [], []
    for (r, lo, hi) in row_units(len(jobs), n_ctu_rows, rank, world):
        idx, _, _ = band_jobs(jobs[r], lo, hi)
        il.append(np.stack([np.full(len(idx), r), idx], axis=1))
        wl.append(jobs[r]["w"][idx])
    if not il:
        return np.zeros((0, 2), np.int64)
    index, w = np.concatenate(il), np.concatenate(wl)
    return index[np.argsort(-w.astype(np.int64), kind="stable")]


def assemble_band_records(jobs, n_ctu_rows, world, gathered):
    """gathered: per rank a uint8 array of 8-byte records (MV x, y int16; cost int32), padded at the end. Returns
    (mv, cost): per reference the MVs [njobs, 2] and costs [njobs] in the order of the unsharded lists; a job no rank
    reported keeps cost -1."""
    mv = [np.zeros((len(j), 2), np.int16) for j in jobs]
    cost = [np.full(len(j), -1, np.int32) for j in jobs]
    for q, t in enumerate(gathered):
        index = band_record_index(jobs, n_ctu_rows, q, world)
        rec = np.asarray(t, np.uint8).reshape(-1, 8)[:len(index)]
        for r in range(len(jobs)):
            sel = index[:, 0] == r
            mv[r][index[sel, 1]] = rec[sel, :4].copy().view(np.int16).reshape(-1, 2)
            cost[r][index[sel, 1]] = rec[sel, 4:].copy().view(np.int32).reshape(-1)
    return mv, cost


def round_robin(n_units, rank, world):
    """Unit indices of `rank` when units of unequal, unknown cost are dealt out in turn (lowres fields: one search per
    (frame, list, distance), slicetype.cpp:1150-1154)."""
    return list(range(rank, n_units, world))


def padded_count(n_units, world):
    """Units of the rank that holds the most (gather buffers are padded to it)."""
    return -(-n_units // world)
