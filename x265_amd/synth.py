"""Deterministic synthetic clips (luma; 4:2:0 chroma for motion compensation) and padded-plane helpers (host side, numpy).

The clips follow the recipe the reference numbers were measured on: a 5x5
box-smoothed uniform random texture translated by (3i mod 32, 2i mod 32) pixels per
frame plus N(0, 2) noise. ``kind`` selects the worst case (pure noise) and the best
case (static) variants.

Plane layout mirrors the reference's padded pictures (TComPicYuv.cpp:69-131,
xExtendPicCompBorder :169): ``margin`` replicated border pixels on every side,
stride = width + 2 * margin_x (optionally rounded up), origin = margin_y * stride +
margin_x.
"""
import numpy as np

LUMA_MARGIN = 304  # 64 + 208 + 8 + 24, TComPicYuv.cpp:85-91 with the fork's search range


def make_clip(width, height, nframes, seed=1234, depth=8, kind="texture"):
    rng = np.random.default_rng(seed)
    maxv = (1 << depth) - 1
    dt = np.uint8 if depth <= 8 else np.uint16
    frames = []
    if kind == "noise":
        for _ in range(nframes):
            frames.append(rng.integers(0, maxv + 1, size=(height, width)).astype(dt))
        return frames
    base = rng.integers(0, 256, size=(height + 64, width + 64)).astype(np.float32)
    p = np.pad(base, 2, mode="edge")
    # 5x5 box filter via a summed-area table
    sat = np.zeros((p.shape[0] + 1, p.shape[1] + 1), np.float64)
    sat[1:, 1:] = p.cumsum(0).cumsum(1)
    b = (sat[5:, 5:] - sat[:-5, 5:] - sat[5:, :-5] + sat[:-5, :-5]) / 25.0
    scale = maxv / 255.0
    for i in range(nframes):
        dx, dy = ((i * 3) % 32, (i * 2) % 32) if kind == "texture" else (0, 0)
        noise = rng.normal(0, 2, (height, width)) if kind == "texture" else 0.0
        y = np.clip((b[dy:dy + height, dx:dx + width] + noise) * scale, 0, maxv)
        frames.append(y.astype(dt))
    return frames


def baseline_clip(width, height, nframes, seed=1234):
    """Luma frames of the 8-bit 4:2:0 clips the reference encoder's bitstream MD5s were recorded on (chroma is constant
    128): the same recipe as make_clip(kind="texture"), but with the box filter taken as a float32 window mean, which
    rounds differently, so the two generators are not interchangeable."""
    from numpy.lib.stride_tricks import sliding_window_view
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(height + 64, width + 64), dtype=np.uint8)
    b = sliding_window_view(np.pad(base.astype(np.float32), 2, mode="edge"), (5, 5)).mean(axis=(2, 3))
    frames = []
    for i in range(nframes):
        dx, dy = (i * 3) % 32, (i * 2) % 32
        frames.append(np.clip(b[dy:dy + height, dx:dx + width] + rng.normal(0, 2, (height, width)), 0, 255).astype(np.uint8))
    return frames


def write_yuv420(path, frames):
    """Planar 8-bit 4:2:0 file of luma frames with constant-128 chroma."""
    with open(path, "wb") as f:
        for y in frames:
            f.write(np.ascontiguousarray(y).tobytes())
            f.write(np.full((y.shape[0] // 2) * (y.shape[1] // 2) * 2, 128, np.uint8).tobytes())


def make_chroma(width, height, nframes, seed=1234, depth=8):
    """Cb / Cr planes of a 4:2:0 clip ((height/2, width/2) each): the luma recipe at half size with other seeds."""
    cb = make_clip(width // 2, height // 2, nframes, seed + 7001, depth)
    cr = make_clip(width // 2, height // 2, nframes, seed + 9001, depth)
    return cb, cr


def pad_plane(img, margin_x=LUMA_MARGIN, margin_y=None, stride_align=1):
    """Return (flat buffer, origin offset in samples, stride in samples)."""
    if margin_y is None:
        margin_y = margin_x
    h, w = img.shape
    stride = w + 2 * margin_x
    if stride % stride_align:
        stride += stride_align - stride % stride_align
    buf = np.zeros((h + 2 * margin_y, stride), img.dtype)
    buf[:, :w + 2 * margin_x] = np.pad(img, ((margin_y, margin_y), (margin_x, margin_x)), mode="edge")
    return buf.reshape(-1), margin_y * stride + margin_x, stride


def lowres_dims(width, height):
    """(lowres width, lines, widthInCU, heightInCU), lowres.cpp:33-44."""
    w, h = width // 2, height // 2
    wcu, hcu = (w + 7) >> 3, (h + 7) >> 3
    return wcu * 8, hcu * 8, wcu, hcu
