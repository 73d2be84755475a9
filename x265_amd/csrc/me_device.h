/* Device-side building blocks of the motion-estimation engine (gfx950, wave64).
 *
 * Work decomposition used by every kernel here: a PU is cut into 4x4-pixel tiles;
 * one lane owns one (candidate, tile) pair, keeps the 16 source and 16 reference
 * (or interpolated) pixels in VGPRs, and produces that tile's SAD or Hadamard
 * cost without any cross-lane traffic. Per-candidate totals are then formed with a
 * segmented butterfly over the lanes that share a candidate.
 *
 * Arithmetic follows the reference's C primitives exactly:
 *   sad / satd            source/common/pixel.cpp:92-109, :201-288
 *   8-tap interpolation   source/common/ipfilter.cpp:194-400
 *   lowres qpel           source/common/lowres.h:78-97
 */
#ifndef X265AMD_ME_DEVICE_H
#define X265AMD_ME_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace x265amd {

/* All LDS of the ME kernels is carved from this one dynamic array, addressed by byte
 * offset, so every access is visibly an LDS access (ds_* instructions, not flat). */
extern __shared__ __attribute__((aligned(16))) unsigned char x265amd_smem[];
#define AS1 __attribute__((address_space(1)))

/* HEVC luma taps, source/Lib/TLibCommon/TComRom.cpp:175-181 */
__constant__ static const int16_t kLumaTaps[4][8] = {
    {  0, 0,   0, 64,  0,   0, 0,  0 },
    { -1, 4, -10, 58, 17,  -5, 1,  0 },
    { -1, 4, -11, 40, 40, -11, 4, -1 },
    {  0, 1,  -5, 17, 58, -10, 4, -1 },
};

/* the square of refinement offsets {0, N, S, W, E, NW, SW, NE, SE}; motion.cpp:147 */
__constant__ static const int8_t kSquare1[9][2] = { {0,0}, {0,-1}, {0,1}, {-1,0}, {1,0}, {-1,-1}, {-1,1}, {1,-1}, {1,1} };

__device__ __forceinline__ void wave_sync()
{
    /* LDS operations of one wave retire in order; this only stops the compiler
     * from moving LDS accesses of different lanes across the hand-off point. */
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

/* v of the lane a DPP control word selects (quad permutes, row mirrors: patterns in which every
 * lane has a source, so no previous value of the result needs to be kept) */
template<int CTRL> __device__ __forceinline__ int dpp_all(int v)
{
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);
}

/* ---- 4 horizontally adjacent pixels ---- */
template<typename P> struct Px;

template<> struct Px<uint8_t>
{
    typedef uint32_t quad;
    static __device__ __forceinline__ quad load(const uint8_t *p)
    {
        typedef uint32_t __attribute__((aligned(1))) u32u;
        return *(const u32u*)p;
    }
    static __device__ __forceinline__ quad load_aligned(const uint8_t *p) { return *(const uint32_t*)p; }
    static __device__ __forceinline__ quad load_g(const uint8_t AS1 *p)
    {
        typedef uint32_t __attribute__((aligned(1))) u32u;
        return *(const u32u AS1*)p;
    }
    /* 4 pixels at an arbitrary byte offset of the LDS array: two aligned dwords + v_alignbyte */
    static __device__ __forceinline__ quad load_lds(int byteoff)
    {
        const uint32_t *q = (const uint32_t*)(x265amd_smem + (byteoff & ~3));
        return __builtin_amdgcn_alignbyte(q[1], q[0], (uint32_t)(byteoff & 3));
    }
    static __device__ __forceinline__ quad load_lds_aligned(int byteoff) { return *(const uint32_t*)(x265amd_smem + byteoff); }
    static __device__ __forceinline__ void store_lds_aligned(int byteoff, quad v) { *(uint32_t*)(x265amd_smem + byteoff) = v; }
    static __device__ __forceinline__ int sad(quad a, quad b, int acc) { return (int)__builtin_amdgcn_sad_u8(a, b, (uint32_t)acc); }
    static __device__ __forceinline__ void unpack(quad q, int v[4])
    {
        v[0] = q & 0xff; v[1] = (q >> 8) & 0xff; v[2] = (q >> 16) & 0xff; v[3] = q >> 24;
    }
    /* rounding average of 4 packed pixels, (a + b + 1) >> 1 per byte */
    static __device__ __forceinline__ quad avg(quad a, quad b) { return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu); }
};

template<> struct Px<uint16_t>
{
    typedef uint2 quad;
    static __device__ __forceinline__ quad load(const uint16_t *p)
    {
        typedef uint32_t __attribute__((aligned(2))) u32u;
        const u32u *q = (const u32u*)p;
        return make_uint2(q[0], q[1]);
    }
    static __device__ __forceinline__ quad load_aligned(const uint16_t *p) { return *(const uint2*)p; }
    static __device__ __forceinline__ quad load_g(const uint16_t AS1 *p)
    {
        typedef uint32_t __attribute__((aligned(2))) u32u;
        const u32u AS1 *q = (const u32u AS1*)p;
        return make_uint2(q[0], q[1]);
    }
    /* 4 pixels (8 bytes) at a 2-byte aligned offset: three aligned dwords + v_alignbit */
    static __device__ __forceinline__ quad load_lds(int byteoff)
    {
        const uint32_t *q = (const uint32_t*)(x265amd_smem + (byteoff & ~3));
        const uint32_t s = (byteoff & 2) * 8;
        const uint32_t lo = q[0], mid = q[1], hi = q[2];
        return make_uint2(__builtin_amdgcn_alignbit(mid, lo, s), __builtin_amdgcn_alignbit(hi, mid, s));
    }
    static __device__ __forceinline__ quad load_lds_aligned(int byteoff) { return *(const uint2*)(x265amd_smem + byteoff); }
    static __device__ __forceinline__ void store_lds_aligned(int byteoff, quad v) { *(uint2*)(x265amd_smem + byteoff) = v; }
    static __device__ __forceinline__ int sad(quad a, quad b, int acc)
    {
        return (int)__builtin_amdgcn_sad_u16(a.y, b.y, __builtin_amdgcn_sad_u16(a.x, b.x, (uint32_t)acc));
    }
    static __device__ __forceinline__ void unpack(quad q, int v[4])
    {
        v[0] = q.x & 0xffff; v[1] = q.x >> 16; v[2] = q.y & 0xffff; v[3] = q.y >> 16;
    }
    static __device__ __forceinline__ quad avg(quad a, quad b)
    {
        return make_uint2((a.x | b.x) - (((a.x ^ b.x) >> 1) & 0x7fff7fffu), (a.y | b.y) - (((a.y ^ b.y) >> 1) & 0x7fff7fffu));
    }
};

/* 4-pixel loads relative to a tile origin, from the staged LDS window or from global memory */
template<typename P> struct LdsLoader
{
    int base, stride; /* byte offset of the tile origin in the LDS array; row pitch in samples */
    __device__ __forceinline__ typename Px<P>::quad operator()(int dx, int dy) const
    {
        return Px<P>::load_lds(base + (dy * stride + dx) * (int)sizeof(P));
    }
};
template<typename P> struct GlobalLoader
{
    const P AS1 *p;
    int stride;
    __device__ __forceinline__ typename Px<P>::quad operator()(int dx, int dy) const
    {
        return Px<P>::load_g(p + (long)dy * stride + dx);
    }
};

/* Sum of |Hadamard4x4(d)|, >> 1. The 16 coefficients all share the parity of the sum
 * of d, so the abs-sum is even and halving per 4x4 tile equals the reference's
 * halving per 8x4 pair (pixel.cpp:226, :251). */
__device__ __forceinline__ int had4x4_half(const int d[4][4])
{
    int t[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        int a0 = d[r][0] + d[r][1], a1 = d[r][0] - d[r][1];
        int a2 = d[r][2] + d[r][3], a3 = d[r][2] - d[r][3];
        t[r][0] = a0 + a2; t[r][1] = a1 + a3; t[r][2] = a0 - a2; t[r][3] = a1 - a3;
    }
    int s = 0;
#pragma unroll
    for (int c = 0; c < 4; c++)
    {
        int a0 = t[0][c] + t[1][c], a1 = t[0][c] - t[1][c];
        int a2 = t[2][c] + t[3][c], a3 = t[2][c] - t[3][c];
        s += abs(a0 + a2) + abs(a1 + a3) + abs(a0 - a2) + abs(a1 - a3);
    }
    return s >> 1;
}

/* The same for differences of 8-bit samples, on packed int16 pairs: E[r] = (d[r][0], d[r][2]),
 * O[r] = (d[r][1], d[r][3]). Every intermediate stays below 2^12. The two vertical stages and
 * the first horizontal one are whole-register v_pk_add / v_pk_sub; the last stage pairs the two
 * halves of a register, and |a + b| + |a - b| = 2 max(|a|, |b|) turns it into a maximum, which
 * is also where the halving goes. */
typedef short pk16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int had4x4_half_pk(const pk16 E[4], const pk16 O[4])
{
    pk16 v[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
    {
        const pk16 *X = i ? O : E;
        const pk16 a0 = X[0] + X[1], a1 = X[0] - X[1], a2 = X[2] + X[3], a3 = X[2] - X[3];
        v[i][0] = a0 + a2; v[i][1] = a1 + a3; v[i][2] = a0 - a2; v[i][3] = a1 - a3;
    }
    pk16 acc = { 0, 0 };
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 2; i++)
        {
            pk16 t = i ? v[0][r] - v[1][r] : v[0][r] + v[1][r];
            t = __builtin_elementwise_max(t, -t);
            const uint32_t u = __builtin_bit_cast(uint32_t, t);
            acc += __builtin_elementwise_max(t, __builtin_bit_cast(pk16, __builtin_amdgcn_alignbit(u, u, 16)));
        }
    return (int)(unsigned short)acc.x;
}
/* even / odd samples of four packed bytes as int16 pairs */
__device__ __forceinline__ pk16 even_bytes(uint32_t q) { return __builtin_bit_cast(pk16, __builtin_amdgcn_perm(q, q, 0x0c020c00u)); }
__device__ __forceinline__ pk16 odd_bytes(uint32_t q) { return __builtin_bit_cast(pk16, __builtin_amdgcn_perm(q, q, 0x0c030c01u)); }

__device__ __forceinline__ int clip_pixel(int v16, int maxv)
{
    /* v16 already passed through the reference's int16_t cast */
    return v16 < 0 ? 0 : (v16 > maxv ? maxv : v16);
}

__device__ __forceinline__ int as_i16(int v) { return (int)(int16_t)v; }

/* Interpolated 4x4 tile at integer position `r` (top-left of the tile in the
 * reference plane) with fractional phase (xf, yf) != (0, 0).
 *   yf == 0 : luma_hpp   (sum + 32) >> 6, int16 cast, clip      ipfilter.cpp:194-233
 *   xf == 0 : luma_vpp                                           ipfilter.cpp:280-319
 *   else    : luma_hps(isRowExt) into int16, then luma_vsp       ipfilter.cpp:235-278, :359-400
 *             exactly as subpelCompare chains them (motion.cpp:1681-1685) */
template<typename P, class LD>
__device__ __forceinline__ void interp_tile(const LD &ld, int xf, int yf, int depth, int o[4][4])
{
    const int maxv = (1 << depth) - 1;
    int ch[8];
#pragma unroll
    for (int k = 0; k < 8; k++) ch[k] = kLumaTaps[xf][k];
    if (yf == 0)
    {
#pragma unroll
        for (int y = 0; y < 4; y++)
        {
            int px[12];
            Px<P>::unpack(ld(-3, y), px);
            Px<P>::unpack(ld(1, y), px + 4);
            Px<P>::unpack(ld(5, y), px + 8);
#pragma unroll
            for (int x = 0; x < 4; x++)
            {
                int sum = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) sum += px[x + k] * ch[k];
                o[y][x] = clip_pixel(as_i16((sum + 32) >> 6), maxv);
            }
        }
        return;
    }
    /* Vertical and 2-D phases: reference rows -3 .. 7 stream through an 8-row shift register
     * (horizontally filtered first when xf != 0); rows are primed in a rolled loop so the code
     * stays small and only one row of loads is in flight. */
    int cv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) cv[k] = kLumaTaps[yf][k];
    const int headRoom = 14 - depth;
    const int shiftH = 6 - headRoom, offH = -(8192 << shiftH);
    const int shiftV = 6 + headRoom, offV = (1 << (shiftV - 1)) + (8192 << 6);
    int win[8][4];
#pragma unroll
    for (int k = 0; k < 8; k++)
#pragma unroll
        for (int x = 0; x < 4; x++) win[k][x] = 0;
    auto feed = [&](int row)
    {
#pragma unroll
        for (int k = 0; k < 7; k++)
#pragma unroll
            for (int x = 0; x < 4; x++) win[k][x] = win[k + 1][x];
        if (xf == 0)
            Px<P>::unpack(ld(0, row), win[7]);
        else
        {
            int px[12];
            Px<P>::unpack(ld(-3, row), px);
            Px<P>::unpack(ld(1, row), px + 4);
            Px<P>::unpack(ld(5, row), px + 8);
#pragma unroll
            for (int x = 0; x < 4; x++)
            {
                int sum = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) sum += px[x + k] * ch[k];
                win[7][x] = as_i16((sum + offH) >> shiftH);
            }
        }
    };
#pragma unroll 1
    for (int rr = -3; rr < 4; rr++) feed(rr);
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        feed(y + 4);
#pragma unroll
        for (int x = 0; x < 4; x++)
        {
            int sum = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) sum += win[k][x] * cv[k];
            o[y][x] = xf == 0 ? clip_pixel(as_i16((sum + 32) >> 6), maxv)
                              : clip_pixel(as_i16((sum + offV) >> shiftV), maxv);
        }
    }
}

/* 4-tap kernels of the Rockchip fractional search: g_chromaFilter[frac * 2]
 * (motion.cpp:1455-1490; table TComRom.cpp:183-193) */
__constant__ static const int16_t kRkTaps[4][4] = { { 0, 64, 0, 0 }, { -4, 54, 16, -2 }, { -4, 36, 36, -4 }, { -2, 16, 54, -4 } };
/* the same taps as four signed bytes (tap 0 lowest) and as two int16 pairs (taps 0-1, taps 2-3) */
#define RK_B(a, b, c, d) ((uint32_t)((a) & 0xff) | ((uint32_t)((b) & 0xff) << 8) | ((uint32_t)((c) & 0xff) << 16) | ((uint32_t)((d) & 0xff) << 24))
#define RK_H(a, b) ((uint32_t)((a) & 0xffff) | ((uint32_t)((b) & 0xffff) << 16))
__constant__ static const uint32_t kRkTapBytes[4] = { RK_B(0, 64, 0, 0), RK_B(-4, 54, 16, -2), RK_B(-4, 36, 36, -4), RK_B(-2, 16, 54, -4) };
__constant__ static const uint32_t kRkTapPairs[4][2] = { { RK_H(0, 64), RK_H(0, 0) }, { RK_H(-4, 54), RK_H(16, -2) },
                                                         { RK_H(-4, 36), RK_H(36, -4) }, { RK_H(-2, 16), RK_H(54, -4) } };
#undef RK_B
#undef RK_H

/* RK flavour of interp_tile: InterpHoriz / InterpVert / FilterHorizontal + FilterVertical
 * with N = 4 (motion.cpp:1495-1648), phase (xf, yf) != (0, 0) */
template<typename P, class LD>
__device__ __forceinline__ void rk_interp_tile(const LD &ld, int xf, int yf, int depth, int o[4][4])
{
    const int maxv = (1 << depth) - 1;
    int ch[4], cv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { ch[k] = kRkTaps[xf][k]; cv[k] = kRkTaps[yf][k]; }
    const int headRoom = 14 - depth;
    const int shiftH = 6 - headRoom, offH = -(8192 << shiftH);
    const int shiftV = 6 + headRoom, offV = (1 << (shiftV - 1)) + (8192 << 6);
    if (yf == 0)
    {
#pragma unroll
        for (int y = 0; y < 4; y++)
        {
            int px[8];
            Px<P>::unpack(ld(-1, y), px);
            Px<P>::unpack(ld(3, y), px + 4);
#pragma unroll
            for (int x = 0; x < 4; x++)
            {
                const int sum = px[x] * ch[0] + px[x + 1] * ch[1] + px[x + 2] * ch[2] + px[x + 3] * ch[3];
                o[y][x] = clip_pixel(as_i16((sum + 32) >> 6), maxv);
            }
        }
        return;
    }
    int win[4][4];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int x = 0; x < 4; x++) win[k][x] = 0;
    auto feed = [&](int row)
    {
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int x = 0; x < 4; x++) win[k][x] = win[k + 1][x];
        if (xf == 0)
            Px<P>::unpack(ld(0, row), win[3]);
        else
        {
            int px[8];
            Px<P>::unpack(ld(-1, row), px);
            Px<P>::unpack(ld(3, row), px + 4);
#pragma unroll
            for (int x = 0; x < 4; x++)
                win[3][x] = as_i16((px[x] * ch[0] + px[x + 1] * ch[1] + px[x + 2] * ch[2] + px[x + 3] * ch[3] + offH) >> shiftH);
        }
    };
#pragma unroll 1
    for (int rr = -1; rr < 2; rr++) feed(rr);
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        feed(y + 2);
#pragma unroll
        for (int x = 0; x < 4; x++)
        {
            const int sum = win[0][x] * cv[0] + win[1][x] * cv[1] + win[2][x] * cv[2] + win[3][x] * cv[3];
            o[y][x] = xf == 0 ? clip_pixel(as_i16((sum + 32) >> 6), maxv)
                              : clip_pixel(as_i16((sum + offV) >> shiftV), maxv);
        }
    }
}

/* cost of one tile given the 4 packed source rows and 16 predicted pixels */
template<typename P>
__device__ __forceinline__ int tile_cost_px(const typename Px<P>::quad f[4], const int o[4][4], bool satd)
{
    if constexpr (sizeof(P) == 1)
        if (satd)
        {
            pk16 E[4], O[4];
#pragma unroll
            for (int y = 0; y < 4; y++)
            {
                E[y] = even_bytes(f[y]) - __builtin_bit_cast(pk16, (uint32_t)(o[y][0] | (o[y][2] << 16)));
                O[y] = odd_bytes(f[y]) - __builtin_bit_cast(pk16, (uint32_t)(o[y][1] | (o[y][3] << 16)));
            }
            return had4x4_half_pk(E, O);
        }
    int d[4][4];
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        int fv[4];
        Px<P>::unpack(f[y], fv);
#pragma unroll
        for (int x = 0; x < 4; x++) d[y][x] = fv[x] - o[y][x];
    }
    if (satd)
        return had4x4_half(d);
    int s = 0;
#pragma unroll
    for (int y = 0; y < 4; y++)
#pragma unroll
        for (int x = 0; x < 4; x++) s += abs(d[y][x]);
    return s;
}

/* cost of one tile against 4 packed reference rows */
template<typename P>
__device__ __forceinline__ int tile_cost_quads(const typename Px<P>::quad f[4], const typename Px<P>::quad q[4], bool satd)
{
    if (!satd)
    {
        int s = 0;
#pragma unroll
        for (int y = 0; y < 4; y++) s = Px<P>::sad(f[y], q[y], s);
        return s;
    }
    if constexpr (sizeof(P) == 1)
    {
        pk16 E[4], O[4];
#pragma unroll
        for (int y = 0; y < 4; y++)
        {
            E[y] = even_bytes(f[y]) - even_bytes(q[y]);
            O[y] = odd_bytes(f[y]) - odd_bytes(q[y]);
        }
        return had4x4_half_pk(E, O);
    }
    int o[4][4];
#pragma unroll
    for (int y = 0; y < 4; y++) Px<P>::unpack(q[y], o[y]);
    return tile_cost_px<P>(f, o, true);
}

} // namespace x265amd
#endif
