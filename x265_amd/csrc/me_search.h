/* One-wavefront motion search: MotionEstimate::motionEstimate re-designed for a
 * 64-lane wave (reference: source/encoder/motion.cpp:549-1146).
 *
 * The search decisions are sequential and data dependent, so one wave owns one PU
 * and walks the reference's decision sequence with wave-uniform control flow.
 * What runs in parallel is every group of candidates the reference evaluates
 * before it next looks at a cost: each (candidate, 4x4 tile) pair goes to one
 * lane (me_device.h), per-candidate sums come from a segmented butterfly, and the
 * candidates' lambda*bits terms are gathered by one lane each. Costs are then
 * compared in the reference's visit order with its strict '<', which keeps every
 * tie-break identical.
 */
#ifndef X265AMD_ME_SEARCH_H
#define X265AMD_ME_SEARCH_H

#include "me_device.h"
#include "../../include/x265_amd_me.h"

namespace x265amd {

#define MAXC 64   /* candidates per batched evaluation: one per lane */
#define MAXP 16   /* candidates of one sub-pel batch (interpolation planes) */

/* radius-2 hexagon with wrap-around entries, (x-1)%6, motion.cpp:145-146 */
__constant__ static const int8_t kHex2[8][2] = { {-1,-2}, {-2,0}, {-1,2}, {1,2}, {2,0}, {1,-2}, {-1,-2}, {-2,0} };
__constant__ static const uint8_t kMod6m1[8] = { 5, 0, 1, 2, 3, 4, 5, 0 };
/* first hexagon ring in the order the reference measures it (motion.cpp:673-681) */
__constant__ static const int8_t kHexRing[6][2] = { {-2,0}, {-1,2}, {1,2}, {2,0}, {1,-2}, {-1,-2} };
/* radius-4 hexagon grid of the UMH search, motion.cpp:149-155 */
__constant__ static const int8_t kHex4[16][2] = { {0,-4}, {0,4}, {-2,-3}, {2,-3}, {-4,-2}, {4,-2}, {-4,-1}, {4,-1},
                                                  {-4,0}, {4,0}, {-4,1}, {4,1}, {-4,2}, {4,2}, {-2,3}, {2,3} };
/* offsets for the two-point search, motion.cpp:156-166 */
__constant__ static const int8_t kTwoPoint[16][2] = { {-1,0}, {0,-1}, {-1,-1}, {1,-1}, {-1,0}, {1,0}, {-1,1}, {-1,-1},
                                                      {1,-1}, {1,1}, {-1,0}, {0,1}, {-1,1}, {1,1}, {1,0}, {0,1} };
/* SubpelWorkload {hpel_iters, hpel_dirs, qpel_iters, qpel_dirs, hpel_satd}, motion.cpp:53-63 */
__constant__ static const int8_t kWorkload[8][5] = { {1,4,0,4,0}, {1,4,1,4,0}, {1,4,1,4,1}, {2,4,1,4,1},
                                                     {2,4,2,4,1}, {1,8,1,8,1}, {2,8,1,8,1}, {2,8,2,8,1} };

/* Packed 8-tap luma filters for the plane-based sub-pel stage, 12 dwords per phase: 4 dwords of
 * int16 tap pairs (c0,c1)..(c6,c7); 5 dwords of the same taps moved down one sample
 * (0,c0),(c1,c2)..(c7,0); 2 dwords of taps as signed bytes; 1 unused */
struct TapWords
{
    int w[48];
    constexpr TapWords() : w()
    {
        const int taps[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
                                 { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
        for (int ph = 0; ph < 4; ph++)
        {
            const int *c = taps[ph];
            for (int i = 0; i < 4; i++) w[ph * 12 + i] = (int)((unsigned)(c[2 * i] & 0xffff) | ((unsigned)(c[2 * i + 1] & 0xffff) << 16));
            for (int j = 0; j < 5; j++)
            {
                const int lo = j ? c[2 * j - 1] : 0, hi = j < 4 ? c[2 * j] : 0;
                w[ph * 12 + 4 + j] = (int)((unsigned)(lo & 0xffff) | ((unsigned)(hi & 0xffff) << 16));
            }
            for (int b = 0; b < 2; b++)
            {
                unsigned v = 0;
                for (int k = 0; k < 4; k++) v |= (unsigned)(c[4 * b + k] & 0xff) << (8 * k);
                w[ph * 12 + 9 + b] = (int)v;
            }
        }
    }
};
__constant__ static const TapWords kTapWords = TapWords();

struct PlaneDev
{
    const void *origin;
    int stride;
};

/* Per-wave LDS layout, byte offsets from the wave's base. Candidate arrays hold MAXC ints. */
enum
{
    L_SX = 0, L_SY = 256,        /* where to measure: full-pel offset, or qpel MV */
    L_MVX = 512, L_MVY = 768,    /* MV charged for the candidate (same units) */
    L_SH = 1024,                 /* left shift applied to mv before the cost lookup, + flavour flags */
    L_META = 1280, L_COSTS = 1536,
    L_CPL = 1792,                /* sub-pel candidates: plane | (row + 16) << 8 | yfrac << 16 | satd << 18 */
    L_CLIST = 1856,              /* candidates served by the planes currently in LDS */
    L_PDESC = 1920,              /* MAXP plane descriptors of 32 bytes */
    L_TAPS = 2432,               /* 4 phases x 12 dwords of packed filter taps */
    L_CTX = 2624,                /* EvalCtx */
    L_FENC = 2880                /* source PU, row pitch 64 samples (FENC_STRIDE) */
};

/* Everything the batched evaluation needs, parked in LDS so that the evaluation routines
 * are ordinary functions of (base, K, mode) and the search logic stays in registers. */
struct EvalCtx
{
    unsigned long long ref[4];   /* global addresses of the plane origins */
    unsigned long long ctab;     /* centred MVD cost table */
    long roff;                   /* pu_y * rstride + pu_x */
    int rstride, w, h, T, Tp, lTp, wt, inv_wt, depth;
    int mvpx, mvpy;
    int rk_have_mvd, rk_mvp[3][2];
    int rk_win, rk_pitch, rk_x0, rk_y0; /* Rockchip FME: the reference region staged in LDS (byte offset, pitch in samples) and the MV of its first sample */
    int planes_off, plane_budget; /* LDS area for the interpolation planes of the sub-pel stage */
};

static_assert(sizeof(EvalCtx) <= L_FENC - L_CTX, "EvalCtx must fit its LDS slot");

/* LDS bytes of the reference region the Rockchip fractional stage reads for a size x size CU:
 * the integer parts of the 49 positions are the RIME vector + {-1, 0}, the 4-tap filters reach
 * one sample before and two after, and tiles are loaded as whole quads */
#define RK_REGION_COLS(size) ((size) + 8)
#define RK_REGION_ROWS(size) ((size) + 4)
template<typename P> constexpr int rk_region_bytes(int size) { return RK_REGION_COLS(size) * RK_REGION_ROWS(size) * (int)sizeof(P) + 16; }

__device__ __forceinline__ int *lds_i(int base, int off) { return (int*)(x265amd_smem + base + off); }
__device__ __forceinline__ EvalCtx *lds_ctx(int base) { return (EvalCtx*)(x265amd_smem + base + L_CTX); }
__device__ __forceinline__ unsigned long long uni64(unsigned long long v)
{
    return ((unsigned long long)(unsigned)uni((int)(v >> 32)) << 32) | (unsigned)uni((int)v);
}

/* LDS bytes of the interpolation planes for PUs up to dim x dim. One plane is at most
 * PLANE_COLS columns of (h + 10) int16 rows (wider PUs go block by block). A refinement round
 * needs one plane per qpel x of its candidates when the samples are 8-bit (three planes), and up
 * to six with deeper samples, where the kind of the first pass splits them further; batches with
 * more planes go through in several chunks. Launches of small 8-bit PUs get exactly those three
 * planes, which lets a CU hold six waves per SIMD; larger launches have fewer waves than the
 * chip has room for and keep room for two planes of a 64-row PU. */
#define PLANE_COLS 16
constexpr int plane_budget(int dim, int sample_bytes)
{
    return (dim <= PLANE_COLS && sample_bytes == 1 ? 3 : 6) * (dim < PLANE_COLS ? dim : PLANE_COLS) * (((dim < PLANE_COLS ? dim : PLANE_COLS) + 10) & ~1) * 2;
}
/* per-wave LDS: candidate arrays + context + source block of `rows` rows [+ planes] */
template<typename P> constexpr int wave_lds_bytes(int rows, bool planes = false)
{
    return L_FENC + rows * 64 * (int)sizeof(P) + (planes ? plane_budget(rows, (int)sizeof(P)) + 16 : 0);
}

/* ceil(65536 / wt) for the tile counts a PU row can have (widths 4 .. 64), so that
 * (t * inv) >> 16 == t / wt for every tile index; a division costs some twenty instructions */
__device__ __forceinline__ int inv_tiles(int wt)
{
    return wt == 1 ? 65536 : wt == 2 ? 32768 : wt == 3 ? 21846 : wt == 4 ? 16384 : wt == 6 ? 10923 : wt == 8 ? 8192
         : wt == 12 ? 5462 : wt == 16 ? 4096 : (65536 + wt - 1) / wt;
}

/* The sub-pel rounds of the partitions no fixed-shape kernel takes (me_small.h): refine_any_kernel, a wave per PU,
 * above 128 samples; refine_tiny_kernel, several PUs per wave, for the shapes whose (tile, column) pairs fill 8, 16 or
 * 32 lanes exactly: 8x4, 4x8 | 16x4, 4x16, 8x8 | 16x8, 8x16. tiny_class: log2 of the lanes such a PU takes, 0 for any
 * other shape. */
__device__ __forceinline__ int tiny_class(int w, int h)
{
    const int a = w * h;
    if ((w | h) & 3 || w > 16 || h > 16 || (w == 12 || h == 12)) return 0;
    return a == 32 ? 3 : (a == 64 ? 4 : (a == 128 ? 5 : 0));
}
__device__ __forceinline__ bool any_refines(int w, int h) { return w * h >= 128 && !tiny_class(w, h); }
/* the PUs star_rect_kernel (me_small.h) takes through a fixed-shape integer stage, a wave per PU: sides of 8 to 64
 * samples, 64 to 512 (tile, column) pairs; the large parts of the asymmetric splits; and 16x16 / 32x32 PUs, for the lists
 * of launches made for another size (the callers test that first) */
__device__ __forceinline__ bool rect_class(int w, int h)
{
    const bool pw = w == 8 || w == 16 || w == 32 || w == 64, ph = h == 8 || h == 16 || h == 32 || h == 64;
    const int lo = w < h ? w : h, hi = w < h ? h : w;
    return (pw && ph && w * h >= 256 && w * h <= 2048) || (4 * lo == 3 * hi && (hi == 16 || hi == 32 || hi == 64));
}

/* sum over each group of 2^l adjacent lanes, left in every lane of the group */
__device__ __forceinline__ int seg_sum(int v, int l)
{
    if (l >= 1) v += dpp_all<0xB1>(v);  /* quad_perm [1,0,3,2] */
    if (l >= 2) v += dpp_all<0x4E>(v);  /* quad_perm [2,3,0,1] */
    if (l >= 3) v += dpp_all<0x141>(v); /* row_half_mirror */
    if (l >= 4) v += dpp_all<0x140>(v); /* row_mirror */
    if (l >= 5) v += __shfl_xor(v, 16);
    if (l >= 6) v += __shfl_xor(v, 32);
    return v;
}

/* minimum over each row of 16 lanes */
__device__ __forceinline__ int row_min16(int v)
{
    int o;
    o = dpp_all<0xB1>(v); v = o < v ? o : v;
    o = dpp_all<0x4E>(v); v = o < v ? o : v;
    o = dpp_all<0x141>(v); v = o < v ? o : v;
    o = dpp_all<0x140>(v); v = o < v ? o : v;
    return v;
}

/* minimum over the wave, as a scalar */
__device__ __forceinline__ int wave_min(int v)
{
    v = row_min16(v);
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    const int ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}

/* table of up to 16 small signed values (-4 .. 11) packed as nibbles, for per-lane lookups
 * without a memory access */
template<int N> constexpr unsigned long long nibs(const int (&v)[N])
{
    unsigned long long r = 0;
    for (int i = 0; i < N; i++) r |= (unsigned long long)((v[i] + 4) & 15) << (4 * i);
    return r;
}
__device__ __forceinline__ int nib(unsigned long long t, int i) { return (int)((t >> (4 * i)) & 15) - 4; }

/* lambda * bits term of candidate `lane` (0 for lanes that have none). PLAIN: full-pel candidates of the classic
 * searches only (no Rockchip predictors, no UMH grid flavour), for kernels that carry nothing else */
template<bool PLAIN = false>
__device__ __forceinline__ int gather_mvcost(int base, int K, int mode, int addmv)
{
    const int lane = threadIdx.x & 63;
    const EvalCtx *cx = lds_ctx(base);
    int mvc = 0;
    /* mode < 0: mixed batch, candidate k carries its own mode in meta bits 0-3 and its
     * "add mvcost" flag in bit 4 */
    if (lane < K && (mode < 0 ? (lds_i(base, L_META)[lane] >> 4) & 1 : addmv))
    {
        const uint16_t AS1 *ctab = (const uint16_t AS1*)uni64(cx->ctab);
        const int mvpx = uni(cx->mvpx), mvpy = uni(cx->mvpy);
        const int s = lds_i(base, L_SH)[lane];
        const int mx = lds_i(base, L_MVX)[lane], my = lds_i(base, L_MVY)[lane];
        if (!PLAIN && (s & 16))
        {
            /* Rockchip stages: min over three predictors, each sum truncated to uint16
             * (motion.cpp:1284-1292) */
            if (uni(cx->rk_have_mvd))
            {
                mvc = 0x7fffffff;
#pragma unroll
                for (int k = 0; k < 3; k++)
                {
                    const int v = ((int)ctab[mx - uni(cx->rk_mvp[k][0])] + (int)ctab[my - uni(cx->rk_mvp[k][1])]) & 0xffff;
                    mvc = v < mvc ? v : mvc;
                }
            }
        }
        else if (!PLAIN && (s & 8))
            /* UMH hexagon grid: table entries added as ints, MV * 4 without the int16 wrap
             * (ADD_MVCOST, motion.cpp:856) */
            mvc = (int)ctab[mx * 4 - mvpx] + (int)ctab[my * 4 - mvpy];
        else
            /* bitcost.h:46 -- the sum is truncated to uint16 */
            mvc = ((int)ctab[as_i16(mx << (s & 7)) - mvpx] + (int)ctab[as_i16(my << (s & 7)) - mvpy]) & 0xffff;
    }
    return mvc;
}

/* wave-uniform geometry of the PU, pulled from the LDS context into scalar registers */
struct Geo
{
    const void AS1 *ref[4];
    long roff;
    int rstride, T, Tp, lTp, wt, inv_wt, depth;
    /* ref[i] for a per-lane i, by selects (indexing the array would put it in scratch memory) */
    __device__ __forceinline__ const void AS1 *pick(int i) const
    {
        const void AS1 *lo = (i & 1) ? ref[1] : ref[0], *hi = (i & 1) ? ref[3] : ref[2];
        return (i & 2) ? hi : lo;
    }
    template<bool ALLREFS> __device__ __forceinline__ void load(int base)
    {
        const EvalCtx *cx = lds_ctx(base);
        ref[0] = (const void AS1*)uni64(cx->ref[0]);
        if (ALLREFS)
            for (int i = 1; i < 4; i++) ref[i] = (const void AS1*)uni64(cx->ref[i]);
        roff = (long)uni64((unsigned long long)cx->roff);
        rstride = uni(cx->rstride); T = uni(cx->T); Tp = uni(cx->Tp); lTp = uni(cx->lTp);
        wt = uni(cx->wt); inv_wt = uni(cx->inv_wt); depth = uni(cx->depth);
    }
};

/* cost of tile t of candidate k measured by one lane on its own. mode 0: full-pel SAD at
 * offset (sx, sy); mode 1 / 2 (lookahead only): SAD / SATD at qpel MV (sx, sy) through
 * lowresQPelCost; mode 3: Rockchip FME */
template<typename P, bool LOWRES, bool RK>
__device__ __forceinline__ int tile_cost(int base, const Geo &c, int k, int t, int mode)
{
    const int ty = __umul24(t, c.inv_wt) >> 16, tx = t - __mul24(ty, c.wt);
    typename Px<P>::quad f[4], q[4];
#pragma unroll
    for (int y = 0; y < 4; y++) f[y] = Px<P>::load_lds_aligned(base + L_FENC + ((ty * 4 + y) * 64 + tx * 4) * (int)sizeof(P));
    const int cx = lds_i(base, L_SX)[k], cy = lds_i(base, L_SY)[k];
    const P AS1 *ref0 = (const P AS1*)c.ref[0];
    if (mode == 0)
    {
        const P AS1 *r = ref0 + c.roff + (long)__mul24(cy + ty * 4, c.rstride) + cx + tx * 4;
#pragma unroll
        for (int y = 0; y < 4; y++) q[y] = Px<P>::load_g(r + y * c.rstride);
        return tile_cost_quads<P>(f, q, false);
    }
    if constexpr (LOWRES)
    {
        /* lowres.h:78-97: pick the hpel plane; odd qpel = average with the next one */
        const bool satd = mode == 2;
        const int hpelA = (cy & 2) | ((cx & 2) >> 1);
        const P AS1 *ra = (const P AS1*)c.pick(hpelA) + c.roff + (long)__mul24((cy >> 2) + ty * 4, c.rstride) + (cx >> 2) + tx * 4;
#pragma unroll
        for (int y = 0; y < 4; y++) q[y] = Px<P>::load_g(ra + y * c.rstride);
        if ((cx | cy) & 1)
        {
            const int bx = as_i16(cx + (cx & 1) * 2), by = as_i16(cy + (cy & 1) * 2);
            const int hpelB = (by & 2) | ((bx & 2) >> 1);
            const P AS1 *rb = (const P AS1*)c.pick(hpelB) + c.roff + (long)__mul24((by >> 2) + ty * 4, c.rstride) + (bx >> 2) + tx * 4;
#pragma unroll
            for (int y = 0; y < 4; y++) q[y] = Px<P>::avg(q[y], Px<P>::load_g(rb + y * c.rstride));
        }
        return tile_cost_quads<P>(f, q, satd);
    }
    else if constexpr (RK)
    {
        const int xf = cx & 3, yf = cy & 3;
        const EvalCtx *ec = lds_ctx(base);
        const int pitch = uni(ec->rk_pitch);
        LdsLoader<P> ld = { base + uni(ec->rk_win) + (__mul24((cy >> 2) - uni(ec->rk_y0) + ty * 4, pitch) + (cx >> 2) - uni(ec->rk_x0) + tx * 4) * (int)sizeof(P), pitch };
        if ((xf | yf) == 0)
        {
#pragma unroll
            for (int y = 0; y < 4; y++) q[y] = ld(0, y);
            return tile_cost_quads<P>(f, q, true);
        }
        int o[4][4];
        rk_interp_tile<P>(ld, xf, yf, c.depth, o);
        return tile_cost_px<P>(f, o, true);
    }
    else
        return 0;   /* sub-pel points of the classic path go through eval_planes */
}

/* costs[k] = distortion(k) [+ mvcost(mv[k] << sh[k])] for k < K, every (candidate, tile) pair
 * measured independently by one lane. Serves all full-pel batches, and the sub-pel batches
 * of the lookahead (half-pel planes) and Rockchip (4-tap) flavours. */
template<typename P, bool LOWRES, bool RK, bool FULLPEL>
__device__ __forceinline__ void eval_batch_body(int base, int K, int mode_, int addmv)
{
    /* FULLPEL: the caller only ever measures full-pel SADs (mode 0); everything else compiles out */
    const int mode = FULLPEL ? 0 : mode_;
    wave_sync();
    Geo c;
    c.load<LOWRES>(base);
    const int lane = threadIdx.x & 63;
    /* the lambda * bits term of candidate `lane` is gathered first so that its table loads
     * overlap the distortion work */
    const bool mixed = mode < 0;
    const int mvc = gather_mvcost<FULLPEL>(base, K, mode, addmv);
    if (c.Tp <= 64)
    {
        const int g = lane >> c.lTp, t = lane & (c.Tp - 1);
        const int cpp = 64 >> c.lTp;
        if (mode == 0)
        {
            /* full-pel SAD: the lane's source tile is loaded once; two passes are issued back
             * to back so that their reference loads are in flight together */
            const int tt = t < c.T ? t : 0;
            const int ty = __umul24(tt, c.inv_wt) >> 16, tx = tt - __mul24(ty, c.wt);
            typename Px<P>::quad f[4];
#pragma unroll
            for (int y = 0; y < 4; y++) f[y] = Px<P>::load_lds_aligned(base + L_FENC + ((ty * 4 + y) * 64 + tx * 4) * (int)sizeof(P));
            const P AS1 *r0 = (const P AS1*)c.ref[0] + c.roff + (long)__mul24(ty * 4, c.rstride) + tx * 4;
            for (int k0 = 0; k0 < K; k0 += 2 * cpp)
            {
                typename Px<P>::quad q[2][4];
                int kk[2];
#pragma unroll
                for (int u = 0; u < 2; u++)
                {
                    kk[u] = k0 + u * cpp + g;
                    const int kc = kk[u] < K ? kk[u] : K - 1;
                    const P AS1 *r = r0 + (long)__mul24(lds_i(base, L_SY)[kc], c.rstride) + lds_i(base, L_SX)[kc];
#pragma unroll
                    for (int y = 0; y < 4; y++) q[u][y] = Px<P>::load_g(r + y * c.rstride);
                }
#pragma unroll
                for (int u = 0; u < 2; u++)
                {
                    int v = 0;
#pragma unroll
                    for (int y = 0; y < 4; y++) v = Px<P>::sad(f[y], q[u][y], v);
                    v = (kk[u] < K && t < c.T) ? v : 0;
                    v = seg_sum(v, c.lTp);
                    if (t == 0 && kk[u] < K) lds_i(base, L_COSTS)[kk[u]] = v;
                }
            }
        }
        else if constexpr (!FULLPEL)
            for (int k0 = 0; k0 < K; k0 += cpp)
            {
                const int k = k0 + g;
                int v = 0;
                if (k < K && t < c.T) v = tile_cost<P, LOWRES, RK>(base, c, k, t, mixed ? lds_i(base, L_META)[k] & 15 : mode);
                v = seg_sum(v, c.lTp);
                if (t == 0 && k < K) lds_i(base, L_COSTS)[k] = v;
            }
    }
    else if (mode == 0)
    {
        /* more than 64 tiles (T is 128, 192 or 256): the lane's up to four source tiles stay in
         * registers across candidates and all loads of a candidate are issued together */
        typename Px<P>::quad f[4][4];
        int toff[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const int t = lane + 64 * i < c.T ? lane + 64 * i : lane;
            const int ty = __umul24(t, c.inv_wt) >> 16, tx = t - __mul24(ty, c.wt);
            toff[i] = __mul24(ty * 4, c.rstride) + tx * 4;
#pragma unroll
            for (int y = 0; y < 4; y++) f[i][y] = Px<P>::load_lds_aligned(base + L_FENC + ((ty * 4 + y) * 64 + tx * 4) * (int)sizeof(P));
        }
        const P AS1 *r0 = (const P AS1*)c.ref[0] + c.roff;
        const int nt = c.T >> 6;
        for (int k = 0; k < K; k++)
        {
            const P AS1 *r = r0 + (long)uni(lds_i(base, L_SY)[k]) * c.rstride + uni(lds_i(base, L_SX)[k]);
            int v = 0;
            /* two tiles (eight loads) at a time: the registers of all sixteen rows would set the
             * kernel's register count, and with it how many waves a SIMD holds */
#pragma unroll
            for (int i0 = 0; i0 < 4; i0 += 2)
            {
                typename Px<P>::quad q[2][4];
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int y = 0; y < 4; y++) q[i][y] = Px<P>::load_g(r + toff[i0 + i] + y * c.rstride);
#pragma unroll
                for (int i = 0; i < 2; i++)
                {
                    int s = 0;
#pragma unroll
                    for (int y = 0; y < 4; y++) s = Px<P>::sad(f[i0 + i][y], q[i][y], s);
                    v += i0 + i < nt ? s : 0;
                }
            }
            v = seg_sum(v, 6);
            if (lane == 0) lds_i(base, L_COSTS)[k] = v;
        }
    }
    else if constexpr (!FULLPEL)
    {
        for (int k = 0; k < K; k++)
        {
            int v = 0;
            const int km = mixed ? uni(lds_i(base, L_META)[k]) & 15 : mode;
            for (int t = lane; t < c.T; t += 64) v += tile_cost<P, LOWRES, RK>(base, c, k, t, km);
            v = seg_sum(v, 6);
            if (lane == 0) lds_i(base, L_COSTS)[k] = v;
        }
    }
    wave_sync();
    if (lane < K) lds_i(base, L_COSTS)[lane] += mvc;
    wave_sync();
}
template<typename P, bool LOWRES, bool RK>
__device__ __noinline__ void eval_batch(int base, int K, int mode, int addmv) { eval_batch_body<P, LOWRES, RK, false>(base, K, mode, addmv); }

/* ---- sub-pel batches of the classic path: separable interpolation through LDS planes ----
 *
 * subpelCompare (motion.cpp:1651-1689) filters every candidate block on its own. Here the
 * candidates of a batch first share their horizontal pass: all candidates with the same qpel
 * x (and the same kind of first pass) read one "plane" -- the horizontally filtered rows they
 * need, computed once by the whole wave into LDS as int16, column-major so that vertically
 * adjacent samples pair up in a dword. The vertical pass then runs per (candidate, 4x4 tile)
 * lane on v_dot2_i32_i16, straight into the SAD / SATD against the source tile.
 *
 * Plane kinds (xf, yf = fractional parts of the candidate MV):
 *   xf != 0, yf != 0  luma_hps intermediates (ipfilter.cpp:235-278), then luma_vsp (:359-400)
 *   xf != 0, yf == 0  finished luma_hpp pixels (:194-233); the vertical pass is the identity
 *   xf == 0           source pixels; the vertical pass is luma_vpp (:280-319) or the identity
 * The identity is the phase-0 filter {0,0,0,64,0,0,0,0} followed by (x + 32) >> 6, so the
 * vertical pass has one code path with per-lane taps, offset and shift. */

typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false);
}

/* four horizontally adjacent 8-tap sums; s points at the sample 3 to the left of the first */
template<typename P> struct HFilter;
template<> struct HFilter<uint8_t>
{
    static __device__ __forceinline__ void run(const uint8_t AS1 *s, const int *taps, int sum[4])
    {
        /* pixels are biased to signed bytes for v_dot4_i32_i8; the taps of every phase add up to
         * 64, so the bias comes back as + 128 * 64 */
        const uint32_t d0 = Px<uint8_t>::load_g(s) ^ 0x80808080u, d1 = Px<uint8_t>::load_g(s + 4) ^ 0x80808080u,
                       d2 = Px<uint8_t>::load_g(s + 8) ^ 0x80808080u;
        const int tlo = taps[9], thi = taps[10];
        sum[0] = __builtin_amdgcn_sdot4(d0, tlo, __builtin_amdgcn_sdot4(d1, thi, 8192, false), false);
#pragma unroll
        for (int x = 1; x < 4; x++)
            sum[x] = __builtin_amdgcn_sdot4(__builtin_amdgcn_alignbyte(d1, d0, x), tlo,
                                            __builtin_amdgcn_sdot4(__builtin_amdgcn_alignbyte(d2, d1, x), thi, 8192, false), false);
    }
};
template<> struct HFilter<uint16_t>
{
    static __device__ __forceinline__ void run(const uint16_t AS1 *s, const int *taps, int sum[4])
    {
        const uint2 a = Px<uint16_t>::load_g(s), b = Px<uint16_t>::load_g(s + 4), c = Px<uint16_t>::load_g(s + 8);
        const uint32_t D[6] = { a.x, a.y, b.x, b.y, c.x, c.y };
        uint32_t A[5];
#pragma unroll
        for (int i = 0; i < 5; i++) A[i] = __builtin_amdgcn_alignbit(D[i + 1], D[i], 16);
        const uint32_t t0 = taps[0], t1 = taps[1], t2 = taps[2], t3 = taps[3];
        sum[0] = dot2(D[0], t0, dot2(D[1], t1, dot2(D[2], t2, dot2(D[3], t3, 0))));
        sum[1] = dot2(A[0], t0, dot2(A[1], t1, dot2(A[2], t2, dot2(A[3], t3, 0))));
        sum[2] = dot2(D[1], t0, dot2(D[2], t1, dot2(D[3], t2, dot2(D[4], t3, 0))));
        sum[3] = dot2(A[1], t0, dot2(A[2], t1, dot2(A[3], t2, dot2(A[4], t3, 0))));
    }
};

/* Vertical pass + cost for column x of tile t of candidate k against the planes in LDS.
 * The four lanes of a quad hold the four columns of one 4x4 tile. SAD: the lane's share is
 * the sum over its column. SATD: the vertical Hadamard runs inside the lane, the horizontal
 * one across the quad (DPP), and each lane returns the abs-sum of its four coefficients; the
 * caller halves the total (every tile's abs-sum is even, pixel.cpp:226). */
struct ColCand
{
    /* what the vertical pass needs to know about a candidate: its plane (byte offset, row pitch),
     * first plane row of the window of tile row 0, output shift / rounding offset, cost flavour,
     * and the packed taps of its vertical phase */
    int poff, prow, R0, shiftV, offV, satd;
    uint32_t ev[4], od[5];
    template<bool UNIFORM> __device__ __forceinline__ void load(int base, int k, int depth, bool all_ps)
    {
#define U_(v) (UNIFORM ? uni(v) : (v))
        const int cpl = U_(lds_i(base, L_CPL)[k]);
        const int *d = lds_i(base, L_PDESC + (cpl & 255) * 32);
        const int ps = all_ps ? 1 : (U_(d[1]) >> 2) & 1, headRoom = 14 - depth;
        poff = U_(d[3]); prow = U_(d[4]);
        R0 = ((cpl >> 8) & 255) - 16;
        satd = (cpl >> 18) & 1;
        shiftV = ps ? 6 + headRoom : 6;
        offV = ps ? (1 << (5 + headRoom)) + (8192 << 6) : 32;
        const int *tp = lds_i(base, L_TAPS + ((cpl >> 16) & 3) * 48);
#pragma unroll
        for (int i = 0; i < 4; i++) ev[i] = U_(tp[i]);
#pragma unroll
        for (int i = 0; i < 5; i++) od[i] = U_(tp[4 + i]);
#undef U_
    }
};

/* where tile t of a column block sits, and the lane's column of the source tile */
template<typename P>
struct ColTile
{
    int ty, col, f[4];
    __device__ __forceinline__ void load(int base, int t, int x, int wt, int inv_wt, int cb)
    {
        ty = __umul24(t, inv_wt) >> 16;
        col = (t - __mul24(ty, wt)) * 4 + x;
        const P *s = (const P*)(x265amd_smem + base + L_FENC) + (ty * 4) * 64 + cb + col;
#pragma unroll
        for (int y = 0; y < 4; y++) f[y] = (int)s[y * 64];
    }
};

template<typename P>
__device__ __forceinline__ int plane_column_cost(int base, const ColCand &cc, const ColTile<P> &ct, int x, int depth)
{
    const int poff = cc.poff, prow = cc.prow, shiftV = cc.shiftV, offV = cc.offV, maxv = (1 << depth) - 1;
    const uint32_t *ev = cc.ev, *od = cc.od;
    const int R = cc.R0 + 4 * ct.ty;
    const uint32_t sh16 = (R & 1) * 16;
    const uint32_t *col = (const uint32_t*)(x265amd_smem + base + poff + (__mul24(ct.col, prow) + (R & ~1)) * 2);
    uint32_t E[7], e[6];
#pragma unroll
    for (int i = 0; i < 7; i++) E[i] = col[i];
#pragma unroll
    for (int i = 0; i < 6; i++) e[i] = __builtin_amdgcn_alignbit(E[i + 1], E[i], sh16);
    int o[4];
    o[0] = dot2(e[0], ev[0], dot2(e[1], ev[1], dot2(e[2], ev[2], dot2(e[3], ev[3], offV))));
    o[1] = dot2(e[0], od[0], dot2(e[1], od[1], dot2(e[2], od[2], dot2(e[3], od[3], dot2(e[4], od[4], offV)))));
    o[2] = dot2(e[1], ev[0], dot2(e[2], ev[1], dot2(e[3], ev[2], dot2(e[4], ev[3], offV))));
    o[3] = dot2(e[1], od[0], dot2(e[2], od[1], dot2(e[3], od[2], dot2(e[4], od[3], dot2(e[5], od[4], offV)))));
    int pr[4];
#pragma unroll
    for (int y = 0; y < 4; y++) pr[y] = clip_pixel(as_i16(o[y] >> shiftV), maxv);
    if constexpr (sizeof(P) == 1)
    {
        /* 8-bit samples: differences and every Hadamard stage fit int16 (16 * 255), so rows
         * (0, 1) and (2, 3) travel as packed pairs through v_pk_* instructions */
        const s16x2 d01 = __builtin_bit_cast(s16x2, (uint32_t)(ct.f[0] | (ct.f[1] << 16))) - __builtin_bit_cast(s16x2, (uint32_t)(pr[0] | (pr[1] << 16)));
        const s16x2 d23 = __builtin_bit_cast(s16x2, (uint32_t)(ct.f[2] | (ct.f[3] << 16))) - __builtin_bit_cast(s16x2, (uint32_t)(pr[2] | (pr[3] << 16)));
        /* sums of magnitudes on v_sad_u16: a signed 16-bit value c as the unsigned c + 0x8000 is |c| away from 0x8000 */
        const uint32_t bias = 0x80008000u;
        if (!cc.satd)
            return (int)__builtin_amdgcn_sad_u16(__builtin_bit_cast(uint32_t, d01) ^ bias, bias,
                                                 __builtin_amdgcn_sad_u16(__builtin_bit_cast(uint32_t, d23) ^ bias, bias, 0u));
        const s16x2 S = d01 + d23, D = d01 - d23;                       /* (r0+r2, r1+r3), (r0-r2, r1-r3) */
        const s16x2 A = __builtin_shufflevector(S, D, 0, 2), B = __builtin_shufflevector(S, D, 1, 3);
        s16x2 U = A + B, W = A - B;                                     /* vertical coefficients (c0, c2), (c1, c3) */
        const short m1 = (short)(1 - 2 * (x & 1)), m2 = (short)(1 - (x & 2));
        const s16x2 s1 = { m1, m1 }, s2 = { m2, m2 };
        U = U * s1 + __builtin_bit_cast(s16x2, dpp_all<0xB1>(__builtin_bit_cast(int, U)));
        W = W * s1 + __builtin_bit_cast(s16x2, dpp_all<0xB1>(__builtin_bit_cast(int, W)));
        U = U * s2 + __builtin_bit_cast(s16x2, dpp_all<0x4E>(__builtin_bit_cast(int, U)));
        W = W * s2 + __builtin_bit_cast(s16x2, dpp_all<0x4E>(__builtin_bit_cast(int, W)));
        return (int)__builtin_amdgcn_sad_u16(__builtin_bit_cast(uint32_t, U) ^ bias, bias,
                                             __builtin_amdgcn_sad_u16(__builtin_bit_cast(uint32_t, W) ^ bias, bias, 0u));
    }
    int dv[4];
#pragma unroll
    for (int y = 0; y < 4; y++) dv[y] = ct.f[y] - pr[y];
    if (!cc.satd)
        return abs(dv[0]) + abs(dv[1]) + abs(dv[2]) + abs(dv[3]);
    const int a0 = dv[0] + dv[1], a1 = dv[0] - dv[1], a2 = dv[2] + dv[3], a3 = dv[2] - dv[3];
    int c[4] = { a0 + a2, a1 + a3, a0 - a2, a1 - a3 };
    /* butterfly with the lane one column over, then two columns over: the lower lane of a pair
     * keeps the sum, the upper one the difference (own value times -1, 24-bit multiply-add) */
    const int s1 = 1 - 2 * (x & 1), s2 = 1 - (x & 2);
    /* stage by stage over the four values, so that no DPP read waits on the write before it */
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = __mul24(c[i], s1) + dpp_all<0xB1>(c[i]);
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = __mul24(c[i], s2) + dpp_all<0x4E>(c[i]);
    const int s = abs(c[0]) + abs(c[1]) + abs(c[2]) + abs(c[3]);
    return s;
}

/* costs[k] for K sub-pel (or mixed) candidates of the classic path; mode as in eval_batch */
template<typename P>
__device__ __noinline__ void eval_planes(int base, int K, int mode, int addmv)
{
    wave_sync();
    const int lane = threadIdx.x & 63;
    const EvalCtx *cx = lds_ctx(base);
    const int mvc = gather_mvcost(base, K, mode, addmv);
    const P AS1 *ref0 = (const P AS1*)uni64(cx->ref[0]) + (long)uni64((unsigned long long)cx->roff);
    /* 8-bit planes always hold 8-bit samples: shifts, offsets and clamps become constants */
    const int rstride = uni(cx->rstride), w = uni(cx->w), h = uni(cx->h), depth = sizeof(P) == 1 ? 8 : uni(cx->depth);
    const int T = uni(cx->T), lTp = uni(cx->lTp), wt = uni(cx->wt), inv_wt = uni(cx->inv_wt);
    const int planes_off = uni(cx->planes_off), budget = uni(cx->plane_budget);

    /* ---- plan: lane k describes candidate k, lane p collects plane p ---- */
    int qx = 0, qy = 0, cm = 0;
    if (lane < K)
    {
        qx = lds_i(base, L_SX)[lane]; qy = lds_i(base, L_SY)[lane];
        cm = mode < 0 ? lds_i(base, L_META)[lane] & 15 : mode;
        if (cm == 0) { qx <<= 2; qy <<= 2; }   /* full-pel member of a mixed batch */
    }
    const int yf = qy & 3, Y = qy >> 2;

    /* ---- candidates at integer positions (the rounded predictor, MV(0,0), the centre of the
     * first refinement round, a refinement point that lands on a sample) need no interpolation:
     * one lane per (candidate, tile) reads the reference itself, and the planes below are built
     * for the others only. SATD is halved per tile here, which equals halving the sum; the lone
     * SATD candidate of a small PU (the centre) stays with the planes, where it rides along ---- */
    const bool direct = lane < K && ((qx | qy) & 3) == 0 && (cm != 2 || T > 16);   /* a small PU's SATD costs no more in the planes */
    {
        const unsigned long long dmask = __ballot(direct);
        if (dmask)
        {
            const int nd = (int)__builtin_popcountll(dmask);
            if (direct) lds_i(base, L_CLIST)[__builtin_popcountll(dmask & ((1ull << lane) - 1))] = lane;
            wave_sync();
            if (T <= 64)
            {
                const int g = lane >> lTp, t = lane & ((1 << lTp) - 1), cpp = 64 >> lTp, tt = t < T ? t : 0;
                const int ty = __umul24(tt, inv_wt) >> 16, tx = tt - __mul24(ty, wt);
                typename Px<P>::quad f[4];
#pragma unroll
                for (int y = 0; y < 4; y++) f[y] = Px<P>::load_lds_aligned(base + L_FENC + ((ty * 4 + y) * 64 + tx * 4) * (int)sizeof(P));
                const P AS1 *r0 = ref0 + __mul24(ty * 4, rstride) + tx * 4;
                for (int j0 = 0; j0 < nd; j0 += cpp)
                {
                    const int j = j0 + g;
                    const bool act = j < nd && t < T;
                    const int k = lds_i(base, L_CLIST)[act ? j : 0];
                    const P AS1 *r = r0 + __mul24(__shfl(Y, k), rstride) + __shfl(qx >> 2, k);
                    typename Px<P>::quad q[4];
#pragma unroll
                    for (int y = 0; y < 4; y++) q[y] = Px<P>::load_g(r + y * rstride);
                    int v = tile_cost_quads<P>(f, q, __shfl(cm, k) == 2);
                    v = seg_sum(act ? v : 0, lTp);
                    if (act && t == 0) lds_i(base, L_COSTS)[k] = v;
                }
            }
            else
                for (int j = 0; j < nd; j++)
                {
                    const int k = uni(lds_i(base, L_CLIST)[j]);
                    const P AS1 *r0 = ref0 + (long)__builtin_amdgcn_readlane(Y, k) * rstride + __builtin_amdgcn_readlane(qx >> 2, k);
                    const bool satd = __builtin_amdgcn_readlane(cm, k) == 2;
                    int v = 0;
                    for (int t = lane; t < T; t += 64)
                    {
                        const int ty = __umul24(t, inv_wt) >> 16, tx = t - __mul24(ty, wt);
                        const P AS1 *r = r0 + __mul24(ty * 4, rstride) + tx * 4;
                        typename Px<P>::quad f[4], q[4];
#pragma unroll
                        for (int y = 0; y < 4; y++)
                        {
                            f[y] = Px<P>::load_lds_aligned(base + L_FENC + ((ty * 4 + y) * 64 + tx * 4) * (int)sizeof(P));
                            q[y] = Px<P>::load_g(r + y * rstride);
                        }
                        v += tile_cost_quads<P>(f, q, satd);
                    }
                    v = seg_sum(v, 6);
                    if (lane == 0) lds_i(base, L_COSTS)[k] = v;
                }
            wave_sync();
        }
    }

    /* With 8-bit samples the luma_hps intermediate keeps every bit (shift 0), and luma_vsp with the
     * phase-0 taps on it gives exactly (sum + 32) >> 6; likewise phase-0 luma_hps of a sample is
     * 64 * sample - 8192 and luma_vsp on that equals luma_vpp. So all candidates of one qpel x can
     * share one plane of hps intermediates, whatever their fractions. Deeper samples lose two bits
     * in the intermediate and keep the three plane kinds. */
    const int key = (qx << 1) | (sizeof(P) == 1 ? 1 : ((qx & 3) && yf ? 1 : 0));
    /* Candidates share a plane when they have the same key and their first rows fall into the
     * same pair (base + 2g, base + 2g + 1), base being the smallest first row of the batch: the
     * candidates of a refinement round span two rows at most, so each of their columns gets one
     * plane. One pass per plane: the lowest unplaced candidate names the plane, a ballot finds
     * its companions. Which candidates share a plane changes the work, never the costs. */
    int pkey = 0, pymin = 0, pymax = 0, papron = 0, mypl = 0, np = 0;
    {
        const bool isc = lane < K && !direct;
        const int ybase = wave_min(isc ? Y : 0x7fffffff);
        const int grp = (Y - ybase) >> 1, odd = (Y - ybase) & 1;
        unsigned long long rem = __ballot(isc);
        while (rem)
        {
            const int lead = (int)__builtin_ctzll(rem);
            const int kk = __builtin_amdgcn_readlane(key, lead), kg = __builtin_amdgcn_readlane(grp, lead);
            const bool mine = isc && key == kk && grp == kg;
            const unsigned long long m = __ballot(mine);
            const bool lo = __ballot(mine && !odd) != 0, hi = __ballot(mine && odd) != 0, ap = __ballot(mine && yf != 0) != 0;
            if (mine) mypl = np;
            if (lane == np)
            {
                pkey = kk;
                pymin = ybase + 2 * kg + (lo ? 0 : 1);
                pymax = ybase + 2 * kg + (hi ? 1 : 0);
                papron = ap;
            }
            rem &= ~m;
            np++;
        }
    }
    const int pY0 = pymin - (papron ? 3 : 0);
    const int pnrows = pymax - pymin + h + (papron ? 7 : 0);
    const int pprow = (pnrows + 2) & ~1;
    {
        const int Y0 = __shfl(pY0, mypl);
        if (lane < K) lds_i(base, L_CPL)[lane] = direct ? 0 : mypl | ((Y - 3 - Y0 + 16) << 8) | (yf << 16) | ((cm == 2 ? 1 : 0) << 18);
    }

    const int headRoom = 14 - depth, maxv = (1 << depth) - 1;
    const int shiftH = 6 - headRoom, offH = -(8192 << shiftH);
    /* PUs wider than PLANE_COLS are worked through in column blocks, which bounds the LDS a plane
     * needs; the candidates' costs add up over the blocks */
    for (int cb = 0; cb < w; cb += PLANE_COLS)
    {
        const int cw = w - cb < PLANE_COLS ? w - cb : PLANE_COLS;
        const int wtb = cw >> 2, inv_wtb = wtb == 4 ? 16384 : (wtb == 2 ? 32768 : (wtb == 1 ? 65536 : 21846)), Tb = wtb * (h >> 2);
        int lTb = 0;
        while ((1 << lTb) < Tb) lTb++;
        const int pbytes = cw * pprow * 2, pitems = pnrows * wtb;
        for (int p0 = 0; p0 < np;)
        {
            /* ---- the planes that fit in LDS together ---- */
            int acc = 0, items = 0, p1 = p0, my_poff = 0, my_istart = 0;
            const int b0 = __builtin_amdgcn_readlane(pbytes, p0), n0 = __builtin_amdgcn_readlane(pitems, p0);
            if (__ballot(lane >= p0 && lane < np && (pbytes != b0 || pitems != n0)) == 0)
            {
                /* planes of one size (every refinement round): no running sums needed */
                int n = 1;
                while (p0 + n < np && (n + 1) * b0 <= budget) n++;
                p1 = p0 + n;
                items = n * n0;
                my_poff = __mul24(lane - p0, b0); my_istart = __mul24(lane - p0, n0);
            }
            else
            while (p1 < np)
            {
                const int b = __builtin_amdgcn_readlane(pbytes, p1);
                if (p1 > p0 && acc + b > budget) break;
                if (lane == p1) { my_poff = acc; my_istart = items; }
                acc += b;
                items += __builtin_amdgcn_readlane(pitems, p1);
                p1++;
            }
            if (lane >= p0 && lane < p1)
            {
                int *d = lds_i(base, L_PDESC + lane * 32);
                const int pqx = pkey >> 1;
                d[0] = __mul24(pY0, rstride) + (pqx >> 2) - 3 + cb;
                d[1] = (pqx & 3) | ((pkey & 1) << 2);
                d[2] = my_istart; d[3] = planes_off + my_poff; d[4] = pprow;
            }
            const bool inch = lane < K && !direct && mypl >= p0 && mypl < p1;
            const unsigned long long cmask = __ballot(inch);
            const int nc = __builtin_popcountll(cmask);
            if (inch) lds_i(base, L_CLIST)[__builtin_popcountll(cmask & ((1ull << lane) - 1))] = lane;
            wave_sync();

            /* ---- horizontal pass: one item = 4 adjacent samples of one plane row ---- */
            if (p1 == p0 + 1)
            {
                /* a single plane (the large-PU case): its description stays in scalar registers */
                const int *d = lds_i(base, L_PDESC + p0 * 32);
                const int src = uni(d[0]), xfps = uni(d[1]), poff = uni(d[3]), prow = uni(d[4]);
                const int *tp = lds_i(base, L_TAPS + (xfps & 3) * 48);
                int taps[11];
#pragma unroll
                for (int i = 0; i < 11; i++) taps[i] = uni(tp[i]);
                const bool ps = sizeof(P) == 1 || (xfps & 4);   /* 8-bit: every plane holds hps intermediates */
                const int off = ps ? offH : 32, sh = ps ? shiftH : 6, lo = ps ? -32768 : 0, hi = ps ? 32767 : maxv;
                for (int i0 = 0; i0 < items; i0 += 64)
                {
                    const bool act = i0 + lane < items;
                    const int i = act ? i0 + lane : items - 1;
                    const int r = __umul24(i, inv_wtb) >> 16, q4 = i - __mul24(r, wtb);
                    int sum[4];
                    HFilter<P>::run(ref0 + src + (long)__mul24(r, rstride) + 4 * q4, taps, sum);
                    int16_t *dst = (int16_t*)(x265amd_smem + base + poff) + __mul24(4 * q4, prow) + r;
                    if (act)
#pragma unroll
                        for (int x = 0; x < 4; x++)
                        {
                            int v = as_i16((sum[x] + off) >> sh);
                            v = v < lo ? lo : (v > hi ? hi : v);
                            dst[x * prow] = (int16_t)v;
                        }
                }
            }
            else
                for (int i0 = 0; i0 < items; i0 += 64)
                {
                    const bool act = i0 + lane < items;
                    const int i = act ? i0 + lane : items - 1;
                    int p = p0;
                    for (int q = p0 + 1; q < p1; q++) p += i >= __builtin_amdgcn_readlane(my_istart, q) ? 1 : 0;
                    const int *d = lds_i(base, L_PDESC + p * 32);
                    const int xfps = d[1], li = i - d[2], poff = d[3], prow = d[4];
                    const int r = __umul24(li, inv_wtb) >> 16, q4 = li - __mul24(r, wtb);
                    int sum[4];
                    HFilter<P>::run(ref0 + d[0] + (long)__mul24(r, rstride) + 4 * q4, lds_i(base, L_TAPS + (xfps & 3) * 48), sum);
                    const bool ps = sizeof(P) == 1 || (xfps & 4);   /* 8-bit: every plane holds hps intermediates */
                    const int off = ps ? offH : 32, sh = ps ? shiftH : 6, lo = ps ? -32768 : 0, hi = ps ? 32767 : maxv;
                    int16_t *dst = (int16_t*)(x265amd_smem + base + poff) + __mul24(4 * q4, prow) + r;
                    if (act)
#pragma unroll
                        for (int x = 0; x < 4; x++)
                        {
                            int v = as_i16((sum[x] + off) >> sh);
                            v = v < lo ? lo : (v > hi ? hi : v);
                            dst[x * prow] = (int16_t)v;
                        }
                }
            wave_sync();

            /* ---- vertical pass + cost: one lane per (candidate, tile, column) ---- */
            if (lTb <= 4)
            {
                const int l4 = lTb + 2, g = lane >> l4, t = (lane >> 2) & ((1 << lTb) - 1), cpp = 64 >> l4;
                /* the lane keeps its tile and source column over all passes */
                ColTile<P> ct;
                ct.load(base, t < Tb ? t : 0, lane & 3, wtb, inv_wtb, cb);
                for (int j0 = 0; j0 < nc; j0 += cpp)
                {
                    const int j = j0 + g;
                    const bool act = j < nc && t < Tb;
                    const int k = lds_i(base, L_CLIST)[act ? j : 0];
                    ColCand cc;
                    cc.load<false>(base, k, depth, sizeof(P) == 1);
                    int v = plane_column_cost<P>(base, cc, ct, lane & 3, depth);
                    v = seg_sum(act ? v : 0, l4);
                    if (act && (lane & ((1 << l4) - 1)) == 0) lds_i(base, L_COSTS)[k] = (cb ? lds_i(base, L_COSTS)[k] : 0) + v;
                }
            }
            else
                /* more than 16 tiles in the block: 16 tiles per pass */
                for (int j = 0; j < nc; j++)
                {
                    const int k = uni(lds_i(base, L_CLIST)[j]);
                    ColCand cc;   /* the same for every lane: kept in scalar registers */
                    cc.load<true>(base, k, depth, sizeof(P) == 1);
                    int v = 0;
                    for (int t0 = 0; t0 < Tb; t0 += 16)
                    {
                        const int t = t0 + (lane >> 2);
                        ColTile<P> ct;
                        ct.load(base, t < Tb ? t : 0, lane & 3, wtb, inv_wtb, cb);
                        const int c = plane_column_cost<P>(base, cc, ct, lane & 3, depth);
                        v += t < Tb ? c : 0;
                    }
                    v = seg_sum(v, 6);
                    if (lane == 0) lds_i(base, L_COSTS)[k] = (cb ? lds_i(base, L_COSTS)[k] : 0) + v;
                }
            wave_sync();
            p0 = p1;
        }
    }
    /* SATD: every tile's abs-sum is even, so halving the total equals halving per tile */
    if (lane < K) lds_i(base, L_COSTS)[lane] = (lds_i(base, L_COSTS)[lane] >> ((lds_i(base, L_CPL)[lane] >> 18) & 1)) + mvc;
    wave_sync();
}

/* The fields of a job as wave-uniform scalars. Jobs may sit in LDS or behind a generic
 * pointer, where the compiler cannot see that every lane reads the same value. */
struct JobRegs
{
    int pux, puy, mnx, mny, mxx, mxy, qmx, qmy, numc, merange;
    int cx[4], cy[4];
    __device__ __forceinline__ void load(const x265amd_job *job)
    {
        const int *p = (const int*)job;
        const int d0 = uni(p[0]), d1 = uni(p[1]), d2 = uni(p[2]), d3 = uni(p[3]), d4 = uni(p[4]);
        pux = as_i16(d0); puy = d0 >> 16;
        numc = (d1 >> 16) & 0xff;
        mnx = as_i16(d2); mny = d2 >> 16; mxx = as_i16(d3); mxy = d3 >> 16; qmx = as_i16(d4); qmy = d4 >> 16;
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const int d = uni(p[5 + i]);
            cx[i] = as_i16(d); cy[i] = d >> 16;
        }
        merange = uni(p[9]);
    }
    __device__ __forceinline__ int mvcx(int i) const { return i == 0 ? cx[0] : (i == 1 ? cx[1] : (i == 2 ? cx[2] : cx[3])); }
    __device__ __forceinline__ int mvcy(int i) const { return i == 0 ? cy[0] : (i == 1 ? cy[1] : (i == 2 ? cy[2] : cy[3])); }
};

/* what the stages of one search hand to each other: the full-pel best (bmv, bcost), the best
 * qpel predictor (bestpre, bprecost) and the rounded predictor pmv (motion.cpp:575-621) */
struct SearchState
{
    int bmx, bmy, bcost, bprecost, bpx, bpy, pmx, pmy;
};

/* LOWRES selects the lookahead flavour at compile time (half-pel planes instead of
 * interpolation), which keeps the 8-tap code out of the lookahead kernel. */
/* LEAN: a searcher that only ever measures full-pel SADs (the continuation of a STAR search whose start-up a
 * fixed-shape kernel has done, and whose refinement another will do): its evaluations are inlined, so the kernel
 * makes no calls and needs no stack. */
template<typename P, int ROWS = 64, bool LOWRES = false, bool RK = false, bool LEAN = false>
struct Searcher
{
    int base;           /* byte offset of this wave's LDS block */
    int w, h, lane;
    int mvpx, mvpy;
    unsigned nsad, nsub;

    __device__ __forceinline__ void eval(int K, int mode, bool addmv)
    {
        if constexpr (LEAN) { eval_batch_body<P, false, false, true>(base, K, 0, 1); nsad += K; return; }
        /* sub-pel (and mixed) batches of the classic path interpolate through LDS planes */
        if (!LOWRES && !RK && mode != 0) eval_planes<P>(base, K, mode, addmv ? 1 : 0);
        else eval_batch<P, LOWRES, RK>(base, K, mode, addmv ? 1 : 0);
        if (mode == 0) nsad += K; else if (mode > 0) nsub += K;
    }

    /* First minimum of costs[k0..k1) in index order if it beats bcost (the reference's chain of
     * strict '<' tests); returns its index or -1. Costs stay below 2^25, so cost << 6 | index
     * orders candidates exactly like the sequential scan. */
    __device__ __forceinline__ int fold_range(int k0, int k1, int &bcost)
    {
        int key = 0x7fffffff;
        if (lane >= k0 && lane < k1) key = (lds_i(base, L_COSTS)[lane] << 6) | lane;
        key = k1 <= 16 ? uni(row_min16(key)) : wave_min(key);
        if ((key >> 6) < bcost) { bcost = key >> 6; return key & 63; }
        return -1;
    }
    __device__ __forceinline__ int fold_min(int K, int &bcost) { return fold_range(0, K, bcost); }

    /* queue one candidate per lane that has one (`ok`), in lane order; returns their number */
    __device__ __forceinline__ int put_lanes(bool ok, int sx_, int sy_, int mx_, int my_, int sh_, int meta_)
    {
        const unsigned long long m = __ballot(ok);
        if (ok)
        {
            const int k = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
            lds_i(base, L_SX)[k] = sx_; lds_i(base, L_SY)[k] = sy_;
            lds_i(base, L_MVX)[k] = mx_; lds_i(base, L_MVY)[k] = my_;
            lds_i(base, L_SH)[k] = sh_; lds_i(base, L_META)[k] = meta_;
        }
        return (int)__builtin_popcountll(m);
    }

    __device__ __forceinline__ void put(int k, int sx_, int sy_, int mx_, int my_, int sh_, int meta_)
    {
        lds_i(base, L_SX)[k] = sx_; lds_i(base, L_SY)[k] = sy_;
        lds_i(base, L_MVX)[k] = mx_; lds_i(base, L_MVY)[k] = my_;
        lds_i(base, L_SH)[k] = sh_; lds_i(base, L_META)[k] = meta_;
    }

    __device__ __forceinline__ int cost_of(int k) const { return uni(lds_i(base, L_COSTS)[k]); }
    __device__ __forceinline__ int mvx_of(int k) const { return uni(lds_i(base, L_MVX)[k]); }
    __device__ __forceinline__ int mvy_of(int k) const { return uni(lds_i(base, L_MVY)[k]); }
    __device__ __forceinline__ int meta_of(int k) const { return uni(lds_i(base, L_META)[k]); }

    /* setMVP (bitcost.h:43) */
    __device__ __forceinline__ void set_mvp(int x, int y)
    {
        mvpx = x; mvpy = y;
        wave_sync();
        if (lane == 0) { lds_ctx(base)->mvpx = x; lds_ctx(base)->mvpy = y; }
    }

    /* single evaluations */
    __device__ __forceinline__ int fpcost(int mx, int my)
    {
        put(0, mx, my, mx, my, 2, 0);
        eval(1, 0, true);
        return cost_of(0);
    }
    __device__ __forceinline__ int qdist(int qx, int qy, int mode)
    {
        put(0, qx, qy, qx, qy, 0, 0);
        eval(1, mode, false);
        return cost_of(0);
    }

    __device__ __forceinline__ bool in_range(int x, int y, int mnx, int mny, int mxx, int mxy) const
    {
        return x >= mnx && x <= mxx && y >= mny && y <= mxy;
    }

    /* Evaluate the K queued full-pel candidates (meta = point | dist << 8) and fold
     * them into the running best in queue order; COST_MV_PT_DIST, motion.cpp:182-194 */
    __device__ __forceinline__ void fold_pt(int K, int &bmx, int &bmy, int &bcost, int &bPointNr, int &bDistance)
    {
        if (!K) return;
        eval(K, 0, true);
        const int k = fold_min(K, bcost);
        if (k >= 0)
        {
            bmx = mvx_of(k); bmy = mvy_of(k);
            const int m = meta_of(k);
            bPointNr = m & 0xff; bDistance = m >> 8;
        }
    }

    /* first lane of star round r when all rounds are laid out back to back: round 0 (distance 1)
     * has 4 points, rounds 1-3 (distance 2, 4, 8) have 8, later rounds 16 */
    static __device__ __forceinline__ int star_start(int r) { return r == 0 ? 0 : (r <= 4 ? 8 * r - 4 : 16 * r - 36); }

    /* StarPatternSearch, motion.cpp:307-547. All rounds measure around the same centre and
     * only look at the running best to count rounds without improvement, so the rounds that
     * are certain to run (earlyExitIters minus the current count) form ONE batch: lane i takes
     * point i of the concatenated rounds, points outside the search range drop out, and the
     * per-round bookkeeping is replayed on the costs in visit order. */
    /* (ox, oy): the centre the rounds are measured around (the reference's omv = bmv on entry); a search that a
     * fixed-shape kernel began continues at round ra0 with rounds0 rounds without improvement behind it. */
    __device__ __forceinline__ void star_pattern(int mnx, int mny, int mxx, int mxy, int &bmx, int &bmy, int &bcost,
                                 int &bPointNr, int &bDistance, int earlyExitIters, int merange,
                                 int ox, int oy, int ra0 = 0, int rounds0 = 0)
    {
        /* point offsets in units of distance (round 0), distance / 2 (rounds 1-3, visit order
         * 2,1,3,4,5,6,8,7), distance / 4 (later rounds); motion.cpp:330-540 */
        constexpr unsigned long long X4 = nibs({ 0, -1, 1, 0 }), Y4 = nibs({ -1, 0, 0, 1 }), P4 = nibs({ 2, 4, 5, 7 });
        constexpr unsigned long long X8 = nibs({ 0, -1, 1, -2, 2, -1, 1, 0 }), Y8 = nibs({ -2, -1, -1, 0, 0, 1, 1, 2 }),
                                     P8 = nibs({ 2, 1, 3, 4, 5, 6, 8, 7 });
        constexpr unsigned long long X16 = nibs({ 0, -4, 4, 0, -1, 1, -1, 1, -2, 2, -2, 2, -3, 3, -3, 3 }),
                                     Y16 = nibs({ -4, 0, 0, 4, -3, -3, 3, 3, -2, -2, 2, 2, -1, -1, 1, 1 });
        int rl = 3;
        for (int d = 16; d <= as_i16(merange) && d > 0; d = as_i16(d << 1)) rl++;
        int rounds = rounds0;
        for (int ra = ra0; ra <= rl;)
        {
            int rb = ra + (earlyExitIters - rounds) - 1;
            if (rb > rl) rb = rl;
            while (star_start(rb + 1) - star_start(ra) > MAXC) rb--;
            const int local = lane + star_start(ra);
            int r, i;
            if (local < 4) { r = 0; i = local; }
            else if (local < 28) { r = 1 + ((local - 4) >> 3); i = (local - 4) & 7; }
            else { r = 4 + ((local - 28) >> 4); i = (local - 28) & 15; }
            const int d = 1 << r;
            const int ux = nib(r == 0 ? X4 : (r < 4 ? X8 : X16), i), uy = nib(r == 0 ? Y4 : (r < 4 ? Y8 : Y16), i);
            const int unit = r == 0 ? 1 : (r < 4 ? d >> 1 : d >> 2);
            const int dx = ux * unit, dy = uy * unit;
            const int pt = r == 0 ? nib(P4, i) : (r < 4 ? nib(P8, i) : 0);
            const int dd = (r >= 1 && r < 4 && (ux & 1)) ? d >> 1 : d;
            const int x = as_i16(ox + dx), y = as_i16(oy + dy);
            /* the reference tests only the bounds a point can cross */
            const bool ok = r <= rb && (dy >= 0 || y >= mny) && (dy <= 0 || y <= mxy) && (dx >= 0 || x >= mnx) && (dx <= 0 || x <= mxx);
            const unsigned long long m = __ballot(ok);
            const int K = put_lanes(ok, x, y, x, y, 2, pt | (dd << 8));
            if (K) eval(K, 0, true);
            int k0 = 0;
            for (int r2 = ra; r2 <= rb; r2++)
            {
                const int n2 = r2 == 0 ? 4 : (r2 < 4 ? 8 : 16);
                const int n = (int)__builtin_popcountll(m & (((1ull << n2) - 1) << (star_start(r2) - star_start(ra))));
                const int saved = bcost;
                const int k = fold_range(k0, k0 + n, bcost);
                if (k >= 0)
                {
                    bmx = mvx_of(k); bmy = mvy_of(k);
                    const int mt = meta_of(k);
                    bPointNr = mt & 0xff; bDistance = mt >> 8;
                }
                k0 += n;
                if (bcost < saved) rounds = 0;
                else if (++rounds >= earlyExitIters) return;
            }
            ra = rb + 1;
        }
    }

    /* Two extra points next to a distance-1 winner, motion.cpp:918-946 / :992-1015 */
    __device__ __forceinline__ void two_point(int bPointNr, int mnx, int mny, int mxx, int mxy,
                                              int &bmx, int &bmy, int &bcost)
    {
        const int i0 = (bPointNr - 1) * 2;
        const int x1 = as_i16(bmx + kTwoPoint[i0][0]), y1 = as_i16(bmy + kTwoPoint[i0][1]);
        const int x2 = as_i16(bmx + kTwoPoint[i0 + 1][0]), y2 = as_i16(bmy + kTwoPoint[i0 + 1][1]);
        int K = 0;
        /* the second point is measured relative to the ORIGINAL bmv and folded after
         * the first, exactly as the two COST_MV calls do */
        if (in_range(x1, y1, mnx, mny, mxx, mxy)) { put(K, x1, y1, x1, y1, 2, 0); K++; }
        if (in_range(x2, y2, mnx, mny, mxx, mxy)) { put(K, x2, y2, x2, y2, 2, 0); K++; }
        int pn = 0, pd = 0;
        fold_pt(K, bmx, bmy, bcost, pn, pd);
    }

    /* candidate queue folded in order; used where the reference evaluates a long run of
     * candidates without looking at the running best in between */
    __device__ __forceinline__ void q_flush(int &K, int &bmx, int &bmy, int &bcost)
    {
        int pn = 0, pd = 0;
        fold_pt(K, bmx, bmy, bcost, pn, pd);
        bmx = as_i16(bmx); bmy = as_i16(bmy);
        K = 0;
    }
    __device__ __forceinline__ void q_add(int &K, int sx_, int sy_, int mx_, int my_, int sh_, int &bmx, int &bmy, int &bcost)
    {
        if (K == MAXC) q_flush(K, bmx, bmy, bcost);
        put(K, sx_, sy_, mx_, my_, sh_, 0);
        K++;
    }

    /* X265_STAR_SEARCH, motion.cpp:911-1019, from round ra0 of its first pattern on (0: all of it). */
    __device__ __forceinline__ void star_rest(int mnx, int mny, int mxx, int mxy, int merange, int ox, int oy, int ra0, int rounds0,
                                              int &bmx, int &bmy, int &bcost, int &bPointNr, int &bDistance)
    {
        star_pattern(mnx, mny, mxx, mxy, bmx, bmy, bcost, bPointNr, bDistance, 3, merange, ox, oy, ra0, rounds0);
        if (bDistance == 1)
        {
            if (!bPointNr) return;
            const int saved = bcost;
            two_point(bPointNr, mnx, mny, mxx, mxy, bmx, bmy, bcost);
            if (bcost == saved) return;
        }

        if (bDistance > 5)
        {
            /* raster refinement with step 5 (motion.cpp:952-990): row-major scan; within a row the
             * points go in groups of four while a whole group fits, and the MV bits of a
             * group's 4th point use tmv << 3 (:975). One lane per point. */
            const int nx = (mxx - mnx) / 5 + 1;
            const int grouped = mxx - mnx >= 15 ? ((mxx - mnx - 15) / 20 + 1) * 4 : 0;
            int pn, pd;
            if (nx <= MAXC)
            {
                const int rpb = MAXC / nx, inv = (65536 + nx - 1) / nx;
                const int rs = (lane * inv) >> 16, j = lane - rs * nx;
                for (int ty = mny; ty <= mxy; ty += 5 * rpb)
                {
                    const int y = ty + 5 * rs, x = mnx + 5 * j;
                    const int K = put_lanes(rs < rpb && y <= mxy, x, y, as_i16(x), y, (j < grouped && (j & 3) == 3) ? 3 : 2, 0);
                    fold_pt(K, bmx, bmy, bcost, pn, pd);
                }
            }
            else
                for (int ty = mny; ty <= mxy; ty += 5)
                    for (int j0 = 0; j0 < nx; j0 += MAXC)
                    {
                        const int j = j0 + lane, x = mnx + 5 * j;
                        const int K = put_lanes(j < nx, x, ty, as_i16(x), ty, (j < grouped && (j & 3) == 3) ? 3 : 2, 0);
                        fold_pt(K, bmx, bmy, bcost, pn, pd);
                    }
        }

        while (bDistance > 0)
        {
            bDistance = 0;
            bPointNr = 0;
            star_pattern(mnx, mny, mxx, mxy, bmx, bmy, bcost, bPointNr, bDistance, 32, merange, bmx, bmy);
            if (bDistance == 1)
            {
                if (!bPointNr) break;
                two_point(bPointNr, mnx, mny, mxx, mxy, bmx, bmy, bcost);
                break;
            }
        }
    }

    /* X265_DIA_SEARCH, motion.cpp:628-649: rounds of four points (N, S, W, E; the packed (cost << 4) + dir compare is
     * a strict '<' scan in that order) until none improves, the count runs out or the centre leaves the range.
     * first = false: the first round has been measured and has moved the centre already. */
    __device__ __forceinline__ void dia_rest(int mnx, int mny, int mxx, int mxy, int merange, bool first, int &bmx, int &bmy, int &bcost)
    {
        constexpr unsigned long long SQX = nibs({ 0, 0, 0, -1, 1, -1, -1, 1, 1 }), SQY = nibs({ 0, -1, 1, 0, 0, -1, 1, -1, 1 });
        int i = merange;
        if (!first && !(--i && in_range(bmx, bmy, mnx, mny, mxx, mxy))) return;
        do
        {
            put_lanes(lane < 4, bmx + nib(SQX, lane + 1), bmy + nib(SQY, lane + 1),
                      as_i16(bmx + nib(SQX, lane + 1)), as_i16(bmy + nib(SQY, lane + 1)), 2, 0);
            eval(4, 0, true);
            const int dir = fold_min(4, bcost);
            if (dir < 0) break;
            bmx = mvx_of(dir); bmy = mvy_of(dir);
        }
        while (--i && in_range(bmx, bmy, mnx, mny, mxx, mxy));
    }

    /* the half hexagons of X265_HEX_SEARCH after the first ring has moved the centre in direction `dir`
     * (motion.cpp:687-707); points are measured without a range check, as in the reference */
    __device__ __forceinline__ void hex_walk(int mnx, int mny, int mxx, int mxy, int merange, int dir, int &bmx, int &bmy, int &bcost)
    {
        constexpr unsigned long long H2X = nibs({ -1, -2, -1, 1, 2, 1, -1, -2 }), H2Y = nibs({ -2, 0, 2, 2, 0, -2, -2, 0 });
        for (int i = (merange >> 1) - 1; i > 0 && in_range(bmx, bmy, mnx, mny, mxx, mxy); i--)
        {
            put_lanes(lane < 3, bmx + nib(H2X, dir + lane), bmy + nib(H2Y, dir + lane),
                      as_i16(bmx + nib(H2X, dir + lane)), as_i16(bmy + nib(H2Y, dir + lane)), 2, 0);
            eval(3, 0, true);
            const int step = fold_min(3, bcost) + 1;
            if (!step) break;
            dir += step - 2;
            dir = kMod6m1[dir + 1];
            bmx = as_i16(bmx + kHex2[dir + 1][0]); bmy = as_i16(bmy + kHex2[dir + 1][1]);
        }
    }

    /* the closing square of X265_HEX_SEARCH, motion.cpp:710-722 */
    __device__ __forceinline__ void hex_square(int &bmx, int &bmy, int &bcost)
    {
        constexpr unsigned long long SQX = nibs({ 0, 0, 0, -1, 1, -1, -1, 1, 1 }), SQY = nibs({ 0, -1, 1, 0, 0, -1, 1, -1, 1 });
        put_lanes(lane < 8, bmx + nib(SQX, lane + 1), bmy + nib(SQY, lane + 1),
                  as_i16(bmx + nib(SQX, lane + 1)), as_i16(bmy + nib(SQY, lane + 1)), 2, 0);
        eval(8, 0, true);
        const int dir = fold_min(8, bcost) + 1;
        bmx = as_i16(bmx + kSquare1[dir][0]); bmy = as_i16(bmy + kSquare1[dir][1]);
    }

    /* The rest of a DIA or HEX search whose start-up candidates and first round / ring a fixed-shape kernel has measured
     * (me_small.h) and which moved the centre: `state` as resume_star reads it, with the centre after the move, its
     * cost and -- HEX -- the direction of the move. Leaves the record as search_staged does. */
    __device__ __forceinline__ void resume_pattern(const x265amd_job *job, const int *state, int method, x265amd_result *res)
    {
        if (method == X265AMD_STAR) { resume_star(job, state, res); return; }
        JobRegs J;
        J.load(job);
        set_mvp(J.qmx, J.qmy);
        const int s1 = uni(state[1]), s3 = uni(state[3]), s5 = uni(state[5]);
        SearchState st;
        int bmx = as_i16(s1), bmy = s1 >> 16, bcost = uni(state[2]);
        st.bprecost = uni(state[4]); st.bpx = as_i16(s5); st.bpy = s5 >> 16;
        nsad = (unsigned)uni(state[6]); nsub = (unsigned)uni(state[7]);
        if (method == X265AMD_HEX)
        {
            hex_walk(J.mnx, J.mny, J.mxx, J.mxy, J.merange, s3 & 0xff, bmx, bmy, bcost);
            hex_square(bmx, bmy, bcost);
        }
        else
            dia_rest(J.mnx, J.mny, J.mxx, J.mxy, J.merange, false, bmx, bmy, bcost);
        st.bmx = bmx; st.bmy = bmy; st.bcost = bcost;
        int qx, qy, c;
        choose(st, qx, qy, c);
        if (lane == 0)
        {
            res->mv[0] = (int16_t)qx; res->mv[1] = (int16_t)qy;
            res->cost = c;
            res->sad_calls = nsad; res->subpel_calls = nsub;
        }
    }

    /* The rest of a STAR search whose start-up candidates and first three pattern rounds a fixed-shape kernel has
     * measured and folded (me_small.h): `state` holds what the reference has in hand at that point. Leaves the
     * choice of motion.cpp:1063-1069 and the counters in the result record, as search_staged does. */
    __device__ __forceinline__ void resume_star(const x265amd_job *job, const int *state, x265amd_result *res)
    {
        JobRegs J;
        J.load(job);
        set_mvp(J.qmx, J.qmy);
        const int s0 = uni(state[0]), s1 = uni(state[1]), s3 = uni(state[3]), s5 = uni(state[5]);
        SearchState st;
        int bmx = as_i16(s1), bmy = s1 >> 16, bcost = uni(state[2]);
        int bPointNr = s3 & 0xff, bDistance = (s3 >> 8) & 0xff;
        st.bprecost = uni(state[4]); st.bpx = as_i16(s5); st.bpy = s5 >> 16;
        nsad = (unsigned)uni(state[6]); nsub = (unsigned)uni(state[7]);
        star_rest(J.mnx, J.mny, J.mxx, J.mxy, J.merange, as_i16(s0), s0 >> 16, 3, (s3 >> 16) & 0xff, bmx, bmy, bcost, bPointNr, bDistance);
        st.bmx = bmx; st.bmy = bmy; st.bcost = bcost;
        int qx, qy, c;
        choose(st, qx, qy, c);
        if (lane == 0)
        {
            res->mv[0] = (int16_t)qx; res->mv[1] = (int16_t)qy;
            res->cost = c;
            res->sad_calls = nsad; res->subpel_calls = nsub;
        }
    }

    /* COST_MV_X4 of the UMH case: SAD measured relative to pix_base (pbx, pby), MV
     * bookkeeping relative to omv (motion.cpp:242-258, :732) */
    __device__ __forceinline__ void umh_x4(int &K, int pbx, int pby, int omx, int omy,
                                           int d0x, int d0y, int d1x, int d1y, int d2x, int d2y, int d3x, int d3y,
                                           int &bmx, int &bmy, int &bcost)
    {
        q_add(K, pbx + d0x, pby + d0y, as_i16(omx + d0x), as_i16(omy + d0y), 2, bmx, bmy, bcost);
        q_add(K, pbx + d1x, pby + d1y, as_i16(omx + d1x), as_i16(omy + d1y), 2, bmx, bmy, bcost);
        q_add(K, pbx + d2x, pby + d2y, as_i16(omx + d2x), as_i16(omy + d2y), 2, bmx, bmy, bcost);
        q_add(K, pbx + d3x, pby + d3y, as_i16(omx + d3x), as_i16(omy + d3y), 2, bmx, bmy, bcost);
    }

    /* CROSS, motion.cpp:281-305. The vertical COST_MV calls expand to
     * fref + omv.x + omv.y + i * stride, i.e. the SAD is taken at (omv.x + omv.y, +-i)
     * while the MV charged is (omv.x, omv.y +- i); reproduced literally. */
    __device__ __forceinline__ void umh_cross(int &K, int start, int x_max, int y_max, int pbx, int pby, int omx, int omy,
                              int mnx, int mny, int mxx, int mxy, int &bmx, int &bmy, int &bcost)
    {
        int i = start;
        if (x_max <= (mxx - omx < omx - mnx ? mxx - omx : omx - mnx))
            for (; i < x_max - 2; i = as_i16(i + 4))
                umh_x4(K, pbx, pby, omx, omy, i, 0, -i, 0, i + 2, 0, -i - 2, 0, bmx, bmy, bcost);
        for (; i < x_max; i = as_i16(i + 2))
        {
            if (omx + i <= mxx) q_add(K, omx + i, omy, as_i16(omx + i), omy, 2, bmx, bmy, bcost);
            if (omx - i >= mnx) q_add(K, omx - i, omy, as_i16(omx - i), omy, 2, bmx, bmy, bcost);
        }
        i = start;
        if (y_max <= (mxy - omy < omy - mny ? mxy - omy : omy - mny))
            for (; i < y_max - 2; i = as_i16(i + 4))
                umh_x4(K, pbx, pby, omx, omy, 0, i, 0, -i, 0, i + 2, 0, -i - 2, bmx, bmy, bcost);
        for (; i < y_max; i = as_i16(i + 2))
        {
            if (omy + i <= mxy) q_add(K, omx + omy, i, omx, as_i16(omy + i), 2, bmx, bmy, bcost);
            if (omy - i >= mny) q_add(K, omx + omy, -i, omx, as_i16(omy - i), 2, bmx, bmy, bcost);
        }
    }

    /* X265_UMH_SEARCH, motion.cpp:725-909. Returns true when the search continues into
     * the hexagon refinement (goto me_hex2) with the possibly rescaled merange. */
    __device__ __forceinline__ bool umh(const JobRegs &J, int mnx, int mny, int mxx, int mxy, int pmx, int pmy,
                        int &merange, int &bmx, int &bmy, int &bcost)
    {
        const int size_scale = (h * h) >> 4; /* motion.cpp:71-72 */
#define SAD_THRESH(v) (bcost < (((v) >> 4) * size_scale))
        const int numc = J.numc;
        const int pbx = bmx, pby = bmy;
        int omx = bmx, omy = bmy;
        int K = 0;
        const int ucost1 = bcost;
        int cross_start = 1;
        omx = pmx; omy = pmy;
        umh_x4(K, pbx, pby, omx, omy, 0, -1, 0, 1, -1, 0, 1, 0, bmx, bmy, bcost);
        q_flush(K, bmx, bmy, bcost);
        if (pmx | pmy)
        {
            omx = 0; omy = 0;
            umh_x4(K, pbx, pby, omx, omy, 0, -1, 0, 1, -1, 0, 1, 0, bmx, bmy, bcost);
            q_flush(K, bmx, bmy, bcost);
        }
        const int ucost2 = bcost;
        if ((bmx | bmy) && !(bmx == pmx && bmy == pmy))
        {
            omx = bmx; omy = bmy;
            umh_x4(K, pbx, pby, omx, omy, 0, -1, 0, 1, -1, 0, 1, 0, bmx, bmy, bcost);
            q_flush(K, bmx, bmy, bcost);
        }
        if (bcost == ucost2) cross_start = 3;

        omx = bmx; omy = bmy;
        if (bcost == ucost2 && SAD_THRESH(2000))
        {
            umh_x4(K, pbx, pby, omx, omy, 0, -2, -1, -1, 1, -1, -2, 0, bmx, bmy, bcost);
            umh_x4(K, pbx, pby, omx, omy, 2, 0, -1, 1, 1, 1, 0, 2, bmx, bmy, bcost);
            q_flush(K, bmx, bmy, bcost);
            if (bcost == ucost1 && SAD_THRESH(500)) return false;
            if (bcost == ucost2)
            {
                const int range = as_i16((merange >> 1) | 1);
                umh_cross(K, 3, range, range, pbx, pby, omx, omy, mnx, mny, mxx, mxy, bmx, bmy, bcost);
                umh_x4(K, pbx, pby, omx, omy, -1, -2, 1, -2, -2, -1, 2, -1, bmx, bmy, bcost);
                umh_x4(K, pbx, pby, omx, omy, -2, 1, 2, 1, -1, 2, 1, 2, bmx, bmy, bcost);
                q_flush(K, bmx, bmy, bcost);
                if (bcost == ucost2) return false;
                cross_start = as_i16(range + 2);
            }
        }

        /* adaptive search range from predictor agreement, motion.cpp:767-819 */
        if (numc)
        {
            /* range multipliers [mvd_ctx][sad_ctx], motion.cpp:769-772 */
            constexpr unsigned long long RANGE_MUL = nibs({ 3, 3, 4, 4, 3, 4, 4, 4, 4, 4, 4, 5, 4, 4, 5, 6 });
            const int qmx = J.qmx, qmy = J.qmy;
            int mvd, denom = 1;
            const bool is64 = (w == 64 && h == 64);
            if (numc == 1)
                mvd = is64 ? 25 : abs(qmx - J.cx[0]) + abs(qmy - J.cy[0]);
            else
            {
                denom = numc - 1;
                mvd = 0;
                if (!is64)
                {
                    mvd = abs(qmx - J.cx[0]) + abs(qmy - J.cy[0]);
                    denom++;
                }
                for (int i = 0; i < numc - 1; i++)
                    mvd += abs(J.mvcx(i) - J.mvcx(i + 1)) + abs(J.mvcy(i) - J.mvcy(i + 1));
            }
            const int sad_ctx = SAD_THRESH(1000) ? 0 : SAD_THRESH(2000) ? 1 : SAD_THRESH(4000) ? 2 : 3;
            const int mvd_ctx = mvd < 10 * denom ? 0 : mvd < 20 * denom ? 1 : mvd < 40 * denom ? 2 : 3;
            merange = (merange * nib(RANGE_MUL, mvd_ctx * 4 + sad_ctx)) >> 2;
        }

        umh_cross(K, cross_start, merange, merange >> 1, pbx, pby, omx, omy, mnx, mny, mxx, mxy, bmx, bmy, bcost);
        umh_x4(K, pbx, pby, omx, omy, -2, -2, -2, 2, 2, -2, 2, 2, bmx, bmy, bcost);
        q_flush(K, bmx, bmy, bcost);

        /* hexagon grid, motion.cpp:826-905 */
        omx = bmx; omy = bmy;
        int i = 1;
        do
        {
            int lim = mxx - omx;
            if (omx - mnx < lim) lim = omx - mnx;
            if (mxy - omy < lim) lim = mxy - omy;
            if (omy - mny < lim) lim = omy - mny;
            if (4 * i > lim)
            {
                for (int j = 0; j < 16; j++)
                {
                    const int mx = as_i16(omx + as_i16(kHex4[j][0] * as_i16(i))), my = as_i16(omy + as_i16(kHex4[j][1] * as_i16(i)));
                    if (in_range(mx, my, mnx, mny, mxx, mxy)) q_add(K, mx, my, mx, my, 2, bmx, bmy, bcost);
                }
            }
            else
            {
                for (int j = 0; j < 16; j++)
                {
                    const int px = omx + kHex4[j][0] * i, py = omy + kHex4[j][1] * i;
                    q_add(K, px, py, px, py, 8 | 2, bmx, bmy, bcost);
                }
            }
            q_flush(K, bmx, bmy, bcost);
            i = (i + 1) & 0xffff;
        }
        while (i <= merange >> 2);
#undef SAD_THRESH
        return in_range(bmx, bmy, mnx, mny, mxx, mxy);
    }

    /* the whole of motionEstimate for one PU */
    __device__ __forceinline__ void search(const x265amd_job *job, int method, int subme, x265amd_result *res)
    {
        JobRegs J;
        J.load(job);
        SearchState st;
#ifdef X265AMD_STAGE_STOP
        /* timing builds only (tools/search_bench.py --lib): leave after the set-up (1), the start-up
         * batch (2) or the integer stage (3); results are then meaningless */
        if (X265AMD_STAGE_STOP == 1) { if (lane == 0) res->cost = J.pux; return; }
        startup(J, st);
        if (X265AMD_STAGE_STOP == 2) { if (lane == 0) res->cost = st.bcost; return; }
        integer(J, method, st);
        if (X265AMD_STAGE_STOP == 3) { if (lane == 0) res->cost = st.bcost; return; }
        finish(subme, st, res);
#else
        startup(J, st);
        integer(J, method, st);
        finish(subme, st, res);
#endif
    }

    /* The stages before the sub-pel refinement, for launches that refine in a kernel of their own (me_small.h):
     * the result record receives the choice between the full-pel winner and the best quarter-pel predictor
     * (motion.cpp:1063-1069) and the counters so far. */
    __device__ __forceinline__ void search_staged(const x265amd_job *job, int method, x265amd_result *res)
    {
        JobRegs J;
        J.load(job);
        SearchState st;
        startup(J, st);
        integer(J, method, st);
        int qx, qy, bcost;
        choose(st, qx, qy, bcost);
        if (lane == 0)
        {
            res->mv[0] = (int16_t)qx; res->mv[1] = (int16_t)qy;
            res->cost = bcost;
            res->sad_calls = nsad; res->subpel_calls = nsub;
        }
    }

    /* motion.cpp:1063-1069 */
    __device__ __forceinline__ void choose(const SearchState &st, int &qx, int &qy, int &bcost) const
    {
        bcost = st.bcost;
        if (st.bprecost < bcost) { qx = st.bpx; qy = st.bpy; bcost = st.bprecost; }
        else { qx = as_i16(st.bmx << 2); qy = as_i16(st.bmy << 2); }
    }

    /* motion.cpp:562-621: predictor, MV(0,0) and the qpel candidates */
    __device__ __forceinline__ void startup(const JobRegs &J, SearchState &st)
    {
        const int mnx = J.mnx, mny = J.mny, mxx = J.mxx, mxy = J.mxy;
        const int qmnx = as_i16(mnx << 2), qmny = as_i16(mny << 2), qmxx = as_i16(mxx << 2), qmxy = as_i16(mxy << 2);
        const int numc = J.numc;
        set_mvp(J.qmx, J.qmy);
        nsad = 0; nsub = 0;

#define CLIPX(v) ((v) > qmxx ? qmxx : ((v) < qmnx ? qmnx : (v)))
#define CLIPY(v) ((v) > qmxy ? qmxy : ((v) < qmny ? qmny : (v)))
        /* Start-up candidates (motion.cpp:575-621): the clipped qpel predictor (no mvcost of its
         * own), its full-pel rounding, MV(0,0) and the qpel candidates are all independent
         * measurements, so they form ONE mixed batch (per-candidate mode in `meta`); the
         * reference's sequential tests -- including "skip a candidate equal to the current
         * bestpre" -- are then replayed on the costs, which also keeps the counters exact. */
        int pmx = CLIPX(mvpx), pmy = CLIPY(mvpy);
        int bpx = pmx, bpy = pmy;
        int bmx = as_i16((pmx + 2) >> 2), bmy = as_i16((pmy + 2) >> 2);
        int K = 0;
        put(K++, pmx, pmy, pmx, pmy, 0, 1);                          /* SAD at qpel, no mvcost */
        const int kFp = ((pmx | pmy) & 3) ? K : -1;
        if (kFp >= 0) put(K++, bmx, bmy, bmx, bmy, 2, 0 | 16);       /* full-pel SAD + mvcost */
        const int kZero = (pmx | pmy) ? K : -1;
        if (kZero >= 0) put(K++, 0, 0, 0, 0, 2, 0 | 16);
        int kc[4], ccx[4], ccy[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            kc[i] = -1; ccx[i] = 0; ccy[i] = 0;
            if (i < numc)
            {
                ccx[i] = CLIPX(J.cx[i]); ccy[i] = CLIPY(J.cy[i]);
                if ((ccx[i] | ccy[i]) && !(ccx[i] == pmx && ccy[i] == pmy))
                {
                    kc[i] = K;
                    put(K++, ccx[i], ccy[i], ccx[i], ccy[i], 0, 1 | 16);
                }
            }
        }
        eval(K, -1, false);
        int bprecost = cost_of(0);
        nsub++;
        int bcost = bprecost;
        if (kFp >= 0) { bcost = cost_of(kFp); nsad++; }
        if (kZero >= 0)
        {
            const int c = cost_of(kZero);
            nsad++;
            if (c < bcost) { bcost = c; bmx = 0; bmy = 0; }
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (kc[i] >= 0 && !(ccx[i] == bpx && ccy[i] == bpy))
            {
                const int c = cost_of(kc[i]);
                nsub++;
                if (c < bprecost) { bprecost = c; bpx = ccx[i]; bpy = ccy[i]; }
            }
#undef CLIPX
#undef CLIPY
        st.pmx = as_i16((pmx + 2) >> 2); st.pmy = as_i16((pmy + 2) >> 2);
        st.bmx = bmx; st.bmy = bmy; st.bcost = bcost; st.bprecost = bprecost; st.bpx = bpx; st.bpy = bpy;
    }

    /* the integer search proper, motion.cpp:626-1061 */
    __device__ __forceinline__ void integer(const JobRegs &J, int method, SearchState &st)
    {
        const int mnx = J.mnx, mny = J.mny, mxx = J.mxx, mxy = J.mxy;
        const int pmx = st.pmx, pmy = st.pmy;
        int merange = J.merange;
        int bmx = st.bmx, bmy = st.bmy, bcost = st.bcost;
        /* the pattern tables of motion.cpp:145-147, :673-681 as nibble-packed immediates */
        constexpr unsigned long long SQX = nibs({ 0, 0, 0, -1, 1, -1, -1, 1, 1 }), SQY = nibs({ 0, -1, 1, 0, 0, -1, 1, -1, 1 });
        constexpr unsigned long long HRX = nibs({ -2, -1, 1, 2, 1, -1 }), HRY = nibs({ 0, 2, 2, 0, -2, -2 });
        constexpr unsigned long long H2X = nibs({ -1, -2, -1, 1, 2, 1, -1, -2 }), H2Y = nibs({ -2, 0, 2, 2, 0, -2, -2, 0 });

        bool run_hex = false;
        switch (method)
        {
        case X265AMD_DIA:
        {
            /* motion.cpp:628-649; the packed (cost << 4) + dir compare is a strict
             * '<' scan in the order N, S, W, E */
            dia_rest(mnx, mny, mxx, mxy, merange, true, bmx, bmy, bcost);
            break;
        }
        case X265AMD_HEX:
            run_hex = true;
            break;
        case X265AMD_UMH:
            run_hex = umh(J, mnx, mny, mxx, mxy, pmx, pmy, merange, bmx, bmy, bcost);
            break;
        case X265AMD_STAR:
        {
            int bPointNr = 0, bDistance = 0;
            star_rest(mnx, mny, mxx, mxy, merange, bmx, bmy, 0, 0, bmx, bmy, bcost, bPointNr, bDistance);
            break;
        }
        case X265AMD_FULL:
        {
            /* motion.cpp:1021-1056: raster scan, strict '<' keeps the first minimum; one lane per
             * position, several short rows per batch */
            const int nx = mxx - mnx + 1;
            int pn, pd;
            if (nx <= MAXC)
            {
                const int rpb = MAXC / nx, inv = (65536 + nx - 1) / nx;
                const int rs = (lane * inv) >> 16, j = lane - rs * nx;
                for (int ty = mny; ty <= mxy; ty += rpb)
                {
                    const int K = put_lanes(rs < rpb && ty + rs <= mxy, mnx + j, ty + rs, mnx + j, ty + rs, 2, 0);
                    fold_pt(K, bmx, bmy, bcost, pn, pd);
                }
            }
            else
                for (int ty = mny; ty <= mxy; ty++)
                    for (int j0 = 0; j0 < nx; j0 += MAXC)
                    {
                        const int K = put_lanes(j0 + lane < nx, mnx + j0 + lane, ty, mnx + j0 + lane, ty, 2, 0);
                        fold_pt(K, bmx, bmy, bcost, pn, pd);
                    }
            break;
        }
        default:
            break;
        }

        if (run_hex)
        {
            /* motion.cpp:651-723. Hexagon points and the closing square are evaluated
             * without a range check, as in the reference. */
            /* first ring: direction codes 2..7 in measuring order */
            put_lanes(lane < 6, bmx + nib(HRX, lane), bmy + nib(HRY, lane), as_i16(bmx + nib(HRX, lane)), as_i16(bmy + nib(HRY, lane)), 2, 0);
            eval(6, 0, true);
            const int code = fold_min(6, bcost) + 2;
            if (code > 1)
            {
                const int dir = code - 2;
                bmx = as_i16(bmx + kHex2[dir + 1][0]); bmy = as_i16(bmy + kHex2[dir + 1][1]);
                hex_walk(mnx, mny, mxx, mxy, merange, dir, bmx, bmy, bcost);
            }
            hex_square(bmx, bmy, bcost);
        }

        st.bmx = bmx; st.bmy = bmy; st.bcost = bcost;
    }

    /* choice between the full-pel winner and the best qpel predictor, then the sub-pel
     * refinement; motion.cpp:1063-1145 */
    __device__ __forceinline__ void finish(int subme, const SearchState &st, x265amd_result *res)
    {
        int bcost, qx, qy;
        choose(st, qx, qy, bcost);

        const int hpel_iters = kWorkload[subme][0], hpel_dirs = kWorkload[subme][1];
        const int qpel_iters = kWorkload[subme][2], qpel_dirs = kWorkload[subme][3], hpel_satd = kWorkload[subme][4];

        if (bcost)
        {
            /* all directions of one refinement round share the same centre, so a
             * round is one batched evaluation (motion.cpp:1077-1141) */
            if (LOWRES)
            {
                refine_round(qx, qy, bcost, hpel_dirs, 2, 1, false);
                refine_round(qx, qy, bcost, qpel_dirs, 1, 2, true);
            }
            else
            {
                /* the SATD re-measure of the centre rides along with the first round that uses SATD */
                bool centre = hpel_satd != 0;
                for (int it = 0; it < hpel_iters; it++)
                {
                    refine_round(qx, qy, bcost, hpel_dirs, 2, hpel_satd ? 2 : 1, centre);
                    centre = false;
                }
                centre = !hpel_satd;
                if (centre && !qpel_iters)
                {
                    put(0, qx, qy, qx, qy, 0, 0);
                    eval(1, 2, true);
                    bcost = cost_of(0);
                }
                for (int it = 0; it < qpel_iters; it++)
                {
                    refine_round(qx, qy, bcost, qpel_dirs, 1, 2, centre);
                    centre = false;
                }
            }
        }

        if (lane == 0)
        {
            res->mv[0] = (int16_t)qx; res->mv[1] = (int16_t)qy;
            res->cost = bcost;
            res->sad_calls = nsad; res->subpel_calls = nsub;
        }
    }

    /* One refinement round: the `dirs` neighbours of (qx, qy) at distance `step` in the order
     * of the square table, one lane each. With `centre`, (qx, qy) itself is measured in the same
     * batch and its cost replaces bcost first (the SATD re-measure of motion.cpp:1086, :1117). */
    __device__ __forceinline__ void refine_round(int &qx, int &qy, int &bcost, int dirs, int step, int mode, bool centre)
    {
        constexpr unsigned long long SQX = nibs({ 0, 0, 0, -1, 1, -1, -1, 1, 1 }), SQY = nibs({ 0, -1, 1, 0, 0, -1, 1, -1, 1 });
        const int k0 = centre ? 1 : 0, i = lane + 1 - k0;
        const int tx = as_i16(qx + nib(SQX, i) * step), ty = as_i16(qy + nib(SQY, i) * step);
        put_lanes(lane < k0 + dirs, tx, ty, tx, ty, 0, 0);
        eval(k0 + dirs, mode, true);
        if (centre) bcost = cost_of(0);
        const int k = fold_range(k0, k0 + dirs, bcost);
        const int bdir = k < 0 ? 0 : k - k0 + 1;
        qx = as_i16(qx + kSquare1[bdir][0] * step); qy = as_i16(qy + kSquare1[bdir][1] * step);
    }

    /* bind to a PU: geometry, tile tables, the source block copy (setSourcePU,
     * motion.cpp:116-142) and the packed filter taps. `fenc_rows` is the height the launch
     * reserved for source blocks; the interpolation planes follow them. */
    __device__ __forceinline__ void bind(int lds_base, const PlaneDev &fencPl, const PlaneDev *refPl, int nref,
                                         const uint16_t *costTab, int depth_, int pu_x, int pu_y, int w_, int h_,
                                         int fenc_rows = ROWS, int budget = 0, bool copy_fenc = true)
    {
        base = lds_base;
        lane = threadIdx.x & 63;
        w = w_; h = h_;
        mvpx = mvpy = 0;
        const int wt = w >> 2, T = wt * (h >> 2);
        int lTp = 0;
        while ((1 << lTp) < T) lTp++;
        const int inv_wt = inv_tiles(wt);
        const int rstride = refPl[0].stride;
        const long roff = (long)pu_y * rstride + pu_x;
        wave_sync();
        if (lane == 0)
        {
            EvalCtx *c = lds_ctx(base);
            for (int i = 0; i < 4; i++) c->ref[i] = (unsigned long long)refPl[i < nref ? i : 0].origin;
            c->ctab = (unsigned long long)costTab;
            c->roff = roff;
            c->rstride = rstride; c->w = w; c->h = h; c->T = T; c->Tp = 1 << lTp; c->lTp = lTp; c->wt = wt; c->inv_wt = inv_wt;
            c->depth = depth_; c->mvpx = 0; c->mvpy = 0; c->rk_have_mvd = 0;
            c->planes_off = L_FENC + fenc_rows * 64 * (int)sizeof(P); c->plane_budget = budget;
        }
        if (!LOWRES && !RK && lane < 48) lds_i(base, L_TAPS)[lane] = kTapWords.w[lane];
        const P AS1 *src = (const P AS1*)fencPl.origin + (long)pu_y * fencPl.stride + pu_x;
        /* a caller that reads the source block from global memory itself reserves no LDS for it */
        for (int t = lane; copy_fenc && t < T; t += 64)
        {
            const int ty = (t * inv_wt) >> 16, tx = t - ty * wt;
#pragma unroll
            for (int y = 0; y < 4; y++)
                Px<P>::store_lds_aligned(base + L_FENC + ((ty * 4 + y) * 64 + tx * 4) * (int)sizeof(P),
                                         Px<P>::load_g(src + (long)(ty * 4 + y) * fencPl.stride + tx * 4));
        }
        wave_sync();
    }

    /* Rockchip stages: the three cost predictors (motion.cpp:1284-1292) */
    __device__ __forceinline__ void set_rk_predictors(const int16_t mvp[3][2], int have_mvd)
    {
        wave_sync();
        if (lane == 0)
        {
            EvalCtx *c = lds_ctx(base);
            c->rk_have_mvd = have_mvd;
            for (int k = 0; k < 3; k++) { c->rk_mvp[k][0] = mvp[k][0]; c->rk_mvp[k][1] = mvp[k][1]; }
        }
        wave_sync();
    }
};

} // namespace x265amd
#endif
