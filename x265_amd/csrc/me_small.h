/* Square PUs of 8-bit samples in fixed-shape kernels (full-resolution planes): the sub-pel refinement
 * (refine_small_kernel for 8x8 and 16x16 PUs, refine_big_kernel for 32x32 and 64x64) and the common case of the STAR
 * integer stage of 8x8 and 16x16 PUs (star_small_kernel).
 *
 * The refinement rounds of MotionEstimate::motionEstimate (motion.cpp:1071-1141) have a fixed shape: the
 * neighbours of one centre at one distance, in the order of the square table, each an 8-tap interpolation
 * plus SAD or SATD (subpelCompare, motion.cpp:1651-1689). The one-wave search (me_search.h) runs them through its
 * general candidate machinery, and an 8x8 PU, four 4x4 tiles, fills a quarter of its passes. Here what is uniform
 * per PU (centre, best cost, predictor) is a value every lane of the PU's group holds, decisions are group-wide
 * minima (DPP), the passes are straight-line code and, for the small sizes, filled by four PUs.
 *
 * Per round and PU the three columns of candidates (x - step, x, x + step) share their horizontal pass: one
 * plane of luma_hps intermediates (ipfilter.cpp:235-278) per column, S + 8 rows x S samples, int16, column-major
 * in LDS. With 8-bit samples that intermediate keeps every bit, luma_vsp (:359-400) with the phase-0 taps on it
 * gives exactly luma_hpp's (sum + 32) >> 6, and the phase-0 intermediate of a sample is 64 * sample - 8192, on
 * which luma_vsp equals luma_vpp: one plane kind serves every fraction, as in eval_planes. The vertical pass
 * and the cost run one lane per (tile, column) in small_column_cost, one candidate of every PU per pass; a lane of a
 * PU above 8x8 looks after four such pairs.
 *
 * The stage before leaves (mv, cost, counters) of the full-pel / predictor choice (motion.cpp:1063-1069) in
 * the PU's result record; the refinement kernels refine that record in place. PUs of other sizes were finished by
 * the one-wave search and are skipped.
 */
#ifndef X265AMD_ME_SMALL_H
#define X265AMD_ME_SMALL_H

/* Waves per SIMD the register allocator is asked to leave room for (512 registers per lane and SIMD; overrides for
 * measurements: -DOCC_...). Measured on the 1080p step (32 frame pairs, all launch classes side by side): the 16x16
 * refinement at 169 registers (two waves) 3.55 ms, at 164 (three) 3.40, at 128 with 16 spilled dwords (four) 3.29, at
 * 96 with 63 spilled (five) 3.53; the other kernels' settings are within +-0.5 % of their neighbours'. */
#ifndef OCC_R16
#define OCC_R16 4
#endif
#ifndef OCC_R8
#define OCC_R8 5
#endif
#ifndef OCC_RB64
#define OCC_RB64 5
#endif
#ifndef OCC_RB32
#define OCC_RB32 6
#endif
#ifndef OCC_S16
#define OCC_S16 6
#endif
#ifndef OCC_S8
#define OCC_S8 6
#endif
#ifndef OCC_SB64
#define OCC_SB64 5
#endif
#ifndef OCC_SB32
#define OCC_SB32 8
#endif
namespace x265amd {

/* whether the launch's star_rect_kernel (or, for shapes up to 16 samples, star_tiny_kernel's pass for them) takes a w x h
 * PU: a shape of rect_class that fits the launch and is not a square of the launch's own fixed-shape sizes */
__device__ __forceinline__ bool rect_takes(const SearchArgs &a, int w, int h)
{
    return a.rect && rect_class(w, h) && (w > h ? w : h) <= a.max_dim && !(w == h && (w == a.staged || w == a.staged2));
}

/* likewise for star_tiny_kernel's own shapes (a PU larger than the launch's size is refused by the one-wave search) */
__device__ __forceinline__ bool tiny_takes(const SearchArgs &a, int w, int h)
{
    return a.tiny && tiny_class(w, h) && (w > h ? w : h) <= a.max_dim && !(w == h && (w == a.staged || w == a.staged2));
}

template<int S> struct SmallGeo
{
    static_assert(S == 8 || S == 16 || S == 32 || S == 64, "square PUs of 8 to 64 samples");
    static constexpr int LTW = S == 8 ? 1 : S == 16 ? 2 : S == 32 ? 3 : 4;
    static constexpr int TW = S / 4;                /* tiles in a PU row */
    static constexpr int LT = 2 * LTW;              /* log2 of the tiles of a PU */
    static constexpr int LG = LT + 2;               /* log2 of the lanes of a PU's group: a lane per (tile, column) */
    static constexpr int LPP = 1 << LG;
    static constexpr int NPU = LPP < 64 ? 64 / LPP : 1;     /* PUs per wave */
    static constexpr int ROWS = S + 8;              /* rows of a plane: the PU, the 7 rows the taps reach and one for a second first row */
    static constexpr int PROW = ROWS + 2;           /* int16 rows of a plane column, with slack for an odd first row */
    static constexpr int PLANE = S * PROW * 2;      /* bytes of one plane */
    static constexpr int IPP = ROWS * TW;           /* horizontal-pass items (4 samples of a row) per plane */
    static constexpr int SIZE_CODE = S | (S << 8);  /* the (w, h) bytes of a job record */
};
#define SM_TAPS 0                                   /* 48 dwords of packed filter taps (TapWords): the horizontal pass reads words 9, 10 of a phase */
#define SM_VTAPS 192                                /* 80 dwords: the vertical pass's taps (VTapWords) */
#define SM_FIXED 512                                /* first byte behind the tap tables */

/* The taps of the vertical pass as the lanes of small_column_cost read them: per (phase, parity of the first row in
 * the column's dword pairs) two sets of five int16 pairs, P for output rows 0 and 2 and Q for rows 1 and 3, each
 * applied to five consecutive dwords (ten plane rows). Even first row: P = the taps on rows 0..7 of the ten, Q = on
 * rows 1..8; odd: P = on rows 1..8, Q = on rows 2..9 -- so a lane needs no realignment of what it loads. The taps are
 * scaled by 16: the sum then carries luma_vsp's (sum + offset) >> 12 (ipfilter.cpp:372-389 with 8-bit samples) in its
 * high half, where the int16 the reference casts to (:389) is taken with one byte permute. */
struct VTapWords
{
    int w[80];
    constexpr VTapWords() : w()
    {
        const int taps[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
                                 { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
        for (int ph = 0; ph < 4; ph++)
            for (int par = 0; par < 2; par++)
                for (int set = 0; set < 2; set++)
                {
                    const int first = par + set;        /* row of the ten the first tap falls on */
                    for (int j = 0; j < 5; j++)
                    {
                        const int a = 2 * j - first, b = 2 * j + 1 - first;
                        const int lo = a >= 0 && a < 8 ? taps[ph][a] * 16 : 0, hi = b >= 0 && b < 8 ? taps[ph][b] * 16 : 0;
                        w[(ph * 2 + par) * 10 + set * 5 + j] = (int)((unsigned)(lo & 0xffff) | ((unsigned)(hi & 0xffff) << 16));
                    }
                }
    }
};
__constant__ static const VTapWords kVTapWords = VTapWords();

/* both tap tables into LDS, by the first 64 threads of the workgroup. 8-bit samples: the table above. Deeper samples:
 * the same layout with the taps scaled by 2^(depth - 4), so that luma_vsp's shift of 20 - depth (ipfilter.cpp:372-375)
 * again ends at bit 16; the scaled taps (at most 58 << 6 with 10 bits) still fit int16. */
template<typename P>
__device__ __forceinline__ void small_load_taps(int tid, int depth)
{
    if (tid < 48) ((int*)(x265amd_smem + SM_TAPS))[tid] = kTapWords.w[tid];
    if constexpr (sizeof(P) == 1)
    {
        if (tid < 64) ((int*)(x265amd_smem + SM_VTAPS))[tid] = kVTapWords.w[tid];
        if (tid < 16) ((int*)(x265amd_smem + SM_VTAPS))[64 + tid] = kVTapWords.w[64 + tid];
    }
    else
        for (int i = tid; i < 80; i += 64)
        {
            const int ph = i / 20, rem = i - ph * 20, par = rem / 10, k = rem - par * 10, set = k / 5, j = k - set * 5;
            const int first = par + set, a = 2 * j - first, b = 2 * j + 1 - first;
            const int lo = a >= 0 && a < 8 ? (int)kLumaTaps[ph][a] << (depth - 4) : 0, hi = b >= 0 && b < 8 ? (int)kLumaTaps[ph][b] << (depth - 4) : 0;
            ((int*)(x265amd_smem + SM_VTAPS))[i] = (int)((unsigned)(lo & 0xffff) | ((unsigned)(hi & 0xffff) << 16));
        }
}

/* the lane's column of a 4x4 source tile, rows (0, 1) and (2, 3) as int16 pairs */
struct ColSrc
{
    int ty, col;
    uint32_t f01, f23;
};

template<int LG> __device__ __forceinline__ int grp_lane(int v, int k) { return __shfl(v, (int)(threadIdx.x & 63 & ~((1 << LG) - 1)) + k); }
template<int LG> __device__ __forceinline__ int grp_min(int v)
{
    if (LG == 3)
    {
        int o;
        o = dpp_all<0xB1>(v); v = o < v ? o : v;
        o = dpp_all<0x4E>(v); v = o < v ? o : v;
        o = dpp_all<0x141>(v); v = o < v ? o : v;
        return v;
    }
    if (LG == 4) return row_min16(v);
    if (LG == 5)
    {
        const int m = row_min16(v), o = __shfl_xor(m, 16);
        return o < m ? o : m;
    }
    return wave_min(v);
}
/* flags of the first 16 lanes of the group */
template<int LG> __device__ __forceinline__ unsigned grp_bits16(bool p)
{
    const unsigned long long b = __ballot(p);
    return (unsigned)((b >> (threadIdx.x & 63 & ~((1 << LG) - 1))) & 0xffff);
}
template<int LG> __device__ __forceinline__ int grp_count(bool p)
{
    const unsigned long long b = __ballot(p);
    return LG == 6 ? (int)__builtin_popcountll(b) : (int)__builtin_popcount((unsigned)((b >> (threadIdx.x & 48)) & 0xffff));
}
/* the same for groups of any size */
template<int LG> __device__ __forceinline__ int grp_count_any(bool p)
{
    const unsigned long long b = __ballot(p);
    return (int)__builtin_popcountll((b >> (threadIdx.x & 63 & ~((1 << LG) - 1))) & ((1ull << (1 << LG)) - 1));
}
template<int LG> __device__ __forceinline__ bool grp_any(bool p)
{
    const unsigned long long b = __ballot(p);
    return LG == 6 ? b != 0 : ((b >> (threadIdx.x & 48)) & 0xffff) != 0;
}

/* four samples of one plane row: luma_hps of 8-bit samples on v_dot4_i32_i8. The samples are biased to signed bytes;
 * the taps of every phase add up to 64, so the sum comes out 8192 low, which is exactly luma_hps's offset
 * (ipfilter.cpp:246-247 with 8-bit samples). `tp`: the packed taps of the phase (TapWords). */
__device__ __forceinline__ void hps_quad(const uint8_t AS1 *s, const int *tp, int16_t *dst, int pitch)
{
    const uint32_t d0 = Px<uint8_t>::load_g(s) ^ 0x80808080u, d1 = Px<uint8_t>::load_g(s + 4) ^ 0x80808080u,
                   d2 = Px<uint8_t>::load_g(s + 8) ^ 0x80808080u;
    const int tlo = tp[9], thi = tp[10];
    dst[0] = (int16_t)__builtin_amdgcn_sdot4(d0, tlo, __builtin_amdgcn_sdot4(d1, thi, 0, false), false);
#pragma unroll
    for (int x = 1; x < 4; x++)
        dst[x * pitch] = (int16_t)__builtin_amdgcn_sdot4(__builtin_amdgcn_alignbyte(d1, d0, x), tlo,
                                                         __builtin_amdgcn_sdot4(__builtin_amdgcn_alignbyte(d2, d1, x), thi, 0, false), false);
}

/* The same for 16-bit samples of `depth` bits: luma_hps's (sum - (8192 << shift)) >> shift with shift = depth - 8
 * (ipfilter.cpp:244-247). The intermediate drops `shift` bits of the sum, and still it serves every fraction:
 * luma_hpp's (sum + 32) >> 6 equals ((sum >> shift) + (32 >> shift)) >> (6 - shift), which is what luma_vsp's formula
 * gives for the phase-0 taps, and a phase-0 intermediate is the sample itself, scaled. */
__device__ __forceinline__ void hps_quad(const uint16_t AS1 *s, const int *tp, int16_t *dst, int pitch, int depth)
{
    int sum[4];
    HFilter<uint16_t>::run(s, tp, sum);
    const int shift = depth - 8, off = -(8192 << shift);
#pragma unroll
    for (int x = 0; x < 4; x++) dst[x * pitch] = (int16_t)((sum[x] + off) >> shift);
}
__device__ __forceinline__ void hps_quad(const uint8_t AS1 *s, const int *tp, int16_t *dst, int pitch, int) { hps_quad(s, tp, dst, pitch); }

/* Vertical pass + cost of the lane's (tile, column) against plane `poff` (samples of `depth` bits, at most 10): first tap row `r0` of tile
 * row 0, vertical phase yf. The four lanes of a quad hold the four columns of a tile. SAD: the sum over the lane's
 * column. SATD: the vertical Hadamard inside the lane, the horizontal one across the quad (DPP); the lane returns the
 * abs-sum of its four coefficients and the caller halves the total (every tile's abs-sum is even, pixel.cpp:226). */
/* N (4 or 5) consecutive output rows of the vertical pass for the lane's column: sums whose high halves are
 * (int16_t)(sum >> (20 - depth)), luma_vsp's result before the clip (ipfilter.cpp:372-389) */
template<int N>
__device__ __forceinline__ void column_rows(int poff, int prow, int r0, int yf, const ColSrc &ct, int depth, int o[N]);
template<int S, int N>
__device__ __forceinline__ void small_column_rows(int poff, int r0, int yf, const ColSrc &ct, int depth, int o[N])
{
    column_rows<N>(poff, SmallGeo<S>::PROW, r0, yf, ct, depth, o);
}
/* prow: int16 rows per plane column (even) */
template<int N>
__device__ __forceinline__ void column_rows(int poff, int prow, int r0, int yf, const ColSrc &ct, int depth, int o[N])
{
    const int R = r0 + 4 * ct.ty;
    const uint32_t *col = (const uint32_t*)(x265amd_smem + poff + (__mul24(ct.col, prow) + (R & ~1)) * 2);
    const uint32_t *tp = (const uint32_t*)(x265amd_smem + SM_VTAPS) + ((yf << 1) | (r0 & 1)) * 10;
    uint32_t E[N + 2], P[5], Q[5];
#pragma unroll
    for (int i = 0; i < N + 2; i++) E[i] = col[i];
#pragma unroll
    for (int i = 0; i < 5; i++) { P[i] = tp[i]; Q[i] = tp[5 + i]; }
    /* luma_vsp's rounding and the intermediates' bias (shift 20 - depth), scaled like the taps */
    const int OFF = ((1 << (19 - depth)) + (8192 << 6)) << (depth - 4);
#pragma unroll
    for (int k = 0; k < N; k++)
    {
        const uint32_t *T = (k & 1) ? Q : P;
        const int e = k >> 1;
        o[k] = dot2(E[e], T[0], dot2(E[e + 1], T[1], dot2(E[e + 2], T[2], dot2(E[e + 3], T[3], dot2(E[e + 4], T[4], OFF)))));
    }
}

/* cost of the lane's column given its four output rows: clipped to the sample range as pairs, then SAD or the lane's
 * share of the 4x4 Hadamard. Differences and every Hadamard stage fit int16 up to 10 bits (16 * 1023) */
__device__ __forceinline__ int small_rows_cost(int o0, int o1, int o2, int o3, bool satd, const ColSrc &ct, int x, int depth)
{
    const short maxv = (short)((1 << depth) - 1);
    const s16x2 lo = { 0, 0 }, hi = { maxv, maxv };
    s16x2 p01 = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm((uint32_t)o1, (uint32_t)o0, 0x07060302u));
    s16x2 p23 = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm((uint32_t)o3, (uint32_t)o2, 0x07060302u));
    p01 = __builtin_elementwise_min(__builtin_elementwise_max(p01, lo), hi);
    p23 = __builtin_elementwise_min(__builtin_elementwise_max(p23, lo), hi);
    /* SAD: source and prediction are unsigned 16-bit pairs, which is what v_sad_u16 takes */
    if (!satd)
        return (int)__builtin_amdgcn_sad_u16(ct.f01, __builtin_bit_cast(uint32_t, p01),
                                             __builtin_amdgcn_sad_u16(ct.f23, __builtin_bit_cast(uint32_t, p23), 0u));
    const s16x2 d01 = __builtin_bit_cast(s16x2, ct.f01) - p01, d23 = __builtin_bit_cast(s16x2, ct.f23) - p23;
    const s16x2 Sm = d01 + d23, Df = d01 - d23;                     /* (r0+r2, r1+r3), (r0-r2, r1-r3) */
    const s16x2 A = __builtin_shufflevector(Sm, Df, 0, 2), B = __builtin_shufflevector(Sm, Df, 1, 3);
    s16x2 U = A + B, W = A - B;                                     /* vertical coefficients (c0, c2), (c1, c3) */
    const short m1 = (short)(1 - 2 * (x & 1)), m2 = (short)(1 - (x & 2));
    const s16x2 s1 = { m1, m1 }, s2 = { m2, m2 };
    U = U * s1 + __builtin_bit_cast(s16x2, dpp_all<0xB1>(__builtin_bit_cast(int, U)));
    W = W * s1 + __builtin_bit_cast(s16x2, dpp_all<0xB1>(__builtin_bit_cast(int, W)));
    U = U * s2 + __builtin_bit_cast(s16x2, dpp_all<0x4E>(__builtin_bit_cast(int, U)));
    W = W * s2 + __builtin_bit_cast(s16x2, dpp_all<0x4E>(__builtin_bit_cast(int, W)));
    /* sum of the four magnitudes: a coefficient c as the unsigned c + 0x8000 is |c| away from 0x8000 */
    const uint32_t bias = 0x80008000u;
    return (int)__builtin_amdgcn_sad_u16(__builtin_bit_cast(uint32_t, U) ^ bias, bias,
                                         __builtin_amdgcn_sad_u16(__builtin_bit_cast(uint32_t, W) ^ bias, bias, 0u));
}

template<int S>
__device__ __forceinline__ int small_column_cost(int poff, int r0, int yf, bool satd, const ColSrc &ct, int x, int depth = 8)
{
    int o[4];
    small_column_rows<S, 4>(poff, r0, yf, ct, depth, o);
    return small_rows_cost(o[0], o[1], o[2], o[3], satd, ct, x, depth);
}

/* Two candidates of one plane and phase whose first rows are r0 and r0 + 1 -- the half-pel neighbours above and below
 * a centre: (y - 2) and (y + 2) quarter-pels are one row apart -- share three of their four output rows: five vertical
 * sums instead of eight. */
template<int S>
__device__ __forceinline__ void small_column_cost_pair(int poff, int r0, int yf, bool satd, const ColSrc &ct, int x, int depth, int &va, int &vb)
{
    int o[5];
    small_column_rows<S, 5>(poff, r0, yf, ct, depth, o);
    va += small_rows_cost(o[0], o[1], o[2], o[3], satd, ct, x, depth);
    vb += small_rows_cost(o[1], o[2], o[3], o[4], satd, ct, x, depth);
}

/* the lane's tile column of the source block */
template<int S, typename P>
__device__ __forceinline__ void small_source_column(const P AS1 *fenc0, int fstride, int pux, int puy, int l, ColSrc &ct)
{
    typedef SmallGeo<S> G;
    const int t = l >> 2, tx = t & (G::TW - 1), ty = t >> G::LTW;
    ct.ty = ty; ct.col = tx * 4 + (l & 3);
    const P AS1 *src = fenc0 + (long)(puy + ty * 4) * fstride + pux + ct.col;
    uint32_t f[4];
#pragma unroll
    for (int y = 0; y < 4; y++) f[y] = (uint32_t)src[(long)y * fstride];
    ct.f01 = f[0] | (f[1] << 16); ct.f23 = f[2] | (f[3] << 16);
}

/* the planes of a job: the launch's own, or its entry of the pair table (as job_planes, per lane) */
template<typename P>
__device__ __forceinline__ void small_planes(const SearchArgs &a, int d1, const P AS1 *&fenc0, int &fstride,
                                             const P AS1 *&ref0, int &rstride)
{
    fenc0 = (const P AS1*)a.fenc.origin; fstride = a.fenc.stride;
    ref0 = (const P AS1*)a.ref[0].origin; rstride = a.ref[0].stride;
    if (a.pairs)
    {
        int pi = (d1 >> 24) & 0xff;
        if (pi >= a.npairs) pi = 0;
        const PairDev *p = a.pairs + pi;
        fenc0 = (const P AS1*)p->fenc.origin; fstride = p->fenc.stride;
        ref0 = (const P AS1*)p->ref.origin; rstride = p->ref.stride;
    }
}

/* ---------------------------------------------------------------- sub-pel refinement */

/* Sixteen lanes per 8x8 PU, four PUs per wave, a (tile, column) pair per lane. Thirty-two lanes per 16x16 PU, two PUs
 * per wave, two pairs per lane, one tile row band apart: the cross-lane sums and the bookkeeping of a candidate are paid
 * once per two pairs. (Sixteen lanes with four pairs issue 9 % fewer instructions, but need 157 registers and 10 KB of
 * LDS per wave -- three waves per SIMD -- and were 1.4 % slower on the headline step; a pair per lane, 6 % slower.) */
template<int S> struct RefineGeo
{
    static_assert(S == 8 || S == 16, "8x8 or 16x16 PUs");
    static constexpr int LG = S == 16 ? 5 : 4, LPP = 1 << LG, NPU = 64 / LPP;
    static constexpr int ITEMS = SmallGeo<S>::LPP / LPP;
};

template<int S> struct RefineLds
{
    typedef SmallGeo<S> G;
    static constexpr int PLANES = SM_FIXED;
    static constexpr int PU_BYTES = 3 * G::PLANE + 4;               /* the odd dword staggers the PUs over the LDS banks */
    /* a 16x16 PU that takes its four 8x8 children along: per candidate of a round the sums of the four quadrants
     * (9 x 4 ints), then the children's predictors (4 x, 4 y) */
    static constexpr int FAM = PLANES + RefineGeo<S>::NPU * PU_BYTES + 32;
    static constexpr int FAM_BYTES = 192;
    static constexpr int TOTAL = FAM + (S == 16 ? RefineGeo<S>::NPU * FAM_BYTES : 0);
};

template<int S, typename P> struct RefineSmall
{
    typedef SmallGeo<S> G;
    typedef RefineGeo<S> RG;
    typedef RefineLds<S> L;
    int depth;                  /* bits per sample: 8 for uint8_t planes, at most 10 otherwise */
    /* per PU, the same in every lane of its group */
    int qx, qy, bcost, mvpx, mvpy, roff;
    unsigned nsub;
    bool active;
    const P AS1 *ref0;          /* origin of the PU's reference plane (PUs of one wave may search different pictures) */
    int rstride;
    /* per lane */
    int lane, grp, l;
    ColSrc ct[RG::ITEMS];
    const uint16_t AS1 *ctab;
    /* a 16x16 PU launched together with the list of the 8x8 PUs (its four children at 4 k .. 4 k + 3 of that list), all
     * five starting from one MV: SAD and SATD are
     * sums over 4x4 tiles, so a candidate's distortion for a child is the sum over the child's quadrant. The lanes of
     * the PU keep the sums of its upper and lower tiles apart and the quadrants' sums go to LDS; lanes 0 .. K - 1 then
     * take every child's decisions beside the parent's with the child's own lambda * bits. The children stay while every
     * decision moves all five to one MV; if one differs they are left to refine_small_kernel<8>. */
    bool quad;                  /* per PU */
    int cb[4];                  /* the children's best costs */
    unsigned cadd;              /* candidates measured while the children were taken along */

    /* the quadrants' sums and the PU's total of one candidate from the lanes' sums over their upper (vT) and lower (vB)
     * tiles: blocks of eight lanes hold two tiles side by side, lanes 16 apart the same columns one tile row down */
    __device__ __forceinline__ int fam_sums(int vT, int vB, int c)
    {
        vT += dpp_all<0xB1>(vT); vB += dpp_all<0xB1>(vB);
        vT += dpp_all<0x4E>(vT); vB += dpp_all<0x4E>(vB);
        vT += dpp_all<0x141>(vT); vB += dpp_all<0x141>(vB);
        vT += __shfl_xor(vT, 16); vB += __shfl_xor(vB, 16);
        int *cell = (int*)(x265amd_smem + L::FAM + grp * L::FAM_BYTES);
        if ((l & 23) == 0) { cell[c * 4 + (l >> 3)] = vT; cell[c * 4 + 2 + (l >> 3)] = vB; }
        int tot = vT + vB;
        tot += dpp_all<0x140>(tot);
        return tot;
    }

    /* One refinement round for every PU of the wave: DIRS neighbours of (qx, qy) at distance `step`, plus the
     * centre itself first when CENTRE (the SATD re-measure of motion.cpp:1086, :1117, whose cost replaces
     * bcost). DIRS = 0 with CENTRE measures the centre alone (the closing SATD of subme 0). The candidate loop is
     * unrolled, so which plane and which of the three first rows / phases a candidate reads is fixed at compile time. */
    template<int DIRS, bool CENTRE, bool HPEL, bool FAM = false>
    __device__ __forceinline__ void round_t(bool satd)
    {
        constexpr int step = HPEL ? 2 : 1;
        constexpr unsigned long long SQX = nibs({ 0, 0, 0, -1, 1, -1, -1, 1, 1 }), SQY = nibs({ 0, -1, 1, 0, 0, -1, 1, -1, 1 });
        constexpr int k0 = CENTRE ? 1 : 0, K = k0 + DIRS;
        /* lambda * bits of the candidate this lane folds (bitcost.h:46: the sum wraps at 16 bits) */
        int mvc = 0, tx, ty;
        {
            const int i = l < K ? l + 1 - k0 : 0;
            tx = as_i16(qx + nib(SQX, i) * step); ty = as_i16(qy + nib(SQY, i) * step);
            if (l < K && active) mvc = ((int)ctab[tx - mvpx] + (int)ctab[ty - mvpy]) & 0xffff;
        }
        /* rows of the planes: the candidates' first rows are ymin or ymin + 1 */
        int r0y[3], phy[3];
#pragma unroll
        for (int d = 0; d < 3; d++)
        {
            const int ty = as_i16(qy + (d - 1) * step);
            r0y[d] = ty >> 2; phy[d] = ty & 3;
        }
        int ymin = r0y[0] < r0y[1] ? r0y[0] : r0y[1];
        ymin = r0y[2] < ymin ? r0y[2] : ymin;
#pragma unroll
        for (int d = 0; d < 3; d++) r0y[d] -= ymin;
        const int pbase = L::PLANES + grp * L::PU_BYTES;
        const int first = roff + __mul24(ymin - 3, rstride) - 3;
        wave_sync();
        /* horizontal pass: the lanes of a PU fill its own three planes, so everything they need is in their registers */
#pragma unroll
        for (int p = 0; p < 3; p++)
        {
            const int xp = as_i16(qx + (p - 1) * step);
            const P AS1 *src = ref0 + (first + (xp >> 2));
            const int *htap = (const int*)(x265amd_smem + SM_TAPS) + (xp & 3) * 12;
#pragma unroll
            for (int it = 0; it < G::IPP / RG::LPP; it++)
            {
                const int item = it * RG::LPP + l, r = item >> G::LTW, q = item & (G::TW - 1);
                hps_quad(src + (__mul24(r, rstride) + 4 * q), htap,
                         (int16_t*)(x265amd_smem + pbase + p * G::PLANE) + (4 * q) * G::PROW + r, G::PROW, depth);
            }
        }
        wave_sync();
        /* vertical pass + cost: candidate c of every PU per pass, a lane per (tile, column) or four of them */
        int mycost = 0;
#pragma unroll
        for (int c = 0; c < K; c++)
        {
            const int i = c + 1 - k0, dx = nib(SQX, i), dy = nib(SQY, i);
            /* half-pel rounds: the table lists (dx, -1) right before (dx, +1), one row apart in the same plane and phase */
            if (HPEL && i >= 2 && dy == 1 && nib(SQY, i - 1) == -1 && nib(SQX, i - 1) == dx) continue;   /* measured with its partner */
            if (HPEL && i >= 1 && dy == -1 && c + 1 < K && nib(SQY, i + 1) == 1 && nib(SQX, i + 1) == dx)
            {
                int va = 0, vb = 0;
                if constexpr (FAM)
                {
                    int vaB = 0, vbB = 0;
                    small_column_cost_pair<S>(pbase + (dx + 1) * G::PLANE, r0y[0], phy[0], satd, ct[0], l & 3, depth, va, vb);
                    small_column_cost_pair<S>(pbase + (dx + 1) * G::PLANE, r0y[0], phy[0], satd, ct[RG::ITEMS - 1], l & 3, depth, vaB, vbB);
                    va = fam_sums(va, vaB, c); vb = fam_sums(vb, vbB, c + 1);
                }
                else
                {
#pragma unroll
                    for (int u = 0; u < RG::ITEMS; u++)
                        small_column_cost_pair<S>(pbase + (dx + 1) * G::PLANE, r0y[0], phy[0], satd, ct[u], l & 3, depth, va, vb);
                    va = seg_sum(va, RG::LG); vb = seg_sum(vb, RG::LG);
                }
                if (l == c) mycost = (satd ? va >> 1 : va) + mvc;
                if (l == c + 1) mycost = (satd ? vb >> 1 : vb) + mvc;
                continue;
            }
            int v = 0;
            if constexpr (FAM)
                v = fam_sums(small_column_cost<S>(pbase + (dx + 1) * G::PLANE, r0y[dy + 1], phy[dy + 1], satd, ct[0], l & 3, depth),
                             small_column_cost<S>(pbase + (dx + 1) * G::PLANE, r0y[dy + 1], phy[dy + 1], satd, ct[RG::ITEMS - 1], l & 3, depth), c);
            else
            {
#pragma unroll
                for (int u = 0; u < RG::ITEMS; u++)
                    v += small_column_cost<S>(pbase + (dx + 1) * G::PLANE, r0y[dy + 1], phy[dy + 1], satd, ct[u], l & 3, depth);
                v = seg_sum(v, RG::LG);
            }
            if (l == c) mycost = (satd ? v >> 1 : v) + mvc;   /* every 4x4 abs-sum is even: halving the total = halving per tile */
        }
        /* the reference's strict '<' over the candidates in table order */
        if (CENTRE && active) bcost = grp_lane<RG::LG>(mycost, 0);
        int bdir = 0;
        if (DIRS)
        {
            int key = 0x7fffffff;
            if (l >= k0 && l < K) key = (mycost << 6) | l;
            key = grp_min<RG::LG>(key);
            if (active && (key >> 6) < bcost) { bcost = key >> 6; bdir = (key & 63) - k0 + 1; }
        }
        if constexpr (FAM)
        {
            /* the children: the same sequence on their quadrants' sums and their own lambda * bits */
            wave_sync();
            const int *cell = (const int*)(x265amd_smem + L::FAM + grp * L::FAM_BYTES);
            bool agree = true;
#pragma unroll
            for (int q = 0; q < 4; q++)
            {
                int cc = 0;
                if (l < K && quad)
                {
                    const int sum = cell[l * 4 + q];
                    cc = (satd ? sum >> 1 : sum) + (((int)ctab[tx - cell[36 + q]] + (int)ctab[ty - cell[40 + q]]) & 0xffff);
                }
                if (CENTRE && quad) cb[q] = grp_lane<RG::LG>(cc, 0);
                int cdir = 0;
                if (DIRS)
                {
                    int key = 0x7fffffff;
                    if (l >= k0 && l < K) key = (cc << 6) | l;
                    key = grp_min<RG::LG>(key);
                    if (quad && (key >> 6) < cb[q]) { cb[q] = key >> 6; cdir = (key & 63) - k0 + 1; }
                }
                agree = agree && cdir == bdir;
            }
            quad = quad && agree;
            if (quad) cadd += (unsigned)K;
            wave_sync();        /* the cells are the next round's */
        }
        qx = as_i16(qx + nib(SQX, bdir) * step); qy = as_i16(qy + nib(SQY, bdir) * step);
        if (active) nsub += (unsigned)K;
    }

    /* the rounds the workload table can ask for (motion.cpp:53-63: four or eight directions) */
    __device__ __forceinline__ void round(int step, int dirs, bool satd, bool centre)
    {
        if constexpr (S == 16)
            if (__ballot(quad) != 0)
            {
                /* a wave with children in tow: the rounds of the levels that matter for throughput (eight directions) */
                if (step == 2 && dirs == 8) { if (centre) round_t<8, true, true, true>(satd); else round_t<8, false, true, true>(satd); return; }
                if (step == 1 && dirs == 8) { if (centre) round_t<8, true, false, true>(satd); else round_t<8, false, false, true>(satd); return; }
                quad = false;       /* other workloads: the children are refined on their own */
            }
        if (step == 2)
        {
            if (dirs == 8) { if (centre) round_t<8, true, true>(satd); else round_t<8, false, true>(satd); }
            else if (dirs == 4) { if (centre) round_t<4, true, true>(satd); else round_t<4, false, true>(satd); }
        }
        else if (dirs == 8) { if (centre) round_t<8, true, false>(satd); else round_t<8, false, false>(satd); }
        else if (dirs == 4) { if (centre) round_t<4, true, false>(satd); else round_t<4, false, false>(satd); }
        else if (centre) round_t<0, true, false>(satd);
    }
};

/* one wave's PUs: the lanes of group `grp` refine job j (valid: there is one; j must be a readable index either way) */
template<int S, typename P>
__device__ __forceinline__ void refine_small_group(const SearchArgs &a, int j, bool valid)
{
    typedef SmallGeo<S> G;
    typedef RefineGeo<S> RG;
    RefineSmall<S, P> R;
    R.depth = sizeof(P) == 1 ? 8 : a.depth;
    R.lane = (int)threadIdx.x; R.grp = R.lane >> RG::LG; R.l = R.lane & (RG::LPP - 1);
    /* a PU the STAR kernels handed on is refined by the list-driven launch that follows its search */
    if (!a.list && a.mark && a.mark[j]) valid = false;
    const int *job = (const int*)(a.jobs + j);
    const int d0 = job[0], d1 = job[1], d4 = job[4];
    const int pux = as_i16(d0), puy = d0 >> 16;
    int *res = (int*)(a.results + j);
    const int r0 = res[0];
    R.qx = as_i16(r0); R.qy = r0 >> 16; R.bcost = res[1]; R.nsub = (unsigned)res[3];
    R.mvpx = as_i16(d4); R.mvpy = d4 >> 16;
    /* a PU whose full-pel cost is 0 is not refined (motion.cpp:1073-1076); PUs of other sizes are final already */
    R.active = valid && (d1 & 0xffff) == G::SIZE_CODE && R.bcost > 0;      /* a negative cost marks a job that was refused */
    if (__ballot(R.active) == 0) return;
    R.quad = false; R.cadd = 0;
    if constexpr (S == 16)
    {
        if (!a.list && a.cjobs)
        {
            /* lane q < 4 looks at child q: PU 4 j + q of the children's list, if it is the 8x8 PU of the same pictures
             * at that place in the quadtree, not handed on by the STAR kernels, starting from the parent's MV, cost above 0 */
            const int q = R.l & 3;
            bool ok = R.active && 4 * j + 3 < a.ncjobs;
            int bc = 0, c4 = 0;
            if (ok)
            {
                const int *cj = (const int*)(a.cjobs + 4 * j + q);
                const int *cr = (const int*)(a.cresults + 4 * j + q);
                const int c0 = cj[0], c1 = cj[1];
                c4 = cj[4]; bc = cr[1];
                ok = (c1 & 0xffff) == SmallGeo<8>::SIZE_CODE && (c1 >> 24) == (d1 >> 24) && as_i16(c0) == pux + 8 * (q & 1) &&
                     (c0 >> 16) == puy + 8 * (q >> 1) && cr[0] == r0 && bc > 0 && !a.cmark[4 * j + q];
            }
            R.quad = (grp_bits16<RG::LG>(ok) & 0xf) == 0xf;
            int *cell = (int*)(x265amd_smem + RefineLds<S>::FAM + R.grp * RefineLds<S>::FAM_BYTES);
            if (R.l < 4) { cell[36 + q] = as_i16(c4); cell[40 + q] = c4 >> 16; }
#pragma unroll
            for (int k = 0; k < 4; k++) R.cb[k] = grp_lane<RG::LG>(bc, k);
        }
    }
    const P AS1 *fenc0;
    int fstride;
    small_planes<P>(a, d1, fenc0, fstride, R.ref0, R.rstride);
    R.roff = puy * R.rstride + pux;
    R.ctab = (const uint16_t AS1*)a.ctab;
#pragma unroll
    for (int u = 0; u < RG::ITEMS; u++) small_source_column<S>(fenc0, fstride, pux, puy, R.l + u * RG::LPP, R.ct[u]);
    /* SubpelWorkload, motion.cpp:53-63, and the round sequence of :1100-1141 */
    const int hpel_iters = kWorkload[a.subme][0], hpel_dirs = kWorkload[a.subme][1];
    const int qpel_iters = kWorkload[a.subme][2], qpel_dirs = kWorkload[a.subme][3], hpel_satd = kWorkload[a.subme][4];
    bool centre = hpel_satd != 0;
    for (int it = 0; it < hpel_iters; it++)
    {
        R.round(2, hpel_dirs, hpel_satd != 0, centre);
        centre = false;
    }
    centre = !hpel_satd;
    if (centre && !qpel_iters) R.round(1, 0, true, true);
    for (int it = 0; it < qpel_iters; it++)
    {
        R.round(1, qpel_dirs, true, centre);
        centre = false;
    }
    if (R.l == 0 && R.active)
    {
        res[0] = (R.qx & 0xffff) | (R.qy << 16);
        res[1] = R.bcost;
        res[3] = (int)R.nsub;
    }
    if constexpr (S == 16)
        if (R.quad && R.l < 4)
        {
            int *cr = (int*)(a.cresults + 4 * j + R.l);
            cr[0] = (R.qx & 0xffff) | (R.qy << 16);
            cr[1] = R.l == 0 ? R.cb[0] : (R.l == 1 ? R.cb[1] : (R.l == 2 ? R.cb[2] : R.cb[3]));
            cr[3] += (int)R.cadd;
            a.cdone[4 * j + R.l] = 1;
        }
}

template<int S, typename P>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(S == 16 ? OCC_R16 : OCC_R8))) void refine_small_kernel(SearchArgs a)
{
    typedef RefineGeo<S> RG;
    const int lane = (int)threadIdx.x, grp = lane >> RG::LG;
    small_load_taps<P>(lane, sizeof(P) == 1 ? 8 : a.depth);
    if (a.list)
    {
        /* the launch's list of marked PUs, every wave walking it */
        const int total = uni(a.redo[0]), ngroups = (total + RG::NPU - 1) / RG::NPU;
        for (int k = (int)blockIdx.x; k < ngroups; k += (int)gridDim.x)
        {
            wave_sync();
            const int i = k * RG::NPU + grp;
            refine_small_group<S, P>(a, a.redo[REDO_LIST + (i < total ? i : total - 1)], i < total);
        }
        return;
    }
    if (a.done)
    {
        /* the children of a family launch: most are refined already. A wave looks at 64 consecutive PUs and refines
         * those left over, a group of lanes each at a time */
        const int base = (int)blockIdx.x * 64, jl = base + lane;
        unsigned long long m = __ballot(jl < a.njobs && !a.done[jl]);
        while (m)
        {
            int j = base;
            bool valid = false;
#pragma unroll
            for (int g = 0; g < RG::NPU; g++)
                if (m)
                {
                    if (g == grp) { j = base + (int)__builtin_ctzll(m); valid = true; }
                    m &= m - 1;
                }
            wave_sync();
            refine_small_group<S, P>(a, j, valid);
        }
        return;
    }
    /* every PU of the launch, a group per wave */
    const int ngroups = (a.njobs + RG::NPU - 1) / RG::NPU, g = xcd_remap((int)blockIdx.x, ngroups);
    if (g >= ngroups) return;
    const int j = g * RG::NPU + grp;
    refine_small_group<S, P>(a, j < a.njobs ? j : a.njobs - 1, j < a.njobs);
}

/* ---------------------------------------------------------------- sub-pel refinement of 32x32 and 64x64 PUs
 *
 * The same rounds for the large PUs: a workgroup of two waves per 32x32 PU and of four waves per 64x64 PU, every lane
 * looking after two or four (tile, column) pairs, one tile row band apart, where the one-wave search walks the PU's 64 or 256 tiles through
 * its general candidate machinery. The planes of a round are filled by all waves together, one after the other
 * through one buffer; every wave measures its tiles of every candidate and adds the wave's sum to the candidate's cell in LDS (one cell array per round,
 * zeroed at the start, so no round has to wait for cells to be cleared); after the barrier every wave reads the same
 * totals and takes the same decisions on its own copy of the PU's values. Workgroups stay small on purpose: a
 * sixteen-wave workgroup per 64x64 PU waits for a compute unit with that much room while launches of small PUs share
 * the chip, and ended every step alone. */
template<int S> struct BigGeo
{
    static_assert(S == 32 || S == 64, "32x32 or 64x64 PUs");
    /* measured on the headline step: one wave per 32x32 PU (four pairs per lane) issues 9 % fewer instructions and is
     * 0.4 % slower than two (its LDS leaves 4.5 waves per SIMD); eight waves per 64x64 PU are 1.4 % slower than four */
    static constexpr int WAVES = S == 32 ? 2 : 4;
    static constexpr int THREADS = WAVES * 64;
    static constexpr int ITEMS = SmallGeo<S>::LPP / THREADS;        /* (tile, column) pairs per lane */
};
template<int S> struct BigLds
{
    typedef SmallGeo<S> G;
    static constexpr int CELLS = SM_FIXED;                          /* per round 16 candidates' sums, one per quadrant of the PU */
    static constexpr int ROUNDS = 5;                                /* motion.cpp:53-63: at most 2 + 2 rounds, or the closing re-measure */
    static constexpr int VOTES = CELLS + ROUNDS * 256;              /* per round the four children's decisions */
    static constexpr int PLANES = VOTES + ROUNDS * 16;
    static constexpr int TOTAL = PLANES + G::PLANE + 32;          /* one plane at a time */
};

template<int S, typename P>
__device__ __forceinline__ void refine_big_pu(const SearchArgs &a, int j)
{
    const int depth = sizeof(P) == 1 ? 8 : a.depth;
    typedef SmallGeo<S> G;
    typedef BigLds<S> L;
    constexpr int THREADS = BigGeo<S>::THREADS, ITEMS = BigGeo<S>::ITEMS;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int *job = (const int*)(a.jobs + j);
    const int d0 = uni(job[0]), d1 = uni(job[1]), d4 = uni(job[4]);
    const int pux = as_i16(d0), puy = d0 >> 16;
    int *res = (int*)(a.results + j);
    const int r0 = uni(res[0]);
    int qx = as_i16(r0), qy = r0 >> 16, bcost = uni(res[1]);
    unsigned nsub = (unsigned)uni(res[3]);
    const int mvpx = as_i16(d4), mvpy = d4 >> 16;
    /* a PU whose full-pel cost is 0 is not refined (motion.cpp:1073-1076), a negative cost marks a job that was
     * refused; PUs of other sizes are final already */
    if ((d1 & 0xffff) != G::SIZE_CODE || bcost <= 0) return;
    if (a.list != 1 && a.done && a.done[j]) return;     /* refined together with its parent already */
    /* a PU the STAR kernels handed on is refined by the list-driven launch that follows its search */
    if (a.list != 1 && a.mark && a.mark[j]) return;
    const P AS1 *fenc0, *ref0;
    int fstride, rstride;
    small_planes<P>(a, d1, fenc0, fstride, ref0, rstride);
    const int roff = puy * rstride + pux;
    const uint16_t AS1 *ctab = (const uint16_t AS1*)a.ctab;
    for (int i = tid; i < L::ROUNDS * 68; i += THREADS) ((int*)(x265amd_smem + L::CELLS))[i] = 0;
    /* A 64x64 PU directly followed in the list by its four 32x32 children (raster order, same pictures), all five
     * starting from the same MV: SAD and SATD are sums over 4x4 tiles, so a candidate's distortion for a child is the
     * sum over the child's quadrant -- wave w keeps the sums of quadrant-aligned halves apart and takes child w's
     * decisions beside the parent's. The children stay with the parent as long as every decision moves all five to
     * the same MV; if one differs, the parent goes on alone and the children are left for their own refinement. */
    const int wave = tid >> 6;
    bool quad = false;
    int cbcost = 0, cmvpx = 0, cmvpy = 0;
    unsigned cnsub = 0;
    if constexpr (S == 64)
    {
        /* (the list-driven launch refines PU by PU: the children of a marked parent were refined on their own) */
        if (a.list != 1 && a.done && j + 4 < a.njobs)
        {
            quad = true;
#pragma unroll
            for (int q = 0; q < 4; q++)
            {
                const int *cj = (const int*)(a.jobs + j + 1 + q);
                const int *cr = (const int*)(a.results + j + 1 + q);
                const int c0 = uni(cj[0]), c1 = uni(cj[1]), c4 = uni(cj[4]), m = uni(cr[0]), bc = uni(cr[1]);
                quad = quad && (c1 & 0xffff) == SmallGeo<32>::SIZE_CODE && (c1 >> 24) == (d1 >> 24) &&
                       as_i16(c0) == pux + 32 * (q & 1) && (c0 >> 16) == puy + 32 * (q >> 1) && m == r0 && bc > 0 &&
                       !(a.mark && a.mark[j + 1 + q]);    /* a marked child's record is not ready yet */
                if (q == wave) { cbcost = bc; cmvpx = as_i16(c4); cmvpy = c4 >> 16; cnsub = (unsigned)uni(cr[3]); }
            }
        }
    }
    ColSrc ct[ITEMS];
#pragma unroll
    for (int u = 0; u < ITEMS; u++) small_source_column<S>(fenc0, fstride, pux, puy, tid + u * THREADS, ct[u]);
    int rnd = 0;
    __syncthreads();

    /* one refinement round, as RefineSmall::round_t */
    auto round = [&](int step, int dirs, bool satd, bool centre)
    {
        constexpr unsigned long long SQX = nibs({ 0, 0, 0, -1, 1, -1, -1, 1, 1 }), SQY = nibs({ 0, -1, 1, 0, 0, -1, 1, -1, 1 });
        const int k0 = centre ? 1 : 0, K = k0 + dirs;
        int mvc = 0, cmvc = 0;
        {
            const int i = lane < K ? lane + 1 - k0 : 0;
            const int tx = as_i16(qx + nib(SQX, i) * step), ty = as_i16(qy + nib(SQY, i) * step);
            if (lane < K) mvc = ((int)ctab[tx - mvpx] + (int)ctab[ty - mvpy]) & 0xffff;
            if (lane < K && quad) cmvc = ((int)ctab[tx - cmvpx] + (int)ctab[ty - cmvpy]) & 0xffff;
        }
        const int ya = as_i16(qy - step) >> 2, yb = qy >> 2, yc = as_i16(qy + step) >> 2;
        int ymin = ya < yb ? ya : yb;
        ymin = yc < ymin ? yc : ymin;
        /* plane by plane through one buffer: all waves filter the plane (horizontal pass), then every wave measures its
         * tiles of the candidates in that column of the square (vertical pass + cost). One plane instead of three keeps
         * the workgroup's LDS small enough to find room on a compute unit shared with the small PUs' kernels. A lane's
         * first half of its pairs lies in the upper half of the PU and the second in the lower (tile rows wave + 4 u),
         * lanes 0-31 in the left half and 32-63 in the right: the sums of the four quadrants stay apart (lane c: left,
         * lane c + 32: right) */
        int partT = 0, partB = 0;
        bool filled = false;
#pragma unroll
        for (int p = 0; p < 3; p++)
        {
            if (p != 1 && dirs < 4) continue;           /* the closing re-measure has the centre only */
            if (filled) __syncthreads();                /* every wave is done with the plane before */
            filled = true;
            {
                const int xp = as_i16(qx + (p - 1) * step);
                const P AS1 *src = ref0 + (roff + __mul24(ymin - 3, rstride) + (xp >> 2) - 3);
                const int *tp = (const int*)(x265amd_smem + SM_TAPS) + (xp & 3) * 12;
                int16_t *plane = (int16_t*)(x265amd_smem + L::PLANES);
                for (int item = tid; item < G::IPP; item += THREADS)
                {
                    const int r = item >> G::LTW, q = item & (G::TW - 1);
                    hps_quad(src + (__mul24(r, rstride) + 4 * q), tp, plane + (4 * q) * G::PROW + r, G::PROW, depth);
                }
            }
            __syncthreads();
            for (int c = 0; c < K; c++)
            {
                const int i = c + 1 - k0;
                if (nib(SQX, i) != p - 1) continue;
                const int ty = as_i16(qy + nib(SQY, i) * step);
                const int poff = L::PLANES, rr = (ty >> 2) - ymin;
                /* half-pel rounds: (dx, -1) and (dx, +1) follow each other in the table and are one row apart in the
                 * same plane and phase -- measured together */
                if (step == 2 && i >= 1 && nib(SQY, i) == -1 && c + 1 < K && nib(SQY, i + 1) == 1 && nib(SQX, i + 1) == nib(SQX, i))
                {
                    int vaT = 0, vbT = 0, vaB = 0, vbB = 0;
#pragma unroll
                    for (int u = 0; u < ITEMS; u++)
                    {
                        if (u < ITEMS / 2) small_column_cost_pair<S>(poff, rr, ty & 3, satd, ct[u], tid & 3, depth, vaT, vbT);
                        else small_column_cost_pair<S>(poff, rr, ty & 3, satd, ct[u], tid & 3, depth, vaB, vbB);
                    }
                    vaT = seg_sum(vaT, 5); vbT = seg_sum(vbT, 5); vaB = seg_sum(vaB, 5); vbB = seg_sum(vbB, 5);
                    if ((lane & 31) == c) { partT = vaT; partB = vaB; }
                    if ((lane & 31) == c + 1) { partT = vbT; partB = vbB; }
                    c++;
                    continue;
                }
                int vT = 0, vB = 0;
#pragma unroll
                for (int u = 0; u < ITEMS; u++)
                {
                    const int v = small_column_cost<S>(poff, rr, ty & 3, satd, ct[u], tid & 3, depth);
                    if (u < ITEMS / 2) vT += v; else vB += v;
                }
                vT = seg_sum(vT, 5); vB = seg_sum(vB, 5);
                if ((lane & 31) == c) { partT = vT; partB = vB; }
            }
        }
        int *cells = (int*)(x265amd_smem + L::CELLS) + rnd * 64;           /* [candidate][TL, TR, BL, BR] */
        if ((lane & 31) < K)
        {
            atomicAdd(cells + (lane & 31) * 4 + (lane >> 5), partT);
            atomicAdd(cells + (lane & 31) * 4 + 2 + (lane >> 5), partB);
        }
        __syncthreads();
        int mycost = 0, ccost = 0;
        if (lane < K)
        {
            const int4 q4 = *(const int4*)(cells + lane * 4);
            const int v = q4.x + q4.y + q4.z + q4.w;
            mycost = (satd ? v >> 1 : v) + mvc;     /* every 4x4 abs-sum is even: halving the total = halving per tile */
            const int vq = wave == 0 ? q4.x : (wave == 1 ? q4.y : (wave == 2 ? q4.z : q4.w));
            ccost = (satd ? vq >> 1 : vq) + cmvc;
        }
        if (centre) bcost = __shfl(mycost, 0);
        int key = 0x7fffffff;
        if (lane >= k0 && lane < K) key = (mycost << 6) | lane;
        key = wave_min(key);
        int bdir = 0;
        if ((key >> 6) < bcost) { bcost = key >> 6; bdir = (key & 63) - k0 + 1; }
        if (quad)
        {
            /* child `wave`: the same sequence on its quadrant's sums and its own lambda * bits */
            if (centre) cbcost = __shfl(ccost, 0);
            int ckey = 0x7fffffff;
            if (lane >= k0 && lane < K) ckey = (ccost << 6) | lane;
            ckey = wave_min(ckey);
            int cdir = 0;
            if ((ckey >> 6) < cbcost) { cbcost = ckey >> 6; cdir = (ckey & 63) - k0 + 1; }
            cnsub += (unsigned)K;
            int *votes = (int*)(x265amd_smem + L::VOTES) + rnd * 4;
            if (lane == 0) votes[wave] = cdir;
            __syncthreads();
            quad = votes[0] == bdir && votes[1] == bdir && votes[2] == bdir && votes[3] == bdir;
        }
        qx = as_i16(qx + nib(SQX, bdir) * step); qy = as_i16(qy + nib(SQY, bdir) * step);
        nsub += (unsigned)K;
        rnd++;
    };

    const int hpel_iters = kWorkload[a.subme][0], hpel_dirs = kWorkload[a.subme][1];
    const int qpel_iters = kWorkload[a.subme][2], qpel_dirs = kWorkload[a.subme][3], hpel_satd = kWorkload[a.subme][4];
    bool centre = hpel_satd != 0;
    for (int it = 0; it < hpel_iters; it++)
    {
        round(2, hpel_dirs, hpel_satd != 0, centre);
        centre = false;
    }
    centre = !hpel_satd;
    if (centre && !qpel_iters) round(1, 0, true, true);
    for (int it = 0; it < qpel_iters; it++)
    {
        round(1, qpel_dirs, true, centre);
        centre = false;
    }
    if (tid == 0)
    {
        res[0] = (qx & 0xffff) | (qy << 16);
        res[1] = bcost;
        res[3] = (int)nsub;
    }
    if (quad && lane == 0)
    {
        int *cr = (int*)(a.results + j + 1 + wave);
        cr[0] = (qx & 0xffff) | (qy << 16);
        cr[1] = cbcost;
        cr[3] = (int)cnsub;
        a.done[j + 1 + wave] = 1;
    }
}

template<int S, typename P>
__global__ __launch_bounds__(BigGeo<S>::THREADS) __attribute__((amdgpu_waves_per_eu(S == 64 ? OCC_RB64 : OCC_RB32))) void refine_big_kernel(SearchArgs a)
{
    small_load_taps<P>((int)threadIdx.x, sizeof(P) == 1 ? 8 : a.depth);
    /* a workgroup per PU of the launch; or (a.list) a list of its PUs, every workgroup walking it: the marked PUs (1), or
     * the unmarked squares of the launch's sizes of a list that may hold few of them (2: otherwise as without a list) */
    const int n = a.list ? uni(a.redo[0]) : a.njobs;
    for (int k = (int)blockIdx.x; k < n; k += a.list ? (int)gridDim.x : n)
    {
        __syncthreads();        /* the workgroup's LDS belongs to the PU before until here */
        const int j = a.list ? uni(a.redo[REDO_LIST + k]) : xcd_remap(k, a.njobs);
        if (j < a.njobs) refine_big_pu<S, P>(a, j);
    }
}

/* ---------------------------------------------------------------- sub-pel refinement of every other partition
 *
 * The rounds of motion.cpp:1071-1141 for a PU of any luma partition (primitives.h:65-73: 2NxN, Nx2N, the asymmetric
 * pairs, and squares that are not the size the launch's fixed-shape kernels take), one wave per PU with the PU's width
 * and height as values: the same planes of luma_hps intermediates (one buffer, a column of candidates at a time, as in
 * refine_big_kernel) and the same vertical pass per (tile, column) pair -- a lane takes every 64th pair of the PU, up to
 * MAXI of them (the launch's largest PU decides MAXI: 1 up to 16 samples, 4 up to 32, 12 up to 64). The one-wave search
 * measures these candidates through its general machinery at about three times the instructions per sample. Measured on
 * 1080p lists of one shape, 16 pictures per launch, STAR: 64x48 0.56 -> 0.38 ms, 64x32 0.83 -> 0.63, 32x24 0.68 -> 0.52,
 * 16x12 1.31 -> 1.09, 16x8 2.06 -> 1.91; the integer stage of these PUs stays a wave per PU and is what is left. */
__device__ __forceinline__ int wave_total(int v);
template<int MAXI> struct AnyLds
{
    static constexpr int DIM = MAXI == 1 ? 16 : (MAXI == 4 ? 32 : 64);
    static constexpr int PLANES = SM_FIXED;
    static constexpr int TOTAL = PLANES + DIM * (DIM + 10) * 2 + 32;
};

template<typename P, int MAXI>
__device__ __forceinline__ void refine_any_pu(const SearchArgs &a, int j)
{
    const int depth = sizeof(P) == 1 ? 8 : a.depth;
    typedef AnyLds<MAXI> L;
    const int lane = (int)threadIdx.x;
    const int *job = (const int*)(a.jobs + j);
    const int d0 = uni(job[0]), d1 = uni(job[1]), d4 = uni(job[4]);
    const int pux = as_i16(d0), puy = d0 >> 16, w = d1 & 0xff, h = (d1 >> 8) & 0xff;
    /* the squares of the launch's fixed-shape sizes have kernels of their own; a PU larger than the launch reserved
     * LDS for was refused by the search */
    if (w == h && (w == a.staged || w == a.staged2)) return;
    if (w > L::DIM || h > L::DIM || !any_refines(w, h) || ((w | h) & 3)) return;
    if (!a.list && a.mark && a.mark[j]) return;         /* handed on by the STAR kernels: the list-driven launch's */
    int *res = (int*)(a.results + j);
    const int r0 = uni(res[0]);
    int qx = as_i16(r0), qy = r0 >> 16, bcost = uni(res[1]);
    unsigned nsub = (unsigned)uni(res[3]);
    /* a PU whose full-pel cost is 0 is not refined (motion.cpp:1073-1076), a negative cost marks a job that was refused */
    if (bcost <= 0) return;
    const int mvpx = as_i16(d4), mvpy = d4 >> 16;
    const P AS1 *fenc0, *ref0;
    int fstride, rstride;
    small_planes<P>(a, d1, fenc0, fstride, ref0, rstride);
    const int roff = puy * rstride + pux;
    const uint16_t AS1 *ctab = (const uint16_t AS1*)a.ctab;
    const int tw = w >> 2, inv = (65536 + tw - 1) / tw;     /* tiles in a row; item / tw = item * inv >> 16 (items < 4096) */
    const int npairs = (w * h) >> 2, prow = h + 10, rows = h + 8, nitems = rows * tw;
    ColSrc ct[MAXI];
#pragma unroll
    for (int u = 0; u < MAXI; u++)
    {
        int l = lane + 64 * u;
        if (l >= npairs) l = npairs - 4 + (lane & 3);       /* the pair of the last tile with the lane's column: never summed */
        const int t = l >> 2, ty = (t * inv) >> 16, tx = t - ty * tw;
        ct[u].ty = ty; ct[u].col = tx * 4 + (l & 3);
        const P AS1 *src = fenc0 + (long)(puy + ty * 4) * fstride + pux + ct[u].col;
        uint32_t f[4];
#pragma unroll
        for (int y = 0; y < 4; y++) f[y] = (uint32_t)src[(long)y * fstride];
        ct[u].f01 = f[0] | (f[1] << 16); ct[u].f23 = f[2] | (f[3] << 16);
    }

    auto round = [&](int step, int dirs, bool satd, bool centre)
    {
        constexpr unsigned long long SQX = nibs({ 0, 0, 0, -1, 1, -1, -1, 1, 1 }), SQY = nibs({ 0, -1, 1, 0, 0, -1, 1, -1, 1 });
        const int k0 = centre ? 1 : 0, K = k0 + dirs;
        int mvc = 0;
        {
            const int i = lane < K ? lane + 1 - k0 : 0;
            const int tx = as_i16(qx + nib(SQX, i) * step), ty = as_i16(qy + nib(SQY, i) * step);
            if (lane < K) mvc = ((int)ctab[tx - mvpx] + (int)ctab[ty - mvpy]) & 0xffff;
        }
        const int ya = as_i16(qy - step) >> 2, yb = qy >> 2, yc = as_i16(qy + step) >> 2;
        int ymin = ya < yb ? ya : yb;
        ymin = yc < ymin ? yc : ymin;
        int mycost = 0;
#pragma unroll 1
        for (int p = 0; p < 3; p++)
        {
            if (p != 1 && dirs < 4) continue;           /* the closing re-measure has the centre only */
            wave_sync();                                /* the plane before has been read */
            {
                const int xp = as_i16(qx + (p - 1) * step);
                const P AS1 *src = ref0 + (roff + __mul24(ymin - 3, rstride) + (xp >> 2) - 3);
                const int *tp = (const int*)(x265amd_smem + SM_TAPS) + (xp & 3) * 12;
                int16_t *plane = (int16_t*)(x265amd_smem + L::PLANES);
                for (int item = lane; item < nitems; item += 64)
                {
                    const int r = (item * inv) >> 16, q = item - r * tw;
                    hps_quad(src + (__mul24(r, rstride) + 4 * q), tp, plane + (4 * q) * prow + r, prow, depth);
                }
            }
            wave_sync();
            for (int c = 0; c < K; c++)
            {
                const int i = c + 1 - k0;
                if (nib(SQX, i) != p - 1) continue;
                const int ty = as_i16(qy + nib(SQY, i) * step);
                const int rr = (ty >> 2) - ymin;
                /* half-pel rounds: (dx, -1) and (dx, +1) follow each other in the table and are one row apart in the
                 * same plane and phase -- measured together */
                if (step == 2 && i >= 1 && nib(SQY, i) == -1 && c + 1 < K && nib(SQY, i + 1) == 1 && nib(SQX, i + 1) == nib(SQX, i))
                {
                    int va = 0, vb = 0;
#pragma unroll
                    for (int u = 0; u < MAXI; u++)
                    {
                        if (64 * u >= npairs) break;        /* (the same for every lane) */
                        int o[5];
                        column_rows<5>(L::PLANES, prow, rr, ty & 3, ct[u], depth, o);
                        const int ca = small_rows_cost(o[0], o[1], o[2], o[3], satd, ct[u], lane & 3, depth);
                        const int cb = small_rows_cost(o[1], o[2], o[3], o[4], satd, ct[u], lane & 3, depth);
                        if (lane + 64 * u < npairs) { va += ca; vb += cb; }
                    }
                    va = wave_total(va); vb = wave_total(vb);
                    if (lane == c) mycost = (satd ? va >> 1 : va) + mvc;
                    if (lane == c + 1) mycost = (satd ? vb >> 1 : vb) + mvc;
                    c++;
                    continue;
                }
                int v = 0;
#pragma unroll
                for (int u = 0; u < MAXI; u++)
                {
                    if (64 * u >= npairs) break;
                    int o[4];
                    column_rows<4>(L::PLANES, prow, rr, ty & 3, ct[u], depth, o);
                    const int cv = small_rows_cost(o[0], o[1], o[2], o[3], satd, ct[u], lane & 3, depth);
                    if (lane + 64 * u < npairs) v += cv;
                }
                v = wave_total(v);
                if (lane == c) mycost = (satd ? v >> 1 : v) + mvc;      /* every 4x4 abs-sum is even: halving the total = halving per tile */
            }
        }
        if (centre) bcost = __shfl(mycost, 0);
        int key = 0x7fffffff;
        if (lane >= k0 && lane < K) key = (mycost << 6) | lane;
        key = wave_min(key);
        int bdir = 0;
        if ((key >> 6) < bcost) { bcost = key >> 6; bdir = (key & 63) - k0 + 1; }
        qx = as_i16(qx + nib(SQX, bdir) * step); qy = as_i16(qy + nib(SQY, bdir) * step);
        nsub += (unsigned)K;
    };

    const int hpel_iters = kWorkload[a.subme][0], hpel_dirs = kWorkload[a.subme][1];
    const int qpel_iters = kWorkload[a.subme][2], qpel_dirs = kWorkload[a.subme][3], hpel_satd = kWorkload[a.subme][4];
    bool centre = hpel_satd != 0;
    for (int it = 0; it < hpel_iters; it++)
    {
        round(2, hpel_dirs, hpel_satd != 0, centre);
        centre = false;
    }
    centre = !hpel_satd;
    if (centre && !qpel_iters) round(1, 0, true, true);
    for (int it = 0; it < qpel_iters; it++)
    {
        round(1, qpel_dirs, true, centre);
        centre = false;
    }
    if (lane == 0)
    {
        res[0] = (qx & 0xffff) | (qy << 16);
        res[1] = bcost;
        res[3] = (int)nsub;
    }
}

/* chunk: PUs of the job list a wave looks at together (a power of two up to 64, a lane each) before it refines those
 * that are this kernel's one after the other: a list of squares of the fixed-shape sizes then costs a wave per `chunk`
 * PUs and not a wave per PU. The host picks it so that a short list still spreads over the chip. */
template<typename P, int MAXI>
__global__ __launch_bounds__(64) void refine_any_kernel(SearchArgs a, int chunk)
{
    small_load_taps<P>((int)threadIdx.x, sizeof(P) == 1 ? 8 : a.depth);
    if (a.list)
    {
        /* the launch's list of PUs the STAR kernels left to others, every wave walking it */
        const int n = uni(a.redo[0]);
        for (int k = (int)blockIdx.x; k < n; k += (int)gridDim.x)
        {
            wave_sync();
            const int j = uni(a.redo[REDO_LIST + k]);
            if (j < a.njobs) refine_any_pu<P, MAXI>(a, j);
        }
        return;
    }
    const int lane = (int)threadIdx.x;
    for (int base = (int)blockIdx.x * chunk; base < a.njobs; base += (int)gridDim.x * chunk)
    {
        const int j = base + lane;
        bool mine = false;
        if (lane < chunk && j < a.njobs)
        {
            const int d1 = ((const int*)(a.jobs + j))[1], w = d1 & 0xff, h = (d1 >> 8) & 0xff;
            mine = !(w == h && (w == a.staged || w == a.staged2)) && any_refines(w, h) && a.results[j].cost > 0;
        }
        unsigned long long m = __ballot(mine);
        while (m)
        {
            wave_sync();
            refine_any_pu<P, MAXI>(a, base + (int)__builtin_ctzll(m));
            m &= m - 1;
        }
    }
}

/* ---------------------------------------------------------------- sub-pel refinement of the small other partitions
 *
 * The same rounds for the shapes of tiny_class, several PUs per wave: 2^LG lanes per PU (a lane per (tile, column) pair),
 * 64 >> LG PUs side by side, each with planes of its own in LDS. The PUs of a wave may differ in shape inside their
 * class; everything about a PU is a value its lanes hold. A candidate's cost is a segmented sum; the reference's strict
 * '<' over the candidates in table order is kept as a running (cost, direction) per PU. */
template<int LG> struct TinyLds
{
    static constexpr int LPP = 1 << LG, NPU = 64 / LPP;
    static constexpr int PLANE = LG == 3 ? 256 : (LG == 4 ? 512 : 640);        /* the largest plane of the class: (h + 10) w int16 */
    static constexpr int PLANES = SM_FIXED;
    static constexpr int TOTAL = PLANES + NPU * 3 * PLANE + 32;
};
#define TINY_LDS_MAX (SM_FIXED + 8 * 3 * 256 + 32)

/* the PUs of class LG among jobs[base + 0 .. 63] (or among the redo list's entries there), NPU at a time */
template<typename P, int LG>
__device__ __forceinline__ void refine_tiny_pass(const SearchArgs &a, int jmine, bool mine)
{
    typedef TinyLds<LG> L;
    constexpr int LPP = L::LPP, NPU = L::NPU;
    const int depth = sizeof(P) == 1 ? 8 : a.depth;
    const int lane = (int)threadIdx.x, grp = lane >> LG, l = lane & (LPP - 1);
    unsigned long long m = __ballot(mine);
    while (m)
    {
        /* group g takes the g-th PU that is left; a group without one goes along with the first (and writes nothing) */
        int src_lane = (int)__builtin_ctzll(m);
        bool valid = false;
#pragma unroll
        for (int g = 0; g < NPU; g++)
            if (m)
            {
                if (g == grp) { src_lane = (int)__builtin_ctzll(m); valid = true; }
                m &= m - 1;
            }
        const int j = __shfl(jmine, src_lane);
        wave_sync();
        const int *job = (const int*)(a.jobs + j);
        const int d0 = job[0], d1 = job[1], d4 = job[4];
        const int pux = as_i16(d0), puy = d0 >> 16, w = d1 & 0xff, h = (d1 >> 8) & 0xff;
        int *res = (int*)(a.results + j);
        const int r0 = res[0];
        int qx = as_i16(r0), qy = r0 >> 16, bcost = res[1];
        unsigned nsub = (unsigned)res[3];
        const int mvpx = as_i16(d4), mvpy = d4 >> 16;
        const bool active = valid;
        const P AS1 *fenc0, *ref0;
        int fstride, rstride;
        small_planes<P>(a, d1, fenc0, fstride, ref0, rstride);
        const int roff = puy * rstride + pux;
        const uint16_t AS1 *ctab = (const uint16_t AS1*)a.ctab;
        const int ltw = w == 16 ? 2 : (w == 8 ? 1 : 0), tw = 1 << ltw;
        const int prow = h + 10, nitems = (h + 8) << ltw;
        ColSrc ct;
        {
            const int t = l >> 2;
            ct.ty = t >> ltw; ct.col = (t & (tw - 1)) * 4 + (l & 3);
            const P AS1 *src = fenc0 + (long)(puy + ct.ty * 4) * fstride + pux + ct.col;
            uint32_t f[4];
#pragma unroll
            for (int y = 0; y < 4; y++) f[y] = (uint32_t)src[(long)y * fstride];
            ct.f01 = f[0] | (f[1] << 16); ct.f23 = f[2] | (f[3] << 16);
        }
        const int pbase = L::PLANES + grp * 3 * L::PLANE;

        auto round = [&](int step, int dirs, bool satd, bool centre)
        {
            constexpr unsigned long long SQX = nibs({ 0, 0, 0, -1, 1, -1, -1, 1, 1 }), SQY = nibs({ 0, -1, 1, 0, 0, -1, 1, -1, 1 });
            const int k0 = centre ? 1 : 0, K = k0 + dirs;
            const int ya = as_i16(qy - step) >> 2, yb = qy >> 2, yc = as_i16(qy + step) >> 2;
            int ymin = ya < yb ? ya : yb;
            ymin = yc < ymin ? yc : ymin;
            wave_sync();
            /* horizontal pass: the lanes of a PU fill its own planes (the closing re-measure has the centre plane only) */
            for (int p = 0; p < 3; p++)
            {
                if (p != 1 && dirs < 4) continue;
                const int xp = as_i16(qx + (p - 1) * step);
                const P AS1 *src = ref0 + (roff + __mul24(ymin - 3, rstride) + (xp >> 2) - 3);
                const int *tp = (const int*)(x265amd_smem + SM_TAPS) + (xp & 3) * 12;
                int16_t *plane = (int16_t*)(x265amd_smem + pbase + p * L::PLANE);
#pragma unroll
                for (int it = 0; it < (LG == 5 ? 2 : 3); it++)
                {
                    const int item = it * LPP + l;
                    if (item < nitems)
                    {
                        const int r = item >> ltw, q = item & (tw - 1);
                        hps_quad(src + (__mul24(r, rstride) + 4 * q), tp, plane + (4 * q) * prow + r, prow, depth);
                    }
                }
            }
            wave_sync();
            int best = 0x7fffffff, bdir = 0;
            for (int c = 0; c < K; c++)
            {
                const int i = c + 1 - k0, dx = nib(SQX, i), dy = nib(SQY, i);
                const int tx = as_i16(qx + dx * step), ty = as_i16(qy + dy * step);
                int mvc = 0;
                if (active) mvc = ((int)ctab[tx - mvpx] + (int)ctab[ty - mvpy]) & 0xffff;
                /* half-pel rounds: (dx, -1) and (dx, +1) follow each other in the table and are one row apart in the same
                 * plane and phase -- measured together, five vertical sums instead of eight */
                if (step == 2 && i >= 1 && dy == -1 && c + 1 < K && nib(SQY, i + 1) == 1 && nib(SQX, i + 1) == dx)
                {
                    const int ty2 = as_i16(qy + step);
                    int mvc2 = 0;
                    if (active) mvc2 = ((int)ctab[tx - mvpx] + (int)ctab[ty2 - mvpy]) & 0xffff;
                    int o[5];
                    column_rows<5>(pbase + (dx + 1) * L::PLANE, prow, (ty >> 2) - ymin, ty & 3, ct, depth, o);
                    int va = small_rows_cost(o[0], o[1], o[2], o[3], satd, ct, l & 3, depth);
                    int vb = small_rows_cost(o[1], o[2], o[3], o[4], satd, ct, l & 3, depth);
                    va = seg_sum(va, LG); vb = seg_sum(vb, LG);
                    const int ca = (satd ? va >> 1 : va) + mvc, cb = (satd ? vb >> 1 : vb) + mvc2;
                    if (ca < best) { best = ca; bdir = i; }
                    if (cb < best) { best = cb; bdir = i + 1; }
                    c++;
                    continue;
                }
                int o[4];
                column_rows<4>(pbase + (dx + 1) * L::PLANE, prow, (ty >> 2) - ymin, ty & 3, ct, depth, o);
                int v = small_rows_cost(o[0], o[1], o[2], o[3], satd, ct, l & 3, depth);
                v = seg_sum(v, LG);
                const int cost = (satd ? v >> 1 : v) + mvc;         /* every 4x4 abs-sum is even: halving the total = halving per tile */
                if (centre && c == 0) { if (active) bcost = cost; }
                else if (cost < best) { best = cost; bdir = i; }
            }
            if (active && best < bcost) { bcost = best; } else bdir = 0;
            qx = as_i16(qx + nib(SQX, bdir) * step); qy = as_i16(qy + nib(SQY, bdir) * step);
            if (active) nsub += (unsigned)K;
        };

        const int hpel_iters = kWorkload[a.subme][0], hpel_dirs = kWorkload[a.subme][1];
        const int qpel_iters = kWorkload[a.subme][2], qpel_dirs = kWorkload[a.subme][3], hpel_satd = kWorkload[a.subme][4];
        bool centre = hpel_satd != 0;
        for (int it = 0; it < hpel_iters; it++)
        {
            round(2, hpel_dirs, hpel_satd != 0, centre);
            centre = false;
        }
        centre = !hpel_satd;
        if (centre && !qpel_iters) round(1, 0, true, true);
        for (int it = 0; it < qpel_iters; it++)
        {
            round(1, qpel_dirs, true, centre);
            centre = false;
        }
        if (l == 0 && active)
        {
            res[0] = (qx & 0xffff) | (qy << 16);
            res[1] = bcost;
            res[3] = (int)nsub;
        }
    }
}

template<typename P>
__global__ __launch_bounds__(64) void refine_tiny_kernel(SearchArgs a, int chunk)
{
    const int lane = (int)threadIdx.x;
    small_load_taps<P>(lane, sizeof(P) == 1 ? 8 : a.depth);
    /* a wave looks at `chunk` PUs of the job list at once (a.list: entries of the launch's list of PUs the STAR kernels
     * left to others), a lane each, and takes those of each class in turn; chunk: a power of two from 8 to 64, picked by
     * the host so that a short list still spreads over the chip */
    const int n = a.list ? uni(a.redo[0]) : a.njobs;
    for (int base = xcd_remap((int)blockIdx.x, (int)gridDim.x) * chunk; base < n; base += (int)gridDim.x * chunk)
    {
        int j = 0, cls = 0;
        if (lane < chunk && base + lane < n)
        {
            j = a.list ? a.redo[REDO_LIST + base + lane] : base + lane;
            if (j < a.njobs)
            {
                const int d1 = ((const int*)(a.jobs + j))[1], w = d1 & 0xff, h = (d1 >> 8) & 0xff;
                /* a PU whose full-pel cost is 0 is not refined (motion.cpp:1073-1076), a negative cost marks a job that was refused */
                if (!(w == h && (w == a.staged || w == a.staged2)) && !(!a.list && a.mark && a.mark[j]) && a.results[j].cost > 0) cls = tiny_class(w, h);
            }
        }
        if (__ballot(cls != 0) == 0) continue;
        refine_tiny_pass<P, 3>(a, j, cls == 3);
        refine_tiny_pass<P, 4>(a, j, cls == 4);
        refine_tiny_pass<P, 5>(a, j, cls == 5);
    }
}

/* ---------------------------------------------------------------- the common case of a STAR search
 *
 * Most PUs of a picture end their integer search where it began: the start-up candidates (motion.cpp:562-621)
 * pick a full-pel start, none of the 20 points of the first three star rounds (distance 1, 2, 4; StarPatternSearch,
 * :307-547) beats it, three rounds without improvement end the pattern (earlyExitIters = 3, :914) and with
 * bDistance == 0 nothing else runs (:916-1017). For such a PU the order of the points does not matter -- only that
 * none is cheaper -- so the whole stage has a fixed shape: the start-up candidates (the quarter-pel ones as one plane +
 * one vertical pass each, all SAD; the rounded predictor and MV 0 as plain SADs), the decisions of :575-621 on group-uniform values, then the 20
 * points as plain SADs from a window of the reference staged in LDS, four points per pass. A PU one of whose points improves is left to the one-wave search from
 * the start, as are the PUs of other sizes: their result records get SMALL_REDO in sad_calls, and the search kernel
 * launched next over all jobs works on exactly those (its other waves leave at once; no counter or queue that a
 * picture full of such PUs could choke). The others get their record of the full-pel / predictor choice (:1063-1069)
 * written here, ready for refine_small_kernel. */
#define SM_SLOTS 7                                  /* clipped predictor, its full-pel rounding, MV 0, four candidates */
#define SM_PLANES 5                                 /* the slots that can lie between samples: all but the rounding and MV 0 */
template<int S> struct StarLds
{
    typedef SmallGeo<S> G;
    static constexpr int PLANES = SM_FIXED;
    static constexpr int PU_BYTES = SM_PLANES * G::PLANE + 4;       /* the planes, later the window of the star points */
    static_assert((S + 8) * (S + 12) * 2 + 8 <= SM_PLANES * G::PLANE, "the window of the star points (16-bit samples) fits the planes");
    static constexpr int TOTAL = PLANES + G::NPU * PU_BYTES + 32;
};

/* point p (0 .. 19) of the first three rounds of the star pattern in visit order: 4 points at distance 1, 8 at
 * distance 2, 8 at distance 4 (motion.cpp:330-452) */
__device__ __forceinline__ void star_point(int p, int &dx, int &dy)
{
    constexpr unsigned long long X4 = nibs({ 0, -1, 1, 0 }), Y4 = nibs({ -1, 0, 0, 1 });
    constexpr unsigned long long X8 = nibs({ 0, -1, 1, -2, 2, -1, 1, 0 }), Y8 = nibs({ -2, -1, -1, 0, 0, 1, 1, 2 });
    const int r = p < 4 ? 0 : (p < 12 ? 1 : 2), i = p < 4 ? p : (p - 4) & 7;
    const int unit = r == 2 ? 2 : 1;
    dx = nib(r == 0 ? X4 : X8, i) * unit; dy = nib(r == 0 ? Y4 : Y8, i) * unit;
}

/* The points a fixed-shape kernel measures around the start of method M in visit order, and how many: STAR the 20 of
 * star_point; HEX the six of the first hexagon ring (motion.cpp:673-686) and then the eight of the closing square
 * (:710-722), which is what the search measures when the ring does not move the centre; DIA the four of its first
 * round (:628-649). */
template<int M> struct PatPoints { static constexpr int N = M == X265AMD_STAR ? 20 : (M == X265AMD_HEX ? 14 : 4); };
template<int M> __device__ __forceinline__ void pat_point(int p, int &dx, int &dy)
{
    if constexpr (M == X265AMD_STAR) star_point(p, dx, dy);
    else
    {
        constexpr unsigned long long SQX = nibs({ 0, 0, 0, -1, 1, -1, -1, 1, 1 }), SQY = nibs({ 0, -1, 1, 0, 0, -1, 1, -1, 1 });
        constexpr unsigned long long HRX = nibs({ -2, -1, 1, 2, 1, -1 }), HRY = nibs({ 0, 2, 2, 0, -2, -2 });
        if (M == X265AMD_HEX && p < 6) { dx = nib(HRX, p); dy = nib(HRY, p); }
        else
        {
            const int i = (M == X265AMD_HEX ? p - 6 : p) + 1;
            dx = nib(SQX, i & 15); dy = nib(SQY, i & 15);
        }
    }
}

/* How a HEX or DIA search stands after the points above, given the first minimum (cost << 6 | point) of the ring /
 * round (k0) and -- HEX -- of the square (k1): true when the search is over (centre and cost updated to the final
 * full-pel choice, `counted` = points the reference has measured), false when the ring / round moved the centre: the
 * centre is then the new one, `dir` the direction of the move, and the rest is Searcher::resume_pattern's. */
template<int M> __device__ __forceinline__ bool pat_fold(int k0, int k1, int &bmx, int &bmy, int &bcost, int &dir, unsigned &counted)
{
    int dx, dy;
    counted = M == X265AMD_HEX ? 6 : 4;
    dir = 0;
    if ((k0 >> 6) < bcost)
    {
        dir = k0 & 63;
        pat_point<M>(dir, dx, dy);
        bcost = k0 >> 6; bmx = as_i16(bmx + dx); bmy = as_i16(bmy + dy);
        return false;
    }
    if constexpr (M == X265AMD_HEX)
    {
        counted += 8;
        if ((k1 >> 6) < bcost)
        {
            pat_point<M>(k1 & 63, dx, dy);
            bcost = k1 >> 6; bmx = as_i16(bmx + dx); bmy = as_i16(bmy + dy);
        }
    }
    return true;
}

/* What StarPatternSearch has in hand after its first three rounds, given each round's first minimum
 * (cost << 6 | point, 0x7fffffff for a round without points): best MV and cost, bPointNr / bDistance of the winner
 * (COST_MV_PT_DIST, motion.cpp:182-194) and the count of rounds without improvement (:443-446). */
struct StarFold
{
    int ox, oy, bmx, bmy, bcost, pt, dist, rounds;
    __device__ __forceinline__ void start(int x, int y, int c) { ox = bmx = x; oy = bmy = y; bcost = c; pt = 0; dist = 0; rounds = 0; }
    __device__ __forceinline__ void round(int key)
    {
        if ((key >> 6) < bcost)
        {
            constexpr unsigned long long P4 = nibs({ 2, 4, 5, 7 }), P8 = nibs({ 2, 1, 3, 4, 5, 6, 8, 7 });
            constexpr unsigned long long X8 = nibs({ 0, -1, 1, -2, 2, -1, 1, 0 });
            const int p = key & 63, r = p < 4 ? 0 : (p < 12 ? 1 : 2), i = p < 4 ? p : (p - 4) & 7;
            int dx, dy;
            star_point(p, dx, dy);
            bcost = key >> 6; bmx = as_i16(ox + dx); bmy = as_i16(oy + dy);
            pt = r == 0 ? nib(P4, i) : nib(P8, i);
            dist = (r >= 1 && (nib(X8, i) & 1)) ? (1 << r) >> 1 : 1 << r;
            rounds = 0;
        }
        else
            rounds++;
    }
    /* the record star_resume_kernel continues from (Searcher::resume_star) */
    __device__ __forceinline__ void store(int *st, int bprecost, int bpx, int bpy, unsigned nsad, unsigned nsub) const
    {
        st[0] = (ox & 0xffff) | (oy << 16); st[1] = (bmx & 0xffff) | (bmy << 16); st[2] = bcost;
        st[3] = pt | (dist << 8) | (rounds << 16);
        st[4] = bprecost; st[5] = (bpx & 0xffff) | (bpy << 16); st[6] = (int)nsad; st[7] = (int)nsub;
    }
};

/* v of the four rows of sixteen lanes of a wave added up, lane by lane (two lane-swap adds) */
__device__ __forceinline__ int rows4_sum(int v)
{
    auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    v = (int)(r[0] + r[1]);
    r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (int)(r[0] + r[1]);
}

/* S = 8 in a family launch (a.pjobs: the list of 16x16 PUs launched with this list, PU k of it the parent of this
 * list's PUs 4 k .. 4 k + 3): a wave holds the four children of one parent. Where the five jobs name the same
 * predictor, candidates and range, every start-up candidate's SAD for the parent is the sum of the children's, the
 * parent's own sequence of tests runs on those sums, and if the children all start their pattern where the parent does
 * its 20 points are the children's sums too: the parent's whole integer stage costs a few lane swaps here and
 * star_small_kernel<16> skips it (a.handled). */
template<int S, typename P, int M = X265AMD_STAR>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(S == 16 ? OCC_S16 : OCC_S8))) void star_small_kernel(SearchArgs a)
{
    const int depth = sizeof(P) == 1 ? 8 : a.depth;
    typedef SmallGeo<S> G;
    typedef StarLds<S> L;
    const int ngroups = (a.njobs + G::NPU - 1) / G::NPU;
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.redo) a.redo[0] = 0;     /* the list of marked PUs starts empty (redo_list_kernel) */
    const int g = xcd_remap((int)blockIdx.x, ngroups);
    if (g >= ngroups) return;
    const int lane = (int)threadIdx.x, grp = lane >> G::LG, l = lane & (G::LPP - 1);
    int j = g * G::NPU + grp;
    const bool valid = j < a.njobs;
    if (!valid) j = a.njobs - 1;
    const int *job = (const int*)(a.jobs + j);
    int d[10];
#pragma unroll
    for (int i = 0; i < 10; i++) d[i] = job[i];
    const bool sized = (d[1] & 0xffff) == G::SIZE_CODE;
    /* the launch for the list's largest size hands every PU that no fixed-shape launch takes to the one-wave search */
    if (valid && !sized && l == 0 && S == a.staged && (d[1] & 0xffff) != (a.staged2 | (a.staged2 << 8)) &&
        !tiny_takes(a, d[1] & 0xff, (d[1] >> 8) & 0xff) && !rect_takes(a, d[1] & 0xff, (d[1] >> 8) & 0xff))
    {
        a.results[j].sad_calls = SMALL_REDO;
        a.mark[j] = 1;
    }
    const bool act = valid && sized;
    if constexpr (S == 16)
        if (a.handled && valid && a.handled[j]) return;        /* searched with its children (one PU per wave: all lanes agree) */
    if (act && l == 0 && a.done) a.done[j] = 0;        /* "refined with its parent": set by the parent's refinement, if at all */
    if (__ballot(act) == 0) return;
    bool fam = false;
    if constexpr (S == 8)
        if (a.pjobs && g < a.npjobs)
        {
            const int *pjob = (const int*)(a.pjobs + g);
            const int p0 = pjob[0], p1 = pjob[1];
            bool same = act && (p1 & 0xffff) == SmallGeo<16>::SIZE_CODE && ((p1 ^ d[1]) >> 16) == 0 &&
                        as_i16(d[0]) == as_i16(p0) + 8 * (grp & 1) && (d[0] >> 16) == (p0 >> 16) + 8 * (grp >> 1);
#pragma unroll
            for (int i = 2; i < 10; i++) same = same && pjob[i] == d[i];
            fam = __ballot(same) == ~0ull;
        }

    const int pux = as_i16(d[0]), puy = d[0] >> 16, numc = (d[1] >> 16) & 0xff;
    const int mnx = as_i16(d[2]), mny = d[2] >> 16, mxx = as_i16(d[3]), mxy = d[3] >> 16;
    const int mvpx = as_i16(d[4]), mvpy = d[4] >> 16;
    const P AS1 *fenc0, *ref0;
    int fstride, rstride;
    small_planes<P>(a, d[1], fenc0, fstride, ref0, rstride);
    const int roff = puy * rstride + pux;
    const uint16_t AS1 *ctab = (const uint16_t AS1*)a.ctab;
    small_load_taps<P>(lane, depth);

    /* ---- start-up candidates, motion.cpp:575-621 (values every lane of the group holds) ---- */
    const int qmnx = as_i16(mnx << 2), qmny = as_i16(mny << 2), qmxx = as_i16(mxx << 2), qmxy = as_i16(mxy << 2);
#define SM_CLIPX(v) ((v) > qmxx ? qmxx : ((v) < qmnx ? qmnx : (v)))
#define SM_CLIPY(v) ((v) > qmxy ? qmxy : ((v) < qmny ? qmny : (v)))
    const int pmx = SM_CLIPX(mvpx), pmy = SM_CLIPY(mvpy);
    int bmx = as_i16((pmx + 2) >> 2), bmy = as_i16((pmy + 2) >> 2);
    const bool fpOn = ((pmx | pmy) & 3) != 0, zeroOn = (pmx | pmy) != 0;
    int ccx[4], ccy[4];
    bool cOn[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        ccx[i] = SM_CLIPX(as_i16(d[5 + i])); ccy[i] = SM_CLIPY(d[5 + i] >> 16);
        cOn[i] = i < numc && (ccx[i] | ccy[i]) != 0 && !(ccx[i] == pmx && ccy[i] == pmy);
    }
#undef SM_CLIPX
#undef SM_CLIPY
    /* lane l < 7 of a group looks after candidate slot l: where it is measured (quarter-pel) and whether it is in use.
     * A group without a PU of this size measures slot 0 only, so that every address below stays defined. */
    int sx = pmx, sy = pmy;
    bool on = l == 0;
    if (l == 1) { sx = as_i16(bmx << 2); sy = as_i16(bmy << 2); on = fpOn; }
    if (l == 2) { sx = 0; sy = 0; on = zeroOn; }
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (l == 3 + i) { sx = ccx[i]; sy = ccy[i]; on = cOn[i]; }
    on = l < SM_SLOTS && (act ? on : l == 0);
    /* slots 1 and 2 are whole-sample positions and measured as plain SADs; the others get a plane each */
    const int planebits = (int)(grp_bits16<G::LG>(on) & 0x79);
    const int nplanes = __builtin_popcount(planebits);
    int kmax = __builtin_amdgcn_readlane(nplanes, 0);
    if (G::NPU > 1)
    {
        const int n1 = __builtin_amdgcn_readlane(nplanes, 16), n2 = __builtin_amdgcn_readlane(nplanes, 32), n3 = __builtin_amdgcn_readlane(nplanes, 48);
        kmax = n1 > kmax ? n1 : kmax; kmax = n2 > kmax ? n2 : kmax; kmax = n3 > kmax ? n3 : kmax;
    }
    /* lambda * bits of the slot (bitcost.h:46); the clipped predictor itself carries none (motion.cpp:579-582) */
    int mvc = 0;
    if (on && act && l != 0) mvc = ((int)ctab[sx - mvpx] + (int)ctab[sy - mvpy]) & 0xffff;
    /* where the slot's plane begins in the reference, and its horizontal phase */
    const int slot_off = roff + __mul24((sy >> 2) - 3, rstride) + (sx >> 2) - 3, slot_ph = sx & 3;

    /* the lane's tile of the source block, for the plain SADs: a lane per (point, tile) */
    const int tile = l & ((1 << G::LT) - 1), tlx = (tile & (G::TW - 1)) * 4, tly = (tile >> G::LTW) * 4;
    typename Px<P>::quad fq[4];
    {
        const P AS1 *src = fenc0 + (long)(puy + tly) * fstride + pux + tlx;
#pragma unroll
        for (int y = 0; y < 4; y++) fq[y] = Px<P>::load_g(src + (long)y * fstride);
    }
    /* slots 1 and 2: point 0 = the rounded predictor, point 1 = MV 0 */
    int vdir = 0;
    {
        const int pnt = l >> G::LT;
        if (pnt < 2)
        {
            const P AS1 *rp = ref0 + roff + (long)__mul24((pnt ? 0 : bmy) + tly, rstride) + (pnt ? 0 : bmx) + tlx;
#pragma unroll
            for (int yy = 0; yy < 4; yy++) vdir = Px<P>::sad(fq[yy], Px<P>::load_g(rp + (long)yy * rstride), vdir);
        }
        vdir = seg_sum(vdir, G::LT);
    }
    /* horizontal pass: the lanes of a PU fill the planes of its own slots, kmax per PU (a PU with fewer repeats its
     * last); slot values come from the lane that looks after the slot */
    const int pbase = L::PLANES + grp * L::PU_BYTES;
    wave_sync();
    {
        int rem = planebits, sk = 0;
        if constexpr (S == 8)
        {
            for (int k = 0; k < kmax; k++)
            {
                if (rem) { sk = __builtin_ctz(rem); rem &= rem - 1; }
                const int so = grp_lane<G::LG>(slot_off, sk), ph = grp_lane<G::LG>(slot_ph, sk);
                const int kk = k < nplanes ? k : nplanes - 1;
                const int *tp = (const int*)(x265amd_smem + SM_TAPS) + ph * 12;
#pragma unroll
                for (int it = 0; it < 2; it++)
                {
                    const int item = it * 16 + l, r = item >> 1, q = item & 1;
                    hps_quad(ref0 + (so + __mul24(r, rstride) + 4 * q), tp,
                             (int16_t*)(x265amd_smem + pbase + kk * G::PLANE) + (4 * q) * G::PROW + r, G::PROW, depth);
                }
            }
        }
        else
        {
            /* 96 items per plane and 64 lanes: two planes in three passes */
            for (int k = 0; k < kmax; k += 2)
            {
                int so[2], ph[2];
#pragma unroll
                for (int u = 0; u < 2; u++)
                {
                    if (rem) { sk = __builtin_ctz(rem); rem &= rem - 1; }
                    so[u] = grp_lane<G::LG>(slot_off, sk); ph[u] = grp_lane<G::LG>(slot_ph, sk);
                }
#pragma unroll
                for (int it = 0; it < 3; it++)
                {
                    const int item = it * 64 + l, pl = item >= G::IPP ? 1 : 0, rr = item - pl * G::IPP;
                    const int r = rr >> G::LTW, q = rr & (G::TW - 1);
                    const int kk = k + pl < nplanes ? k + pl : nplanes - 1;
                    hps_quad(ref0 + ((pl ? so[1] : so[0]) + __mul24(r, rstride) + 4 * q),
                             (const int*)(x265amd_smem + SM_TAPS) + (pl ? ph[1] : ph[0]) * 12,
                             (int16_t*)(x265amd_smem + pbase + kk * G::PLANE) + (4 * q) * G::PROW + r, G::PROW, depth);
                }
            }
        }
    }
    wave_sync();
    /* vertical pass + SAD: the k-th plane of every PU per pass */
    ColSrc ct;
    small_source_column<S>(fenc0, fstride, pux, puy, l, ct);
    int mycost = 0;
    {
        int rem = planebits, sk = 0;
        for (int k = 0; k < kmax; k++)
        {
            if (rem) { sk = __builtin_ctz(rem); rem &= rem - 1; }
            const int ysk = grp_lane<G::LG>(sy, sk);
            int v = small_column_cost<S>(pbase + (k < nplanes ? k : nplanes - 1) * G::PLANE, 0, ysk & 3, false, ct, l & 3, depth);
            v = seg_sum(v, G::LG);
            if (l == sk) mycost = v + mvc;
        }
    }
    /* the reference's sequence of tests on the measured costs (motion.cpp:579-621) */
    int bprecost = grp_lane<G::LG>(mycost, 0), bcost = bprecost;
    unsigned nsad = 0, nsub = 1;
    {
        const int cFp = grp_lane<G::LG>(vdir, 0) + grp_lane<G::LG>(mvc, 1);
        const int cZero = grp_lane<G::LG>(vdir, 1 << G::LT) + grp_lane<G::LG>(mvc, 2);
        if (fpOn) { bcost = cFp; nsad++; }
        if (zeroOn)
        {
            nsad++;
            if (cZero < bcost) { bcost = cZero; bmx = 0; bmy = 0; }
        }
    }
    int bpx = pmx, bpy = pmy;
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const int c = grp_lane<G::LG>(mycost, 3 + i);
        if (cOn[i] && !(ccx[i] == bpx && ccy[i] == bpy))
        {
            nsub++;
            if (c < bprecost) { bprecost = c; bpx = ccx[i]; bpy = ccy[i]; }
        }
    }
    /* the parent's start-up: the same slots with the children's SADs added up, its own sequence of tests */
    int Pbmx = 0, Pbmy = 0, Pbcost = 0, Pbprecost = 0, Pbpx = pmx, Pbpy = pmy;
    unsigned Pnsad = 0, Pnsub = 1;
    int prkey0 = 0x7fffffff, prkey1 = 0x7fffffff, prkey2 = 0x7fffffff;
    if constexpr (S == 8)
        if (fam)
        {
            const int pcost = rows4_sum(mycost - mvc) + mvc, pdir = rows4_sum(vdir);
            Pbprecost = grp_lane<G::LG>(pcost, 0); Pbcost = Pbprecost;
            Pbmx = as_i16((pmx + 2) >> 2); Pbmy = as_i16((pmy + 2) >> 2);
            const int cFp = grp_lane<G::LG>(pdir, 0) + grp_lane<G::LG>(mvc, 1);
            const int cZero = grp_lane<G::LG>(pdir, 1 << G::LT) + grp_lane<G::LG>(mvc, 2);
            if (fpOn) { Pbcost = cFp; Pnsad++; }
            if (zeroOn)
            {
                Pnsad++;
                if (cZero < Pbcost) { Pbcost = cZero; Pbmx = 0; Pbmy = 0; }
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                const int c = grp_lane<G::LG>(pcost, 3 + i);
                if (cOn[i] && !(ccx[i] == Pbpx && ccy[i] == Pbpy))
                {
                    Pnsub++;
                    if (c < Pbprecost) { Pbprecost = c; Pbpx = ccx[i]; Pbpy = ccy[i]; }
                }
            }
            /* the parent's points are the children's only where all five patterns have one centre */
            fam = __ballot(bmx == Pbmx && bmy == Pbmy) == ~0ull;
        }
    /* ---- the first three star rounds around (bmx, bmy): 4 + 8 + 8 points (motion.cpp:330-452), does any beat bcost?
     * A lane per (point, tile), four points per pass ---- */
    int rkey0 = 0x7fffffff, rkey1 = 0x7fffffff, rkey2 = 0x7fffffff;     /* per round: min of cost << 6 | point in visit order */
    unsigned npts = 0;
#ifndef X265AMD_STAR_GLOBAL_LOADS
    /* the (S + 8)^2 samples the 20 points can touch, staged once per PU into the LDS of the start-up planes (dead by
     * now); the points are measured from there. (X265AMD_STAR_GLOBAL_LOADS, tools builds only: the points read global
     * memory instead, the variant this one was measured against.) */
    constexpr int BPS = (int)sizeof(P);
    constexpr int WP = (S + 12) * BPS;              /* window row pitch in bytes: S + 8 samples and what a lane reads past them */
    const int woff = L::PLANES + grp * L::PU_BYTES;
    wave_sync();
    {
        typedef uint32_t __attribute__((aligned(1))) u32u;      /* the window starts at any sample */
        const char AS1 *w0 = (const char AS1*)(ref0 + roff + (long)__mul24(bmy - 4, rstride) + bmx - 4);
        for (int i = l; i < (S + 8) * (WP / 4); i += G::LPP)
        {
            const int r = i / (WP / 4), q = i - r * (WP / 4);
            *(uint32_t*)(x265amd_smem + woff + r * WP + 4 * q) = *(const u32u AS1*)(w0 + (long)r * rstride * BPS + 4 * q);
        }
    }
    wave_sync();
#endif
#pragma unroll 1
    for (int pass = 0; pass < (PatPoints<M>::N + 3) / 4; pass++)
    {
        const int p = pass * 4 + (l >> G::LT);
        int dx, dy;
        pat_point<M>(p < PatPoints<M>::N ? p : 0, dx, dy);
        const int x = as_i16(bmx + dx), y = as_i16(bmy + dy);
        /* STAR: the reference tests only the bounds a point can cross; the hexagon, square and diamond points are
         * measured unchecked */
        const bool ok = act && p < PatPoints<M>::N &&
                        (M != X265AMD_STAR || ((dy >= 0 || y >= mny) && (dy <= 0 || y <= mxy) && (dx >= 0 || x >= mnx) && (dx <= 0 || x <= mxx)));
        int v = 0, cost = 0;
        if (ok)
        {
#ifndef X265AMD_STAR_GLOBAL_LOADS
            const int bo = woff + (4 + dy + tly) * WP + (4 + dx + tlx) * BPS;
#pragma unroll
            for (int yy = 0; yy < 4; yy++) v = Px<P>::sad(fq[yy], Px<P>::load_lds(bo + yy * WP), v);
#else
            const P AS1 *rp = ref0 + roff + (long)__mul24(y + tly, rstride) + x + tlx;
#pragma unroll
            for (int yy = 0; yy < 4; yy++) v = Px<P>::sad(fq[yy], Px<P>::load_g(rp + (long)yy * rstride), v);
#endif
            cost = ((int)ctab[as_i16(x << 2) - mvpx] + (int)ctab[as_i16(y << 2) - mvpy]) & 0xffff;
        }
        v = seg_sum(v, G::LT);      /* the tiles of a point are adjacent lanes */
        const int key = ok && tile == 0 ? ((v + cost) << 6) | p : 0x7fffffff;
        /* the parent's key of the point: same centre, range and predictor, so the same points pass the bounds tests and
         * carry the same lambda * bits */
        int pkey = 0x7fffffff;
        if constexpr (S == 8)
            if (fam)
            {
                const int pv = rows4_sum(v);
                pkey = ok && tile == 0 ? ((pv + cost) << 6) | p : 0x7fffffff;
            }
        if constexpr (M == X265AMD_HEX)
        {
            /* rkey0: the ring (points 0 .. 5), rkey1: the square */
            const int kr = grp_min<G::LG>(p < 6 ? key : 0x7fffffff), ks = grp_min<G::LG>(p >= 6 ? key : 0x7fffffff);
            rkey0 = kr < rkey0 ? kr : rkey0; rkey1 = ks < rkey1 ? ks : rkey1;
            if (S == 8 && fam)
            {
                const int pr = grp_min<G::LG>(p < 6 ? pkey : 0x7fffffff), ps = grp_min<G::LG>(p >= 6 ? pkey : 0x7fffffff);
                prkey0 = pr < prkey0 ? pr : prkey0; prkey1 = ps < prkey1 ? ps : prkey1;
            }
        }
        else
        {
            const int km = grp_min<G::LG>(key);
            if (pass == 0) rkey0 = km;
            else if (pass < 3) rkey1 = km < rkey1 ? km : rkey1;
            else rkey2 = km < rkey2 ? km : rkey2;
            if (S == 8 && fam)
            {
                const int pkm = grp_min<G::LG>(pkey);
                if (pass == 0) prkey0 = pkm;
                else if (pass < 3) prkey1 = pkm < prkey1 ? pkm : prkey1;
                else prkey2 = pkm < prkey2 ? pkm : prkey2;
            }
        }
        npts += (unsigned)grp_count<G::LG>(ok && tile == 0);
    }
    if constexpr (S == 8)
        if (a.pjobs && lane == 0 && g < a.npjobs)
        {
            a.handled[g] = fam;
            if (fam)
            {
                StarFold F;
                F.start(Pbmx, Pbmy, Pbcost);
                bool over;
                if constexpr (M == X265AMD_STAR)
                {
                    F.round(prkey0); F.round(prkey1); F.round(prkey2);
                    Pnsad += npts;
                    over = F.rounds >= 3;
                }
                else
                {
                    int dir;
                    unsigned counted;
                    over = pat_fold<M>(prkey0, prkey1, Pbmx, Pbmy, Pbcost, dir, counted);
                    Pnsad += counted;
                    F.start(Pbmx, Pbmy, Pbcost);
                    F.pt = dir;
                }
                int *pres = (int*)(a.presults + g);
                a.pmark[g] = !over;
                if (!over)
                {
                    F.store(a.pstate + 8 * (long)g, Pbprecost, Pbpx, Pbpy, Pnsad, Pnsub);
                    pres[2] = (int)SMALL_RESUME;
                }
                else
                {
                    int qx, qy, c = Pbcost;
                    if (Pbprecost < c) { qx = Pbpx; qy = Pbpy; c = Pbprecost; }
                    else { qx = as_i16(Pbmx << 2); qy = as_i16(Pbmy << 2); }
                    pres[0] = (qx & 0xffff) | (qy << 16);
                    pres[1] = c;
                    pres[2] = (int)Pnsad;
                    pres[3] = (int)Pnsub;
                }
            }
        }
    if (!act || l != 0) return;
    if constexpr (M != X265AMD_STAR)
    {
        int dir;
        unsigned counted;
        const bool over = pat_fold<M>(rkey0, rkey1, bmx, bmy, bcost, dir, counted);
        nsad += counted;
        int *res = (int*)(a.results + j);
        a.mark[j] = !over;
        if (!over)
        {
            StarFold F;
            F.start(bmx, bmy, bcost);
            F.pt = dir;
            F.store(a.state + 8 * (long)j, bprecost, bpx, bpy, nsad, nsub);
            res[2] = (int)SMALL_RESUME;
            return;
        }
        int qx, qy;
        if (bprecost < bcost) { qx = bpx; qy = bpy; bcost = bprecost; }
        else { qx = as_i16(bmx << 2); qy = as_i16(bmy << 2); }
        res[0] = (qx & 0xffff) | (qy << 16);
        res[1] = bcost;
        res[2] = (int)nsad;
        res[3] = (int)nsub;
        return;
    }
    /* the bookkeeping of StarPatternSearch over these rounds (motion.cpp:323-452): strict '<' in visit order, a
     * counter of rounds without improvement */
    StarFold F;
    F.start(bmx, bmy, bcost);
    F.round(rkey0); F.round(rkey1); F.round(rkey2);
    nsad += npts;
    int *res = (int*)(a.results + j);
    a.mark[j] = F.rounds < 3;
    if (F.rounds < 3)
    {
        /* a point improved: the pattern goes on to larger distances, which star_resume_kernel measures */
        F.store(a.state + 8 * (long)j, bprecost, bpx, bpy, nsad, nsub);
        res[2] = (int)SMALL_RESUME;
        return;
    }
    /* three rounds without improvement end the pattern with the start point standing; motion.cpp:1063-1069 */
    int qx, qy;
    if (bprecost < bcost) { qx = bpx; qy = bpy; bcost = bprecost; }
    else { qx = as_i16(bmx << 2); qy = as_i16(bmy << 2); }
    res[0] = (qx & 0xffff) | (qy << 16);
    res[1] = bcost;
    res[2] = (int)nsad;
    res[3] = (int)nsub;
}

/* ---------------------------------------------------------------- the same stage for the small other partitions
 *
 * star_small_kernel's stage for the shapes of tiny_class, 2, 4 or 8 PUs per wave: start-up candidates, the points of the
 * method's first rounds from a window in LDS, the record or the hand-over state. A wave looks at 64 PUs of the job list
 * and takes those of each class in turn (as refine_tiny_kernel does); width and height of a PU are values its lanes
 * hold, only the lanes per PU are fixed per class. */
template<int LG, int BPS> struct StarTinyLds
{
    static constexpr int LPP = 1 << LG, NPU = 64 / LPP;
    static constexpr int PLANE = LG == 3 ? 224 : (LG == 4 ? 448 : 576);        /* the largest plane of the class: (h + 10) w int16 */
    static constexpr int WINDOW = (LG == 3 ? 256 : (LG == 4 ? 384 : 480)) * BPS; /* its largest window: (h + 8) (w + 12) samples */
    /* one plane at a time (filled, then measured), later the window of the points */
    static constexpr int PU_BYTES = (PLANE > WINDOW ? PLANE : WINDOW) + 4;
    static constexpr int TOTAL = SM_FIXED + NPU * PU_BYTES + 32;
};
template<int BPS> struct StarTinyMax
{
    static constexpr int A = StarTinyLds<3, BPS>::TOTAL, B = StarTinyLds<4, BPS>::TOTAL, C = StarTinyLds<5, BPS>::TOTAL;
    static constexpr int TOTAL = A > B ? (A > C ? A : C) : (B > C ? B : C);     /* one LDS size for the three classes */
};

template<typename P, int M, int LG>
__device__ __forceinline__ void star_tiny_pass(const SearchArgs &a, int jmine, bool mine)
{
    constexpr int BPS = (int)sizeof(P);
    typedef StarTinyLds<LG, BPS> L;
    constexpr int LPP = L::LPP, NPU = L::NPU, LT = LG - 2;
    const int depth = sizeof(P) == 1 ? 8 : a.depth;
    const int lane = (int)threadIdx.x, grp = lane >> LG, l = lane & (LPP - 1);
    unsigned long long m = __ballot(mine);
    while (m)
    {
        /* group g takes the g-th PU that is left; a group without one goes along with the first (and writes nothing) */
        int src_lane = (int)__builtin_ctzll(m);
        bool act = false;
#pragma unroll
        for (int g = 0; g < NPU; g++)
            if (m)
            {
                if (g == grp) { src_lane = (int)__builtin_ctzll(m); act = true; }
                m &= m - 1;
            }
        const int j = __shfl(jmine, src_lane);
        wave_sync();
        const int *job = (const int*)(a.jobs + j);
        int d[10];
#pragma unroll
        for (int i = 0; i < 10; i++) d[i] = job[i];
        const int pux = as_i16(d[0]), puy = d[0] >> 16, w = d[1] & 0xff, h = (d[1] >> 8) & 0xff, numc = (d[1] >> 16) & 0xff;
        const int ltw = w == 16 ? 2 : (w == 8 ? 1 : 0), tw = 1 << ltw, prow = h + 10, nitems = (h + 8) << ltw;
        const int mnx = as_i16(d[2]), mny = d[2] >> 16, mxx = as_i16(d[3]), mxy = d[3] >> 16;
        const int mvpx = as_i16(d[4]), mvpy = d[4] >> 16;
        const P AS1 *fenc0, *ref0;
        int fstride, rstride;
        small_planes<P>(a, d[1], fenc0, fstride, ref0, rstride);
        const int roff = puy * rstride + pux;
        const uint16_t AS1 *ctab = (const uint16_t AS1*)a.ctab;

        /* ---- start-up candidates, motion.cpp:575-621 (values every lane of the group holds) ---- */
        const int qmnx = as_i16(mnx << 2), qmny = as_i16(mny << 2), qmxx = as_i16(mxx << 2), qmxy = as_i16(mxy << 2);
#define SM_CLIPX(v) ((v) > qmxx ? qmxx : ((v) < qmnx ? qmnx : (v)))
#define SM_CLIPY(v) ((v) > qmxy ? qmxy : ((v) < qmny ? qmny : (v)))
        const int pmx = SM_CLIPX(mvpx), pmy = SM_CLIPY(mvpy);
        int bmx = as_i16((pmx + 2) >> 2), bmy = as_i16((pmy + 2) >> 2);
        const bool fpOn = ((pmx | pmy) & 3) != 0, zeroOn = (pmx | pmy) != 0;
        int ccx[4], ccy[4];
        bool cOn[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            ccx[i] = SM_CLIPX(as_i16(d[5 + i])); ccy[i] = SM_CLIPY(d[5 + i] >> 16);
            cOn[i] = i < numc && (ccx[i] | ccy[i]) != 0 && !(ccx[i] == pmx && ccy[i] == pmy);
        }
#undef SM_CLIPX
#undef SM_CLIPY
        /* lane l < 7 of a group looks after candidate slot l: where it is measured (quarter-pel) and whether it is in use */
        int sx = pmx, sy = pmy;
        bool on = l == 0;
        if (l == 1) { sx = as_i16(bmx << 2); sy = as_i16(bmy << 2); on = fpOn; }
        if (l == 2) { sx = 0; sy = 0; on = zeroOn; }
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (l == 3 + i) { sx = ccx[i]; sy = ccy[i]; on = cOn[i]; }
        on = l < SM_SLOTS && on;
        /* slots 1 and 2 are whole-sample positions and measured as plain SADs; the others get a plane each */
        const int planebits = (int)((__ballot(on) >> (lane & ~(LPP - 1))) & 0x79);
        const int nplanes = __builtin_popcount(planebits);
        int kmax = 1;
        while (__ballot(nplanes > kmax)) kmax++;
        int mvc = 0;
        if (on && l != 0) mvc = ((int)ctab[sx - mvpx] + (int)ctab[sy - mvpy]) & 0xffff;
        const int slot_off = roff + __mul24((sy >> 2) - 3, rstride) + (sx >> 2) - 3, slot_ph = sx & 3;

        /* the lane's tile of the source block, for the plain SADs: a lane per (point, tile) */
        const int tile = l & ((1 << LT) - 1), tlx = (tile & (tw - 1)) * 4, tly = (tile >> ltw) * 4;
        typename Px<P>::quad fq[4];
        {
            const P AS1 *src = fenc0 + (long)(puy + tly) * fstride + pux + tlx;
#pragma unroll
            for (int y = 0; y < 4; y++) fq[y] = Px<P>::load_g(src + (long)y * fstride);
        }
        /* slots 1 and 2: point 0 = the rounded predictor, point 1 = MV 0 */
        int vdir = 0;
        {
            const int pnt = l >> LT;
            if (pnt < 2)
            {
                const P AS1 *rp = ref0 + roff + (long)__mul24((pnt ? 0 : bmy) + tly, rstride) + (pnt ? 0 : bmx) + tlx;
#pragma unroll
                for (int yy = 0; yy < 4; yy++) vdir = Px<P>::sad(fq[yy], Px<P>::load_g(rp + (long)yy * rstride), vdir);
            }
            vdir = seg_sum(vdir, LT);
        }
        const int pbase = SM_FIXED + grp * L::PU_BYTES;
        ColSrc ct;
        {
            const int t = l >> 2;
            ct.ty = t >> ltw; ct.col = (t & (tw - 1)) * 4 + (l & 3);
            const P AS1 *src = fenc0 + (long)(puy + ct.ty * 4) * fstride + pux + ct.col;
            uint32_t f[4];
#pragma unroll
            for (int y = 0; y < 4; y++) f[y] = (uint32_t)src[(long)y * fstride];
            ct.f01 = f[0] | (f[1] << 16); ct.f23 = f[2] | (f[3] << 16);
        }
        /* the k-th slot of every PU per pass (a PU with fewer repeats its last): horizontal pass by the lanes of the PU
         * into its plane, then vertical pass + SAD, a lane per (tile, column) */
        int mycost = 0;
        {
            int rem = planebits, sk = 0;
            for (int k = 0; k < kmax; k++)
            {
                if (rem) { sk = __builtin_ctz(rem); rem &= rem - 1; }
                const int so = grp_lane<LG>(slot_off, sk), ph = grp_lane<LG>(slot_ph, sk), ysk = grp_lane<LG>(sy, sk);
                const int *tp = (const int*)(x265amd_smem + SM_TAPS) + ph * 12;
                int16_t *plane = (int16_t*)(x265amd_smem + pbase);
                wave_sync();                            /* the plane before has been read */
#pragma unroll
                for (int it = 0; it < (LG == 5 ? 2 : 3); it++)
                {
                    const int item = it * LPP + l;
                    if (item < nitems)
                    {
                        const int r = item >> ltw, q = item & (tw - 1);
                        hps_quad(ref0 + (so + __mul24(r, rstride) + 4 * q), tp, plane + (4 * q) * prow + r, prow, depth);
                    }
                }
                wave_sync();
                int o[4];
                column_rows<4>(pbase, prow, 0, ysk & 3, ct, depth, o);
                int v = small_rows_cost(o[0], o[1], o[2], o[3], false, ct, l & 3, depth);
                v = seg_sum(v, LG);
                if (l == sk) mycost = v + mvc;
            }
        }
        /* the reference's sequence of tests on the measured costs (motion.cpp:579-621) */
        int bprecost = grp_lane<LG>(mycost, 0), bcost = bprecost;
        unsigned nsad = 0, nsub = 1;
        {
            const int cFp = grp_lane<LG>(vdir, 0) + grp_lane<LG>(mvc, 1);
            const int cZero = grp_lane<LG>(vdir, 1 << LT) + grp_lane<LG>(mvc, 2);
            if (fpOn) { bcost = cFp; nsad++; }
            if (zeroOn)
            {
                nsad++;
                if (cZero < bcost) { bcost = cZero; bmx = 0; bmy = 0; }
            }
        }
        int bpx = pmx, bpy = pmy;
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const int c = grp_lane<LG>(mycost, 3 + i);
            if (cOn[i] && !(ccx[i] == bpx && ccy[i] == bpy))
            {
                nsub++;
                if (c < bprecost) { bprecost = c; bpx = ccx[i]; bpy = ccy[i]; }
            }
        }
        /* ---- the points of the method's first rounds around (bmx, bmy), from the (w + 8) x (h + 8) samples they can
         * touch, staged into the LDS of the start-up plane (dead by now). A lane per (point, tile), four points per pass ---- */
        int rkey0 = 0x7fffffff, rkey1 = 0x7fffffff, rkey2 = 0x7fffffff;
        unsigned npts = 0;
        const int WQ = ((w + 12) * BPS) >> 2, WP = WQ * 4, nwin = (h + 8) * WQ;     /* window row pitch in dwords and bytes */
        int inv = w == 4 ? 16384 : (w == 8 ? 13108 : 9363);                          /* i / WQ = i * inv >> 16 for i < nwin */
        if (BPS == 2) inv = (inv + 1) >> 1;
        wave_sync();
        {
            typedef uint32_t __attribute__((aligned(1))) u32u;      /* the window starts at any sample */
            const char AS1 *w0 = (const char AS1*)(ref0 + roff + (long)__mul24(bmy - 4, rstride) + bmx - 4);
            for (int i = l; __ballot(i < nwin) != 0; i += LPP)
                if (i < nwin)
                {
                    const int r = (i * inv) >> 16, q = i - r * WQ;
                    *(uint32_t*)(x265amd_smem + pbase + r * WP + 4 * q) = *(const u32u AS1*)(w0 + (long)r * rstride * BPS + 4 * q);
                }
        }
        wave_sync();
#pragma unroll 1
        for (int pass = 0; pass < (PatPoints<M>::N + 3) / 4; pass++)
        {
            const int p = pass * 4 + (l >> LT);
            int dx, dy;
            pat_point<M>(p < PatPoints<M>::N ? p : 0, dx, dy);
            const int x = as_i16(bmx + dx), y = as_i16(bmy + dy);
            /* STAR: the reference tests only the bounds a point can cross; the hexagon, square and diamond points are
             * measured unchecked */
            const bool ok = p < PatPoints<M>::N &&
                            (M != X265AMD_STAR || ((dy >= 0 || y >= mny) && (dy <= 0 || y <= mxy) && (dx >= 0 || x >= mnx) && (dx <= 0 || x <= mxx)));
            int v = 0, cost = 0;
            if (ok)
            {
                const int bo = pbase + (4 + dy + tly) * WP + (4 + dx + tlx) * BPS;
#pragma unroll
                for (int yy = 0; yy < 4; yy++) v = Px<P>::sad(fq[yy], Px<P>::load_lds(bo + yy * WP), v);
                cost = ((int)ctab[as_i16(x << 2) - mvpx] + (int)ctab[as_i16(y << 2) - mvpy]) & 0xffff;
            }
            v = seg_sum(v, LT);         /* the tiles of a point are adjacent lanes */
            const int key = ok && tile == 0 ? ((v + cost) << 6) | p : 0x7fffffff;
            if constexpr (M == X265AMD_HEX)
            {
                /* rkey0: the ring (points 0 .. 5), rkey1: the square */
                const int kr = grp_min<LG>(p < 6 ? key : 0x7fffffff), ks = grp_min<LG>(p >= 6 ? key : 0x7fffffff);
                rkey0 = kr < rkey0 ? kr : rkey0; rkey1 = ks < rkey1 ? ks : rkey1;
            }
            else
            {
                const int km = grp_min<LG>(key);
                if (pass == 0) rkey0 = km;
                else if (pass < 3) rkey1 = km < rkey1 ? km : rkey1;
                else rkey2 = km < rkey2 ? km : rkey2;
            }
            npts += (unsigned)grp_count_any<LG>(ok && tile == 0);
        }
        if (act && l == 0)
        {
            int *res = (int*)(a.results + j);
            bool over;
            StarFold F;
            if constexpr (M != X265AMD_STAR)
            {
                int dir;
                unsigned counted;
                over = pat_fold<M>(rkey0, rkey1, bmx, bmy, bcost, dir, counted);
                nsad += counted;
                F.start(bmx, bmy, bcost);
                F.pt = dir;
            }
            else
            {
                /* the bookkeeping of StarPatternSearch over these rounds (motion.cpp:323-452) */
                F.start(bmx, bmy, bcost);
                F.round(rkey0); F.round(rkey1); F.round(rkey2);
                nsad += npts;
                over = F.rounds >= 3;
            }
            a.mark[j] = !over;
            if (!over)
            {
                /* the search goes on: star_resume_kernel's */
                F.store(a.state + 8 * (long)j, bprecost, bpx, bpy, nsad, nsub);
                res[2] = (int)SMALL_RESUME;
            }
            else
            {
                /* motion.cpp:1063-1069 */
                int qx, qy;
                if (bprecost < bcost) { qx = bpx; qy = bpy; bcost = bprecost; }
                else { qx = as_i16(bmx << 2); qy = as_i16(bmy << 2); }
                res[0] = (qx & 0xffff) | (qy << 16);
                res[1] = bcost;
                res[2] = (int)nsad;
                res[3] = (int)nsub;
            }
        }
    }
}

#ifndef OCC_ST
#define OCC_ST 4
#endif
template<int D, typename P, int M>
__device__ __forceinline__ void star_rect_shape(const SearchArgs &a, int j, const int *job, int code);   /* (below) */
template<typename P, int M>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(OCC_ST))) void star_tiny_kernel(SearchArgs a, int chunk)
{
    const int lane = (int)threadIdx.x;
    small_load_taps<P>(lane, sizeof(P) == 1 ? 8 : a.depth);
    for (int base = xcd_remap((int)blockIdx.x, (int)gridDim.x) * chunk; base < a.njobs; base += (int)gridDim.x * chunk)
    {
        const int j = base + lane;
        int cls = 0;
        if (lane < chunk && j < a.njobs)
        {
            const int d1 = ((const int*)(a.jobs + j))[1], w = d1 & 0xff, h = (d1 >> 8) & 0xff;
            if (tiny_takes(a, w, h)) cls = tiny_class(w, h);
            /* 16x12, 12x16 and 16x16, a wave per PU: star_rect_kernel's shapes whose list this kernel screens anyway */
            if (rect_takes(a, w, h) && (w > h ? w : h) == 16) cls = 6;
        }
        if (__ballot(cls != 0) == 0) continue;
        star_tiny_pass<P, M, 3>(a, j, cls == 3);
        star_tiny_pass<P, M, 4>(a, j, cls == 4);
        star_tiny_pass<P, M, 5>(a, j, cls == 5);
        unsigned long long m = __ballot(cls == 6);
        while (m)
        {
            const int jj = base + (int)__builtin_ctzll(m);
            m &= m - 1;
            const int *job = (const int*)(a.jobs + jj);
            wave_sync();
            star_rect_shape<16, P, M>(a, jj, job, uni(job[1]) & 0xffff);
        }
    }
}

/* four samples at byte offset base + coff of the LDS array, base a multiple of four and coff a compile-time constant
 * once the caller's loops are unrolled: the realignment disappears where coff is a multiple of four itself */
template<typename P> __device__ __forceinline__ typename Px<P>::quad lds_quad_const(int base, int coff);
template<> __device__ __forceinline__ uint32_t lds_quad_const<uint8_t>(int base, int coff)
{
    const uint32_t *q = (const uint32_t*)(x265amd_smem + base + (coff & ~3));
    return (coff & 3) ? __builtin_amdgcn_alignbyte(q[1], q[0], (uint32_t)(coff & 3)) : q[0];
}
template<> __device__ __forceinline__ uint2 lds_quad_const<uint16_t>(int base, int coff)
{
    const uint32_t *q = (const uint32_t*)(x265amd_smem + base + (coff & ~3));
    if (!(coff & 2)) return make_uint2(q[0], q[1]);
    return make_uint2(__builtin_amdgcn_alignbit(q[1], q[0], 16), __builtin_amdgcn_alignbit(q[2], q[1], 16));
}

/* sum over the wave as a scalar: four steps inside the rows of sixteen lanes, then the rows' sums handed up by the
 * row broadcasts (lane 15 of a row into the next row, lane 31 into the upper half); the total arrives in lane 63 */
__device__ __forceinline__ int wave_total(int v)
{
    v += dpp_all<0xB1>(v);
    v += dpp_all<0x4E>(v);
    v += dpp_all<0x141>(v);
    v += dpp_all<0x140>(v);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      /* row_bcast:15 into rows 1 and 3 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      /* row_bcast:31 into rows 2 and 3 */
    return __builtin_amdgcn_readlane(v, 63);
}

/* ---------------------------------------------------------------- the same for 32x32 and 64x64 PUs
 *
 * A workgroup of one / two waves per PU (StarBigGeo), everything about the PU uniform across it. The start-up candidates go
 * one after the other: a whole-sample one is a plain SAD of the reference block, every thread taking the quad rows of its
 * tile; a fractional one fills a plane of luma_hps intermediates (all threads), then every thread runs the vertical
 * pass + SAD of its (tile, column) pairs -- two plane buffers in turn, so a fill never waits for the pass before it.
 * The sums meet in LDS cells. The 20 points of the first three rounds are SADs from the (S + 8)^2 window staged over
 * the dead planes, every thread a strip four samples wide, the sums of sixteen lanes at a time added to the point's cell. The
 * fold over the rounds and the hand-over to star_resume_kernel are star_small_kernel's. */
/* Small workgroups on purpose: next to launches of one-wave workgroups a workgroup of four waves waits long for a
 * compute unit with four free places at once (measured: the launch of the 32x32 PUs took 2.3 ms in the step against
 * 0.4 ms alone). */
template<int S> struct StarBigGeo
{
    static constexpr int WAVES = S == 32 ? 1 : 2;
    static constexpr int THREADS = WAVES * 64;
    static constexpr int ITEMS = SmallGeo<S>::LPP / THREADS;        /* (tile, column) pairs per thread */
    static constexpr int NQ = S * S / (4 * THREADS);                /* quad rows per thread: a strip four samples wide */
};

template<int S> struct StarBigLds
{
    typedef SmallGeo<S> G;
    static constexpr int CELLS = SM_FIXED;                          /* per wave: 8 cells of the start-up slots, then 20 of the star points */
    static constexpr int PLANES = CELLS + 128 * StarBigGeo<S>::WAVES;
    static_assert((S + 8) * (S + 12) * 2 + 8 <= 2 * G::PLANE, "the window of the star points (16-bit samples) fits the two planes");
    static constexpr int TOTAL = PLANES + 2 * G::PLANE + 32;
};

/* offsets of the pattern points as compile-time tables (the big kernel unrolls over them) */
template<int M> struct PatTable
{
    int dx[20], dy[20];
    constexpr PatTable() : dx(), dy()
    {
        const int sx[20] = { 0, -1, 1, 0, 0, -1, 1, -2, 2, -1, 1, 0, 0, -2, 2, -4, 4, -2, 2, 0 };
        const int sy[20] = { -1, 0, 0, 1, -2, -1, -1, 0, 0, 1, 1, 2, -4, -2, -2, 0, 0, 2, 2, 4 };
        const int hx[14] = { -2, -1, 1, 2, 1, -1, 0, 0, -1, 1, -1, -1, 1, 1 };
        const int hy[14] = { 0, 2, 2, 0, -2, -2, -1, 1, 0, 0, -1, 1, -1, 1 };
        for (int p = 0; p < 20; p++)
        {
            if (M == X265AMD_STAR) { dx[p] = sx[p]; dy[p] = sy[p]; }
            else if (M == X265AMD_HEX) { dx[p] = p < 14 ? hx[p] : 0; dy[p] = p < 14 ? hy[p] : 0; }
            else { dx[p] = p < 4 ? hx[6 + p] : 0; dy[p] = p < 4 ? hy[6 + p] : 0; }
        }
    }
};

template<int S, typename P, int M = X265AMD_STAR>
__global__ __launch_bounds__(StarBigGeo<S>::THREADS) __attribute__((amdgpu_waves_per_eu(S == 64 ? OCC_SB64 : OCC_SB32))) void star_big_kernel(SearchArgs a)
{
    const int depth = sizeof(P) == 1 ? 8 : a.depth;
    typedef SmallGeo<S> G;
    typedef StarBigLds<S> L;
    constexpr int THREADS = StarBigGeo<S>::THREADS, ITEMS = StarBigGeo<S>::ITEMS, NQ = StarBigGeo<S>::NQ, BPS = (int)sizeof(P);
    if (blockIdx.x == 0 && threadIdx.x == 0 && S == a.staged) a.redo[0] = 0;   /* the list of marked PUs starts empty (redo_list_kernel) */
    const int j = xcd_remap((int)blockIdx.x, a.njobs);
    if (j >= a.njobs) return;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    if (tid == 0 && a.done && S == a.staged) a.done[j] = 0;     /* "refined with its parent": set by the parent's refinement, if at all */
    const int *job = (const int*)(a.jobs + j);
    int d[10];
    d[1] = uni(job[1]);
    if ((d[1] & 0xffff) != G::SIZE_CODE)
    {
        /* the launch for the list's largest size hands every PU that no fixed-shape launch takes to the one-wave search */
        if (S == a.staged && (d[1] & 0xffff) != (a.staged2 | (a.staged2 << 8)) && tid == 0 &&
            !tiny_takes(a, d[1] & 0xff, (d[1] >> 8) & 0xff) && !rect_takes(a, d[1] & 0xff, (d[1] >> 8) & 0xff))
        { a.results[j].sad_calls = SMALL_REDO; a.mark[j] = 1; }
        return;
    }
#pragma unroll
    for (int i = 0; i < 10; i++) d[i] = uni(job[i]);
    const int pux = as_i16(d[0]), puy = d[0] >> 16, numc = (d[1] >> 16) & 0xff;
    const int mnx = as_i16(d[2]), mny = d[2] >> 16, mxx = as_i16(d[3]), mxy = d[3] >> 16;
    const int mvpx = as_i16(d[4]), mvpy = d[4] >> 16;
    const P AS1 *fenc0, *ref0;
    int fstride, rstride;
    small_planes<P>(a, d[1], fenc0, fstride, ref0, rstride);
    const int roff = puy * rstride + pux;
    const uint16_t AS1 *ctab = (const uint16_t AS1*)a.ctab;
    small_load_taps<P>(tid, depth);
    constexpr int WAVES = StarBigGeo<S>::WAVES;
    int *cells = (int*)(x265amd_smem + L::CELLS);                   /* [wave][32]: every wave's sums; a slot that is off stays unread */
    const int wave = tid >> 6;

    /* ---- start-up candidates, motion.cpp:575-621 ---- */
    const int qmnx = as_i16(mnx << 2), qmny = as_i16(mny << 2), qmxx = as_i16(mxx << 2), qmxy = as_i16(mxy << 2);
#define SM_CLIPX(v) ((v) > qmxx ? qmxx : ((v) < qmnx ? qmnx : (v)))
#define SM_CLIPY(v) ((v) > qmxy ? qmxy : ((v) < qmny ? qmny : (v)))
    const int pmx = SM_CLIPX(mvpx), pmy = SM_CLIPY(mvpy);
    int bmx = as_i16((pmx + 2) >> 2), bmy = as_i16((pmy + 2) >> 2);
    const bool fpOn = ((pmx | pmy) & 3) != 0, zeroOn = (pmx | pmy) != 0;
    int ccx[4], ccy[4];
    bool cOn[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        ccx[i] = SM_CLIPX(as_i16(d[5 + i])); ccy[i] = SM_CLIPY(d[5 + i] >> 16);
        cOn[i] = i < numc && (ccx[i] | ccy[i]) != 0 && !(ccx[i] == pmx && ccy[i] == pmy);
    }
#undef SM_CLIPX
#undef SM_CLIPY
    /* lane l < 7 of every wave looks after candidate slot l, as in star_small_kernel */
    int sx = pmx, sy = pmy;
    bool on = lane == 0;
    if (lane == 1) { sx = as_i16(bmx << 2); sy = as_i16(bmy << 2); on = fpOn; }
    if (lane == 2) { sx = 0; sy = 0; on = zeroOn; }
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (lane == 3 + i) { sx = ccx[i]; sy = ccy[i]; on = cOn[i]; }
    on = on && lane < SM_SLOTS;
    int mvc = 0;
    if (on && lane != 0) mvc = ((int)ctab[sx - mvpx] + (int)ctab[sy - mvpy]) & 0xffff;

    /* the thread's part of the source block: its (tile, column) pairs for the vertical passes, its quad rows for the SADs */
    ColSrc ct[ITEMS];
#pragma unroll
    for (int u = 0; u < ITEMS; u++) small_source_column<S>(fenc0, fstride, pux, puy, tid + u * THREADS, ct[u]);
    const int tlx = (tid & (G::TW - 1)) * 4, tly = (tid >> G::LTW) * NQ;
    typename Px<P>::quad fq[NQ];
    {
        const P AS1 *src = fenc0 + (long)(puy + tly) * fstride + pux + tlx;
#pragma unroll
        for (int y = 0; y < NQ; y++) fq[y] = Px<P>::load_g(src + (long)y * fstride);
    }
    __syncthreads();
    int pb = 0;
#pragma unroll 1
    for (int k = 0; k < SM_SLOTS; k++)
    {
        if (!__builtin_amdgcn_readlane((int)on, k)) continue;
        const int kx = __builtin_amdgcn_readlane(sx, k), ky = __builtin_amdgcn_readlane(sy, k);
        int v = 0;
        if (((kx | ky) & 3) == 0)
        {
            const P AS1 *rp = ref0 + roff + (long)__mul24((ky >> 2) + tly, rstride) + (kx >> 2) + tlx;
#pragma unroll
            for (int y = 0; y < NQ; y++) v = Px<P>::sad(fq[y], Px<P>::load_g(rp + (long)y * rstride), v);
        }
        else
        {
            const int poff = L::PLANES + pb * G::PLANE;
            const P AS1 *src = ref0 + (roff + __mul24((ky >> 2) - 3, rstride) + (kx >> 2) - 3);
            const int *tp = (const int*)(x265amd_smem + SM_TAPS) + (kx & 3) * 12;
            for (int item = tid; item < G::IPP; item += THREADS)
            {
                const int r = item >> G::LTW, q = item & (G::TW - 1);
                hps_quad(src + (__mul24(r, rstride) + 4 * q), tp, (int16_t*)(x265amd_smem + poff) + (4 * q) * G::PROW + r, G::PROW, depth);
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < ITEMS; u++) v += small_column_cost<S>(poff, 0, ky & 3, false, ct[u], tid & 3, depth);
            pb ^= 1;
        }
        v = wave_total(v);
        if (lane == 0) cells[wave * 32 + k] = v;
    }
    __syncthreads();
    /* the reference's sequence of tests on the measured costs (motion.cpp:579-621) */
    int cst[SM_SLOTS];
#pragma unroll
    for (int k = 0; k < SM_SLOTS; k++)
    {
        int v = __builtin_amdgcn_readlane(mvc, k);
#pragma unroll
        for (int w = 0; w < WAVES; w++) v += uni(cells[w * 32 + k]);
        cst[k] = v;
    }
    int bprecost = cst[0], bcost = bprecost;
    unsigned nsad = 0, nsub = 1;
    if (fpOn) { bcost = cst[1]; nsad++; }
    if (zeroOn)
    {
        nsad++;
        if (cst[2] < bcost) { bcost = cst[2]; bmx = 0; bmy = 0; }
    }
    int bpx = pmx, bpy = pmy;
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (cOn[i] && !(ccx[i] == bpx && ccy[i] == bpy))
        {
            nsub++;
            if (cst[3 + i] < bprecost) { bprecost = cst[3 + i]; bpx = ccx[i]; bpy = ccy[i]; }
        }
    /* ---- the window of the 20 points over the planes (every thread is past its last vertical pass) ---- */
    constexpr int WP = (S + 12) * BPS;
    const int woff = L::PLANES;
    {
        typedef uint32_t __attribute__((aligned(1))) u32u;
        const char AS1 *w0 = (const char AS1*)(ref0 + roff + (long)__mul24(bmy - 4, rstride) + bmx - 4);
        for (int i = tid; i < (S + 8) * (WP / 4); i += THREADS)
        {
            const int r = i / (WP / 4), q = i - r * (WP / 4);
            *(uint32_t*)(x265amd_smem + woff + r * WP + 4 * q) = *(const u32u AS1*)(w0 + (long)r * rstride * BPS + 4 * q);
        }
    }
    __syncthreads();
    /* every point's SAD over the thread's quad rows: the loops are unrolled, so a point's offset in the window is a
     * constant next to the thread's own (dword-aligned) corner -- reads shared between points are made once, and a
     * point a multiple of four samples away needs no realignment */
    unsigned npts = 0;
    {
        constexpr PatTable<M> T = PatTable<M>();
        constexpr int NP = PatPoints<M>::N;
        const int room_l = bmx - mnx, room_r = mxx - bmx, room_u = bmy - mny, room_d = mxy - bmy;
        const int wbase = woff + tly * WP + tlx * BPS;
        int pv[NP];
#pragma unroll
        for (int p = 0; p < NP; p++)
        {
            int v = 0;
#pragma unroll
            for (int y2 = 0; y2 < NQ; y2++)
                v = Px<P>::sad(fq[y2], lds_quad_const<P>(wbase, (4 + T.dy[p] + y2) * WP + (4 + T.dx[p]) * BPS), v);
            pv[p] = v;
        }
#pragma unroll
        for (int p = 0; p < NP; p++)
        {
            /* STAR: the reference tests only the bounds a point can cross (motion.cpp:330-452); the other patterns'
             * points are measured unchecked */
            const bool okx = M != X265AMD_STAR || T.dx[p] == 0 || (T.dx[p] < 0 ? room_l >= -T.dx[p] : room_r >= T.dx[p]);
            const bool oky = M != X265AMD_STAR || T.dy[p] == 0 || (T.dy[p] < 0 ? room_u >= -T.dy[p] : room_d >= T.dy[p]);
            if (okx && oky)
            {
                npts++;
                const int v = wave_total(pv[p]);
                if (lane == 0) cells[wave * 32 + 8 + p] = v;
            }
        }
    }
    __syncthreads();
    if (tid >= 64) return;
    int key = 0x7fffffff;
    if (lane < PatPoints<M>::N)
    {
        int dx, dy;
        pat_point<M>(lane, dx, dy);
        const int x = as_i16(bmx + dx), y = as_i16(bmy + dy);
        if (M != X265AMD_STAR || ((dy >= 0 || y >= mny) && (dy <= 0 || y <= mxy) && (dx >= 0 || x >= mnx) && (dx <= 0 || x <= mxx)))
        {
            int v = 0;
#pragma unroll
            for (int w = 0; w < WAVES; w++) v += cells[w * 32 + 8 + lane];
            key = ((v + (((int)ctab[as_i16(x << 2) - mvpx] + (int)ctab[as_i16(y << 2) - mvpy]) & 0xffff)) << 6) | lane;
        }
    }
    if constexpr (M != X265AMD_STAR)
    {
        const int k0 = wave_min(lane < (M == X265AMD_HEX ? 6 : 4) ? key : 0x7fffffff), k1 = wave_min(lane >= 6 ? key : 0x7fffffff);
        if (tid != 0) return;
        int dir;
        unsigned counted;
        const bool over = pat_fold<M>(k0, k1, bmx, bmy, bcost, dir, counted);
        nsad += counted;
        int *res = (int*)(a.results + j);
        a.mark[j] = !over;
        if (!over)
        {
            StarFold F;
            F.start(bmx, bmy, bcost);
            F.pt = dir;
            F.store(a.state + 8 * (long)j, bprecost, bpx, bpy, nsad, nsub);
            res[2] = (int)SMALL_RESUME;
            return;
        }
        int qx, qy;
        if (bprecost < bcost) { qx = bpx; qy = bpy; bcost = bprecost; }
        else { qx = as_i16(bmx << 2); qy = as_i16(bmy << 2); }
        res[0] = (qx & 0xffff) | (qy << 16);
        res[1] = bcost;
        res[2] = (int)nsad;
        res[3] = (int)nsub;
        return;
    }
    const int rkey0 = wave_min(lane < 4 ? key : 0x7fffffff), rkey1 = wave_min(lane >= 4 && lane < 12 ? key : 0x7fffffff),
              rkey2 = wave_min(lane >= 12 ? key : 0x7fffffff);
    if (tid != 0) return;
    StarFold F;
    F.start(bmx, bmy, bcost);
    F.round(rkey0); F.round(rkey1); F.round(rkey2);
    nsad += npts;
    int *res = (int*)(a.results + j);
    a.mark[j] = F.rounds < 3;
    if (F.rounds < 3)
    {
        F.store(a.state + 8 * (long)j, bprecost, bpx, bpy, nsad, nsub);
        res[2] = (int)SMALL_RESUME;
        return;
    }
    int qx, qy;
    if (bprecost < bcost) { qx = bpx; qy = bpy; bcost = bprecost; }
    else { qx = as_i16(bmx << 2); qy = as_i16(bmy << 2); }
    res[0] = (qx & 0xffff) | (qy << 16);
    res[1] = bcost;
    res[2] = (int)nsad;
    res[3] = (int)nsub;
}

/* ---------------------------------------------------------------- the same stage for the larger 2NxN / Nx2N / pow-2 partitions
 *
 * star_big_kernel's stage for the other partitions of rect_class (2NxN / Nx2N from 32x16 up, every asymmetric shape from
 * 16x12 up), one wave per PU: a lane looks after W H / 256 (tile, column) pairs for the vertical passes and as many
 * quad rows for the plain SADs. One kernel for all shapes -- the wave reads its PU's shape and takes that shape's
 * straight-line code -- because the shapes of a device-resident list are not known to the host. */
template<int W, int H> struct RectGeo
{
    static constexpr int LTW = W == 8 ? 1 : W == 16 ? 2 : W == 32 ? 3 : 4;
    static constexpr int TW = W / 4;
    static constexpr int PAIRS = W * H / 4;                 /* (tile, column) pairs = quad rows of the PU */
    static constexpr int ITEMS = (PAIRS + 63) / 64;         /* of either, per lane */
    /* a power of two of tiles per row and whole items per lane: a lane's quad rows are a strip, one below the other;
     * otherwise (12, 24, 48 samples wide, or 16x12) they are every 64th quad row of the PU in raster order */
    static constexpr bool STRIP = (W == 8 || W == 16 || W == 32 || W == 64) && PAIRS % 64 == 0;
    static constexpr int PROW = H + 10;                     /* int16 rows of a plane column */
    static constexpr int PLANE = W * PROW * 2;
    static constexpr int IPP = (H + 8) * TW;                /* horizontal-pass items per plane */
    static constexpr int SIZE_CODE = W | (H << 8);
    static constexpr int CELLS = SM_FIXED;                  /* 8 cells of the start-up slots, then 20 of the points */
    static constexpr int PLANES = CELLS + 128;
    static constexpr int window(int bps) { return (H + 8) * (W + 12) * bps + 8; }
    static constexpr int total(int bps) { return PLANES + (2 * PLANE > window(bps) ? 2 * PLANE : window(bps)) + 32; }
};
/* LDS of the kernel for the shapes whose longer side is d samples (16, 32, 64): the largest of them */
static inline int star_rect_lds(int d, int bps)
{
    int m = 0;
    if (d == 16)
    {
        const int t[3] = { RectGeo<16, 12>::total(bps), RectGeo<12, 16>::total(bps), RectGeo<16, 16>::total(bps) };
        for (int i = 0; i < 3; i++) m = t[i] > m ? t[i] : m;
    }
    else if (d == 32)
    {
        const int t[7] = { RectGeo<32, 16>::total(bps), RectGeo<16, 32>::total(bps), RectGeo<32, 8>::total(bps), RectGeo<8, 32>::total(bps),
                           RectGeo<32, 24>::total(bps), RectGeo<24, 32>::total(bps), RectGeo<32, 32>::total(bps) };
        for (int i = 0; i < 7; i++) m = t[i] > m ? t[i] : m;
    }
    else
    {
        const int t[6] = { RectGeo<64, 32>::total(bps), RectGeo<32, 64>::total(bps), RectGeo<64, 16>::total(bps), RectGeo<16, 64>::total(bps),
                           RectGeo<64, 48>::total(bps), RectGeo<48, 64>::total(bps) };
        for (int i = 0; i < 6; i++) m = t[i] > m ? t[i] : m;
    }
    return m;
}

template<int W, int H, typename P, int M>
__device__ __forceinline__ void star_rect_pu(const SearchArgs &a, int j, const int *job)
{
    typedef RectGeo<W, H> G;
    constexpr int ITEMS = G::ITEMS, NQ = G::ITEMS, BPS = (int)sizeof(P);
    const int depth = sizeof(P) == 1 ? 8 : a.depth;
    const int lane = (int)threadIdx.x;
    int d[10];
#pragma unroll
    for (int i = 0; i < 10; i++) d[i] = uni(job[i]);
    const int pux = as_i16(d[0]), puy = d[0] >> 16, numc = (d[1] >> 16) & 0xff;
    const int mnx = as_i16(d[2]), mny = d[2] >> 16, mxx = as_i16(d[3]), mxy = d[3] >> 16;
    const int mvpx = as_i16(d[4]), mvpy = d[4] >> 16;
    const P AS1 *fenc0, *ref0;
    int fstride, rstride;
    small_planes<P>(a, d[1], fenc0, fstride, ref0, rstride);
    const int roff = puy * rstride + pux;
    const uint16_t AS1 *ctab = (const uint16_t AS1*)a.ctab;
    int *cells = (int*)(x265amd_smem + G::CELLS);

    /* ---- start-up candidates, motion.cpp:575-621 ---- */
    const int qmnx = as_i16(mnx << 2), qmny = as_i16(mny << 2), qmxx = as_i16(mxx << 2), qmxy = as_i16(mxy << 2);
#define SM_CLIPX(v) ((v) > qmxx ? qmxx : ((v) < qmnx ? qmnx : (v)))
#define SM_CLIPY(v) ((v) > qmxy ? qmxy : ((v) < qmny ? qmny : (v)))
    const int pmx = SM_CLIPX(mvpx), pmy = SM_CLIPY(mvpy);
    int bmx = as_i16((pmx + 2) >> 2), bmy = as_i16((pmy + 2) >> 2);
    const bool fpOn = ((pmx | pmy) & 3) != 0, zeroOn = (pmx | pmy) != 0;
    int ccx[4], ccy[4];
    bool cOn[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        ccx[i] = SM_CLIPX(as_i16(d[5 + i])); ccy[i] = SM_CLIPY(d[5 + i] >> 16);
        cOn[i] = i < numc && (ccx[i] | ccy[i]) != 0 && !(ccx[i] == pmx && ccy[i] == pmy);
    }
#undef SM_CLIPX
#undef SM_CLIPY
    /* lane l < 7 looks after candidate slot l, as in star_small_kernel */
    int sx = pmx, sy = pmy;
    bool on = lane == 0;
    if (lane == 1) { sx = as_i16(bmx << 2); sy = as_i16(bmy << 2); on = fpOn; }
    if (lane == 2) { sx = 0; sy = 0; on = zeroOn; }
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (lane == 3 + i) { sx = ccx[i]; sy = ccy[i]; on = cOn[i]; }
    on = on && lane < SM_SLOTS;
    int mvc = 0;
    if (on && lane != 0) mvc = ((int)ctab[sx - mvpx] + (int)ctab[sy - mvpy]) & 0xffff;

    /* the lane's part of the source block: its (tile, column) pairs for the vertical passes, its quad rows for the SADs */
    ColSrc ct[ITEMS];
    /* a lane beyond the PU's last pair (16x12, 12x16: 48 pairs) measures a pair of the first tile and adds nothing */
    const bool mine = G::PAIRS % 64 == 0 || lane < G::PAIRS % 64;
#pragma unroll
    for (int u = 0; u < ITEMS; u++)
    {
        const int l = (u < ITEMS - 1 || mine) ? lane + 64 * u : (lane & 3), t = l >> 2;
        ct[u].ty = t / G::TW; ct[u].col = (t % G::TW) * 4 + (l & 3);
        const P AS1 *src = fenc0 + (long)(puy + ct[u].ty * 4) * fstride + pux + ct[u].col;
        uint32_t f[4];
#pragma unroll
        for (int y = 0; y < 4; y++) f[y] = (uint32_t)src[(long)y * fstride];
        ct[u].f01 = f[0] | (f[1] << 16); ct[u].f23 = f[2] | (f[3] << 16);
    }
    /* the lane's quad rows: row sqy[u] of the PU, samples sqx[u] .. sqx[u] + 3 */
    int sqx[NQ], sqy[NQ];
#pragma unroll
    for (int u = 0; u < NQ; u++)
    {
        if constexpr (G::STRIP) { sqx[u] = (lane % G::TW) * 4; sqy[u] = (lane / G::TW) * NQ + u; }
        else
        {
            const int q = (u < NQ - 1 || mine) ? lane + 64 * u : 0;
            sqx[u] = (q % G::TW) * 4; sqy[u] = q / G::TW;
        }
    }
    typename Px<P>::quad fq[NQ];
#pragma unroll
    for (int y = 0; y < NQ; y++) fq[y] = Px<P>::load_g(fenc0 + (long)(puy + sqy[y]) * fstride + pux + sqx[y]);
    wave_sync();
    int pb = 0;
#pragma unroll 1
    for (int k = 0; k < SM_SLOTS; k++)
    {
        if (!__builtin_amdgcn_readlane((int)on, k)) continue;
        const int kx = __builtin_amdgcn_readlane(sx, k), ky = __builtin_amdgcn_readlane(sy, k);
        int v = 0;
        if (((kx | ky) & 3) == 0)
        {
            const P AS1 *rp = ref0 + roff + (long)__mul24(ky >> 2, rstride) + (kx >> 2);
#pragma unroll
            for (int y = 0; y < NQ; y++)
            {
                const int c = Px<P>::sad(fq[y], Px<P>::load_g(rp + (long)__mul24(sqy[y], rstride) + sqx[y]), 0);
                v += (y < NQ - 1 || mine) ? c : 0;
            }
        }
        else
        {
            const int poff = G::PLANES + pb * G::PLANE;
            const P AS1 *src = ref0 + (roff + __mul24((ky >> 2) - 3, rstride) + (kx >> 2) - 3);
            const int *tp = (const int*)(x265amd_smem + SM_TAPS) + (kx & 3) * 12;
            for (int item = lane; item < G::IPP; item += 64)
            {
                const int r = item / G::TW, q = item % G::TW;
                hps_quad(src + (__mul24(r, rstride) + 4 * q), tp, (int16_t*)(x265amd_smem + poff) + (4 * q) * G::PROW + r, G::PROW, depth);
            }
            wave_sync();
#pragma unroll
            for (int u = 0; u < ITEMS; u++)
            {
                int o[4];
                column_rows<4>(poff, G::PROW, 0, ky & 3, ct[u], depth, o);
                const int c = small_rows_cost(o[0], o[1], o[2], o[3], false, ct[u], lane & 3, depth);
                v += (u < ITEMS - 1 || mine) ? c : 0;
            }
            pb ^= 1;
        }
        v = wave_total(v);
        if (lane == 0) cells[k] = v;
    }
    wave_sync();
    /* the reference's sequence of tests on the measured costs (motion.cpp:579-621) */
    int cst[SM_SLOTS];
#pragma unroll
    for (int k = 0; k < SM_SLOTS; k++) cst[k] = __builtin_amdgcn_readlane(mvc, k) + uni(cells[k]);
    int bprecost = cst[0], bcost = bprecost;
    unsigned nsad = 0, nsub = 1;
    if (fpOn) { bcost = cst[1]; nsad++; }
    if (zeroOn)
    {
        nsad++;
        if (cst[2] < bcost) { bcost = cst[2]; bmx = 0; bmy = 0; }
    }
    int bpx = pmx, bpy = pmy;
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (cOn[i] && !(ccx[i] == bpx && ccy[i] == bpy))
        {
            nsub++;
            if (cst[3 + i] < bprecost) { bprecost = cst[3 + i]; bpx = ccx[i]; bpy = ccy[i]; }
        }
    /* ---- the window of the points over the planes (every lane is past its last vertical pass) ---- */
    constexpr int WP = (W + 12) * BPS;
    const int woff = G::PLANES;
    {
        typedef uint32_t __attribute__((aligned(1))) u32u;
        const char AS1 *w0 = (const char AS1*)(ref0 + roff + (long)__mul24(bmy - 4, rstride) + bmx - 4);
        for (int i = lane; i < (H + 8) * (WP / 4); i += 64)
        {
            const int r = i / (WP / 4), q = i - r * (WP / 4);
            *(uint32_t*)(x265amd_smem + woff + r * WP + 4 * q) = *(const u32u AS1*)(w0 + (long)r * rstride * BPS + 4 * q);
        }
    }
    wave_sync();
    /* every point's SAD over the lane's quad rows, unrolled: a point's offset in the window is a constant next to the
     * lane's own (dword-aligned) corner */
    unsigned npts = 0;
    {
        constexpr PatTable<M> T = PatTable<M>();
        constexpr int NP = PatPoints<M>::N;
        const int room_l = bmx - mnx, room_r = mxx - bmx, room_u = bmy - mny, room_d = mxy - bmy;
        /* (a strip's rows are the first row's address plus a constant: reads shared between points are made once) */
        int wbase[NQ];
#pragma unroll
        for (int y2 = 0; y2 < NQ; y2++) wbase[y2] = woff + (G::STRIP ? sqy[0] : sqy[y2]) * WP + sqx[y2] * BPS;
#pragma unroll
        for (int p = 0; p < NP; p++)
        {
            int v = 0;
#pragma unroll
            for (int y2 = 0; y2 < NQ; y2++)
            {
                const int c = Px<P>::sad(fq[y2], lds_quad_const<P>(wbase[y2], (4 + T.dy[p] + (G::STRIP ? y2 : 0)) * WP + (4 + T.dx[p]) * BPS), 0);
                v += (y2 < NQ - 1 || mine) ? c : 0;
            }
            /* STAR: the reference tests only the bounds a point can cross (motion.cpp:330-452); the other patterns'
             * points are measured unchecked */
            const bool okx = M != X265AMD_STAR || T.dx[p] == 0 || (T.dx[p] < 0 ? room_l >= -T.dx[p] : room_r >= T.dx[p]);
            const bool oky = M != X265AMD_STAR || T.dy[p] == 0 || (T.dy[p] < 0 ? room_u >= -T.dy[p] : room_d >= T.dy[p]);
            if (okx && oky)
            {
                npts++;
                v = wave_total(v);
                if (lane == 0) cells[8 + p] = v;
            }
        }
    }
    wave_sync();
    int key = 0x7fffffff;
    if (lane < PatPoints<M>::N)
    {
        int dx, dy;
        pat_point<M>(lane, dx, dy);
        const int x = as_i16(bmx + dx), y = as_i16(bmy + dy);
        if (M != X265AMD_STAR || ((dy >= 0 || y >= mny) && (dy <= 0 || y <= mxy) && (dx >= 0 || x >= mnx) && (dx <= 0 || x <= mxx)))
            key = ((cells[8 + lane] + (((int)ctab[as_i16(x << 2) - mvpx] + (int)ctab[as_i16(y << 2) - mvpy]) & 0xffff)) << 6) | lane;
    }
    StarFold F;
    bool over;
    if constexpr (M != X265AMD_STAR)
    {
        const int k0 = wave_min(lane < (M == X265AMD_HEX ? 6 : 4) ? key : 0x7fffffff), k1 = wave_min(lane >= 6 ? key : 0x7fffffff);
        int dir;
        unsigned counted;
        over = pat_fold<M>(k0, k1, bmx, bmy, bcost, dir, counted);
        nsad += counted;
        F.start(bmx, bmy, bcost);
        F.pt = dir;
    }
    else
    {
        const int rkey0 = wave_min(lane < 4 ? key : 0x7fffffff), rkey1 = wave_min(lane >= 4 && lane < 12 ? key : 0x7fffffff),
                  rkey2 = wave_min(lane >= 12 ? key : 0x7fffffff);
        F.start(bmx, bmy, bcost);
        F.round(rkey0); F.round(rkey1); F.round(rkey2);
        nsad += npts;
        over = F.rounds >= 3;
    }
    if (lane == 0)
    {
        int *res = (int*)(a.results + j);
        a.mark[j] = !over;
        if (!over)
        {
            F.store(a.state + 8 * (long)j, bprecost, bpx, bpy, nsad, nsub);
            res[2] = (int)SMALL_RESUME;
        }
        else
        {
            int qx, qy;
            if (bprecost < bcost) { qx = bpx; qy = bpy; bcost = bprecost; }
            else { qx = as_i16(bmx << 2); qy = as_i16(bmy << 2); }
            res[0] = (qx & 0xffff) | (qy << 16);
            res[1] = bcost;
            res[2] = (int)nsad;
            res[3] = (int)nsub;
        }
    }
}

/* D: the longer side of the shapes this instance takes (32 or 64) -- a kernel per class, so that the smaller shapes do not
 * run at the register count of 48x64's twelve pairs per lane. The two shapes with a longer side of 16 are taken by
 * star_tiny_kernel, which screens a max_dim-16 list anyway. */
template<int D, typename P, int M>
__device__ __forceinline__ void star_rect_shape(const SearchArgs &a, int j, const int *job, int code);

template<typename P, int M, int D>
__global__ __launch_bounds__(64) void star_rect_kernel(SearchArgs a)
{
    const int j = xcd_remap((int)blockIdx.x, a.njobs);
    if (j >= a.njobs) return;
    const int *job = (const int*)(a.jobs + j);
    const int code = uni(job[1]) & 0xffff, w = code & 0xff, h = code >> 8;
    if (!rect_takes(a, w, h) || (w > h ? w : h) != D) return;
    small_load_taps<P>((int)threadIdx.x, sizeof(P) == 1 ? 8 : a.depth);
    star_rect_shape<D, P, M>(a, j, job, code);
}

template<int D, typename P, int M>
__device__ __forceinline__ void star_rect_shape(const SearchArgs &a, int j, const int *job, int code)
{
    if constexpr (D == 16)
        switch (code)
        {
        case RectGeo<16, 12>::SIZE_CODE: star_rect_pu<16, 12, P, M>(a, j, job); break;
        case RectGeo<12, 16>::SIZE_CODE: star_rect_pu<12, 16, P, M>(a, j, job); break;
        case RectGeo<16, 16>::SIZE_CODE: star_rect_pu<16, 16, P, M>(a, j, job); break;
        }
    else if constexpr (D == 32)
        switch (code)
        {
        case RectGeo<32, 16>::SIZE_CODE: star_rect_pu<32, 16, P, M>(a, j, job); break;
        case RectGeo<16, 32>::SIZE_CODE: star_rect_pu<16, 32, P, M>(a, j, job); break;
        case RectGeo<32, 8>::SIZE_CODE: star_rect_pu<32, 8, P, M>(a, j, job); break;
        case RectGeo<8, 32>::SIZE_CODE: star_rect_pu<8, 32, P, M>(a, j, job); break;
        case RectGeo<32, 24>::SIZE_CODE: star_rect_pu<32, 24, P, M>(a, j, job); break;
        case RectGeo<24, 32>::SIZE_CODE: star_rect_pu<24, 32, P, M>(a, j, job); break;
        case RectGeo<32, 32>::SIZE_CODE: star_rect_pu<32, 32, P, M>(a, j, job); break;
        }
    else
        switch (code)
        {
        case RectGeo<64, 32>::SIZE_CODE: star_rect_pu<64, 32, P, M>(a, j, job); break;
        case RectGeo<32, 64>::SIZE_CODE: star_rect_pu<32, 64, P, M>(a, j, job); break;
        case RectGeo<64, 16>::SIZE_CODE: star_rect_pu<64, 16, P, M>(a, j, job); break;
        case RectGeo<16, 64>::SIZE_CODE: star_rect_pu<16, 64, P, M>(a, j, job); break;
        case RectGeo<64, 48>::SIZE_CODE: star_rect_pu<64, 48, P, M>(a, j, job); break;
        case RectGeo<48, 64>::SIZE_CODE: star_rect_pu<48, 64, P, M>(a, j, job); break;
        }
}

} // namespace x265amd
#endif
