/* Lookahead frame cost on the device: what EstimateRow::estimateCUCost computes for every
 * 8x8 lowres CU besides the motion search, and the sums CostEstimate::estimateFrameCost
 * forms from it (reference: source/encoder/slicetype.cpp:1471-1581, :1193-1204).
 *
 * None of it depends on a neighbouring CU, so unlike the search (a dependency wavefront)
 * these kernels are flat: one workgroup per (frame, CU row), which also makes the row sums
 * plain stores:
 *   la_cost_init_kernel   clears the frame totals
 *   la_intra_kernel       35-mode intra SATD: a wave takes every 8th CU of the row, one lane
 *                         per (mode, 4x4 tile), prediction and Hadamard in registers
 *   la_cost_kernel        bidirectional candidates, list / intra choice: four lanes per CU
 *                         (one per 4x4 tile)
 * Frame totals get one atomic add per row; sums are integer, so the order does not matter.
 */
#ifndef X265AMD_ME_LOOKAHEAD_COST_H
#define X265AMD_ME_LOOKAHEAD_COST_H

#include "me_search.h"

namespace x265amd {

struct FrameCostDev
{
    PlaneDev fenc;          /* lowresPlane[0] of frame b */
    PlaneDev ref0[4];       /* hpel planes of frame p0 */
    PlaneDev ref1[4];       /* hpel planes of frame p1 */
    int d0, d1;             /* b - p0, p1 - b */
    int intra_calculated, pad;
    const int16_t *mvs0; const int32_t *mv_costs0;
    const int16_t *mvs1; const int32_t *mv_costs1;
    const int32_t *inv_qscale;
    int32_t *intra_cost;
    uint16_t *lowres_costs;
    int32_t *row_satds_intra;
    int32_t *row_satds;
    long long *totals;      /* costEst, costEstAq, costIntra, costIntraAq, intraMbs */
};

#define LA_COST_MAX (1 << 28)   /* MotionEstimate::COST_MAX, motion.h:62 */

/* should this CU's cost contribute to the frame cost? slicetype.cpp:1412-1413 */
__device__ __forceinline__ bool la_scored(int cux, int cuy, int W, int H)
{
    return (cux > 0 && cux < W - 1 && cuy > 0 && cuy < H - 1) || W <= 2 || H <= 2;
}

__global__ void la_cost_init_kernel(const FrameCostDev *frames)
{
    if (threadIdx.x < 5) frames[blockIdx.x].totals[threadIdx.x] = 0;
}

/* Sum of up to three per-thread values over a workgroup of NW waves; the result is valid in
 * thread 0. Per-wave partial sums stay below 2^31 (a wave adds at most a few hundred costs). */
template<int NW>
__device__ __forceinline__ void la_block_sum3(int a, int b, int c, long long out[3])
{
    __shared__ int part[NW][3];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    a = seg_sum(a, 6); b = seg_sum(b, 6); c = seg_sum(c, 6);
    if (lane == 0) { part[wave][0] = a; part[wave][1] = b; part[wave][2] = c; }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = 0; k < 3; k++)
        {
            long long t = 0;
            for (int w = 0; w < NW; w++) t += part[w][k];
            out[k] = t;
        }
}

/* ---- intra: slicetype.cpp:1491-1559 ----
 * Per wave, LDS holds
 *   edge[f][s][0..16]  the block's reference edges as the reference builds them (:1495-1524):
 *                      f = 0 raw / 1 smoothed with [1 2 1] (ends kept), s = 0 above / 1 left,
 *                      index 0 the corner, 9..16 a repeat of index 8;
 *   src, srcT          the 8x8 source and its transpose (horizontal modes are compared against
 *                      the transpose, :1542-1543);
 *   ext[m - 11]        for the 15 modes with a negative angle: the main edge with its eight
 *                      projected side samples in front (intrapred.cpp:181-195), so that every
 *                      angular mode reads its main edge through one base address with plain
 *                      signed indexing. */
#define LA_INTRA_WAVES 8
#define LA_EDGE_BYTES 20
#define LA_EXT_BYTES 28
#define LA_OFF_SRC (4 * LA_EDGE_BYTES)
#define LA_OFF_SRCT (LA_OFF_SRC + 64)
#define LA_OFF_EXT (LA_OFF_SRCT + 64)
#define LA_INTRA_LDS (LA_OFF_EXT + 15 * LA_EXT_BYTES + 4)

/* 8192 / angle step of |mode - 10| or |mode - 26| (intrapred.cpp:167-170) */
__constant__ static const int16_t kInvAngle[9] = { 0, 4096, 1638, 910, 630, 482, 390, 315, 256 };

__device__ __forceinline__ int la_abs_angle(int idx)
{
    const unsigned long long angs = 0x1a15110d09050200ull;     /* 0 2 5 9 13 17 21 26 (a byte each), then 32 */
    return idx == 8 ? 32 : (int)((angs >> (8 * idx)) & 0xff);
}

/* DC with edge smoothing (intrapred.cpp:45-104) and planar from the smoothed edges (:106-154),
 * sample (row k, column l) */
__device__ __forceinline__ int la_pred_dc_planar(const uint8_t *e, int mode, int k, int l, int dc)
{
    if (mode == 0)
    {
        const uint8_t *a = e, *lf = e + LA_EDGE_BYTES;
        if ((k | l) == 0) return (a[1] + lf[1] + 2 * dc + 2) >> 2;
        if (k == 0) return (a[l + 1] + 3 * dc + 2) >> 2;
        if (l == 0) return (lf[k + 1] + 3 * dc + 2) >> 2;
        return dc;
    }
    const uint8_t *a = e + 2 * LA_EDGE_BYTES + 1, *lf = e + 3 * LA_EDGE_BYTES + 1;
    return ((lf[k] << 3) + 8 + (l + 1) * ((int)a[8] - lf[k]) + (a[l] << 3) + (k + 1) * ((int)lf[8] - a[l])) >> 4;
}

__global__ __launch_bounds__(LA_INTRA_WAVES * 64) void la_intra_kernel(const FrameCostDev *frames, int W, int H)
{
    __shared__ __attribute__((aligned(4))) uint8_t lds[LA_INTRA_WAVES][LA_INTRA_LDS];
    const FrameCostDev &f = frames[blockIdx.y];
    if (f.intra_calculated) return;                            /* block-uniform */
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cuy = blockIdx.x;
    const long stride = f.fenc.stride;
    int sum_cost = 0, sum_scored = 0, sum_aq = 0;              /* lane 0 of each wave accumulates */
    /* what a lane reads of a CU: a sample of the raw edges (lanes 0-16 above, 32-48 left) and one of the block. The
     * next CU's are fetched while this one is worked on. */
    const int ei = lane & 31, es = lane >> 5, ej = ei < 8 ? ei : 8;
    const long eoff = -1 - stride + (es ? ej * stride : ej), poff = (lane >> 3) * stride + (lane & 7);
    const uint8_t AS1 *row = (const uint8_t AS1*)f.fenc.origin + (long)(8 * cuy) * stride;
    uint8_t nedge = 0, npx = 0;
    if (wave < W) { nedge = row[8 * wave + eoff]; npx = row[8 * wave + poff]; }
    for (int cux = wave; cux < W; cux += LA_INTRA_WAVES)
    {
    const int cu = cuy * W + cux;
    uint8_t *e = lds[wave], *src = e + LA_OFF_SRC, *srct = e + LA_OFF_SRCT, *ext = e + LA_OFF_EXT;
    {
        if (ei <= 16) e[es * LA_EDGE_BYTES + ei] = nedge;
        src[lane] = npx;
        srct[(lane & 7) * 8 + (lane >> 3)] = npx;
        const int nx = cux + LA_INTRA_WAVES;
        if (nx < W) { nedge = row[8 * nx + eoff]; npx = row[8 * nx + poff]; }
    }
    wave_sync();
    {
        const int i = lane & 31, s = lane >> 5;
        if (i <= 16)
        {
            const uint8_t *r = e + s * LA_EDGE_BYTES;
            e[(2 + s) * LA_EDGE_BYTES + i] = (i == 0 || i == 16) ? r[i] : (uint8_t)((r[i - 1] + 2 * r[i] + r[i + 1] + 2) >> 2);
        }
    }
    wave_sync();
    /* extended main edges of modes 11..25: entry j < 8 is main index j - 8 = side[(128 + (8 - j) * invAngle) >> 8]
     * (entries the mode never reads may project past the side edge and are clamped), entries 8..24 the main edge
     * itself, copied as five dwords (the edges and the extended edges start on dwords) */
    for (int t = lane; t < 15 * 8; t += 64)
    {
        const int m = t >> 3, j = t & 7, mode = m + 11;
        const int fl = mode == 18, ver = mode >= 18;
        const uint8_t *main_ = e + (fl * 2 + (ver ? 0 : 1)) * LA_EDGE_BYTES, *side = e + (fl * 2 + (ver ? 1 : 0)) * LA_EDGE_BYTES;
        const int idx = ver ? 26 - mode : mode - 10;
        const int si = (128 + (8 - j) * kInvAngle[idx]) >> 8;
        ext[m * LA_EXT_BYTES + j] = side[si > 16 ? 16 : si];
        if (j < 5) *(uint32_t*)(ext + m * LA_EXT_BYTES + 8 + 4 * j) = *(const uint32_t*)(main_ + 4 * j);
    }
    wave_sync();
    /* DC: the eight samples above and the eight to the left (indices 1..8 of the raw edges), summed as bytes of dwords */
    int dc;
    {
        const uint32_t *ea = (const uint32_t*)e, *el = (const uint32_t*)(e + LA_EDGE_BYTES);
        uint32_t t = __builtin_amdgcn_sad_u8(ea[0] & 0xffffff00u, 0u, 8u);
        t = __builtin_amdgcn_sad_u8(ea[1], 0u, t);
        t = __builtin_amdgcn_sad_u8(ea[2] & 0xffu, 0u, t);
        t = __builtin_amdgcn_sad_u8(el[0] & 0xffffff00u, 0u, t);
        t = __builtin_amdgcn_sad_u8(el[1], 0u, t);
        t = __builtin_amdgcn_sad_u8(el[2] & 0xffu, 0u, t);
        dc = (int)(t >> 4);
    }
    int icost = LA_COST_MAX;
    for (int pass = 0; pass < 3; pass++)
    {
        const int task = pass * 64 + lane, mode = task >> 2, ty = (task >> 1) & 1, tx = task & 1;
        int v = 0;
        if (mode < 35)
        {
            /* the tile's prediction as packed pairs: columns (0, 2) and (1, 3) of every row, the layout the packed
             * Hadamard takes */
            typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
            pk16 P02[4], P13[4];
            if (mode < 2)
            {
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    int p[4];
#pragma unroll
                    for (int l = 0; l < 4; l++) p[l] = la_pred_dc_planar(e, mode, 4 * ty + k, 4 * tx + l, dc);
                    P02[k] = __builtin_bit_cast(pk16, (uint32_t)(p[0] | (p[2] << 16)));
                    P13[k] = __builtin_bit_cast(pk16, (uint32_t)(p[1] | (p[3] << 16)));
                }
            }
            else
            {
                /* angular (intrapred.cpp:156-259), modes below 18 left transposed as all_angs_pred_c
                 * leaves them (:262-289) */
                const int fl = (mode == 2 || mode == 18 || mode == 34);   /* IntraFilterType of 8x8 blocks, intrapred.cpp:35 */
                const int ver = mode >= 18;
                int ang = ver ? mode - 26 : 10 - mode;
                const int absAng = la_abs_angle(ang < 0 ? -ang : ang);
                const uint8_t *refMain = ang < 0 ? ext + (mode - 11) * LA_EXT_BYTES + 8 : e + (fl * 2 + (ver ? 0 : 1)) * LA_EDGE_BYTES;
                ang = ang < 0 ? -absAng : absAng;
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const int dp = (4 * ty + k + 1) * ang;
                    const unsigned short fr = (unsigned short)(dp & 31), fc = (unsigned short)(32 - (dp & 31));
                    /* five edge samples r0..r4 from two aligned dwords (the wave's LDS block is dword aligned) */
                    const uint8_t *r = refMain + 4 * tx + (dp >> 5) + 1;
                    const unsigned sh = (unsigned)(uintptr_t)r & 3u;
                    const uint32_t *w = (const uint32_t*)(r - sh);
                    const uint32_t q0 = w[0], q1 = w[1];
                    const uint32_t lo = __builtin_amdgcn_alignbyte(q1, q0, sh), r4 = (q1 >> (8 * sh)) & 0xffu;
                    const u16x2 A = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(lo, lo, 0x0c020c00u));   /* r0, r2 */
                    const u16x2 B = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(lo, lo, 0x0c030c01u));   /* r1, r3 */
                    const u16x2 C = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(r4, lo, 0x0c040c02u));   /* r2, r4 */
                    const u16x2 wf = { fr, fr }, wc = { fc, fc }, rnd = { 16, 16 };
                    /* with fr = 0 this is the plain copy of :233-240 */
                    P02[k] = __builtin_bit_cast(pk16, (u16x2)((A * wc + B * wf + rnd) >> 5));
                    P13[k] = __builtin_bit_cast(pk16, (u16x2)((B * wc + C * wf + rnd) >> 5));
                }
                if (absAng == 0 && tx == 0)
                {
                    /* edge filter of the pure horizontal / vertical modes, bFilter = (cuSize <= 16) (:203-209) */
                    const uint8_t *refSide = e + (ver ? 1 : 0) * LA_EDGE_BYTES;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                    {
                        const int t = (int)(unsigned short)P02[k].x + (((int)refSide[4 * ty + k + 1] - (int)refSide[0]) >> 1);
                        P02[k].x = (short)(t < 0 ? 0 : (t > 255 ? 255 : t));
                    }
                }
            }
            const uint8_t *sp = ((mode >= 2 && mode < 18) ? srct : src) + (4 * ty) * 8 + 4 * tx;
            pk16 E[4], O[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const uint32_t q = *(const uint32_t*)(sp + 8 * k);
                E[k] = even_bytes(q) - P02[k];
                O[k] = odd_bytes(q) - P13[k];
            }
            v = had4x4_half_pk(E, O);
        }
        v = seg_sum(v, 2);
        v = mode < 35 ? v : LA_COST_MAX;
        v = wave_min(v);
        icost = v < icost ? v : icost;
    }
    if (lane == 0)
    {
        f.intra_cost[cu] = icost;
        sum_cost += icost;
        if (la_scored(cux, cuy, W, H))
        {
            sum_scored += icost;
            if (f.inv_qscale) sum_aq += (int)((unsigned)icost * (unsigned)f.inv_qscale[cu] + 128u) >> 8;
        }
    }
    wave_sync();                                               /* the next CU overwrites this wave's LDS */
    }
    long long tot[3];
    la_block_sum3<LA_INTRA_WAVES>(sum_cost, sum_scored, sum_aq, tot);
    if (threadIdx.x == 0)
    {
        f.row_satds_intra[cuy] = (int)tot[0];
        atomicAdd((unsigned long long*)&f.totals[2], (unsigned long long)tot[1]);
        if (f.inv_qscale) atomicAdd((unsigned long long*)&f.totals[3], (unsigned long long)tot[2]);
    }
}

/* ---- inter / bidir choice: slicetype.cpp:1432-1490, :1560-1581 ---- */

/* four pixels of ReferencePlanes::lowresMC (lowres.h:56-76) at qpel MV (qx, qy) */
template<typename P>
__device__ __forceinline__ typename Px<P>::quad la_mc_quad(const PlaneDev *planes, long off, int qx, int qy, int y)
{
    const int hpelA = (qy & 2) | ((qx & 2) >> 1);
    const long stride = planes[0].stride;
    const P AS1 *a = (const P AS1*)planes[hpelA].origin + off + (long)((qy >> 2) + y) * stride + (qx >> 2);
    typename Px<P>::quad q = Px<P>::load_g(a);
    if ((qx | qy) & 1)
    {
        const int bx = as_i16(qx + (qx & 1) * 2), by = as_i16(qy + (qy & 1) * 2);
        const int hpelB = (by & 2) | ((bx & 2) >> 1);
        const P AS1 *b = (const P AS1*)planes[hpelB].origin + off + (long)((by >> 2) + y) * stride + (bx >> 2);
        q = Px<P>::avg(q, Px<P>::load_g(b));
    }
    return q;
}

#define LA_COST_THREADS 256
template<typename P>
__global__ __launch_bounds__(LA_COST_THREADS) void la_cost_kernel(const FrameCostDev *frames, int W, int H)
{
    const FrameCostDev &f = frames[blockIdx.y];
    const int cuy = blockIdx.x, tile = threadIdx.x & 3;
    const bool bidir = f.d1 > 0, inter = f.d0 + f.d1 > 0;
    int sum_cost = 0, sum_scored = 0, sum_aq = 0, sum_intra = 0;   /* tile-0 lanes accumulate */
    for (int cx0 = 0; cx0 < W; cx0 += LA_COST_THREADS / 4)
    {
        const int cux_ = cx0 + (threadIdx.x >> 2);
        const bool live = cux_ < W;
        const int cux = live ? cux_ : 0, cu = cuy * W + cux;
        int bcost = LA_COST_MAX, listused = 0;
        if (inter)
        {
            const int c0 = f.mv_costs0[cu];
            if (c0 < bcost) { bcost = c0; listused = 1; }
        }
        if (bidir)
        {
            const int c1 = f.mv_costs1[cu];
            if (c1 < bcost) { bcost = c1; listused = 2; }
            const long off = (long)(8 * cuy + 4 * (tile >> 1)) * f.fenc.stride + 8 * cux + 4 * (tile & 1);
            const int m0x = f.mvs0[2 * cu], m0y = f.mvs0[2 * cu + 1], m1x = f.mvs1[2 * cu], m1y = f.mvs1[2 * cu + 1];
            typename Px<P>::quad s[4], q[4], z[4];
#pragma unroll
            for (int y = 0; y < 4; y++)
            {
                s[y] = Px<P>::load_g((const P AS1*)f.fenc.origin + off + (long)y * f.fenc.stride);
                q[y] = Px<P>::avg(la_mc_quad<P>(f.ref0, off, m0x, m0y, y), la_mc_quad<P>(f.ref1, off, m1x, m1y, y));
                z[y] = Px<P>::avg(Px<P>::load_g((const P AS1*)f.ref0[0].origin + off + (long)y * f.ref0[0].stride),
                                  Px<P>::load_g((const P AS1*)f.ref1[0].origin + off + (long)y * f.ref1[0].stride));
            }
            const int bi = seg_sum(tile_cost_quads<P>(s, q, true), 2);
            const int bz = seg_sum(tile_cost_quads<P>(s, z, true), 2);
            if (bi < bcost) { bcost = bi; listused = 3; }
            if (bz < bcost) { bcost = bz; listused = 3; }
        }
        if (live && !tile)
        {
            const bool scored = la_scored(cux, cuy, W, H);
            if (!bidir)
            {
                const int ic = f.intra_cost[cu];
                if (ic < bcost)
                {
                    if (scored) sum_intra++;
                    bcost = ic;
                    listused = 0;
                }
            }
            if (inter)
            {
                sum_cost += bcost;
                if (scored)
                {
                    sum_scored += bcost;
                    if (f.inv_qscale) sum_aq += (int)((unsigned)bcost * (unsigned)f.inv_qscale[cu] + 128u) >> 8;
                }
            }
            f.lowres_costs[cu] = (uint16_t)((bcost < 16383 ? bcost : 16383) | (listused << 14));
        }
    }
    long long tot[3], tot2[3];
    la_block_sum3<LA_COST_THREADS / 64>(sum_cost, sum_scored, sum_aq, tot);
    __syncthreads();
    la_block_sum3<LA_COST_THREADS / 64>(sum_intra, 0, 0, tot2);
    if (threadIdx.x == 0)
    {
        if (inter)
        {
            f.row_satds[cuy] = (int)tot[0];
            atomicAdd((unsigned long long*)&f.totals[0], (unsigned long long)tot[1]);
            if (f.inv_qscale) atomicAdd((unsigned long long*)&f.totals[1], (unsigned long long)tot[2]);
        }
        if (tot2[0]) atomicAdd((unsigned long long*)&f.totals[4], (unsigned long long)tot2[0]);
    }
}

/* ---- explicit weighted prediction ---- */

/* weight_pp_c (pixel.cpp:512-529) on four packed samples: through the 14-bit intermediate as
 * int16, (w0 * v + round) >> shift, + offset, clipped to the sample range */
template<typename P>
__device__ __forceinline__ typename Px<P>::quad weight_quad(typename Px<P>::quad q, int w0, int round, int shift, int offset, int depth)
{
    int v[4];
    Px<P>::unpack(q, v);
    const int maxv = (1 << depth) - 1;
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        int r = ((w0 * as_i16(v[i] << (14 - depth)) + round) >> shift) + offset;
        v[i] = r < 0 ? 0 : (r > maxv ? maxv : r);
    }
    if constexpr (sizeof(P) == 1)
        return (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
    else
        return make_uint2((uint32_t)v[0] | ((uint32_t)v[1] << 16), (uint32_t)v[2] | ((uint32_t)v[3] << 16));
}

/* A whole padded buffer (the lookahead weights Lowres::buffer[i] end to end, slicetype.cpp:1333-1337):
 * flat, 16 bytes per thread. Both buffers start 16-byte aligned and are padded to a multiple of 16. */
template<typename P>
__global__ void weight_buffer_kernel(const uint4 *src, uint4 *dst, long nvec, int w0, int round, int shift, int offset, int depth)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvec) return;
    const uint4 a = src[i];
    uint4 o;
    if constexpr (sizeof(P) == 1)
    {
        o.x = weight_quad<P>(a.x, w0, round, shift, offset, depth); o.y = weight_quad<P>(a.y, w0, round, shift, offset, depth);
        o.z = weight_quad<P>(a.z, w0, round, shift, offset, depth); o.w = weight_quad<P>(a.w, w0, round, shift, offset, depth);
    }
    else
    {
        const uint2 lo = weight_quad<P>(make_uint2(a.x, a.y), w0, round, shift, offset, depth);
        const uint2 hi = weight_quad<P>(make_uint2(a.z, a.w), w0, round, shift, offset, depth);
        o = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
    dst[i] = o;
}

/* MotionReference::applyWeight over a whole picture (reference.cpp:211-256) in one pass. The
 * reference weights the picture rows and then replicates the weighted edge samples into the
 * margins; weighting is pointwise, so the padded output is weight(source sample at the clamped
 * position). Flat over the padded buffer, 16 bytes per thread, all stores aligned; a chunk that
 * lies inside one picture row reads its 16 source bytes at once. Columns between width + 2 *
 * margin and the stride are cleared. src and dst have the same geometry. */
template<typename P>
__global__ void apply_weight_kernel(const P *src_base, P *dst_base, long nvec, int stride, int width, int height, int mx, int my,
                                    int w0, int round, int shift, int offset, int depth)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvec) return;
    constexpr int N = 16 / (int)sizeof(P);
    const long e0 = i * N;
    const int row = (int)(e0 / stride), col = (int)(e0 - (long)row * stride);
    const int y = row - my, x = col - mx;
    uint4 o;
    if (y >= 0 && y < height && x >= 0 && x + N <= width)
    {
        const uint4 a = *(const uint4*)(src_base + e0);
        if constexpr (sizeof(P) == 1)
        {
            o.x = weight_quad<P>(a.x, w0, round, shift, offset, depth); o.y = weight_quad<P>(a.y, w0, round, shift, offset, depth);
            o.z = weight_quad<P>(a.z, w0, round, shift, offset, depth); o.w = weight_quad<P>(a.w, w0, round, shift, offset, depth);
        }
        else
        {
            const uint2 lo = weight_quad<P>(make_uint2(a.x, a.y), w0, round, shift, offset, depth);
            const uint2 hi = weight_quad<P>(make_uint2(a.z, a.w), w0, round, shift, offset, depth);
            o = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    }
    else
    {
        const int maxv = (1 << depth) - 1, rows = height + 2 * my;
        uint32_t w[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (int k = 0; k < N; k++)
        {
            int c = col + k, r = row;
            if (c >= stride) { c -= stride; r++; }
            int v = 0;
            if (c < width + 2 * mx && r < rows)
            {
                int sx = c - mx, sy = r - my;
                sx = sx < 0 ? 0 : (sx >= width ? width - 1 : sx);
                sy = sy < 0 ? 0 : (sy >= height ? height - 1 : sy);
                const int p = src_base[(long)(sy + my) * stride + sx + mx];
                v = ((w0 * as_i16(p << (14 - depth)) + round) >> shift) + offset;
                v = v < 0 ? 0 : (v > maxv ? maxv : v);
            }
            w[k * (int)sizeof(P) / 4] |= (uint32_t)v << (8 * ((k * (int)sizeof(P)) & 3));
        }
        o = make_uint4(w[0], w[1], w[2], w[3]);
    }
    *(uint4*)(dst_base + e0) = o;
}

/* CostEstimate::weightCostLuma (slicetype.cpp:1223-1256): sum over the 8x8 blocks of
 * min(SATD(weighted reference block, source block), intraCost). One workgroup per CU row, four
 * lanes per block, the weighting applied to the loaded samples instead of a weighted copy. */
template<typename P>
__global__ __launch_bounds__(LA_COST_THREADS) void la_weight_cost_kernel(PlaneDev fenc, PlaneDev srcp, const int32_t *intra_cost, int W,
                                                                         int use_wp, int w0, int round, int shift, int offset, int depth,
                                                                         unsigned int *out)
{
    const int cuy = blockIdx.x, tile = threadIdx.x & 3;
    int sum = 0;
    for (int cx0 = 0; cx0 < W; cx0 += LA_COST_THREADS / 4)
    {
        const int cux_ = cx0 + (threadIdx.x >> 2);
        const bool live = cux_ < W;
        const int cux = live ? cux_ : 0;
        const long off = (long)(8 * cuy + 4 * (tile >> 1)) * fenc.stride + 8 * cux + 4 * (tile & 1);
        typename Px<P>::quad s[4], q[4];
#pragma unroll
        for (int y = 0; y < 4; y++)
        {
            s[y] = Px<P>::load_g((const P AS1*)fenc.origin + off + (long)y * fenc.stride);
            q[y] = Px<P>::load_g((const P AS1*)srcp.origin + off + (long)y * srcp.stride);
            if (use_wp) q[y] = weight_quad<P>(q[y], w0, round, shift, offset, depth);
        }
        /* satd(src, fenc): the Hadamard abs-sum does not depend on the sign of the difference */
        const int satd = seg_sum(tile_cost_quads<P>(s, q, true), 2);
        if (live && !tile)
        {
            const int ic = intra_cost[cuy * W + cux];
            sum += satd < ic ? satd : ic;
        }
    }
    long long tot[3];
    la_block_sum3<LA_COST_THREADS / 64>(sum, 0, 0, tot);
    if (threadIdx.x == 0) atomicAdd(out, (unsigned int)tot[0]);
}

/* ---- bi-prediction estimate of predInterSearch (TEncSearch.cpp:4346-4368) ----
 * One wave per PU, a lane per 4x4 tile (lanes loop when the PU has more than 64 tiles): both
 * predictions of the tile are interpolated in registers from global memory
 * (xPredInterLumaBlk = the luma_hpp / vpp / hps+vsp chain of interp_tile), averaged with
 * rounding, and the 4x4 Hadamard of the difference to the source is summed over the tiles. */
struct BipredArgs
{
    PlaneDev fenc, ref0, ref1;
    int depth, njobs;
    const x265amd_bipred_job *jobs;
    x265amd_bipred_result *results;
};

template<typename P>
__device__ __forceinline__ void bipred_tile(const P AS1 *r, int stride, int qx, int qy, int depth, int o[4][4])
{
    const int xf = qx & 3, yf = qy & 3;
    GlobalLoader<P> ld = { r + (long)(qy >> 2) * stride + (qx >> 2), stride };
    if ((xf | yf) == 0)
    {
#pragma unroll
        for (int y = 0; y < 4; y++) Px<P>::unpack(ld(0, y), o[y]);
    }
    else
        interp_tile<P>(ld, xf, yf, depth, o);
}

template<typename P>
__global__ __launch_bounds__(64) void bipred_kernel(BipredArgs a)
{
    const int j = blockIdx.x;
    if (j >= a.njobs) return;
    const x265amd_bipred_job *job = a.jobs + j;
    const int lane = threadIdx.x & 63;
    const int w = uni(job->w), h = uni(job->h), pux = uni(job->pu_x), puy = uni(job->pu_y);
    const int m0x = uni(job->mv0[0]), m0y = uni(job->mv0[1]), m1x = uni(job->mv1[0]), m1y = uni(job->mv1[1]);
    const int wt = w >> 2, T = wt * (h >> 2);
    int bi = 0, zero = 0;
    for (int t = lane; t < T; t += 64)
    {
        const int ty = t / wt, tx = t - ty * wt;
        const long fo = (long)(puy + 4 * ty) * a.fenc.stride + pux + 4 * tx;
        const long r0o = (long)(puy + 4 * ty) * a.ref0.stride + pux + 4 * tx, r1o = (long)(puy + 4 * ty) * a.ref1.stride + pux + 4 * tx;
        typename Px<P>::quad f[4], z[4];
#pragma unroll
        for (int y = 0; y < 4; y++)
        {
            f[y] = Px<P>::load_g((const P AS1*)a.fenc.origin + fo + (long)y * a.fenc.stride);
            z[y] = Px<P>::avg(Px<P>::load_g((const P AS1*)a.ref0.origin + r0o + (long)y * a.ref0.stride),
                              Px<P>::load_g((const P AS1*)a.ref1.origin + r1o + (long)y * a.ref1.stride));
        }
        zero += tile_cost_quads<P>(f, z, true);
        int o0[4][4], o1[4][4];
        bipred_tile<P>((const P AS1*)a.ref0.origin + r0o, a.ref0.stride, m0x, m0y, a.depth, o0);
        bipred_tile<P>((const P AS1*)a.ref1.origin + r1o, a.ref1.stride, m1x, m1y, a.depth, o1);
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 4; x++) o0[y][x] = (o0[y][x] + o1[y][x] + 1) >> 1;
        bi += tile_cost_px<P>(f, o0, true);
    }
    bi = seg_sum(bi, 6);
    zero = seg_sum(zero, 6);
    if (lane == 0) { a.results[j].satd = bi; a.results[j].satd_zero = zero; }
}

/* ---- merge-candidate estimate of the CU analysis (compress.cpp:248-271) ----
 * A lane per 4x4 tile (merge_flat_kernel, further down). Lanes are ordered so that a quad of lanes is an
 * 8x8 block (2 x 2 tiles) and sixteen lanes a 16x16 block: the 8x8 Hadamard of sa8d is the 4x4
 * Hadamard of each tile followed by sum / difference steps across the quad (DPP), and the 16x16
 * rounding of sa8d_16x16 (pixel.cpp:332-342) is a sum over sixteen lanes. */
struct MergeArgs
{
    PlaneDev fenc;
    const PlaneDev *refs;      /* device array of nrefs planes */
    int depth, njobs;
    const x265amd_merge_job *jobs;
    int32_t *out;
};

/* xPredInterLumaBlk into 14-bit intermediates (TComPrediction.cpp:441-476): luma_p2s, luma_hps,
 * filterVertical_ps or luma_hps + filterVertical_ss (ipfilter.cpp:84-172), one 4x4 tile */
template<typename P>
__device__ __forceinline__ void mc_short_tile(const P AS1 *r, int stride, int qx, int qy, int depth, int o[4][4])
{
    const int xf = qx & 3, yf = qy & 3, headRoom = 14 - depth;
    const int shift = 6 - headRoom, off = -(8192 << shift);
    GlobalLoader<P> ld = { r + (long)(qy >> 2) * stride + (qx >> 2), stride };
    int ch[8], cv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { ch[k] = kLumaTaps[xf][k]; cv[k] = kLumaTaps[yf][k]; }
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
                /* with xf = 0 the taps are {0,0,0,64,..}: (64 p - 8192 << shift) >> shift = (p << headRoom) - 8192 */
                o[y][x] = as_i16((sum + off) >> shift);
            }
        }
        return;
    }
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
                win[7][x] = as_i16((sum + off) >> shift);
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
            o[y][x] = xf == 0 ? as_i16((sum + off) >> shift) : as_i16(sum >> 6);
        }
    }
}

/* ---- motion compensation into a prediction picture, luma + 4:2:0 chroma ----
 * TComPrediction::motionCompensation without weighted prediction (TComPrediction.cpp:201-286).
 * The unit of work is a 4x4 luma tile together with the 2x2 block of Cb and of Cr under it (chroma
 * blocks of every HEVC partition are even-sized). The tiles of all PUs of a launch are numbered
 * consecutively (PU j owns tiles [first[j], first[j+1])); a wave takes 64 consecutive tiles, whatever
 * PUs they belong to, so 8x8 PUs fill a wave as well as 64x64 ones do. The host supplies first[] and,
 * per wave, the PU its first tile belongs to; the lanes find their own PU from the 64 entries of
 * first[] that follow (a flag per tile where a PU starts, one ballot, a bit count).
 * All samples are interpolated in registers from global memory and stored once. */
__constant__ static const int16_t kChromaTaps[8][4] = {        /* g_chromaFilter, TComRom.cpp:183-193 */
    {  0, 64,  0,  0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
    { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };

struct McRef                   /* one reference picture: Y, Cb, Cr planes and its explicit weights */
{
    PlaneDev pl[3];
    x265amd_mc_weight wp;
};

struct McArgs
{
    const McRef *refs;         /* device array, one entry per reference picture */
    PlaneDev pred[3];
    int depth, njobs, nchunks;
    const x265amd_mc_job *jobs;
    const int32_t *first;      /* njobs + 64 entries: first tile of each PU, then the tile total repeated */
    const int32_t *chunk_job;  /* nchunks entries: the PU that tile 64 * k belongs to */
    const x265amd_result *res0, *res1;   /* optional: search results that supply the MVs of list 0 / list 1 */
};

/* Parameters of explicit weighted prediction for one component, as getWpScaling derives them
 * (TComWeightPrediction.cpp:534-560), and the sample formulas weightUnidir / weightBidir (:46-64).
 * Samples are the 14-bit intermediates (value - 8192); `bias` adds the 8192 back. With weights (1, 1),
 * bias 0 and round = 1 << (shift - 1) the two-list formula is TComYuv::addAvg's, whose offset
 * 2 * 8192 stands for the same bias. */
struct McWeights
{
    int w0, w1, shift, round, offset, bias;
    __device__ __forceinline__ int apply(int s0, int s1, bool two, int maxv) const
    {
        int v;
        if (two) v = (w0 * (s0 + bias) + w1 * (s1 + bias) + round + offset) >> shift;
        else v = ((w0 * (s0 + bias) + round) >> shift) + offset;
        return v < 0 ? 0 : (v > maxv ? maxv : v);
    }
};
/* a: the weights of list A (the only list, or list 0), b: of list 1 */
__device__ __forceinline__ McWeights mc_weights(const x265amd_mc_weight &a, const x265amd_mc_weight &b, int comp, bool two, int depth)
{
    McWeights W;
    const int scale = 1 << (depth - 8), shiftNum = 14 - depth;
    W.bias = 8192;
    W.w0 = a.w[comp]; W.w1 = b.w[comp];
    if (two)
    {
        W.shift = a.log2denom[comp] + 1 + shiftNum;
        W.round = 1 << (W.shift - 1);
        W.offset = (a.offset[comp] + b.offset[comp]) * scale * (1 << (W.shift - 1));   /* added before the shift */
    }
    else
    {
        W.shift = a.log2denom[comp] + shiftNum;
        W.round = W.shift ? 1 << (W.shift - 1) : 0;
        W.offset = a.offset[comp] * scale;                                              /* added after the shift */
    }
    return W;
}

/* N consecutive samples starting at p, as one wide load (any alignment) */
template<typename P, int N> struct RowLoad;
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
/* Horizontal filter sums of a loaded row on the dot-product instructions. 8-bit samples are biased to signed bytes for
 * v_dot4_i32_i8; the taps of every phase add up to 64, so the bias comes back as + 128 * 64. 16-bit samples (at most 12
 * bits) are signed halves for v_dot2_i32_i16 as they are. */
__device__ __forceinline__ int mc_dot2(uint32_t a, uint32_t b, int c)
{
    typedef short h2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b), c, false);
}
#define MC_B(a, b, c, d) ((uint32_t)((a) & 0xff) | ((uint32_t)((b) & 0xff) << 8) | ((uint32_t)((c) & 0xff) << 16) | ((uint32_t)((d) & 0xff) << 24))
#define MC_H(a, b) ((uint32_t)((a) & 0xffff) | ((uint32_t)((b) & 0xffff) << 16))
#define MC_L8(a, b, c, d, e, f, g, h) { MC_B(a, b, c, d), MC_B(e, f, g, h) }
#define MC_L16(a, b, c, d, e, f, g, h) { MC_H(a, b), MC_H(c, d), MC_H(e, f), MC_H(g, h) }
__constant__ static const uint32_t kMcLuma8[4][2] = { MC_L8(0, 0, 0, 64, 0, 0, 0, 0), MC_L8(-1, 4, -10, 58, 17, -5, 1, 0),
                                                      MC_L8(-1, 4, -11, 40, 40, -11, 4, -1), MC_L8(0, 1, -5, 17, 58, -10, 4, -1) };
__constant__ static const uint32_t kMcLuma16[4][4] = { MC_L16(0, 0, 0, 64, 0, 0, 0, 0), MC_L16(-1, 4, -10, 58, 17, -5, 1, 0),
                                                       MC_L16(-1, 4, -11, 40, 40, -11, 4, -1), MC_L16(0, 1, -5, 17, 58, -10, 4, -1) };
__constant__ static const uint32_t kMcChroma8[8] = { MC_B(0, 64, 0, 0), MC_B(-2, 58, 10, -2), MC_B(-4, 54, 16, -2), MC_B(-6, 46, 28, -4),
                                                     MC_B(-4, 36, 36, -4), MC_B(-4, 28, 46, -6), MC_B(-2, 16, 54, -4), MC_B(-2, 10, 58, -2) };
__constant__ static const uint32_t kMcChroma16[8][2] = { { MC_H(0, 64), MC_H(0, 0) }, { MC_H(-2, 58), MC_H(10, -2) }, { MC_H(-4, 54), MC_H(16, -2) },
                                                         { MC_H(-6, 46), MC_H(28, -4) }, { MC_H(-4, 36), MC_H(36, -4) }, { MC_H(-4, 28), MC_H(46, -6) },
                                                         { MC_H(-2, 16), MC_H(54, -4) }, { MC_H(-2, 10), MC_H(58, -2) } };
/* packed taps of a lane's horizontal phases: l = luma (2 or 4 words), c = chroma (1 or 2 words) */
template<typename P> struct McTaps;
template<> struct McTaps<uint8_t>
{
    uint32_t l[2], c[1];
    __device__ __forceinline__ void load(int xf, int cxf) { l[0] = kMcLuma8[xf][0]; l[1] = kMcLuma8[xf][1]; c[0] = kMcChroma8[cxf]; }
};
template<> struct McTaps<uint16_t>
{
    uint32_t l[4], c[2];
    __device__ __forceinline__ void load(int xf, int cxf)
    {
#pragma unroll
        for (int i = 0; i < 4; i++) l[i] = kMcLuma16[xf][i];
        c[0] = kMcChroma16[cxf][0]; c[1] = kMcChroma16[cxf][1];
    }
};

template<> struct RowLoad<uint8_t, 12>
{
    u32x3_t v;
    /* H[x] = sum over k < 8 of sample(x + k) * tap[k], x < 4 */
    __device__ __forceinline__ void hsum(const McTaps<uint8_t> &t, int H[4]) const
    {
        const uint32_t d0 = v[0] ^ 0x80808080u, d1 = v[1] ^ 0x80808080u, d2 = v[2] ^ 0x80808080u;
        H[0] = __builtin_amdgcn_sdot4(d0, t.l[0], __builtin_amdgcn_sdot4(d1, t.l[1], 8192, false), false);
#pragma unroll
        for (int x = 1; x < 4; x++)
            H[x] = __builtin_amdgcn_sdot4(__builtin_amdgcn_alignbyte(d1, d0, x), t.l[0],
                                          __builtin_amdgcn_sdot4(__builtin_amdgcn_alignbyte(d2, d1, x), t.l[1], 8192, false), false);
    }
    __device__ __forceinline__ void load(const uint8_t AS1 *p) { typedef u32x3_t __attribute__((aligned(1))) T; v = *(const T AS1*)p; }
    __device__ __forceinline__ void zero() { v = (u32x3_t)(0u); }
    __device__ __forceinline__ int px(int i) const { return (int)((v[i >> 2] >> (8 * (i & 3))) & 0xff); }
};
template<> struct RowLoad<uint8_t, 8>
{
    u32x2_t v;
    /* H[x] = sum over k < 4 of sample(x + k) * tap[k], x < 2 */
    __device__ __forceinline__ void hsum(const McTaps<uint8_t> &t, int H[2]) const
    {
        const uint32_t d0 = v[0] ^ 0x80808080u, d1 = v[1] ^ 0x80808080u;
        H[0] = __builtin_amdgcn_sdot4(d0, t.c[0], 8192, false);
        H[1] = __builtin_amdgcn_sdot4(__builtin_amdgcn_alignbyte(d1, d0, 1), t.c[0], 8192, false);
    }
    __device__ __forceinline__ void load(const uint8_t AS1 *p) { typedef u32x2_t __attribute__((aligned(1))) T; v = *(const T AS1*)p; }
    __device__ __forceinline__ void zero() { v = (u32x2_t)(0u); }
    __device__ __forceinline__ int px(int i) const { return (int)((v[i >> 2] >> (8 * (i & 3))) & 0xff); }
};
template<> struct RowLoad<uint16_t, 12>
{
    u32x4_t lo; u32x2_t hi;
    __device__ __forceinline__ void hsum(const McTaps<uint16_t> &t, int H[4]) const
    {
        const uint32_t D[6] = { lo[0], lo[1], lo[2], lo[3], hi[0], hi[1] };
        uint32_t A[5];
#pragma unroll
        for (int i = 0; i < 5; i++) A[i] = __builtin_amdgcn_alignbit(D[i + 1], D[i], 16);
        H[0] = mc_dot2(D[0], t.l[0], mc_dot2(D[1], t.l[1], mc_dot2(D[2], t.l[2], mc_dot2(D[3], t.l[3], 0))));
        H[1] = mc_dot2(A[0], t.l[0], mc_dot2(A[1], t.l[1], mc_dot2(A[2], t.l[2], mc_dot2(A[3], t.l[3], 0))));
        H[2] = mc_dot2(D[1], t.l[0], mc_dot2(D[2], t.l[1], mc_dot2(D[3], t.l[2], mc_dot2(D[4], t.l[3], 0))));
        H[3] = mc_dot2(A[1], t.l[0], mc_dot2(A[2], t.l[1], mc_dot2(A[3], t.l[2], mc_dot2(A[4], t.l[3], 0))));
    }
    __device__ __forceinline__ void load(const uint16_t AS1 *p)
    {
        typedef u32x4_t __attribute__((aligned(2))) T4;
        typedef u32x2_t __attribute__((aligned(2))) T2;
        lo = *(const T4 AS1*)p; hi = *(const T2 AS1*)(p + 8);
    }
    __device__ __forceinline__ void zero() { lo = (u32x4_t)(0u); hi = (u32x2_t)(0u); }
    __device__ __forceinline__ int px(int i) const { const uint32_t d = i < 8 ? lo[i >> 1] : hi[(i - 8) >> 1]; return (int)((d >> (16 * (i & 1))) & 0xffff); }
};
template<> struct RowLoad<uint16_t, 8>        /* six samples are enough for the chroma footprint */
{
    u32x3_t v;
    __device__ __forceinline__ void hsum(const McTaps<uint16_t> &t, int H[2]) const
    {
        H[0] = mc_dot2(v[0], t.c[0], mc_dot2(v[1], t.c[1], 0));
        H[1] = mc_dot2(__builtin_amdgcn_alignbit(v[1], v[0], 16), t.c[0], mc_dot2(__builtin_amdgcn_alignbit(v[2], v[1], 16), t.c[1], 0));
    }
    __device__ __forceinline__ void load(const uint16_t AS1 *p) { typedef u32x3_t __attribute__((aligned(2))) T; v = *(const T AS1*)p; }
    __device__ __forceinline__ void zero() { v = (u32x3_t)(0u); }
    __device__ __forceinline__ int px(int i) const { return (int)((v[i >> 1] >> (16 * (i & 1))) & 0xffff); }
};

/* Predictions of one tile from one list: luma 4x4, Cb and Cr 2x2 */
struct McTile
{
    int y[4][4], c[2][2][2];
};

/* xPredInterLumaBlk + xPredInterChromaBlk for one tile and one list (TComPrediction.cpp:405-593), pixel flavour or
 * (inter14) 14-bit flavour, chosen per lane. Every row the filters can touch is fetched first -- one wide load per row,
 * apron rows only where the vertical phase needs them -- so the loads of a tile are all in flight together; then the
 * separable filters run on registers. Phases and flavours differ between lanes; the reference's formulas
 *   no vertical phase:  pixel = clip((H + 32) >> 6) or a copy;       14-bit = (H + offH) >> shiftH or p2s
 *   vertical phase on samples (xf = 0): pixel = clip((V + 32) >> 6); 14-bit = (V + offH) >> shiftH
 *   vertical phase on intermediates:    pixel = clip((V + offV) >> shiftV); 14-bit = V >> 6
 * (each through an int16 cast; ipfilter.cpp:40-172, :194-435) differ only in offset, shift and clamp, which become
 * per-lane parameters of one code path. */
template<typename P, bool CHROMA>
__device__ __forceinline__ void mc_predict(const PlaneDev (&pls)[3], int lx, int ly, int cx, int cy, int mvx, int mvy, int depth, bool inter14,
                                           bool luma, bool chroma, McTile &o)
{
    const int maxv = (1 << depth) - 1, headRoom = 14 - depth;
    const int shiftH = 6 - headRoom, offH = -(8192 << shiftH);
    const int shiftV = 6 + headRoom, offV = (1 << (shiftV - 1)) + (8192 << 6);
    const int xf = mvx & 3, yf = mvy & 3, cxf = mvx & 7, cyf = mvy & 7;
    /* filter taps first: nothing below has to wait for them behind the row loads */
    McTaps<P> th;
    th.load(xf, cxf);
    int lv[8], kv[4];
#pragma unroll
    for (int k = 0; k < 8; k++) lv[k] = kLumaTaps[yf][k];
#pragma unroll
    for (int k = 0; k < 4; k++) kv[k] = kChromaTaps[cyf][k];
    RowLoad<P, 12> L[11];
    RowLoad<P, 8> C[2][5];
    {
        const PlaneDev pl = pls[0];
        const P AS1 *base = (const P AS1*)pl.origin + (long)(ly + (mvy >> 2) - 3) * pl.stride + lx + (mvx >> 2) - 3;
#pragma unroll
        for (int r = 0; r < 11; r++) L[r].zero();
        if (luma)
        {
#pragma unroll
            for (int r = 3; r < 7; r++) L[r].load(base + (long)r * pl.stride);
            if (yf)
            {
#pragma unroll
                for (int r = 0; r < 3; r++) L[r].load(base + (long)r * pl.stride);
#pragma unroll
                for (int r = 7; r < 11; r++) L[r].load(base + (long)r * pl.stride);
            }
        }
    }
#pragma unroll
    for (int c = 0; CHROMA && c < 2; c++)
    {
        const PlaneDev pl = pls[1 + c];
        const P AS1 *base = (const P AS1*)pl.origin + (long)(cy + (mvy >> 3) - 1) * pl.stride + cx + (mvx >> 3) - 1;
#pragma unroll
        for (int r = 0; r < 5; r++) C[c][r].zero();
        if (chroma)
        {
            C[c][1].load(base + pl.stride);
            C[c][2].load(base + 2L * pl.stride);
            if (cyf)
            {
                C[c][0].load(base);
                C[c][3].load(base + 3L * pl.stride);
                C[c][4].load(base + 4L * pl.stride);
            }
        }
    }
    /* Every stage is value = int16((sum + off) >> shift), then a clamp for finished samples; which flavour and which phases a
     * lane has only changes off, shift and the clamp bounds, so they are per-lane parameters and the code is one formula.
     * A zero horizontal phase runs through the phase-0 taps {.., 64, ..}: (64 p + 32) >> 6 = p, (64 p) >> 6 = p and
     * (64 p + offH) >> shiftH = (p << headRoom) - 8192 are the copy, the vertical filter's sample input and p2s exactly. */
    const int lo = inter14 ? -32768 : 0, hi = inter14 ? 32767 : maxv;
    const int offR = inter14 ? offH : 32, shR = inter14 ? shiftH : 6;            /* result of the horizontal stage alone */
    /* ---- luma ---- */
    {
        const int (&cv)[8] = lv;
        const int offI = xf ? offH : 0, shI = xf ? shiftH : 6;                  /* input of the vertical stage */
        const int offA = yf ? offI : offR, shA = yf ? shI : shR;
        const int offB = xf ? (inter14 ? 0 : offV) : offR, shB = xf ? (inter14 ? 6 : shiftV) : shR;   /* vertical stage */
        int im[11][4];          /* rows -3 .. 7 after the horizontal stage */
        auto hrow = [&](int r)
        {
            int Hs[4];
            L[r].hsum(th, Hs);
#pragma unroll
            for (int x = 0; x < 4; x++) im[r][x] = as_i16((Hs[x] + offA) >> shA);
        };
#pragma unroll
        for (int r = 3; r < 7; r++) hrow(r);
        if (yf)
        {
#pragma unroll
            for (int r = 0; r < 3; r++) hrow(r);
#pragma unroll
            for (int r = 7; r < 11; r++) hrow(r);
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++)
                {
                    int V = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) V += im[y + k][x] * cv[k];
                    const int v = as_i16((V + offB) >> shB);
                    o.y[y][x] = v < lo ? lo : (v > hi ? hi : v);
                }
        }
        else
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++) { const int v = im[y + 3][x]; o.y[y][x] = v < lo ? lo : (v > hi ? hi : v); }
    }
    /* ---- chroma ---- */
    if (CHROMA)
    {
        const int (&cv)[4] = kv;
        const int offI = cxf ? offH : 0, shI = cxf ? shiftH : 6;
        const int offA = cyf ? offI : offR, shA = cyf ? shI : shR;
        const int offB = cxf ? (inter14 ? 0 : offV) : offR, shB = cxf ? (inter14 ? 6 : shiftV) : shR;
#pragma unroll
        for (int c = 0; c < 2; c++)
        {
            int im[5][2];
            auto hrow = [&](int r)
            {
                int Hs[2];
                C[c][r].hsum(th, Hs);
#pragma unroll
                for (int x = 0; x < 2; x++) im[r][x] = as_i16((Hs[x] + offA) >> shA);
            };
            hrow(1); hrow(2);
            if (cyf)
            {
                hrow(0); hrow(3); hrow(4);
#pragma unroll
                for (int y = 0; y < 2; y++)
#pragma unroll
                    for (int x = 0; x < 2; x++)
                    {
                        int V = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) V += im[y + k][x] * cv[k];
                        const int v = as_i16((V + offB) >> shB);
                        o.c[c][y][x] = v < lo ? lo : (v > hi ? hi : v);
                    }
            }
            else
#pragma unroll
                for (int y = 0; y < 2; y++)
#pragma unroll
                    for (int x = 0; x < 2; x++) { const int v = im[y + 1][x]; o.c[c][y][x] = v < lo ? lo : (v > hi ? hi : v); }
        }
    }
}

template<typename P> struct PxStore;
template<> struct PxStore<uint8_t>
{
    static __device__ __forceinline__ void four(uint8_t AS1 *d, const int v[4])
    {
        typedef uint32_t __attribute__((aligned(1))) u32u;
        *(u32u AS1*)d = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
    }
    static __device__ __forceinline__ void two(uint8_t AS1 *d, int a, int b)
    {
        typedef uint16_t __attribute__((aligned(1))) u16u;
        *(u16u AS1*)d = (uint16_t)(a | (b << 8));
    }
};
template<> struct PxStore<uint16_t>
{
    static __device__ __forceinline__ void four(uint16_t AS1 *d, const int v[4])
    {
        typedef uint32_t __attribute__((aligned(2))) u32u;
        ((u32u AS1*)d)[0] = (uint32_t)v[0] | ((uint32_t)v[1] << 16);
        ((u32u AS1*)d)[1] = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
    }
    static __device__ __forceinline__ void two(uint16_t AS1 *d, int a, int b)
    {
        typedef uint32_t __attribute__((aligned(2))) u32u;
        *(u32u AS1*)d = (uint32_t)a | ((uint32_t)b << 16);
    }
};

/* Lane -> (PU, tile) for kernels whose waves take 64 consecutive tiles of a launch, whatever PUs they belong to.
 * first[j] = first tile of PU j (njobs + 64 entries, the tail repeating the tile total); chunk_job[k] = the PU that tile
 * 64 k belongs to. The lanes find their PU from the 64 entries of first[] after the chunk's PU: a flag in LDS where a PU
 * starts, one ballot, a bit count. Returns false for lanes past the last tile; with may_leave = false those lanes get the
 * last PU and tile 0, so that a kernel with cross-lane steps can keep the whole wave running. */
__device__ __forceinline__ bool flat_tile_of_lane(const int32_t *first, const int32_t *chunk_job, int nchunks, int njobs, int &j, int &t,
                                                  bool may_leave = true)
{
    __shared__ int32_t starts[64];
    __shared__ int32_t flags[64];
    const int k = xcd_remap(blockIdx.x, nchunks);
    if (k >= nchunks) { j = njobs - 1; t = 0; return false; }       /* never taken: the grid is nchunks */
    const int lane = threadIdx.x & 63;
    const int j0 = uni(chunk_job[k]), g = 64 * k + lane;
    const int my_start = first[j0 + lane];
    starts[lane] = my_start;
    flags[lane] = 0;
    wave_sync();
    /* PU j0 holds tile 64 k, so the PUs after it start inside (64 k, ...) */
    if (lane > 0 && (unsigned)(my_start - 64 * k) < 64u) flags[my_start - 64 * k] = 1;
    wave_sync();
    const int f = flags[lane];
    const unsigned long long m = __ballot(f);
    const int off = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0)) + f;
    j = j0 + off;
    if (j >= njobs)                           /* tiles past the last PU (only in the last wave) */
    {
        j = njobs - 1; t = 0;
        return false;
    }
    t = g - starts[off];
    return true;
}

/* 125 VGPRs (8-bit) / 160 (16-bit): four / three waves per SIMD. Asking for more (amdgpu_waves_per_eu) makes the
 * compiler spill the row registers and was measured slower. */
template<typename P>
__global__ __launch_bounds__(64) void mc_kernel(McArgs a)
{
    int j, t;
    if (!flat_tile_of_lane(a.first, a.chunk_job, a.nchunks, a.njobs, j, t)) return;
    const x265amd_mc_job *job = a.jobs + j;
    const int w = job->w, dir = job->dir, flags_lc = job->flags;
    const int pux = job->pu_x, puy = job->pu_y;
    int m0x = job->mv0[0], m0y = job->mv0[1], m1x = job->mv1[0], m1y = job->mv1[1];
    if (a.res0) { m0x = a.res0[j].mv[0]; m0y = a.res0[j].mv[1]; }
    if (a.res1) { m1x = a.res1[j].mv[0]; m1y = a.res1[j].mv[1]; }
    const McRef *p0 = a.refs + job->ref0, *p1 = a.refs + job->ref1;
    const int wt = w >> 2;
    const bool weighted = (flags_lc & 4) != 0, two = dir == 3, inter14 = two || weighted;
    const int maxv = (1 << a.depth) - 1;
    const int ty = t / wt, tx = t - ty * wt;
    const int lx = pux + 4 * tx, ly = puy + 4 * ty, cx = (pux >> 1) + 2 * tx, cy = (puy >> 1) + 2 * ty;
    const bool luma = flags_lc & 1, chroma = flags_lc & 2;
    /* list A = the only list of a uni-predicted PU, or list 0; list B = list 1 of a bi-predicted PU */
    const McRef *pa = dir == 2 ? p1 : p0;
    McTile A, B;
    {
        /* plane descriptors before anything that waits on the row loads */
        const PlaneDev pla[3] = { pa->pl[0], pa->pl[1], pa->pl[2] };
        mc_predict<P, true>(pla, lx, ly, cx, cy, dir == 2 ? m1x : m0x, dir == 2 ? m1y : m0y, a.depth, inter14, luma, chroma, A);
    }
    if (two)
    {
        const PlaneDev plb[3] = { p1->pl[0], p1->pl[1], p1->pl[2] };
        mc_predict<P, true>(plb, lx, ly, cx, cy, m1x, m1y, a.depth, true, luma, chroma, B);
    }
    if (inter14)
    {
        /* TComYuv::addAvg (TComYuv.cpp:596-606, :616-650) as weights (1, 1), or the explicit weights */
        const int shiftNum = 15 - a.depth;
#pragma unroll
        for (int c = 0; c < 3; c++)
        {
            McWeights W;
            if (weighted) W = mc_weights(pa->wp, p1->wp, c, two, a.depth);
            else { W.w0 = W.w1 = 1; W.shift = shiftNum; W.round = 1 << (shiftNum - 1); W.offset = 0; W.bias = 8192; }
            if (c == 0)
            {
#pragma unroll
                for (int yy = 0; yy < 4; yy++)
#pragma unroll
                    for (int xx = 0; xx < 4; xx++) A.y[yy][xx] = W.apply(A.y[yy][xx], B.y[yy][xx], two, maxv);
            }
            else
            {
#pragma unroll
                for (int yy = 0; yy < 2; yy++)
#pragma unroll
                    for (int xx = 0; xx < 2; xx++) A.c[c - 1][yy][xx] = W.apply(A.c[c - 1][yy][xx], B.c[c - 1][yy][xx], two, maxv);
            }
        }
    }
    if (luma)
    {
        P AS1 *d = (P AS1*)a.pred[0].origin + (long)ly * a.pred[0].stride + lx;
#pragma unroll
        for (int yy = 0; yy < 4; yy++) PxStore<P>::four(d + (long)yy * a.pred[0].stride, A.y[yy]);
    }
    if (chroma)
#pragma unroll
        for (int c = 0; c < 2; c++)
        {
            P AS1 *d = (P AS1*)a.pred[1 + c].origin + (long)cy * a.pred[1 + c].stride + cx;
            PxStore<P>::two(d, A.c[c][0][0], A.c[c][0][1]);
            PxStore<P>::two(d + a.pred[1 + c].stride, A.c[c][1][0], A.c[c][1][1]);
        }
}

/* The bi-prediction estimate with 64 consecutive tiles per wave (flat_tile_of_lane): every lane adds its tile's two
 * SATDs to its PU's result, which the launch zeroed first. SATD is halved per tile, which equals halving the sum. */
struct BipredFlatArgs
{
    BipredArgs b;
    const int32_t *first, *chunk_job;
    int nchunks;
    const x265amd_result *res0, *res1;   /* optional: search results that supply the MVs of list 0 / list 1 */
};

template<typename P>
__global__ __launch_bounds__(64) void bipred_flat_kernel(BipredFlatArgs fa)
{
    const BipredArgs &a = fa.b;
    int j, t;
    if (!flat_tile_of_lane(fa.first, fa.chunk_job, fa.nchunks, a.njobs, j, t)) return;
    const x265amd_bipred_job *job = a.jobs + j;
    const int wt = job->w >> 2, pux = job->pu_x, puy = job->pu_y;
    const int ty = t / wt, tx = t - ty * wt;
    const long fo = (long)(puy + 4 * ty) * a.fenc.stride + pux + 4 * tx;
    const long r0o = (long)(puy + 4 * ty) * a.ref0.stride + pux + 4 * tx, r1o = (long)(puy + 4 * ty) * a.ref1.stride + pux + 4 * tx;
    typename Px<P>::quad f[4], z[4];
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        f[y] = Px<P>::load_g((const P AS1*)a.fenc.origin + fo + (long)y * a.fenc.stride);
        z[y] = Px<P>::avg(Px<P>::load_g((const P AS1*)a.ref0.origin + r0o + (long)y * a.ref0.stride),
                          Px<P>::load_g((const P AS1*)a.ref1.origin + r1o + (long)y * a.ref1.stride));
    }
    const int zero = tile_cost_quads<P>(f, z, true);
    int m0x = job->mv0[0], m0y = job->mv0[1], m1x = job->mv1[0], m1y = job->mv1[1];
    if (fa.res0) { m0x = fa.res0[j].mv[0]; m0y = fa.res0[j].mv[1]; }
    if (fa.res1) { m1x = fa.res1[j].mv[0]; m1y = fa.res1[j].mv[1]; }
    /* xPredInterLumaBlk's pixel flavour through the one-formula predictor: lanes of different phases do not diverge */
    McTile A, B;
    {
        const PlaneDev p0[3] = { a.ref0, a.ref0, a.ref0 }, p1[3] = { a.ref1, a.ref1, a.ref1 };
        mc_predict<P, false>(p0, pux + 4 * tx, puy + 4 * ty, 0, 0, m0x, m0y, a.depth, false, true, false, A);
        mc_predict<P, false>(p1, pux + 4 * tx, puy + 4 * ty, 0, 0, m1x, m1y, a.depth, false, true, false, B);
    }
    int (&o0)[4][4] = A.y;
#pragma unroll
    for (int y = 0; y < 4; y++)
#pragma unroll
        for (int x = 0; x < 4; x++) o0[y][x] = (A.y[y][x] + B.y[y][x] + 1) >> 1;
    int bi = tile_cost_px<P>(f, o0, true), zr = zero;
    /* the lanes of one PU are neighbours: sum them down to the PU's first lane in the wave (a lane adds its partner's
     * partial sum when the partner belongs to the same PU), then one atomic per PU and wave */
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
        const int pj = __shfl_down(j, d), pb = __shfl_down(bi, d), pz = __shfl_down(zr, d);
        const bool same = lane + d < 64 && pj == j;
        bi += same ? pb : 0;
        zr += same ? pz : 0;
    }
    const int prev = __shfl_up(j, 1);
    if (lane == 0 || prev != j)
    {
        atomicAdd(&a.results[j].satd, bi);
        atomicAdd(&a.results[j].satd_zero, zr);
    }
}

/* The merge-candidate estimate with 64 consecutive tiles per wave. The plan starts every CU at a multiple of 4 tiles
 * (8x8 CUs) or 16 tiles (larger ones), so a quad of lanes is still an 8x8 block and sixteen lanes a 16x16 block, as the
 * sa8d butterflies across lanes need; lanes in the padding between two CUs compute a dummy tile and add nothing. The
 * launch zeroes the results; per CU and wave the block sums are added up by neighbours and one lane adds them in. */
struct MergeFlatArgs
{
    MergeArgs m;
    const int32_t *first, *chunk_job;
    int nchunks;
};

template<typename P>
__global__ __launch_bounds__(64) void merge_flat_kernel(MergeFlatArgs fa)
{
    const MergeArgs &a = fa.m;
    const int lane = threadIdx.x & 63;
    int j, t;
    bool live = flat_tile_of_lane(fa.first, fa.chunk_job, fa.nchunks, a.njobs, j, t, false);
    const x265amd_merge_job *job = a.jobs + j;
    const int size = job->size, dir = job->dir, pux = job->pu_x, puy = job->pu_y;
    const int m0x = job->mv0[0], m0y = job->mv0[1], m1x = job->mv1[0], m1y = job->mv1[1];
    const PlaneDev r0 = a.refs[job->ref0], r1 = a.refs[job->ref1];
    const int T = (size >> 2) * (size >> 2), g16 = size >> 4;
    live = live && t < T;
    const int maxv = (1 << a.depth) - 1;
    /* tile number -> tile: bits 0-1 the tile inside its 8x8 block, bits 2-3 the 8x8 block inside its 16x16 block */
    const int tt = live ? t : 0, g = tt >> 4;
    const int gy = g16 ? g / g16 : 0, gx = g16 ? g - gy * g16 : 0;
    const int ty = gy * 4 + ((tt >> 3) & 1) * 2 + ((tt >> 1) & 1), tx = gx * 4 + ((tt >> 2) & 1) * 2 + (tt & 1);
    const long fo = (long)(puy + 4 * ty) * a.fenc.stride + pux + 4 * tx;
    const long o0 = (long)(puy + 4 * ty) * r0.stride + pux + 4 * tx, o1 = (long)(puy + 4 * ty) * r1.stride + pux + 4 * tx;
    /* list A = the only list of a uni-predicted candidate, or list 0; list B = list 1 of a bi-predicted one */
    McTile A, B;
    {
        const bool two = dir == 3;
        const PlaneDev ra = dir == 2 ? r1 : r0;
        const PlaneDev pa[3] = { ra, ra, ra }, pb[3] = { r1, r1, r1 };
        const int max_ = dir == 2 ? m1x : m0x, may_ = dir == 2 ? m1y : m0y;
        if (!two && ((max_ | may_) & 3) == 0)
            /* a whole-sample MV of a single list reads the block itself and nothing around it: the host lets such a
             * candidate use a reference without margins (the distortion of a finished prediction picture) */
            bipred_tile<P>((const P AS1*)ra.origin + (long)(puy + 4 * ty) * ra.stride + pux + 4 * tx, ra.stride, max_, may_, a.depth, A.y);
        else
            mc_predict<P, false>(pa, pux + 4 * tx, puy + 4 * ty, 0, 0, max_, may_, a.depth, two, true, false, A);
        if (two)
        {
            mc_predict<P, false>(pb, pux + 4 * tx, puy + 4 * ty, 0, 0, m1x, m1y, a.depth, true, true, false, B);
            /* TComYuv::addAvg, TComYuv.cpp:596-606 */
            const int shiftNum = 15 - a.depth, offset = (1 << (shiftNum - 1)) + 2 * 8192;
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++)
                {
                    const int v = (A.y[y][x] + B.y[y][x] + offset) >> shiftNum;
                    A.y[y][x] = v < 0 ? 0 : (v > maxv ? maxv : v);
                }
        }
    }
    int (&p)[4][4] = A.y;
    int d[4][4];
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        int fv[4];
        Px<P>::unpack(Px<P>::load_g((const P AS1*)a.fenc.origin + fo + (long)y * a.fenc.stride), fv);
#pragma unroll
        for (int x = 0; x < 4; x++) d[y][x] = fv[x] - p[y][x];
    }
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        const int a0 = d[y][0] + d[y][1], a1 = d[y][0] - d[y][1], a2 = d[y][2] + d[y][3], a3 = d[y][2] - d[y][3];
        d[y][0] = a0 + a2; d[y][1] = a1 + a3; d[y][2] = a0 - a2; d[y][3] = a1 - a3;
    }
    int s = 0;
    const int n1 = -(lane & 1), n2 = -((lane >> 1) & 1);
#pragma unroll
    for (int x = 0; x < 4; x++)
    {
        const int a0 = d[0][x] + d[1][x], a1 = d[0][x] - d[1][x], a2 = d[2][x] + d[3][x], a3 = d[2][x] - d[3][x];
        int c[4] = { a0 + a2, a1 + a3, a0 - a2, a1 - a3 };
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            int v = c[i];
            v = ((v ^ n1) - n1) + dpp_all<0xB1>(v);
            v = ((v ^ n2) - n2) + dpp_all<0x4E>(v);
            s += abs(v);
        }
    }
    s = live ? s : 0;
    /* sa8d_8x8 rounds per 8x8 block, sa8d_16x16 once per 16x16 block (pixel.cpp:290-342) */
    const int s4 = seg_sum(s, 2), s16 = seg_sum(s, 4);
    int total = g16 ? ((live && (lane & 15) == 0) ? (s16 + 2) >> 2 : 0) : ((live && (lane & 3) == 0) ? (s4 + 2) >> 2 : 0);
    const int key = live ? j : -1 - lane;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1)
    {
        const int pk = __shfl_down(key, dd), pv = __shfl_down(total, dd);
        total += (lane + dd < 64 && pk == key) ? pv : 0;
    }
    const int prev = __shfl_up(key, 1);
    if (live && (lane == 0 || prev != key)) atomicAdd(&a.out[j], total);
}

/* subpelCompare (motion.cpp:1651-1689) for a batch of PUs with 64 consecutive tiles per wave: the PU interpolated at
 * its quarter-pel MV (pixel flavour of the one-formula predictor; a whole-sample MV is the copy the reference compares
 * directly) against the source, SAD or SATD per tile, summed per PU as in the bi-prediction estimate. */
struct SubpelFlatArgs
{
    PlaneDev fenc, ref;
    int depth, njobs, satd;
    const x265amd_job *jobs;
    x265amd_result *results;   /* zeroed by the launch */
    const int32_t *first, *chunk_job;
    int nchunks;
};

template<typename P>
__global__ __launch_bounds__(64) void subpel_flat_kernel(SubpelFlatArgs a)
{
    int j, t;
    if (!flat_tile_of_lane(a.first, a.chunk_job, a.nchunks, a.njobs, j, t)) return;
    const x265amd_job *job = a.jobs + j;
    const int wt = job->w >> 2, ty = t / wt, tx = t - ty * wt;
    const int x = job->pu_x + 4 * tx, y = job->pu_y + 4 * ty, qx = job->qmvp[0], qy = job->qmvp[1];
    typename Px<P>::quad f[4];
#pragma unroll
    for (int r = 0; r < 4; r++) f[r] = Px<P>::load_g((const P AS1*)a.fenc.origin + (long)(y + r) * a.fenc.stride + x);
    McTile A;
    const PlaneDev pl[3] = { a.ref, a.ref, a.ref };
    mc_predict<P, false>(pl, x, y, 0, 0, qx, qy, a.depth, false, true, false, A);
    int v = tile_cost_px<P>(f, A.y, a.satd != 0);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
        const int pj = __shfl_down(j, d), pv = __shfl_down(v, d);
        v += (lane + d < 64 && pj == j) ? pv : 0;
    }
    const int prev = __shfl_up(j, 1);
    if (lane == 0 || prev != j)
    {
        atomicAdd(&a.results[j].cost, v);
        a.results[j].mv[0] = (int16_t)qx; a.results[j].mv[1] = (int16_t)qy;
        a.results[j].subpel_calls = 1;
    }
}

} // namespace x265amd







#endif
