/* Lookahead frame cost on the device: what EstimateRow::estimateCUCost computes for every
 * 8x8 lowres CU besides the motion search, and the sums CostEstimate::estimateFrameCost
 * forms from it (reference: source/encoder/slicetype.cpp:1471-1581, :1193-1204).
 *
 * None of it depends on a neighbouring CU, so unlike the search (an anti-diagonal
 * wavefront) these kernels are flat over the CUs of all frames of a batch:
 *   la_cost_init_kernel   clears the accumulators of a frame
 *   la_intra_kernel       35-mode intra SATD of a CU: one wave per CU, one lane per
 *                         (mode, 4x4 tile), prediction and Hadamard in registers
 *   la_cost_kernel        bidirectional candidates, list / intra choice, row and frame
 *                         sums: four lanes per CU (one per 4x4 tile)
 * Sums are integer, so atomic accumulation gives the reference's totals in any order.
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

__global__ void la_cost_init_kernel(const FrameCostDev *frames, int hcu)
{
    const FrameCostDev &f = frames[blockIdx.x];
    for (int i = threadIdx.x; i < hcu; i += blockDim.x)
    {
        if (!f.intra_calculated) f.row_satds_intra[i] = 0;
        if (f.d0 + f.d1 > 0) f.row_satds[i] = 0;
    }
    if (threadIdx.x < 5) f.totals[threadIdx.x] = 0;
}

/* ---- intra: slicetype.cpp:1491-1559 ----
 * Per wave, LDS holds the block's reference edges the way the reference builds them
 * (:1495-1524): edge[f][s][0..16], f = 0 raw / 1 smoothed with [1 2 1] (ends kept),
 * s = 0 above / 1 left, index 0 the corner, 9..16 a repeat of index 8; and the 8x8 source. */
#define LA_INTRA_WAVES 4
#define LA_EDGE_BYTES 20
#define LA_INTRA_LDS (4 * LA_EDGE_BYTES + 64)

/* Sample (row k, column l) of prediction `mode` as the reference's cost loop sees it: 0 = DC
 * with edge smoothing (intrapred.cpp:45-104), 1 = planar from the smoothed edges (:106-154),
 * 2..34 angular (:156-259) with modes below 18 left transposed (:262-289). */
__device__ __forceinline__ int la_pred(const uint8_t *e, int mode, int k, int l, int dc)
{
    /* e + (f * 2 + s) * LA_EDGE_BYTES */
    if (mode == 0)
    {
        const uint8_t *a = e, *lf = e + LA_EDGE_BYTES;
        if ((k | l) == 0) return (a[1] + lf[1] + 2 * dc + 2) >> 2;
        if (k == 0) return (a[l + 1] + 3 * dc + 2) >> 2;
        if (l == 0) return (lf[k + 1] + 3 * dc + 2) >> 2;
        return dc;
    }
    if (mode == 1)
    {
        const uint8_t *a = e + 2 * LA_EDGE_BYTES + 1, *lf = e + 3 * LA_EDGE_BYTES + 1;
        return ((lf[k] << 3) + 8 + (l + 1) * ((int)a[8] - lf[k]) + (a[l] << 3) + (k + 1) * ((int)lf[8] - a[l])) >> 4;
    }
    /* angle steps of |mode - 10| / |mode - 26| = 0..8 and 8192 / step (intrapred.cpp:167-170), packed */
    const int f = (mode == 2 || mode == 18 || mode == 34);   /* IntraFilterType of 8x8 blocks, intrapred.cpp:35 */
    const int ver = mode >= 18;
    const uint8_t *refMain = e + (f * 2 + (ver ? 0 : 1)) * LA_EDGE_BYTES;
    const uint8_t *refSide = e + (f * 2 + (ver ? 1 : 0)) * LA_EDGE_BYTES;
    int ang = ver ? mode - 26 : 10 - mode;
    const int idx = ang < 0 ? -ang : ang;
    const unsigned long long angs = 0x1a15110d09050200ull;             /* 0 2 5 9 13 17 21 26, a byte each; step 8 = 32 */
    const int absAng = (idx == 8) ? 32 : (int)((angs >> (8 * idx)) & 0xff);
    if (absAng == 0)
    {
        int v = refMain[l + 1];
        if (l == 0)
        {
            v += ((int)refSide[k + 1] - (int)refSide[0]) >> 1;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
        return v;
    }
    int invAngle = 256;
    switch (idx) { case 1: invAngle = 4096; break; case 2: invAngle = 1638; break; case 3: invAngle = 910; break;
                   case 4: invAngle = 630; break; case 5: invAngle = 482; break; case 6: invAngle = 390; break;
                   case 7: invAngle = 315; break; default: break; }
    ang = ang < 0 ? -absAng : absAng;
    const int deltaPos = (k + 1) * ang, deltaInt = deltaPos >> 5, fract = deltaPos & 31;
    const int i0 = l + deltaInt + 1, i1 = i0 + 1;
    /* negative indices of the main edge are projections of the side edge (intrapred.cpp:181-195) */
    const int v0 = i0 >= 0 ? refMain[i0] : refSide[(128 - i0 * invAngle) >> 8];
    if (!fract) return v0;
    const int v1 = i1 >= 0 ? refMain[i1] : refSide[(128 - i1 * invAngle) >> 8];
    return ((32 - fract) * v0 + fract * v1 + 16) >> 5;
}

__global__ __launch_bounds__(LA_INTRA_WAVES * 64) void la_intra_kernel(const FrameCostDev *frames, int W, int H)
{
    __shared__ uint8_t lds[LA_INTRA_WAVES][LA_INTRA_LDS];
    const FrameCostDev &f = frames[blockIdx.y];
    if (f.intra_calculated) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cu = blockIdx.x * LA_INTRA_WAVES + wave;
    if (cu >= W * H) return;                                   /* wave-uniform; no block barrier below */
    const int cuy = cu / W, cux = cu - cuy * W;
    const long stride = f.fenc.stride;
    const uint8_t AS1 *cur = (const uint8_t AS1*)f.fenc.origin + (long)(8 * cuy) * stride + 8 * cux;
    uint8_t *e = lds[wave], *src = e + 4 * LA_EDGE_BYTES;
    /* raw edges: lanes 0-16 above, 32-48 left */
    {
        const int i = lane & 31, s = lane >> 5, j = i < 8 ? i : 8;
        if (i <= 16)
            e[s * LA_EDGE_BYTES + i] = s ? cur[-1 - stride + j * stride] : cur[-1 - stride + j];
        src[lane] = cur[(lane >> 3) * stride + (lane & 7)];
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
    int dc = 8;
#pragma unroll
    for (int i = 1; i <= 8; i++) dc += e[i] + e[LA_EDGE_BYTES + i];
    dc >>= 4;
    int icost = LA_COST_MAX;
    for (int pass = 0; pass < 3; pass++)
    {
        const int task = pass * 64 + lane, mode = task >> 2, ty = (task >> 1) & 1, tx = task & 1;
        int v = LA_COST_MAX;
        if (mode < 35)
        {
            const bool hor = mode >= 2 && mode < 18;           /* compared with the transposed source, :1542-1543 */
            int d[4][4];
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int l = 0; l < 4; l++)
                {
                    const int r = 4 * ty + k, c = 4 * tx + l;
                    d[k][l] = (int)(hor ? src[c * 8 + r] : src[r * 8 + c]) - la_pred(e, mode, r, c, dc);
                }
            v = had4x4_half(d);
        }
        v = seg_sum(mode < 35 ? v : 0, 2);
        v = mode < 35 ? v : LA_COST_MAX;
        v = wave_min(v);
        icost = v < icost ? v : icost;
    }
    if (lane == 0)
    {
        f.intra_cost[cu] = icost;
        atomicAdd(&f.row_satds_intra[cuy], icost);
        if (la_scored(cux, cuy, W, H))
        {
            atomicAdd((unsigned long long*)&f.totals[2], (unsigned long long)icost);
            if (f.inv_qscale)
                atomicAdd((unsigned long long*)&f.totals[3], (unsigned long long)(long long)((int)((unsigned)icost * (unsigned)f.inv_qscale[cu] + 128u) >> 8));
        }
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
    const int ncu = W * H;
    const int cu = blockIdx.x * (LA_COST_THREADS / 4) + (threadIdx.x >> 2), tile = threadIdx.x & 3;
    const bool live = cu < ncu;
    const int cuc = live ? cu : 0;
    const int cuy = cuc / W, cux = cuc - cuy * W;
    const bool bidir = f.d1 > 0, inter = f.d0 + f.d1 > 0;
    int bcost = LA_COST_MAX, listused = 0;
    if (inter)
    {
        const int c0 = f.mv_costs0[cuc];
        if (c0 < bcost) { bcost = c0; listused = 1; }
    }
    if (bidir)
    {
        const int c1 = f.mv_costs1[cuc];
        if (c1 < bcost) { bcost = c1; listused = 2; }
        const long off = (long)(8 * cuy + 4 * (tile >> 1)) * f.fenc.stride + 8 * cux + 4 * (tile & 1);
        const int m0x = f.mvs0[2 * cuc], m0y = f.mvs0[2 * cuc + 1], m1x = f.mvs1[2 * cuc], m1y = f.mvs1[2 * cuc + 1];
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
    if (!live || tile) return;
    const bool scored = la_scored(cux, cuy, W, H);
    if (!bidir)
    {
        const int ic = f.intra_cost[cu];
        if (ic < bcost)
        {
            if (scored) atomicAdd((unsigned long long*)&f.totals[4], 1ull);
            bcost = ic;
            listused = 0;
        }
    }
    if (inter)
    {
        atomicAdd(&f.row_satds[cuy], bcost);
        if (scored)
        {
            atomicAdd((unsigned long long*)&f.totals[0], (unsigned long long)bcost);
            if (f.inv_qscale)
                atomicAdd((unsigned long long*)&f.totals[1], (unsigned long long)(long long)((int)((unsigned)bcost * (unsigned)f.inv_qscale[cu] + 128u) >> 8));
        }
    }
    f.lowres_costs[cu] = (uint16_t)((bcost < 16383 ? bcost : 16383) | (listused << 14));
}

} // namespace x265amd
#endif
