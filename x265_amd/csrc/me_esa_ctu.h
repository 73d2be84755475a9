/* Exhaustive integer search (X265_FULL_SEARCH, source/encoder/motion.cpp:1021-1056) of all 85 square PUs of a 64x64
 * CTU at once.
 *
 * The SAD of a PU at a displacement is the sum of the SADs of its 8x8 blocks at that displacement, so the four quadtree
 * depths of a CTU (TEncCu::xCompressCU, Lib/TLibEncoder/TEncCu.cpp:2352, searches every one of them) need the 8x8 level
 * measured once: the reference's own hardware model does the same for its refinement windows
 * (source/hardwareC/inter.cpp:2521-2624). One workgroup takes a CTU. It scans the bounding box of the PUs' search
 * ranges in absolute displacements (the ranges of the PUs of a CTU differ where their predictors do): strips of the
 * box's reference window go through LDS, and per task a wave measures a patch of 4 x 8 displacements with a lane per
 * 8x8 block (8-bit samples: v_qsad_pk_u16_u8 as in esa_patch, the block's source samples in registers; deeper samples:
 * 2 x 8 displacements with v_sad_u16, esa_patch16, sums 32 bits wide throughout). The lanes are in Morton
 * order, so a 16x16 PU is a quad of lanes and a 32x32 PU a row of sixteen: the packed 16-bit sums go up to 16x16 by
 * two DPP adds per dword (16 x 16 x 255 still fits 16 bits), then every lane of a quad keeps a quarter of the patch,
 * widens it, and two row rotations give the 32x32 sums; of those every lane keeps a pair of displacements and two
 * lane-swap adds give the 64x64 sums. Every level folds its sums with its own PU's lambda * bits -- per-PU tables of
 * (cost << 5 | index in the patch) per column and per row of the strip, a sentinel outside the PU's own range -- into
 * the lane's running minimum of (cost, row, column), which is the position the reference's strict-'<' raster scan of
 * that PU keeps. esa_finish_kernel and the refinement kernels then run per PU size as after esa_scan_kernel.
 *
 * Job order of a launch of n CTUs: the 64 n 8x8 PUs (CTU-major, Morton order inside a CTU), then the 16 n 16x16 PUs
 * (likewise), the 4 n 32x32 PUs (raster order inside a CTU = Morton order), the n 64x64 PUs. A slot whose w is 0 is a
 * PU that is not searched (CTUs cut by the picture's edge); slot 0 of a CTU's 8x8 PUs always carries the CTU's origin
 * and, in a launch over several (source, reference) pairs, the CTU's pair (job.reserved).
 */
#ifndef X265AMD_ME_ESA_CTU_H
#define X265AMD_ME_ESA_CTU_H

#include "me_esa.h"

namespace x265amd {

#define ECTU_THREADS 512
#define ECTU_WAVES (ECTU_THREADS / 64)
#define ECTU_PUS 85
#define ECTU_YG 4                               /* groups of 8 rows per strip at most */
#define ECTU_KYP (ECTU_YG * 8 + 4)
#define ECTU_WROWS (ECTU_YG * 8 + 63)
#define ECTU_L_HDR 0
#define ECTU_L_PU 64                            /* per PU 8 ints: mnx, mxx, mny, mxy, qmx, qmy, present, job index */
#define ECTU_L_KX (ECTU_L_PU + ECTU_PUS * 32)
#define ECTU_L_RED ECTU_L_KX                    /* the waves' minima, over the column tables once the scan is over */
#define ECTU_MAX_SPAN 1024                      /* columns / rows of the bounding box at most (10 bits each in a lane's index) */

/* What depends on the sample size. 8-bit: a lane measures 4 columns x 8 rows per task with the packed quad-SAD and keeps
 * its block's source samples in registers. 16-bit (samples of up to 12 bits): 2 columns x 8 rows with v_sad_u16
 * (esa_patch16), the CTU's source samples in LDS, sums 32 bits wide from the start. */
template<typename P> struct Ectu
{
    static constexpr int BPS = (int)sizeof(P);
    static constexpr int CX = BPS == 1 ? 4 : 2, CXL = BPS == 1 ? 2 : 1;     /* columns of a lane's patch, and their log2 */
    static constexpr int XG = BPS == 1 ? 24 : 32;                          /* groups of CX columns per strip at most */
    static constexpr int XS = XG * CX;                                     /* = 96 / 64 columns */
    /* dwords per PU of the column / row tables: a multiple of 4 that is no multiple of 8, so that the 64 lanes of a
     * 16-byte read, one PU each, spread over all banks */
    static constexpr int KXP = XS + 4;
    /* bytes per window row: the strip's columns + 64 of the CTU. 8-bit: eight rows (one block row) a multiple of 16
     * dwords that is no multiple of 32 -- lanes 0 .. 31 (block rows 0 .. 3, two dwords per block column) then hit every
     * bank once; 16-bit: a multiple of 32 dwords that is no multiple of 64 (two lanes per bank, the least a block of
     * four dwords allows) */
    static constexpr int PITCH = BPS == 1 ? 168 : 272;
    static constexpr int KSH = 3 + CXL;                                    /* key = cost << KSH | row of the patch << CXL | column */
    static constexpr uint32_t SENT = EsaLane<P>::SENT;                     /* added to the cost of a position outside a PU's range */
    static constexpr int L_KY = ECTU_L_KX + ECTU_PUS * KXP * 4;
    static constexpr int L_WIN = L_KY + ECTU_PUS * ECTU_KYP * 4;
    static constexpr int L_FENC = L_WIN + ECTU_WROWS * PITCH;              /* 16-bit: the CTU's source samples, 128 bytes a row */
    static constexpr int L_TOTAL = L_FENC + (BPS == 1 ? 0 : 64 * 128);
};

enum { EH_BX0 = 0, EH_BX1, EH_BY0, EH_BY1, EH_GX0, EH_GX1, EH_GY0, EH_GY1, EH_SX1, EH_SY1, EH_ANY };

/* where PU slot p of a CTU lies in it and how large it is */
__host__ __device__ inline void ectu_slot(int p, int &ox, int &oy, int &s)
{
    int k = p, sh = 3;
    s = 8;
    if (p >= 84) { k = 0; s = 64; sh = 6; }
    else if (p >= 80) { k = p - 80; s = 32; sh = 5; }
    else if (p >= 64) { k = p - 64; s = 16; sh = 4; }
    const int mx = (k & 1) | ((k >> 1) & 2) | ((k >> 2) & 4), my = ((k >> 1) & 1) | ((k >> 2) & 2) | ((k >> 3) & 4);
    ox = mx << sh; oy = my << sh;
}

/* index of slot p of CTU c in the job list of a launch of n CTUs */
__host__ __device__ inline int ectu_job(int p, int c, int n)
{
    if (p < 64) return c * 64 + p;
    if (p < 80) return 64 * n + c * 16 + (p - 64);
    if (p < 84) return 80 * n + c * 4 + (p - 80);
    return 84 * n + c;
}

__device__ __forceinline__ uint32_t ectu_mad_lo(uint32_t a, uint32_t k)
{
    uint32_t r;
    asm("v_mad_u32_u16 %0, %1, 32, %2" : "=v"(r) : "v"(a), "v"(k));
    return r;
}
__device__ __forceinline__ uint32_t ectu_mad_hi(uint32_t a, uint32_t k)
{
    uint32_t r;
    asm("v_mad_u32_u16 %0, %1, 32, %2 op_sel:[1,0,0,0]" : "=v"(r) : "v"(a), "v"(k));
    return r;
}
__device__ __forceinline__ uint32_t ectu_min(uint32_t a, uint32_t b) { return a < b ? a : b; }
/* b where the mask is all ones, a where it is zero: one v_bfi_b32 (written as a choice between array elements by a
 * lane-dependent condition the compiler puts the array into scratch memory and indexes it) */
__device__ __forceinline__ uint32_t ectu_pick(uint32_t m, uint32_t a, uint32_t b) { return (b & m) | (a & ~m); }

/* a level's best key of one patch into the lane's running minimum: key = cost << (3 + CXL) | row of the patch << CXL |
 * column; base = first row of the patch << 10 | its first column */
template<int CXL>
__device__ __forceinline__ void ectu_update(uint32_t &bc, uint32_t &bi, uint32_t key, uint32_t base)
{
    const uint32_t cost = key >> (3 + CXL), ri = base + (((key >> CXL) & 7u) << 10) + (key & ((1u << CXL) - 1u));
    const unsigned long long k = ((unsigned long long)cost << 32) | ri, b = ((unsigned long long)bc << 32) | bi;
    const bool better = k < b;
    bc = better ? cost : bc; bi = better ? ri : bi;
}

/* minimum of (cost, index) over the lanes whose numbers differ in the bits of `mask` */
__device__ __forceinline__ void ectu_lane_min(uint32_t &bc, uint32_t &bi, int mask)
{
    for (int o = 1; o < 64; o <<= 1)
    {
        if (!(o & mask)) continue;
        const uint32_t oc = (uint32_t)__shfl_xor((int)bc, o), oi = (uint32_t)__shfl_xor((int)bi, o);
        if (oc < bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
    }
}

template<typename P>
__global__ __launch_bounds__(ECTU_THREADS) __attribute__((amdgpu_waves_per_eu(4))) void esa_ctu_kernel(SearchArgs a, int nctu)
{
    typedef Ectu<P> E;
    constexpr int CX = E::CX, CXL = E::CXL, BPS = E::BPS, KSH = E::KSH, KXP = E::KXP, PITCH = E::PITCH;
    const int c = xcd_remap(blockIdx.x, nctu);
    if (c >= nctu) return;
    const int tid = threadIdx.x, wave = uni(tid >> 6), lane = tid & 63;     /* the wave's number as a scalar: so are its tasks' places */
    int *hdr = lds_i(0, ECTU_L_HDR);
    if (tid == 0)
    {
        hdr[EH_BX0] = hdr[EH_BY0] = hdr[EH_GX0] = hdr[EH_GY0] = 0x7fffffff;
        hdr[EH_BX1] = hdr[EH_BY1] = hdr[EH_GX1] = hdr[EH_GY1] = hdr[EH_SX1] = hdr[EH_SY1] = -0x7fffffff;
        hdr[EH_ANY] = 0;
    }
    /* the CTU's origin: slot 0 carries it whether or not its PU is searched */
    const int o0 = uni(((const int*)(a.jobs + ectu_job(0, c, nctu)))[0]);
    const int ctx = as_i16(o0), cty = o0 >> 16;
    __syncthreads();
    if (tid < ECTU_PUS)
    {
        const int j = ectu_job(tid, c, nctu);
        const int *p = (const int*)(a.jobs + j);
        const int d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4];
        int ox, oy, s;
        ectu_slot(tid, ox, oy, s);
        const int mnx = as_i16(d2), mny = d2 >> 16, mxx = as_i16(d3), mxy = d3 >> 16;
        /* searched: a square of the slot's size at the slot's place with a range that is not empty */
        const bool present = (d1 & 0xffff) == (s | (s << 8)) && as_i16(d0) == ctx + ox && (d0 >> 16) == cty + oy &&
                             mnx <= mxx && mny <= mxy;
        int *q = lds_i(0, ECTU_L_PU) + tid * 8;
        q[0] = mnx; q[1] = mxx; q[2] = mny; q[3] = mxy; q[4] = as_i16(d4); q[5] = d4 >> 16; q[6] = present; q[7] = j;
        if (present)
        {
            atomicMin(hdr + EH_BX0, mnx); atomicMax(hdr + EH_BX1, mxx);
            atomicMin(hdr + EH_BY0, mny); atomicMax(hdr + EH_BY1, mxy);
            /* what the PUs may read of the reference (their windows) and of the source, relative to the CTU */
            atomicMin(hdr + EH_GX0, ox + mnx); atomicMax(hdr + EH_GX1, ox + s - 1 + mxx);
            atomicMin(hdr + EH_GY0, oy + mny); atomicMax(hdr + EH_GY1, oy + s - 1 + mxy);
            atomicMax(hdr + EH_SX1, ox + s - 1); atomicMax(hdr + EH_SY1, oy + s - 1);
            hdr[EH_ANY] = 1;
        }
        else if ((d1 & 0xffff) != 0)
        {
            /* not what the slot holds: refused (esa_finish_kernel passes the mark on) */
            a.results[j].cost = -1; a.results[j].sad_calls = 0;
        }
    }
    __syncthreads();
    if (!hdr[EH_ANY]) return;
    const int bx0 = hdr[EH_BX0], by0 = hdr[EH_BY0];
    const int nxb = hdr[EH_BX1] - bx0 + 1, nyb = hdr[EH_BY1] - by0 + 1;
    const int gx0 = hdr[EH_GX0], gx1 = hdr[EH_GX1], gy0 = hdr[EH_GY0], gy1 = hdr[EH_GY1];
    if (nxb > ECTU_MAX_SPAN || nyb > ECTU_MAX_SPAN)
    {
        if (tid < ECTU_PUS && lds_i(0, ECTU_L_PU)[tid * 8 + 6]) { const int j = lds_i(0, ECTU_L_PU)[tid * 8 + 7]; a.results[j].cost = -1; a.results[j].sad_calls = 0; }
        return;
    }
    const uint16_t AS1 *ctab = (const uint16_t AS1*)a.ctab;
    /* the CTU's pictures: the launch's own, or the entry of the launch's pair table that slot 0 names */
    PlaneDev pfenc, pref[4];
    job_planes(a, a.jobs + ectu_job(0, c, nctu), pfenc, pref);
    const char AS1 *ref = (const char AS1*)((const P AS1*)pref[0].origin + (long)cty * pref[0].stride + ctx);
    const long rstride = (long)pref[0].stride * BPS;           /* bytes */

    /* the lane's 8x8 block (Morton order) */
    const int bx8 = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4), by8 = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
    uint32_t f[BPS == 1 ? 16 : 1];
    typedef uint32_t __attribute__((aligned(1))) u32u;
    {
        /* Source samples. Blocks of PUs that are not searched read the last searched block's samples instead of running
         * off the plane. 8-bit: the lane's own block into registers, 8 rows of two dwords. 16-bit: the CTU into LDS, a
         * thread per row and run of eight samples. */
        const int sx8 = hdr[EH_SX1] >> 3, sy8 = hdr[EH_SY1] >> 3;
        const long fstride = (long)pfenc.stride * BPS;
        const char AS1 *src = (const char AS1*)((const P AS1*)pfenc.origin + (long)cty * pfenc.stride + ctx);
        if constexpr (BPS == 1)
        {
            const int sbx = bx8 < sx8 ? bx8 : sx8, sby = by8 < sy8 ? by8 : sy8;
            src += (long)(sby * 8) * fstride + sbx * 8;
#pragma unroll
            for (int y = 0; y < 8; y++)
            {
                f[2 * y] = *(const u32u AS1*)(src + (long)y * fstride);
                f[2 * y + 1] = *(const u32u AS1*)(src + (long)y * fstride + 4);
            }
        }
        else
        {
            const int y = tid >> 3, xb = tid & 7;          /* 512 threads: 64 rows x 8 runs */
            const int sy = (y >> 3) < sy8 ? y : sy8 * 8 + (y & 7), sb = xb < sx8 ? xb : sx8;
            const char AS1 *s8 = src + (long)sy * fstride + sb * 16;
            uint32_t *dst = (uint32_t*)(x265amd_smem + E::L_FENC + y * 128 + xb * 16);
#pragma unroll
            for (int k = 0; k < 4; k++) dst[k] = *(const u32u AS1*)(s8 + 4 * k);
            f[0] = 0;
        }
    }
    const int q = lane & 3, Q = (lane >> 2) & 3;
    const uint32_t qm1 = 0u - (uint32_t)(q & 1), qm2 = 0u - (uint32_t)(q >> 1), Qm1 = 0u - (uint32_t)(Q & 1), Qm2 = 0u - (uint32_t)(Q >> 1);
    uint32_t bc8 = 0xffffffffu, bi8 = 0xffffffffu, bc16 = 0xffffffffu, bi16 = 0xffffffffu;
    uint32_t bc32 = 0xffffffffu, bi32 = 0xffffffffu, bc64 = 0xffffffffu, bi64 = 0xffffffffu;

    const int nxg = (nxb + CX - 1) >> CXL, nyg = (nyb + 7) >> 3;
    const int nxs = (nxg + E::XG - 1) / E::XG, xgs = (nxg + nxs - 1) / nxs;       /* strips evened out */
    const int nys = (nyg + ECTU_YG - 1) / ECTU_YG, ygs = (nyg + nys - 1) / nys;
    for (int xs0 = 0; xs0 < nxg; xs0 += xgs)
    {
        const int xgc = nxg - xs0 < xgs ? nxg - xs0 : xgs;
        if (xs0) __syncthreads();               /* every wave is done with the tables of the strip before */
        /* lambda * bits of the strip's columns per PU (bitcost.h:46), as (cost << KSH | column in its group) */
        for (int p = wave; p < ECTU_PUS; p += ECTU_WAVES)
        {
            const int *pq = lds_i(0, ECTU_L_PU) + p * 8;
            const int mnx = uni(pq[0]), mxx = uni(pq[1]), qmx = uni(pq[4]), present = uni(pq[6]);
            for (int i = lane; i < xgc * CX; i += 64)
            {
                const int mvx = bx0 + xs0 * CX + i;
                uint32_t v = E::SENT;
                if (present && mvx >= mnx && mvx <= mxx) v = ctab[as_i16(mvx << 2) - qmx];
                ((uint32_t*)(x265amd_smem + ECTU_L_KX))[p * KXP + i] = (v << KSH) | (uint32_t)(i & (CX - 1));
            }
        }
        for (int ys0 = 0; ys0 < nyg; ys0 += ygs)
        {
            const int ygc = nyg - ys0 < ygs ? nyg - ys0 : ygs;
            if (ys0) __syncthreads();
            for (int p = wave; p < ECTU_PUS; p += ECTU_WAVES)
            {
                const int *pq = lds_i(0, ECTU_L_PU) + p * 8;
                const int mny = uni(pq[2]), mxy = uni(pq[3]), qmy = uni(pq[5]), present = uni(pq[6]);
                const int i = lane;
                if (i < ygc * 8)
                {
                    const int mvy = by0 + ys0 * 8 + i;
                    uint32_t v = E::SENT;
                    if (present && mvy >= mny && mvy <= mxy) v = ctab[as_i16(mvy << 2) - qmy];
                    ((uint32_t*)(x265amd_smem + E::L_KY))[p * ECTU_KYP + i] = (v << KSH) | (uint32_t)((i & 7) << CXL);
                }
            }
            {
                /* the window of the strip: a wave copies whole rows (lane = dword; two dwords with 16-bit samples). Rows
                 * and dwords are held inside what the searched PUs' own windows (with their 8-sample apron) cover:
                 * positions that read beyond that are in no PU's range */
                const int W0 = bx0 + xs0 * CX, Y0 = by0 + ys0 * 8;      /* relative to the CTU */
                const int rows = ygc * 8 + 63, nd = ((xgc * CX + 64) * BPS) >> 2;
                /* dword d holds samples W0 + d * SPD .. + SPD - 1 (SPD = 4 / BPS) */
                constexpr int SPD = 4 / BPS;
                int dlo = (gx0 - 8 - W0 + SPD - 1) / SPD, dhi = (gx1 + 8 - (SPD - 1) - W0) / SPD;
                if (gx0 - 8 - W0 < 0) dlo = 0;
                {
                    const int dw = lane;            /* nd <= 64 */
                    int d = dw < dhi ? dw : dhi;
                    d = d > dlo ? d : dlo;
                    const char AS1 *src0 = ref + (long)W0 * BPS + 4 * d;
                    for (int r0 = wave; r0 < rows; r0 += 4 * ECTU_WAVES)
                    {
                        uint32_t v[4];
#pragma unroll
                        for (int u = 0; u < 4; u++)
                        {
                            int y = Y0 + r0 + u * ECTU_WAVES;
                            y = y < gy0 ? gy0 : (y > gy1 ? gy1 : y);
                            v[u] = *(const u32u AS1*)(src0 + (long)y * rstride);
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++)
                        {
                            const int r = r0 + u * ECTU_WAVES;
                            if (r < rows && dw < nd) *(uint32_t*)(x265amd_smem + E::L_WIN + r * PITCH + 4 * dw) = v[u];
                        }
                    }
                }
            }
            __syncthreads();

            const int ntask = xgc * ygc;
            for (int t = wave; t < ntask; t += ECTU_WAVES)
            {
                const int yg = t / xgc, xg = t - yg * xgc;
                const int org = E::L_WIN + (by8 * 8 + yg * 8) * PITCH + bx8 * 8 * BPS + xg * 4;
                const uint32_t base = (uint32_t)(((ys0 + yg) * 8) << 10) | (uint32_t)((xs0 + xg) * CX);
                /* s[2 r], s[2 r + 1]: row r of the 16x16 PU's sums -- packed pairs of the four columns (8-bit) or the
                 * two columns (16-bit) */
                uint32_t s[16];
                if constexpr (BPS == 1)
                {
                    uint64_t a16[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) a16[k] = 0;
#pragma unroll
                    for (int j = 0; j < 2; j++)
                    {
                        uint64_t rw[15];
#pragma unroll
                        for (int k = 0; k < 15; k++) rw[k] = lds_pair(org + 4 * j + k * PITCH);
#pragma unroll
                        for (int yy = 0; yy < 8; yy++)
#pragma unroll
                            for (int k = 0; k < 8; k++)
                                a16[k] = __builtin_amdgcn_qsad_pk_u16_u8(rw[yy + k], f[2 * yy + j], a16[k]);
                    }
                    /* 8x8: the lane's own PU */
                    const uint4 kx = *(const uint4*)(x265amd_smem + ECTU_L_KX + (lane * KXP + xg * 4) * 4);
                    const uint4 k0 = *(const uint4*)(x265amd_smem + E::L_KY + (lane * ECTU_KYP + yg * 8) * 4);
                    const uint4 k1 = *(const uint4*)(x265amd_smem + E::L_KY + (lane * ECTU_KYP + yg * 8 + 4) * 4);
                    const uint32_t ky[8] = { k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w };
                    uint32_t best = 0xffffffffu;
#pragma unroll
                    for (int k = 0; k < 8; k++)
                    {
                        const uint32_t lo = (uint32_t)a16[k], hi = (uint32_t)(a16[k] >> 32);
                        uint32_t m = ectu_min(ectu_min(ectu_mad_lo(lo, kx.x), ectu_mad_hi(lo, kx.y)),
                                              ectu_min(ectu_mad_lo(hi, kx.z), ectu_mad_hi(hi, kx.w)));
                        best = ectu_min(best, m + ky[k]);
                        s[2 * k] = lo; s[2 * k + 1] = hi;
                    }
                    ectu_update<CXL>(bc8, bi8, best, base);
                }
                else
                {
                    uint32_t tot[ESA_R][2];
                    esa_patch16<8, 8>(org, PITCH, E::L_FENC + by8 * 8 * 128 + bx8 * 16, 8, 8, tot);
                    const uint2 kx = *(const uint2*)(x265amd_smem + ECTU_L_KX + (lane * KXP + xg * 2) * 4);
                    const uint4 k0 = *(const uint4*)(x265amd_smem + E::L_KY + (lane * ECTU_KYP + yg * 8) * 4);
                    const uint4 k1 = *(const uint4*)(x265amd_smem + E::L_KY + (lane * ECTU_KYP + yg * 8 + 4) * 4);
                    const uint32_t ky[8] = { k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w };
                    uint32_t best = 0xffffffffu;
#pragma unroll
                    for (int k = 0; k < 8; k++)
                    {
                        best = ectu_min(best, ectu_min((tot[k][0] << KSH) + kx.x, (tot[k][1] << KSH) + kx.y) + ky[k]);
                        s[2 * k] = tot[k][0]; s[2 * k + 1] = tot[k][1];
                    }
                    ectu_update<CXL>(bc8, bi8, best, base);
                }
                /* 16x16: the quad's sums; lane q of the quad keeps rows 2q and 2q + 1 of the patch */
                uint32_t sel[4];
#pragma unroll
                for (int k = 0; k < 16; k++)
                {
                    s[k] += (uint32_t)dpp_all<0xB1>((int)s[k]);
                    s[k] += (uint32_t)dpp_all<0x4E>((int)s[k]);
                }
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const uint32_t t0 = ectu_pick(qm1, s[k], s[4 + k]), t1 = ectu_pick(qm1, s[8 + k], s[12 + k]);
                    sel[k] = ectu_pick(qm2, t0, t1);
                }
                /* w: the lane's share of the patch 32 bits wide, row by row (2 rows x CX columns) */
                uint32_t w[2 * CX];
                {
                    const int p = 64 + (lane >> 2);
                    const uint2 ky = *(const uint2*)(x265amd_smem + E::L_KY + (p * ECTU_KYP + yg * 8 + 2 * q) * 4);
                    if constexpr (BPS == 1)
                    {
                        /* still packed: 16 x 16 x 255 fits 16 bits */
                        const uint4 kx = *(const uint4*)(x265amd_smem + ECTU_L_KX + (p * KXP + xg * 4) * 4);
                        const uint32_t mA = ectu_min(ectu_min(ectu_mad_lo(sel[0], kx.x), ectu_mad_hi(sel[0], kx.y)),
                                                     ectu_min(ectu_mad_lo(sel[1], kx.z), ectu_mad_hi(sel[1], kx.w))) + ky.x;
                        const uint32_t mB = ectu_min(ectu_min(ectu_mad_lo(sel[2], kx.x), ectu_mad_hi(sel[2], kx.y)),
                                                     ectu_min(ectu_mad_lo(sel[3], kx.z), ectu_mad_hi(sel[3], kx.w))) + ky.y;
                        ectu_update<CXL>(bc16, bi16, ectu_min(mA, mB), base);
#pragma unroll
                        for (int k = 0; k < 4; k++) { w[2 * k] = sel[k] & 0xffffu; w[2 * k + 1] = sel[k] >> 16; }
                    }
                    else
                    {
                        const uint2 kx = *(const uint2*)(x265amd_smem + ECTU_L_KX + (p * KXP + xg * 2) * 4);
                        const uint32_t mA = ectu_min((sel[0] << KSH) + kx.x, (sel[1] << KSH) + kx.y) + ky.x;
                        const uint32_t mB = ectu_min((sel[2] << KSH) + kx.x, (sel[3] << KSH) + kx.y) + ky.y;
                        ectu_update<CXL>(bc16, bi16, ectu_min(mA, mB), base);
#pragma unroll
                        for (int k = 0; k < 4; k++) w[k] = sel[k];
                    }
                }
                /* 32x32: the sums of the four quads of a row of sixteen lanes */
                {
#pragma unroll
                    for (int k = 0; k < 2 * CX; k++)
                    {
                        w[k] += (uint32_t)dpp_all<0x124>((int)w[k]);     /* row_ror:4 */
                        w[k] += (uint32_t)dpp_all<0x128>((int)w[k]);     /* row_ror:8 */
                    }
                    const int p = 80 + (lane >> 4);
                    const uint2 ky = *(const uint2*)(x265amd_smem + E::L_KY + (p * ECTU_KYP + yg * 8 + 2 * q) * 4);
                    uint32_t mA, mB;
                    if constexpr (BPS == 1)
                    {
                        const uint4 kx = *(const uint4*)(x265amd_smem + ECTU_L_KX + (p * KXP + xg * 4) * 4);
                        mA = ectu_min(ectu_min((w[0] << KSH) + kx.x, (w[1] << KSH) + kx.y), ectu_min((w[2] << KSH) + kx.z, (w[3] << KSH) + kx.w)) + ky.x;
                        mB = ectu_min(ectu_min((w[4] << KSH) + kx.x, (w[5] << KSH) + kx.y), ectu_min((w[6] << KSH) + kx.z, (w[7] << KSH) + kx.w)) + ky.y;
                    }
                    else
                    {
                        const uint2 kx = *(const uint2*)(x265amd_smem + ECTU_L_KX + (p * KXP + xg * 2) * 4);
                        mA = ectu_min((w[0] << KSH) + kx.x, (w[1] << KSH) + kx.y) + ky.x;
                        mB = ectu_min((w[2] << KSH) + kx.x, (w[3] << KSH) + kx.y) + ky.y;
                    }
                    ectu_update<CXL>(bc32, bi32, ectu_min(mA, mB), base);
                }
                /* 64x64: the four rows of sixteen lanes are added up by the two lane swaps; lane (Q, q) of a row keeps,
                 * of patch row 2q + (Q >> 1), columns 2 (Q & 1), + 1 (8-bit) or column Q & 1 (16-bit) */
                if constexpr (BPS == 1)
                {
                    const uint32_t t0 = ectu_pick(Qm1, w[0], w[2]), t1 = ectu_pick(Qm1, w[4], w[6]);
                    const uint32_t v0 = ectu_pick(Qm1, w[1], w[3]), v1 = ectu_pick(Qm1, w[5], w[7]);
                    uint32_t u0 = ectu_pick(Qm2, t0, t1), u1 = ectu_pick(Qm2, v0, v1);
                    auto r0 = __builtin_amdgcn_permlane16_swap(u0, u0, false, false);
                    u0 = r0[0] + r0[1];
                    auto r1 = __builtin_amdgcn_permlane16_swap(u1, u1, false, false);
                    u1 = r1[0] + r1[1];
                    r0 = __builtin_amdgcn_permlane32_swap(u0, u0, false, false);
                    u0 = r0[0] + r0[1];
                    r1 = __builtin_amdgcn_permlane32_swap(u1, u1, false, false);
                    u1 = r1[0] + r1[1];
                    const uint2 kx = *(const uint2*)(x265amd_smem + ECTU_L_KX + (84 * KXP + xg * 4 + 2 * (Q & 1)) * 4);
                    const uint32_t ky = *(const uint32_t*)(x265amd_smem + E::L_KY + (84 * ECTU_KYP + yg * 8 + 2 * q + (Q >> 1)) * 4);
                    ectu_update<CXL>(bc64, bi64, ectu_min((u0 << KSH) + kx.x, (u1 << KSH) + kx.y) + ky, base);
                }
                else
                {
                    uint32_t u0 = ectu_pick(Qm2, ectu_pick(Qm1, w[0], w[1]), ectu_pick(Qm1, w[2], w[3]));
                    auto r0 = __builtin_amdgcn_permlane16_swap(u0, u0, false, false);
                    u0 = r0[0] + r0[1];
                    r0 = __builtin_amdgcn_permlane32_swap(u0, u0, false, false);
                    u0 = r0[0] + r0[1];
                    const uint32_t kx = *(const uint32_t*)(x265amd_smem + ECTU_L_KX + (84 * KXP + xg * 2 + (Q & 1)) * 4);
                    const uint32_t ky = *(const uint32_t*)(x265amd_smem + E::L_KY + (84 * ECTU_KYP + yg * 8 + 2 * q + (Q >> 1)) * 4);
                    ectu_update<CXL>(bc64, bi64, (u0 << KSH) + kx + ky, base);
                }
            }
        }
    }

    /* the PUs' minima: over the lanes that share a PU, then over the waves */
    ectu_lane_min(bc16, bi16, 3);
    ectu_lane_min(bc32, bi32, 15);
    ectu_lane_min(bc64, bi64, 63);
    __syncthreads();
    {
        uint32_t *red = (uint32_t*)(x265amd_smem + ECTU_L_RED) + wave * ECTU_PUS * 2;
        red[lane * 2] = bc8; red[lane * 2 + 1] = bi8;
        if (q == 0) { red[(64 + (lane >> 2)) * 2] = bc16; red[(64 + (lane >> 2)) * 2 + 1] = bi16; }
        if ((lane & 15) == 0) { red[(80 + (lane >> 4)) * 2] = bc32; red[(80 + (lane >> 4)) * 2 + 1] = bi32; }
        if (lane == 0) { red[84 * 2] = bc64; red[84 * 2 + 1] = bi64; }
    }
    __syncthreads();
    if (tid < ECTU_PUS)
    {
        const int *pq = lds_i(0, ECTU_L_PU) + tid * 8;
        if (pq[6])
        {
            uint32_t bc = 0xffffffffu, bi = 0xffffffffu;
            for (int k = 0; k < ECTU_WAVES; k++)
            {
                const uint32_t *red = (const uint32_t*)(x265amd_smem + ECTU_L_RED) + (k * ECTU_PUS + tid) * 2;
                const uint32_t oc = red[0], oi = red[1];
                if (oc < bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
            }
            /* as esa_scan_kernel leaves it: cost and the raster index inside the PU's own range */
            const int mvx = bx0 + (int)(bi & 1023u), mvy = by0 + (int)(bi >> 10);
            const int j = pq[7];
            a.results[j].cost = (int)bc;
            a.results[j].sad_calls = (uint32_t)((mvy - pq[2]) * (pq[1] - pq[0] + 1) + (mvx - pq[0]));
        }
    }
}

} // namespace x265amd
#endif
