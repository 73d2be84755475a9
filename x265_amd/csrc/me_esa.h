/* Exhaustive integer search (X265_FULL_SEARCH, source/encoder/motion.cpp:1021-1056) as a
 * workgroup-per-PU kernel (8-bit planes: packed quad-SAD; 16-bit planes: v_sad_u16).
 *
 * The reference scans [mvmin, mvmax] in raster order and keeps the first position whose
 * SAD + lambda * bits beats everything before it. Here the reference window of the PU is staged
 * once into LDS, and every lane owns a 4 x ESA_R patch of candidate positions:
 *   - v_qsad_pk_u16_u8 measures four horizontally adjacent candidates against four source
 *     pixels in one instruction (the 64-bit operand is the sliding reference window, so no
 *     byte realignment is needed: a lane's first candidate sits on a dword of the LDS window);
 *   - the ESA_R vertically adjacent candidates share every reference row the lane loads: the
 *     row that is source row y for candidate c is source row y - 1 for candidate c + 1.
 * Sums stay packed in 16 bits for at most 64 accumulations (64 * 4 * 255 < 65536) and are
 * widened per column of 4 pixels. The winner is the minimum of (cost, raster index), which is
 * the position the sequential strict-'<' scan would have kept.
 *
 * The stages around the scan (start-up candidates, sub-pel refinement) run in a second kernel,
 * one wave per PU through the Searcher, so that the scan's workgroups hold their LDS only
 * while all of their waves are busy.
 */
#ifndef X265AMD_ME_ESA_H
#define X265AMD_ME_ESA_H

#include "me_search.h"

namespace x265amd {

#ifndef ESA_THREADS
#define ESA_THREADS 256
#endif
#define ESA_R 8                  /* candidate rows per lane */
/* candidate columns a lane measures at once and bytes per sample: four with the packed quad-SAD of 8-bit samples,
 * two (an even and an odd one) with v_sad_u16. SENT is added to the cost of positions outside the range: one bit
 * above every real cost (a 64x64 SAD of 12-bit samples plus two table entries stays below 2^25). */
template<typename P> struct EsaLane
{
    static constexpr int CX = sizeof(P) == 1 ? 4 : 2, BPS = (int)sizeof(P);
    static constexpr uint32_t SENT = sizeof(P) == 1 ? 0x1000000u : 0x2000000u;
};

/* how one PU's scan is laid out; the same values in every thread of the workgroup */
struct EsaGeo
{
    int nx, ny;       /* positions per row / rows of the search range */
    int nxg, nyg;     /* groups of cx columns / ESA_R rows */
    int xgs, ygs;     /* groups per LDS strip */
    int pitch;        /* window row pitch in bytes, a whole number of dwords */
    int off_cx, off_cy, off_win;   /* LDS byte offsets */
};

__host__ __device__ inline int esa_pitch(int xgs, int w, int cx = 4, int bps = 1)
{
    return (((xgs * cx + w) * bps + 3) & ~3) + 4;   /* the last lane reads one dword past its columns */
}

/* LDS bytes a strip of xgs x ygs groups needs after the fixed part */
__host__ __device__ inline int esa_strip_bytes(int xgs, int ygs, int w, int h, int cx = 4, int bps = 1)
{
    return 4 * cx * xgs + 4 * ESA_R * ygs + esa_pitch(xgs, w, cx, bps) * (ygs * ESA_R + h - 1);
}

/* Largest strips that fit `budget` bytes, evened out over the range; false if not even one
 * group fits. */
__host__ __device__ inline bool esa_plan(int nx, int ny, int w, int h, int budget, EsaGeo &g, int cx = 4, int bps = 1)
{
    g.nx = nx; g.ny = ny;
    g.nxg = (nx + cx - 1) / cx; g.nyg = (ny + ESA_R - 1) / ESA_R;
    int xgs = g.nxg;
    while (xgs > 1 && esa_strip_bytes(xgs, 1, w, h, cx, bps) > budget) xgs = (xgs + 1) >> 1;
    if (esa_strip_bytes(xgs, 1, w, h, cx, bps) > budget) return false;
    const int nxs = (g.nxg + xgs - 1) / xgs;
    xgs = (g.nxg + nxs - 1) / nxs;
    const int p = esa_pitch(xgs, w, cx, bps);
    int ygs = (budget - 4 * cx * xgs - p * (h - 1)) / (ESA_R * (p + 4));
    if (ygs > g.nyg) ygs = g.nyg;
    const int nys = (g.nyg + ygs - 1) / ygs;
    ygs = (g.nyg + nys - 1) / nys;
    g.xgs = xgs; g.ygs = ygs; g.pitch = p;
    return true;
}

/* two consecutive dwords of LDS at a 4-byte aligned offset */
__device__ __forceinline__ uint64_t lds_pair(int off)
{
    const uint32_t *q = (const uint32_t*)(x265amd_smem + off);
    return ((uint64_t)q[1] << 32) | q[0];
}

/* SADs of the 4 x ESA_R candidates whose first one has its top-left reference pixel at LDS byte
 * `org`; the source block (w x h) sits at `fenc` with a pitch of 64 bytes. W, H != 0 fix the PU
 * size at compile time: the loops unroll, the accumulators start from the instruction's zero
 * operand and the single widening needs no running totals (8x8 PUs spend more instructions on
 * that bookkeeping than on SADs otherwise). */
/* PU sizes whose packed 16-bit sums hold a whole candidate (at most 64 accumulations of 4 pixels): esa_patch then
 * leaves the sums packed in a16 and does not fill tot. */
template<int W, int H> struct EsaPacked { static constexpr bool value = W && H && (W >> 2) * H <= 64; };

template<int W, int H>
__device__ __forceinline__ void esa_patch(int org, int pitch, int fenc, int w_, int h_, uint32_t tot[ESA_R][4], uint64_t a16[ESA_R])
{
    const int w = W ? W : w_, h = H ? H : h_;
    static_assert(ESA_R == 8, "the row bookkeeping below is written for 8 candidate rows");
#pragma unroll
    for (int c = 0; c < ESA_R; c++)
    {
        a16[c] = 0;
        if (!EsaPacked<W, H>::value)
#pragma unroll
            for (int i = 0; i < 4; i++) tot[c][i] = 0;
    }
    int cnt = 0;
    const int ncol = w >> 2;
#pragma unroll
    for (int j = 0; j < ncol; j++)
    {
        const int col = org + 4 * j, fcol = fenc + 4 * j;
        /* Source row y meets reference rows y .. y + 7 (one per candidate row). A block of N
         * source rows starting at y0 therefore needs rows y0 .. y0 + 6, kept from the block
         * before (old), and N new ones; all loads of a block are issued before its SADs. */
        uint64_t old[7];
#pragma unroll
        for (int k = 0; k < 7; k++) old[k] = lds_pair(col + k * pitch);
        /* the first source dword of a block is fetched one block ahead, so that a block's SADs
         * start at once and its own loads have the first seven SADs to arrive in */
        uint32_t f0 = *(const uint32_t*)(x265amd_smem + fcol);
        int y0 = 0;
#pragma unroll
        for (; y0 + 8 <= h; y0 += 8)
        {
            uint64_t nw[8];
            uint32_t f[8];
            f[0] = f0;
#pragma unroll
            for (int k = 0; k < 8; k++)
            {
                nw[k] = lds_pair(col + (y0 + 7 + k) * pitch);
                if (k) f[k] = *(const uint32_t*)(x265amd_smem + fcol + (y0 + k) * 64);
            }
            f0 = *(const uint32_t*)(x265amd_smem + fcol + (y0 + 8 < h ? y0 + 8 : y0) * 64);
#pragma unroll
            for (int yy = 0; yy < 8; yy++)
#pragma unroll
                for (int c = 0; c < ESA_R; c++)
                    a16[c] = __builtin_amdgcn_qsad_pk_u16_u8(yy + c < 7 ? old[yy + c < 7 ? yy + c : 0] : nw[yy + c < 7 ? 0 : yy + c - 7], f[yy], a16[c]);
#pragma unroll
            for (int k = 0; k < 7; k++) old[k] = nw[k + 1];
        }
        if (h & 4)
        {
            /* heights 4 and 12: a last block of four source rows */
            uint64_t nw[4];
            uint32_t f[4];
            f[0] = f0;
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                nw[k] = lds_pair(col + (y0 + 7 + k) * pitch);
                if (k) f[k] = *(const uint32_t*)(x265amd_smem + fcol + (y0 + k) * 64);
            }
#pragma unroll
            for (int yy = 0; yy < 4; yy++)
#pragma unroll
                for (int c = 0; c < ESA_R; c++)
                    a16[c] = __builtin_amdgcn_qsad_pk_u16_u8(yy + c < 7 ? old[yy + c < 7 ? yy + c : 0] : nw[yy + c < 7 ? 0 : yy + c - 7], f[yy], a16[c]);
        }
        cnt += h;
        if (!EsaPacked<W, H>::value && (cnt + h > 64 || j == ncol - 1))
        {
#pragma unroll
            for (int c = 0; c < ESA_R; c++)
            {
                const uint32_t lo = (uint32_t)a16[c], hi = (uint32_t)(a16[c] >> 32);
                tot[c][0] += lo & 0xffff; tot[c][1] += lo >> 16;
                tot[c][2] += hi & 0xffff; tot[c][3] += hi >> 16;
                a16[c] = 0;
            }
            cnt = 0;
        }
    }
}

/* The same for 16-bit samples: the 2 x ESA_R candidates whose first one has its top-left reference sample at LDS byte
 * `org` (a dword boundary), source pitch 128 bytes. v_sad_u16 measures two samples; the even candidate's operand is
 * the reference dword itself, the odd one's the pair of dwords shifted by one sample. Sums are 32 bits wide. */
template<int W, int H>
__device__ __forceinline__ void esa_patch16(int org, int pitch, int fenc, int w_, int h_, uint32_t tot[ESA_R][2])
{
    const int w = W ? W : w_, h = H ? H : h_;
#pragma unroll
    for (int c = 0; c < ESA_R; c++) tot[c][0] = tot[c][1] = 0;
    const int ncol = w >> 1;
#pragma unroll
    for (int j = 0; j < ncol; j++)
    {
        const int col = org + 4 * j, fcol = fenc + 4 * j;
        uint32_t oe[7], oo[7];
#pragma unroll
        for (int k = 0; k < 7; k++)
        {
            const uint64_t pr = lds_pair(col + k * pitch);
            oe[k] = (uint32_t)pr; oo[k] = __builtin_amdgcn_alignbyte((uint32_t)(pr >> 32), (uint32_t)pr, 2);
        }
        int y0 = 0;
#pragma unroll
        for (; y0 + 8 <= h; y0 += 8)
        {
            uint32_t ne[8], no[8], f[8];
#pragma unroll
            for (int k = 0; k < 8; k++)
            {
                const uint64_t pr = lds_pair(col + (y0 + 7 + k) * pitch);
                ne[k] = (uint32_t)pr; no[k] = __builtin_amdgcn_alignbyte((uint32_t)(pr >> 32), (uint32_t)pr, 2);
                f[k] = *(const uint32_t*)(x265amd_smem + fcol + (y0 + k) * 128);
            }
#pragma unroll
            for (int yy = 0; yy < 8; yy++)
#pragma unroll
                for (int c = 0; c < ESA_R; c++)
                {
                    tot[c][0] = __builtin_amdgcn_sad_u16(yy + c < 7 ? oe[yy + c < 7 ? yy + c : 0] : ne[yy + c < 7 ? 0 : yy + c - 7], f[yy], tot[c][0]);
                    tot[c][1] = __builtin_amdgcn_sad_u16(yy + c < 7 ? oo[yy + c < 7 ? yy + c : 0] : no[yy + c < 7 ? 0 : yy + c - 7], f[yy], tot[c][1]);
                }
#pragma unroll
            for (int k = 0; k < 7; k++) { oe[k] = ne[k + 1]; oo[k] = no[k + 1]; }
        }
        if (h & 4)
        {
            /* heights 4 and 12: a last block of four source rows */
            uint32_t ne[4], no[4], f[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const uint64_t pr = lds_pair(col + (y0 + 7 + k) * pitch);
                ne[k] = (uint32_t)pr; no[k] = __builtin_amdgcn_alignbyte((uint32_t)(pr >> 32), (uint32_t)pr, 2);
                f[k] = *(const uint32_t*)(x265amd_smem + fcol + (y0 + k) * 128);
            }
#pragma unroll
            for (int yy = 0; yy < 4; yy++)
#pragma unroll
                for (int c = 0; c < ESA_R; c++)
                {
                    tot[c][0] = __builtin_amdgcn_sad_u16(yy + c < 7 ? oe[yy + c < 7 ? yy + c : 0] : ne[yy + c < 7 ? 0 : yy + c - 7], f[yy], tot[c][0]);
                    tot[c][1] = __builtin_amdgcn_sad_u16(yy + c < 7 ? oo[yy + c < 7 ? yy + c : 0] : no[yy + c < 7 ? 0 : yy + c - 7], f[yy], tot[c][1]);
                }
        }
    }
}

/* LDS layout of the scan kernel: [counter, per-wave winners][source block, pitch 64][cost of
 * each column's MV x][cost of each row's MV y][reference window strip] */
#define ESA_L_CNT 0
#define ESA_L_RED 16
#define ESA_L_FENC 64
__host__ __device__ inline int esa_fixed_lds(int max_dim, int bps = 1) { return ESA_L_FENC + max_dim * 64 * bps; }

/* A job the scan cannot take (PU larger than the launch allows, empty range, window budget too
 * small for a single group) gets cost -1 from both kernels. */
__device__ __forceinline__ bool esa_job_ok(const JobRegs &J, int w, int h, int max_dim, int win_budget, EsaGeo &g, int cx, int bps)
{
    const int nx = J.mxx - J.mnx + 1, ny = J.mxy - J.mny + 1;
    return w <= max_dim && h <= max_dim && nx > 0 && ny > 0 && esa_plan(nx, ny, w, h, win_budget, g, cx, bps);
}

/* The raster scan of motion.cpp:1025-1053 for one PU per workgroup: writes the smallest
 * (SAD + lambda * bits, raster index) of the range into results[j].cost / .sad_calls, where
 * esa_finish_kernel picks it up. */
/* The scan of one PU by its workgroup: every strip of the range staged and measured; leaves this lane's best (cost,
 * raster index) in bc / bi. W, H != 0: the PU size, fixed at compile time. */
template<typename P, int W, int H>
__device__ __forceinline__ void esa_scan_pu(const SearchArgs &a, const JobRegs &J, const EsaGeo &g, const PlaneDev *pref, int w, int h,
                                            int nowrap, int tid, int wave, int lane, uint32_t &bc, uint32_t &bi)
{
    constexpr int CX = EsaLane<P>::CX, BPS = EsaLane<P>::BPS;
    const int nx = g.nx, ny = g.ny;
    const uint16_t AS1 *ctab = (const uint16_t AS1*)a.ctab;
    const P AS1 *ref = (const P AS1*)pref[0].origin + (long)J.puy * pref[0].stride + J.pux;
    const long rstride = (long)pref[0].stride * BPS;          /* bytes */
    const int pd = g.pitch >> 2;
    bool first = true;
    for (int ys0 = 0; ys0 < g.nyg; ys0 += g.ygs)
        for (int xs0 = 0; xs0 < g.nxg; xs0 += g.xgs)
        {
            const int xgc = g.nxg - xs0 < g.xgs ? g.nxg - xs0 : g.xgs;   /* groups in this strip */
            const int ygc = g.nyg - ys0 < g.ygs ? g.nyg - ys0 : g.ygs;
            if (!first) __syncthreads();
            first = false;
            /* lambda * bits of each column / row of the strip, as the reference's two table
             * operands (bitcost.h:46); positions outside the range carry the sentinel */
            for (int i = tid; i < CX * xgc; i += ESA_THREADS)
            {
                const int dxi = xs0 * CX + i;
                lds_i(0, g.off_cx)[i] = dxi < nx ? (int)ctab[as_i16((J.mnx + dxi) << 2) - J.qmx] : (int)EsaLane<P>::SENT;
            }
            for (int i = tid; i < ESA_R * ygc; i += ESA_THREADS)
            {
                const int dyi = ys0 * ESA_R + i;
                lds_i(0, g.off_cy)[i] = dyi < ny ? (int)ctab[as_i16((J.mny + dyi) << 2) - J.qmy] : (int)EsaLane<P>::SENT;
            }
            {
                /* window strip: rows below the last row any valid candidate needs repeat that
                 * row, columns stop within 7 bytes of the last needed one. A wave copies whole
                 * rows (lane = dword), eight rows in flight before the first is stored. */
                const int rows = ygc * ESA_R + h - 1;
                const int rlast = (ny - 1 - ys0 * ESA_R) + h - 1;
                const int dlast = (((nx - 1 - xs0 * CX + w - 1) * BPS) >> 2) + 1;
                const char AS1 *src0 = (const char AS1*)(ref + J.mnx + xs0 * CX) + (long)(J.mny + ys0 * ESA_R) * rstride;
#ifdef X265AMD_ESA_SKIP
                /* timing builds only: 1 = no window staging, 2 = no scan; results are then meaningless */
                if (X265AMD_ESA_SKIP != 1)
#endif
                for (int d0 = 0; d0 < pd; d0 += 64)
                {
                    const int d = d0 + lane;
                    const unsigned voff = 4u * (unsigned)(d < dlast ? d : dlast);
                    for (int r0 = wave; r0 < rows; r0 += 8 * (ESA_THREADS / 64))
                    {
                        uint32_t v[8];
#pragma unroll
                        for (int u = 0; u < 8; u++)
                        {
                            int r = r0 + u * (ESA_THREADS / 64);
                            r = r < rlast ? r : rlast;
                            typedef uint32_t __attribute__((aligned(1))) u32u;
                            v[u] = *(const u32u AS1*)(src0 + r * rstride + voff);
                        }
#pragma unroll
                        for (int u = 0; u < 8; u++)
                        {
                            const int r = r0 + u * (ESA_THREADS / 64);
                            if (r < rows && d < pd) *(uint32_t*)(x265amd_smem + g.off_win + r * g.pitch + 4 * d) = v[u];
                        }
                    }
                }
            }
            if (tid == 0) *lds_i(0, ESA_L_CNT) = 0;
            __syncthreads();

#ifdef X265AMD_ESA_SKIP
            const int ni = X265AMD_ESA_SKIP == 2 ? 0 : xgc * ygc, inv = (65536 + xgc - 1) / xgc;
#else
            const int ni = xgc * ygc, inv = (65536 + xgc - 1) / xgc;
#endif
            for (;;)
            {
                int i0 = 0;
                if (lane == 0) i0 = atomicAdd(lds_i(0, ESA_L_CNT), 64);
                i0 = uni(i0);
                if (i0 >= ni) break;
                const bool live = i0 + lane < ni;
                const int it = live ? i0 + lane : ni - 1;
                const int yg = (it * inv) >> 16, xg = it - yg * xgc;
                uint32_t tot[ESA_R][CX];
                uint64_t a16[ESA_R];
                const int porg = g.off_win + yg * ESA_R * g.pitch + xg * 4;     /* CX * BPS = 4 bytes per group */
                constexpr bool packed = BPS == 1 && EsaPacked<W, H>::value;
                if constexpr (BPS == 1) esa_patch<W, H>(porg, g.pitch, ESA_L_FENC, w, h, tot, a16);
                else esa_patch16<W, H>(porg, g.pitch, ESA_L_FENC, w, h, tot);
                int cx[CX];
#pragma unroll
                for (int i = 0; i < CX; i++) cx[i] = lds_i(0, g.off_cx)[xg * CX + i];
                uint32_t best = 0xffffffffu;
                if constexpr (BPS == 2)
                {
                    /* 16-bit samples: (cost << 4 | patch index); cost < 2^27 with the sentinel */
#pragma unroll
                    for (int c = 0; c < ESA_R; c++)
                    {
                        const int cy = lds_i(0, g.off_cy)[yg * ESA_R + c];
#pragma unroll
                        for (int i = 0; i < CX; i++)
                        {
                            const uint32_t cost = tot[c][i] + ((uint32_t)(cx[i] + cy) & (0xffffu | (3u * EsaLane<P>::SENT)));
                            const uint32_t key = (cost << 4) | (uint32_t)(c * CX + i);
                            best = key < best ? key : best;
                        }
                    }
                }
                else if (packed && nowrap)
                {
                    /* as below, straight from the packed sums: key = sum16 * 32 + (cx << 5 | column) in one
                     * multiply-add that picks its 16-bit half */
                    uint32_t kx[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) kx[i] = ((uint32_t)cx[i] << 5) | (uint32_t)i;
#pragma unroll
                    for (int c = 0; c < ESA_R; c++)
                    {
                        const uint32_t ky = ((uint32_t)lds_i(0, g.off_cy)[yg * ESA_R + c] << 5) | (uint32_t)(c * 4);
                        const uint32_t lo = (uint32_t)a16[c], hi = (uint32_t)(a16[c] >> 32);
                        uint32_t k0, k1, k2, k3;
                        asm("v_mad_u32_u16 %0, %1, 32, %2" : "=v"(k0) : "v"(lo), "v"(kx[0]));
                        asm("v_mad_u32_u16 %0, %1, 32, %2 op_sel:[1,0,0,0]" : "=v"(k1) : "v"(lo), "v"(kx[1]));
                        asm("v_mad_u32_u16 %0, %1, 32, %2" : "=v"(k2) : "v"(hi), "v"(kx[2]));
                        asm("v_mad_u32_u16 %0, %1, 32, %2 op_sel:[1,0,0,0]" : "=v"(k3) : "v"(hi), "v"(kx[3]));
                        uint32_t m = k0 < k1 ? k0 : k1;
                        m = k2 < m ? k2 : m;
                        m = k3 < m ? k3 : m;
                        m += ky;
                        best = m < best ? m : best;
                    }
                }
                else
                {
                if (packed)
#pragma unroll
                    for (int c = 0; c < ESA_R; c++)
                    {
                        const uint32_t lo = (uint32_t)a16[c], hi = (uint32_t)(a16[c] >> 32);
                        tot[c][0] = lo & 0xffff; tot[c][1] = lo >> 16; tot[c][2] = hi & 0xffff; tot[c][3] = hi >> 16;
                    }
                if (nowrap)
                {
                    /* No two entries of this QP's table add up to 2^16 (the host checked), so the
                     * uint16 wrap of bitcost.h:46 never acts and cost = SAD + cx + cy. The key
                     * (cost << 5 | patch index) then splits into a per-column and a per-row term,
                     * and adding the row term after the minimum over a row's four columns keeps
                     * the order: two instructions per position instead of five. */
                    uint32_t kx[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) kx[i] = ((uint32_t)cx[i] << 5) | (uint32_t)i;
#pragma unroll
                    for (int c = 0; c < ESA_R; c++)
                    {
                        const uint32_t ky = ((uint32_t)lds_i(0, g.off_cy)[yg * ESA_R + c] << 5) | (uint32_t)(c * 4);
                        const uint32_t k0 = (tot[c][0] << 5) + kx[0], k1 = (tot[c][1] << 5) + kx[1];
                        const uint32_t k2 = (tot[c][2] << 5) + kx[2], k3 = (tot[c][3] << 5) + kx[3];
                        uint32_t m = k0 < k1 ? k0 : k1;
                        m = k2 < m ? k2 : m;
                        m = k3 < m ? k3 : m;
                        m += ky;
                        best = m < best ? m : best;
                    }
                }
                else
#pragma unroll
                for (int c = 0; c < ESA_R; c++)
                {
                    const int cy = lds_i(0, g.off_cy)[yg * ESA_R + c];
#pragma unroll
                    for (int i = 0; i < 4; i++)
                    {
                        /* uint16 wrap of the sum of the two table entries; the sentinel bits survive */
                        const uint32_t cost = tot[c][i] + ((uint32_t)(cx[i] + cy) & (0xffffu | (3u * EsaLane<P>::SENT)));
                        const uint32_t key = (cost << 5) | (uint32_t)(c * 4 + i);
                        best = key < best ? key : best;
                    }
                }
                }
                /* cost and patch index (row * 4 + column) of the lane's best position */
                const uint32_t cost = BPS == 2 ? best >> 4 : best >> 5;
                const uint32_t ci = BPS == 2 ? (((best & 15) >> 1) << 2) | (best & 1) : best & 31;
                const uint32_t idx = (uint32_t)((ys0 + yg) * ESA_R + (ci >> 2)) * (uint32_t)nx + (uint32_t)((xs0 + xg) * CX + (ci & 3));
                if (live && cost < EsaLane<P>::SENT && (cost < bc || (cost == bc && idx < bi))) { bc = cost; bi = idx; }
            }
        }
}

/* SZ: 8 or 16 when the launch is one of 8x8 or 16x16 PUs (PUs of other sizes still go through, by the general
 * code); each size gets a kernel of its own so that its registers are allocated for its own unrolled patch. */
template<typename P, int SZ>
__global__ __launch_bounds__(ESA_THREADS) __attribute__((amdgpu_waves_per_eu(4))) void esa_scan_kernel(SearchArgs a, int win_budget, int nowrap)
{
    constexpr int CX = EsaLane<P>::CX, BPS = EsaLane<P>::BPS;
    const int j = xcd_remap(blockIdx.x, a.njobs);
    if (j >= a.njobs) return;
    const x265amd_job *job = a.jobs + j;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#ifdef X265AMD_TRACE
    /* debug builds only: per-wave start / end time (100 MHz counter) and placement, in a side
     * buffer behind the result records */
    unsigned long long *trace = (unsigned long long*)(a.results + a.njobs) + ((size_t)blockIdx.x * (ESA_THREADS / 64) + wave) * 4;
    if (lane == 0)
    {
        trace[0] = wall_clock64();
        trace[2] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);
    }
#endif
    JobRegs J;
    J.load(job);
    const int w = uni(job->w), h = uni(job->h);
    EsaGeo g;
    if (!esa_job_ok(J, w, h, a.max_dim, win_budget, g, CX, BPS))
    {
        if (tid == 0) { a.results[j].cost = -1; a.results[j].sad_calls = 0; }
        return;
    }
    PlaneDev pfenc, pref[4];
    job_planes(a, job, pfenc, pref);
    g.off_cx = esa_fixed_lds(a.max_dim, BPS);
    g.off_cy = g.off_cx + 4 * CX * g.xgs;
    g.off_win = g.off_cy + 4 * ESA_R * g.ygs;

    {
        /* source block into LDS, pitch 64 samples (setSourcePU, motion.cpp:116-142) */
        const char AS1 *src = (const char AS1*)((const P AS1*)pfenc.origin + (long)J.puy * pfenc.stride + J.pux);
        const int wd = (w * BPS) >> 2;
        for (int i = tid; i < wd * h; i += ESA_THREADS)
        {
            const int y = i / wd, x = i - y * wd;
            typedef uint32_t __attribute__((aligned(1))) u32u;
            *(uint32_t*)(x265amd_smem + ESA_L_FENC + y * 64 * BPS + 4 * x) = *(const u32u AS1*)(src + (long)y * pfenc.stride * BPS + 4 * x);
        }
    }
    uint32_t bc = 0xffffffffu, bi = 0xffffffffu;   /* this lane's best (cost, raster index) */
    if (SZ && w == SZ && h == SZ) esa_scan_pu<P, SZ, SZ>(a, J, g, pref, w, h, nowrap, tid, wave, lane, bc, bi);
    else esa_scan_pu<P, 0, 0>(a, J, g, pref, w, h, nowrap, tid, wave, lane, bc, bi);
    for (int o = 1; o < 64; o <<= 1)
    {
        const uint32_t oc = __shfl_xor(bc, o), oi = __shfl_xor(bi, o);
        if (oc < bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
    }
    if (lane == 0) { lds_i(0, ESA_L_RED)[2 * wave] = (int)bc; lds_i(0, ESA_L_RED)[2 * wave + 1] = (int)bi; }
    __syncthreads();
    if (tid == 0)
    {
        for (int k = 1; k < ESA_THREADS / 64; k++)
        {
            const uint32_t oc = (uint32_t)lds_i(0, ESA_L_RED)[2 * k], oi = (uint32_t)lds_i(0, ESA_L_RED)[2 * k + 1];
            if (oc < bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
        }
        a.results[j].cost = (int)bc;
        a.results[j].sad_calls = bi;
    }
#ifdef X265AMD_TRACE
    if (lane == 0) trace[1] = trace[3] = wall_clock64();
#endif
}

/* The stages around the scan, one wave per PU: the start-up candidates (motion.cpp:562-621, which
 * do not depend on the scan), the scan's winner folded in with the reference's strict '<', then
 * the sub-pel refinement (or, in launches of 8x8 / 16x16 PUs, the hand-over to refine_small_kernel). */
template<typename P>
__global__ __launch_bounds__(64) void esa_finish_kernel(SearchArgs a, int win_budget)
{
    const int j = xcd_remap(blockIdx.x, a.njobs);
    if (j >= a.njobs) return;
    const x265amd_job *job = a.jobs + j;
    JobRegs J;
    J.load(job);
    const int w = uni(job->w), h = uni(job->h);
    EsaGeo g;
    const uint32_t bc = (uint32_t)uni(a.results[j].cost), bi = (uint32_t)uni((int)a.results[j].sad_calls);
    /* refused: a job the scan cannot take, a slot of a CTU launch that holds no PU (w = 0), or one the scan marked */
    if (w == 0 || bc == 0xffffffffu || !esa_job_ok(J, w, h, a.max_dim, win_budget, g, EsaLane<P>::CX, EsaLane<P>::BPS))
    {
        if (threadIdx.x == 0)
        {
            x265amd_result r = { { 0, 0 }, -1, 0, 0 };
            a.results[j] = r;
        }
        return;
    }
    Searcher<P, 64> S;
    SearchState st;
    PlaneDev pfenc, pref[4];
    job_planes(a, job, pfenc, pref);
    S.bind(0, pfenc, pref, a.nref, a.ctab, a.depth, J.pux, J.puy, w, h, a.max_dim, plane_budget(a.max_dim, (int)sizeof(P)));
    S.startup(J, st);
    if (bc < (uint32_t)st.bcost)
    {
        const int by = (int)(bi / (uint32_t)g.nx);
        st.bcost = (int)bc; st.bmy = J.mny + by; st.bmx = J.mnx + (int)(bi - (uint32_t)by * (uint32_t)g.nx);
    }
    S.nsad += (unsigned)(g.nx * g.ny);
    if (a.staged && w == a.staged && h == a.staged)
    {
        /* the sub-pel refinement of these PUs runs in refine_small_kernel: leave it the choice of motion.cpp:1063-1069 */
        int qx, qy, bcost;
        S.choose(st, qx, qy, bcost);
        if (threadIdx.x == 0)
        {
            x265amd_result r = { { (int16_t)qx, (int16_t)qy }, bcost, S.nsad, S.nsub };
            a.results[j] = r;
        }
    }
    else S.finish(a.subme, st, a.results + j);
}

} // namespace x265amd
#endif
