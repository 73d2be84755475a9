/* Rockchip path, stages after CIME: RIME for a whole CTU at once (8-bit planes) and the
 * fractional stage FME for one CU per wave.
 *
 * RIME
 *
 * Reference: source/encoder/motion.cpp:1305-1361 (C-model twin hardwareC/inter.cpp:2521-2560).
 * Every CU of a CTU scans the same three windows of 16 x 12 full-pel positions -- the window
 * centres Pmv[0..2] are per CTU -- and adds the same lambda * bits term per position. Because
 * all CUs move by the same displacement, the SAD of a 16x16, 32x32 or 64x64 CU at one position
 * is the sum of the SADs of the 8x8 blocks it covers. So one workgroup per CTU
 *   1. copies the CTU and the three reference windows into LDS,
 *   2. measures every 8x8 block at every position (v_qsad_pk_u16_u8: four horizontally adjacent
 *      positions per instruction, three vertically adjacent ones sharing the loaded rows),
 *   3. adds the block surfaces up the quadtree,
 *   4. takes, per CU, the first minimum of SAD + lambda * bits in the reference's visit order
 *      (window, row, column; strict '<').
 * The 85 results go to x265amd_rk_result::rime; the fractional stage reads them from there.
 */
#ifndef X265AMD_ME_RK_H
#define X265AMD_ME_RK_H

#include "me_search.h"

namespace x265amd {

#define RIME_THREADS 1024
#define RIME_NCAND 576            /* 3 windows x 12 rows x 16 columns; index = visit order */
#define RIME_WPITCH 84            /* bytes per window row: 64 + 15 columns and the dword a lane reads past them */
#define RIME_WROWS 75             /* 64 + 11 rows */
#define RIME_RR 3                 /* vertically adjacent positions per lane */

enum
{
    RL_FENC = 0,                                  /* 64 x 64 source, pitch 64 */
    RL_MVC = 4096,                                /* int[576]: lambda * bits per position */
    RL_BEST = RL_MVC + 4 * RIME_NCAND,            /* uint[85]: running minimum key per CU */
    RL_WIN = RL_BEST + 4 * 88,                    /* 3 windows */
    RL_S8 = RL_WIN + ((3 * RIME_WROWS * RIME_WPITCH + 15) & ~15),   /* uint16[64 blocks][576] */
    RL_S16 = RL_S8 + 2 * 64 * RIME_NCAND,         /* uint16[16][576] */
    RL_S32 = RL_S16 + 2 * 16 * RIME_NCAND,        /* uint32[4][576] */
    RL_S64 = RL_S32 + 4 * 4 * RIME_NCAND,         /* uint32[576] */
    RL_TOTAL = RL_S64 + 4 * RIME_NCAND
};

__device__ __forceinline__ uint64_t rime_pair(int off)
{
    const uint32_t *q = (const uint32_t*)(x265amd_smem + off);
    return ((uint64_t)q[1] << 32) | q[0];
}

template<typename P>
__global__ __launch_bounds__(RIME_THREADS) void rk_rime_kernel(RkArgs a)
{
    static_assert(sizeof(P) == 1, "the packed quad-SAD path is for 8-bit samples");
    const x265amd_rk_ctu &j = a.ctus[blockIdx.x];
    x265amd_rk_result &res = a.results[blockIdx.x];
    const int tid = threadIdx.x;
    const int pmv[3][2] = { { res.cime.mv[0], res.cime.mv[1] }, { j.pmv[0][0], j.pmv[0][1] }, { j.pmv[1][0], j.pmv[1][1] } };

    /* ---- 1. source CTU, windows, lambda * bits ---- */
    {
        typedef uint32_t __attribute__((aligned(1))) u32u;
        const char AS1 *src = (const char AS1*)a.fenc.origin + (long)j.ctu_y * a.fenc.stride + j.ctu_x;
        {
            const int y = tid >> 4, x = tid & 15;     /* 64 rows x 16 dwords */
            *(uint32_t*)(x265amd_smem + RL_FENC + y * 64 + 4 * x) = *(const u32u AS1*)(src + (long)y * a.fenc.stride + 4 * x);
        }
        const char AS1 *rec = (const char AS1*)a.recon.origin + (long)j.ctu_y * a.recon.stride + j.ctu_x;
        const int wd = RIME_WPITCH / 4;
        for (int i = tid; i < 3 * RIME_WROWS * wd; i += RIME_THREADS)
        {
            const int p = i / (RIME_WROWS * wd), r = (i - p * RIME_WROWS * wd) / wd, d = i - (p * RIME_WROWS + r) * wd;
            const int px = as_i16(pmv[p][0] - 8), py = as_i16(pmv[p][1] - 6);   /* first position of the window */
            *(uint32_t*)(x265amd_smem + RL_WIN + (p * RIME_WROWS + r) * RIME_WPITCH + 4 * d) =
                *(const u32u AS1*)(rec + (long)(py + r) * a.recon.stride + px + 4 * d);
        }
        if (tid < RIME_NCAND)
        {
            /* the smallest of the three predictors' costs, each sum truncated to uint16
             * (motion.cpp:1347-1355) */
            const int p = tid / 192, i = (tid - p * 192) >> 4, jx = tid & 15;
            const int tx = as_i16(pmv[p][0] + jx - 8), ty = as_i16(pmv[p][1] + i - 6);
            int mvc = 0;
            if (j.have_mvd)
            {
                const uint16_t AS1 *ctab = (const uint16_t AS1*)a.ctab;
                mvc = 0x7fffffff;
#pragma unroll
                for (int k = 0; k < 3; k++)
                {
                    const int v = ((int)ctab[tx - j.mvp[k][0]] + (int)ctab[ty - j.mvp[k][1]]) & 0xffff;
                    mvc = v < mvc ? v : mvc;
                }
            }
            lds_i(0, RL_MVC)[tid] = mvc;
        }
        if (tid < 85) ((uint32_t*)(x265amd_smem + RL_BEST))[tid] = 0xffffffffu;
    }
    __syncthreads();

    /* ---- 2. 8x8 block surfaces: item = (block, window, row group of 3, column group of 4) ---- */
    const int nbx = j.valid_w >> 3, nby = j.valid_h >> 3;     /* blocks inside the picture */
    for (int it = tid; it < 64 * 3 * 4 * 4; it += RIME_THREADS)
    {
        const int xg = it & 3, yg = (it >> 2) & 3, pb = it >> 4;
        const int p = pb % 3, blk = pb / 3;
        const int bx = blk & 7, by = blk >> 3;
        if (bx >= nbx || by >= nby) continue;
        const int org = RL_WIN + (p * RIME_WROWS + by * 8 + yg * RIME_RR) * RIME_WPITCH + bx * 8 + xg * 4;
        const int fo = RL_FENC + by * 8 * 64 + bx * 8;
        uint64_t acc[RIME_RR];
#pragma unroll
        for (int c = 0; c < RIME_RR; c++) acc[c] = 0;
#pragma unroll
        for (int col = 0; col < 2; col++)
        {
            uint64_t rows[8 + RIME_RR - 1];
            uint32_t f[8];
#pragma unroll
            for (int k = 0; k < 8 + RIME_RR - 1; k++) rows[k] = rime_pair(org + 4 * col + k * RIME_WPITCH);
#pragma unroll
            for (int k = 0; k < 8; k++) f[k] = *(const uint32_t*)(x265amd_smem + fo + 4 * col + k * 64);
#pragma unroll
            for (int y = 0; y < 8; y++)
#pragma unroll
                for (int c = 0; c < RIME_RR; c++) acc[c] = __builtin_amdgcn_qsad_pk_u16_u8(rows[y + c], f[y], acc[c]);
        }
        /* an 8x8 SAD is at most 16320: the packed 16-bit sums are the surface entries */
#pragma unroll
        for (int c = 0; c < RIME_RR; c++)
            *(uint64_t*)(x265amd_smem + RL_S8 + 2 * (blk * RIME_NCAND + p * 192 + (yg * RIME_RR + c) * 16 + xg * 4)) = acc[c];
    }
    __syncthreads();

    /* ---- 3. up the quadtree ---- */
    const uint16_t *s8 = (const uint16_t*)(x265amd_smem + RL_S8);
    uint16_t *s16 = (uint16_t*)(x265amd_smem + RL_S16);
    uint32_t *s32 = (uint32_t*)(x265amd_smem + RL_S32), *s64 = (uint32_t*)(x265amd_smem + RL_S64);
    for (int i = tid; i < 16 * RIME_NCAND; i += RIME_THREADS)
    {
        const int cu = i / RIME_NCAND, cand = i - cu * RIME_NCAND;
        const int b0 = (cu >> 2) * 16 + (cu & 3) * 2;      /* top-left 8x8 block of the 16x16 CU */
        /* at most 4 x 16320 = 65280: still 16 bits */
        s16[i] = (uint16_t)(s8[b0 * RIME_NCAND + cand] + s8[(b0 + 1) * RIME_NCAND + cand] +
                            s8[(b0 + 8) * RIME_NCAND + cand] + s8[(b0 + 9) * RIME_NCAND + cand]);
    }
    __syncthreads();
    for (int i = tid; i < 4 * RIME_NCAND; i += RIME_THREADS)
    {
        const int cu = i / RIME_NCAND, cand = i - cu * RIME_NCAND;
        const int c0 = (cu >> 1) * 8 + (cu & 1) * 2;       /* top-left 16x16 CU of the 32x32 CU */
        s32[i] = (uint32_t)s16[c0 * RIME_NCAND + cand] + s16[(c0 + 1) * RIME_NCAND + cand] +
                 s16[(c0 + 4) * RIME_NCAND + cand] + s16[(c0 + 5) * RIME_NCAND + cand];
    }
    __syncthreads();
    if (tid < RIME_NCAND) s64[tid] = s32[tid] + s32[RIME_NCAND + tid] + s32[2 * RIME_NCAND + tid] + s32[3 * RIME_NCAND + tid];
    __syncthreads();

    /* ---- 4. per CU: first minimum in visit order; 12 threads share a CU ---- */
    uint32_t *best = (uint32_t*)(x265amd_smem + RL_BEST);
    if (tid < 85 * 12)
    {
        const int idx = tid / 12, part = tid - idx * 12;
        int size, ox, oy;
        if (idx < 64) { size = 8; ox = (idx & 7) * 8; oy = (idx >> 3) * 8; }
        else if (idx < 80) { size = 16; ox = ((idx - 64) & 3) * 16; oy = ((idx - 64) >> 2) * 16; }
        else if (idx < 84) { size = 32; ox = ((idx - 80) & 1) * 32; oy = ((idx - 80) >> 1) * 32; }
        else { size = 64; ox = 0; oy = 0; }
        if (j.ctu_x + ox + size <= a.pic_w && j.ctu_y + oy + size <= a.pic_h)
        {
            const int *mvc = lds_i(0, RL_MVC);
            uint32_t key = 0xffffffffu;
            /* cost < 2^21, so (cost << 10 | visit index) orders like the sequential strict-'<' scan */
            for (int k = 0; k < 48; k++)
            {
                const int cand = part * 48 + k;
                uint32_t sad;
                if (idx < 64) sad = s8[idx * RIME_NCAND + cand];
                else if (idx < 80) sad = s16[(idx - 64) * RIME_NCAND + cand];
                else if (idx < 84) sad = s32[(idx - 80) * RIME_NCAND + cand];
                else sad = s64[cand];
                const uint32_t kk = ((sad + (uint32_t)mvc[cand]) << 10) | (uint32_t)cand;
                key = kk < key ? kk : key;
            }
            atomicMin(&best[idx], key);
        }
    }
    __syncthreads();
    if (tid < 85)
    {
        const uint32_t key = best[tid];
        if (key == 0xffffffffu)
        {
            /* CU outside the picture (TEncCu.cpp:475-493) */
            res.rime[tid].mv[0] = res.rime[tid].mv[1] = 0; res.rime[tid].cost = -1;
        }
        else
        {
            const int cand = key & 1023, p = cand / 192, i = (cand - p * 192) >> 4, jx = cand & 15;
            res.rime[tid].mv[0] = (int16_t)(pmv[p][0] + jx - 8);
            res.rime[tid].mv[1] = (int16_t)(pmv[p][1] + i - 6);
            res.rime[tid].cost = (int)(key >> 10);
        }
    }
}


/* ---- RIME of a whole CTU for 16-bit planes of up to 10 bits per sample ----
 *
 * The same four steps as rk_rime_kernel. v_sad_u16 measures two samples per lane, so a lane's four horizontally
 * adjacent positions are two that start on a dword of the window row and two that start one sample later, measured
 * against the pairs of dwords shifted by one sample (v_alignbyte); the three vertically adjacent positions share every
 * row and its shifted copies. An 8x8 SAD of 10-bit samples still fits 16 bits (64 * 1023), the 16x16 sums do not, and
 * SAD + lambda * bits of a 64x64 CU needs more than the 22 bits a 32-bit (cost, visit index) key has room for: the
 * upper surfaces are 32 bits wide (they reuse the LDS of the source CTU and the windows, which nothing reads after
 * step 2) and the minimum runs on 64-bit keys. */
#define RIME16_WPITCH 164         /* bytes per window row: 64 + 15 samples, a last dword pair read past them, odd number of dwords */
enum
{
    R16_FENC = 0,                                                   /* 64 x 64 source, pitch 128 bytes */
    R16_WIN = 8192,                                                 /* 3 windows */
    R16_S16 = 0,                                                    /* after step 2: uint32[16][576] ... */
    R16_S64 = R16_S16 + 4 * 16 * RIME_NCAND,                        /* ... and uint32[576] over the same bytes */
    R16_MVC = (R16_WIN + 3 * RIME_WROWS * RIME16_WPITCH + 15) & ~15, /* int[576] */
    R16_BEST = R16_MVC + 4 * RIME_NCAND,                            /* uint64[85]: running minimum key per CU */
    R16_S8 = R16_BEST + 8 * 86,                                     /* uint16[64][576] */
    R16_S32 = R16_S8 + 2 * 64 * RIME_NCAND,                         /* uint32[4][576] */
    R16_TOTAL = R16_S32 + 4 * 4 * RIME_NCAND
};
static_assert(R16_S64 + 4 * RIME_NCAND <= R16_MVC, "the upper surfaces fit the bytes of the source CTU and the windows");

__global__ __launch_bounds__(RIME_THREADS) void rk_rime16_kernel(RkArgs a)
{
    const x265amd_rk_ctu &j = a.ctus[blockIdx.x];
    x265amd_rk_result &res = a.results[blockIdx.x];
    const int tid = threadIdx.x;
    const int pmv[3][2] = { { res.cime.mv[0], res.cime.mv[1] }, { j.pmv[0][0], j.pmv[0][1] }, { j.pmv[1][0], j.pmv[1][1] } };
    unsigned long long *best = (unsigned long long*)(x265amd_smem + R16_BEST);

    /* ---- 1. source CTU, windows, lambda * bits ---- */
    {
        typedef uint32_t __attribute__((aligned(2))) u32u;
        const char AS1 *src = (const char AS1*)((const uint16_t AS1*)a.fenc.origin + (long)j.ctu_y * a.fenc.stride + j.ctu_x);
        for (int i = tid; i < 64 * 32; i += RIME_THREADS)
        {
            const int y = i >> 5, x = i & 31;         /* 64 rows x 32 dwords */
            *(uint32_t*)(x265amd_smem + R16_FENC + y * 128 + 4 * x) = *(const u32u AS1*)(src + (long)y * a.fenc.stride * 2 + 4 * x);
        }
        const uint16_t AS1 *rec = (const uint16_t AS1*)a.recon.origin + (long)j.ctu_y * a.recon.stride + j.ctu_x;
        const int wd = RIME16_WPITCH / 4;
        for (int i = tid; i < 3 * RIME_WROWS * wd; i += RIME_THREADS)
        {
            const int p = i / (RIME_WROWS * wd), r = (i - p * RIME_WROWS * wd) / wd, d = i - (p * RIME_WROWS + r) * wd;
            const int px = as_i16(pmv[p][0] - 8), py = as_i16(pmv[p][1] - 6);   /* first position of the window */
            *(uint32_t*)(x265amd_smem + R16_WIN + (p * RIME_WROWS + r) * RIME16_WPITCH + 4 * d) =
                *(const u32u AS1*)((const char AS1*)(rec + (long)(py + r) * a.recon.stride + px) + 4 * d);
        }
        if (tid < RIME_NCAND)
        {
            const int p = tid / 192, i = (tid - p * 192) >> 4, jx = tid & 15;
            const int tx = as_i16(pmv[p][0] + jx - 8), ty = as_i16(pmv[p][1] + i - 6);
            int mvc = 0;
            if (j.have_mvd)
            {
                const uint16_t AS1 *ctab = (const uint16_t AS1*)a.ctab;
                mvc = 0x7fffffff;
#pragma unroll
                for (int k = 0; k < 3; k++)
                {
                    const int v = ((int)ctab[tx - j.mvp[k][0]] + (int)ctab[ty - j.mvp[k][1]]) & 0xffff;
                    mvc = v < mvc ? v : mvc;
                }
            }
            lds_i(0, R16_MVC)[tid] = mvc;
        }
        if (tid < 85) best[tid] = ~0ull;
    }
    __syncthreads();

    /* ---- 2. 8x8 block surfaces: item = (block, window, row group of 3, column group of 4) ---- */
    const int nbx = j.valid_w >> 3, nby = j.valid_h >> 3;     /* blocks inside the picture */
    for (int it = tid; it < 64 * 3 * 4 * 4; it += RIME_THREADS)
    {
        const int xg = it & 3, yg = (it >> 2) & 3, pb = it >> 4;
        const int p = pb % 3, blk = pb / 3;
        const int bx = blk & 7, by = blk >> 3;
        if (bx >= nbx || by >= nby) continue;
        const int org = R16_WIN + (p * RIME_WROWS + by * 8 + yg * RIME_RR) * RIME16_WPITCH + (bx * 8 + xg * 4) * 2;
        const int fo = R16_FENC + by * 8 * 128 + bx * 16;
        uint32_t f[8][4];
#pragma unroll
        for (int y = 0; y < 8; y++)
        {
            const uint64_t lo = *(const uint64_t*)(x265amd_smem + fo + y * 128), hi = *(const uint64_t*)(x265amd_smem + fo + y * 128 + 8);
            f[y][0] = (uint32_t)lo; f[y][1] = (uint32_t)(lo >> 32); f[y][2] = (uint32_t)hi; f[y][3] = (uint32_t)(hi >> 32);
        }
        uint32_t acc[RIME_RR][4];
#pragma unroll
        for (int c = 0; c < RIME_RR; c++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[c][i] = 0;
#pragma unroll
        for (int r = 0; r < 8 + RIME_RR - 1; r++)
        {
            uint32_t d[6], s[5];
#pragma unroll
            for (int k = 0; k < 3; k++)
            {
                const uint64_t pr = rime_pair(org + r * RIME16_WPITCH + 8 * k);
                d[2 * k] = (uint32_t)pr; d[2 * k + 1] = (uint32_t)(pr >> 32);
            }
#pragma unroll
            for (int k = 0; k < 5; k++) s[k] = __builtin_amdgcn_alignbyte(d[k + 1], d[k], 2);
#pragma unroll
            for (int c = 0; c < RIME_RR; c++)
            {
                const int y = r - c;              /* the source row this reference row meets for position row c */
                if (y < 0 || y > 7) continue;
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    acc[c][0] = __builtin_amdgcn_sad_u16(d[k], f[y][k], acc[c][0]);
                    acc[c][1] = __builtin_amdgcn_sad_u16(s[k], f[y][k], acc[c][1]);
                    acc[c][2] = __builtin_amdgcn_sad_u16(d[k + 1], f[y][k], acc[c][2]);
                    acc[c][3] = __builtin_amdgcn_sad_u16(s[k + 1], f[y][k], acc[c][3]);
                }
            }
        }
        /* an 8x8 SAD of 10-bit samples is at most 65472: the surface entries are 16 bits */
#pragma unroll
        for (int c = 0; c < RIME_RR; c++)
            *(uint64_t*)(x265amd_smem + R16_S8 + 2 * (blk * RIME_NCAND + p * 192 + (yg * RIME_RR + c) * 16 + xg * 4)) =
                (uint64_t)(acc[c][0] | (acc[c][1] << 16)) | ((uint64_t)(acc[c][2] | (acc[c][3] << 16)) << 32);
    }
    __syncthreads();

    /* ---- 3. up the quadtree (the source CTU and the windows are dead: their bytes hold the upper surfaces) ---- */
    const uint16_t *s8 = (const uint16_t*)(x265amd_smem + R16_S8);
    uint32_t *s16 = (uint32_t*)(x265amd_smem + R16_S16);
    uint32_t *s32 = (uint32_t*)(x265amd_smem + R16_S32), *s64 = (uint32_t*)(x265amd_smem + R16_S64);
    for (int i = tid; i < 16 * RIME_NCAND; i += RIME_THREADS)
    {
        const int cu = i / RIME_NCAND, cand = i - cu * RIME_NCAND;
        const int b0 = (cu >> 2) * 16 + (cu & 3) * 2;      /* top-left 8x8 block of the 16x16 CU */
        s16[i] = (uint32_t)s8[b0 * RIME_NCAND + cand] + s8[(b0 + 1) * RIME_NCAND + cand] +
                 s8[(b0 + 8) * RIME_NCAND + cand] + s8[(b0 + 9) * RIME_NCAND + cand];
    }
    __syncthreads();
    for (int i = tid; i < 4 * RIME_NCAND; i += RIME_THREADS)
    {
        const int cu = i / RIME_NCAND, cand = i - cu * RIME_NCAND;
        const int c0 = (cu >> 1) * 8 + (cu & 1) * 2;       /* top-left 16x16 CU of the 32x32 CU */
        s32[i] = s16[c0 * RIME_NCAND + cand] + s16[(c0 + 1) * RIME_NCAND + cand] +
                 s16[(c0 + 4) * RIME_NCAND + cand] + s16[(c0 + 5) * RIME_NCAND + cand];
    }
    __syncthreads();
    if (tid < RIME_NCAND) s64[tid] = s32[tid] + s32[RIME_NCAND + tid] + s32[2 * RIME_NCAND + tid] + s32[3 * RIME_NCAND + tid];
    __syncthreads();

    /* ---- 4. per CU: first minimum in visit order; 12 threads share a CU ---- */
    if (tid < 85 * 12)
    {
        const int idx = tid / 12, part = tid - idx * 12;
        int size, ox, oy;
        if (idx < 64) { size = 8; ox = (idx & 7) * 8; oy = (idx >> 3) * 8; }
        else if (idx < 80) { size = 16; ox = ((idx - 64) & 3) * 16; oy = ((idx - 64) >> 2) * 16; }
        else if (idx < 84) { size = 32; ox = ((idx - 80) & 1) * 32; oy = ((idx - 80) >> 1) * 32; }
        else { size = 64; ox = 0; oy = 0; }
        if (j.ctu_x + ox + size <= a.pic_w && j.ctu_y + oy + size <= a.pic_h)
        {
            const int *mvc = lds_i(0, R16_MVC);
            unsigned long long key = ~0ull;
            for (int k = 0; k < 48; k++)
            {
                const int cand = part * 48 + k;
                uint32_t sad;
                if (idx < 64) sad = s8[idx * RIME_NCAND + cand];
                else if (idx < 80) sad = s16[(idx - 64) * RIME_NCAND + cand];
                else if (idx < 84) sad = s32[(idx - 80) * RIME_NCAND + cand];
                else sad = s64[cand];
                const unsigned long long kk = ((unsigned long long)(sad + (uint32_t)mvc[cand]) << 10) | (unsigned)cand;
                key = kk < key ? kk : key;
            }
            atomicMin(&best[idx], key);
        }
    }
    __syncthreads();
    if (tid < 85)
    {
        const unsigned long long key = best[tid];
        if (key == ~0ull)
        {
            /* CU outside the picture (TEncCu.cpp:475-493) */
            res.rime[tid].mv[0] = res.rime[tid].mv[1] = 0; res.rime[tid].cost = -1;
        }
        else
        {
            const int cand = (int)(key & 1023), p = cand / 192, i = (cand - p * 192) >> 4, jx = cand & 15;
            res.rime[tid].mv[0] = (int16_t)(pmv[p][0] + jx - 8);
            res.rime[tid].mv[1] = (int16_t)(pmv[p][1] + i - 6);
            res.rime[tid].cost = (int)(key >> 10);
        }
    }
}

/* ---- FME: the 7 x 7 quarter-pel grid around the RIME vector (motion.cpp:1364-1403; filters
 * InterpHoriz / InterpVert / FilterHorizontal + FilterVertical with N = 4, :1495-1648) ----
 *
 * The grid has seven x offsets and seven y offsets. One lane owns a (x offset, 4x4 tile) pair:
 * it runs the horizontal 4-tap pass once, over the eight reference rows the seven y offsets
 * need between them, and then the vertical pass + Hadamard cost for each y offset. The
 * reference region (RK_REGION_COLS x RK_REGION_ROWS samples, first sample at RIME vector
 * - (2, 2)) and the source block are in LDS. costs[i * 7 + j] is left in the wave's L_COSTS
 * array, i = x offset index (outer in the reference's visit order), j = y offset index. */
template<typename P, bool FENC_GLOBAL>
__device__ __forceinline__ void rk_fme_grid(int base, int region, int pitch, int size, int bxq, int byq, int depth,
                                            const P AS1 *fenc_g, int fenc_stride)
{
    const int lane = threadIdx.x & 63;
    const int wt = size >> 2, T = wt * wt;
    int lT = 0;
    while ((1 << lT) < T) lT++;
    const int lG = lT < 6 ? lT : 6;       /* lanes that share a candidate within one pass */
    const int maxv = (1 << depth) - 1;
    const int headRoom = 14 - depth;
    const int shiftH = 6 - headRoom, offH = -(8192 << shiftH);
    const int shiftV = 6 + headRoom, offV = (1 << (shiftV - 1)) + (8192 << 6);
    if (lane < 49) lds_i(base, L_COSTS)[lane] = 0;
    wave_sync();
    /* An 8x8 CU has 28 (x offset, tile) pairs. Both halves of the wave take them and share out the seven y offsets:
     * lanes 0-31 take offsets 0 .. 3, lanes 32-63 offsets 4 .. 6 one region row lower, where they use the same rows
     * of their registers and the same phases as offsets 0 .. 2 -- four vertical passes instead of seven. */
    const bool split = T == 4;
    const int hs = split && lane >= 32 ? 1 : 0;
    for (int task0 = 0; task0 < 7 * T; task0 += 64)
    {
        const int task = split ? (lane & 31) : task0 + lane;
        const bool live = task < 7 * T;
        const int i = live ? task >> lT : 0, t = task & (T - 1);
        const int ty = t >> (lT >> 1), tx = t & (wt - 1);
        const int qx = as_i16(bxq + i - 3), xf = qx & 3;
        /* region column of the sample one to the left of the tile's first output */
        const int col0 = ((qx >> 2) - (bxq >> 2) + 1) + tx * 4;
        typename Px<P>::quad fq[4];
#pragma unroll
        for (int y = 0; y < 4; y++)
            fq[y] = FENC_GLOBAL ? Px<P>::load_g(fenc_g + (long)(ty * 4 + y) * fenc_stride + tx * 4)
                                : Px<P>::load_lds_aligned(base + L_FENC + ((ty * 4 + y) * 64 + tx * 4) * (int)sizeof(P));
        if constexpr (sizeof(P) == 1)
        {
            /* 8-bit samples. Horizontal pass on v_dot4_i32_i8: samples biased to signed bytes, and as
             * the taps of every phase (phase 0 = {0, 64, 0, 0} included) add up to 64, the result is
             * D = sum - 8192, which is the reference's 16-bit intermediate (shift 0 at this depth).
             * Everything else follows from D without telling the phases apart:
             *   yf != 0: (sum_k cv[k] * D[k] + offV) >> 12 -- for xf == 0 that is (64 S + 2048) >> 12
             *            = (S + 32) >> 6 with S the vertical sum over samples, the one-pass rounding;
             *   yf == 0: (D + 8192 + 32) >> 6, the one-pass horizontal rounding (the sample for xf == 0).
             * The vertical pass takes rows in pairs through v_dot2_i32_i16. */
            const uint32_t tH = kRkTapBytes[xf];
            const int off0 = base + region + (ty * 4 * pitch + col0);
            const uint32_t sh = (uint32_t)(off0 & 3);     /* the pitch is a multiple of 4: the same for every row */
            int D[8][4];
#pragma unroll
            for (int r = 0; r < 8; r++)
            {
                const int rr = r + hs < 8 ? r + hs : 7;   /* the shifted half never uses its last row */
                const uint32_t *q = (const uint32_t*)(x265amd_smem + ((off0 + rr * pitch) & ~3));
                const uint32_t q0 = q[0], q1 = q[1], q2 = q[2];
                const uint32_t d0 = __builtin_amdgcn_alignbyte(q1, q0, sh) ^ 0x80808080u, d1 = __builtin_amdgcn_alignbyte(q2, q1, sh) ^ 0x80808080u;
                D[r][0] = __builtin_amdgcn_sdot4(d0, tH, 0, false);
#pragma unroll
                for (int x = 1; x < 4; x++) D[r][x] = __builtin_amdgcn_sdot4(__builtin_amdgcn_alignbyte(d1, d0, x), tH, 0, false);
            }
            uint32_t pk[7][4];
#pragma unroll
            for (int r = 0; r < 7; r++)
#pragma unroll
                for (int x = 0; x < 4; x++) pk[r][x] = __builtin_amdgcn_perm((uint32_t)D[r + 1][x], (uint32_t)D[r][x], 0x05040100u);
#pragma unroll
            for (int k = 0; k < 7; k++)
            {
                if (split && k >= 4) break;
                /* y offset of this pass: k, or with the wave split k for the first half and k + 4 for the second, whose
                 * rows sit one lower in its registers: offsets 4 .. 6 then read like 0 .. 2 */
                const int jj = k + 4 * hs;
                const bool mine = live && jj < 7;
                const int yf = (k + 1) & 3, o = k < 3 || split ? 0 : 1;
                int pr[4][4];
#pragma unroll
                for (int y = 0; y < 4; y++)
#pragma unroll
                    for (int x = 0; x < 4; x++)
                    {
                        const int sum = yf ? dot2(pk[y + o][x], kRkTapPairs[yf][0], dot2(pk[y + o + 2][x], kRkTapPairs[yf][1], offV)) >> shiftV
                                           : (D[y + 2][x] + 8192 + 32) >> 6;
                        pr[y][x] = clip_pixel(sum, maxv);
                    }
                int v = mine ? tile_cost_px<P>(fq, pr, true) : 0;
                v = seg_sum(v, lG);
                if (mine && (lane & ((1 << lG) - 1)) == 0) lds_i(base, L_COSTS)[i * 7 + jj] += v;
                if (T > 64) wave_sync();   /* several passes add to the same entry */
            }
            if (split) break;
            continue;
        }
        int ch[4];
#pragma unroll
        for (int k = 0; k < 4; k++) ch[k] = kRkTaps[xf][k];
        /* horizontal pass over region rows ty*4 .. ty*4 + 7: raw sums, or the samples when xf == 0 */
        int hr[8][4];
#pragma unroll
        for (int r = 0; r < 8; r++)
        {
            const int rr = r + hs < 8 ? r + hs : 7;
            const int off = base + region + ((ty * 4 + rr) * pitch + col0) * (int)sizeof(P);
            int px[8];
            Px<P>::unpack(Px<P>::load_lds(off), px);
            Px<P>::unpack(Px<P>::load_lds(off + 4 * (int)sizeof(P)), px + 4);
#pragma unroll
            for (int x = 0; x < 4; x++)
                hr[r][x] = xf ? px[x] * ch[0] + px[x + 1] * ch[1] + px[x + 2] * ch[2] + px[x + 3] * ch[3] : px[x + 1];
        }
        /* what the vertical filter reads: the 16-bit intermediates of the two-pass case, or samples */
        int hv[8][4];
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int x = 0; x < 4; x++) hv[r][x] = xf ? as_i16((hr[r][x] + offH) >> shiftH) : hr[r][x];
        const int sv = xf ? shiftV : 6, ov = xf ? offV : 32;
        for (int k = 0; k < (split ? 4 : 7); k++)
        {
            const int jj = k + 4 * hs;
            const bool mine = live && jj < 7;
            const int yf = (k + 1) & 3;
            int pr[4][4];   /* the predicted tile */
            if (yf == 0)
            {
                /* integer row: luma-hpp style rounding of the raw horizontal sum, or the sample itself */
#pragma unroll
                for (int y = 0; y < 4; y++)
#pragma unroll
                    for (int x = 0; x < 4; x++)
                        pr[y][x] = xf ? clip_pixel(as_i16((hr[y + 2][x] + 32) >> 6), maxv) : hr[y + 2][x];
            }
            else
            {
                int cv[4];
#pragma unroll
                for (int k = 0; k < 4; k++) cv[k] = kRkTaps[yf][k];
                if (k < 3 || split)
                {
#pragma unroll
                    for (int y = 0; y < 4; y++)
#pragma unroll
                        for (int x = 0; x < 4; x++)
                        {
                            const int sum = hv[y][x] * cv[0] + hv[y + 1][x] * cv[1] + hv[y + 2][x] * cv[2] + hv[y + 3][x] * cv[3];
                            pr[y][x] = clip_pixel(as_i16((sum + ov) >> sv), maxv);
                        }
                }
                else
                {
#pragma unroll
                    for (int y = 0; y < 4; y++)
#pragma unroll
                        for (int x = 0; x < 4; x++)
                        {
                            const int sum = hv[y + 1][x] * cv[0] + hv[y + 2][x] * cv[1] + hv[y + 3][x] * cv[2] + hv[y + 4][x] * cv[3];
                            pr[y][x] = clip_pixel(as_i16((sum + ov) >> sv), maxv);
                        }
                }
            }
            int v = mine ? tile_cost_px<P>(fq, pr, true) : 0;
            v = seg_sum(v, lG);
            if (mine && (lane & ((1 << lG) - 1)) == 0) lds_i(base, L_COSTS)[i * 7 + jj] += v;
            if (T > 64) wave_sync();   /* several passes add to the same entry */
        }
        if (split) break;
    }
    wave_sync();
}

} // namespace x265amd
#endif
