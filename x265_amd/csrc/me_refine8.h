/* 8x8 PUs four to a wave (8-bit samples, full-resolution planes): the sub-pel refinement (refine8_kernel) and the common
 * case of the STAR integer stage (star8_kernel, further down).
 *
 * The refinement rounds of MotionEstimate::motionEstimate (motion.cpp:1071-1141) have a fixed shape: the
 * neighbours of one centre at one distance, in the order of the square table, each an 8-tap interpolation
 * plus SAD or SATD (subpelCompare, motion.cpp:1651-1689). An 8x8 PU has four 4x4 tiles, so a wave that owns
 * one PU runs its passes a quarter full and pays its whole bookkeeping per PU. Here a wave owns four PUs, one
 * per row of 16 lanes: what is uniform per PU (centre, best cost, predictor) is a value every lane of the row
 * holds, decisions are row-wide minima (DPP), and the filter passes are filled by all four PUs.
 *
 * Per round and PU the three columns of candidates (x - step, x, x + step) share their horizontal pass: one
 * plane of luma_hps intermediates (ipfilter.cpp:235-278) per column, 16 rows x 8 samples, int16, column-major
 * in LDS. With 8-bit samples that intermediate keeps every bit, luma_vsp (:359-400) with the phase-0 taps on it
 * gives exactly luma_hpp's (sum + 32) >> 6, and the phase-0 intermediate of a sample is 64 * sample - 8192, on
 * which luma_vsp equals luma_vpp: one plane kind serves every fraction, as in eval_planes. The vertical pass
 * and the cost run one lane per (tile, column) in plane_column_cost, one candidate of every PU per pass.
 *
 * The search kernel leaves (mv, cost, counters) of the full-pel / predictor choice (motion.cpp:1063-1069) in
 * the PU's result record; this kernel refines that record in place. PUs of other sizes were finished by the
 * search kernel and are skipped.
 */
#ifndef X265AMD_ME_REFINE8_H
#define X265AMD_ME_REFINE8_H

namespace x265amd {

#define R8_PROW 18                          /* int16 rows of a plane column: 16 filled + slack for an odd first row */
#define R8_PLANE (8 * R8_PROW * 2)          /* bytes of one plane */
#define R8_PU_BYTES (3 * R8_PLANE + 4)      /* planes of one PU; the odd dword staggers the PUs over the LDS banks */
enum
{
    R8_TAPS = 0,                            /* 48 dwords of packed filter taps (TapWords) */
    R8_PARAMS = 192,                        /* per PU, 48 bytes: 3 source offsets, row pitch, 3 phases, -, plane origin (64 bit) */
    R8_PLANES = 192 + 4 * 48
};
#define R8_LDS (R8_PLANES + 4 * R8_PU_BYTES + 32)

/* value of lane 0 of the own row, in every lane of the row */
__device__ __forceinline__ int row_first(int v) { return __shfl(v, (int)(threadIdx.x & 48)); }

struct Refine8
{
    /* per row (= per PU), the same in its 16 lanes */
    int qx, qy, bcost, mvpx, mvpy, roff;
    unsigned nsub;
    bool active;
    const uint8_t AS1 *ref0;    /* origin of the PU's reference plane (PUs of one wave may search different pictures) */
    int rstride;
    /* per lane */
    int lane, row, l, f[4];
    const uint16_t AS1 *ctab;

    /* One refinement round for all four PUs: `dirs` neighbours of (qx, qy) at distance `step`, plus the
     * centre itself first when `centre` (the SATD re-measure of motion.cpp:1086, :1117, whose cost replaces
     * bcost). dirs = 0 with centre measures the centre alone (the closing SATD of subme 0). */
    __device__ __forceinline__ void round(int step, int dirs, bool satd, bool centre)
    {
        constexpr unsigned long long SQX = nibs({ 0, 0, 0, -1, 1, -1, -1, 1, 1 }), SQY = nibs({ 0, -1, 1, 0, 0, -1, 1, -1, 1 });
        const int k0 = centre ? 1 : 0, K = k0 + dirs;
        /* lambda * bits of the candidate this lane folds (bitcost.h:46: the sum wraps at 16 bits) */
        int mvc = 0;
        {
            const int i = l < K ? l + 1 - k0 : 0;
            const int tx = as_i16(qx + nib(SQX, i) * step), ty = as_i16(qy + nib(SQY, i) * step);
            if (l < K && active) mvc = ((int)ctab[tx - mvpx] + (int)ctab[ty - mvpy]) & 0xffff;
        }
        /* rows of the planes: the candidates' first rows are Ymin or Ymin + 1 */
        const int ya = as_i16(qy - step) >> 2, yb = qy >> 2, yc = as_i16(qy + step) >> 2;
        int ymin = ya < yb ? ya : yb;
        ymin = yc < ymin ? yc : ymin;
        wave_sync();
        if (l < 3)
        {
            const int xp = as_i16(qx + (l - 1) * step);
            int *prm = (int*)(x265amd_smem + R8_PARAMS + row * 48);
            prm[l] = roff + __mul24(ymin - 3, rstride) + (xp >> 2) - 3;
            prm[4 + l] = xp & 3;
            if (l == 0)
            {
                prm[3] = rstride;
                *(unsigned long long*)(prm + 8) = (unsigned long long)ref0;
            }
        }
        wave_sync();
        /* horizontal pass: 96 items per PU (3 planes x 16 rows x 2 quads of samples), 384 per wave */
#pragma unroll 2
        for (int pass = 0; pass < 6; pass++)
        {
            const int item = pass * 64 + lane;
            const int pu = (item * 683) >> 16, rem = item - pu * 96;
            const int p = rem >> 5, r = (rem >> 1) & 15, q = rem & 1;
            const int *prm = (const int*)(x265amd_smem + R8_PARAMS + pu * 48);
            const int *tp = (const int*)(x265amd_smem + R8_TAPS) + prm[4 + p] * 12;
            const uint8_t AS1 *s = (const uint8_t AS1*)*(const unsigned long long*)(prm + 8) + prm[p] + __mul24(r, prm[3]) + 4 * q;
            /* samples biased to signed bytes for v_dot4_i32_i8; the taps of every phase add up to 64, so the sum
             * comes out 8192 low, which is exactly luma_hps's offset (ipfilter.cpp:246-247 with 8-bit samples) */
            const uint32_t d0 = Px<uint8_t>::load_g(s) ^ 0x80808080u, d1 = Px<uint8_t>::load_g(s + 4) ^ 0x80808080u,
                           d2 = Px<uint8_t>::load_g(s + 8) ^ 0x80808080u;
            const int tlo = tp[9], thi = tp[10];
            int16_t *dst = (int16_t*)(x265amd_smem + R8_PLANES + pu * R8_PU_BYTES + p * R8_PLANE) + (4 * q) * R8_PROW + r;
            dst[0] = (int16_t)__builtin_amdgcn_sdot4(d0, tlo, __builtin_amdgcn_sdot4(d1, thi, 0, false), false);
#pragma unroll
            for (int x = 1; x < 4; x++)
                dst[x * R8_PROW] = (int16_t)__builtin_amdgcn_sdot4(__builtin_amdgcn_alignbyte(d1, d0, x), tlo,
                                                                   __builtin_amdgcn_sdot4(__builtin_amdgcn_alignbyte(d2, d1, x), thi, 0, false), false);
        }
        wave_sync();
        /* vertical pass + cost: candidate c of every PU per pass, a lane per (tile, column) */
        ColTile<uint8_t> ct;
        ct.ty = l >> 3; ct.col = ((l >> 2) & 1) * 4 + (l & 3);
#pragma unroll
        for (int y = 0; y < 4; y++) ct.f[y] = f[y];
        int mycost = 0;
        for (int c = 0; c < K; c++)
        {
            const int i = c + 1 - k0;
            const int dx = nib(SQX, i), dy = nib(SQY, i);
            const int ty = as_i16(qy + dy * step);
            ColCand cc;
            cc.poff = R8_PLANES + row * R8_PU_BYTES + (dx + 1) * R8_PLANE;
            cc.prow = R8_PROW;
            cc.R0 = (ty >> 2) - ymin;
            cc.shiftV = 12;                                  /* 6 + headroom of 8-bit samples */
            cc.offV = (1 << 11) + (8192 << 6);
            cc.satd = satd ? 1 : 0;
            const int *tp = (const int*)(x265amd_smem + R8_TAPS) + (ty & 3) * 12;
#pragma unroll
            for (int k = 0; k < 4; k++) cc.ev[k] = (uint32_t)tp[k];
#pragma unroll
            for (int k = 0; k < 5; k++) cc.od[k] = (uint32_t)tp[4 + k];
            int v = plane_column_cost<uint8_t>(0, cc, ct, l & 3, 8);
            v = seg_sum(v, 4);
            if (l == c) mycost = (satd ? v >> 1 : v) + mvc;   /* every 4x4 abs-sum is even: halving the total = halving per tile */
        }
        /* the reference's strict '<' over the candidates in table order */
        if (centre && active) bcost = row_first(mycost);
        int key = 0x7fffffff;
        if (l >= k0 && l < K) key = (mycost << 6) | l;
        key = row_min16(key);
        int bdir = 0;
        if (active && (key >> 6) < bcost) { bcost = key >> 6; bdir = (key & 63) - k0 + 1; }
        qx = as_i16(qx + nib(SQX, bdir) * step); qy = as_i16(qy + nib(SQY, bdir) * step);
        if (active) nsub += (unsigned)K;
    }
};

__global__ __launch_bounds__(64) void refine8_kernel(SearchArgs a)
{
    const int ngroups = (a.njobs + 3) >> 2;
    const int g = xcd_remap((int)blockIdx.x, ngroups);
    if (g >= ngroups) return;
    Refine8 R;
    R.lane = (int)threadIdx.x; R.row = R.lane >> 4; R.l = R.lane & 15;
    int j = g * 4 + R.row;
    const bool valid = j < a.njobs;
    if (!valid) j = a.njobs - 1;
    const int *job = (const int*)(a.jobs + j);
    const int d0 = job[0], d1 = job[1], d4 = job[4];
    const int pux = as_i16(d0), puy = d0 >> 16;
    int *res = (int*)(a.results + j);
    const int r0 = res[0];
    R.qx = as_i16(r0); R.qy = r0 >> 16; R.bcost = res[1]; R.nsub = (unsigned)res[3];
    R.mvpx = as_i16(d4); R.mvpy = d4 >> 16;
    /* a PU whose full-pel cost is 0 is not refined (motion.cpp:1073-1076); PUs of other sizes are final already */
    R.active = valid && (d1 & 0xffff) == 0x0808 && R.bcost != 0;
    if (__ballot(R.active) == 0) return;
    /* the planes of the row's PU: the launch's own, or its entry of the pair table (as job_planes) */
    const uint8_t AS1 *fenc0 = (const uint8_t AS1*)a.fenc.origin;
    int fstride = a.fenc.stride;
    R.ref0 = (const uint8_t AS1*)a.ref[0].origin;
    R.rstride = a.ref[0].stride;
    if (a.pairs)
    {
        int pi = (d1 >> 24) & 0xff;
        if (pi >= a.npairs) pi = 0;
        const PairDev *p = a.pairs + pi;
        fenc0 = (const uint8_t AS1*)p->fenc.origin; fstride = p->fenc.stride;
        R.ref0 = (const uint8_t AS1*)p->ref.origin; R.rstride = p->ref.stride;
    }
    R.roff = puy * R.rstride + pux;
    R.ctab = (const uint16_t AS1*)a.ctab;
    if (R.lane < 48) ((int*)(x265amd_smem + R8_TAPS))[R.lane] = kTapWords.w[R.lane];
    {
        /* the lane's column of its tile of the source block */
        const int t = R.l >> 2;
        const uint8_t AS1 *src = fenc0 + (long)(puy + (t >> 1) * 4) * fstride + pux + (t & 1) * 4 + (R.l & 3);
#pragma unroll
        for (int y = 0; y < 4; y++) R.f[y] = (int)src[(long)y * fstride];
    }
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
}

/* ---- the common case of a STAR search of an 8x8 PU, four PUs per wave ----
 *
 * Most PUs of a picture end their integer search where it began: the start-up candidates (motion.cpp:562-621)
 * pick a full-pel start, none of the 20 points of the first three star rounds (distance 1, 2, 4; StarPatternSearch,
 * :307-547) beats it, three rounds without improvement end the pattern (earlyExitIters = 3, :914) and with
 * bDistance == 0 nothing else runs (:916-1017). For such a PU the order of the points does not matter -- only that
 * none is cheaper -- so the whole stage has a fixed shape, and this kernel runs it for four PUs per wave, a PU per
 * row of 16 lanes: the start-up candidates as one plane + one vertical pass each (all SAD; full-pel ones are the
 * phase-0 case), the decisions of :575-621 on row-uniform values, then the 20 points as plain SADs, four points x
 * four tiles per pass. A PU one of whose points improves is left to search_queue_kernel (the one-wave search from
 * the start), as are the PUs of other sizes; their indices go to a queue in device memory. The others get their
 * record of the full-pel / predictor choice (:1063-1069) written here, ready for refine8_kernel.
 *
 * queue[0] = number of entries, queue[1] unused, queue[2 ...] = job index | (1 << 31 when the PU is not 8x8 and wants
 * the complete search). */
#define S8_SLOTS 7                                  /* clipped predictor, its full-pel rounding, MV 0, four candidates */
#define S8_PU_BYTES (S8_SLOTS * R8_PLANE + 4)
enum
{
    S8_TAPS = 0,
    S8_PU = 192,                                    /* per PU, 16 bytes: plane origin (64 bit), row pitch, candidates in use */
    S8_CAND = 192 + 4 * 16,                         /* per PU and candidate, 8 bytes: source offset of the plane, x phase */
    S8_PLANES = 192 + 4 * 16 + 4 * S8_SLOTS * 8
};
#define S8_LDS (S8_PLANES + 4 * S8_PU_BYTES + 32)

__device__ __forceinline__ int row_lane(int v, int k) { return __shfl(v, (int)(threadIdx.x & 48) + k); }

__global__ __launch_bounds__(64) void star8_kernel(SearchArgs a, int *queue)
{
    const int ngroups = (a.njobs + 3) >> 2;
    const int g = xcd_remap((int)blockIdx.x, ngroups);
    if (g >= ngroups) return;
    const int lane = (int)threadIdx.x, row = lane >> 4, l = lane & 15;
    int j = g * 4 + row;
    const bool valid = j < a.njobs;
    if (!valid) j = a.njobs - 1;
    const int *job = (const int*)(a.jobs + j);
    int d[10];
#pragma unroll
    for (int i = 0; i < 10; i++) d[i] = job[i];
    const bool is88 = (d[1] & 0xffff) == 0x0808;
    if (valid && !is88 && l == 0) queue[2 + atomicAdd(&queue[0], 1)] = j | (int)0x80000000;
    const bool act = valid && is88;
    if (__ballot(act) == 0) return;

    const int pux = as_i16(d[0]), puy = d[0] >> 16, numc = (d[1] >> 16) & 0xff;
    const int mnx = as_i16(d[2]), mny = d[2] >> 16, mxx = as_i16(d[3]), mxy = d[3] >> 16;
    const int mvpx = as_i16(d[4]), mvpy = d[4] >> 16;
    const uint8_t AS1 *fenc0 = (const uint8_t AS1*)a.fenc.origin, *ref0 = (const uint8_t AS1*)a.ref[0].origin;
    int fstride = a.fenc.stride, rstride = a.ref[0].stride;
    if (a.pairs)
    {
        int pi = (d[1] >> 24) & 0xff;
        if (pi >= a.npairs) pi = 0;
        const PairDev *p = a.pairs + pi;
        fenc0 = (const uint8_t AS1*)p->fenc.origin; fstride = p->fenc.stride;
        ref0 = (const uint8_t AS1*)p->ref.origin; rstride = p->ref.stride;
    }
    const int roff = puy * rstride + pux;
    const uint16_t AS1 *ctab = (const uint16_t AS1*)a.ctab;
    if (lane < 48) ((int*)(x265amd_smem + S8_TAPS))[lane] = kTapWords.w[lane];

    /* ---- start-up candidates, motion.cpp:575-621 (values every lane of the row holds) ---- */
    const int qmnx = as_i16(mnx << 2), qmny = as_i16(mny << 2), qmxx = as_i16(mxx << 2), qmxy = as_i16(mxy << 2);
#define S8_CLIPX(v) ((v) > qmxx ? qmxx : ((v) < qmnx ? qmnx : (v)))
#define S8_CLIPY(v) ((v) > qmxy ? qmxy : ((v) < qmny ? qmny : (v)))
    const int pmx = S8_CLIPX(mvpx), pmy = S8_CLIPY(mvpy);
    int bmx = as_i16((pmx + 2) >> 2), bmy = as_i16((pmy + 2) >> 2);
    const bool fpOn = ((pmx | pmy) & 3) != 0, zeroOn = (pmx | pmy) != 0;
    int ccx[4], ccy[4];
    bool cOn[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        ccx[i] = S8_CLIPX(as_i16(d[5 + i])); ccy[i] = S8_CLIPY(d[5 + i] >> 16);
        cOn[i] = i < numc && (ccx[i] | ccy[i]) != 0 && !(ccx[i] == pmx && ccy[i] == pmy);
    }
#undef S8_CLIPX
#undef S8_CLIPY
    /* lane l < 7 of a row looks after candidate slot l: where it is measured (quarter-pel) and whether it is in use.
     * A row without an 8x8 PU measures slot 0 only, so that every address below stays defined. */
    int sx = pmx, sy = pmy;
    bool on = l == 0;
    if (l == 1) { sx = as_i16(bmx << 2); sy = as_i16(bmy << 2); on = fpOn; }
    if (l == 2) { sx = 0; sy = 0; on = zeroOn; }
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (l == 3 + i) { sx = ccx[i]; sy = ccy[i]; on = cOn[i]; }
    on = l < S8_SLOTS && (act ? on : l == 0);
    const int rowbits = (int)((__ballot(on) >> (row * 16)) & 0x7f);
    const int cidx = __builtin_popcount(rowbits & ((1 << l) - 1)), nact = __builtin_popcount(rowbits);
    int kmax = __builtin_amdgcn_readlane(nact, 0);
    {
        const int n1 = __builtin_amdgcn_readlane(nact, 16), n2 = __builtin_amdgcn_readlane(nact, 32), n3 = __builtin_amdgcn_readlane(nact, 48);
        kmax = n1 > kmax ? n1 : kmax; kmax = n2 > kmax ? n2 : kmax; kmax = n3 > kmax ? n3 : kmax;
    }
    /* lambda * bits of the slot (bitcost.h:46); the clipped predictor itself carries none (motion.cpp:579-582) */
    int mvc = 0;
    if (on && act && l != 0) mvc = ((int)ctab[sx - mvpx] + (int)ctab[sy - mvpy]) & 0xffff;
    if (on)
    {
        int *cd = (int*)(x265amd_smem + S8_CAND + (row * S8_SLOTS + cidx) * 8);
        cd[0] = roff + __mul24((sy >> 2) - 3, rstride) + (sx >> 2) - 3;
        cd[1] = sx & 3;
    }
    if (l == 0)
    {
        int *pu = (int*)(x265amd_smem + S8_PU + row * 16);
        *(unsigned long long*)pu = (unsigned long long)ref0;
        pu[2] = rstride; pu[3] = nact;
    }
    wave_sync();
    /* horizontal pass: candidate k of every PU, 16 rows x 2 quads of samples each */
    for (int k = 0; k < kmax; k++)
#pragma unroll
        for (int hh = 0; hh < 2; hh++)
        {
            const int item = hh * 64 + lane;
            const int pu = item >> 5, r = (item >> 1) & 15, q = item & 1;
            const int *pp = (const int*)(x265amd_smem + S8_PU + pu * 16);
            const int kk = k < pp[3] ? k : pp[3] - 1;
            const int *cd = (const int*)(x265amd_smem + S8_CAND + (pu * S8_SLOTS + kk) * 8);
            const int *tp = (const int*)(x265amd_smem + S8_TAPS) + cd[1] * 12;
            const uint8_t AS1 *s = (const uint8_t AS1*)*(const unsigned long long*)pp + cd[0] + __mul24(r, pp[2]) + 4 * q;
            const uint32_t d0 = Px<uint8_t>::load_g(s) ^ 0x80808080u, d1 = Px<uint8_t>::load_g(s + 4) ^ 0x80808080u,
                           d2 = Px<uint8_t>::load_g(s + 8) ^ 0x80808080u;
            const int tlo = tp[9], thi = tp[10];
            int16_t *dst = (int16_t*)(x265amd_smem + S8_PLANES + pu * S8_PU_BYTES + kk * R8_PLANE) + (4 * q) * R8_PROW + r;
            dst[0] = (int16_t)__builtin_amdgcn_sdot4(d0, tlo, __builtin_amdgcn_sdot4(d1, thi, 0, false), false);
#pragma unroll
            for (int x = 1; x < 4; x++)
                dst[x * R8_PROW] = (int16_t)__builtin_amdgcn_sdot4(__builtin_amdgcn_alignbyte(d1, d0, x), tlo,
                                                                   __builtin_amdgcn_sdot4(__builtin_amdgcn_alignbyte(d2, d1, x), thi, 0, false), false);
        }
    wave_sync();
    /* vertical pass + SAD: the k-th candidate in use of every PU per pass */
    ColTile<uint8_t> ct;
    ct.ty = l >> 3; ct.col = ((l >> 2) & 1) * 4 + (l & 3);
    {
        const int t = l >> 2;
        const uint8_t AS1 *src = fenc0 + (long)(puy + (t >> 1) * 4) * fstride + pux + (t & 1) * 4 + (l & 3);
#pragma unroll
        for (int y = 0; y < 4; y++) ct.f[y] = (int)src[(long)y * fstride];
    }
    int mycost = 0;
    {
        int rem = rowbits;
        for (int k = 0; k < kmax; k++)
        {
            const int sk = rem ? __builtin_ctz(rem) : 15;
            rem &= rem - 1;
            const int ysk = row_lane(sy, sk < S8_SLOTS ? sk : 0);
            ColCand cc;
            cc.poff = S8_PLANES + row * S8_PU_BYTES + (k < nact ? k : nact - 1) * R8_PLANE;
            cc.prow = R8_PROW; cc.R0 = 0; cc.shiftV = 12; cc.offV = (1 << 11) + (8192 << 6); cc.satd = 0;
            const int *tp = (const int*)(x265amd_smem + S8_TAPS) + (ysk & 3) * 12;
#pragma unroll
            for (int i = 0; i < 4; i++) cc.ev[i] = (uint32_t)tp[i];
#pragma unroll
            for (int i = 0; i < 5; i++) cc.od[i] = (uint32_t)tp[4 + i];
            int v = plane_column_cost<uint8_t>(0, cc, ct, l & 3, 8);
            v = seg_sum(v, 4);
            if (l == sk) mycost = v + mvc;
        }
    }
    /* the reference's sequence of tests on the measured costs (motion.cpp:579-621) */
    int bprecost = row_lane(mycost, 0), bcost = bprecost;
    unsigned nsad = 0, nsub = 1;
    {
        const int cFp = row_lane(mycost, 1), cZero = row_lane(mycost, 2);
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
        const int c = row_lane(mycost, 3 + i);
        if (cOn[i] && !(ccx[i] == bpx && ccy[i] == bpy))
        {
            nsub++;
            if (c < bprecost) { bprecost = c; bpx = ccx[i]; bpy = ccy[i]; }
        }
    }
    /* ---- the first three star rounds around (bmx, bmy): 4 + 8 + 8 points (motion.cpp:330-452), does any beat bcost? ---- */
    constexpr unsigned long long X4 = nibs({ 0, -1, 1, 0 }), Y4 = nibs({ -1, 0, 0, 1 });
    constexpr unsigned long long X8 = nibs({ 0, -1, 1, -2, 2, -1, 1, 0 }), Y8 = nibs({ -2, -1, -1, 0, 0, 1, 1, 2 });
    uint32_t fq[4];
    {
        const int t = l & 3;
        const uint8_t AS1 *src = fenc0 + (long)(puy + (t >> 1) * 4) * fstride + pux + (t & 1) * 4;
#pragma unroll
        for (int y = 0; y < 4; y++) fq[y] = Px<uint8_t>::load_g(src + (long)y * fstride);
    }
    bool improve = false;
    unsigned npts = 0;
#pragma unroll 1
    for (int pass = 0; pass < 5; pass++)
    {
        const int p = pass * 4 + (l >> 2);
        const int r = p < 4 ? 0 : (p < 12 ? 1 : 2), i = p < 4 ? p : (p - 4) & 7;
        const int unit = r == 2 ? 2 : 1;
        const int dx = nib(r == 0 ? X4 : X8, i) * unit, dy = nib(r == 0 ? Y4 : Y8, i) * unit;
        const int x = as_i16(bmx + dx), y = as_i16(bmy + dy);
        /* the reference tests only the bounds a point can cross */
        const bool ok = act && (dy >= 0 || y >= mny) && (dy <= 0 || y <= mxy) && (dx >= 0 || x >= mnx) && (dx <= 0 || x <= mxx);
        int v = 0, cost = 0;
        if (ok)
        {
            const int t = l & 3;
            const uint8_t AS1 *rp = ref0 + roff + (long)__mul24(y + (t >> 1) * 4, rstride) + x + (t & 1) * 4;
#pragma unroll
            for (int yy = 0; yy < 4; yy++) v = Px<uint8_t>::sad(fq[yy], Px<uint8_t>::load_g(rp + (long)yy * rstride), v);
            cost = ((int)ctab[as_i16(x << 2) - mvpx] + (int)ctab[as_i16(y << 2) - mvpy]) & 0xffff;
        }
        v += dpp_all<0xB1>(v);      /* the four tiles of a point sit in one quad */
        v += dpp_all<0x4E>(v);
        improve = improve || (ok && v + cost < bcost);
        npts += (unsigned)__builtin_popcount((unsigned)((__ballot(ok && (l & 3) == 0) >> (row * 16)) & 0xffff));
    }
    const bool rowImprove = ((__ballot(improve) >> (row * 16)) & 0xffff) != 0;
    if (!act || l != 0) return;
    if (rowImprove)
    {
        queue[2 + atomicAdd(&queue[0], 1)] = j;
        return;
    }
    /* three rounds without improvement end the pattern with the start point standing; motion.cpp:1063-1069 */
    nsad += npts;
    int qx, qy;
    if (bprecost < bcost) { qx = bpx; qy = bpy; bcost = bprecost; }
    else { qx = as_i16(bmx << 2); qy = as_i16(bmy << 2); }
    int *res = (int*)(a.results + j);
    res[0] = (qx & 0xffff) | (qy << 16);
    res[1] = bcost;
    res[2] = (int)nsad;
    res[3] = (int)nsub;
}

/* The one-wave search for the PUs star8_kernel queued: wave b of a launch of G waves takes entries b, b + G, ...
 * (a shared hand-out counter would cost more than the few entries a picture usually leaves). An 8x8 PU is searched
 * up to the full-pel / predictor choice (refine8_kernel follows), a PU of another size completely. */
template<typename P>
__global__ __launch_bounds__(64) void search_queue_kernel(SearchArgs a, int *queue)
{
    const int lane = (int)threadIdx.x;
    const int count = uni(queue[0]);
    for (int t = (int)blockIdx.x; t < count; t += (int)gridDim.x)
    {
        const int e = uni(queue[2 + t]), j = e & 0x7fffffff;
        const x265amd_job *job = a.jobs + j;
        const int w = uni((int)job->w), h = uni((int)job->h);
        if (w > a.max_dim || h > a.max_dim)
        {
            if (lane == 0)
            {
                x265amd_result r = { { 0, 0 }, -1, 0, 0 };
                a.results[j] = r;
            }
            continue;
        }
        Searcher<P, 64, false> S;
        PlaneDev fenc, ref[4];
        job_planes(a, job, fenc, ref);
        S.bind(0, fenc, ref, a.nref, a.ctab, a.depth, job->pu_x, job->pu_y, w, h, a.max_dim, plane_budget(a.max_dim, (int)sizeof(P)));
        if (e >= 0) S.search_staged(job, a.method, a.results + j);
        else S.search(job, a.method, a.subme, a.results + j);
    }
}

} // namespace x265amd
#endif
