/* Sub-pel refinement of 8x8 PUs, four PUs per wave (8-bit samples, full-resolution planes).
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

} // namespace x265amd
#endif
