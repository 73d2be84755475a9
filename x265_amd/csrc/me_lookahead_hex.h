/* Lookahead MV fields, hexagon search, four (or two) CU rows per wave.
 *
 * The lookahead's search (EstimateRow::estimateCUCost, slicetype.cpp:1401-1470) is always X265_HEX_SEARCH on 8x8
 * CUs of the half-resolution planes (slicetype.h:65-72), a CU taking its predictors from its right neighbour and the
 * three CUs below (:1442-1466). The general one-wave search spends ~2000 instructions on such a CU, nearly all of it
 * bookkeeping: a CU has four 4x4 tiles and a hexagon step three to eight candidates. Here a wave owns a band of four
 * CU rows of one field, a row per group of 16 lanes = 4 candidates x 4 tiles per pass (launches too small to fill
 * the chip: two rows, 32 lanes and 8 candidates per pass -- shorter steps for more lane work). The rows run right to left in
 * lockstep, each two CUs behind the row below it (the lag of the reference's own wavefront, :1362-1379), so inside a
 * band the three CUs below are always done -- their MVs wait in LDS -- and only the bottom row of a band polls the
 * top row of the band below (device-scope loads of the MV array, preset to x = 0x7FFF, the reference's "not searched"
 * mark, :1150). Bands are handed out bottom-up by a ticket, a band only waits on a smaller ticket, so every wait ends.
 *
 * Per CU, as MotionEstimate::motionEstimate does for a lowres reference (motion.cpp:549-1146): the start-up
 * candidates (:575-621), the hexagon (:651-723: ring, half-hexagon walk, closing square, none range-checked), the
 * choice of :1063-1069, then one half-pel round on SAD and one quarter-pel round on SATD with the centre re-measured
 * (:1077-1099); every fractional position through lowresQPelCost (lowres.h:78-97: the half-pel plane of the
 * position, averaged with the next one for odd quarter-pels). Decisions are the reference's strict '<' in its visit
 * order; all of it is per-lane code that the lanes of a group take together.
 */
#ifndef X265AMD_ME_LOOKAHEAD_HEX_H
#define X265AMD_ME_LOOKAHEAD_HEX_H

namespace x265amd {

#define LB_ROWS 4                       /* most CU rows of a band = groups of a wave */
#define LB_WIN_W 12                     /* the sub-pel window of a CU: samples per row, rows */
#define LB_WIN_H 11
/* LDS of a wave: per CU row a ring of four MV words (a row reads the row below at its own column +-1 while that row
 * is two columns further on), then the four groups' sub-pel windows of four planes each */
#define LB_RING_BYTES (LB_ROWS * 4 * 4)
template<typename P> constexpr int band_window_bytes() { return 4 * LB_WIN_H * LB_WIN_W * (int)sizeof(P) + 16; }
/* LG: log2 of the lanes of a group (4 or 5: four or two CU rows per wave) */
template<typename P, int LG> constexpr int band_lds_bytes() { return LB_RING_BYTES + (64 >> LG) * band_window_bytes<P>(); }

template<typename P, int LG> struct BandCu
{
    static constexpr int LPP = 1 << LG, CPP = LPP / 4;      /* lanes of the group; candidates per pass (a lane per 4x4 tile) */
    /* group-uniform */
    const P AS1 *ref[4];                /* the four half-pel planes, at this CU's position */
    const uint16_t AS1 *ctab;
    int rstride, mvpx, mvpy, depth;
    unsigned nsad, nsub;
    int wbase, wx0, wy0;                /* the staged sub-pel window: LDS byte offset, origin relative to the CU */
    /* per lane */
    int l, tile;
    typename Px<P>::quad f[4];          /* the lane's tile of the source CU */

    __device__ __forceinline__ const P AS1 *pick(int i) const
    {
        const P AS1 *lo = (i & 1) ? ref[1] : ref[0], *hi = (i & 1) ? ref[3] : ref[2];
        return (i & 2) ? hi : lo;
    }
    /* The sub-pel rounds stay within 3 quarter-pels of where the integer search ended, so all they read lies in a
     * window of LB_WIN_W x LB_WIN_H samples of each half-pel plane; the group stages it in LDS once (one row segment
     * per lane and pass) instead of fetching every candidate's tile from memory. The origin is pulled back where the
     * window would leave the 12 margin samples the planes are sure to have: what is cut off no candidate reads. */
    __device__ __forceinline__ void stage_window(int qx, int qy, int xlim, int ylim)
    {
        wx0 = min((qx >> 2) - 1, xlim); wy0 = min((qy >> 2) - 1, ylim);
        constexpr int NSEG = 4 * LB_WIN_H, NW = LB_WIN_W * (int)sizeof(P) / 4;
        uint32_t v[(NSEG + LPP - 1) / LPP][NW];
        /* every load of the lane before the first store, so that they are in flight together */
#pragma unroll
        for (int i = 0; i < (NSEG + LPP - 1) / LPP; i++)
        {
            const int sg = l + LPP * i;
            const int p = (sg >= LB_WIN_H) + (sg >= 2 * LB_WIN_H) + (sg >= 3 * LB_WIN_H), r = sg - p * LB_WIN_H;
            const P AS1 *src = pick(p & 3) + __mul24(wy0 + r, rstride) + wx0;
            if (sg < NSEG)
            {
                if constexpr (sizeof(P) == 1)
                {
                    typedef uint32_t __attribute__((ext_vector_type(2), aligned(1))) u32x2u;
                    typedef uint32_t __attribute__((aligned(1))) u32u;
                    const u32x2u a = *(const u32x2u AS1*)src;
                    v[i][0] = a.x; v[i][1] = a.y; v[i][2] = *(const u32u AS1*)(src + 8);
                }
                else
                {
                    typedef uint32_t __attribute__((ext_vector_type(4), aligned(2))) u32x4u;
                    typedef uint32_t __attribute__((ext_vector_type(2), aligned(2))) u32x2u;
                    const u32x4u a = *(const u32x4u AS1*)src;
                    const u32x2u b = *(const u32x2u AS1*)(src + 8);
                    v[i][0] = a.x; v[i][1] = a.y; v[i][2] = a.z; v[i][3] = a.w; v[i][4] = b.x; v[i][5] = b.y;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < (NSEG + LPP - 1) / LPP; i++)
        {
            const int sg = l + LPP * i;
            uint32_t *dst = (uint32_t*)(x265amd_smem + wbase + sg * LB_WIN_W * (int)sizeof(P));
            if (sg < NSEG)
            {
#pragma unroll
                for (int k = 0; k < NW; k++) dst[k] = v[i][k];
            }
        }
    }
    /* lowresQPelCost of the lane's tile at quarter-pel (cx, cy), from the staged window */
    __device__ __forceinline__ int tile_cost_window(int cx, int cy, bool satd) const
    {
        const int toff = ((tile >> 1) * 4 - wy0) * LB_WIN_W + (tile & 1) * 4 - wx0;
        typename Px<P>::quad q[4];
        const int pa = (cy & 2) | ((cx & 2) >> 1);
        const int oa = wbase + ((pa * LB_WIN_H + (cy >> 2)) * LB_WIN_W + (cx >> 2) + toff) * (int)sizeof(P);
#pragma unroll
        for (int y = 0; y < 4; y++) q[y] = Px<P>::load_lds(oa + y * LB_WIN_W * (int)sizeof(P));
        if ((cx | cy) & 1)
        {
            const int bx = as_i16(cx + (cx & 1) * 2), by = as_i16(cy + (cy & 1) * 2);
            const int pb = (by & 2) | ((bx & 2) >> 1);
            const int ob = wbase + ((pb * LB_WIN_H + (by >> 2)) * LB_WIN_W + (bx >> 2) + toff) * (int)sizeof(P);
#pragma unroll
            for (int y = 0; y < 4; y++) q[y] = Px<P>::avg(q[y], Px<P>::load_lds(ob + y * LB_WIN_W * (int)sizeof(P)));
        }
        return tile_cost_quads<P>(f, q, satd);
    }
    /* lowresQPelCost of the lane's tile at quarter-pel (cx, cy); lowres.h:78-97 */
    __device__ __forceinline__ int tile_cost(int cx, int cy, bool satd) const
    {
        const int toff = __mul24((tile >> 1) * 4, rstride) + (tile & 1) * 4;
        typename Px<P>::quad q[4];
        const P AS1 *ra = pick((cy & 2) | ((cx & 2) >> 1)) + __mul24(cy >> 2, rstride) + (cx >> 2) + toff;
#pragma unroll
        for (int y = 0; y < 4; y++) q[y] = Px<P>::load_g(ra + y * rstride);
        if ((cx | cy) & 1)
        {
            const int bx = as_i16(cx + (cx & 1) * 2), by = as_i16(cy + (cy & 1) * 2);
            const P AS1 *rb = pick((by & 2) | ((bx & 2) >> 1)) + __mul24(by >> 2, rstride) + (bx >> 2) + toff;
#pragma unroll
            for (int y = 0; y < 4; y++) q[y] = Px<P>::avg(q[y], Px<P>::load_g(rb + y * rstride));
        }
        return tile_cost_quads<P>(f, q, satd);
    }
    /* bitcost.h:46 for a quarter-pel MV (full-pel candidates come in as mv << 2) */
    __device__ __forceinline__ int mvcost(int qx, int qy) const
    {
        return ((int)ctab[qx - mvpx] + (int)ctab[qy - mvpy]) & 0xffff;
    }
    /* Costs of K candidates, CPP per pass: pos(k, x, y) names candidate k's quarter-pel position. Lane k < K of the
     * group gets candidate k's distortion (+ lambda * bits when addmv(k)); the other lanes INT_MAX. */
    template<bool WINDOW = false, class POS, class ADD>
    __device__ __forceinline__ int eval(int K, bool satd, POS pos, ADD addmv) const
    {
        int mine = 0x7fffffff;
        int mx = 0, my = 0;
        if (l < K) pos(l, mx, my);
        const int mvc = l < K && addmv(l) ? mvcost(mx, my) : 0;
        for (int k0 = 0; k0 < K; k0 += CPP)
        {
            const int k = k0 + (l >> 2);
            int cx = 0, cy = 0;
            pos(k < K ? k : K - 1, cx, cy);
            int v = WINDOW ? tile_cost_window(cx, cy, satd) : tile_cost(cx, cy, satd);
            v += dpp_all<0xB1>(v);
            v += dpp_all<0x4E>(v);
            /* candidate k0 + i sits in quad i of the group; lane k0 + i of the group takes it */
            const int c = grp_lane<LG>(v, ((l - k0) & (CPP - 1)) * 4);
            if (l >= k0 && l < k0 + CPP && l < K) mine = c + mvc;
        }
        return mine;
    }
    /* eval for the integer stage (SAD, at most CPP * NP candidates, FULLPEL: all of them full-pel positions): the loads
     * of every pass are issued before the first cost is formed, so the passes cost one trip to memory, not NP */
    template<int NP, bool FULLPEL, class POS, class ADD>
    __device__ __forceinline__ int eval_sad(int K, POS pos, ADD addmv) const
    {
        int mine = 0x7fffffff;
        int mx = 0, my = 0;
        if (l < K) pos(l, mx, my);
        const int mvc = l < K && addmv(l) ? mvcost(mx, my) : 0;
        const int toff = __mul24((tile >> 1) * 4, rstride) + (tile & 1) * 4;
        typename Px<P>::quad qa[NP][4], qb[NP][4];
        bool odd[NP];
#pragma unroll
        for (int p = 0; p < NP; p++)
        {
            odd[p] = false;
            if (p == 0 || p * CPP < K)
            {
                const int k = p * CPP + (l >> 2);
                int cx = 0, cy = 0;
                pos(k < K ? k : K - 1, cx, cy);
                const P AS1 *ra = pick((cy & 2) | ((cx & 2) >> 1)) + __mul24(cy >> 2, rstride) + (cx >> 2) + toff;
#pragma unroll
                for (int y = 0; y < 4; y++) qa[p][y] = Px<P>::load_g(ra + y * rstride);
                if constexpr (!FULLPEL)
                {
                    odd[p] = ((cx | cy) & 1) != 0;
                    if (odd[p])
                    {
                        const int bx = as_i16(cx + (cx & 1) * 2), by = as_i16(cy + (cy & 1) * 2);
                        const P AS1 *rb = pick((by & 2) | ((bx & 2) >> 1)) + __mul24(by >> 2, rstride) + (bx >> 2) + toff;
#pragma unroll
                        for (int y = 0; y < 4; y++) qb[p][y] = Px<P>::load_g(rb + y * rstride);
                    }
                }
            }
        }
#pragma unroll
        for (int p = 0; p < NP; p++)
            if (p == 0 || p * CPP < K)
            {
                if constexpr (!FULLPEL)
                    if (odd[p])
                    {
#pragma unroll
                        for (int y = 0; y < 4; y++) qa[p][y] = Px<P>::avg(qa[p][y], qb[p][y]);
                    }
                int v = tile_cost_quads<P>(f, qa[p], false);
                v += dpp_all<0xB1>(v);
                v += dpp_all<0x4E>(v);
                const int c = grp_lane<LG>(v, ((l - p * CPP) & (CPP - 1)) * 4);
                if (l >= p * CPP && l < p * CPP + CPP && l < K) mine = c + mvc;
            }
        return mine;
    }
    /* first minimum of the lanes [k0, k1) in lane order if it beats bcost: its lane, or -1 */
    __device__ __forceinline__ int fold(int mine, int k0, int k1, int &bcost) const
    {
        int key = 0x7fffffff;
        if (l >= k0 && l < k1) key = (mine << 6) | l;
        key = grp_min<LG>(key);
        if ((key >> 6) < bcost) { bcost = key >> 6; return key & 63; }
        return -1;
    }
};

__device__ __forceinline__ int mv_word(int x, int y) { return (x & 0xffff) | (y << 16); }

/* Measurement build (-DX265AMD_PHASE_PROFILE): a wave adds up, in registers, the shader-clock cycles it spends between
 * the marks of a CU step and writes the sums once, with plain vector stores from lane 0, when it has no band left:
 * phase[wave][0..LB_NPHASE-1] cycles, [LB_NPHASE] steps, [LB_NPHASE + 1] hexagon-walk iterations. The product build
 * has none of it. */
#ifdef X265AMD_PHASE_PROFILE
#define LB_NPHASE 10
#define LB_MARK(i) do { const long long t_ = (long long)__builtin_readcyclecounter(); ph[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define LB_MARK(i) do { } while (0)
#endif

template<typename P, int LG>
__global__ __launch_bounds__(64) void lookahead_band_kernel(LookaheadArgs a)
{
#ifdef X265AMD_PHASE_PROFILE
    long long ph[LB_NPHASE + 2] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    long long tlast = (long long)__builtin_readcyclecounter();
#endif
    constexpr int ROWS = 64 >> LG;                  /* CU rows of a band */
    /* passes of the integer stage's evaluations: start-up (at most 7 candidates), hexagon (6), closing square (8) */
    constexpr int CPP = BandCu<P, LG>::CPP, NP7 = (7 + CPP - 1) / CPP, NP6 = (6 + CPP - 1) / CPP, NP8 = (8 + CPP - 1) / CPP;
    int *rowmv = (int*)x265amd_smem;                /* per CU row of this band the MV words of its last four CUs */
    const int lane = (int)threadIdx.x, g = lane >> LG, l = lane & ((1 << LG) - 1);
    const int W = a.wcu, H = a.hcu, ncu = W * H;
    const int bands = (H + ROWS - 1) / ROWS, nbands = a.nfields * bands;
    constexpr unsigned long long SQX = nibs({ 0, 0, 0, -1, 1, -1, -1, 1, 1 }), SQY = nibs({ 0, -1, 1, 0, 0, -1, 1, -1, 1 });
    constexpr unsigned long long HRX = nibs({ -2, -1, 1, 2, 1, -1 }), HRY = nibs({ 0, 2, 2, 0, -2, -2 });
    constexpr unsigned long long H2X = nibs({ -1, -2, -1, 1, 2, 1, -1, -2 }), H2Y = nibs({ -2, 0, 2, 2, 0, -2, -2, 0 });
    constexpr unsigned long long MOD6M1 = nibs({ 5, 0, 1, 2, 3, 4, 5, 0 });
    const int hpel_dirs = kWorkload[a.subme][1], qpel_dirs = kWorkload[a.subme][3];
    for (;;)
    {
        int t = 0;
        if (lane == 0) t = atomicAdd(a.ticket, 1);
        t = uni(t);
        if (t >= nbands) break;
        const int fi = t % a.nfields, band = t / a.nfields;
        const FieldDev &fd = a.fields[fi];
        const int cuy = H - 1 - band * ROWS - g;             /* this group's CU row; rows above the field idle */
        int *mvw = (int*)(a.mvs + (size_t)fi * ncu * 2);
        int32_t *costs = a.costs + (size_t)fi * ncu;
        BandCu<P, LG> C;
        C.l = l; C.tile = l & 3;
        C.wbase = LB_RING_BYTES + g * band_window_bytes<P>();
        C.ctab = (const uint16_t AS1*)a.ctab;
        C.rstride = fd.ref[0].stride;
        C.depth = a.depth;
        int right = 0;
        wave_sync();
        for (int s = 0; s < W + 2 * (ROWS - 1); s++)
        {
            const int cux = W - 1 - (s - 2 * g);
            LB_MARK(9);                                     /* between steps: ticket, loop, wave_sync */
#ifdef X265AMD_PHASE_PROFILE
            ph[LB_NPHASE]++;
#endif
            if (cuy >= 0 && cux >= 0 && cux < W)
            {
                const int cuXY = cux + cuy * W;
                /* ---- predictors: right neighbour and the three CUs below (slicetype.cpp:1442-1466) ---- */
                int below = 0, bleft = 0, bright = 0;
                if (cuy < H - 1)
                {
                    if (g == 0)
                    {
                        int nb = 0;
                        const int dx = l == 0 ? 0 : (l == 1 ? -1 : 1);
                        if (l < 3 && cux + dx >= 0 && cux + dx < W)
                        {
                            const int *src = mvw + cuXY + W + dx;
                            nb = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            while ((nb & 0xffff) == 0x7FFF)
                            {
                                __builtin_amdgcn_s_sleep(4);
                                nb = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                        below = grp_lane<LG>(nb, 0); bleft = grp_lane<LG>(nb, 1); bright = grp_lane<LG>(nb, 2);
                    }
                    else
                    {
                        const int *rb = rowmv + (g - 1) * 4;
                        below = rb[cux & 3];
                        if (cux > 0) bleft = rb[(cux - 1) & 3];
                        if (cux < W - 1) bright = rb[(cux + 1) & 3];
                    }
                }
                LB_MARK(0);                                 /* the neighbours' MVs: polling the band below, or LDS */
                int cx[4] = { 0, 0, 0, 0 }, cy[4] = { 0, 0, 0, 0 }, numc = 0;
#define LB_PUSH(word) do { const int w_ = (word); if (numc == 0) { cx[0] = as_i16(w_); cy[0] = w_ >> 16; } else if (numc == 1) { cx[1] = as_i16(w_); cy[1] = w_ >> 16; } \
                           else if (numc == 2) { cx[2] = as_i16(w_); cy[2] = w_ >> 16; } else { cx[3] = as_i16(w_); cy[3] = w_ >> 16; } numc++; } while (0)
                if (cux < W - 1) LB_PUSH(right);
                if (cuy < H - 1)
                {
                    LB_PUSH(below);
                    if (cux > 0) LB_PUSH(bleft);
                    if (cux < W - 1) LB_PUSH(bright);
                }
#undef LB_PUSH
                int mvpx = cx[0], mvpy = cy[0];
                if (numc > 1) { mvpx = median3(cx[0], cx[1], cx[2]); mvpy = median3(cy[0], cy[1], cy[2]); }
                C.mvpx = mvpx; C.mvpy = mvpy;
                /* bounds through the reference's uint16 casts (slicetype.cpp:1427-1430) */
                const int mnx = as_i16(-cux * 8 - 8), mny = as_i16(-cuy * 8 - 8);
                const int mxx = as_i16((W - cux - 1) * 8 + 8), mxy = as_i16((H - cuy - 1) * 8 + 8);
                {
                    const long off = (long)(cuy * 8) * C.rstride + cux * 8;
#pragma unroll
                    for (int i = 0; i < 4; i++) C.ref[i] = (const P AS1*)fd.ref[i].origin + off;
                    const P AS1 *src = (const P AS1*)fd.fenc.origin + (long)(cuy * 8 + (C.tile >> 1) * 4) * fd.fenc.stride + cux * 8 + (C.tile & 1) * 4;
#pragma unroll
                    for (int y = 0; y < 4; y++) C.f[y] = Px<P>::load_g(src + (long)y * fd.fenc.stride);
                }
                C.nsad = 0; C.nsub = 0;

                /* ---- start-up candidates, motion.cpp:575-621 ---- */
                const int qmnx = as_i16(mnx << 2), qmny = as_i16(mny << 2), qmxx = as_i16(mxx << 2), qmxy = as_i16(mxy << 2);
#define LB_CLIPX(v) ((v) > qmxx ? qmxx : ((v) < qmnx ? qmnx : (v)))
#define LB_CLIPY(v) ((v) > qmxy ? qmxy : ((v) < qmny ? qmny : (v)))
                const int pmx = LB_CLIPX(mvpx), pmy = LB_CLIPY(mvpy);
                int bmx = as_i16((pmx + 2) >> 2), bmy = as_i16((pmy + 2) >> 2);
                const bool fpOn = ((pmx | pmy) & 3) != 0, zeroOn = (pmx | pmy) != 0;
                bool cOn[4];
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    cx[i] = LB_CLIPX(cx[i]); cy[i] = LB_CLIPY(cy[i]);
                    cOn[i] = i < numc && (cx[i] | cy[i]) != 0 && !(cx[i] == pmx && cy[i] == pmy);
                }
#undef LB_CLIPX
#undef LB_CLIPY
                int bcost, bprecost, bpx = pmx, bpy = pmy;
                LB_MARK(1);                                 /* predictor, bounds, source tile loads issued, candidate clipping */
                {
                    const int fx = as_i16(bmx << 2), fy = as_i16(bmy << 2);
                    /* slots: 0 clipped predictor (no lambda * bits of its own, :579-582), 1 its full-pel rounding, 2 MV 0, 3.. candidates */
                    auto pos = [&](int k, int &x, int &y)
                    {
                        x = pmx; y = pmy;
                        if (k == 1) { x = fx; y = fy; }
                        if (k == 2) { x = 0; y = 0; }
                        if (k == 3) { x = cx[0]; y = cy[0]; }
                        if (k == 4) { x = cx[1]; y = cy[1]; }
                        if (k == 5) { x = cx[2]; y = cy[2]; }
                        if (k == 6) { x = cx[3]; y = cy[3]; }
                    };
                    /* (16-bit samples: both passes' tiles and their averaging partners at once cost a wave of occupancy) */
                    int mine;
                    if constexpr (sizeof(P) == 1 || NP7 == 1) mine = C.template eval_sad<NP7, false>(3 + numc, pos, [](int k) { return k != 0; });
                    else mine = C.eval(3 + numc, false, pos, [](int k) { return k != 0; });
                    bprecost = grp_lane<LG>(mine, 0);
                    C.nsub++;
                    bcost = bprecost;
                    if (fpOn) { bcost = grp_lane<LG>(mine, 1); C.nsad++; }
                    if (zeroOn)
                    {
                        const int c = grp_lane<LG>(mine, 2);
                        C.nsad++;
                        if (c < bcost) { bcost = c; bmx = 0; bmy = 0; }
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++)
                    {
                        const int c = grp_lane<LG>(mine, 3 + i);
                        if (cOn[i] && !(cx[i] == bpx && cy[i] == bpy))
                        {
                            C.nsub++;
                            if (c < bprecost) { bprecost = c; bpx = cx[i]; bpy = cy[i]; }
                        }
                    }
                }
                LB_MARK(2);                                 /* start-up candidates and their tests */
                /* ---- hexagon search, motion.cpp:651-723 ---- */
                {
                    const int ox = bmx, oy = bmy;
                    const int mine = C.template eval_sad<NP6, true>(6, [&](int k, int &x, int &y) { x = as_i16(as_i16(ox + nib(HRX, k)) << 2); y = as_i16(as_i16(oy + nib(HRY, k)) << 2); },
                                            [](int) { return true; });
                    C.nsad += 6;
                    int dir = C.fold(mine, 0, 6, bcost);
                    LB_MARK(3);                             /* the hexagon's six points */
                    if (dir >= 0)
                    {
                        bmx = as_i16(bmx + nib(H2X, dir + 1)); bmy = as_i16(bmy + nib(H2Y, dir + 1));
                        for (int i = (a.merange >> 1) - 1; i > 0 && bmx >= mnx && bmx <= mxx && bmy >= mny && bmy <= mxy; i--)
                        {
                            const int hx = bmx, hy = bmy, hd = dir;
                            const int m3 = C.template eval_sad<1, true>(3, [&](int k, int &x, int &y) { x = as_i16(as_i16(hx + nib(H2X, hd + k)) << 2); y = as_i16(as_i16(hy + nib(H2Y, hd + k)) << 2); },
                                                  [](int) { return true; });
                            C.nsad += 3;
                            const int step = C.fold(m3, 0, 3, bcost) + 1;
#ifdef X265AMD_PHASE_PROFILE
                            ph[LB_NPHASE + 1]++;
#endif
                            if (!step) break;
                            dir += step - 2;
                            dir = nib(MOD6M1, dir + 1);
                            bmx = as_i16(bmx + nib(H2X, dir + 1)); bmy = as_i16(bmy + nib(H2Y, dir + 1));
                        }
                    }
                    LB_MARK(4);                             /* the walk: three points per iteration */
                    const int sx = bmx, sy = bmy;
                    const int m8 = C.template eval_sad<NP8, true>(8, [&](int k, int &x, int &y) { x = as_i16(as_i16(sx + nib(SQX, k + 1)) << 2); y = as_i16(as_i16(sy + nib(SQY, k + 1)) << 2); },
                                          [](int) { return true; });
                    C.nsad += 8;
                    const int sq = C.fold(m8, 0, 8, bcost) + 1;
                    bmx = as_i16(bmx + nib(SQX, sq)); bmy = as_i16(bmy + nib(SQY, sq));
                }
                LB_MARK(5);                                 /* the closing square */
                /* ---- motion.cpp:1063-1069, then the lowres sub-pel rounds, :1077-1099 ---- */
                int qx, qy;
                if (bprecost < bcost) { qx = bpx; qy = bpy; bcost = bprecost; }
                else { qx = as_i16(bmx << 2); qy = as_i16(bmy << 2); }
                if (bcost)
                {
                    C.stage_window(qx, qy, (W - cux) * 8, (H - cuy) * 8 + 12 - LB_WIN_H);
                    wave_sync();
                    LB_MARK(6);                             /* the sub-pel window staged in LDS */
                    {
                        const int ox = qx, oy = qy;
                        const int mine = C.template eval<true>(hpel_dirs, false, [&](int k, int &x, int &y) { x = as_i16(ox + nib(SQX, k + 1) * 2); y = as_i16(oy + nib(SQY, k + 1) * 2); },
                                                [](int) { return true; });
                        C.nsub += (unsigned)hpel_dirs;
                        const int k = C.fold(mine, 0, hpel_dirs, bcost);
                        const int bdir = k < 0 ? 0 : k + 1;
                        qx = as_i16(qx + nib(SQX, bdir) * 2); qy = as_i16(qy + nib(SQY, bdir) * 2);
                    }
                    LB_MARK(7);                             /* half-pel round */
                    {
                        const int ox = qx, oy = qy;
                        const int mine = C.template eval<true>(1 + qpel_dirs, true, [&](int k, int &x, int &y) { x = as_i16(ox + nib(SQX, k)); y = as_i16(oy + nib(SQY, k)); },
                                                [](int) { return true; });
                        C.nsub += (unsigned)(1 + qpel_dirs);
                        bcost = grp_lane<LG>(mine, 0);
                        const int k = C.fold(mine, 1, 1 + qpel_dirs, bcost);
                        const int bdir = k < 0 ? 0 : k;
                        qx = as_i16(qx + nib(SQX, bdir)); qy = as_i16(qy + nib(SQY, bdir));
                    }
                }
                LB_MARK(8);                                 /* quarter-pel round */
                right = mv_word(qx, qy);
                rowmv[g * 4 + (cux & 3)] = right;
                if (l == 0)
                {
                    costs[cuXY] = bcost;
                    if (a.stats)
                    {
                        x265amd_result r;
                        r.mv[0] = (int16_t)qx; r.mv[1] = (int16_t)qy; r.cost = bcost; r.sad_calls = C.nsad; r.subpel_calls = C.nsub;
                        a.stats[(size_t)fi * ncu + cuXY] = r;
                    }
                    __hip_atomic_store(mvw + cuXY, right, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            wave_sync();
        }
    }
#ifdef X265AMD_PHASE_PROFILE
    if (a.phase && lane == 0)
        for (int i = 0; i < LB_NPHASE + 2; i++) a.phase[(size_t)blockIdx.x * 16 + i] = ph[i];
#endif
}

} // namespace x265amd
#endif
