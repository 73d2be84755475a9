/* TEST INFRASTRUCTURE ONLY -- see me_oracle.h. Plain-C restatement of the
 * reference's ME path; the shipped HIP engine never links or calls this file. */
#include "me_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <math.h>

/* HEVC 8-tap luma interpolation taps; source/Lib/TLibCommon/TComRom.cpp:175-181 */
static const int16_t orc_luma_filter[4][8] = {
    {  0, 0,   0, 64,  0,   0, 0,  0 },
    { -1, 4, -10, 58, 17,  -5, 1,  0 },
    { -1, 4, -11, 40, 40, -11, 4, -1 },
    {  0, 1,  -5, 17, 58, -10, 4, -1 },
};

/* HEVC 4-tap chroma interpolation taps (eighth-pel); source/Lib/TLibCommon/TComRom.cpp:183-193 */
static const int16_t orc_chroma_filter[8][4] = {
    {  0, 64,  0,  0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
    { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 },
};

/* lambda^2 for non-I slices per QP; source/Lib/TLibCommon/TComRom.cpp:529-541 */
static const double orc_lambda2_non_I[70] = {
    0.05231, 0.060686, 0.07646, 0.096333333, 0.151715667, 0.15292, 0.192666667, 0.242745, 0.382299,
    0.385333333, 0.485489333, 0.611678333, 0.963333333, 0.970979, 1.223356667, 1.541333333, 2.427447667,
    2.446714, 3.082666667, 3.883916667, 6.116785, 6.165333333, 7.767833333, 9.786856667, 15.41333333,
    16.57137733, 22.183542, 29.5936, 39.357022, 52.19656867, 69.05173333, 91.14257667, 150.0651347,
    157.8325333, 207.1422197, 271.4221573, 443.904, 447.4271947, 563.722941, 710.2464, 1118.567987,
    1127.445883, 1420.4928, 1789.70878, 2818.614706, 2840.9856, 3579.41756, 4509.78353, 7102.464,
    7158.83512, 9019.56706, 11363.9424, 14317.66967, 18039.13269, 22727.8821, 28635.33594, 36078.2611,
    45455.7588, 57270.66508, 72156.51362, 90911.5068, 114541.3166, 144313.0101, 181822.992, 229082.6059,
    288625.9859, 363645.9408, 458165.1574, 577251.9032, 727291.7952
};

/* optional trace of every full-pel SAD position (debugging aid for the tests) */
static int64_t *orc_trace_buf; static int orc_trace_n, orc_trace_cap;
void orc_trace_begin(int64_t *buf, int cap) { orc_trace_buf = buf; orc_trace_cap = cap; orc_trace_n = 0; }
int orc_trace_end(void) { orc_trace_buf = 0; return orc_trace_n; }
static inline void orc_trace(int64_t v) { if (orc_trace_buf) { if (orc_trace_n < orc_trace_cap) orc_trace_buf[orc_trace_n] = v; orc_trace_n++; } }

#define PIXEL uint8_t
#define FN(n) n##_8
#include "me_oracle_impl.h"
#undef PIXEL
#undef FN

#define PIXEL uint16_t
#define FN(n) n##_16
#include "me_oracle_impl.h"
#undef PIXEL
#undef FN

/* partition enum order of source/common/primitives.h:65-73 */
static const uint8_t orc_part_wh[25][2] = {
    {4,4}, {8,8}, {8,4}, {4,8}, {16,16}, {16,8}, {8,16}, {16,12}, {12,16}, {16,4}, {4,16},
    {32,32}, {32,16}, {16,32}, {32,24}, {24,32}, {32,8}, {8,32},
    {64,64}, {64,32}, {32,64}, {64,48}, {48,64}, {64,16}, {16,64}
};

int orc_partition(int w, int h)
{
    for (int i = 0; i < 25; i++)
        if (orc_part_wh[i][0] == w && orc_part_wh[i][1] == h)
            return i;
    return -1;
}

#define DISPATCH(depth, call8, call16) do { if ((depth) <= 8) { call8; } else { call16; } } while (0)

int orc_sad(int depth, const void *a, intptr_t sa, const void *b, intptr_t sb, int w, int h)
{
    if (depth <= 8) return sad_8((const uint8_t*)a, sa, (const uint8_t*)b, sb, w, h);
    return sad_16((const uint16_t*)a, sa, (const uint16_t*)b, sb, w, h);
}

int orc_satd(int depth, const void *a, intptr_t sa, const void *b, intptr_t sb, int w, int h)
{
    if (depth <= 8) return satd_8((const uint8_t*)a, sa, (const uint8_t*)b, sb, w, h);
    return satd_16((const uint16_t*)a, sa, (const uint16_t*)b, sb, w, h);
}

int orc_sa8d_inter(int depth, const void *a, intptr_t sa, const void *b, intptr_t sb, int w, int h)
{
    if (depth <= 8) return sa8d_inter_8((const uint8_t*)a, sa, (const uint8_t*)b, sb, w, h);
    return sa8d_inter_16((const uint16_t*)a, sa, (const uint16_t*)b, sb, w, h);
}

/* sad_x3 / sad_x4 share one 64-stride fenc block; pixel.cpp:111-156 */
void orc_sad_x3(int depth, const void *fenc, const void *r0, const void *r1, const void *r2,
                intptr_t rs, int w, int h, int32_t *res)
{
    res[0] = orc_sad(depth, fenc, 64, r0, rs, w, h);
    res[1] = orc_sad(depth, fenc, 64, r1, rs, w, h);
    res[2] = orc_sad(depth, fenc, 64, r2, rs, w, h);
}

void orc_sad_x4(int depth, const void *fenc, const void *r0, const void *r1, const void *r2, const void *r3,
                intptr_t rs, int w, int h, int32_t *res)
{
    orc_sad_x3(depth, fenc, r0, r1, r2, rs, w, h, res);
    res[3] = orc_sad(depth, fenc, 64, r3, rs, w, h);
}

void orc_pixelavg_pp(int depth, void *dst, intptr_t ds, const void *a, intptr_t sa, const void *b, intptr_t sb, int w, int h)
{
    DISPATCH(depth, pixelavg_8((uint8_t*)dst, ds, (const uint8_t*)a, sa, (const uint8_t*)b, sb, w, h),
             pixelavg_16((uint16_t*)dst, ds, (const uint16_t*)a, sa, (const uint16_t*)b, sb, w, h));
}

void orc_luma_hpp(int depth, const void *src, intptr_t ss, void *dst, intptr_t ds, int ci, int w, int h)
{
    DISPATCH(depth, luma_hpp_8((const uint8_t*)src, ss, (uint8_t*)dst, ds, ci, w, h, depth),
             luma_hpp_16((const uint16_t*)src, ss, (uint16_t*)dst, ds, ci, w, h, depth));
}

void orc_luma_vpp(int depth, const void *src, intptr_t ss, void *dst, intptr_t ds, int ci, int w, int h)
{
    DISPATCH(depth, luma_vpp_8((const uint8_t*)src, ss, (uint8_t*)dst, ds, ci, w, h, depth),
             luma_vpp_16((const uint16_t*)src, ss, (uint16_t*)dst, ds, ci, w, h, depth));
}

void orc_luma_hps(int depth, const void *src, intptr_t ss, int16_t *dst, intptr_t ds, int ci, int ext, int w, int h)
{
    DISPATCH(depth, luma_hps_8((const uint8_t*)src, ss, dst, ds, ci, ext, w, h, depth),
             luma_hps_16((const uint16_t*)src, ss, dst, ds, ci, ext, w, h, depth));
}

void orc_luma_vsp(int depth, const int16_t *src, intptr_t ss, void *dst, intptr_t ds, int ci, int w, int h)
{
    DISPATCH(depth, luma_vsp_8(src, ss, (uint8_t*)dst, ds, ci, w, h, depth),
             luma_vsp_16(src, ss, (uint16_t*)dst, ds, ci, w, h, depth));
}

void orc_lowres_core(int depth, const void *src, void *d0, void *dh, void *dv, void *dc,
                     intptr_t ss, intptr_t ds, int w, int h)
{
    DISPATCH(depth, lowres_core_8((const uint8_t*)src, (uint8_t*)d0, (uint8_t*)dh, (uint8_t*)dv, (uint8_t*)dc, ss, ds, w, h),
             lowres_core_16((const uint16_t*)src, (uint16_t*)d0, (uint16_t*)dh, (uint16_t*)dv, (uint16_t*)dc, ss, ds, w, h));
}

/* bitcost.cpp:71-83 (bit-size estimate, float) and :47-54 (scale by
 * sqrt(lambda2) in double, +0.5f, clamp to 65535, truncate) */
void orc_cost_table(int qp, uint16_t *out)
{
    double lambda = sqrt(orc_lambda2_non_I[qp]);
    float log2_2 = 2.0f / logf(2.0f);
    for (int i = 0; i < 32768; i++)
    {
        float bits = (i == 0) ? 0.718f : logf((float)(i + 1)) * log2_2 + 1.718f;
        double v = bits * lambda + 0.5f;
        if (v > 65535.0) v = 65535.0;
        uint16_t c = (uint16_t)v;
        if (i < 32768 && 32767 + i < 65535) out[32767 + i] = c;
        out[32767 - i] = c;
    }
}

int orc_me_search(const orc_me_ctx *c, intptr_t off, int w, int h, orc_mv mvmin, orc_mv mvmax, orc_mv qmvp,
                  int numc, const orc_mv *mvc, int merange, orc_mv *out, orc_me_stats *st)
{
    if (c->depth <= 8)
        return me_search_8(c, off, w, h, mvmin, mvmax, qmvp, numc, mvc, merange, out, st);
    return me_search_16(c, off, w, h, mvmin, mvmax, qmvp, numc, mvc, merange, out, st);
}

void orc_me_search_batch(const orc_me_ctx *c, const orc_job *jobs, int njobs, orc_result *out)
{
    for (int i = 0; i < njobs; i++)
    {
        const orc_job *j = &jobs[i];
        orc_mv mn = { j->mvmin[0], j->mvmin[1] }, mx = { j->mvmax[0], j->mvmax[1] }, p = { j->qmvp[0], j->qmvp[1] };
        orc_mv mvc[4], o;
        orc_me_stats st;
        memset(&st, 0, sizeof(st));
        for (int k = 0; k < 4; k++) { mvc[k].x = j->mvc[k][0]; mvc[k].y = j->mvc[k][1]; }
        out[i].cost = orc_me_search(c, (intptr_t)j->pu_y * c->fenc_stride + j->pu_x, j->w, j->h, mn, mx, p,
                                    j->numc, mvc, j->merange, &o, &st);
        out[i].mv[0] = o.x; out[i].mv[1] = o.y;
        out[i].sad_calls = st.sad_calls; out[i].subpel_calls = st.subpel_calls;
    }
}

void orc_rk_downsample(int depth, const void *src, int width, int height, intptr_t sstride, int16_t *dst, intptr_t dstride)
{
    for (int i = 0; i < height / 4; i++)
        for (int k = 0; k < width / 4; k++)
        {
            int16_t sum = 0;
            for (int m = 0; m < 4; m++)
                for (int n = 0; n < 4; n++)
                    sum = (int16_t)(sum + (depth <= 8 ? ((const uint8_t*)src)[(i * 4 + m) * sstride + k * 4 + n]
                                                      : ((const uint16_t*)src)[(i * 4 + m) * sstride + k * 4 + n]));
            dst[i * dstride + k] = sum;
        }
}

void orc_rk_ctu_batch(const orc_me_ctx *c, const int16_t *ds_origin, intptr_t ds_stride, int pic_w, int pic_h,
                      const orc_rk_ctu *jobs, int n, orc_rk_result *out)
{
    for (int i = 0; i < n; i++)
    {
        if (c->depth <= 8) rk_ctu_8(c, ds_origin, ds_stride, pic_w, pic_h, &jobs[i], &out[i]);
        else rk_ctu_16(c, ds_origin, ds_stride, pic_w, pic_h, &jobs[i], &out[i]);
    }
}

int orc_subpel_compare(const orc_me_ctx *c, intptr_t off, int w, int h, orc_mv qmv, int cmp)
{
    if (c->depth <= 8)
    {
        me_t_8 m;
        memset(&m, 0, sizeof(m));
        m.c = c; m.w = w; m.h = h; m.off = off; m.stride = c->ref_stride;
        for (int y = 0; y < h; y++)
            memcpy(m.fenc + 64 * y, (const uint8_t*)c->fenc_plane + off + y * c->fenc_stride, w);
        return subpel_cmp_8(&m, qmv.x, qmv.y, cmp ? satd_8 : sad_8);
    }
    else
    {
        me_t_16 m;
        memset(&m, 0, sizeof(m));
        m.c = c; m.w = w; m.h = h; m.off = off; m.stride = c->ref_stride;
        for (int y = 0; y < h; y++)
            memcpy(m.fenc + 64 * y, (const uint16_t*)c->fenc_plane + off + y * c->fenc_stride, w * 2);
        return subpel_cmp_16(&m, qmv.x, qmv.y, cmp ? satd_16 : sad_16);
    }
}

static int16_t median3(int16_t a, int16_t b, int16_t c)
{
    /* slicetype.cpp:42-57 evaluates to the median of the three values */
    int16_t lo = a < b ? a : b, hi = a < b ? b : a;
    int16_t m = lo > c ? lo : c;
    return m < hi ? m : hi;
}

/* slicetype.cpp:1401-1470 (ME part), visiting order of slicetype.cpp:1344,1356 */
void orc_lookahead_field(const orc_me_ctx *c, int widthInCU, int heightInCU, int merange,
                         orc_mv *mvs, int32_t *costs, orc_me_stats *st)
{
    for (int cuy = heightInCU - 1; cuy >= 0; cuy--)
    {
        for (int cux = widthInCU - 1; cux >= 0; cux--)
        {
            const int cuXY = cux + cuy * widthInCU;
            const intptr_t pelOffset = 8 * cux + 8 * cuy * c->fenc_stride;
            orc_mv mvmin, mvmax, mvc[4], mvp;
            /* bounds pass through uint16 and land in int16 (slicetype.cpp:1427-1430) */
            mvmin.x = (int16_t)(uint16_t)(-cux * 8 - 8);
            mvmin.y = (int16_t)(uint16_t)(-cuy * 8 - 8);
            mvmax.x = (int16_t)(uint16_t)((widthInCU - cux - 1) * 8 + 8);
            mvmax.y = (int16_t)(uint16_t)((heightInCU - cuy - 1) * 8 + 8);
            int numc = 0;
            memset(mvc, 0, sizeof(mvc));
            if (cux < widthInCU - 1) mvc[numc++] = mvs[cuXY + 1];
            if (cuy < heightInCU - 1)
            {
                mvc[numc++] = mvs[cuXY + widthInCU];
                if (cux > 0) mvc[numc++] = mvs[cuXY + widthInCU - 1];
                if (cux < widthInCU - 1) mvc[numc++] = mvs[cuXY + widthInCU + 1];
            }
            if (numc <= 1) mvp = mvc[0];
            else
            {
                mvp.x = median3(mvc[0].x, mvc[1].x, mvc[2].x);
                mvp.y = median3(mvc[0].y, mvc[1].y, mvc[2].y);
            }
            costs[cuXY] = orc_me_search(c, pelOffset, 8, 8, mvmin, mvmax, mvp, numc, mvc, merange, &mvs[cuXY], st);
        }
    }
}


void orc_lookahead_frame_cost(int depth, const void *fenc, const void *const ref0[4], const void *const ref1[4],
                              intptr_t stride, int widthInCU, int heightInCU, int d0, int d1, int intra_calculated,
                              const int32_t *inv_qscale, const orc_mv *mvs0, const int32_t *mvcosts0,
                              const orc_mv *mvs1, const int32_t *mvcosts1, int32_t *intra_cost, uint16_t *lowres_costs,
                              int32_t *row_satds_intra, int32_t *row_satds, int64_t totals[5])
{
    if (depth <= 8)
        la_frame_cost_8(depth, fenc, ref0, ref1, stride, widthInCU, heightInCU, d0, d1, intra_calculated, inv_qscale,
                        mvs0, mvcosts0, mvs1, mvcosts1, intra_cost, lowres_costs, row_satds_intra, row_satds, totals);
    else
        la_frame_cost_16(depth, fenc, ref0, ref1, stride, widthInCU, heightInCU, d0, d1, intra_calculated, inv_qscale,
                         mvs0, mvcosts0, mvs1, mvcosts1, intra_cost, lowres_costs, row_satds_intra, row_satds, totals);
}


void orc_weight_pp(int depth, const void *src, void *dst, intptr_t ss, intptr_t ds, int width, int height,
                   int w0, int round, int shift, int offset)
{
    if (depth <= 8) weight_pp_8((const uint8_t*)src, (uint8_t*)dst, ss, ds, width, height, w0, round, shift, offset, depth);
    else weight_pp_16((const uint16_t*)src, (uint16_t*)dst, ss, ds, width, height, w0, round, shift, offset, depth);
}

void orc_apply_weight(int depth, const void *src_origin, void *dst_origin, intptr_t stride, int width, int height,
                      int mx, int my, int inputWeight, int log2Denom, int inputOffset)
{
    if (depth <= 8) apply_weight_8((const uint8_t*)src_origin, (uint8_t*)dst_origin, stride, width, height, mx, my, inputWeight, log2Denom, inputOffset, depth);
    else apply_weight_16((const uint16_t*)src_origin, (uint16_t*)dst_origin, stride, width, height, mx, my, inputWeight, log2Denom, inputOffset, depth);
}

uint32_t orc_weight_cost_luma(int depth, const void *fenc, const void *src, intptr_t stride, int wcu, int hcu,
                              const int32_t *intra_cost, int use_wp, int scale, int denom, int offset)
{
    if (depth <= 8) return weight_cost_luma_8((const uint8_t*)fenc, (const uint8_t*)src, stride, wcu, hcu, intra_cost, use_wp, scale, denom, offset, depth);
    return weight_cost_luma_16((const uint16_t*)fenc, (const uint16_t*)src, stride, wcu, hcu, intra_cost, use_wp, scale, denom, offset, depth);
}

/* CostEstimate::weightsAnalyse, slicetype.cpp:1258-1340. Returns 1 when a weight was chosen
 * (weightedRef.isWeighted); out = { inputWeight, log2WeightDenom, inputOffset }, scores =
 * { origscore, minscore }. The caller weights the four planes with orc_weight_pp as :1333-1337
 * do (scale, 1 << (denom - 1 + correction), denom + correction, offset << (depth - 8)). */
int orc_weights_analyse(int depth, const void *fenc, const void *ref, intptr_t stride, int wcu, int hcu,
                        const int32_t *intra_cost, const uint64_t stats[4], int out[3], uint32_t scores[2])
{
    const float epsilon = 1.f / 128.f;
    const int lines = hcu * 8, width = wcu * 8;
    float guessScale, fencMean, refMean;
    if (stats[0] && stats[2]) guessScale = sqrtf((float)stats[0] / stats[2]);
    else guessScale = 1.0f;
    fencMean = (float)stats[1] / (lines * width) / (1 << (depth - 8));
    refMean = (float)stats[3] / (lines * width) / (1 << (depth - 8));
    scores[0] = 1; scores[1] = 0;
    if (fabsf(refMean - fencMean) < 0.5f && fabsf(1.f - guessScale) < epsilon) return 0;

    /* wpScalingParam::setFromWeightAndOffset, TComSlice.h:1274-1286 */
    int mindenom = 7, minscale = (int)(guessScale * 128 + 0.5f), minoff = 0, found = 0;
    while (mindenom > 0 && minscale > 127) { mindenom--; minscale >>= 1; }
    if (minscale > 127) minscale = 127;

    unsigned int minscore, origscore;
    origscore = minscore = orc_weight_cost_luma(depth, fenc, ref, stride, wcu, hcu, intra_cost, 0, 0, 0, 0);
    scores[0] = origscore; scores[1] = minscore;
    if (!minscore) return 0;

    int curScale = minscale;
    int curOffset = (int)(fencMean - refMean * curScale / (1 << mindenom) + 0.5f);
    if (curOffset < -128 || curOffset > 127)
    {
        curOffset = curOffset < -128 ? -128 : 127;
        curScale = (int)((1 << mindenom) * (fencMean - curOffset) / refMean + 0.5f);
        curScale = curScale < 0 ? 0 : (curScale > 127 ? 127 : curScale);
    }
    const unsigned int s = orc_weight_cost_luma(depth, fenc, ref, stride, wcu, hcu, intra_cost, 1, curScale, mindenom, curOffset);
    if (s < minscore) { minscore = s; minscale = curScale; minoff = curOffset; found = 1; }
    while (mindenom > 0 && !(minscale & 1)) { mindenom--; minscale >>= 1; }
    scores[1] = minscore;
    if (!found || (minscale == 1 << mindenom && minoff == 0) || (float)minscore / origscore > 0.998f) return 0;
    out[0] = minscale; out[1] = mindenom; out[2] = minoff;
    return 1;
}


void orc_bipred_batch(int depth, const void *fenc, intptr_t fstride, const void *ref0, const void *ref1, intptr_t rstride,
                      const orc_bipred_job *jobs, int n, int32_t *out /* 2 per job */)
{
    for (int i = 0; i < n; i++)
    {
        const orc_bipred_job *j = jobs + i;
        const intptr_t fo = (intptr_t)j->pu_y * fstride + j->pu_x, ro = (intptr_t)j->pu_y * rstride + j->pu_x;
        if (depth <= 8)
            bipred_cost_8((const uint8_t*)fenc + fo, fstride, (const uint8_t*)ref0 + ro, (const uint8_t*)ref1 + ro, rstride,
                          j->w, j->h, depth, j->mv0[0], j->mv0[1], j->mv1[0], j->mv1[1], out + 2 * i);
        else
            bipred_cost_16((const uint16_t*)fenc + fo, fstride, (const uint16_t*)ref0 + ro, (const uint16_t*)ref1 + ro, rstride,
                           j->w, j->h, depth, j->mv0[0], j->mv0[1], j->mv1[0], j->mv1[1], out + 2 * i);
    }
}


void orc_merge_sa8d_batch(int depth, const void *fenc, intptr_t fstride, const void *const *refs, const intptr_t *rstrides,
                          const orc_merge_job *jobs, int n, int32_t *out)
{
    for (int i = 0; i < n; i++)
    {
        const orc_merge_job *j = jobs + i;
        const intptr_t fo = (intptr_t)j->pu_y * fstride + j->pu_x;
        const intptr_t s0 = rstrides[j->ref0], s1 = rstrides[j->ref1];
        const intptr_t o0 = (intptr_t)j->pu_y * s0 + j->pu_x, o1 = (intptr_t)j->pu_y * s1 + j->pu_x;
        if (depth <= 8)
            out[i] = merge_sa8d_8((const uint8_t*)fenc + fo, fstride, (const uint8_t*)refs[j->ref0] + o0, s0,
                                  (const uint8_t*)refs[j->ref1] + o1, s1, j->size, depth, j->dir, j->mv0[0], j->mv0[1], j->mv1[0], j->mv1[1]);
        else
            out[i] = merge_sa8d_16((const uint16_t*)fenc + fo, fstride, (const uint16_t*)refs[j->ref0] + o0, s0,
                                   (const uint16_t*)refs[j->ref1] + o1, s1, j->size, depth, j->dir, j->mv0[0], j->mv0[1], j->mv1[0], j->mv1[1]);
    }
}


void orc_mc_batch(int depth, const void *const *ref_planes, const intptr_t *ref_strides, const orc_mc_weight *weights,
                  void *const pred[3], const intptr_t pred_strides[2], const orc_mc_job *jobs, int n)
{
    for (int i = 0; i < n; i++)
    {
        const orc_mc_job *j = jobs + i;
        const void *const *p0 = ref_planes + 3 * j->ref0, *const *p1 = ref_planes + 3 * j->ref1;
        const intptr_t *s0 = ref_strides + 2 * j->ref0, *s1 = ref_strides + 2 * j->ref1;
        const orc_mc_weight *w0 = weights ? weights + j->ref0 : 0, *w1 = weights ? weights + j->ref1 : 0;
        if (depth <= 8)
            mc_pu_8((const uint8_t *const *)p0, s0, (const uint8_t *const *)p1, s1, (uint8_t *const *)pred, pred_strides,
                    j->pu_x, j->pu_y, j->w, j->h, depth, j->dir, j->flags, j->mv0[0], j->mv0[1], j->mv1[0], j->mv1[1], w0, w1);
        else
            mc_pu_16((const uint16_t *const *)p0, s0, (const uint16_t *const *)p1, s1, (uint16_t *const *)pred, pred_strides,
                     j->pu_x, j->pu_y, j->w, j->h, depth, j->dir, j->flags, j->mv0[0], j->mv0[1], j->mv1[0], j->mv1[1], w0, w1);
    }
}
