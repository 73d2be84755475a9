/* TEST INFRASTRUCTURE ONLY -- not part of the shipped engine.
 *
 * Thin extern "C" wrapper around the REFERENCE's own classes, compiled together
 * with the reference's source files (where they lie under /root/reference) by
 * oracle/Makefile into oracle/_ref/libx265ref{8,10}.so.  Nothing from the
 * reference is copied: this file only #includes its headers at build time and
 * calls MotionEstimate / EncoderPrimitives / BitCost as the encoder does.
 *
 * Used to (1) validate the plain-C restatement in oracle/me_oracle.c and
 * (2) generate the golden vectors committed under tests/golden/.
 *
 * Call sites mirrored:
 *   primitives table setup      source/common/primitives.cpp:124-144
 *   MotionEstimate API          source/encoder/motion.h:66-108
 *   lookahead CU driver         source/encoder/slicetype.cpp:1401-1470
 *   lowres plane generation     source/common/lowres.cpp:155-166
 *   lookahead frame cost        source/encoder/slicetype.cpp:1132-1221, :1342-1387, :1401-1582
 */
#include "TLibCommon/TComRom.h"
#include "primitives.h"
#include "common.h"
#include "lowres.h"
#include "motion.h"
#include "bitcost.h"
#include "mv.h"
#include "reference.h"
#include "hardwareC/inter.h"
#include "TLibCommon/TComSlice.h"
#include "TLibCommon/TComPicYuv.h"
#include "slicetype.h"
#include "TLibCommon/TComYuv.h"
#include "TShortYUV.h"
#include "TLibCommon/TComWeightPrediction.h"

#include <string.h>
#include <stdlib.h>
#include <stdint.h>

namespace x265 {
void Setup_C_PixelPrimitives(EncoderPrimitives &p);
void Setup_C_IPFilterPrimitives(EncoderPrimitives &p);
void Setup_C_IPredPrimitives(EncoderPrimitives &p);
}

using namespace x265;

namespace {
/* MotionEstimate keeps its cost pointers protected; a subclass lets the
 * harness read the table the reference built. */
struct ProbeME : public MotionEstimate
{
    const uint16_t *costTable() const { return m_cost; }
};

bool g_ready = false;
}

extern "C" {

int ref_pixel_size(void) { return (int)sizeof(pixel); }

/* Fill the global primitive table with the reference's C primitives and apply
 * the sa8d_inter -> satd aliasing of primitives.cpp:138-144 (the full
 * x265_setup_primitives additionally initialises DCT/intra/CABAC tables that
 * this path never touches). */
void ref_setup(int bitDepth)
{
#if HIGH_BIT_DEPTH
    g_bitDepth = bitDepth;
#else
    (void)bitDepth;
#endif
    if (g_ready) return;
    Setup_C_PixelPrimitives(primitives);
    Setup_C_IPFilterPrimitives(primitives);
    Setup_C_IPredPrimitives(primitives);
    initROM();                  /* g_convertToBit, read by estimateCUCost and the planar predictor */
    primitives.sa8d_inter[LUMA_4x4]   = primitives.satd[LUMA_4x4];
    primitives.sa8d_inter[LUMA_4x8]   = primitives.satd[LUMA_4x8];
    primitives.sa8d_inter[LUMA_4x16]  = primitives.satd[LUMA_4x16];
    primitives.sa8d_inter[LUMA_8x4]   = primitives.satd[LUMA_8x4];
    primitives.sa8d_inter[LUMA_16x4]  = primitives.satd[LUMA_16x4];
    primitives.sa8d_inter[LUMA_16x12] = primitives.satd[LUMA_16x12];
    primitives.sa8d_inter[LUMA_12x16] = primitives.satd[LUMA_12x16];
    g_maxCUWidth = 64;
    g_maxCUHeight = 64;
    g_ready = true;
}

int ref_partition(int w, int h) { return partitionFromSizes(w, h); }

int ref_sad(int part, pixel *a, intptr_t sa, pixel *b, intptr_t sb)  { return primitives.sad[part](a, sa, b, sb); }
int ref_satd(int part, pixel *a, intptr_t sa, pixel *b, intptr_t sb) { return primitives.satd[part](a, sa, b, sb); }
int ref_sa8d_inter(int part, pixel *a, intptr_t sa, pixel *b, intptr_t sb) { return primitives.sa8d_inter[part](a, sa, b, sb); }
void ref_sad_x3(int part, pixel *fenc, pixel *r0, pixel *r1, pixel *r2, intptr_t s, int32_t *res)
{ primitives.sad_x3[part](fenc, r0, r1, r2, s, res); }
void ref_sad_x4(int part, pixel *fenc, pixel *r0, pixel *r1, pixel *r2, pixel *r3, intptr_t s, int32_t *res)
{ primitives.sad_x4[part](fenc, r0, r1, r2, r3, s, res); }
void ref_pixelavg_pp(int part, pixel *dst, intptr_t ds, pixel *a, intptr_t sa, pixel *b, intptr_t sb)
{ primitives.pixelavg_pp[part](dst, ds, a, sa, b, sb, 32); }
void ref_luma_hpp(int part, pixel *src, intptr_t ss, pixel *dst, intptr_t ds, int c) { primitives.luma_hpp[part](src, ss, dst, ds, c); }
void ref_luma_vpp(int part, pixel *src, intptr_t ss, pixel *dst, intptr_t ds, int c) { primitives.luma_vpp[part](src, ss, dst, ds, c); }
void ref_luma_hps(int part, pixel *src, intptr_t ss, int16_t *dst, intptr_t ds, int c, int ext) { primitives.luma_hps[part](src, ss, dst, ds, c, ext); }
void ref_luma_vsp(int part, int16_t *src, intptr_t ss, pixel *dst, intptr_t ds, int c) { primitives.luma_vsp[part](src, ss, dst, ds, c); }
void ref_lowres_core(pixel *src, pixel *d0, pixel *dh, pixel *dv, pixel *dc, intptr_t ss, intptr_t ds, int w, int h)
{ primitives.frame_init_lowres_core(src, d0, dh, dv, dc, ss, ds, w, h); }

/* lambda-scaled MVD cost table for one QP, entries [-32767, 32767] written to
 * out[0..65534] (bitcost.cpp:34-59). */
void ref_cost_table(int qp, uint16_t *out)
{
    ProbeME me;
    me.setQP(qp);
    const uint16_t *t = me.costTable();
    for (int i = -32767; i <= 32767; i++)
        out[i + 32767] = t[i];
}

/* ---- classic MotionEstimate::motionEstimate, one PU ---- */
struct RefME
{
    MotionEstimate me;
    ReferencePlanes ref;
};

void *ref_me_create(void) { return new RefME; }
void ref_me_destroy(void *h) { delete (RefME*)h; }

void ref_me_config(void *h, pixel *fencPlane, intptr_t fencStride, int qp, int method, int subme)
{
    RefME *r = (RefME*)h;
    r->me.setSourcePlane(fencPlane, fencStride);
    r->me.setQP(qp);
    r->me.setSearchMethod(method);
    r->me.setSubpelRefine(subme);
}

/* planes[0] = full-pel plane origin; for lowres planes[0..3] = the four hpel
 * planes (lowres.h:44). mv arguments are {x,y} int16 pairs. */
int ref_me_search(void *h, int offset, int w, int hgt,
                  pixel **planes, int stride, int isLowres,
                  const int16_t *mvmin, const int16_t *mvmax, const int16_t *qmvp,
                  int numc, const int16_t *mvc, int merange, int16_t *outmv)
{
    RefME *r = (RefME*)h;
    r->ref.fpelPlane = planes[0];
    r->ref.lumaStride = stride;
    r->ref.isLowres = !!isLowres;
    if (isLowres)
        for (int i = 0; i < 4; i++) r->ref.lowresPlane[i] = planes[i];
    r->me.setSourcePU(offset, w, hgt);
    MV cand[8];
    for (int i = 0; i < numc && i < 8; i++) cand[i] = MV(mvc[2 * i], mvc[2 * i + 1]);
    MV out;
    int cost = r->me.motionEstimate(&r->ref, MV(mvmin[0], mvmin[1]), MV(mvmax[0], mvmax[1]),
                                    MV(qmvp[0], qmvp[1]), numc, cand, merange, out);
    outmv[0] = out.x;
    outmv[1] = out.y;
    return cost;
}

/* Batch form of ref_me_search for timing the reference without per-call binding overhead:
 * jobs / results use the 40-byte / 16-byte records of include/x265_amd_me.h (sad_calls and
 * subpel_calls are left 0: the reference does not count them). */
struct HarnessJob { int16_t pu_x, pu_y; uint8_t w, h, numc, reserved; int16_t mvmin[2], mvmax[2], qmvp[2]; int16_t mvc[4][2]; int32_t merange; };
struct HarnessResult { int16_t mv[2]; int32_t cost; uint32_t sad_calls, subpel_calls; };
void ref_me_search_batch(void *h, pixel *plane, int stride, const HarnessJob *jobs, int n, HarnessResult *out)
{
    RefME *r = (RefME*)h;
    r->ref.fpelPlane = plane;
    r->ref.lumaStride = stride;
    r->ref.isLowres = false;
    for (int i = 0; i < n; i++)
    {
        const HarnessJob &j = jobs[i];
        r->me.setSourcePU(j.pu_y * stride + j.pu_x, j.w, j.h);
        MV cand[4], mv;
        for (int k = 0; k < j.numc && k < 4; k++) cand[k] = MV(j.mvc[k][0], j.mvc[k][1]);
        out[i].cost = r->me.motionEstimate(&r->ref, MV(j.mvmin[0], j.mvmin[1]), MV(j.mvmax[0], j.mvmax[1]),
                                           MV(j.qmvp[0], j.qmvp[1]), j.numc, cand, j.merange, mv);
        out[i].mv[0] = mv.x; out[i].mv[1] = mv.y;
        out[i].sad_calls = out[i].subpel_calls = 0;
    }
}

/* subpelCompare at one qpel MV with sad (cmp=0) or satd (cmp=1); motion.cpp:1651 */
int ref_me_subpel_compare(void *h, int offset, int w, int hgt, pixel *plane, int stride,
                          int qx, int qy, int cmp)
{
    RefME *r = (RefME*)h;
    r->ref.fpelPlane = plane;
    r->ref.lumaStride = stride;
    r->ref.isLowres = false;
    r->me.setSourcePU(offset, w, hgt);
    int part = partitionFromSizes(w, hgt);
    return r->me.subpelCompare(&r->ref, MV(qx, qy), cmp ? primitives.satd[part] : primitives.sad[part]);
}

/* ---- lookahead: one lowres MV field (one list), ME part of
 * EstimateRow::estimateCUCost, slicetype.cpp:1401-1470, visited in the
 * reference's order (rows bottom->top, CUs right->left, slicetype.cpp:1344,1356).
 * The MotionEstimate and ReferencePlanes objects are the reference's. */
static inline int16_t med3(int16_t a, int16_t b, int16_t c)
{
    /* median of three (slicetype.cpp:42-51 computes the same value branch-free) */
    int16_t lo = a < b ? a : b, hi = a < b ? b : a;
    int16_t m = lo > c ? lo : c;
    return m < hi ? m : hi;
}

void ref_lookahead_field(pixel *fencLowres, pixel **refPlanes, int stride,
                         int widthInCU, int heightInCU, int qp, int method, int subme, int merange,
                         int16_t *outMvs /* 2*ncu */, int32_t *outCosts /* ncu */)
{
    MotionEstimate me;
    ReferencePlanes ref;
    me.setQP(qp);
    me.setSearchMethod(method);
    me.setSubpelRefine(subme);
    me.setSourcePlane(fencLowres, stride);
    ref.fpelPlane = refPlanes[0];
    ref.lumaStride = stride;
    ref.isLowres = true;
    for (int i = 0; i < 4; i++) ref.lowresPlane[i] = refPlanes[i];
    MV *mvs = (MV*)outMvs;
    const int cuSize = 8;
    for (int cuy = heightInCU - 1; cuy >= 0; cuy--)
    {
        for (int cux = widthInCU - 1; cux >= 0; cux--)
        {
            const int cuXY = cux + cuy * widthInCU;
            const int pelOffset = cuSize * cux + cuSize * cuy * stride;
            me.setSourcePU(pelOffset, cuSize, cuSize);
            MV mvmin, mvmax;
            mvmin.x = (uint16_t)(-cux * cuSize - 8);
            mvmin.y = (uint16_t)(-cuy * cuSize - 8);
            mvmax.x = (uint16_t)((widthInCU - cux - 1) * cuSize + 8);
            mvmax.y = (uint16_t)((heightInCU - cuy - 1) * cuSize + 8);
            int numc = 0;
            MV mvc[4], mvp;
            MV *fenc_mv = &mvs[cuXY];
            mvc[0] = 0;
            mvc[2] = 0;
            if (cux < widthInCU - 1) mvc[numc++] = fenc_mv[1];
            if (cuy < heightInCU - 1)
            {
                mvc[numc++] = fenc_mv[widthInCU];
                if (cux > 0) mvc[numc++] = fenc_mv[widthInCU - 1];
                if (cux < widthInCU - 1) mvc[numc++] = fenc_mv[widthInCU + 1];
            }
            if (numc <= 1) mvp = mvc[0];
            else { mvp.x = med3(mvc[0].x, mvc[1].x, mvc[2].x); mvp.y = med3(mvc[0].y, mvc[1].y, mvc[2].y); }
            outCosts[cuXY] = me.motionEstimate(&ref, mvmin, mvmax, mvp, numc, mvc, merange, mvs[cuXY]);
        }
    }
}

/* ---- optional call trace: records the reference-plane address of every
 * full-pel sad / sad_x3 / sad_x4 call of one partition, so visit order and
 * candidate counts can be compared with the restatement ---- */
static pixelcmp_t    t_sad0;
static pixelcmp_x3_t t_x30;
static pixelcmp_x4_t t_x40;
static int64_t *t_buf; static int t_n, t_cap, t_part = -1; static pixel *t_base;
static inline void t_log(pixel *p) { if (t_n < t_cap) t_buf[t_n] = (int64_t)(p - t_base); t_n++; }
static int t_sad(pixel *a, intptr_t sa, pixel *b, intptr_t sb) { t_log(b); return t_sad0(a, sa, b, sb); }
static void t_x3(pixel *f, pixel *a, pixel *b, pixel *c, intptr_t s, int32_t *r) { t_log(a); t_log(b); t_log(c); t_x30(f, a, b, c, s, r); }
static void t_x4(pixel *f, pixel *a, pixel *b, pixel *c, pixel *d, intptr_t s, int32_t *r) { t_log(a); t_log(b); t_log(c); t_log(d); t_x40(f, a, b, c, d, s, r); }

void ref_trace_begin(int part, pixel *base, int64_t *buf, int cap)
{
    t_part = part; t_base = base; t_buf = buf; t_cap = cap; t_n = 0;
    t_sad0 = primitives.sad[part]; t_x30 = primitives.sad_x3[part]; t_x40 = primitives.sad_x4[part];
    primitives.sad[part] = t_sad; primitives.sad_x3[part] = t_x3; primitives.sad_x4[part] = t_x4;
}

int ref_trace_end(void)
{
    if (t_part >= 0)
    {
        primitives.sad[t_part] = t_sad0; primitives.sad_x3[t_part] = t_x30; primitives.sad_x4[t_part] = t_x40;
        t_part = -1;
    }
    return t_n;
}

/* ---- Rockchip path: the 15-argument MotionEstimate::motionEstimate overload
 * (motion.cpp:1250-1410) = CIME + RIME + FME for one CU, called the way
 * TEncSearch::predInterSearch does (TEncSearch.cpp:3820-3822). The overload keeps
 * the CTU's PMVs in a function-static array, so CUs of one CTU must be issued
 * after that CTU's isSavePmv call, as the encoder does. ---- */
void ref_rk_ds_plane(pixel *origOrigin, int width, int height, int stride, short *dsOrigin, int dsStride,
                     int marginY4, int marginX4)
{
    MotionReference mr;
    mr.downSampleAndSum(origOrigin, width, height, stride, dsOrigin, dsStride);
    mr.extendEdge(dsOrigin, marginY4, marginX4, dsStride, width / 4, height / 4);
}

int ref_rk_me(void *h, pixel *fencPlane, intptr_t fencStride, int qp,
              pixel *reconOrigin, int stride, short *dsOrigin,
              int offset, int size, const int16_t *mvmin, const int16_t *mvmax,
              const int16_t *mvNeighBor /* 10 x {x,y} */, int nMvNeighBor, int isHaveMvd, int isSavePmv,
              int offsIdx, int depth, intptr_t blockOffset_ds, int stride_ds, int validW, int validH, int refIdx,
              int32_t *out /* qmv.x qmv.y cost | rime x y cost | fme x y cost | cime x y cost */)
{
    RefME *r = (RefME*)h;
    r->me.setSourcePlane(fencPlane, fencStride);
    r->me.setQP(qp);
    r->ref.fpelPlane = reconOrigin;
    r->ref.fpelPlaneOrigDS = dsOrigin;
    r->ref.lumaStride = stride;
    r->ref.isLowres = false;
    r->me.setSourcePU(offset, size, size);
    MV nb[10];
    for (int i = 0; i < 10; i++) nb[i] = MV(mvNeighBor[2 * i], mvNeighBor[2 * i + 1]);
    MV o;
    int cost = r->me.motionEstimate(&r->ref, MV(mvmin[0], mvmin[1]), MV(mvmax[0], mvmax[1]), o, nb, nMvNeighBor,
                                    !!isHaveMvd, !!isSavePmv, offsIdx, depth, blockOffset_ds, stride_ds, validW, validH, refIdx);
    out[0] = o.x; out[1] = o.y; out[2] = cost;
    out[3] = g_RimeMv[refIdx][offsIdx].x; out[4] = g_RimeMv[refIdx][offsIdx].y; out[5] = g_RimeMv[refIdx][offsIdx].nSadCost;
    out[6] = g_FmeMv[refIdx][offsIdx].x; out[7] = g_FmeMv[refIdx][offsIdx].y; out[8] = g_FmeMv[refIdx][offsIdx].nSadCost;
    out[9] = g_leftPMV[refIdx].x; out[10] = g_leftPMV[refIdx].y; out[11] = g_leftPMV[refIdx].nSadCost;
    return cost;
}

/* ---- C-model twin of the coarse stage: CoarseIntMotionEstimation::CIME
 * (hardwareC/inter.cpp:679-734) driven the way TEncCu::CimePrepare does
 * (TEncCu.cpp:361-474): picture size, one P-slice reference, search-window size, CTU
 * address, QP, the original picture, the reference to down-sample, the 28 neighbour MVs.
 * The model prefetches its own edge-replicated window (inter.cpp:296-647) and derives
 * its three cost predictors with setMvp (:755-842); both are returned so that the
 * x265-side path can be given the same predictors. 8-bit only, as the model is. ---- */
#if !HIGH_BIT_DEPTH
int ref_cmodel_cime(const unsigned char *cur, int curStride, const unsigned char *refPic, int refStride,
                    int picW, int picH, int qp, int ctuAddr, int swW, int swH,
                    const int16_t *neigh /* nAllNeighMv x {x,y} */, int16_t *out /* mv.x mv.y mvp[3].{x,y} */)
{
    CoarseIntMotionEstimation c;
    c.setPicWidth(picW); c.setPicHeight(picH);
    c.setRefPicNum(1, 0); c.setRefPicNum(0, 1);
    c.setSliceType(p_slice);
    c.Create(swW, swH);
    c.setCtuPosInPic(ctuAddr);
    c.setQP((char)qp);
    c.setOrigPic(const_cast<unsigned char*>(cur), curStride);
    c.setDownSamplePic(const_cast<unsigned char*>(refPic), refStride, 0);
    for (int i = 0; i < nAllNeighMv; i++)
    {
        Mv m; m.x = neigh[2 * i]; m.y = neigh[2 * i + 1];
        c.setNeighMv(m, 0, i);
    }
    c.CIME(c.getOrigPic(), c.getRefPicDS(0), 0);
    Mv r = c.getCimeMv(0), mvp[3];
    c.setMvp(mvp, 0);
    out[0] = r.x; out[1] = r.y;
    for (int i = 0; i < 3; i++) { out[2 + 2 * i] = mvp[i].x; out[3 + 2 * i] = mvp[i].y; }
    const int cost = r.nSadCost;
    c.Destroy();
    return cost;
}

/* C-model twin of the refine + fractional stages: RefineIntMotionEstimation::RimeAndFme
 * (hardwareC/inter.cpp:2521-2624) for all 85 CUs of one CTU, set up the way
 * TEncCu::RimeAndFmePrepare does (TEncCu.cpp:509-687): pictures copied in, the CTU's
 * neighbour MVs and coarse MV given to setPmv, the three CTU search windows prefetched,
 * then per CU its own windows. Chroma is not part of the motion search; flat planes stand
 * in for it. out: per CU {rime.x, rime.y, rime.cost, fme.x, fme.y, fme.cost} (cost -1 for CUs
 * outside the picture), then the three window centres and the three cost predictors the
 * model derived. */
int ref_cmodel_rime_fme(const unsigned char *cur, int curStride, const unsigned char *refPic, int refStride,
                        int picW, int picH, int qp, int ctuAddr, const int16_t *neigh, const int16_t *cimeMv,
                        int32_t *out /* 85 * 6 + 12 */)
{
    static RefineIntMotionEstimation r;
    r.setPicWidth(picW); r.setPicHeight(picH);
    r.setRefPicNum(1, 0); r.setRefPicNum(0, 1);
    r.setSliceType(p_slice);
    r.CreatePicInfo();
    const int cw = picW / 2, ch = picH / 2;
    unsigned char *flat = new unsigned char[cw * ch];
    memset(flat, 128, cw * ch);
    r.setOrigPic(const_cast<unsigned char*>(cur), curStride, flat, cw, flat, cw);
    r.setRefPic(const_cast<unsigned char*>(refPic), refStride, flat, cw, flat, cw, 0);
    r.setCtuSize(64);
    r.setCtuPosInPic(ctuAddr);
    r.setQP((char)qp);
    Mv nb[nAllNeighMv], cm;
    for (int i = 0; i < nAllNeighMv; i++) { nb[i].x = neigh[2 * i]; nb[i].y = neigh[2 * i + 1]; }
    cm.x = cimeMv[0]; cm.y = cimeMv[1];
    r.setPmv(nb, cm, 0);
    r.PrefetchCtuLumaInfo(0);
    r.PrefetchCtuChromaInfo(0);
    for (int cu = 0; cu < 85; cu++)
    {
        const int size = cu < 64 ? 8 : (cu < 80 ? 16 : (cu < 84 ? 32 : 64));
        r.setCuPosInCtu(cu);
        r.CreateCuInfo(size + 2 * (nRimeWidth + 4), size + 2 * (nRimeHeight + 4), size);
        int32_t *o = out + 6 * cu;
        if (r.PrefetchCuInfo(0))
        {
            r.RimeAndFme(0);
            Mv a = r.getRimeMv(0), b = r.getFmeMv(0);
            o[0] = a.x; o[1] = a.y; o[2] = a.nSadCost; o[3] = b.x; o[4] = b.y; o[5] = b.nSadCost;
        }
        else { o[0] = o[1] = o[3] = o[4] = 0; o[2] = o[5] = -1; }
        r.DestroyCuInfo();
    }
    Mv mvp[3];
    r.setMvp(mvp, nb);
    for (int i = 0; i < 3; i++)
    {
        Mv p; r.getPmv(p, 0, i);
        out[510 + 2 * i] = p.x; out[511 + 2 * i] = p.y;
        out[516 + 2 * i] = mvp[i].x; out[517 + 2 * i] = mvp[i].y;
    }
    r.DestroyPicInfo();
    delete[] flat;
    return 0;
}
#else
int ref_cmodel_rime_fme(const unsigned char*, int, const unsigned char*, int, int, int, int, int, const int16_t*, const int16_t*, int32_t*)
{
    return -1;
}
int ref_cmodel_cime(const unsigned char*, int, const unsigned char*, int, int, int, int, int, int, int, const int16_t*, int16_t*)
{
    return -1;
}
#endif

/* ---- lookahead: cost of one frame b against (p0, p1): the reference's own
 * EstimateRow::estimateCUCost (slicetype.cpp:1401-1582) called for every CU in the
 * serial order of CostEstimate::processRow (:1344, :1356: rows bottom->top, CUs
 * right->left), then the row sums added up as estimateFrameCost does (:1193-1204).
 * Lowres objects are filled by hand with the planes and output arrays
 * (lowres.h:101-140). d0 = b - p0, d1 = p1 - b. Arrays not used by the call may be
 * NULL. out64: [0] sum of rows' costEst (the score before the b != p1 scaling of
 * :1208-1209), [1] costEstAq, [2] costIntra, [3] costIntraAq, [4] intraMbs. ---- */
void ref_estimate_frame(pixel **fencPlanes, pixel **ref0Planes, pixel **ref1Planes, int stride,
                        int widthInCU, int heightInCU, int d0, int d1,
                        int doSearch0, int doSearch1, int intraCalculated, int32_t *invQscale,
                        int16_t *mvs0, int32_t *mvCosts0, int16_t *mvs1, int32_t *mvCosts1,
                        int32_t *intraCost, uint16_t *lowresCosts, int32_t *rowSatdsIntra, int32_t *rowSatds,
                        int64_t *out64)
{
    const int p0 = 0, b = d0, p1 = d0 + d1;
    Lowres *frames[2 * X265_BFRAME_MAX + 4];
    memset(frames, 0, sizeof(frames));
    Lowres *lf[3] = { new Lowres, new Lowres, new Lowres };
    pixel **pl[3] = { fencPlanes, ref0Planes, ref1Planes };
    for (int k = 0; k < 3; k++)
    {
        if (!pl[k]) continue;
        for (int i = 0; i < 4; i++) lf[k]->lowresPlane[i] = pl[k][i];
        lf[k]->fpelPlane = pl[k][0];
        lf[k]->lumaStride = stride;
        lf[k]->isLowres = true;
        lf[k]->width = widthInCU * 8;
        lf[k]->lines = heightInCU * 8;
    }
    frames[b] = lf[0];
    if (d0) frames[p0] = lf[1];
    if (d1) frames[p1] = lf[2];
    Lowres *fenc = frames[b];
    fenc->bIntraCalculated = !!intraCalculated;
    fenc->intraCost = intraCost;
    fenc->invQscaleFactor = invQscale;
    fenc->rowSatds[0][0] = rowSatdsIntra;
    fenc->rowSatds[d0][d1] = (d0 || d1) ? rowSatds : rowSatdsIntra;
    fenc->lowresCosts[d0][d1] = lowresCosts;
    if (d0) { fenc->lowresMvs[0][d0 - 1] = (MV*)mvs0; fenc->lowresMvCosts[0][d0 - 1] = mvCosts0; }
    if (d1) { fenc->lowresMvs[1][d1 - 1] = (MV*)mvs1; fenc->lowresMvCosts[1][d1 - 1] = mvCosts1; }
    bool bDoSearch[2] = { !!doSearch0, !!doSearch1 };

    EstimateRow *rows = new EstimateRow[heightInCU];
    for (int i = 0; i < heightInCU; i++)
    {
        rows[i].widthInCU = widthInCU;
        rows[i].heightInCU = heightInCU;
        rows[i].init();
        rows[i].me.setSourcePlane(fenc->lowresPlane[0], fenc->lumaStride);
    }
    for (int row = 0; row < heightInCU; row++)
    {
        const int realrow = heightInCU - 1 - row;
        if (!fenc->bIntraCalculated) fenc->rowSatds[0][0][realrow] = 0;
        fenc->rowSatds[d0][d1][realrow] = 0;
        for (int i = widthInCU - 1; i >= 0; i--)
            rows[row].estimateCUCost(frames, frames[p0], i, realrow, p0, p1, b, bDoSearch);
    }
    memset(out64, 0, 5 * sizeof(int64_t));
    for (int row = 0; row < heightInCU; row++)
    {
        out64[0] += rows[row].costEst;
        out64[1] += rows[row].costEstAq;
        out64[2] += rows[row].costIntra;
        out64[3] += rows[row].costIntraAq;
        out64[4] += rows[row].intraMbs;
    }
    delete[] rows;
    for (int k = 0; k < 3; k++) delete lf[k];
}

/* ---- lookahead weighted prediction: the reference's own CostEstimate::weightsAnalyse and
 * weightCostLuma (slicetype.cpp:1223-1340), reached through a subclass because they are
 * protected. The object is used the way CostEstimate::init leaves it when weightp is on
 * (:1112-1130): four weighted buffers of stride x paddedLines samples and weightedRef pointing
 * at their origins; no thread pool, no TEncCfg (neither function reads it).
 * fencPlanes / refPlanes: origins of the 4 lowres planes of frames b and p0; refBuffers: the
 * start of the ref's padded buffers (Lowres::buffer[i]). stats = { fenc wp_ssd[0], fenc
 * wp_sum[0], ref wp_ssd[0], ref wp_sum[0] } (ratecontrol.cpp:44-56, :185-192).
 * out: [0] isWeighted, [1] inputWeight, [2] log2WeightDenom, [3] inputOffset.
 * wbuf: 4 * stride * paddedLines samples, the weighted buffers when isWeighted. ---- */
namespace {
struct ProbeCE : public CostEstimate
{
    ProbeCE() : CostEstimate(NULL) {}
    void analyse(Lowres **frames, int b, int p0) { weightsAnalyse(frames, b, p0); }
    uint32_t cost(Lowres **frames, int b, pixel *src, wpScalingParam *wp) { return weightCostLuma(frames, b, src, wp); }
};
}

uint32_t ref_weight_cost_luma(pixel *fencPlane0, pixel *src, int stride, int widthInCU, int heightInCU, int32_t *intraCost,
                              int useWp, int scale, int denom, int offset, pixel *scratch /* stride * lines */)
{
    ProbeCE ce;
    Lowres fenc;
    fenc.fpelPlane = fenc.lowresPlane[0] = fencPlane0;
    fenc.lumaStride = stride; fenc.width = widthInCU * 8; fenc.lines = heightInCU * 8;
    fenc.intraCost = intraCost;
    Lowres *frames[1] = { &fenc };
    ce.weightedRef.fpelPlane = scratch;
    wpScalingParam wp;
    SET_WEIGHT(wp, 1, scale, denom, offset);
    return ce.cost(frames, 0, src, useWp ? &wp : NULL);
}

void ref_weights_analyse(pixel **fencPlanes, pixel **refPlanes, pixel **refBuffers, int stride, int paddedLines, int padOffset,
                         int widthInCU, int heightInCU, int32_t *intraCost, const uint64_t *stats, int deltaIndex,
                         pixel *wbuf, int32_t *out, double *costDelta)
{
    ProbeCE ce;
    Lowres fenc, ref;
    for (int i = 0; i < 4; i++)
    {
        fenc.lowresPlane[i] = fencPlanes[i]; ref.lowresPlane[i] = refPlanes[i];
        ref.buffer[i] = refBuffers[i];
        ce.wbuffer[i] = wbuf + (size_t)i * stride * paddedLines;
        ce.weightedRef.lowresPlane[i] = ce.wbuffer[i] + padOffset;
    }
    fenc.fpelPlane = fencPlanes[0]; ref.fpelPlane = refPlanes[0];
    fenc.lumaStride = ref.lumaStride = stride;
    fenc.width = ref.width = widthInCU * 8; fenc.lines = ref.lines = heightInCU * 8;
    fenc.intraCost = intraCost;
    fenc.wp_ssd[0] = stats[0]; fenc.wp_sum[0] = stats[1]; ref.wp_ssd[0] = stats[2]; ref.wp_sum[0] = stats[3];
    fenc.frameNum = deltaIndex; ref.frameNum = 0;
    for (int i = 0; i < X265_BFRAME_MAX + 2; i++) fenc.weightedCostDelta[i] = -1.0;
    ce.weightedRef.fpelPlane = ce.weightedRef.lowresPlane[0];
    ce.weightedRef.lumaStride = stride;
    ce.weightedRef.isLowres = true;
    ce.weightedRef.isWeighted = false;
    ce.paddedLines = paddedLines;
    Lowres *frames[2] = { &ref, &fenc };
    ce.analyse(frames, 1, 0);
    out[0] = ce.weightedRef.isWeighted; out[1] = ce.w.inputWeight; out[2] = ce.w.log2WeightDenom; out[3] = ce.w.inputOffset;
    *costDelta = fenc.weightedCostDelta[deltaIndex];
    for (int i = 0; i < 4; i++) ce.wbuffer[i] = NULL;   /* not the object's to free */
}

void ref_weight_pp(pixel *src, pixel *dst, intptr_t srcStride, intptr_t dstStride, int width, int height,
                   int w0, int round, int shift, int offset)
{
    primitives.weight_pp(src, dst, srcStride, dstStride, width, height, w0, round, shift, offset);
}

/* MotionReference::init + applyWeight (reference.cpp:39-71, :211-256) over all CTU rows of a
 * reconstructed picture described by hand in a TComPicYuv (buffer with margins, stride):
 * the weighted, border-extended plane comes back in outBuf (same geometry). */
int ref_apply_weight(pixel *buf, int width, int height, int marginX, int marginY, int stride,
                     int inputWeight, int log2Denom, int inputOffset, pixel *outBuf)
{
    TComPicYuv pic;
    pic.m_picBufY = buf;
    pic.m_picOrgY = buf + (size_t)marginY * stride + marginX;
    pic.m_picWidth = width; pic.m_picHeight = height; pic.m_stride = stride;
    pic.m_lumaMarginX = marginX; pic.m_lumaMarginY = marginY;
    pic.m_numCuInHeight = (height + g_maxCUHeight - 1) / g_maxCUHeight;
    wpScalingParam wp;
    SET_WEIGHT(wp, 1, inputWeight, log2Denom, inputOffset);
    MotionReference mref;
    if (mref.init(&pic, &wp)) return -1;
    /* CTU row by CTU row, as the frame encoder asks for reference rows (the last row's height is
     * only right when it is weighted on its own, reference.cpp:221-223) */
    const int numRows = pic.m_numCuInHeight;
    for (int r = 1; r <= numRows; r++) mref.applyWeight(r, numRows);
    const size_t rows = (size_t)height + 2 * marginY;
    memcpy(outBuf, mref.fpelPlane - ((size_t)marginY * stride + marginX), rows * stride * sizeof(pixel));
    return 0;
}

/* ---- bi-prediction estimate: the primitive calls of TEncSearch::predInterSearch
 * (TEncSearch.cpp:4346-4368) and of TComPrediction::xPredInterLumaBlk (TComPrediction.cpp:
 * 405-438) that it makes, issued in their order on the reference's own primitives. out[0] =
 * satd against the averaged predictions, out[1] = against the averaged co-located blocks. ---- */
static void harness_pred_luma(pixel *refPU, intptr_t stride, int qx, int qy, int part, int width, pixel *dst, intptr_t dstStride,
                              int16_t *immed)
{
    pixel *src = refPU + (qx >> 2) + (qy >> 2) * stride;
    const int xFrac = qx & 3, yFrac = qy & 3;
    if ((yFrac | xFrac) == 0) primitives.luma_copy_pp[part](dst, dstStride, src, stride);
    else if (yFrac == 0) primitives.luma_hpp[part](src, stride, dst, dstStride, xFrac);
    else if (xFrac == 0) primitives.luma_vpp[part](src, stride, dst, dstStride, yFrac);
    else
    {
        primitives.luma_hps[part](src, stride, immed, width, xFrac, 1);
        primitives.luma_vsp[part](immed + 3 * width, width, dst, dstStride, yFrac);
    }
}

void ref_bipred_cost(pixel *fencPU, intptr_t fencStride, pixel *ref0PU, pixel *ref1PU, intptr_t refStride, int w, int h,
                     const int16_t *mv0, const int16_t *mv1, int32_t *out)
{
    const int part = partitionFromSizes(w, h);
    pixel *p0 = (pixel*)malloc(sizeof(pixel) * 64 * 64 * 3);
    pixel *p1 = p0 + 64 * 64, *avg = p1 + 64 * 64;
    int16_t *immed = (int16_t*)malloc(sizeof(int16_t) * 64 * (64 + 8));
    harness_pred_luma(ref0PU, refStride, mv0[0], mv0[1], part, w, p0, 64, immed);
    harness_pred_luma(ref1PU, refStride, mv1[0], mv1[1], part, w, p1, 64, immed);
    primitives.pixelavg_pp[part](avg, w, p0, 64, p1, 64, 32);
    out[0] = primitives.satd[part](fencPU, fencStride, avg, w);
    primitives.pixelavg_pp[part](avg, w, ref0PU, refStride, ref1PU, refStride, 32);
    out[1] = primitives.satd[part](fencPU, fencStride, avg, w);
    free(immed);
    free(p0);
}

/* ---- merge-candidate estimate (compress.cpp:248-271): the primitive calls that
 * TComPrediction::motionCompensation makes for the luma of one candidate (xPredInterUni ->
 * xPredInterLumaBlk, TComPrediction.cpp:405-476; two lists: the TShortYUV flavour and the reference's own
 * TComYuv::addAvg, TComYuv.cpp:573-612, on CU buffers made with its create()), then
 * primitives.sa8d[log2 size - 3] as compress.cpp:265 calls it. ---- */

/* The 64x64 CU buffers motionCompensation works in (TComPrediction::m_predShortYuv[2], the caller's predYuv): the
 * two 14-bit predictions and the prediction samples, created by the reference's own create() so that strides and
 * chroma geometry are the reference's. */
struct CuBuffers
{
    TShortYUV s0, s1;
    TComYuv pred;
    CuBuffers()
    {
        s0.create(MAX_CU_SIZE, MAX_CU_SIZE, X265_CSP_I420);
        s1.create(MAX_CU_SIZE, MAX_CU_SIZE, X265_CSP_I420);
        pred.create(MAX_CU_SIZE, MAX_CU_SIZE, X265_CSP_I420);
    }
    ~CuBuffers() { s0.destroy(); s1.destroy(); pred.destroy(); }
};
static void harness_pred_luma_short(pixel *refPU, intptr_t stride, int qx, int qy, int part, int width, int height, int16_t *dst,
                                    int16_t *immed)
{
    pixel *ref = refPU + (qx >> 2) + (qy >> 2) * stride;
    const int xFrac = qx & 3, yFrac = qy & 3;
    if ((yFrac | xFrac) == 0) primitives.luma_p2s(ref, stride, dst, width, height);
    else if (yFrac == 0) primitives.luma_hps[part](ref, stride, dst, MAX_CU_SIZE, xFrac, 0);
    else if (xFrac == 0) primitives.ipfilter_ps[FILTER_V_P_S_8](ref, stride, dst, MAX_CU_SIZE, width, height, g_lumaFilter[yFrac]);
    else
    {
        primitives.luma_hps[part](ref, stride, immed, width, xFrac, 1);
        primitives.ipfilter_ss[FILTER_V_S_S_8](immed + 3 * width, width, dst, MAX_CU_SIZE, width, height, yFrac);
    }
}

int ref_merge_sa8d(pixel *fencPU, intptr_t fencStride, pixel *ref0PU, intptr_t ref0Stride, pixel *ref1PU, intptr_t ref1Stride,
                   int size, int dir, const int16_t *mv0, const int16_t *mv1)
{
    const int part = partitionFromSizes(size, size);
    pixel *pred = (pixel*)malloc(sizeof(pixel) * 64 * 64);
    int16_t *immed = (int16_t*)malloc(sizeof(int16_t) * 64 * (64 + 8));
    if (dir == 1) harness_pred_luma(ref0PU, ref0Stride, mv0[0], mv0[1], part, size, pred, 64, immed);
    else if (dir == 2) harness_pred_luma(ref1PU, ref1Stride, mv1[0], mv1[1], part, size, pred, 64, immed);
    else
    {
        CuBuffers cu;
        harness_pred_luma_short(ref0PU, ref0Stride, mv0[0], mv0[1], part, size, size, cu.s0.getLumaAddr(), immed);
        harness_pred_luma_short(ref1PU, ref1Stride, mv1[0], mv1[1], part, size, size, cu.s1.getLumaAddr(), immed);
        cu.pred.addAvg(&cu.s0, &cu.s1, 0, size, size, true, false);         /* TComYuv.cpp:573 */
        for (int y = 0; y < size; y++) memcpy(pred + y * 64, cu.pred.getLumaAddr() + y * cu.pred.getStride(), size * sizeof(pixel));
    }
    const int cost = primitives.sa8d[g_convertToBit[size]](fencPU, fencStride, pred, 64);
    free(immed);
    free(pred);
    return cost;
}

/* ---- motion compensation into prediction samples, luma + 4:2:0 chroma: the primitive calls of
 * TComPrediction::motionCompensation without weighted prediction (TComPrediction.cpp:201-286) for
 * one PU, issued as xPredInterLumaBlk (:405-478) and xPredInterChromaBlk (:491-593) issue them --
 * the chroma tables indexed by the luma partition -- and TComYuv::addAvg's arithmetic
 * (TComYuv.cpp:573-650; the loop is repeated because TComYuv / TComDataCU need a whole encoder
 * around them). refs / pred hold the PU's own sample in each plane (Y, Cb, Cr). ---- */
static void harness_pred_chroma(pixel *refBlk, intptr_t stride, int ex, int ey, int part, int cw, int ch, pixel *dst, intptr_t dstStride,
                                int16_t *immed)
{
    pixel *src = refBlk + (ex >> 3) + (ey >> 3) * stride;
    const int xFrac = ex & 7, yFrac = ey & 7, csp = X265_CSP_I420;
    if ((yFrac | xFrac) == 0) primitives.chroma[csp].copy_pp[part](dst, dstStride, src, stride);
    else if (yFrac == 0) primitives.chroma[csp].filter_hpp[part](src, stride, dst, dstStride, xFrac);
    else if (xFrac == 0) primitives.chroma[csp].filter_vpp[part](src, stride, dst, dstStride, yFrac);
    else
    {
        primitives.chroma[csp].filter_hps[part](src, stride, immed, cw, xFrac, 1);
        primitives.chroma_vsp(immed + (NTAPS_CHROMA / 2 - 1) * cw, cw, dst, dstStride, cw, ch, yFrac);
    }
}

static void harness_pred_chroma_short(pixel *refBlk, intptr_t stride, int ex, int ey, int part, int cw, int ch, int16_t *dst, int16_t *immed)
{
    pixel *src = refBlk + (ex >> 3) + (ey >> 3) * stride;
    const int xFrac = ex & 7, yFrac = ey & 7, csp = X265_CSP_I420, dstStride = MAX_CU_SIZE / 2;
    if ((yFrac | xFrac) == 0) primitives.chroma_p2s(src, stride, dst, cw, ch);
    else if (yFrac == 0) primitives.chroma[csp].filter_hps[part](src, stride, dst, dstStride, xFrac, 0);
    else if (xFrac == 0) primitives.ipfilter_ps[FILTER_V_P_S_4](src, stride, dst, dstStride, cw, ch, g_chromaFilter[yFrac]);
    else
    {
        primitives.chroma[csp].filter_hps[part](src, stride, immed, cw, xFrac, 1);
        primitives.ipfilter_ss[FILTER_V_S_S_4](immed + (NTAPS_CHROMA / 2 - 1) * cw, cw, dst, dstStride, cw, ch, yFrac);
    }
}

struct HarnessWeight { int16_t w[3], offset[3]; uint8_t log2denom[3], reserved; };

/* the derived fields of wpScalingParam as TComWeightPrediction::getWpScaling fills them (TComWeightPrediction.cpp:
 * 534-560; that function itself reads the tables of a slice through a TComDataCU) */
static void harness_wp_scaling(const HarnessWeight *h0, const HarnessWeight *h1, bool bidir, wpScalingParam wp0[3], wpScalingParam wp1[3])
{
    for (int c = 0; c < 3; c++)
    {
        const HarnessWeight *hs[2] = { h0, h1 };
        wpScalingParam *ws[2] = { &wp0[c], &wp1[c] };
        for (int l = 0; l < 2; l++)
        {
            if (!hs[l]) continue;
            ws[l]->bPresentFlag = true;
            ws[l]->log2WeightDenom = hs[l]->log2denom[c]; ws[l]->inputWeight = hs[l]->w[c]; ws[l]->inputOffset = hs[l]->offset[c];
            ws[l]->w = ws[l]->inputWeight;
            ws[l]->o = ws[l]->inputOffset * (1 << (X265_DEPTH - 8));
        }
        if (bidir)
        {
            wp0[c].offset = wp0[c].o + wp1[c].o;
            wp0[c].shift = wp0[c].log2WeightDenom + 1;
            wp0[c].round = 1 << wp0[c].log2WeightDenom;
            wp1[c].offset = wp0[c].offset; wp1[c].shift = wp0[c].shift; wp1[c].round = wp0[c].round;
        }
        else
            for (int l = 0; l < 2; l++)
                if (hs[l])
                {
                    ws[l]->offset = ws[l]->o;
                    ws[l]->shift = ws[l]->log2WeightDenom;
                    ws[l]->round = ws[l]->log2WeightDenom >= 1 ? 1 << (ws[l]->log2WeightDenom - 1) : 0;
                }
    }
}

void ref_mc_pu(pixel **ref0, const intptr_t *ref0Strides, pixel **ref1, const intptr_t *ref1Strides, pixel **pred,
               const intptr_t *predStrides, int w, int h, int dir, int flags, const int16_t *mv0, const int16_t *mv1,
               const HarnessWeight *wp0, const HarnessWeight *wp1)
{
    const int part = partitionFromSizes(w, h), cw = w >> 1, ch = h >> 1;
    const bool weighted = (flags & 4) != 0, bLuma = (flags & 1) != 0, bChroma = (flags & 2) != 0;
    int16_t *immed = (int16_t*)malloc(sizeof(int16_t) * 64 * (64 + 8));
    if (dir == 3 || weighted)
    {
        /* both lists (or any weighted PU): 14-bit predictions into the CU buffers, then the reference's own
         * TComYuv::addAvg (TComYuv.cpp:573-650), TComWeightPrediction::addWeightBi (TComWeightPrediction.cpp:203-313) or
         * addWeightUni (:433-485), as xPredInterBi / xWeightedPredictionBi / xWeightedPredictionUni call them */
        CuBuffers cu;
        int16_t *sh0[3] = { cu.s0.getLumaAddr(), cu.s0.getCbAddr(), cu.s0.getCrAddr() };
        int16_t *sh1[3] = { cu.s1.getLumaAddr(), cu.s1.getCbAddr(), cu.s1.getCrAddr() };
        if (bLuma)
        {
            if (dir & 1) harness_pred_luma_short(ref0[0], ref0Strides[0], mv0[0], mv0[1], part, w, h, sh0[0], immed);
            if (dir & 2) harness_pred_luma_short(ref1[0], ref1Strides[0], mv1[0], mv1[1], part, w, h, sh1[0], immed);
        }
        if (bChroma)
            for (int c = 1; c < 3; c++)
            {
                if (dir & 1) harness_pred_chroma_short(ref0[c], ref0Strides[1], mv0[0], mv0[1], part, cw, ch, sh0[c], immed);
                if (dir & 2) harness_pred_chroma_short(ref1[c], ref1Strides[1], mv1[0], mv1[1], part, cw, ch, sh1[c], immed);
            }
        if (!weighted)
            cu.pred.addAvg(&cu.s0, &cu.s1, 0, w, h, bLuma, bChroma);
        else
        {
            wpScalingParam p0[3], p1[3];
            memset(p0, 0, sizeof(p0)); memset(p1, 0, sizeof(p1));
            harness_wp_scaling((dir & 1) ? wp0 : NULL, (dir & 2) ? wp1 : NULL, dir == 3, p0, p1);
            TComWeightPrediction wpred;
            if (dir == 3) wpred.addWeightBi(&cu.s0, &cu.s1, 0, w, h, p0, p1, &cu.pred, true, bLuma, bChroma);
            else if (dir == 1) wpred.addWeightUni(&cu.s0, 0, w, h, p0, &cu.pred, bLuma, bChroma);
            else wpred.addWeightUni(&cu.s1, 0, w, h, p1, &cu.pred, bLuma, bChroma);
        }
        if (bLuma)
            for (int y = 0; y < h; y++) memcpy(pred[0] + y * predStrides[0], cu.pred.getLumaAddr() + y * cu.pred.getStride(), w * sizeof(pixel));
        if (bChroma)
            for (int y = 0; y < ch; y++)
            {
                memcpy(pred[1] + y * predStrides[1], cu.pred.getCbAddr() + y * cu.pred.getCStride(), cw * sizeof(pixel));
                memcpy(pred[2] + y * predStrides[1], cu.pred.getCrAddr() + y * cu.pred.getCStride(), cw * sizeof(pixel));
            }
    }
    else
    {
        pixel **ref = dir == 1 ? ref0 : ref1;
        const intptr_t *strides = dir == 1 ? ref0Strides : ref1Strides;
        const int16_t *mv = dir == 1 ? mv0 : mv1;
        if (bLuma) harness_pred_luma(ref[0], strides[0], mv[0], mv[1], part, w, pred[0], predStrides[0], immed);
        if (bChroma)
            for (int c = 1; c < 3; c++)
                harness_pred_chroma(ref[c], strides[1], mv[0], mv[1], part, cw, ch, pred[c], predStrides[1], immed);
    }
    free(immed);
}

} // extern "C"
