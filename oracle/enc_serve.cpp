/* TEST INFRASTRUCTURE ONLY -- not part of the shipped engine.
 *
 * The REFERENCE encoder with its motion estimation answered by the HIP engine through the engine's C ABI: the binding a
 * maintainer of the reference would write, kept outside the reference's sources by the linker's --wrap option.
 * oracle/Makefile (target `served`) compiles the reference's own encoder + CLI where the files lie and links this file
 * and x265_amd/libx265_amd_me.so; nothing of the reference is edited or copied, its headers are only #included.
 *
 *   MotionEstimate::motionEstimate, RK overload    caller Lib/TLibEncoder/TEncSearch.cpp:3821
 *       the first call of a CTU (isSavePmv) becomes one x265amd_rk_ctu_batch of that CTU; this call and the calls
 *       for the CTU's other CUs are answered from its result record, including the globals the overload leaves
 *       behind for the hardware C-model (encoder/motion.cpp:1300-1302, :1330-1333, :1361, :1403)
 *   MotionEstimate::motionEstimate, classic        caller encoder/slicetype.cpp:1468 (lookahead)
 *       the first call of a (frame, reference) field becomes one x265amd_lookahead_fields of that field; the
 *       field's calls are answered from it
 *   MotionReference::init(recon, orig, wp)         caller encoder/frameencoder.cpp:625-627
 *       uploads the reconstructed reference and builds its 4x4-sum plane on the device (x265amd_rk_ds_plane)
 *   Lowres::init                                   caller encoder/slicetype.cpp:93-110
 *       uploads the original picture and runs x265amd_lowres_init
 *   DPB::prepareEncode                             caller encoder/encoder.cpp:344 (the frame about to be compressed)
 *   BitCost::setQP                                 the QP of each MotionEstimate object
 *
 * Environment:
 *   X265AMD_SERVE        "gpu"    answer from the engine
 *                        "check"  answer from the engine, run the reference's own function as well and abort on the
 *                                 first difference in MV, cost or the globals
 *                        unset    the reference's own functions, timed (the CPU leg of the comparison)
 *   X265AMD_SERVE_STATS  file that receives one JSON line at exit: calls, launches, seconds spent inside the wrapped
 *                        entry points
 * Anything the binding does not cover (weighted references, planes it has not seen) falls through to the reference's
 * own function and is counted as `fallback`.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>
#include <pthread.h>
#include <map>
#include <vector>

#define protected public
#define private public
#include "TLibCommon/TComRom.h"
#include "TLibCommon/TComPic.h"
#include "TLibCommon/TComPicYuv.h"
#include "TLibCommon/TComSlice.h"
#include "primitives.h"
#include "common.h"
#include "lowres.h"
#include "motion.h"
#include "bitcost.h"
#include "mv.h"
#include "reference.h"
#include "TLibEncoder/TEncCfg.h"
#include "dpb.h"
#include "hardwareC/inter.h"
#undef protected
#undef private

#include "../include/x265_amd_me.h"

using namespace x265;

extern short OffsFromCtu64[85][2];      /* hardwareC/inter.cpp:23 */

namespace {

enum { MODE_CPU = 0, MODE_GPU = 1, MODE_CHECK = 2 };
int g_mode = -1;
x265amd_ctx *g_ctx = NULL;
pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;

struct Stats
{
    long rk_calls, rk_ctus, la_calls, la_fields, fallback, planes_up, lowres_up;
    double rk_s, la_s, up_s;
} g_st;

double now()
{
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

void die(const char *what)
{
    fprintf(stderr, "enc_serve: %s%s%s\n", what, g_ctx ? ": " : "", g_ctx ? x265amd_last_error(g_ctx) : "");
    abort();
}

void report()
{
    const char *path = getenv("X265AMD_SERVE_STATS");
    FILE *f = path && *path ? fopen(path, "w") : NULL;
    if (!f) return;
    fprintf(f, "{\"mode\": \"%s\", \"rk_calls\": %ld, \"rk_ctu_launches\": %ld, \"lookahead_calls\": %ld, "
            "\"lookahead_field_launches\": %ld, \"fallback_calls\": %ld, \"planes_uploaded\": %ld, \"lowres_inits\": %ld, "
            "\"rk_seconds\": %.6f, \"lookahead_seconds\": %.6f, \"upload_seconds\": %.6f}\n",
            g_mode == MODE_GPU ? "gpu" : g_mode == MODE_CHECK ? "check" : "cpu", g_st.rk_calls, g_st.rk_ctus, g_st.la_calls,
            g_st.la_fields, g_st.fallback, g_st.planes_up, g_st.lowres_up, g_st.rk_s, g_st.la_s, g_st.up_s);
    fclose(f);
}

int mode()
{
    if (g_mode < 0)
    {
        const char *m = getenv("X265AMD_SERVE");
        g_mode = !m || !*m ? MODE_CPU : !strcmp(m, "gpu") ? MODE_GPU : !strcmp(m, "check") ? MODE_CHECK : MODE_CPU;
        if (g_mode != MODE_CPU && x265amd_create(0, 8, &g_ctx) != X265AMD_OK)
        {
            fprintf(stderr, "enc_serve: no engine (x265amd_create failed)\n");
            abort();
        }
        atexit(report);
    }
    return g_mode;
}

struct Locked
{
    Locked() { pthread_mutex_lock(&g_lock); }
    ~Locked() { pthread_mutex_unlock(&g_lock); }
};

/* ---- device copies of the encoder's pictures ---- */

x265amd_plane upload(TComPicYuv *pic)
{
    x265amd_plane p;
    if (x265amd_plane_upload(g_ctx, pic->getBufY(), pic->getWidth(), pic->getHeight(), pic->getLumaMarginX(), pic->getLumaMarginY(),
                             pic->getStride(), &p) != X265AMD_OK)
        die("plane upload");
    g_st.planes_up++;
    return p;
}

struct RefPlanes { x265amd_plane recon, ds; };
std::map<const void*, RefPlanes> g_refs;          /* MotionReference::fpelPlane -> device planes */
struct LowPlanes { x265amd_plane p[4]; int wcu, hcu; };
std::map<const void*, LowPlanes> g_low;           /* Lowres::lowresPlane[0] -> device planes */
std::map<const void*, int> g_qpOf;                /* BitCost object -> last setQP */

TComPicYuv *g_curOrg = NULL;                      /* source picture of the frame being compressed */
bool g_fencValid = false;
x265amd_plane g_fenc;

/* one CTU's answers per reference index, and the overload's function-static Pmv array (motion.cpp:1269) */
struct CtuAnswers { bool valid; int x, y; x265amd_rk_result res; };
CtuAnswers g_ctu[nMaxRefPic];
MV g_pmv[nMaxRefPic][nNeightMv + 1];

/* one lookahead field's answers */
struct Field { std::vector<int16_t> mv; std::vector<int32_t> cost; int left; };
typedef std::pair<const void*, const void*> FieldKey;
std::map<FieldKey, Field> g_fields;

void dropFieldsOf(const void *plane)
{
    for (std::map<FieldKey, Field>::iterator it = g_fields.begin(); it != g_fields.end();)
        if (it->first.first == plane || it->first.second == plane) g_fields.erase(it++);
        else ++it;
}

}

extern "C" {

/* ---- DPB::prepareEncode ---- */
void __real__ZN4x2653DPB13prepareEncodeEPNS_7TComPicE(DPB *self, TComPic *pic);
void __wrap__ZN4x2653DPB13prepareEncodeEPNS_7TComPicE(DPB *self, TComPic *pic)
{
    __real__ZN4x2653DPB13prepareEncodeEPNS_7TComPicE(self, pic);
    if (mode() == MODE_CPU) return;
    Locked guard;
    g_curOrg = pic->getPicYuvOrg();
    if (g_fencValid) x265amd_plane_free(g_ctx, &g_fenc);
    g_fencValid = false;                          /* uploaded by the frame's first search */
    for (int i = 0; i < nMaxRefPic; i++) g_ctu[i].valid = false;
}

/* ---- BitCost::setQP ---- */
void __real__ZN4x2657BitCost5setQPEj(BitCost *self, unsigned int qp);
void __wrap__ZN4x2657BitCost5setQPEj(BitCost *self, unsigned int qp)
{
    {
        Locked guard;
        g_qpOf[self] = (int)qp;
    }
    __real__ZN4x2657BitCost5setQPEj(self, qp);
}

/* ---- MotionReference::init(recon, orig, wp) ---- */
int __real__ZN4x26515MotionReference4initEPNS_10TComPicYuvES2_PNS_14WpScalingParamE(MotionReference *self, TComPicYuv *recon,
                                                                                   TComPicYuv *orig, wpScalingParam *w);
int __wrap__ZN4x26515MotionReference4initEPNS_10TComPicYuvES2_PNS_14WpScalingParamE(MotionReference *self, TComPicYuv *recon,
                                                                                   TComPicYuv *orig, wpScalingParam *w)
{
    const int rc = __real__ZN4x26515MotionReference4initEPNS_10TComPicYuvES2_PNS_14WpScalingParamE(self, recon, orig, w);
    if (mode() == MODE_CPU) return rc;
    Locked guard;
    const double t0 = now();
    std::map<const void*, RefPlanes>::iterator it = g_refs.find(self->fpelPlane);
    if (it != g_refs.end())
    {
        x265amd_plane_free(g_ctx, &it->second.recon);
        x265amd_plane_free(g_ctx, &it->second.ds);
        g_refs.erase(it);
    }
    if (!w && rc == 0)                            /* a weighted reference is searched by the reference's own code */
    {
        RefPlanes r;
        r.recon = upload(recon);
        x265amd_plane o = upload(orig);
        if (x265amd_rk_ds_plane(g_ctx, &o, &r.ds) != X265AMD_OK) die("rk_ds_plane");
        x265amd_plane_free(g_ctx, &o);
        g_refs[self->fpelPlane] = r;
    }
    g_st.up_s += now() - t0;
    return rc;
}

/* ---- the RK overload ---- */
int __real__ZN4x26514MotionEstimate14motionEstimateEPNS_15ReferencePlanesERKNS_2MVES5_RS3_PS3_ibbjjliiii(
    MotionEstimate *self, ReferencePlanes *ref, const MV *mvmin, const MV *mvmax, MV *outQMv, MV *mvNeighBor, int nMvNeighBor,
    bool isHaveMvd, bool isSavePmv, uint32_t offsIdx, uint32_t depth, intptr_t blockOffset_ds, int stride_ds,
    int nValidCtuWidth, int nValidCtuHeight, int nRefPicIdx);
int __wrap__ZN4x26514MotionEstimate14motionEstimateEPNS_15ReferencePlanesERKNS_2MVES5_RS3_PS3_ibbjjliiii(
    MotionEstimate *self, ReferencePlanes *ref, const MV *mvmin, const MV *mvmax, MV *outQMv, MV *mvNeighBor, int nMvNeighBor,
    bool isHaveMvd, bool isSavePmv, uint32_t offsIdx, uint32_t depth, intptr_t blockOffset_ds, int stride_ds,
    int nValidCtuWidth, int nValidCtuHeight, int nRefPicIdx)
{
#define REAL_RK() __real__ZN4x26514MotionEstimate14motionEstimateEPNS_15ReferencePlanesERKNS_2MVES5_RS3_PS3_ibbjjliiii( \
        self, ref, mvmin, mvmax, outQMv, mvNeighBor, nMvNeighBor, isHaveMvd, isSavePmv, offsIdx, depth, blockOffset_ds, stride_ds, \
        nValidCtuWidth, nValidCtuHeight, nRefPicIdx)
    const double t0 = now();
    if (mode() == MODE_CPU)
    {
        const int c = REAL_RK();
        Locked guard;
        g_st.rk_calls++;
        g_st.rk_s += now() - t0;
        return c;
    }
    Locked guard;
    g_st.rk_calls++;
    const int x = (int)(self->blockOffset % self->fencLumaStride), y = (int)(self->blockOffset / self->fencLumaStride);
    std::map<const void*, RefPlanes>::iterator rp = g_refs.find(ref->fpelPlane);
    const bool covered = rp != g_refs.end() && g_curOrg && self->fencplane == g_curOrg->getLumaAddr() &&
                         g_qpOf.count(static_cast<BitCost*>(self)) && nRefPicIdx >= 0 && nRefPicIdx < nMaxRefPic &&
                         (isSavePmv || (g_ctu[nRefPicIdx].valid && g_ctu[nRefPicIdx].x == (x & ~63) && g_ctu[nRefPicIdx].y == (y & ~63)));
    if (!covered)
    {
        g_st.fallback++;
        if (nRefPicIdx >= 0 && nRefPicIdx < nMaxRefPic) g_ctu[nRefPicIdx].valid = false;
        const int c = REAL_RK();
        g_st.rk_s += now() - t0;
        return c;
    }
    CtuAnswers &a = g_ctu[nRefPicIdx];
    MV *pmv = g_pmv[nRefPicIdx];
    /* motion.cpp:1305-1312: the window centres are refreshed by every call */
    for (int i = 0; i < nMvNeighBor && i < nNeightMv; i++) pmv[i + 1] = mvNeighBor[i];
    if (nMvNeighBor < nNeightMv) pmv[2] = pmv[1];
    if (isSavePmv)
    {
        if (!g_fencValid)
        {
            const double t1 = now();
            g_fenc = upload(g_curOrg);
            g_fencValid = true;
            g_st.up_s += now() - t1;
        }
        x265amd_rk_ctu job;
        memset(&job, 0, sizeof(job));
        job.ctu_x = (int16_t)(x & ~63); job.ctu_y = (int16_t)(y & ~63);
        job.valid_w = (int16_t)nValidCtuWidth; job.valid_h = (int16_t)nValidCtuHeight;
        job.mvmin[0] = mvmin->x; job.mvmin[1] = mvmin->y; job.mvmax[0] = mvmax->x; job.mvmax[1] = mvmax->y;
        for (int i = 0; i < 2; i++) { job.pmv[i][0] = pmv[i + 1].x; job.pmv[i][1] = pmv[i + 1].y; }
        for (int i = 0; i < 3; i++) { job.mvp[i][0] = mvNeighBor[7 + i].x; job.mvp[i][1] = mvNeighBor[7 + i].y; }
        job.have_mvd = isHaveMvd;
        if (x265amd_rk_ctu_batch(g_ctx, &g_fenc, &rp->second.recon, &rp->second.ds, g_qpOf[static_cast<BitCost*>(self)], &job, 1,
                                 &a.res) != X265AMD_OK)
            die("rk_ctu_batch");
        a.valid = true; a.x = job.ctu_x; a.y = job.ctu_y;
        g_st.rk_ctus++;
    }
    Mv exp_left, exp_Left;
    if (g_mode == MODE_CHECK) { exp_left = g_leftPmv[nRefPicIdx]; exp_Left = g_leftPMV[nRefPicIdx]; }
    int cost = a.res.fme[offsIdx].cost;
    MV out(a.res.fme[offsIdx].mv[0], a.res.fme[offsIdx].mv[1]);
    if (g_mode == MODE_CHECK)
    {
        MV neigh[10];
        for (int i = 0; i < 10; i++) neigh[i] = mvNeighBor[i];
        MV rout;
        MV *keep = outQMv;
        outQMv = &rout;
        const int rcost = REAL_RK();
        outQMv = keep;
        const Mv &c = g_leftPMV[nRefPicIdx], &r = g_RimeMv[nRefPicIdx][offsIdx], &f = g_FmeMv[nRefPicIdx][offsIdx];
        const x265amd_rk_mv &gc = a.res.cime, &gr = a.res.rime[offsIdx], &gf = a.res.fme[offsIdx];
        bool ok = rcost == cost && rout.x == out.x && rout.y == out.y && r.x == gr.mv[0] && r.y == gr.mv[1] && r.nSadCost == gr.cost &&
                  f.x == gf.mv[0] && f.y == gf.mv[1] && f.nSadCost == gf.cost;
        if (isSavePmv) ok = ok && c.x == gc.mv[0] && c.y == gc.mv[1] && c.nSadCost == gc.cost;
        if (!ok)
        {
            fprintf(stderr, "enc_serve: RK mismatch at (%d,%d) cu %u ref %d: reference mv (%d,%d) cost %d rime (%d,%d,%d) cime (%d,%d,%d); "
                    "engine mv (%d,%d) cost %d rime (%d,%d,%d) cime (%d,%d,%d)\n", x, y, offsIdx, nRefPicIdx, rout.x, rout.y, rcost,
                    r.x, r.y, r.nSadCost, c.x, c.y, c.nSadCost, out.x, out.y, cost, gr.mv[0], gr.mv[1], gr.cost, gc.mv[0], gc.mv[1], gc.cost);
            abort();
        }
        /* put the globals back so that the code below is what sets them */
        g_leftPmv[nRefPicIdx] = exp_left; g_leftPMV[nRefPicIdx] = exp_Left;
    }
    /* what the overload leaves behind (motion.cpp:1300-1302, :1328-1334, :1361, :1403) */
    if (isSavePmv)
    {
        g_leftPmv[nRefPicIdx] = g_leftPMV[nRefPicIdx];
        pmv[0] = MV(a.res.cime.mv[0], a.res.cime.mv[1]);
        g_leftPMV[nRefPicIdx].x = pmv[0].x; g_leftPMV[nRefPicIdx].y = pmv[0].y; g_leftPMV[nRefPicIdx].nSadCost = a.res.cime.cost;
    }
    const short size = (short)(64 >> depth), offsX = OffsFromCtu64[offsIdx][0], offsY = OffsFromCtu64[offsIdx][1];
    for (int i = 0; i <= nNeightMv; i++)
    {
        g_Mvmin[nRefPicIdx][i].x = pmv[i].x - offsX - (short)nRimeWidth;
        g_Mvmin[nRefPicIdx][i].y = pmv[i].y - offsY - (short)nRimeHeight;
        g_Mvmax[nRefPicIdx][i].x = pmv[i].x + (short)g_maxCUWidth + (short)nRimeWidth - offsX - size;
        g_Mvmax[nRefPicIdx][i].y = pmv[i].y + (short)g_maxCUWidth + (short)nRimeHeight - offsY - size;
    }
    Mv &r = g_RimeMv[nRefPicIdx][offsIdx], &f = g_FmeMv[nRefPicIdx][offsIdx];
    r.x = a.res.rime[offsIdx].mv[0]; r.y = a.res.rime[offsIdx].mv[1]; r.nSadCost = a.res.rime[offsIdx].cost;
    f.x = out.x; f.y = out.y; f.nSadCost = cost;
    self->setMVP(mvNeighBor[9]);                  /* the last predictor the overload selected (:1393) */
    *outQMv = out;
    g_st.rk_s += now() - t0;
    return cost;
#undef REAL_RK
}

/* ---- Lowres::init ---- */
void __real__ZN4x2656Lowres4initEPNS_10TComPicYuvEiii(Lowres *self, TComPicYuv *orig, int poc, int type, int bframes);
void __wrap__ZN4x2656Lowres4initEPNS_10TComPicYuvEiii(Lowres *self, TComPicYuv *orig, int poc, int type, int bframes)
{
    __real__ZN4x2656Lowres4initEPNS_10TComPicYuvEiii(self, orig, poc, type, bframes);
    if (mode() == MODE_CPU) return;
    Locked guard;
    const double t0 = now();
    const void *key = self->lowresPlane[0];
    dropFieldsOf(key);
    std::map<const void*, LowPlanes>::iterator it = g_low.find(key);
    if (it != g_low.end())
    {
        for (int i = 0; i < 4; i++) x265amd_plane_free(g_ctx, &it->second.p[i]);
        g_low.erase(it);
    }
    /* x265amd_lowres_init makes planes of exactly half the picture; the reference rounds the lowres size up to whole
     * 8x8 CUs (lowres.cpp:38-44), which is the same thing when the picture is a multiple of 16 */
    if (orig->getWidth() % 16 == 0 && orig->getHeight() % 16 == 0)
    {
        LowPlanes lp;
        x265amd_plane o = upload(orig);
        if (x265amd_lowres_init(g_ctx, &o, lp.p) != X265AMD_OK) die("lowres_init");
        x265amd_plane_free(g_ctx, &o);
        lp.wcu = self->width / 8; lp.hcu = self->lines / 8;
        g_low[key] = lp;
        g_st.lowres_up++;
    }
    g_st.up_s += now() - t0;
}

/* ---- the classic search: one lowres CU of the lookahead ---- */
int __real__ZN4x26514MotionEstimate14motionEstimateEPNS_15ReferencePlanesERKNS_2MVES5_S5_iPS4_iRS3_(
    MotionEstimate *self, ReferencePlanes *ref, const MV *mvmin, const MV *mvmax, const MV *qmvp, int numCandidates, const MV *mvc,
    int merange, MV *outQMv);
int __wrap__ZN4x26514MotionEstimate14motionEstimateEPNS_15ReferencePlanesERKNS_2MVES5_S5_iPS4_iRS3_(
    MotionEstimate *self, ReferencePlanes *ref, const MV *mvmin, const MV *mvmax, const MV *qmvp, int numCandidates, const MV *mvc,
    int merange, MV *outQMv)
{
#define REAL_CL() __real__ZN4x26514MotionEstimate14motionEstimateEPNS_15ReferencePlanesERKNS_2MVES5_S5_iPS4_iRS3_( \
        self, ref, mvmin, mvmax, qmvp, numCandidates, mvc, merange, outQMv)
    const double t0 = now();
    if (mode() == MODE_CPU)
    {
        const int c = REAL_CL();
        Locked guard;
        g_st.la_calls++;
        g_st.la_s += now() - t0;
        return c;
    }
    Locked guard;
    g_st.la_calls++;
    std::map<const void*, LowPlanes>::iterator pf = g_low.end(), pr = g_low.end();
    if (ref->isLowres && self->blockwidth == 8 && self->blockheight == 8)
    {
        pf = g_low.find(self->fencplane);
        pr = g_low.find(ref->lowresPlane[0]);
    }
    if (pf == g_low.end() || pr == g_low.end() || !g_qpOf.count(static_cast<BitCost*>(self)))
    {
        g_st.fallback++;
        const int c = REAL_CL();
        g_st.la_s += now() - t0;
        return c;
    }
    const int wcu = pf->second.wcu, hcu = pf->second.hcu, ncu = wcu * hcu;
    const int cu = (int)(self->blockOffset / self->fencLumaStride) / 8 * wcu + (int)(self->blockOffset % self->fencLumaStride) / 8;
    const FieldKey key(self->fencplane, ref->lowresPlane[0]);
    std::map<FieldKey, Field>::iterator f = g_fields.find(key);
    if (f == g_fields.end())
    {
        /* the first CU of a field: search the whole field (slicetype.cpp:1342-1387 visits every CU of it, each
         * predicted from the final MVs of its right / lower neighbours, which is what the engine's field driver does) */
        Field nf;
        nf.mv.resize(2 * ncu); nf.cost.resize(ncu); nf.left = ncu;
        x265amd_field fd;
        fd.fenc = pf->second.p[0];
        for (int i = 0; i < 4; i++) fd.ref[i] = pr->second.p[i];
        if (x265amd_lookahead_fields(g_ctx, &fd, 1, wcu, hcu, g_qpOf[static_cast<BitCost*>(self)], self->searchMethod, self->subpelRefine,
                                     merange, &nf.mv[0], &nf.cost[0], NULL) != X265AMD_OK)
            die("lookahead_fields");
        f = g_fields.insert(std::make_pair(key, nf)).first;
        g_st.la_fields++;
    }
    const MV out(f->second.mv[2 * cu], f->second.mv[2 * cu + 1]);
    const int cost = f->second.cost[cu];
    if (g_mode == MODE_CHECK)
    {
        MV rout;
        MV *keep = outQMv;
        outQMv = &rout;
        const int rcost = REAL_CL();
        outQMv = keep;
        if (rcost != cost || rout.x != out.x || rout.y != out.y)
        {
            fprintf(stderr, "enc_serve: lookahead mismatch at CU %d: reference mv (%d,%d) cost %d, engine mv (%d,%d) cost %d\n",
                    cu, rout.x, rout.y, rcost, out.x, out.y, cost);
            abort();
        }
    }
    if (--f->second.left == 0) g_fields.erase(f);
    self->setMVP(*qmvp);                          /* motion.cpp:562 */
    *outQMv = out;
    g_st.la_s += now() - t0;
    return cost;
#undef REAL_CL
}

}
