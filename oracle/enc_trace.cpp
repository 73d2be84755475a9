/* TEST INFRASTRUCTURE ONLY -- not part of the shipped engine.
 *
 * Call-stream recorder for the REFERENCE encoder. oracle/Makefile compiles the reference's own sources where they
 * lie and links this file in with the linker's --wrap option, so that the reference's calls of
 *
 *   MotionEstimate::motionEstimate, RK overload    caller Lib/TLibEncoder/TEncSearch.cpp:3821
 *   MotionEstimate::motionEstimate, classic        caller encoder/slicetype.cpp:1468 (lookahead)
 *   BitCost::setQP                                 caller Lib/TLibEncoder/TEncSearch.cpp:304-308, slicetype.cpp
 *   MotionReference::init(recon, orig, wp)         caller encoder/frameencoder.cpp:625-627
 *   Lowres::init                                   caller encoder/slicetype.cpp:93-110
 *   DPB::prepareEncode                             caller encoder/encoder.cpp:344 (the frame handed to the frame encoder)
 *
 * arrive at the __wrap_ functions below, which call the real member functions (__real_) and append what went in
 * and what came out to the file named by the environment variable X265AMD_TRACE. Nothing of the reference is edited
 * or copied; its headers are only #included at build time. tests/golden/make_callstream.py runs the traced encoder
 * and turns the file into the committed fixture.
 *
 * Record layout (little endian): 4-byte tag, then the fields appended by each wrapper, in order.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <pthread.h>
#include <map>
#include <set>
#include <string>

/* the recorder reads a few protected members (source plane and block offset of a MotionEstimate) */
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

using namespace x265;

namespace {

FILE *g_out = NULL;
bool g_tried = false;
int g_curPoc = -1;
int g_nextRefId = 0;
std::set<int> g_rkPocs;                        /* frames whose RK calls and reference planes are recorded; empty = all */
std::map<const void*, int> g_qpOf;             /* BitCost object -> last setQP */
std::map<const void*, int> g_refIdOf;          /* fpelPlane of a MotionReference -> id of its latest MREF record */

FILE *out()
{
    if (!g_tried)
    {
        g_tried = true;
        const char *path = getenv("X265AMD_TRACE");
        if (path && *path) g_out = fopen(path, "wb");
        const char *pocs = getenv("X265AMD_TRACE_RK_POCS");     /* e.g. "1,2,3" */
        for (const char *p = pocs; p && *p;)
        {
            g_rkPocs.insert(atoi(p));
            p = strchr(p, ',');
            if (p) p++;
        }
    }
    return g_out;
}

/* The frame encoder runs in a thread of its own next to the lookahead (encoder.cpp:351-352), so the wrappers are
 * entered from two threads: every wrapper holds g_lock while it touches the maps and builds its record in g_rec,
 * and the record goes out in one piece. The real functions run outside the lock. */
pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
std::string g_rec;
struct Locked
{
    Locked() { pthread_mutex_lock(&g_lock); g_rec.clear(); }
    ~Locked() { if (g_out && !g_rec.empty()) fwrite(g_rec.data(), 1, g_rec.size(), g_out); pthread_mutex_unlock(&g_lock); }
};

void tag(const char *t) { g_rec.append(t, 4); }
void i32(int32_t v) { g_rec.append((const char*)&v, 4); }
void u64(uint64_t v) { g_rec.append((const char*)&v, 8); }
void mv(const MV &m) { int16_t v[2] = { m.x, m.y }; g_rec.append((const char*)v, 4); }
bool rkWanted() { return g_rkPocs.empty() || g_rkPocs.count(g_curPoc); }

void plane(const pixel *origin, intptr_t stride, int w, int h)
{
    for (int y = 0; y < h; y++) g_rec.append((const char*)(origin + y * stride), sizeof(pixel) * w);
}

}

extern "C" {

/* ---- DPB::prepareEncode: the next frame to be compressed (one frame is in flight with --threads 1) ---- */
void __real__ZN4x2653DPB13prepareEncodeEPNS_7TComPicE(DPB *self, TComPic *pic);
void __wrap__ZN4x2653DPB13prepareEncodeEPNS_7TComPicE(DPB *self, TComPic *pic)
{
    __real__ZN4x2653DPB13prepareEncodeEPNS_7TComPicE(self, pic);
    Locked guard;
    if (out())
    {
        g_curPoc = pic->getPOC();
        tag("FRAM");
        i32(g_curPoc);
        i32((int)pic->getSlice()->getSliceType());
        i32(pic->getPicYuvOrg()->getWidth());
        i32(pic->getPicYuvOrg()->getHeight());
    }
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

/* ---- MotionReference::init(recon, orig, wp): the planes a frame's searches will read ---- */
int __real__ZN4x26515MotionReference4initEPNS_10TComPicYuvES2_PNS_14WpScalingParamE(MotionReference *self, TComPicYuv *recon,
                                                                                   TComPicYuv *orig, wpScalingParam *w);
int __wrap__ZN4x26515MotionReference4initEPNS_10TComPicYuvES2_PNS_14WpScalingParamE(MotionReference *self, TComPicYuv *recon,
                                                                                   TComPicYuv *orig, wpScalingParam *w)
{
    const int rc = __real__ZN4x26515MotionReference4initEPNS_10TComPicYuvES2_PNS_14WpScalingParamE(self, recon, orig, w);
    Locked guard;
    if (out() && rkWanted())
    {
        const int id = g_nextRefId++;
        g_refIdOf[self->fpelPlane] = id;
        tag("MREF");
        i32(g_curPoc); i32(id); i32(recon->getWidth()); i32(recon->getHeight());
        i32(recon->getLumaMarginX()); i32(recon->getLumaMarginY()); i32(w ? 1 : 0);
        plane(orig->getLumaAddr(), orig->getStride(), orig->getWidth(), orig->getHeight());
        plane(recon->getLumaAddr(), recon->getStride(), recon->getWidth(), recon->getHeight());
    }
    return rc;
}

/* ---- the RK overload: CIME + RIME + FME of one CU ---- */
int __real__ZN4x26514MotionEstimate14motionEstimateEPNS_15ReferencePlanesERKNS_2MVES5_RS3_PS3_ibbjjliiii(
    MotionEstimate *self, ReferencePlanes *ref, const MV *mvmin, const MV *mvmax, MV *outQMv, MV *mvNeighBor, int nMvNeighBor,
    bool isHaveMvd, bool isSavePmv, uint32_t offsIdx, uint32_t depth, intptr_t blockOffset_ds, int stride_ds,
    int nValidCtuWidth, int nValidCtuHeight, int nRefPicIdx);
int __wrap__ZN4x26514MotionEstimate14motionEstimateEPNS_15ReferencePlanesERKNS_2MVES5_RS3_PS3_ibbjjliiii(
    MotionEstimate *self, ReferencePlanes *ref, const MV *mvmin, const MV *mvmax, MV *outQMv, MV *mvNeighBor, int nMvNeighBor,
    bool isHaveMvd, bool isSavePmv, uint32_t offsIdx, uint32_t depth, intptr_t blockOffset_ds, int stride_ds,
    int nValidCtuWidth, int nValidCtuHeight, int nRefPicIdx)
{
    MV neigh[10];
    for (int i = 0; i < 10; i++) neigh[i] = mvNeighBor[i];
    const int cost = __real__ZN4x26514MotionEstimate14motionEstimateEPNS_15ReferencePlanesERKNS_2MVES5_RS3_PS3_ibbjjliiii(
        self, ref, mvmin, mvmax, outQMv, mvNeighBor, nMvNeighBor, isHaveMvd, isSavePmv, offsIdx, depth, blockOffset_ds, stride_ds,
        nValidCtuWidth, nValidCtuHeight, nRefPicIdx);
    Locked guard;
    if (out() && rkWanted())
    {
        tag("RKME");
        i32(g_curPoc);
        i32(g_refIdOf.count(ref->fpelPlane) ? g_refIdOf[ref->fpelPlane] : -1);
        i32((int)(self->blockOffset % self->fencLumaStride)); i32((int)(self->blockOffset / self->fencLumaStride));
        i32(self->blockwidth); i32(self->blockheight);
        i32(g_qpOf.count(static_cast<BitCost*>(self)) ? g_qpOf[static_cast<BitCost*>(self)] : -1);
        mv(*mvmin); mv(*mvmax);
        for (int i = 0; i < 10; i++) mv(neigh[i]);
        i32(nMvNeighBor); i32(isHaveMvd); i32(isSavePmv); i32((int)offsIdx); i32((int)depth);
        i32((int)blockOffset_ds); i32(stride_ds); i32(nValidCtuWidth); i32(nValidCtuHeight); i32(nRefPicIdx);
        mv(*outQMv); i32(cost);
        const Mv &c = g_leftPMV[nRefPicIdx], &r = g_RimeMv[nRefPicIdx][offsIdx], &f = g_FmeMv[nRefPicIdx][offsIdx];
        i32(c.x); i32(c.y); i32(c.nSadCost);
        i32(r.x); i32(r.y); i32(r.nSadCost);
        i32(f.x); i32(f.y); i32(f.nSadCost);
    }
    return cost;
}

/* ---- Lowres::init: which lowres planes belong to which frame ---- */
void __real__ZN4x2656Lowres4initEPNS_10TComPicYuvEiii(Lowres *self, TComPicYuv *orig, int poc, int type, int bframes);
void __wrap__ZN4x2656Lowres4initEPNS_10TComPicYuvEiii(Lowres *self, TComPicYuv *orig, int poc, int type, int bframes)
{
    __real__ZN4x2656Lowres4initEPNS_10TComPicYuvEiii(self, orig, poc, type, bframes);
    Locked guard;
    if (out())
    {
        tag("LOWR");
        i32(poc); u64((uint64_t)(uintptr_t)self->lowresPlane[0]);
        i32(self->width); i32(self->lines); i32(self->lumaStride);
        i32(orig->getLumaMarginX()); i32(orig->getLumaMarginY());
    }
}

/* ---- the classic search: one lowres CU of the lookahead (or a PU of an rd < 5 encode) ---- */
int __real__ZN4x26514MotionEstimate14motionEstimateEPNS_15ReferencePlanesERKNS_2MVES5_S5_iPS4_iRS3_(
    MotionEstimate *self, ReferencePlanes *ref, const MV *mvmin, const MV *mvmax, const MV *qmvp, int numCandidates, const MV *mvc,
    int merange, MV *outQMv);
int __wrap__ZN4x26514MotionEstimate14motionEstimateEPNS_15ReferencePlanesERKNS_2MVES5_S5_iPS4_iRS3_(
    MotionEstimate *self, ReferencePlanes *ref, const MV *mvmin, const MV *mvmax, const MV *qmvp, int numCandidates, const MV *mvc,
    int merange, MV *outQMv)
{
    const int cost = __real__ZN4x26514MotionEstimate14motionEstimateEPNS_15ReferencePlanesERKNS_2MVES5_S5_iPS4_iRS3_(
        self, ref, mvmin, mvmax, qmvp, numCandidates, mvc, merange, outQMv);
    Locked guard;
    if (out())
    {
        tag("CLME");
        u64((uint64_t)(uintptr_t)self->fencplane);
        u64((uint64_t)(uintptr_t)(ref->isLowres ? ref->lowresPlane[0] : ref->fpelPlane));
        i32(ref->isLowres ? 1 : 0);
        i32((int)(self->blockOffset % self->fencLumaStride)); i32((int)(self->blockOffset / self->fencLumaStride));
        i32(self->blockwidth); i32(self->blockheight);
        i32(self->searchMethod); i32(self->subpelRefine);
        i32(g_qpOf.count(static_cast<BitCost*>(self)) ? g_qpOf[static_cast<BitCost*>(self)] : -1);
        mv(*mvmin); mv(*mvmax); mv(*qmvp);
        i32(numCandidates);
        for (int i = 0; i < 4; i++) mv(i < numCandidates ? mvc[i] : MV(0, 0));
        i32(merange);
        mv(*outQMv); i32(cost);
    }
    return cost;
}

}
