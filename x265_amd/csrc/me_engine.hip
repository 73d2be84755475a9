/* Host side of the C ABI (include/x265_amd_me.h) and the kernels it launches.
 * Built for gfx950 only: hipcc --offload-arch=gfx950. */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "me_search.h"

using namespace x265amd;

/* ------------------------------------------------------------------ kernels */

/* XCD-aware job order: consecutive workgroup ids land on different XCDs (each with its
 * own L2). Jobs are dealt to the XCDs in runs of 32 neighbours, so PUs that read
 * overlapping reference pixels share an L2 while every XCD still sees the same mix of
 * PU sizes (a batch usually lists the large, expensive PUs first). */
__device__ __forceinline__ int xcd_remap(int bid, int n)
{
    const int C = 32;
    if (bid >= (n & ~(8 * C - 1))) return bid;
    const int x = bid & 7, i = bid >> 3;
    return ((i / C) * 8 + x) * C + (i % C);
}

/* sad_calls of a result record while a launch is under way: the fixed-shape STAR kernels (star_small_kernel,
 * star_big_kernel) left this PU to the one-wave search from the start (SMALL_REDO: not a size of theirs), or measured
 * its start-up candidates and first three pattern rounds, found an improving point and left the rest of the integer
 * search to star_resume_kernel (SMALL_RESUME: the state record of the PU says where it stands) */
#define SMALL_REDO 0xFFFFFFFFu
#define SMALL_RESUME 0xFFFFFFFEu

/* one (source picture, reference picture) pair of a launch that searches several pairs at once */
struct PairDev
{
    PlaneDev fenc, ref;
};

struct SearchArgs
{
    PlaneDev fenc;
    PlaneDev ref[4];
    const PairDev *pairs;   /* null, or the pair table of the launch: job.reserved picks the entry (full-resolution searches) */
    int npairs;
    int nref, lowres, depth, method, subme;
    const uint16_t *ctab;
    const x265amd_job *jobs;
    x265amd_result *results;
    int njobs;
    int max_dim;   /* largest PU width / height the launch reserved LDS for */
    int *redo;         /* null, or the list redo_list_kernel made of the PUs whose result record carries SMALL_REDO (left over
                          by star_small_kernel): redo[0] = their number, redo[REDO_LIST..] = their job indices. The launch
                          then searches only these, each wave taking every gridDim.x-th entry */
    int plane_lds; /* LDS bytes a wave of search_kernel has for interpolation planes (0: plane_budget of max_dim) */
    int staged;    /* 8, 16, 32 or 64: square PUs of this size stop before the sub-pel refinement, which refine_small_kernel
                      (8, 16) or refine_big_kernel (32, 64) does; 0: none */
    int staged2;   /* 32 in a launch staged for 64: its 32x32 PUs stop there too and are refined with their parents where
                      the list has them behind one (refine_big_kernel<64>), by refine_big_kernel<32> otherwise; else 0 */
    unsigned char *done;   /* null, or per job: refined already (set by refine_big_kernel<64> for the children it took along) */
    int *state;            /* per job 8 ints: where a STAR search stands that a fixed-shape kernel began (StarFold::store) */
    /* null, or the list of 8x8 PUs launched with this list of 16x16 PUs: PU 4 k + q of it is taken along by PU k of this
     * list through the sub-pel rounds where it is that PU's quadrant q and starts from the same MV */
    const x265amd_job *cjobs;
    x265amd_result *cresults;
    int ncjobs;
    unsigned char *cmark, *cdone;
    unsigned char *mark;   /* null, or per job: 1 = a STAR kernel handed this PU on (it is in the redo list), 0 = it did not */
    int list;              /* refinement kernels: 1 = refine the PUs of the redo list instead of those of the job list;
                              refine_big_kernel: 2 = `redo` lists the PUs the job-list launch would refine (square_list_kernel) */
    /* family launches: the list of 8x8 PUs carries the parents' list (pjobs ...), whose PU k a wave of
     * star_small_kernel<8> searches along with its children 4 k .. 4 k + 3 where it can; handled[k] says so to both lists'
     * launches (in the children's arguments: written; in the parents': read) */
    const x265amd_job *pjobs;
    x265amd_result *presults;
    int npjobs;
    unsigned char *pmark;
    int *pstate;
    unsigned char *handled;
    int tiny;              /* 1: the launch has star_tiny_kernel for the small other partitions (tiny_class): the square
                              kernels do not hand those to the one-wave search */
    int rect;              /* 1: the launch has star_rect_kernel for the larger other partitions (rect_class), likewise */
    int stage_all;         /* 1: the one-wave search stops every PU before the sub-pel refinement; the PUs that are not squares of
                              the fixed-shape sizes are refined by refine_any_kernel */
};

/* the planes job `job` searches: the launch's own, or its entry of the launch's pair table. Device-resident jobs
 * are not validated, so an index beyond the table falls back to entry 0 instead of reading past it. */
__device__ __forceinline__ void job_planes(const SearchArgs &a, const x265amd_job *job, PlaneDev &fenc, PlaneDev ref[4])
{
    if (a.pairs)
    {
        int pi = uni((int)job->reserved);
        if (pi >= a.npairs) pi = 0;
        const PairDev *p = a.pairs + pi;
        fenc.origin = (const void*)uni64((unsigned long long)p->fenc.origin); fenc.stride = uni(p->fenc.stride);
        ref[0].origin = (const void*)uni64((unsigned long long)p->ref.origin); ref[0].stride = uni(p->ref.stride);
        ref[1] = ref[2] = ref[3] = ref[0];
    }
    else
    {
        fenc = a.fenc;
        for (int i = 0; i < 4; i++) ref[i] = a.ref[i];
    }
}

/* The list of the PUs star_small_kernel left to the one-wave search, made by one small kernel so that the search
 * launch behind it costs what the marked PUs cost and not a wave per PU of the launch. `redo` (ints): [0] the number
 * of marked PUs (star_small_kernel zeroes it), [REDO_LIST ..) their job indices, a block's entries together and in job
 * order, the blocks in the order their reservations arrive (the searches do not depend on each other). */
#define REDO_CHUNK 1024                /* jobs per block (one wave: a block of several waves waits long for room next to
                                          launches of one-wave blocks): one atomic reservation per block */
#define REDO_LIST 64                   /* ints in front of the list */
#define REDO_GRID 8192                 /* waves of the search launch that walks the list */

__global__ __launch_bounds__(64) void redo_list_kernel(const x265amd_result *results, int njobs, int *redo)
{
    const int base = blockIdx.x * REDO_CHUNK, lane = threadIdx.x;
    unsigned long long bal[REDO_CHUNK / 64];
    int n = 0;
#pragma unroll
    for (int k = 0; k < REDO_CHUNK / 64; k++)
    {
        const int j = base + k * 64 + lane;
        bal[k] = __ballot(j < njobs && results[j].sad_calls >= SMALL_RESUME);
        n += __popcll(bal[k]);
    }
    if (!n) return;
    int first = 0;
    if (lane == 0) first = atomicAdd(redo, n);
    first = __builtin_amdgcn_readfirstlane(first);
#pragma unroll
    for (int k = 0; k < REDO_CHUNK / 64; k++)
    {
        if ((bal[k] >> lane) & 1) redo[REDO_LIST + first + __popcll(bal[k] & ((1ull << lane) - 1))] = base + k * 64 + lane;
        first += __popcll(bal[k]);
    }
}

/* The square PUs of the launch's fixed-shape sizes (w | h << 8 == code1 or code2) that no STAR kernel handed on, as a list
 * of the redo list's form: for the refinement of a list that may hold few of them, where a workgroup of several waves per
 * job that leaves at once costs more than the PUs themselves (a 4K picture's 172 020 square PUs of all sizes in one launch:
 * refine_big_kernel<64> 0.26 ms for 1 980 PUs of its own). */
__global__ __launch_bounds__(64) void square_list_kernel(const x265amd_job *jobs, const unsigned char *mark, int njobs, int code1, int code2, int *list)
{
    const int base = blockIdx.x * REDO_CHUNK, lane = threadIdx.x;
    unsigned long long bal[REDO_CHUNK / 64];
    int n = 0;
#pragma unroll
    for (int k = 0; k < REDO_CHUNK / 64; k++)
    {
        const int j = base + k * 64 + lane;
        bool take = false;
        if (j < njobs)
        {
            const int code = ((const int*)(jobs + j))[1] & 0xffff;
            take = (code == code1 || code == code2) && !mark[j];
        }
        bal[k] = __ballot(take);
        n += __popcll(bal[k]);
    }
    if (!n) return;
    int first = 0;
    if (lane == 0) first = atomicAdd(list, n);
    first = __builtin_amdgcn_readfirstlane(first);
#pragma unroll
    for (int k = 0; k < REDO_CHUNK / 64; k++)
    {
        if ((bal[k] >> lane) & 1) list[REDO_LIST + first + __popcll(bal[k] & ((1ull << lane) - 1))] = base + k * 64 + lane;
        first += __popcll(bal[k]);
    }
}

template<typename P, bool LOWRES>
__global__ __launch_bounds__(64) void search_kernel(SearchArgs a)
{
    /* one PU per wave at a time: every gridDim.x-th job of the launch, or of the redo list (there only the PUs left
     * to this kernel from the start; star_resume_kernel takes the others). The waves of this kernel call functions
     * and so carry a stack, which makes them slow to launch (tens of nanoseconds each): a grid of a few thousand
     * waves that walk the jobs costs less than a wave per job. */
    const int total = a.redo ? uni(a.redo[0]) : a.njobs;
    for (int k = blockIdx.x; k < total; k += (int)gridDim.x)
    {
        const int j = a.redo ? uni(a.redo[REDO_LIST + k]) : xcd_remap(k, a.njobs);
        if (j >= a.njobs) return;
        if (a.redo && uni((int)a.results[j].sad_calls) != (int)SMALL_REDO) continue;
        const x265amd_job *job = a.jobs + j;
        Searcher<P, LOWRES ? 8 : 64, LOWRES> S;
        const int w = job->w, h = job->h;
        if (w > a.max_dim || h > a.max_dim)
        {
            /* the launch did not reserve LDS for a PU this large */
            if (threadIdx.x == 0)
            {
                x265amd_result r = { { 0, 0 }, -1, 0, 0 };
                a.results[j] = r;
            }
            continue;
        }
        PlaneDev fenc, ref[4];
        job_planes(a, job, fenc, ref);
        S.bind(0, fenc, ref, a.nref, a.ctab, a.depth, job->pu_x, job->pu_y, w, h, a.max_dim,
               a.plane_lds ? a.plane_lds : plane_budget(a.max_dim, (int)sizeof(P)));
        if (a.staged && ((a.stage_all && (any_refines(w, h) || tiny_class(w, h))) || (w == h && (w == a.staged || w == a.staged2)))) S.search_staged(job, a.method, a.results + j);
        else S.search(job, a.method, a.subme, a.results + j);
    }
}

#include "me_esa.h"   /* esa_scan_kernel, esa_finish_kernel: use SearchArgs and xcd_remap */
#include "me_esa_ctu.h"   /* esa_ctu_kernel: the scan of all 85 square PUs of a CTU at once */
#include "me_small.h"   /* star_small_kernel, refine_small_kernel: 8x8 / 16x16 PUs in fixed-shape kernels, four / one per wave */
#include "me_lookahead_cost.h"

/* The rest of the STAR searches a fixed-shape kernel began and could not end (SMALL_RESUME): later pattern rounds,
 * the two-point and raster refinements, the re-centred patterns (motion.cpp:916-1017) -- full-pel SADs only, one PU
 * per wave at a time from the redo list. The evaluations are inlined (Searcher LEAN): no calls, no stack, so the
 * waves of a launch that finds nothing to do cost next to nothing. Leaves the PU's record as search_staged does. */
template<typename P>
__global__ __launch_bounds__(64) void star_resume_kernel(SearchArgs a)
{
    const int total = uni(a.redo[0]);
    for (int k = blockIdx.x; k < total; k += (int)gridDim.x)
    {
        const int j = uni(a.redo[REDO_LIST + k]);
        if (j >= a.njobs) return;
        if (uni((int)a.results[j].sad_calls) != (int)SMALL_RESUME) continue;
        const x265amd_job *job = a.jobs + j;
        Searcher<P, 64, false, false, true> S;
        PlaneDev fenc, ref[4];
        job_planes(a, job, fenc, ref);
        S.bind(0, fenc, ref, 1, a.ctab, a.depth, job->pu_x, job->pu_y, job->w, job->h, a.max_dim, 0);
        S.resume_pattern(job, a.state + 8 * (long)j, a.method, a.results + j);
    }
}

/* Lookahead: one workgroup per lowres field, LA_WAVES waves walking the reverse-raster
 * dependency wavefront. CU (x, y) needs the final MVs of (x+1, y) and of
 * (x-1 .. x+1, y+1) (slicetype.cpp:1450-1459), so all CUs with equal
 * (W-1-x) + 2 (H-1-y) are independent; one barrier separates the steps. */
#define LA_WAVES 16
struct FieldDev
{
    PlaneDev fenc;
    PlaneDev ref[4];
};

struct LookaheadArgs
{
    const FieldDev *fields;
    int wcu, hcu, depth, method, subme, merange;
    const uint16_t *ctab;
    int16_t *mvs;        /* nfields * ncu * 2 */
    int32_t *costs;      /* nfields * ncu */
    x265amd_result *stats; /* nfields * ncu or null */
    int nfields;
    int *ticket;         /* row hand-out counter of lookahead_rows_kernel, zero at launch */
#ifdef X265AMD_PHASE_PROFILE
    long long *phase;    /* measurement build: 16 counters per wave of lookahead_band_kernel, or null */
#endif
};

__device__ __forceinline__ int median3(int a, int b, int c)
{
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    const int m = lo > c ? lo : c;
    return m < hi ? m : hi;
}

template<typename P>
__global__ __launch_bounds__(LA_WAVES * 64) void lookahead_kernel(LookaheadArgs a)
{
    __shared__ x265amd_job jobs[LA_WAVES];
    __shared__ x265amd_result results[LA_WAVES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int W = a.wcu, H = a.hcu, ncu = W * H;
    const FieldDev &f = a.fields[blockIdx.x];
    int16_t *mvs = a.mvs + (size_t)blockIdx.x * ncu * 2;
    int32_t *costs = a.costs + (size_t)blockIdx.x * ncu;
    const int nsteps = (W - 1) + 2 * (H - 1) + 1;
    for (int s = 0; s < nsteps; s++)
    {
        /* CUs of this step: yy = H-1-y in [ylo, yhi], xx = s - 2 yy in [0, W-1] */
        int ylo = (s - (W - 1) + 1) >> 1;
        if (ylo < 0) ylo = 0;
        int yhi = s >> 1;
        if (yhi > H - 1) yhi = H - 1;
        for (int yy = ylo + wave; yy <= yhi; yy += LA_WAVES)
        {
            const int xx = s - 2 * yy;
            const int cux = W - 1 - xx, cuy = H - 1 - yy;
            const int cuXY = cux + cuy * W;
            x265amd_job *job = &jobs[wave];
            if (lane == 0)
            {
                /* slicetype.cpp:1427-1466 */
                job->pu_x = (int16_t)(cux * 8); job->pu_y = (int16_t)(cuy * 8);
                job->w = 8; job->h = 8;
                job->mvmin[0] = (int16_t)(uint16_t)(-cux * 8 - 8);
                job->mvmin[1] = (int16_t)(uint16_t)(-cuy * 8 - 8);
                job->mvmax[0] = (int16_t)(uint16_t)((W - cux - 1) * 8 + 8);
                job->mvmax[1] = (int16_t)(uint16_t)((H - cuy - 1) * 8 + 8);
                int numc = 0;
                int16_t (*mvc)[2] = job->mvc;
                for (int i = 0; i < 4; i++) { mvc[i][0] = 0; mvc[i][1] = 0; }
                const int16_t *m = mvs + 2 * cuXY;
#define MVC(idx) do { mvc[numc][0] = m[2 * (idx)]; mvc[numc][1] = m[2 * (idx) + 1]; numc++; } while (0)
                if (cux < W - 1) MVC(1);
                if (cuy < H - 1)
                {
                    MVC(W);
                    if (cux > 0) MVC(W - 1);
                    if (cux < W - 1) MVC(W + 1);
                }
#undef MVC
                if (numc <= 1) { job->qmvp[0] = mvc[0][0]; job->qmvp[1] = mvc[0][1]; }
                else
                {
                    job->qmvp[0] = (int16_t)median3(mvc[0][0], mvc[1][0], mvc[2][0]);
                    job->qmvp[1] = (int16_t)median3(mvc[0][1], mvc[1][1], mvc[2][1]);
                }
                job->numc = (uint8_t)numc;
                job->merange = a.merange;
            }
            wave_sync();
            Searcher<P, 8, true> S;
            S.bind(wave * wave_lds_bytes<P>(8), f.fenc, f.ref, 4, a.ctab, a.depth, cux * 8, cuy * 8, 8, 8);
            S.search(job, a.method, a.subme, &results[wave]);
            wave_sync();
            if (lane == 0)
            {
                mvs[2 * cuXY] = results[wave].mv[0];
                mvs[2 * cuXY + 1] = results[wave].mv[1];
                costs[cuXY] = results[wave].cost;
                if (a.stats) a.stats[(size_t)blockIdx.x * ncu + cuXY] = results[wave];
            }
        }
        __syncthreads();
    }
}

/* Lookahead, row-parallel flavour: one wave per CU row, rows of all fields handed out bottom-up
 * through a ticket counter, any number of workgroups. A row runs right to left like
 * CostEstimate::processRow (slicetype.cpp:1342-1387) and, like the reference's wavefront rows,
 * follows the row below it at a distance of two CUs. The MV array is the synchronisation: the
 * launch presets every MV to x = 0x7FFF (the reference's own "not searched yet" mark,
 * slicetype.cpp:1150), a CU's MV is published with one 32-bit device-scope store, and a wave
 * polls the three MVs it needs from the row below (device-scope loads) until none carries the
 * mark. Rows are taken in ticket order and a row only waits for a row with a smaller ticket,
 * which a resident wave already holds, so every wait ends. A field of W x H CUs takes about
 * W + 2H searches of latency instead of W * H / 16. */
#define LA_MV_UNSET 0x7FFF7FFF
template<typename P>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void lookahead_rows_kernel(LookaheadArgs a)
{
    __shared__ x265amd_job jobs[1];
    __shared__ x265amd_result results[1];
    const int lane = threadIdx.x & 63;
    const int W = a.wcu, H = a.hcu, ncu = W * H;
    const int nrows = a.nfields * H;
    for (;;)
    {
        int t = 0;
        if (lane == 0) t = atomicAdd(a.ticket, 1);
        t = uni(t);
        if (t >= nrows) break;
        const int fi = t % a.nfields, cuy = H - 1 - t / a.nfields;
        const FieldDev &f = a.fields[fi];
        int *mvw = (int*)(a.mvs + (size_t)fi * ncu * 2);     /* one 32-bit word per MV: x low, y high */
        int32_t *costs = a.costs + (size_t)fi * ncu;
        int right = 0;                                         /* MV of (cux + 1, cuy): this wave's previous result */
        for (int cux = W - 1; cux >= 0; cux--)
        {
            const int cuXY = cux + cuy * W;
            /* lanes 0..2: below, below-left, below-right */
            int nb = 0;
            const int dx = lane == 0 ? 0 : (lane == 1 ? -1 : 1);
            const bool want = lane < 3 && cuy < H - 1 && cux + dx >= 0 && cux + dx < W;
            if (want)
            {
                const int *src = mvw + cuXY + W + dx;
                nb = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while ((nb & 0xffff) == 0x7FFF)
                {
                    __builtin_amdgcn_s_sleep(8);
                    nb = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            const int below = __builtin_amdgcn_readlane(nb, 0), bleft = __builtin_amdgcn_readlane(nb, 1),
                      bright = __builtin_amdgcn_readlane(nb, 2);
            x265amd_job *job = &jobs[0];
            if (lane == 0)
            {
                /* slicetype.cpp:1427-1466 */
                job->pu_x = (int16_t)(cux * 8); job->pu_y = (int16_t)(cuy * 8);
                job->w = 8; job->h = 8;
                job->mvmin[0] = (int16_t)(uint16_t)(-cux * 8 - 8);
                job->mvmin[1] = (int16_t)(uint16_t)(-cuy * 8 - 8);
                job->mvmax[0] = (int16_t)(uint16_t)((W - cux - 1) * 8 + 8);
                job->mvmax[1] = (int16_t)(uint16_t)((H - cuy - 1) * 8 + 8);
                int numc = 0;
                int16_t (*mvc)[2] = job->mvc;
                for (int i = 0; i < 4; i++) { mvc[i][0] = 0; mvc[i][1] = 0; }
#define MVC(word) do { mvc[numc][0] = (int16_t)((word) & 0xffff); mvc[numc][1] = (int16_t)((word) >> 16); numc++; } while (0)
                if (cux < W - 1) MVC(right);
                if (cuy < H - 1)
                {
                    MVC(below);
                    if (cux > 0) MVC(bleft);
                    if (cux < W - 1) MVC(bright);
                }
#undef MVC
                if (numc <= 1) { job->qmvp[0] = mvc[0][0]; job->qmvp[1] = mvc[0][1]; }
                else
                {
                    job->qmvp[0] = (int16_t)median3(mvc[0][0], mvc[1][0], mvc[2][0]);
                    job->qmvp[1] = (int16_t)median3(mvc[0][1], mvc[1][1], mvc[2][1]);
                }
                job->numc = (uint8_t)numc;
                job->merange = a.merange;
            }
            wave_sync();
            Searcher<P, 8, true> S;
            S.bind(0, f.fenc, f.ref, 4, a.ctab, a.depth, cux * 8, cuy * 8, 8, 8);
            S.search(job, a.method, a.subme, &results[0]);
            wave_sync();
            right = uni(((int)(uint16_t)results[0].mv[0]) | ((int)(uint16_t)results[0].mv[1] << 16));
            if (lane == 0)
            {
                costs[cuXY] = results[0].cost;
                if (a.stats) a.stats[(size_t)fi * ncu + cuXY] = results[0];
                __hip_atomic_store(mvw + cuXY, right, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            wave_sync();
        }
    }
}

#include "me_lookahead_hex.h"   /* lookahead_band_kernel: hexagon search, four CU rows per wave */

/* frame_init_lowres_core (pixel.cpp:595-618): one thread per lowres pixel, four outputs */
template<typename P>
__global__ void lowres_core_kernel(const P *src, P *d0, P *dh, P *dv, P *dc, long ss, long ds, int w, int h)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    const P *s0 = src + (long)2 * y * ss + 2 * x, *s1 = s0 + ss, *s2 = s1 + ss;
    const int a0 = s0[0], a1 = s0[1], a2 = s0[2], b0 = s1[0], b1 = s1[1], b2 = s1[2], c0 = s2[0], c1 = s2[1], c2 = s2[2];
#define F4(p, q, r, s) ((((((p) + (q) + 1) >> 1) + (((r) + (s) + 1) >> 1) + 1) >> 1))
    d0[y * ds + x] = (P)F4(a0, b0, a1, b1);
    dh[y * ds + x] = (P)F4(a1, b1, a2, b2);
    dv[y * ds + x] = (P)F4(b0, c0, b1, c1);
    dc[y * ds + x] = (P)F4(b1, c1, b2, c2);
#undef F4
}

/* xExtendPicCompBorder (TComPicYuv.cpp:169-): replicate the picture edge into the
 * margins; one thread per padded-plane sample outside the picture */
template<typename P>
__global__ void extend_border_kernel(P *origin, long stride, int w, int h, int mx, int my)
{
    const int x = (int)(blockIdx.x * blockDim.x + threadIdx.x) - mx, y = (int)(blockIdx.y * blockDim.y + threadIdx.y) - my;
    if (x >= w + mx || y >= h + my) return;
    if (x >= 0 && x < w && y >= 0 && y < h) return;
    const int sx = x < 0 ? 0 : (x >= w ? w - 1 : x), sy = y < 0 ? 0 : (y >= h ? h - 1 : y);
    origin[y * stride + x] = origin[sy * stride + sx];
}


/* ---- Rockchip path kernels ---- */

/* downSampleAndSum (reference.cpp:143-160): int16 sum of each 4x4 block */
template<typename P>
__global__ void rk_downsample_kernel(const P *src, long ss, int16_t *dst, long ds, int w4, int h4)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w4 || y >= h4) return;
    int sum = 0;
#pragma unroll
    for (int m = 0; m < 4; m++)
    {
        int v[4];
        Px<P>::unpack(Px<P>::load(src + (long)(4 * y + m) * ss + 4 * x), v);
        sum += v[0] + v[1] + v[2] + v[3];
    }
    dst[y * ds + x] = (int16_t)sum;
}

struct RkArgs
{
    PlaneDev fenc, recon;
    const int16_t *ds;
    int ds_stride;
    int pic_w, pic_h, depth;
    const uint16_t *ctab;
    const x265amd_rk_ctu *ctus;
    x265amd_rk_result *results;
    int nctus;
};

/* CIME (motion.cpp:1272-1303): one workgroup per CTU. The 16x16 block of 4x4 sums of the CTU,
 * the downsampled reference window and the lambda * bits of every column / row (three
 * predictors each) live in LDS. A thread measures two horizontally adjacent positions at a time:
 * both read the same dwords of a window row (sample pairs), the odd one through v_alignbit, and
 * v_sad_u16 takes two samples per instruction. The raster-order strict-'<' winner is the minimum
 * of the 64-bit key (cost << 32 | visit index). */
#define CIME_THREADS 1024
#define CIME_MAX_NX 96
#define CIME_MAX_NY 80
template<typename P>
__global__ __launch_bounds__(CIME_THREADS) void rk_cime_kernel(RkArgs a)
{
    extern __shared__ int16_t cime_lds[];
    __shared__ unsigned long long red[CIME_THREADS / 64];
    __shared__ uint16_t costx[3][CIME_MAX_NX], costy[3][CIME_MAX_NY];
    const x265amd_rk_ctu &j = a.ctus[blockIdx.x];
    int16_t *ctu = cime_lds;            /* 256: the block as DownSample leaves it, pitch 16 */
    int16_t *blk = cime_lds + 256;      /* 256: the block as SAD reads it (pitch = valid width), re-pitched to 16 and zero padded */
    int16_t *win = cime_lds + 512;      /* (nx + 15, rounded up to even) x (ny + 15) */
    const int x0 = j.mvmin[0] / 4, x1 = j.mvmax[0] / 4, y0 = j.mvmin[1] / 4, y1 = j.mvmax[1] / 4;
    const int nx = x1 - x0 + 1, ny = y1 - y0 + 1;
    const int ww = (nx + 15 + 1) & ~1, wh = ny + 15;
    const int tid = threadIdx.x;
    const int cw = j.valid_w / 4, ch = j.valid_h / 4;
    if (tid < 256)
    {
        /* DownSample (motion.cpp:1232-1249) of the 64x64 block cached at the CTU origin */
        const P *src = (const P*)a.fenc.origin + (long)j.ctu_y * a.fenc.stride + j.ctu_x;
        const int bx = tid & 15, by = tid >> 4;
        int sum = 0;
#pragma unroll
        for (int m = 0; m < 4; m++)
        {
            int v[4];
            Px<P>::unpack(Px<P>::load(src + (long)(4 * by + m) * a.fenc.stride + 4 * bx), v);
            sum += v[0] + v[1] + v[2] + v[3];
        }
        ctu[tid] = (int16_t)sum;
    }
    const int16_t *ref = a.ds + j.ctu_x / 4 + (long)(j.ctu_y / 4) * a.ds_stride + x0 + (long)y0 * a.ds_stride;
    for (int i = tid; i < ww * wh; i += CIME_THREADS)
    {
        const int yy = i / ww, xx = i - yy * ww;
        win[i] = xx < nx + 15 ? ref[(long)yy * a.ds_stride + xx] : (int16_t)0;
    }
    if (j.have_mvd)
    {
        /* mvcost(tmv) per predictor = costx + costy truncated to uint16 (bitcost.h:46) */
        for (int i = tid; i < 3 * nx; i += CIME_THREADS) { const int k = i / nx, c = i - k * nx; costx[k][c] = a.ctab[x0 + c - j.mvp[k][0]]; }
        for (int i = tid; i < 3 * ny; i += CIME_THREADS) { const int k = i / ny, c = i - k * ny; costy[k][c] = a.ctab[y0 + c - j.mvp[k][1]]; }
    }
    __syncthreads();
    if (tid < 256)
    {
        /* the CTU block is indexed with the valid width as its pitch (motion.cpp:1227) */
        const int k = tid & 15, i = tid >> 4;
        blk[tid] = (k < cw && i < ch) ? ctu[k + i * cw] : (int16_t)0;
    }
    __syncthreads();
    const int nd = (cw + 1) >> 1;       /* dwords (sample pairs) per block row */
    const uint32_t lastmask = (cw & 1) ? 0xffffu : 0xffffffffu;
    const int npx = (nx + 1) >> 1;      /* position pairs per row */
    unsigned long long best = ~0ull;
    for (int c = tid; c < npx * ny; c += CIME_THREADS)
    {
        const int cy = c / npx, cx = (c - cy * npx) * 2;
        uint32_t se = 0, so = 0;
        for (int i = 0; i < ch; i++)
        {
            const uint32_t *r = (const uint32_t*)(win + (cy + i) * ww + cx);
            const uint32_t *f = (const uint32_t*)(blk + i * 16);
            uint32_t rr[9];
#pragma unroll
            for (int d = 0; d < 9; d++) rr[d] = r[d];
#pragma unroll
            for (int d = 0; d < 8; d++)
                if (d < nd)
                {
                    /* an odd last sample of the block row is paired with a zero on both sides */
                    const uint32_t m = d == nd - 1 ? lastmask : 0xffffffffu;
                    se = __builtin_amdgcn_sad_u16(f[d], rr[d] & m, se);
                    so = __builtin_amdgcn_sad_u16(f[d], __builtin_amdgcn_alignbit(rr[d + 1], rr[d], 16) & m, so);
                }
        }
#pragma unroll
        for (int o = 0; o < 2; o++)
        {
            const int px = cx + o;
            if (px >= nx) break;
            int mvc = 0;
            if (j.have_mvd)
            {
                mvc = 0x7fffffff;
#pragma unroll
                for (int k = 0; k < 3; k++)
                {
                    const int v = ((int)costx[k][px] + (int)costy[k][cy]) & 0xffff;
                    mvc = v < mvc ? v : mvc;
                }
            }
            const unsigned long long key = ((unsigned long long)((o ? so : se) + (uint32_t)mvc) << 32) | (unsigned)(cy * nx + px);
            best = key < best ? key : best;
        }
    }
    for (int o = 1; o < 64; o <<= 1)
    {
        const unsigned long long other = __shfl_xor(best, o);
        best = other < best ? other : best;
    }
    if ((tid & 63) == 0) red[tid >> 6] = best;
    __syncthreads();
    if (tid == 0)
    {
        for (int k = 1; k < CIME_THREADS / 64; k++) best = red[k] < best ? red[k] : best;
        const int c = (int)(best & 0xffffffffu);
        const int cy = c / nx, cx = c - cy * nx;
        x265amd_rk_result &r = a.results[blockIdx.x];
        r.cime.mv[0] = (int16_t)((x0 + cx) * 4);
        r.cime.mv[1] = (int16_t)((y0 + cy) * 4);
        r.cime.cost = (int)(best >> 32);
    }
}

/* RIME + FME (motion.cpp:1305-1403): one wave per (CTU, CU). RIME scans three 16x12
 * full-pel windows row by row (16 candidates per batched evaluation); FME scans the 7x7
 * quarter-pel grid around the RIME winner, x offset outer, y inner. */
#include "me_rk.h"   /* rk_rime_kernel (RIME of a whole CTU from 8x8 block surfaces; uses RkArgs), rk_fme_grid */

/* HAVE_RIME: the full-pel stage was done by rk_rime_kernel, only FME runs here */
template<typename P, bool HAVE_RIME>
__global__ __launch_bounds__(64) void rk_cu_kernel(RkArgs a)
{
    /* Longest work first: a 64x64 CU is 28 passes of the fractional grid, an 8x8 CU one. Workgroups
     * start in index order, so all 64x64 CUs of the batch come first, then the 32x32, 16x16 and 8x8
     * ones, and the launch ends on short waves instead of waiting for a last large CU. */
    const int n = a.nctus, b = blockIdx.x;
    int ctuIdx, idx;
    if (b < n) { ctuIdx = b; idx = 84; }
    else if (b < 5 * n) { ctuIdx = (b - n) >> 2; idx = 80 + ((b - n) & 3); }
    else if (b < 21 * n) { ctuIdx = (b - 5 * n) >> 4; idx = 64 + ((b - 5 * n) & 15); }
    else { ctuIdx = (b - 21 * n) >> 6; idx = (b - 21 * n) & 63; }
    const x265amd_rk_ctu &j = a.ctus[ctuIdx];
    x265amd_rk_result &res = a.results[ctuIdx];
    const int lane = threadIdx.x & 63;
    int size, ox, oy;
    if (idx < 64) { size = 8; ox = (idx & 7) * 8; oy = (idx >> 3) * 8; }
    else if (idx < 80) { size = 16; ox = ((idx - 64) & 3) * 16; oy = ((idx - 64) >> 2) * 16; }
    else if (idx < 84) { size = 32; ox = ((idx - 80) & 1) * 32; oy = ((idx - 80) >> 1) * 32; }
    else { size = 64; ox = 0; oy = 0; }
    if (j.ctu_x + ox + size > a.pic_w || j.ctu_y + oy + size > a.pic_h)
    {
        if (lane == 0)
        {
            res.rime[idx].mv[0] = res.rime[idx].mv[1] = 0; res.rime[idx].cost = -1;
            res.fme[idx].mv[0] = res.fme[idx].mv[1] = 0; res.fme[idx].cost = -1;
        }
        return;
    }
    Searcher<P, 64, false, true> S;
    /* with the full-pel stage done elsewhere only the fractional grid reads the source block, one
     * tile per lane straight from global memory: no LDS copy, and the reference region sits where
     * the copy would be, which lets five waves instead of three share a SIMD's LDS */
    S.bind(0, a.fenc, &a.recon, 1, a.ctab, a.depth, j.ctu_x + ox, j.ctu_y + oy, size, size, 64, 0, !HAVE_RIME);
    S.nsad = S.nsub = 0;
    S.set_rk_predictors(j.mvp, j.have_mvd);
    const int woff = HAVE_RIME ? L_FENC : wave_lds_bytes<P>(64);
    int bcost = 0x7fffffff, bx = 0, by = 0;
    if (HAVE_RIME)
    {
        bx = uni(res.rime[idx].mv[0]); by = uni(res.rime[idx].mv[1]);
    }
    else
    {
        int pmv[3][2] = { { res.cime.mv[0], res.cime.mv[1] }, { j.pmv[0][0], j.pmv[0][1] }, { j.pmv[1][0], j.pmv[1][1] } };
        for (int p = 0; p < 3; p++)
            for (int i0 = -6; i0 < 6; i0 += 4)
            {
                /* four 16-wide rows of the window per batch, one lane per position */
                const int tx = as_i16(pmv[p][0] + (lane & 15) - 8), ty = as_i16(pmv[p][1] + i0 + (lane >> 4));
                S.put_lanes(true, tx, ty, tx, ty, 16, 0);
                S.eval(64, 0, true);
                const int kb = S.fold_min(64, bcost);
                if (kb >= 0) { bx = S.mvx_of(kb); by = S.mvy_of(kb); }
            }
        if (lane == 0) { res.rime[idx].mv[0] = (int16_t)bx; res.rime[idx].mv[1] = (int16_t)by; res.rime[idx].cost = bcost; }
    }
    const int BX = as_i16(bx << 2), BY = as_i16(by << 2);
    bcost = 0x7fffffff;
    int fx = 0, fy = 0;
    {
        /* the reference region of the 49 fractional positions, copied once into LDS (the
         * reference copies the same neighbourhood into pfref, motion.cpp:1382-1389) */
        const int cols = RK_REGION_COLS(size), rows = RK_REGION_ROWS(size), rx0 = bx - 2, ry0 = by - 2;
        const P AS1 *src = (const P AS1*)a.recon.origin + (long)(j.ctu_y + oy + ry0) * a.recon.stride + j.ctu_x + ox + rx0;
        const int qd = cols >> 2;
        for (int i = lane; i < rows * qd; i += 64)
        {
            const int r = i / qd, q = i - r * qd;
            Px<P>::store_lds_aligned(woff + (r * cols + 4 * q) * (int)sizeof(P), Px<P>::load_g(src + (long)r * a.recon.stride + 4 * q));
        }
        if (lane == 0)
        {
            EvalCtx *ec = lds_ctx(0);
            ec->rk_win = woff; ec->rk_pitch = cols; ec->rk_x0 = rx0; ec->rk_y0 = ry0;
        }
        wave_sync();
    }
    {
        /* the 7x7 grid: distortions from rk_fme_grid, lambda * bits per position, then the first
         * minimum in visit order, x offset outer (motion.cpp:1377-1401): lane c -> (c / 7, c % 7) */
        const int cx = (lane * 37) >> 8, cy = lane - cx * 7;
        const int qx = as_i16(BX + cx - 3), qy = as_i16(BY + cy - 3);
        S.put_lanes(lane < 49, qx, qy, qx, qy, 16, 0);
        wave_sync();
        const int mvc = gather_mvcost(0, 49, 3, 1);
        rk_fme_grid<P, HAVE_RIME>(0, woff, RK_REGION_COLS(size), size, BX, BY, a.depth,
                                  (const P AS1*)a.fenc.origin + (long)(j.ctu_y + oy) * a.fenc.stride + j.ctu_x + ox, a.fenc.stride);
        if (lane < 49) lds_i(0, L_COSTS)[lane] += mvc;
        wave_sync();
        S.nsub += 49;
        const int kb = S.fold_min(49, bcost);
        if (kb >= 0) { fx = S.mvx_of(kb); fy = S.mvy_of(kb); }
    }
    if (lane == 0) { res.fme[idx].mv[0] = (int16_t)fx; res.fme[idx].mv[1] = (int16_t)fy; res.fme[idx].cost = bcost; }
}

/* ---- per-call primitives (EncoderPrimitives mirror); operands are staged copies ---- */

/* sad / satd / sa8d of up to 4 reference blocks against one source block.
 * mode 0 sad, 1 satd, 2 sa8d_inter. One wave; lane-per-tile (4x4, or 8x8 for sa8d). */
template<typename P>
__global__ __launch_bounds__(64) void prim_cmp_kernel(const P *fenc, long fs, const P *r0, const P *r1, const P *r2, const P *r3,
                                                      long rs, int w, int h, int mode, int nref, int32_t *out)
{
    const int lane = threadIdx.x;
    const P *refs[4] = { r0, r1, r2, r3 };
    for (int n = 0; n < nref; n++)
    {
        const P *ref = refs[n];
        int v = 0;
        const bool use8 = mode == 2 && !(w & 7) && !(h & 7);
        if (!use8)
        {
            const int wt = w >> 2, T = wt * (h >> 2);
            for (int t = lane; t < T; t += 64)
            {
                const int ty = t / wt, tx = t - ty * wt;
                typename Px<P>::quad f[4], q[4];
                for (int y = 0; y < 4; y++)
                {
                    f[y] = Px<P>::load(fenc + (ty * 4 + y) * fs + tx * 4);
                    q[y] = Px<P>::load(ref + (ty * 4 + y) * rs + tx * 4);
                }
                v += tile_cost_quads<P>(f, q, mode != 0);
            }
            for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
        }
        else
        {
            /* sa8d: 8x8 Hadamard per lane (pixel.cpp:290-325); 16-multiples round once per
             * 16x16 (pixel.cpp:332-342, :361-376), others per 8x8 (:344-359) */
            const int wb = w >> 3, B = wb * (h >> 3);
            const bool per16 = !(w & 15) && !(h & 15);
            for (int b0 = 0; b0 < B; b0 += 64)
            {
                const int b = b0 + lane;
                int s = 0;
                if (b < B)
                {
                    int bx, by;
                    if (per16)
                    {
                        /* order blocks so that the 4 of one 16x16 sit in adjacent lanes */
                        const int g = b >> 2, sub = b & 3, gw = w >> 4;
                        by = (g / gw) * 2 + (sub >> 1); bx = (g % gw) * 2 + (sub & 1);
                    }
                    else { by = b / wb; bx = b - by * wb; }
                    int m[8][8];
                    for (int y = 0; y < 8; y++)
                    {
                        int fa[8], ra[8];
                        Px<P>::unpack(Px<P>::load(fenc + (by * 8 + y) * fs + bx * 8), fa);
                        Px<P>::unpack(Px<P>::load(fenc + (by * 8 + y) * fs + bx * 8 + 4), fa + 4);
                        Px<P>::unpack(Px<P>::load(ref + (by * 8 + y) * rs + bx * 8), ra);
                        Px<P>::unpack(Px<P>::load(ref + (by * 8 + y) * rs + bx * 8 + 4), ra + 4);
                        for (int x = 0; x < 8; x++) m[y][x] = fa[x] - ra[x];
                    }
#pragma unroll
                    for (int y = 0; y < 8; y++)
#pragma unroll
                        for (int st = 1; st < 8; st <<= 1)
#pragma unroll
                            for (int x = 0; x < 8; x++)
                                if (!(x & st)) { int p = m[y][x], q = m[y][x + st]; m[y][x] = p + q; m[y][x + st] = p - q; }
#pragma unroll
                    for (int x = 0; x < 8; x++)
#pragma unroll
                        for (int st = 1; st < 8; st <<= 1)
#pragma unroll
                            for (int y = 0; y < 8; y++)
                                if (!(y & st)) { int p = m[y][x], q = m[y + st][x]; m[y][x] = p + q; m[y + st][x] = p - q; }
#pragma unroll
                    for (int y = 0; y < 8; y++)
#pragma unroll
                        for (int x = 0; x < 8; x++) s += abs(m[y][x]);
                }
                if (per16)
                {
                    s += __shfl_xor(s, 1);
                    s += __shfl_xor(s, 2);
                    s = ((lane & 3) == 0 && b < B) ? ((s + 2) >> 2) : 0;
                }
                else
                    s = b < B ? ((s + 2) >> 2) : 0;
                for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
                v += s;
            }
        }
        if (lane == 0) out[n] = v;
    }
}

/* mode 0 luma_hpp, 1 luma_vpp, 2 luma_hps, 3 luma_vsp; one thread per output sample */
template<typename P>
__global__ void prim_filter_kernel(const void *srcv, long ss, void *dstv, long ds, int ci, int w, int rows, int mode, int depth)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= rows) return;
    const int maxv = (1 << depth) - 1;
    int sum = 0;
    if (mode == 0 || mode == 2)
    {
        const P *s = (const P*)srcv + y * ss + x - 3;
        for (int k = 0; k < 8; k++) sum += (int)s[k] * kLumaTaps[ci][k];
    }
    else if (mode == 1)
    {
        const P *s = (const P*)srcv + (y - 3) * ss + x;
        for (int k = 0; k < 8; k++) sum += (int)s[k * ss] * kLumaTaps[ci][k];
    }
    else
    {
        const int16_t *s = (const int16_t*)srcv + (y - 3) * ss + x;
        for (int k = 0; k < 8; k++) sum += (int)s[k * ss] * kLumaTaps[ci][k];
    }
    const int headRoom = 14 - depth;
    if (mode == 0 || mode == 1)
        ((P*)dstv)[y * ds + x] = (P)clip_pixel(as_i16((sum + 32) >> 6), maxv);
    else if (mode == 2)
    {
        const int shift = 6 - headRoom;
        ((int16_t*)dstv)[y * ds + x] = (int16_t)((sum - (8192 << shift)) >> shift);
    }
    else
    {
        const int shift = 6 + headRoom;
        ((P*)dstv)[y * ds + x] = (P)clip_pixel(as_i16((sum + (1 << (shift - 1)) + (8192 << 6)) >> shift), maxv);
    }
}

template<typename P>
__global__ void prim_avg_kernel(P *dst, long ds, const P *a, long sa, const P *b, long sb, int w, int h)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    dst[y * ds + x] = (P)(((int)a[y * sa + x] + (int)b[y * sb + x] + 1) >> 1);
}

/* ------------------------------------------------------------------ host */

/* lambda^2 for non-I slices, TComRom.cpp:529-541 (x265_lambda2_non_I) */
static const double kLambda2NonI[70] = {
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

static const uint8_t kPartWH[X265AMD_NUM_LUMA_PARTITIONS][2] = {
    {4,4}, {8,8}, {8,4}, {4,8}, {16,16}, {16,8}, {8,16}, {16,12}, {12,16}, {16,4}, {4,16},
    {32,32}, {32,16}, {16,32}, {32,24}, {24,32}, {32,8}, {8,32},
    {64,64}, {64,32}, {32,64}, {64,48}, {48,64}, {64,16}, {16,64}
};

struct x265amd_ctx
{
    int device;
    int depth;
    hipStream_t stream;
    std::map<int, uint16_t*> tables; /* qp -> device table, 65536 entries, index mvd + 32768 */
    std::map<int, int> table_nowrap; /* qp -> 1 when no two entries of the table add up to 2^16 or more */
    std::string err;
    void *d_jobs; size_t jobs_cap;
    void *d_results; size_t results_cap;
    void *d_stage; size_t stage_cap;   /* staging for the per-call primitives */
    int probe_khz;                     /* constant-rate clock of the last x265amd_clock_probe_start, kHz */
    long long *d_probe;                /* the probe wave's two counters (a buffer of its own: other calls may run meanwhile) */
    /* Working arrays of a search launch, per job of its list; two sets, the second for the children's list of a
     * launch of parents and children (x265amd_search_batch_pairs_family_dev) */
    struct LaunchBufs
    {
        void *d_done; size_t done_cap;     /* refined with its parent already */
        void *d_redo; size_t redo_cap;     /* list of the PUs a fixed-shape kernel handed on (redo_list_kernel) */
        void *d_state; size_t state_cap;   /* where the STAR search of a handed-on PU stands (StarFold::store) */
        void *d_mark; size_t mark_cap;     /* handed on by the STAR kernels */
        void *d_handled; size_t handled_cap; /* parents of a family launch: searched along with their children */
        void *d_squares; size_t squares_cap; /* square_list_kernel's list */
    } bufs[2];
    hipStream_t aux;                   /* second stream of STAR launches: the marked PUs' searches run beside the others' refinement;
                                          made by the first such launch (a context that never searches this way holds no queue for it) */
    int prio;
    hipEvent_t ev_fork, ev_join, ev_a, ev_b;
    bool redo_valid;                   /* d_redo[0] holds the marked-PU count of this context's last search launch */
    int32_t *d_out;
    int *d_ticket;                     /* row hand-out counter of the row-parallel lookahead kernel */
    int esa_window_bytes;              /* LDS the exhaustive search may use per PU for its window; 0 = the library's choice */
    unsigned char *scratch;            /* this device's copy of the static scratch array (x265amd_lookahead_fields_static) */
    bool esa_lds_raised, rime_lds_raised, ctu_lds_raised; /* dynamic-LDS limits raised on this context's device (the attribute is per device;
                                                             a context has one sample size, so one instance of esa_ctu_kernel) */
    std::map<const void*, void*> plane_base; /* plane origin -> start of its allocation, whatever the sample size */
    struct FieldsInfo { int nfields, wcu, hcu, method, merange; };
    std::map<const void*, FieldsInfo> uploads; /* descriptor arrays of x265amd_fields_upload and what they were checked for */
};

static x265amd_ctx *g_prim_ctx = 0;    /* context behind the per-call primitives table (me_primitives.inc) */

/* Device memory of x265amd_lookahead_fields_static, the entry point that allocates nothing: the row ticket of
 * the row-parallel lookahead kernel, the field descriptors and the MV / cost / statistics arrays the kernel writes
 * before they are copied to the caller. Every GPU the library runs on holds its own copy of the array until the
 * process ends. `scratch` is only a host-side handle of the symbol: the address of a device's copy comes from
 * hipGetSymbolAddress (x265amd_create, with that device current) and is what every launch, memset and copy uses. */
#define SCRATCH_BYTES X265AMD_STATIC_SCRATCH_BYTES
#define SCRATCH_HEAD 256               /* the ticket, alone in its 256 bytes */
#define SCRATCH_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];
static void *g_scratch_ptr[SCRATCH_MAX_DEVICES];   /* per device: address of its copy of `scratch`, 0 = not asked yet */
static std::mutex g_scratch_mu;                    /* guards g_scratch_ptr (contexts may be created from several threads) */
static std::atomic<int> g_scratch_busy[SCRATCH_MAX_DEVICES];   /* 1 while a call uses the device's array: a second one is refused */

#define ESA_WINDOW_DEFAULT 36864
#define ESA_WINDOW_MIN 8192
#define ESA_WINDOW_MAX 131072

static int fail(x265amd_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    return code;
}

#define HIPCHK(c, call) do { hipError_t _e = (call); if (_e != hipSuccess) \
        return fail((c), X265AMD_EHIP, std::string(#call) + ": " + hipGetErrorString(_e)); } while (0)

static int ensure(x265amd_ctx *c, void **p, size_t *cap, size_t need)
{
    if (*cap >= need) return 0;
    if (*p) { HIPCHK(c, hipFree(*p)); *p = 0; *cap = 0; }
    size_t n = need + need / 2 + 4096;
    HIPCHK(c, hipMalloc(p, n));
    /* tests only: working arrays start out as garbage, so a kernel that reads one before a launch has written it shows */
    static const bool poison = getenv("X265AMD_POISON") != 0;
    if (poison) HIPCHK(c, hipMemset(*p, 0xA5, n));
    *cap = n;
    return 0;
}

/* the stages around an exhaustive scan (esa_finish_kernel) and the sub-pel refinement of the launch's fixed-shape size */
template<typename P>
static void esa_finish(x265amd_ctx *c, const SearchArgs &b, int window_bytes, hipStream_t st)
{
    const int n = b.njobs;
    hipLaunchKernelGGL(esa_finish_kernel<P>, dim3(n), dim3(64), wave_lds_bytes<P>(b.max_dim, true), st, b, window_bytes);
    if (b.staged == 8)
        hipLaunchKernelGGL((refine_small_kernel<8, P>), dim3((n + 3) / 4), dim3(64), RefineLds<8>::TOTAL, st, b);
    else if (b.staged == 16)
        hipLaunchKernelGGL((refine_small_kernel<16, P>), dim3((n + RefineGeo<16>::NPU - 1) / RefineGeo<16>::NPU), dim3(64), RefineLds<16>::TOTAL, st, b);
    else if (b.staged == 32)
        hipLaunchKernelGGL((refine_big_kernel<32, P>), dim3(n), dim3(BigGeo<32>::THREADS), BigLds<32>::TOTAL, st, b);
    else if (b.staged == 64)
        hipLaunchKernelGGL((refine_big_kernel<64, P>), dim3(n), dim3(BigGeo<64>::THREADS), BigLds<64>::TOTAL, st, b);
}

/* X265_FULL_SEARCH: workgroup-per-PU scan through an LDS window, then the one-wave stages around it */
template<typename P>
static int launch_esa(x265amd_ctx *c, const SearchArgs &a, int window_bytes, int nowrap)
{
    constexpr int BPS = (int)sizeof(P);
    if (!c->esa_lds_raised)
    {
        const int most = esa_fixed_lds(64, 2) + ESA_WINDOW_MAX;
        HIPCHK(c, hipFuncSetAttribute((const void*)esa_scan_kernel<uint8_t, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, most));
        HIPCHK(c, hipFuncSetAttribute((const void*)esa_scan_kernel<uint8_t, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, most));
        HIPCHK(c, hipFuncSetAttribute((const void*)esa_scan_kernel<uint8_t, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, most));
        HIPCHK(c, hipFuncSetAttribute((const void*)esa_scan_kernel<uint16_t, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, most));
        HIPCHK(c, hipFuncSetAttribute((const void*)esa_scan_kernel<uint16_t, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, most));
        HIPCHK(c, hipFuncSetAttribute((const void*)esa_scan_kernel<uint16_t, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, most));
        c->esa_lds_raised = true;
    }
    /* the square PUs of the launch's largest size (8- and 10-bit samples) are refined by the fixed-shape kernels */
    static const bool no_staged = getenv("X265AMD_NO_STAGED") != 0;      /* measurements only */
    SearchArgs b = a;
    b.staged = (BPS == 1 || c->depth <= 10) && !no_staged ? a.max_dim : 0;     /* 8, 16, 32 or 64 */
    const size_t scan_lds = esa_fixed_lds(a.max_dim, BPS) + window_bytes;
    if (a.max_dim == 8)
        hipLaunchKernelGGL((esa_scan_kernel<P, 8>), dim3(a.njobs), dim3(ESA_THREADS), scan_lds, c->stream, b, window_bytes, nowrap);
    else if (a.max_dim == 16)
        hipLaunchKernelGGL((esa_scan_kernel<P, 16>), dim3(a.njobs), dim3(ESA_THREADS), scan_lds, c->stream, b, window_bytes, nowrap);
    else
        hipLaunchKernelGGL((esa_scan_kernel<P, 0>), dim3(a.njobs), dim3(ESA_THREADS), scan_lds, c->stream, b, window_bytes, nowrap);
    esa_finish<P>(c, b, window_bytes, c->stream);
    HIPCHK(c, hipGetLastError());
    return X265AMD_OK;
}

/* X265_FULL_SEARCH of whole CTUs: one scan over all of a CTU's PUs, then the stages around it per PU
 * size. a.jobs: 85 slots per CTU in the order me_esa_ctu.h describes. */
static int aux_stream(x265amd_ctx *c);
template<typename P>
static int launch_esa_ctu(x265amd_ctx *c, const SearchArgs &a, int nctu)
{
    if (!c->ctu_lds_raised)
    {
        HIPCHK(c, hipFuncSetAttribute((const void*)esa_ctu_kernel<P>, hipFuncAttributeMaxDynamicSharedMemorySize, Ectu<P>::L_TOTAL));
        c->ctu_lds_raised = true;
    }
    static const bool no_staged = getenv("X265AMD_NO_STAGED") != 0;      /* measurements only */
    hipLaunchKernelGGL(esa_ctu_kernel<P>, dim3(nctu), dim3(ECTU_THREADS), Ectu<P>::L_TOTAL, c->stream, a, nctu);
    /* the stages around the scan are short launches that depend on the scan alone: the two small sizes on the
     * context's stream, the two large ones beside them on its second stream */
    int rc = aux_stream(c);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->aux, c->ev_fork, 0));
    static const int first[5] = { 0, 64, 80, 84, 85 };
    for (int k = 0; k < 4; k++)
    {
        SearchArgs b = a;
        b.jobs = a.jobs + (size_t)first[k] * nctu; b.results = a.results + (size_t)first[k] * nctu;
        b.njobs = (first[k + 1] - first[k]) * nctu;
        b.max_dim = 8 << k;
        b.staged = no_staged || c->depth > 10 ? 0 : b.max_dim;
        esa_finish<P>(c, b, ESA_WINDOW_DEFAULT, k < 2 ? c->stream : c->aux);
    }
    HIPCHK(c, hipEventRecord(c->ev_join, c->aux));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    HIPCHK(c, hipGetLastError());
    return X265AMD_OK;
}

/* the sub-pel refinement of the launch's fixed-shape sizes: every PU of the job list (list_grid == 0), or the PUs of the
 * redo list, walked by list_grid workgroups */
template<typename P>
static void launch_refine(const SearchArgs &b, hipStream_t st, int list_grid)
{
    const int n = b.njobs;
    if (b.staged == 8)
        hipLaunchKernelGGL((refine_small_kernel<8, P>), dim3(list_grid ? list_grid : b.done ? (n + 63) / 64 : (n + 3) / 4), dim3(64),
                           RefineLds<8>::TOTAL, st, b);
    else if (b.staged == 16)
        hipLaunchKernelGGL((refine_small_kernel<16, P>), dim3(list_grid ? list_grid : (n + RefineGeo<16>::NPU - 1) / RefineGeo<16>::NPU), dim3(64),
                           RefineLds<16>::TOTAL, st, b);
    else if (b.staged == 32)
        hipLaunchKernelGGL((refine_big_kernel<32, P>), dim3(list_grid ? list_grid : n), dim3(BigGeo<32>::THREADS), BigLds<32>::TOTAL, st, b);
    else if (b.staged == 64)
    {
        hipLaunchKernelGGL((refine_big_kernel<64, P>), dim3(list_grid ? list_grid : n), dim3(BigGeo<64>::THREADS), BigLds<64>::TOTAL, st, b);
        if (b.staged2)      /* the 32x32 PUs their parents did not take along */
            hipLaunchKernelGGL((refine_big_kernel<32, P>), dim3(list_grid ? list_grid : n), dim3(BigGeo<32>::THREADS), BigLds<32>::TOTAL, st, b);
    }
}

/* PUs of the job list a wave of the small partitions' kernels looks at together: as many as leave several thousand waves
 * (its groups take 2 to 8 PUs at a time) */
static int tiny_chunk(int njobs)
{
    static const int fixed = getenv("X265AMD_TINY_CHUNK") ? atoi(getenv("X265AMD_TINY_CHUNK")) : 0;     /* measurements only */
    if (fixed >= 8 && fixed <= 64 && !(fixed & (fixed - 1))) return fixed;
    int chunk = 64;
    while (chunk > 8 && njobs / chunk < 16384) chunk >>= 1;
    return chunk;
}

/* the sub-pel refinement of the small other partitions, several PUs per wave: the job list or the redo list */
template<typename P>
static void launch_tiny(const SearchArgs &b, hipStream_t st, int list_grid)
{
    static const bool no_tiny = getenv("X265AMD_NO_REFINE_TINY") != 0;     /* measurements only: then they are nobody's */
    const int chunk = list_grid ? 8 : tiny_chunk(b.njobs);
    int tgrid = list_grid ? list_grid : (b.njobs + chunk - 1) / chunk;
    if (tgrid > 65536) tgrid = 65536;
    if (!no_tiny) hipLaunchKernelGGL((refine_tiny_kernel<P>), dim3(tgrid), dim3(64), TINY_LDS_MAX, st, b, chunk);
}

/* the sub-pel refinement of the PUs no fixed-shape kernel takes (other partitions, squares of other sizes): every PU of
 * the job list, or of the redo list walked by list_grid waves */
template<typename P>
static void launch_any(const SearchArgs &b, hipStream_t st, int list_grid)
{
    /* a wave looks at `chunk` PUs of the job list at once: as many as leave sixteen thousand waves (measured: 32x16 lists
     * 0.58 ms with four thousand, 0.52 with sixteen; a list of squares pays 1.5 % for the finer screening) */
    static const int any_waves = getenv("X265AMD_ANY_WAVES") ? atoi(getenv("X265AMD_ANY_WAVES")) : 16384;     /* measurements only */
    int chunk = 64;
    while (chunk > 1 && b.njobs / chunk < any_waves) chunk >>= 1;
    int grid = list_grid ? list_grid : (b.njobs + chunk - 1) / chunk;
    if (grid > 65536) grid = 65536;
    if (b.max_dim <= 16)
        hipLaunchKernelGGL((refine_any_kernel<P, 1>), dim3(grid), dim3(64), AnyLds<1>::TOTAL, st, b, chunk);
    else if (b.max_dim == 32)
        hipLaunchKernelGGL((refine_any_kernel<P, 4>), dim3(grid), dim3(64), AnyLds<4>::TOTAL, st, b, chunk);
    else
        hipLaunchKernelGGL((refine_any_kernel<P, 12>), dim3(grid), dim3(64), AnyLds<12>::TOTAL, st, b, chunk);
    launch_tiny<P>(b, st, list_grid);
}

/* waves of a launch of search_kernel: a wave per PU up to this many, then every wave walks the list */
#define SEARCH_GRID 8192

/* A full-resolution search launch in phases. prepare: which sizes take the fixed-shape kernels, working arrays.
 * STAR launches of those sizes then run star (start-up candidates + first three pattern rounds of every PU, on the
 * context's stream), side (the PUs handed on: list, rest of their searches, their refinement, on the stream given)
 * and refine (everything else, on the context's stream); other launches run the one-wave search and refine. */
struct SearchLaunch
{
    SearchArgs b;
    bool star, square_only;
    int set;
};

template<typename P>
static int search_prepare(x265amd_ctx *c, const SearchArgs &a, bool square_only, int set, SearchLaunch &sl)
{
    static const bool no_staged = getenv("X265AMD_NO_STAGED") != 0;                            /* measurements only */
    static const bool no_star = getenv("X265AMD_NO_STAR8") != 0;
    static const bool no_big = getenv("X265AMD_NO_STAGED_BIG") != 0;
    static const bool no_star_big = getenv("X265AMD_NO_STAR_BIG") != 0;
    static const bool no_quad = getenv("X265AMD_NO_QUAD") != 0;
    x265amd_ctx::LaunchBufs &B = c->bufs[set];
    SearchArgs &b = sl.b;
    b = a;
    sl.square_only = square_only && !(no_quad && a.max_dim == 64); sl.set = set;     /* (without the quad path the 32x32 PUs are the one-wave search's) */
    /* launches of PUs up to 8x8, 16x16, 32x32 or 64x64 (samples of at most 10 bits): the square PUs of that size are
     * refined by a second kernel (four 8x8 PUs or two 16x16 PUs per wave; two waves per 32x32 PU, four per 64x64 PU) */
    b.staged = (sizeof(P) == 1 || c->depth <= 10) && !no_staged &&
               (a.max_dim == 8 || a.max_dim == 16 || ((a.max_dim == 32 || a.max_dim == 64) && !no_big)) ? a.max_dim : 0;
    if (b.staged == 64 && !no_quad)
    {
        /* a launch for 64x64 PUs takes the 32x32 PUs of its list through the fixed-shape kernels as well; they are
         * refined together with their parent where the list has the four children right behind it */
        if (ensure(c, &B.d_done, &B.done_cap, (size_t)a.njobs)) return X265AMD_EHIP;      /* cleared by the STAR kernels */
        b.staged2 = 32;
        b.done = (unsigned char*)B.d_done;
    }
    static const bool no_pat = getenv("X265AMD_NO_PATTERN_FIXED") != 0;        /* HEX / DIA through the one-wave search, as before */
    static const bool no_any = getenv("X265AMD_NO_REFINE_ANY") != 0;           /* other partitions refined by the one-wave search, as before */
    b.stage_all = b.staged && !sl.square_only && !no_any;
    sl.star = b.staged && (a.method == X265AMD_STAR || ((a.method == X265AMD_HEX || a.method == X265AMD_DIA) && !no_pat)) &&
              !no_star && (b.staged <= 16 || !no_star_big);
    if (sl.star)
    {
        if (ensure(c, &B.d_redo, &B.redo_cap, sizeof(int) * ((size_t)REDO_LIST + a.njobs))) return X265AMD_EHIP;
        if (ensure(c, &B.d_state, &B.state_cap, 32 * (size_t)a.njobs)) return X265AMD_EHIP;
        if (ensure(c, &B.d_mark, &B.mark_cap, (size_t)a.njobs)) return X265AMD_EHIP;
        b.redo = (int*)B.d_redo;
        b.state = (int*)B.d_state;
        b.mark = (unsigned char*)B.d_mark;
    }
    return X265AMD_OK;
}

static int aux_stream(x265amd_ctx *c)
{
    if (!c->aux) HIPCHK(c, hipStreamCreateWithPriority(&c->aux, hipStreamNonBlocking, c->prio));
    return X265AMD_OK;
}

/* STAR: start-up candidates and the first three pattern rounds of every PU in fixed-shape kernels. Most PUs end
 * there; the others are marked and listed. */
template<typename P, int M>
static void search_star_m(const SearchArgs &b, hipStream_t st)
{
    if (b.staged == 8)
        hipLaunchKernelGGL((star_small_kernel<8, P, M>), dim3((b.njobs + 3) / 4), dim3(64), StarLds<8>::TOTAL, st, b);
    else if (b.staged == 16)
        hipLaunchKernelGGL((star_small_kernel<16, P, M>), dim3(b.njobs), dim3(64), StarLds<16>::TOTAL, st, b);
    else
    {
        if (b.staged == 64)
            hipLaunchKernelGGL((star_big_kernel<64, P, M>), dim3(b.njobs), dim3(StarBigGeo<64>::THREADS), StarBigLds<64>::TOTAL, st, b);
        if (b.staged == 32 || b.staged2 == 32)
            hipLaunchKernelGGL((star_big_kernel<32, P, M>), dim3(b.njobs), dim3(StarBigGeo<32>::THREADS), StarBigLds<32>::TOTAL, st, b);
    }
}

/* the larger other partitions: a kernel per class of shapes (longer side 32, 64) the list can hold; 16x12 / 12x16 are
 * star_tiny_kernel's */
template<typename P, int M>
static void search_rect_m(const SearchArgs &b, hipStream_t st)
{
    const int bps = (int)sizeof(P);
    if (b.max_dim >= 32) hipLaunchKernelGGL((star_rect_kernel<P, M, 32>), dim3(b.njobs), dim3(64), star_rect_lds(32, bps), st, b);
    if (b.max_dim >= 64) hipLaunchKernelGGL((star_rect_kernel<P, M, 64>), dim3(b.njobs), dim3(64), star_rect_lds(64, bps), st, b);
}

template<typename P>
static int search_star(x265amd_ctx *c, SearchLaunch &sl, hipStream_t st)
{
    const SearchArgs &b = sl.b;
    if (b.method == X265AMD_HEX) search_star_m<P, X265AMD_HEX>(b, st);
    else if (b.method == X265AMD_DIA) search_star_m<P, X265AMD_DIA>(b, st);
    else search_star_m<P, X265AMD_STAR>(b, st);
    return X265AMD_OK;
}

/* the PUs the STAR kernels handed on: star_resume_kernel takes their searches to the end, the PUs that are no
 * fixed-shape kernel's (other shapes in the list) go through the one-wave search from the start, and the list-driven
 * refinement finishes them */
template<typename P>
static void search_side(x265amd_ctx *c, SearchLaunch &sl, hipStream_t side)
{
    static const int pad = getenv("X265AMD_LDS_PAD") ? atoi(getenv("X265AMD_LDS_PAD")) : 0;   /* occupancy experiments */
    const SearchArgs &b = sl.b;
    const int blocks = (b.njobs + REDO_CHUNK - 1) / REDO_CHUNK, lgrid = b.njobs < REDO_GRID ? b.njobs : REDO_GRID;
    hipLaunchKernelGGL(redo_list_kernel, dim3(blocks), dim3(64), 0, side, b.results, b.njobs, b.redo);
    hipLaunchKernelGGL((star_resume_kernel<P>), dim3(lgrid), dim3(64), wave_lds_bytes<P>(b.max_dim), side, b);
    /* a caller that states its list holds only the squares these kernels take saves the launch that looks for others */
    if (!sl.square_only)
        hipLaunchKernelGGL((search_kernel<P, false>), dim3(lgrid), dim3(64), wave_lds_bytes<P>(b.max_dim, true) + pad, side, b);
    SearchArgs l = b;
    l.list = 1;
    launch_refine<P>(l, side, lgrid);
    if (b.stage_all) launch_any<P>(l, side, lgrid);
}

template<typename P>
static int launch_search(x265amd_ctx *c, const SearchArgs &a, int is_lowres, bool square_only)
{
    c->redo_valid = false;
    if (is_lowres)
    {
        hipLaunchKernelGGL((search_kernel<P, true>), dim3(a.njobs < SEARCH_GRID ? a.njobs : SEARCH_GRID), dim3(64), wave_lds_bytes<P>(8), c->stream, a);
        HIPCHK(c, hipGetLastError());
        return X265AMD_OK;
    }
    SearchLaunch sl;
    int rc = search_prepare<P>(c, a, square_only, 0, sl);
    if (rc) return rc;
    if (sl.star)
    {
        /* the small other partitions of the list through a fixed-shape stage of their own */
        static const bool no_star_tiny = getenv("X265AMD_NO_STAR_TINY") != 0;                    /* measurements only */
        sl.b.tiny = sl.b.stage_all && !no_star_tiny;
        static const bool no_star_rect = getenv("X265AMD_NO_STAR_RECT") != 0;
        sl.b.rect = sl.b.tiny && a.max_dim >= 16 && !no_star_rect;
        if ((rc = search_star<P>(c, sl, c->stream))) return rc;
        if (sl.b.tiny)
        {
            const int chunk = tiny_chunk(a.njobs);
            const int rect_lds = star_rect_lds(16, (int)sizeof(P));
            const int tiny_lds = StarTinyMax<(int)sizeof(P)>::TOTAL > rect_lds ? StarTinyMax<(int)sizeof(P)>::TOTAL : rect_lds;
            int tgrid = (a.njobs + chunk - 1) / chunk;
            if (tgrid > 65536) tgrid = 65536;
            if (a.method == X265AMD_HEX) hipLaunchKernelGGL((star_tiny_kernel<P, X265AMD_HEX>), dim3(tgrid), dim3(64), tiny_lds, c->stream, sl.b, chunk);
            else if (a.method == X265AMD_DIA) hipLaunchKernelGGL((star_tiny_kernel<P, X265AMD_DIA>), dim3(tgrid), dim3(64), tiny_lds, c->stream, sl.b, chunk);
            else hipLaunchKernelGGL((star_tiny_kernel<P, X265AMD_STAR>), dim3(tgrid), dim3(64), tiny_lds, c->stream, sl.b, chunk);
        }
        if (sl.b.rect)
        {
            if (a.method == X265AMD_HEX) search_rect_m<P, X265AMD_HEX>(sl.b, c->stream);
            else if (a.method == X265AMD_DIA) search_rect_m<P, X265AMD_DIA>(sl.b, c->stream);
            else search_rect_m<P, X265AMD_STAR>(sl.b, c->stream);
        }
        /* The PUs handed on are few as a rule and each a long chain of dependent steps in one wave: their
         * searches and refinement run on the context's second stream while this one refines all the others. */
        static const bool never_fork = getenv("X265AMD_NO_FORK") != 0;                           /* measurements only */
        static const int fork_min = getenv("X265AMD_FORK_MIN") ? atoi(getenv("X265AMD_FORK_MIN")) : 2048;
        /* a short list is a chain of short kernels either way, and the two waits between the streams cost more than the
         * overlap gives (the 1 200 large PUs of a CTU row of 8 frame pairs: 0.23 ms per row forked, 0.17 in a row; at
         * 4 800 PUs the two forms measure alike, from some ten thousand on the fork pays) */
        const bool no_fork = never_fork || a.njobs < fork_min;
        if (!no_fork)
        {
            if ((rc = aux_stream(c))) return rc;
            HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->aux, c->ev_fork, 0));
        }
        search_side<P>(c, sl, no_fork ? c->stream : c->aux);
        if (!no_fork) HIPCHK(c, hipEventRecord(c->ev_join, c->aux));
        /* the squares of a 32 / 64 launch whose list may hold anything: through a list of them (the order of the list is
         * the blocks' arrival, so a 64x64 PU still finds its children behind it in the job list) */
        static const bool no_square_list = getenv("X265AMD_NO_SQUARE_LIST") != 0;                /* measurements only */
        if (sl.b.staged >= 32 && !sl.square_only && !no_square_list)
        {
            x265amd_ctx::LaunchBufs &B = c->bufs[0];
            if (ensure(c, &B.d_squares, &B.squares_cap, sizeof(int) * ((size_t)REDO_LIST + a.njobs))) return X265AMD_EHIP;
            HIPCHK(c, hipMemsetAsync(B.d_squares, 0, sizeof(int), c->stream));
            const int code1 = sl.b.staged | (sl.b.staged << 8), code2 = sl.b.staged2 ? sl.b.staged2 | (sl.b.staged2 << 8) : code1;
            hipLaunchKernelGGL(square_list_kernel, dim3((a.njobs + REDO_CHUNK - 1) / REDO_CHUNK), dim3(64), 0, c->stream,
                               sl.b.jobs, sl.b.mark, a.njobs, code1, code2, (int*)B.d_squares);
            SearchArgs l2 = sl.b;
            l2.list = 2;
            l2.redo = (int*)B.d_squares;
            launch_refine<P>(l2, c->stream, a.njobs < 32768 ? a.njobs : 32768);     /* (a workgroup per PU up to there) */
        }
        else
        launch_refine<P>(sl.b, c->stream, 0);
        /* (the PUs star_tiny_kernel / star_rect_kernel ended; the others are in the list) */
        if (sl.b.rect) launch_any<P>(sl.b, c->stream, 0);
        else if (sl.b.tiny) launch_tiny<P>(sl.b, c->stream, 0);
        if (!no_fork) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
        c->redo_valid = true;
    }
    else
    {
        static const int pad = getenv("X265AMD_LDS_PAD") ? atoi(getenv("X265AMD_LDS_PAD")) : 0;   /* occupancy experiments */
        /* a launch of large PUs that refines elsewhere interpolates only its quarter-pel start-up candidates: room
         * for one plane instead of two lets a compute unit hold a third more of its waves */
        static const bool lean = getenv("X265AMD_NO_LEAN") == 0;
        int lds = wave_lds_bytes<P>(a.max_dim, true);
        if (sl.b.staged >= 32 && lean)
        {
            sl.b.plane_lds = plane_budget(a.max_dim, (int)sizeof(P)) / 2;
            lds -= sl.b.plane_lds;
        }
        /* "refined with its parent": the STAR kernels clear these marks for their launches; here nothing else does */
        if (sl.b.done) HIPCHK(c, hipMemsetAsync(sl.b.done, 0, (size_t)a.njobs, c->stream));
        hipLaunchKernelGGL((search_kernel<P, false>), dim3(a.njobs < SEARCH_GRID ? a.njobs : SEARCH_GRID), dim3(64), lds + pad, c->stream, sl.b);
        launch_refine<P>(sl.b, c->stream, 0);
        if (sl.b.stage_all) launch_any<P>(sl.b, c->stream, 0);
    }
    HIPCHK(c, hipGetLastError());
    return X265AMD_OK;
}

/* A list of 16x16 PUs and the list of their 8x8 children as one launch: both lists' STAR kernels, then both lists'
 * handed-on PUs on the second stream while the 16x16 PUs are refined with the children they can take along, then the
 * remaining 8x8 PUs. Results are those of two launches of their own. */
template<typename P>
static int launch_family(x265amd_ctx *c, const SearchArgs &parents, const SearchArgs &children)
{
    c->redo_valid = false;
    SearchLaunch pa, ch;
    int rc = search_prepare<P>(c, parents, true, 0, pa);
    if (!rc) rc = search_prepare<P>(c, children, true, 1, ch);
    if (rc) return rc;
    if (!pa.star || !ch.star || pa.b.staged != 16 || ch.b.staged != 8) return fail(c, X265AMD_EINVAL, "family launch: STAR, HEX or DIA search of 16x16 and 8x8 PUs, samples of at most 10 bits");
    static const bool no_quad = getenv("X265AMD_NO_QUAD") != 0;                                  /* measurements only */
    if (!no_quad)
    {
        /* which children their parents have refined */
        x265amd_ctx::LaunchBufs &B = c->bufs[1];
        if (ensure(c, &B.d_done, &B.done_cap, (size_t)children.njobs)) return X265AMD_EHIP;    /* cleared by the STAR kernel */
        ch.b.done = (unsigned char*)B.d_done;
        pa.b.cjobs = ch.b.jobs; pa.b.cresults = ch.b.results; pa.b.ncjobs = ch.b.njobs;
        pa.b.cmark = ch.b.mark; pa.b.cdone = ch.b.done;
    }
    static const bool no_share = getenv("X265AMD_NO_STAR_SHARE") != 0;                           /* measurements only */
    const bool share = !no_share;
    if ((rc = aux_stream(c))) return rc;
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->aux, c->ev_fork, 0));
    if (share)
    {
        /* the children's kernel searches the parents along where the five jobs ask the same (their SADs are the
         * children's sums); the parents' kernel then takes the parents left over */
        x265amd_ctx::LaunchBufs &B = c->bufs[0];
        if (ensure(c, &B.d_handled, &B.handled_cap, (size_t)parents.njobs)) return X265AMD_EHIP;
        HIPCHK(c, hipMemsetAsync(B.d_handled, 0, (size_t)parents.njobs, c->aux));
        ch.b.pjobs = pa.b.jobs; ch.b.presults = pa.b.results; ch.b.npjobs = pa.b.njobs;
        ch.b.pmark = pa.b.mark; ch.b.pstate = pa.b.state;
        ch.b.handled = pa.b.handled = (unsigned char*)B.d_handled;
        if ((rc = search_star<P>(c, ch, c->aux))) return rc;
        HIPCHK(c, hipEventRecord(c->ev_b, c->aux));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_b, 0));
        if ((rc = search_star<P>(c, pa, c->stream))) return rc;
        HIPCHK(c, hipEventRecord(c->ev_a, c->stream));
    }
    else
    {
        /* the two lists' STAR kernels side by side on the two streams */
        if ((rc = search_star<P>(c, pa, c->stream)) || (rc = search_star<P>(c, ch, c->aux))) return rc;
        HIPCHK(c, hipEventRecord(c->ev_a, c->stream));
        HIPCHK(c, hipEventRecord(c->ev_b, c->aux));
    }
    /* then the handed-on PUs of both lists on the second stream */
    search_side<P>(c, ch, c->aux);                          /* needs the children's STAR kernel only */
    HIPCHK(c, hipStreamWaitEvent(c->aux, c->ev_a, 0));
    if (!share) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_b, 0));
    search_side<P>(c, pa, c->aux);
    HIPCHK(c, hipEventRecord(c->ev_join, c->aux));
    launch_refine<P>(pa.b, c->stream, 0);
    launch_refine<P>(ch.b, c->stream, 0);
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    c->redo_valid = true;
    HIPCHK(c, hipGetLastError());
    return X265AMD_OK;
}

/* Both lookahead kernels give the same fields. One workgroup per field keeps the launch free of
 * global synchronisation and is enough when hundreds of fields are in flight; the row-parallel
 * kernel spreads a field over up to min(H, W / 2) waves and wins whenever the fields alone cannot
 * fill the chip. LA_ROWS_MAX_FIELDS is where a launch switches over. */
#define LA_ROWS_MAX_FIELDS 512
#define LB_LATENCY_WAVES 1024      /* band waves up to which a lookahead launch trades lane work for latency */
#ifdef X265AMD_PHASE_PROFILE
static long long *g_phase_buffer = 0;
/* measurement build only: where the next lookahead_band_kernel launches write their per-wave phase counters
 * (8192 waves x 16 counters of 8 bytes, zeroed by the caller), or null */
extern "C" void x265amd_debug_phase_buffer(void *d_buffer) { g_phase_buffer = (long long*)d_buffer; }
#endif

static int launch_lookahead(x265amd_ctx *c, LookaheadArgs &a, int nfields, int *d_ticket)
{
#ifdef X265AMD_PHASE_PROFILE
    a.phase = g_phase_buffer;
#endif
    const char *force = getenv("X265AMD_LOOKAHEAD_KERNEL");   /* "field", "rows" or "band": measurements only */
    /* the hexagon search -- the only method the reference's lookahead uses -- has a kernel of its own (four CU rows per
     * wave, any number of fields) */
    bool band = a.method == X265AMD_HEX && (force ? !strcmp(force, "band") : true);
    bool rows = band || (force ? strcmp(force, "field") != 0 : nfields <= LA_ROWS_MAX_FIELDS);
    /* the row-parallel kernels mark unfinished CUs with MV x = 0x7FFF: a grid wide enough for a real quarter-pel MV
     * of that value (range end (wcu * 8 + 8) * 4 + 3) goes to the kernel that needs no mark */
    if ((a.wcu * 8 + 8) * 4 + 3 >= 0x7FFF) rows = band = false;
    a.nfields = nfields;
    a.ticket = d_ticket;
    if (!rows)
    {
        if (c->depth <= 8)
            hipLaunchKernelGGL(lookahead_kernel<uint8_t>, dim3(nfields), dim3(LA_WAVES * 64), LA_WAVES * wave_lds_bytes<uint8_t>(8), c->stream, a);
        else
            hipLaunchKernelGGL(lookahead_kernel<uint16_t>, dim3(nfields), dim3(LA_WAVES * 64), LA_WAVES * wave_lds_bytes<uint16_t>(8), c->stream, a);
    }
    else
    {
        const size_t words = (size_t)nfields * a.wcu * a.hcu;
        HIPCHK(c, hipMemsetD32Async((hipDeviceptr_t)a.mvs, LA_MV_UNSET, words, c->stream));
        HIPCHK(c, hipMemsetAsync(d_ticket, 0, sizeof(int), c->stream));
        if (band)
        {
            /* CU rows per wave: a CU on 32 lanes measures eight candidates per pass instead of four and is through
             * ~10 % sooner, for a third more lane work (on all 64 lanes it is no faster still). A field's latency is its
             * dependency chain of W + 2 (H - 1) CU searches whatever the rows per wave, so a launch that leaves most of the
             * chip idle takes the wide form and one that can fill it the narrow one. */
            const char *fr = getenv("X265AMD_LOOKAHEAD_ROWS");                     /* "2" or "4": tests and measurements */
            const int force_rows = fr ? atoi(fr) : 0;
            int rows = (long)nfields * ((a.hcu + 1) / 2) <= LB_LATENCY_WAVES ? 2 : LB_ROWS;
            if (force_rows == 2 || force_rows == 4) rows = force_rows;
            long nb = (long)nfields * ((a.hcu + rows - 1) / rows);
            if (nb > 8192) nb = 8192;
            const bool p8 = c->depth <= 8;
            if (rows == 4)
            {
                if (p8) hipLaunchKernelGGL((lookahead_band_kernel<uint8_t, 4>), dim3((unsigned)nb), dim3(64), (band_lds_bytes<uint8_t, 4>()), c->stream, a);
                else hipLaunchKernelGGL((lookahead_band_kernel<uint16_t, 4>), dim3((unsigned)nb), dim3(64), (band_lds_bytes<uint16_t, 4>()), c->stream, a);
            }
            else
            {
                if (p8) hipLaunchKernelGGL((lookahead_band_kernel<uint8_t, 5>), dim3((unsigned)nb), dim3(64), (band_lds_bytes<uint8_t, 5>()), c->stream, a);
                else hipLaunchKernelGGL((lookahead_band_kernel<uint16_t, 5>), dim3((unsigned)nb), dim3(64), (band_lds_bytes<uint16_t, 5>()), c->stream, a);
            }
            HIPCHK(c, hipGetLastError());
            return X265AMD_OK;
        }
        long waves = (long)nfields * a.hcu;
        if (waves > 8192) waves = 8192;     /* more than the chip holds at once; the rest of the rows go to waves that finish */
        if (c->depth <= 8)
            hipLaunchKernelGGL(lookahead_rows_kernel<uint8_t>, dim3((unsigned)waves), dim3(64), wave_lds_bytes<uint8_t>(8), c->stream, a);
        else
            hipLaunchKernelGGL(lookahead_rows_kernel<uint16_t>, dim3((unsigned)waves), dim3(64), wave_lds_bytes<uint16_t>(8), c->stream, a);
    }
    HIPCHK(c, hipGetLastError());
    return X265AMD_OK;
}

extern "C" {

int x265amd_partition_from_sizes(int w, int h)
{
    for (int i = 0; i < X265AMD_NUM_LUMA_PARTITIONS; i++)
        if (kPartWH[i][0] == w && kPartWH[i][1] == h) return i;
    return -1;
}

static int create_ctx(int device, int bit_depth, int high_priority, x265amd_ctx **out);

int x265amd_create(int device, int bit_depth, x265amd_ctx **out) { return create_ctx(device, bit_depth, 0, out); }
int x265amd_create_priority(int device, int bit_depth, int high_priority, x265amd_ctx **out)
{
    return create_ctx(device, bit_depth, high_priority, out);
}

static int create_ctx(int device, int bit_depth, int high_priority, x265amd_ctx **out)
{
    if (!out || !(bit_depth == 8 || bit_depth == 10 || bit_depth == 12)) return X265AMD_EINVAL;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return X265AMD_EHIP;
    x265amd_ctx *c = new x265amd_ctx();
    c->device = device; c->depth = bit_depth;
    memset(c->bufs, 0, sizeof(c->bufs)); c->redo_valid = false;
    c->d_jobs = c->d_results = c->d_stage = 0; c->d_ticket = 0; c->d_probe = 0;
    c->jobs_cap = c->results_cap = c->stage_cap = 0; c->d_out = 0;
    c->esa_window_bytes = 0;
    c->probe_khz = 0;
    c->esa_lds_raised = c->rime_lds_raised = c->ctu_lds_raised = false;
    int prio_least = 0, prio_greatest = 0;
    if (hipSetDevice(device) != hipSuccess || hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess)
    {
        delete c;
        return X265AMD_EHIP;
    }
    const int prio = high_priority > 0 ? prio_greatest : high_priority < 0 ? prio_least : prio_least + (prio_greatest - prio_least) / 2;
    c->aux = 0; c->prio = prio;
    if (hipStreamCreateWithPriority(&c->stream, hipStreamDefault, prio) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_a, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_b, hipEventDisableTiming) != hipSuccess ||
        hipMalloc((void**)&c->d_out, 64) != hipSuccess || hipMalloc((void**)&c->d_ticket, 256) != hipSuccess)
    {
        delete c;
        return X265AMD_EHIP;
    }
    /* this device's copy of the static scratch array (the device is current here) */
    c->scratch = 0;
    if (device < SCRATCH_MAX_DEVICES)
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        if (!g_scratch_ptr[device])
        {
            void *ptr = 0;
            if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)) != hipSuccess || !ptr)
            {
                (void)hipFree(c->d_out); (void)hipFree(c->d_ticket); (void)hipStreamDestroy(c->stream);
                (void)hipEventDestroy(c->ev_fork); (void)hipEventDestroy(c->ev_join); (void)hipEventDestroy(c->ev_a); (void)hipEventDestroy(c->ev_b);
                delete c;
                return X265AMD_EHIP;
            }
            g_scratch_ptr[device] = ptr;
        }
        c->scratch = (unsigned char*)g_scratch_ptr[device];
    }
    *out = c;
    return X265AMD_OK;
}

void x265amd_destroy(x265amd_ctx *c)
{
    if (!c) return;
    if (g_prim_ctx == c) g_prim_ctx = 0;   /* the primitives table it served now fails loudly instead of reading freed memory */
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (std::map<int, uint16_t*>::iterator it = c->tables.begin(); it != c->tables.end(); ++it) (void)hipFree(it->second);
    if (c->d_jobs) (void)hipFree(c->d_jobs);
    if (c->d_results) (void)hipFree(c->d_results);
    if (c->d_stage) (void)hipFree(c->d_stage);
    for (int i = 0; i < 2; i++)
    {
        if (c->bufs[i].d_redo) (void)hipFree(c->bufs[i].d_redo);
        if (c->bufs[i].d_done) (void)hipFree(c->bufs[i].d_done);
        if (c->bufs[i].d_state) (void)hipFree(c->bufs[i].d_state);
        if (c->bufs[i].d_mark) (void)hipFree(c->bufs[i].d_mark);
        if (c->bufs[i].d_handled) (void)hipFree(c->bufs[i].d_handled);
        if (c->bufs[i].d_squares) (void)hipFree(c->bufs[i].d_squares);
    }
    if (c->aux)
    {
        (void)hipStreamSynchronize(c->aux);
        (void)hipStreamDestroy(c->aux);
    }
    (void)hipEventDestroy(c->ev_fork);
    (void)hipEventDestroy(c->ev_join);
    (void)hipEventDestroy(c->ev_a);
    (void)hipEventDestroy(c->ev_b);
    if (c->d_probe) (void)hipFree(c->d_probe);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->d_ticket) (void)hipFree(c->d_ticket);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *x265amd_last_error(const x265amd_ctx *c) { return c ? c->err.c_str() : "null context"; }
void *x265amd_stream(x265amd_ctx *c) { return c ? (void*)c->stream : 0; }

/* BitCost::CalculateLogs + setQP (bitcost.cpp:34-83): bit sizes in float,
 * scaled by sqrt(lambda2) in double, +0.5f, clamped to 65535, truncated. */
static void build_cost_table(int qp, uint16_t *tab /* 65536, index mvd + 32768 */)
{
    const double lambda = sqrt(kLambda2NonI[qp]);
    const float log2_2 = 2.0f / logf(2.0f);
    tab[0] = 0;
    for (int i = 0; i < 32768; i++)
    {
        const float bits = i ? logf((float)(i + 1)) * log2_2 + 1.718f : 0.718f;
        double v = bits * lambda + 0.5f;
        if (v > 65535.0) v = 65535.0;
        const uint16_t cst = (uint16_t)v;
        tab[32768 + i] = cst;
        tab[32768 - i] = cst;
    }
}

int x265amd_set_qp(x265amd_ctx *c, int qp)
{
    if (!c || qp < 0 || qp >= 70) return fail(c, X265AMD_EINVAL, "qp out of range [0, 69]");
    if (c->tables.count(qp)) return X265AMD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<uint16_t> tab(65536);
    build_cost_table(qp, tab.data());
    uint16_t *d = 0;
    HIPCHK(c, hipMalloc((void**)&d, 65536 * sizeof(uint16_t)));
    HIPCHK(c, hipMemcpy(d, tab.data(), 65536 * sizeof(uint16_t), hipMemcpyHostToDevice));
    c->tables[qp] = d;
    c->table_nowrap[qp] = 2 * (int)tab[65535] < 65536;   /* the table grows with |mvd|: its last entry is the largest */
    return X265AMD_OK;
}

int x265amd_cost_table(x265amd_ctx *c, int qp, uint16_t *out)
{
    if (!c || !out || qp < 0 || qp >= 70) return fail(c, X265AMD_EINVAL, "bad argument");
    /* the device copy, i.e. what the kernels read */
    int rc = x265amd_set_qp(c, qp);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(out, c->tables[qp] + 1, 65535 * sizeof(uint16_t), hipMemcpyDeviceToHost));
    return X265AMD_OK;
}

static size_t plane_bytes(const x265amd_ctx *c, int height, int margin_y, int stride)
{
    return (size_t)(height + 2 * margin_y) * stride * (c->depth > 8 ? 2 : 1);
}

static int plane_alloc(x265amd_ctx *c, int width, int height, int mx, int my, int stride, x265amd_plane *out)
{
    const int bpp = c->depth > 8 ? 2 : 1;
    void *d = 0;
    HIPCHK(c, hipSetDevice(c->device));
    /* + slack: window staging reads whole 16-byte chunks and may run past the last row */
    HIPCHK(c, hipMalloc(&d, plane_bytes(c, height, my, stride) + 256));
    /* samples between width + 2*margin and stride are written by no kernel: keep them defined */
    HIPCHK(c, hipMemsetAsync(d, 0, plane_bytes(c, height, my, stride) + 256, c->stream));
    out->origin = (char*)d + ((size_t)my * stride + mx) * bpp;
    out->stride = stride; out->width = width; out->height = height; out->margin_x = mx; out->margin_y = my;
    c->plane_base[out->origin] = d;
    return X265AMD_OK;
}

int x265amd_plane_upload(x265amd_ctx *c, const void *buf, int width, int height, int mx, int my, int stride,
                         x265amd_plane *out)
{
    if (!c || !buf || !out || width <= 0 || height <= 0 || mx < 0 || my < 0 || stride < width + 2 * mx)
        return fail(c, X265AMD_EINVAL, "bad plane geometry");
    int rc = plane_alloc(c, width, height, mx, my, stride, out);
    if (rc) return rc;
    const int bpp = c->depth > 8 ? 2 : 1;
    void *base = (char*)out->origin - ((size_t)my * stride + mx) * bpp;
    HIPCHK(c, hipMemcpyAsync(base, buf, plane_bytes(c, height, my, stride), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return X265AMD_OK;
}

int x265amd_plane_free(x265amd_ctx *c, x265amd_plane *p)
{
    if (!c || !p || !p->origin) return fail(c, X265AMD_EINVAL, "null plane");
    /* the allocation is looked up, not derived from the geometry: the int16 plane of x265amd_rk_ds_plane has
     * another sample size than the context's pictures */
    std::map<const void*, void*>::iterator it = c->plane_base.find(p->origin);
    if (it == c->plane_base.end()) return fail(c, X265AMD_EINVAL, "plane was not made by this context (or was freed already)");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(it->second));
    c->plane_base.erase(it);
    p->origin = 0;
    return X265AMD_OK;
}

int x265amd_plane_download(x265amd_ctx *c, const x265amd_plane *p, void *buf)
{
    if (!c || !p || !p->origin || !buf) return fail(c, X265AMD_EINVAL, "null plane");
    const int bpp = c->depth > 8 ? 2 : 1;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(buf, (char*)p->origin - ((size_t)p->margin_y * p->stride + p->margin_x) * bpp,
                        plane_bytes(c, p->height, p->margin_y, p->stride), hipMemcpyDeviceToHost));
    return X265AMD_OK;
}

int x265amd_download(x265amd_ctx *c, const void *d_src, void *dst, size_t bytes)
{
    if (!c || !d_src || !dst) return fail(c, X265AMD_EINVAL, "null argument");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return X265AMD_OK;
}

int x265amd_lowres_init(x265amd_ctx *c, const x265amd_plane *src, x265amd_plane out[4])
{
    if (!c || !src || !src->origin || !out) return fail(c, X265AMD_EINVAL, "null plane");
    /* lowres.cpp:33-44: half size rounded up to whole 8x8 CUs; stride rounded to 32 */
    int w = src->width / 2, h = src->height / 2;
    w = ((w + 7) >> 3) * 8; h = ((h + 7) >> 3) * 8;
    if (2 * w + 2 > src->width + src->margin_x || 2 * h + 2 > src->height + src->margin_y)
        return fail(c, X265AMD_EINVAL, "source margins too small for the lowres filter footprint");
    int stride = w + 2 * src->margin_x;
    if (stride & 31) stride += 32 - (stride & 31);
    for (int i = 0; i < 4; i++)
    {
        int rc = plane_alloc(c, w, h, src->margin_x, src->margin_y, stride, &out[i]);
        if (rc) return rc;
    }
    dim3 blk(64, 4), grd((w + 63) / 64, (h + 3) / 4);
    dim3 grd2((w + 2 * src->margin_x + 63) / 64, (h + 2 * src->margin_y + 3) / 4);
    if (c->depth <= 8)
    {
        hipLaunchKernelGGL(lowres_core_kernel<uint8_t>, grd, blk, 0, c->stream, (const uint8_t*)src->origin,
                           (uint8_t*)out[0].origin, (uint8_t*)out[1].origin, (uint8_t*)out[2].origin, (uint8_t*)out[3].origin,
                           (long)src->stride, (long)stride, w, h);
        for (int i = 0; i < 4; i++)
            hipLaunchKernelGGL(extend_border_kernel<uint8_t>, grd2, blk, 0, c->stream, (uint8_t*)out[i].origin, (long)stride,
                               w, h, src->margin_x, src->margin_y);
    }
    else
    {
        hipLaunchKernelGGL(lowres_core_kernel<uint16_t>, grd, blk, 0, c->stream, (const uint16_t*)src->origin,
                           (uint16_t*)out[0].origin, (uint16_t*)out[1].origin, (uint16_t*)out[2].origin, (uint16_t*)out[3].origin,
                           (long)src->stride, (long)stride, w, h);
        for (int i = 0; i < 4; i++)
            hipLaunchKernelGGL(extend_border_kernel<uint16_t>, grd2, blk, 0, c->stream, (uint16_t*)out[i].origin, (long)stride,
                               w, h, src->margin_x, src->margin_y);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return X265AMD_OK;
}

/* Every address a job can touch must lie inside the padded planes: candidates in
 * [mvmin-3, mvmax+3] (hexagon / square steps are unchecked), the 8-tap footprint of
 * the sub-pel refinement around them, and MV (0,0). */
/* what a search needs to know of a plane to check a job against it */
struct PlaneGeo
{
    int width, height, margin_x, margin_y;
};
static PlaneGeo geo_of(const x265amd_plane *p)
{
    PlaneGeo g = { p->width, p->height, p->margin_x, p->margin_y };
    return g;
}

/* the luma partitions motionEstimate searches (primitives.h:65-73 without 4x4, motion.cpp:119): an edge of 8 .. 64
 * and the other edge the same, a half, or -- from 16 up -- a quarter or three quarters of it */
__host__ __device__ static inline bool searchable_partition(int w, int h)
{
    const int m = w > h ? w : h, n = w > h ? h : w;
    if (!(m == 8 || m == 16 || m == 32 || m == 64)) return false;
    if (n == m || n * 2 == m) return true;
    return m > 8 && (n * 4 == m * 3 || n * 4 == m);
}

/* One job against its planes: 0, or why a search must not run it. The same function checks host job arrays before a
 * launch and device-resident ones in check_jobs_kernel. */
enum { JOB_OK = 0, JOB_PARTITION, JOB_LOWRES_SIZE, JOB_NUMC, JOB_RANGE, JOB_SOURCE, JOB_WINDOW, JOB_PAIR };
__host__ __device__ static inline int job_fault(const x265amd_job &j, const PlaneGeo &f, const PlaneGeo &r, int is_lowres, int method)
{
    if (!searchable_partition(j.w, j.h)) return JOB_PARTITION;
    if (is_lowres && !(j.w == 8 && j.h == 8)) return JOB_LOWRES_SIZE;
    if (j.numc > 4) return JOB_NUMC;
    if (j.mvmin[0] > j.mvmax[0] || j.mvmin[1] > j.mvmax[1]) return JOB_RANGE;
    if (j.pu_x < 0 || j.pu_y < 0 || j.pu_x + j.w > f.width + f.margin_x || j.pu_y + j.h > f.height + f.margin_y) return JOB_SOURCE;
    const int apron = 8;
    const int lox = j.mvmin[0] < 0 ? j.mvmin[0] : 0, hix = j.mvmax[0] > 0 ? j.mvmax[0] : 0;
    const int loy = j.mvmin[1] < 0 ? j.mvmin[1] : 0, hiy = j.mvmax[1] > 0 ? j.mvmax[1] : 0;
    int x0 = j.pu_x + lox - apron, x1 = j.pu_x + j.w + hix + apron;
    int y0 = j.pu_y + loy - apron, y1 = j.pu_y + j.h + hiy + apron;
    if (method == X265AMD_UMH)
    {
        /* UMH measures its cross / diamond groups relative to the block it started from and
         * its vertical cross at x = omv.x + omv.y (motion.cpp:732, :301-303), up to the
         * rescaled range (x 6/4) away */
        const int b = ((j.merange * 6) >> 2) + 6;
        x0 = j.pu_x + lox + loy - b - apron; x1 = j.pu_x + j.w + hix + hiy + b + apron;
        y0 = j.pu_y + loy - b - apron; y1 = j.pu_y + j.h + hiy + b + apron;
    }
    if (x0 < -r.margin_x || y0 < -r.margin_y || x1 > r.width + r.margin_x || y1 > r.height + r.margin_y) return JOB_WINDOW;
    return JOB_OK;
}
static const char *job_fault_text(int code)
{
    static const char *const text[] = { "ok", "not a searchable luma partition", "lowres searches are 8x8 only", "numc > 4",
                                        "empty search range", "PU outside the source plane",
                                        "search window leaves the padded reference plane", "pair index beyond the table" };
    return text[code];
}

static int validate_jobs(x265amd_ctx *c, const x265amd_plane *fenc, const x265amd_plane *ref, int is_lowres,
                         int method, int subme, const x265amd_job *jobs, int njobs)
{
    if (method < 0 || method > X265AMD_FULL || subme < 0 || subme > 7) return fail(c, X265AMD_EINVAL, "bad method / subme");
    const PlaneGeo f = geo_of(fenc), r = geo_of(ref);
    for (int i = 0; i < njobs; i++)
    {
        const int code = job_fault(jobs[i], f, r, is_lowres, method);
        if (code) return fail(c, X265AMD_EINVAL, std::string("job: ") + job_fault_text(code));
    }
    return 0;
}

/* device-resident jobs: out[0] = number of jobs that must not run, out[1] = the first of them (preset to njobs).
 * `geo`: per pair the source and the reference plane (job.reserved picks the pair), or one pair for all jobs. */
__global__ __launch_bounds__(256) void check_jobs_kernel(const x265amd_job *jobs, int njobs, const PlaneGeo *geo, int npairs,
                                                          int by_pair, int is_lowres, int method, int *out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= njobs) return;
    const x265amd_job j = jobs[i];
    const int pi = by_pair ? (int)j.reserved : 0;
    const int code = pi >= npairs ? JOB_PAIR : job_fault(j, geo[2 * pi], geo[2 * pi + 1], is_lowres, method);
    if (code)
    {
        atomicAdd(out, 1);
        atomicMin(out + 1, i);
    }
}

static void fill_args(x265amd_ctx *c, SearchArgs &a, const x265amd_plane *fenc, const x265amd_plane *ref, int is_lowres,
                      int qp, int method, int subme)
{
    a.fenc.origin = fenc->origin; a.fenc.stride = fenc->stride;
    a.nref = is_lowres ? 4 : 1;
    for (int i = 0; i < 4; i++)
    {
        const x265amd_plane *r = &ref[i < a.nref ? i : 0];
        a.ref[i].origin = r->origin; a.ref[i].stride = r->stride;
    }
    a.lowres = is_lowres; a.depth = c->depth; a.method = method; a.subme = subme; a.max_dim = is_lowres ? 8 : 64;
    a.staged = 0; a.staged2 = 0; a.done = 0; a.redo = 0; a.pairs = 0; a.npairs = 0; a.plane_lds = 0; a.state = 0; a.mark = 0; a.list = 0; a.stage_all = 0; a.tiny = 0; a.rect = 0; a.pjobs = 0; a.presults = 0; a.npjobs = 0; a.pmark = 0; a.pstate = 0; a.handled = 0; a.cjobs = 0; a.cresults = 0; a.ncjobs = 0; a.cmark = 0; a.cdone = 0;
    a.ctab = c->tables[qp] + 32768;
}

/* The window a launch gets when the caller set none: 36 KB keeps four workgroups on a compute unit, which is what
 * the 8-bit scans and the small 16-bit PUs want; the rows of a large 16-bit PU are so long that fewer, larger strips
 * win (measured on 4K, merange 92). */
static int esa_window_for(const x265amd_ctx *c, int max_dim)
{
    if (c->esa_window_bytes) return c->esa_window_bytes;
    if (c->depth <= 8) return ESA_WINDOW_DEFAULT;
    return max_dim > 32 ? 65536 : max_dim > 16 ? 49152 : ESA_WINDOW_DEFAULT;
}

int x265amd_set_esa_window_bytes(x265amd_ctx *c, int bytes)
{
    /* the smallest window still holds one lane patch of a 64x64 PU: 71 rows of 72 bytes (8-bit), of 136 bytes (16-bit) */
    if (!c || bytes < (c->depth <= 8 ? ESA_WINDOW_MIN : 2 * ESA_WINDOW_MIN) || bytes > ESA_WINDOW_MAX)
        return fail(c, X265AMD_EINVAL, "window bytes out of range");
    c->esa_window_bytes = bytes & ~15;
    return X265AMD_OK;
}

static int search_dev(x265amd_ctx *c, const x265amd_plane *fenc, const x265amd_plane *ref, int is_lowres,
                      int qp, int method, int subme, const x265amd_job *d_jobs, int njobs, x265amd_result *d_results,
                      int max_pu_dim, int esa_window, const PairDev *d_pairs = 0, int npairs = 0)
{
    if (!c || !d_jobs || !d_results || njobs < 0 || (!d_pairs && (!fenc || !ref))) return fail(c, X265AMD_EINVAL, "null argument");
    if (method < 0 || method > X265AMD_FULL || subme < 0 || subme > 7)
        return fail(c, X265AMD_EINVAL, "bad method / subme");
    const bool square_only = (max_pu_dim & X265AMD_SQUARE_ONLY) != 0;
    max_pu_dim &= ~X265AMD_SQUARE_ONLY;
    if (max_pu_dim == 0) max_pu_dim = 64;
    if (!(max_pu_dim == 8 || max_pu_dim == 16 || max_pu_dim == 32 || max_pu_dim == 64))
        return fail(c, X265AMD_EINVAL, "max_pu_dim must be 0, 8, 16, 32 or 64 (+ X265AMD_SQUARE_ONLY)");
    int rc = x265amd_set_qp(c, qp);
    if (rc) return rc;
    if (!njobs) return X265AMD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    SearchArgs a;
    x265amd_plane none;
    memset(&none, 0, sizeof(none));
    fill_args(c, a, d_pairs ? &none : fenc, d_pairs ? &none : ref, is_lowres, qp, method, subme);
    a.pairs = d_pairs; a.npairs = npairs;
    a.jobs = d_jobs; a.results = d_results; a.njobs = njobs;
    if (!is_lowres) a.max_dim = max_pu_dim;
    if (method == X265AMD_FULL && !is_lowres)
        return c->depth <= 8 ? launch_esa<uint8_t>(c, a, esa_window, c->table_nowrap[qp])
                             : launch_esa<uint16_t>(c, a, esa_window, c->table_nowrap[qp]);
    return c->depth <= 8 ? launch_search<uint8_t>(c, a, is_lowres, square_only) : launch_search<uint16_t>(c, a, is_lowres, square_only);
}

int x265amd_search_batch_dev(x265amd_ctx *c, const x265amd_plane *fenc, const x265amd_plane *ref, int is_lowres,
                             int qp, int method, int subme, const x265amd_job *d_jobs, int njobs, x265amd_result *d_results,
                             int max_pu_dim)
{
    const int dim = max_pu_dim & ~X265AMD_SQUARE_ONLY;
    return search_dev(c, fenc, ref, is_lowres, qp, method, subme, d_jobs, njobs, d_results, max_pu_dim,
                      c ? esa_window_for(c, dim > 0 ? dim : 64) : 0);
}

/* ---- searches over several (source, reference) pairs in one launch ---- */

static int pairs_check(x265amd_ctx *c, const x265amd_pair *pairs, int npairs, std::vector<PairDev> &hp)
{
    if (!c || !pairs || npairs <= 0 || npairs > 256) return fail(c, X265AMD_EINVAL, "1 .. 256 pairs");
    hp.resize(npairs);
    for (int i = 0; i < npairs; i++)
    {
        if (!pairs[i].fenc.origin || !pairs[i].ref.origin) return fail(c, X265AMD_EINVAL, "pair: null plane");
        hp[i].fenc.origin = pairs[i].fenc.origin; hp[i].fenc.stride = pairs[i].fenc.stride;
        hp[i].ref.origin = pairs[i].ref.origin; hp[i].ref.stride = pairs[i].ref.stride;
    }
    return X265AMD_OK;
}

int x265amd_pairs_upload(x265amd_ctx *c, const x265amd_pair *pairs, int npairs, void **d_pairs)
{
    if (!d_pairs) return fail(c, X265AMD_EINVAL, "null argument");
    std::vector<PairDev> hp;
    int rc = pairs_check(c, pairs, npairs, hp);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(d_pairs, npairs * sizeof(PairDev)));
    HIPCHK(c, hipMemcpy(*d_pairs, hp.data(), npairs * sizeof(PairDev), hipMemcpyHostToDevice));
    return X265AMD_OK;
}

int x265amd_search_batch_pairs_family_dev(x265amd_ctx *c, const void *d_pairs, int npairs, int qp, int subme,
                                          const x265amd_job *d_parents, int nparents, x265amd_result *d_parent_results,
                                          const x265amd_job *d_children, int nchildren, x265amd_result *d_child_results)
{
    return x265amd_search_batch_pairs_family_method_dev(c, d_pairs, npairs, qp, X265AMD_STAR, subme, d_parents, nparents, d_parent_results,
                                                        d_children, nchildren, d_child_results);
}

int x265amd_search_batch_pairs_family_method_dev(x265amd_ctx *c, const void *d_pairs, int npairs, int qp, int method, int subme,
                                                 const x265amd_job *d_parents, int nparents, x265amd_result *d_parent_results,
                                                 const x265amd_job *d_children, int nchildren, x265amd_result *d_child_results)
{
    if (!c || !d_pairs || npairs <= 0 || npairs > 256) return fail(c, X265AMD_EINVAL, "1 .. 256 pairs");
    if (method != X265AMD_STAR && method != X265AMD_HEX && method != X265AMD_DIA)
        return fail(c, X265AMD_EINVAL, "family launch: X265_STAR_SEARCH, X265_HEX_SEARCH or X265_DIA_SEARCH");
    if (!d_parents || !d_parent_results || !d_children || !d_child_results || nparents <= 0 || nchildren <= 0)
        return fail(c, X265AMD_EINVAL, "null argument");
    if (subme < 0 || subme > 7) return fail(c, X265AMD_EINVAL, "bad subme");
    if (c->depth > 10) return fail(c, X265AMD_EINVAL, "family launch: samples of at most 10 bits");
    int rc = x265amd_set_qp(c, qp);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    SearchArgs pa, ch;
    x265amd_plane none;
    memset(&none, 0, sizeof(none));
    fill_args(c, pa, &none, &none, 0, qp, method, subme);
    pa.pairs = (const PairDev*)d_pairs; pa.npairs = npairs;
    ch = pa;
    pa.jobs = d_parents; pa.results = d_parent_results; pa.njobs = nparents; pa.max_dim = 16;
    ch.jobs = d_children; ch.results = d_child_results; ch.njobs = nchildren; ch.max_dim = 8;
    return c->depth <= 8 ? launch_family<uint8_t>(c, pa, ch) : launch_family<uint16_t>(c, pa, ch);
}

int x265amd_search_batch_pairs_dev(x265amd_ctx *c, const void *d_pairs, int npairs, int qp, int method, int subme,
                                   const x265amd_job *d_jobs, int njobs, x265amd_result *d_results, int max_pu_dim)
{
    if (!c || !d_pairs || npairs <= 0 || npairs > 256) return fail(c, X265AMD_EINVAL, "1 .. 256 pairs");
    const int dim = max_pu_dim & ~X265AMD_SQUARE_ONLY;
    return search_dev(c, 0, 0, 0, qp, method, subme, d_jobs, njobs, d_results, max_pu_dim, esa_window_for(c, dim > 0 ? dim : 64),
                      (const PairDev*)d_pairs, npairs);
}

/* the jobs of a device-resident list against their planes, before a *_dev entry point runs them unchecked */
static int check_jobs_dev(x265amd_ctx *c, const std::vector<PlaneGeo> &geo, int by_pair, int is_lowres, int method,
                          const x265amd_job *d_jobs, int njobs, int *n_bad, int *first_bad)
{
    if (!c || !d_jobs || njobs < 0) return fail(c, X265AMD_EINVAL, "null argument");
    if (method < 0 || method > X265AMD_FULL) return fail(c, X265AMD_EINVAL, "bad method");
    if (n_bad) *n_bad = 0;
    if (first_bad) *first_bad = -1;
    if (!njobs) return X265AMD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t gbytes = geo.size() * sizeof(PlaneGeo);
    if (ensure(c, &c->d_stage, &c->stage_cap, 16 + gbytes)) return X265AMD_EHIP;
    int out[2] = { 0, njobs };
    HIPCHK(c, hipMemcpyAsync(c->d_stage, out, sizeof(out), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync((char*)c->d_stage + 16, geo.data(), gbytes, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(check_jobs_kernel, dim3((njobs + 255) / 256), dim3(256), 0, c->stream, d_jobs, njobs,
                       (const PlaneGeo*)((char*)c->d_stage + 16), (int)(geo.size() / 2), by_pair, is_lowres, method, (int*)c->d_stage);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, c->d_stage, sizeof(out), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (n_bad) *n_bad = out[0];
    if (!out[0]) return X265AMD_OK;
    if (first_bad) *first_bad = out[1];
    x265amd_job j;
    HIPCHK(c, hipMemcpy(&j, d_jobs + out[1], sizeof(j), hipMemcpyDeviceToHost));
    const int pi = by_pair ? (int)j.reserved : 0;
    const int code = pi >= (int)(geo.size() / 2) ? JOB_PAIR : job_fault(j, geo[2 * pi], geo[2 * pi + 1], is_lowres, method);
    return fail(c, X265AMD_EINVAL, std::to_string(out[0]) + " job(s) must not run; the first is job " + std::to_string(out[1]) + ": " +
                job_fault_text(code));
}

int x265amd_check_jobs_dev(x265amd_ctx *c, const x265amd_plane *fenc, const x265amd_plane *ref, int is_lowres, int method,
                           const x265amd_job *d_jobs, int njobs, int *n_bad, int *first_bad)
{
    if (!c || !fenc || !ref) return fail(c, X265AMD_EINVAL, "null argument");
    std::vector<PlaneGeo> geo;
    geo.push_back(geo_of(fenc));
    geo.push_back(geo_of(ref));
    return check_jobs_dev(c, geo, 0, is_lowres, method, d_jobs, njobs, n_bad, first_bad);
}

int x265amd_check_jobs_pairs_dev(x265amd_ctx *c, const x265amd_pair *pairs, int npairs, int method,
                                 const x265amd_job *d_jobs, int njobs, int *n_bad, int *first_bad)
{
    if (!c || !pairs || npairs <= 0 || npairs > 256) return fail(c, X265AMD_EINVAL, "1 .. 256 pairs");
    std::vector<PlaneGeo> geo;
    for (int i = 0; i < npairs; i++)
    {
        geo.push_back(geo_of(&pairs[i].fenc));
        geo.push_back(geo_of(&pairs[i].ref));
    }
    return check_jobs_dev(c, geo, 1, 0, method, d_jobs, njobs, n_bad, first_bad);
}

int x265amd_search_batch_pairs(x265amd_ctx *c, const x265amd_pair *pairs, int npairs, int qp, int method, int subme,
                               const x265amd_job *jobs, int njobs, x265amd_result *results)
{
    if (!c || !jobs || !results || njobs < 0) return fail(c, X265AMD_EINVAL, "null argument");
    std::vector<PairDev> hp;
    int rc = pairs_check(c, pairs, npairs, hp);
    if (rc) return rc;
    for (int i = 0; i < njobs; i++)
    {
        if (jobs[i].reserved >= npairs) return fail(c, X265AMD_EINVAL, "job: pair index (reserved) beyond the pair table");
        if ((rc = validate_jobs(c, &pairs[jobs[i].reserved].fenc, &pairs[jobs[i].reserved].ref, 0, method, subme, jobs + i, 1))) return rc;
    }
    if (!njobs) return X265AMD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t jb = ((size_t)njobs * sizeof(x265amd_job) + 255) & ~(size_t)255;
    if ((rc = ensure(c, &c->d_jobs, &c->jobs_cap, jb + npairs * sizeof(PairDev)))) return rc;
    if ((rc = ensure(c, &c->d_results, &c->results_cap, (size_t)njobs * sizeof(x265amd_result)))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_jobs, jobs, (size_t)njobs * sizeof(x265amd_job), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync((char*)c->d_jobs + jb, hp.data(), npairs * sizeof(PairDev), hipMemcpyHostToDevice, c->stream));
    int dim = 8;
    for (int i = 0; i < njobs; i++)
    {
        const int d = jobs[i].w > jobs[i].h ? jobs[i].w : jobs[i].h;
        while (dim < d) dim <<= 1;
    }
    rc = search_dev(c, 0, 0, 0, qp, method, subme, (const x265amd_job*)c->d_jobs, njobs, (x265amd_result*)c->d_results, dim,
                    esa_window_for(c, dim), (const PairDev*)((char*)c->d_jobs + jb), npairs);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(results, c->d_results, (size_t)njobs * sizeof(x265amd_result), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return X265AMD_OK;
}

int x265amd_sync(x265amd_ctx *c)
{
    if (!c) return X265AMD_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return X265AMD_OK;
}

int x265amd_marked_count(x265amd_ctx *c, int *marked)
{
    if (!c || !marked) return fail(c, X265AMD_EINVAL, "null argument");
    *marked = 0;
    if (!c->redo_valid) return X265AMD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(marked, c->bufs[0].d_redo, sizeof(int), hipMemcpyDeviceToHost));
    return X265AMD_OK;
}

int x265amd_search_batch(x265amd_ctx *c, const x265amd_plane *fenc, const x265amd_plane *ref, int is_lowres,
                         int qp, int method, int subme, const x265amd_job *jobs, int njobs, x265amd_result *results)
{
    if (!c || !fenc || !ref || !jobs || !results || njobs < 0) return fail(c, X265AMD_EINVAL, "null argument");
    int rc = validate_jobs(c, fenc, ref, is_lowres, method, subme, jobs, njobs);
    if (rc) return rc;
    if (!njobs) return X265AMD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = ensure(c, &c->d_jobs, &c->jobs_cap, (size_t)njobs * sizeof(x265amd_job)))) return rc;
    if ((rc = ensure(c, &c->d_results, &c->results_cap, (size_t)njobs * sizeof(x265amd_result)))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_jobs, jobs, (size_t)njobs * sizeof(x265amd_job), hipMemcpyHostToDevice, c->stream));
    /* LDS per PU is sized for the largest PU of the batch */
    int dim = 8;
    for (int i = 0; i < njobs; i++)
    {
        const int d = jobs[i].w > jobs[i].h ? jobs[i].w : jobs[i].h;
        while (dim < d) dim <<= 1;
    }
    /* does the list hold only what the fixed-shape kernels take (squares of the largest size; with 64 also of 32)? */
    bool squares = true;
    for (int i = 0; i < njobs && squares; i++)
        squares = jobs[i].w == jobs[i].h && (jobs[i].w == dim || (dim == 64 && jobs[i].w == 32));
    /* exhaustive search: no more window LDS than the largest range of the batch needs */
    int window = c->depth <= 8 ? ESA_WINDOW_MIN : 2 * ESA_WINDOW_MIN;
    if (method == X265AMD_FULL)
        for (int i = 0; i < njobs; i++)
        {
            const int cx = c->depth <= 8 ? 4 : 2, bps = c->depth <= 8 ? 1 : 2;
            const int nxg = (jobs[i].mvmax[0] - jobs[i].mvmin[0] + cx) / cx;
            const int nyg = (jobs[i].mvmax[1] - jobs[i].mvmin[1] + ESA_R) / ESA_R;
            const long need = esa_strip_bytes(nxg, nyg, jobs[i].w, jobs[i].h, cx, bps);
            if (need > window) window = need > esa_window_for(c, dim) ? esa_window_for(c, dim) : (int)((need + 15) & ~15);
        }
    rc = search_dev(c, fenc, ref, is_lowres, qp, method, subme, (const x265amd_job*)c->d_jobs, njobs,
                    (x265amd_result*)c->d_results, dim | (squares ? X265AMD_SQUARE_ONLY : 0), window);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(results, c->d_results, (size_t)njobs * sizeof(x265amd_result), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return X265AMD_OK;
}

/* ---- exhaustive search of whole CTUs ---- */

/* the launch for a checked or device-resident CTU list: the shared scan where the cost table's entries never wrap when
 * added, the per-size scans otherwise -- same results either way */
static int esa_ctu_dev(x265amd_ctx *c, const x265amd_plane *fenc, const x265amd_plane *ref, int qp, int subme,
                       const x265amd_job *d_jobs, int nctu, x265amd_result *d_results, const PairDev *d_pairs = 0, int npairs = 0)
{
    if (!c || (!d_pairs && (!fenc || !ref)) || !d_jobs || !d_results || nctu < 0) return fail(c, X265AMD_EINVAL, "null argument");
    if (subme < 0 || subme > 7) return fail(c, X265AMD_EINVAL, "bad subme");
    if (nctu > (1 << 24) / ECTU_PUS) return fail(c, X265AMD_EINVAL, "too many CTUs in one launch");
    int rc = x265amd_set_qp(c, qp);
    if (rc) return rc;
    if (!nctu) return X265AMD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    SearchArgs a;
    x265amd_plane none;
    memset(&none, 0, sizeof(none));
    fill_args(c, a, d_pairs ? &none : fenc, d_pairs ? &none : ref, 0, qp, X265AMD_FULL, subme);
    a.pairs = d_pairs; a.npairs = npairs;
    a.jobs = d_jobs; a.results = d_results; a.njobs = ECTU_PUS * nctu;
    static const bool no_share = getenv("X265AMD_NO_ESA_CTU") != 0;      /* measurements only */
    if (c->table_nowrap[qp] && !no_share) return c->depth <= 8 ? launch_esa_ctu<uint8_t>(c, a, nctu) : launch_esa_ctu<uint16_t>(c, a, nctu);
    static const int first[5] = { 0, 64, 80, 84, 85 };
    for (int k = 0; k < 4; k++)
    {
        SearchArgs b = a;
        b.jobs = d_jobs + (size_t)first[k] * nctu; b.results = d_results + (size_t)first[k] * nctu;
        b.njobs = (first[k + 1] - first[k]) * nctu;
        b.max_dim = 8 << k;
        rc = c->depth <= 8 ? launch_esa<uint8_t>(c, b, esa_window_for(c, b.max_dim), c->table_nowrap[qp])
                           : launch_esa<uint16_t>(c, b, esa_window_for(c, b.max_dim), c->table_nowrap[qp]);
        if (rc) return rc;
    }
    return X265AMD_OK;
}

int x265amd_esa_ctu_batch_dev(x265amd_ctx *c, const x265amd_plane *fenc, const x265amd_plane *ref, int qp, int subme,
                              const x265amd_job *d_jobs, int nctu, x265amd_result *d_results)
{
    return esa_ctu_dev(c, fenc, ref, qp, subme, d_jobs, nctu, d_results);
}

int x265amd_esa_ctu_batch_pairs_dev(x265amd_ctx *c, const void *d_pairs, int npairs, int qp, int subme,
                                    const x265amd_job *d_jobs, int nctu, x265amd_result *d_results)
{
    if (!c || !d_pairs || npairs <= 0 || npairs > 256) return fail(c, X265AMD_EINVAL, "1 .. 256 pairs");
    return esa_ctu_dev(c, 0, 0, qp, subme, d_jobs, nctu, d_results, (const PairDev*)d_pairs, npairs);
}

int x265amd_esa_ctu_batch(x265amd_ctx *c, const x265amd_plane *fenc, const x265amd_plane *ref, int qp, int subme,
                          const x265amd_job *jobs, int nctu, x265amd_result *results)
{
    if (!c || !fenc || !ref || !jobs || !results || nctu < 0) return fail(c, X265AMD_EINVAL, "null argument");
    if (nctu > (1 << 24) / ECTU_PUS) return fail(c, X265AMD_EINVAL, "too many CTUs in one launch");
    /* every slot: empty (w = h = 0), or the square of its size at its place in the CTU, a job x265amd_search_batch would take */
    for (int k = 0; k < nctu; k++)
    {
        const x265amd_job &o = jobs[ectu_job(0, k, nctu)];
        int lox = 0x7fffffff, hix = -0x7fffffff, loy = 0x7fffffff, hiy = -0x7fffffff;
        for (int p = 0; p < ECTU_PUS; p++)
        {
            const x265amd_job &j = jobs[ectu_job(p, k, nctu)];
            if (!j.w && !j.h) continue;
            int ox, oy, s;
            ectu_slot(p, ox, oy, s);
            if (j.w != s || j.h != s || j.pu_x != o.pu_x + ox || j.pu_y != o.pu_y + oy)
                return fail(c, X265AMD_EINVAL, "CTU list: a slot holds neither nothing nor the square PU of its place and size");
            int rc = validate_jobs(c, fenc, ref, 0, X265AMD_FULL, subme, &j, 1);
            if (rc) return rc;
            lox = j.mvmin[0] < lox ? j.mvmin[0] : lox; hix = j.mvmax[0] > hix ? j.mvmax[0] : hix;
            loy = j.mvmin[1] < loy ? j.mvmin[1] : loy; hiy = j.mvmax[1] > hiy ? j.mvmax[1] : hiy;
        }
        if (hix >= lox && (hix - lox >= ECTU_MAX_SPAN || hiy - loy >= ECTU_MAX_SPAN))
            return fail(c, X265AMD_EINVAL, "CTU list: the search ranges of a CTU's PUs span more than 1024 positions");
    }
    if (!nctu) return X265AMD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)ECTU_PUS * nctu;
    int rc;
    if ((rc = ensure(c, &c->d_jobs, &c->jobs_cap, n * sizeof(x265amd_job)))) return rc;
    if ((rc = ensure(c, &c->d_results, &c->results_cap, n * sizeof(x265amd_result)))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_jobs, jobs, n * sizeof(x265amd_job), hipMemcpyHostToDevice, c->stream));
    rc = esa_ctu_dev(c, fenc, ref, qp, subme, (const x265amd_job*)c->d_jobs, nctu, (x265amd_result*)c->d_results);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(results, c->d_results, n * sizeof(x265amd_result), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return X265AMD_OK;
}

static void flat_plan(const void *jobs, size_t job_size, int njobs, const std::vector<int> &tiles, std::vector<char> &blob,
                      size_t &first_off, size_t &chunk_off, int &nchunks, int align16 = 0);

int x265amd_subpel_compare_batch(x265amd_ctx *c, const x265amd_plane *fenc, const x265amd_plane *ref, int cmp,
                                 const x265amd_job *jobs, int njobs, x265amd_result *results)
{
    if (!c || !fenc || !ref || !jobs || !results || njobs < 0) return fail(c, X265AMD_EINVAL, "null argument");
    if (!njobs) return X265AMD_OK;
    int rc = x265amd_set_qp(c, 12);
    if (rc) return rc;
    if (!fenc->origin || !ref->origin) return fail(c, X265AMD_EINVAL, "null plane");
    for (int i = 0; i < njobs; i++)
    {
        const x265amd_job &j = jobs[i];
        if (x265amd_partition_from_sizes(j.w, j.h) < 0) return fail(c, X265AMD_EINVAL, "job: not a luma partition");
        if (j.pu_x < -fenc->margin_x || j.pu_y < -fenc->margin_y || j.pu_x + j.w > fenc->width + fenc->margin_x ||
            j.pu_y + j.h > fenc->height + fenc->margin_y)
            return fail(c, X265AMD_EINVAL, "job: PU outside the source plane");
        /* the 8-tap footprint of the prediction, rows read 12 samples wide */
        const int x0 = j.pu_x + (j.qmvp[0] >> 2) - 3, x1 = j.pu_x + j.w + (j.qmvp[0] >> 2) + 5;
        const int y0 = j.pu_y + (j.qmvp[1] >> 2) - 3, y1 = j.pu_y + j.h + (j.qmvp[1] >> 2) + 4;
        if (x0 < -ref->margin_x || y0 < -ref->margin_y || x1 > ref->width + ref->margin_x || y1 > ref->height + ref->margin_y)
            return fail(c, X265AMD_EINVAL, "job: prediction block leaves the padded reference plane");
    }
    HIPCHK(c, hipSetDevice(c->device));
    /* tiles numbered here (the PU sizes are visible), 64 per wave whatever PUs they belong to */
    std::vector<int> tiles(njobs);
    for (int i = 0; i < njobs; i++) tiles[i] = (jobs[i].w >> 2) * (jobs[i].h >> 2);
    std::vector<char> blob;
    SubpelFlatArgs a;
    size_t first_off, chunk_off;
    flat_plan(jobs, sizeof(x265amd_job), njobs, tiles, blob, first_off, chunk_off, a.nchunks);
    if ((rc = ensure(c, &c->d_jobs, &c->jobs_cap, blob.size()))) return rc;
    if ((rc = ensure(c, &c->d_results, &c->results_cap, (size_t)njobs * sizeof(x265amd_result)))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_jobs, blob.data(), blob.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_results, 0, (size_t)njobs * sizeof(x265amd_result), c->stream));
    a.fenc.origin = fenc->origin; a.fenc.stride = fenc->stride;
    a.ref.origin = ref->origin; a.ref.stride = ref->stride;
    a.depth = c->depth; a.njobs = njobs; a.satd = cmp;
    a.jobs = (const x265amd_job*)c->d_jobs; a.results = (x265amd_result*)c->d_results;
    a.first = (const int32_t*)((const char*)c->d_jobs + first_off);
    a.chunk_job = (const int32_t*)((const char*)c->d_jobs + chunk_off);
    if (c->depth <= 8) hipLaunchKernelGGL(subpel_flat_kernel<uint8_t>, dim3(a.nchunks), dim3(64), 0, c->stream, a);
    else hipLaunchKernelGGL(subpel_flat_kernel<uint16_t>, dim3(a.nchunks), dim3(64), 0, c->stream, a);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(results, c->d_results, (size_t)njobs * sizeof(x265amd_result), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return X265AMD_OK;
}

/* Every sample a lookahead search of the wcu x hcu grid can touch must lie inside the padded planes. The search range
 * of a CU reaches 8 samples outside the grid (slicetype.cpp:1427-1430); hexagon / square / diamond steps go up to 3
 * further without a range check (motion.cpp:635, :673-678, :711-721) and the quarter-pel average reads one more sample.
 * UMH measures its cross relative to the block it started from, up to the rescaled range (x 6/4) away, and its vertical
 * cross at x = omv.x + omv.y (motion.cpp:732, :301-303): the same bound validate_jobs applies per job, taken over the
 * corner CUs of the grid. Fills the device descriptors on success. */
static int lookahead_fields_check(x265amd_ctx *c, const x265amd_field *fields, int nfields, int wcu, int hcu, int method,
                                  int merange, FieldDev *hf /* nfields entries */)
{
    if (method < 0 || method > X265AMD_FULL) return fail(c, X265AMD_EINVAL, "bad method");
    if (merange < 0 || merange > 4096) return fail(c, X265AMD_EINVAL, "merange out of range");
    int need_x = 8 + 3 + 1, need_y = 8 + 3 + 1;
    if (method == X265AMD_UMH)
    {
        const int b = ((merange * 6) >> 2) + 6, apron = 8;
        need_x = 16 + (hcu - 1) * 8 + 8 + b + apron;
        need_y = 16 + b + apron;
    }
    for (int i = 0; i < nfields; i++)
    {
        const x265amd_plane &f = fields[i].fenc;
        if (!f.origin || f.width < wcu * 8 || f.height < hcu * 8 || f.stride < f.width)
            return fail(c, X265AMD_EINVAL, "lookahead field: source plane does not cover the CU grid");
        hf[i].fenc.origin = f.origin; hf[i].fenc.stride = f.stride;
        for (int k = 0; k < 4; k++)
        {
            const x265amd_plane &p = fields[i].ref[k];
            if (!p.origin || p.width < wcu * 8 || p.height < hcu * 8)
                return fail(c, X265AMD_EINVAL, "lookahead field: reference plane does not cover the CU grid");
            if (p.stride != fields[i].ref[0].stride)
                return fail(c, X265AMD_EINVAL, "lookahead field: the four half-pel planes of a reference must share one stride");
            if (p.margin_x < need_x || p.margin_y < need_y || p.stride < p.width + 2 * p.margin_x)
                return fail(c, X265AMD_EINVAL, "lookahead field: reference margins too small for this search method "
                                               "(12 samples; UMH: 32 + 8 * (height_in_cu - 1) + 6 * merange / 4 + 6 wide, 30 + 6 * merange / 4 high)");
            hf[i].ref[k].origin = p.origin; hf[i].ref[k].stride = p.stride;
        }
    }
    return X265AMD_OK;
}

int x265amd_lookahead_fields(x265amd_ctx *c, const x265amd_field *fields, int nfields, int wcu, int hcu,
                             int qp, int method, int subme, int merange,
                             int16_t *out_mvs, int32_t *out_costs, x265amd_result *out_stats)
{
    if (!c || !fields || nfields < 0 || wcu <= 0 || hcu <= 0 || !out_mvs || !out_costs)
        return fail(c, X265AMD_EINVAL, "null argument");
    if (method < 0 || method > X265AMD_FULL || subme < 0 || subme > 7)
        return fail(c, X265AMD_EINVAL, "bad method / subme");
    if (!nfields) return X265AMD_OK;
    std::vector<FieldDev> hf(nfields);
    int rc = lookahead_fields_check(c, fields, nfields, wcu, hcu, method, merange, hf.data());
    if (rc) return rc;
    if ((rc = x265amd_set_qp(c, qp))) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t ncu = (size_t)wcu * hcu;
    const size_t fbytes = nfields * sizeof(FieldDev), mvbytes = nfields * ncu * 4, cbytes = nfields * ncu * 4;
    const size_t sbytes = out_stats ? nfields * ncu * sizeof(x265amd_result) : 0;
    if ((rc = ensure(c, &c->d_jobs, &c->jobs_cap, fbytes))) return rc;
    if ((rc = ensure(c, &c->d_results, &c->results_cap, mvbytes + cbytes + sbytes))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_jobs, hf.data(), fbytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_results, 0, mvbytes + cbytes + sbytes, c->stream));
    LookaheadArgs a;
    a.fields = (const FieldDev*)c->d_jobs;
    a.wcu = wcu; a.hcu = hcu; a.depth = c->depth; a.method = method; a.subme = subme; a.merange = merange;
    a.ctab = c->tables[qp] + 32768;
    a.mvs = (int16_t*)c->d_results;
    a.costs = (int32_t*)((char*)c->d_results + mvbytes);
    a.stats = out_stats ? (x265amd_result*)((char*)c->d_results + mvbytes + cbytes) : 0;
    if ((rc = launch_lookahead(c, a, nfields, c->d_ticket))) return rc;
    HIPCHK(c, hipMemcpyAsync(out_mvs, a.mvs, mvbytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(out_costs, a.costs, cbytes, hipMemcpyDeviceToHost, c->stream));
    if (out_stats) HIPCHK(c, hipMemcpyAsync(out_stats, a.stats, sbytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return X265AMD_OK;
}

int x265amd_fields_upload(x265amd_ctx *c, const x265amd_field *fields, int nfields, int wcu, int hcu, int method, int merange,
                          void **d_fields)
{
    if (!c || !fields || nfields <= 0 || wcu <= 0 || hcu <= 0 || !d_fields) return fail(c, X265AMD_EINVAL, "null argument");
    std::vector<FieldDev> hf(nfields);
    int rc = lookahead_fields_check(c, fields, nfields, wcu, hcu, method, merange, hf.data());
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(d_fields, nfields * sizeof(FieldDev)));
    HIPCHK(c, hipMemcpy(*d_fields, hf.data(), nfields * sizeof(FieldDev), hipMemcpyHostToDevice));
    const x265amd_ctx::FieldsInfo info = { nfields, wcu, hcu, method, merange };
    c->uploads[*d_fields] = info;
    return X265AMD_OK;
}

int x265amd_abi_version(void) { return X265AMD_ABI_VERSION; }

int x265amd_device_mem_info(x265amd_ctx *c, size_t *free_bytes, size_t *total_bytes)
{
    if (!c || !free_bytes || !total_bytes) return fail(c, X265AMD_EINVAL, "null argument");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemGetInfo(free_bytes, total_bytes));
    return X265AMD_OK;
}

int x265amd_device_free(x265amd_ctx *c, void *d_ptr)
{
    if (!c) return X265AMD_EINVAL;
    c->uploads.erase(d_ptr);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipFree(d_ptr));
    return X265AMD_OK;
}

int x265amd_lookahead_fields_dev(x265amd_ctx *c, const void *d_fields, int nfields, int wcu, int hcu,
                                 int qp, int method, int subme, int merange,
                                 int16_t *d_mvs, int32_t *d_costs, x265amd_result *d_stats)
{
    if (!c || !d_fields || nfields < 0 || wcu <= 0 || hcu <= 0 || !d_mvs || !d_costs)
        return fail(c, X265AMD_EINVAL, "null argument");
    if (method < 0 || method > X265AMD_FULL || subme < 0 || subme > 7)
        return fail(c, X265AMD_EINVAL, "bad method / subme");
    /* the descriptors were checked against the planes for one grid, method and range: nothing else may run on them */
    {
        std::map<const void*, x265amd_ctx::FieldsInfo>::const_iterator it = c->uploads.upper_bound(d_fields);
        if (it == c->uploads.begin()) return fail(c, X265AMD_EINVAL, "d_fields is no array of x265amd_fields_upload on this context");
        --it;
        const size_t off = (size_t)((const char*)d_fields - (const char*)it->first);
        const x265amd_ctx::FieldsInfo &u = it->second;
        if (off % sizeof(FieldDev) || off / sizeof(FieldDev) + (size_t)nfields > (size_t)u.nfields)
            return fail(c, X265AMD_EINVAL, "d_fields is no array of x265amd_fields_upload on this context, or holds fewer fields");
        if (wcu != u.wcu || hcu != u.hcu || merange > u.merange || (method == X265AMD_UMH && u.method != X265AMD_UMH))
            return fail(c, X265AMD_EINVAL, "d_fields was uploaded for another CU grid, a smaller merange or without the margins of UMH");
    }
    int rc = x265amd_set_qp(c, qp);
    if (rc) return rc;
    if (!nfields) return X265AMD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    LookaheadArgs a;
    a.fields = (const FieldDev*)d_fields;
    a.wcu = wcu; a.hcu = hcu; a.depth = c->depth; a.method = method; a.subme = subme; a.merange = merange;
    a.ctab = c->tables[qp] + 32768;
    a.mvs = d_mvs; a.costs = d_costs; a.stats = d_stats;
    if ((rc = launch_lookahead(c, a, nfields, c->d_ticket))) return rc;
    return X265AMD_OK;
}

/* x265amd_lookahead_fields without a single allocation: descriptors, result arrays and the row ticket live in the
 * static scratch array of the context's device, the cost table must exist already. Calls that overlap on one GPU
 * (two contexts, two threads) race on that array. */
int x265amd_lookahead_fields_static(x265amd_ctx *c, const x265amd_field *fields, int nfields, int wcu, int hcu,
                                    int qp, int method, int subme, int merange,
                                    int16_t *out_mvs, int32_t *out_costs, x265amd_result *out_stats)
{
    if (!c || !fields || nfields < 0 || wcu <= 0 || hcu <= 0 || !out_mvs || !out_costs)
        return fail(c, X265AMD_EINVAL, "null argument");
    if (method < 0 || method > X265AMD_FULL || subme < 0 || subme > 7)
        return fail(c, X265AMD_EINVAL, "bad method / subme");
    if (!c->scratch) return fail(c, X265AMD_EINVAL, "no static scratch array on this device (device index >= 64)");
    if (qp < 0 || qp >= 70 || !c->tables.count(qp))
        return fail(c, X265AMD_EINVAL, "the cost table of this qp does not exist yet: call x265amd_set_qp first (building it allocates)");
    if (!nfields) return X265AMD_OK;
    if (nfields > X265AMD_STATIC_MAX_FIELDS) return fail(c, X265AMD_EINVAL, "more than X265AMD_STATIC_MAX_FIELDS fields in one call");
    const size_t ncu = (size_t)wcu * hcu;
    const size_t fbytes = ((size_t)nfields * sizeof(FieldDev) + 255) & ~(size_t)255;
    const size_t mvbytes = nfields * ncu * 4, cbytes = nfields * ncu * 4;
    const size_t sbytes = out_stats ? nfields * ncu * sizeof(x265amd_result) : 0;
    if (ncu > SCRATCH_BYTES || SCRATCH_HEAD + fbytes + mvbytes + cbytes + sbytes > SCRATCH_BYTES)
        return fail(c, X265AMD_EINVAL, "fields too large for the static scratch array: nfields * width_in_cu * height_in_cu * "
                                       "8 bytes (24 with statistics) + 80 bytes per field must fit X265AMD_STATIC_SCRATCH_BYTES - 256");
    FieldDev hf[X265AMD_STATIC_MAX_FIELDS];
    int rc = lookahead_fields_check(c, fields, nfields, wcu, hcu, method, merange, hf);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    /* one call at a time per GPU: a second one (another context, another thread) is refused, not left to race */
    struct Busy
    {
        std::atomic<int> &f; bool mine;
        Busy(std::atomic<int> &f_) : f(f_) { int z = 0; mine = f.compare_exchange_strong(z, 1); }
        ~Busy() { if (mine) f.store(0); }
    } busy(g_scratch_busy[c->device]);
    if (!busy.mine) return fail(c, X265AMD_EBUSY, "another x265amd_lookahead_fields_static call is using this GPU's static scratch array");
    unsigned char *base = c->scratch;
    LookaheadArgs a;
    a.fields = (const FieldDev*)(base + SCRATCH_HEAD);
    a.wcu = wcu; a.hcu = hcu; a.depth = c->depth; a.method = method; a.subme = subme; a.merange = merange;
    a.ctab = c->tables[qp] + 32768;
    a.mvs = (int16_t*)(base + SCRATCH_HEAD + fbytes);
    a.costs = (int32_t*)(base + SCRATCH_HEAD + fbytes + mvbytes);
    a.stats = out_stats ? (x265amd_result*)(base + SCRATCH_HEAD + fbytes + mvbytes + cbytes) : 0;
    /* the array keeps the last call's contents: the ticket and (row kernel) the MV marks are set again by
     * launch_lookahead, costs and statistics are written for every CU. `hf` is on this stack frame, so no path leaves
     * before the stream has drained. */
    hipError_t e = hipMemcpyAsync(base + SCRATCH_HEAD, hf, (size_t)nfields * sizeof(FieldDev), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
    {
        rc = launch_lookahead(c, a, nfields, (int*)base);
        if (!rc)
        {
            e = hipMemcpyAsync(out_mvs, a.mvs, mvbytes, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(out_costs, a.costs, cbytes, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess && out_stats) e = hipMemcpyAsync(out_stats, a.stats, sbytes, hipMemcpyDeviceToHost, c->stream);
        }
    }
    const hipError_t es = hipStreamSynchronize(c->stream);
    if (rc) return rc;
    HIPCHK(c, e);
    HIPCHK(c, es);
    return X265AMD_OK;
}

/* ---- lookahead frame cost ---- */

static int frame_cost_check(x265amd_ctx *c, const x265amd_frame_cost *fr, int n, int wcu, int hcu, std::vector<FrameCostDev> &out)
{
    out.resize(n);
    for (int i = 0; i < n; i++)
    {
        const x265amd_frame_cost &s = fr[i];
        FrameCostDev &d = out[i];
        if (s.d0 < 0 || s.d1 < 0 || (s.d1 > 0 && s.d0 == 0))
            return fail(c, X265AMD_EINVAL, "frame cost: need d0 >= 0, d1 >= 0 and a list-0 reference for a bidirectional frame");
        if (!s.fenc.origin || !s.lowres_costs || !s.intra_cost || !s.totals || (!s.intra_calculated && !s.row_satds_intra))
            return fail(c, X265AMD_EINVAL, "frame cost: null output array");
        if (!s.intra_calculated && c->depth > 8)
            return fail(c, X265AMD_EINVAL, "frame cost: the intra estimate exists for 8-bit pictures only (pass intra_cost)");
        if (s.fenc.margin_x < 16 || s.fenc.margin_y < 16 || s.fenc.width < wcu * 8 || s.fenc.height < hcu * 8)
            return fail(c, X265AMD_EINVAL, "frame cost: lowres plane geometry does not cover the CU grid + 16 px");
        if (s.d0 && (!s.ref0[0].origin || !s.mvs0 || !s.mv_costs0 || !s.row_satds))
            return fail(c, X265AMD_EINVAL, "frame cost: list 0 needs planes, MVs, MV costs and row_satds");
        if (s.d1 && (!s.ref1[0].origin || !s.mvs1 || !s.mv_costs1))
            return fail(c, X265AMD_EINVAL, "frame cost: list 1 needs planes, MVs and MV costs");
        d.fenc.origin = s.fenc.origin; d.fenc.stride = s.fenc.stride;
        for (int k = 0; k < 4; k++)
        {
            d.ref0[k].origin = s.ref0[k].origin; d.ref0[k].stride = s.ref0[k].stride;
            d.ref1[k].origin = s.ref1[k].origin; d.ref1[k].stride = s.ref1[k].stride;
        }
        d.d0 = s.d0; d.d1 = s.d1; d.intra_calculated = s.intra_calculated; d.pad = 0;
        d.mvs0 = s.mvs0; d.mv_costs0 = s.mv_costs0; d.mvs1 = s.mvs1; d.mv_costs1 = s.mv_costs1;
        d.inv_qscale = s.inv_qscale; d.intra_cost = s.intra_cost; d.lowres_costs = s.lowres_costs;
        d.row_satds_intra = s.row_satds_intra; d.row_satds = s.row_satds; d.totals = (long long*)s.totals;
    }
    return X265AMD_OK;
}

int x265amd_frame_costs_upload(x265amd_ctx *c, const x265amd_frame_cost *frames, int nframes, int wcu, int hcu, void **d_frames)
{
    if (!c || !frames || nframes <= 0 || wcu <= 0 || hcu <= 0 || !d_frames) return fail(c, X265AMD_EINVAL, "null argument");
    std::vector<FrameCostDev> hf;
    int rc = frame_cost_check(c, frames, nframes, wcu, hcu, hf);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(d_frames, nframes * sizeof(FrameCostDev)));
    HIPCHK(c, hipMemcpy(*d_frames, hf.data(), nframes * sizeof(FrameCostDev), hipMemcpyHostToDevice));
    return X265AMD_OK;
}

int x265amd_lookahead_frame_costs_dev(x265amd_ctx *c, const void *d_frames, int nframes, int wcu, int hcu)
{
    if (!c || !d_frames || nframes < 0 || wcu <= 0 || hcu <= 0) return fail(c, X265AMD_EINVAL, "null argument");
    if (!nframes) return X265AMD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const FrameCostDev *fr = (const FrameCostDev*)d_frames;
    hipLaunchKernelGGL(la_cost_init_kernel, dim3(nframes), dim3(64), 0, c->stream, fr);
    if (c->depth <= 8)
    {
        hipLaunchKernelGGL(la_intra_kernel, dim3(hcu, nframes), dim3(LA_INTRA_WAVES * 64), 0, c->stream, fr, wcu, hcu);
        hipLaunchKernelGGL(la_cost_kernel<uint8_t>, dim3(hcu, nframes), dim3(LA_COST_THREADS), 0, c->stream, fr, wcu, hcu);
    }
    else
        hipLaunchKernelGGL(la_cost_kernel<uint16_t>, dim3(hcu, nframes), dim3(LA_COST_THREADS), 0, c->stream, fr, wcu, hcu);
    HIPCHK(c, hipGetLastError());
    return X265AMD_OK;
}

int x265amd_device_alloc(x265amd_ctx *c, size_t bytes, void **d_ptr)
{
    if (!c || !d_ptr || !bytes) return fail(c, X265AMD_EINVAL, "null argument");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(d_ptr, bytes));
    HIPCHK(c, hipMemsetAsync(*d_ptr, 0, bytes, c->stream));
    return X265AMD_OK;
}

int x265amd_upload(x265amd_ctx *c, void *d_dst, const void *src, size_t bytes)
{
    if (!c || !d_dst || !src) return fail(c, X265AMD_EINVAL, "null argument");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return X265AMD_OK;
}

/* One wave that sleeps for `ticks` of the constant-rate clock and counts the shader clock meanwhile: what the chip
 * clocks at while other streams keep it busy (the shader clock follows power, so a loaded chip runs below its
 * nominal rate and an issue-rate bound has to be priced at the measured one). */
__global__ __launch_bounds__(64) void clock_probe_kernel(long long *out, long long ticks)
{
    const long long w0 = wall_clock64(), c0 = clock64();
    while (wall_clock64() - w0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0)
    {
        out[0] = clock64() - c0;
        out[1] = wall_clock64() - w0;
    }
}

int x265amd_clock_probe_start(x265amd_ctx *c, double ms)
{
    if (!c || !(ms > 0 && ms < 1000)) return fail(c, X265AMD_EINVAL, "probe time out of range");
    HIPCHK(c, hipSetDevice(c->device));
    int khz = 0;
    HIPCHK(c, hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device));
    if (khz <= 0) return fail(c, X265AMD_EHIP, "no constant-rate clock on this device");
    if (!c->d_probe) HIPCHK(c, hipMalloc((void**)&c->d_probe, 16));
    HIPCHK(c, hipMemsetAsync(c->d_probe, 0, 16, c->stream));
    c->probe_khz = khz;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, c->stream, c->d_probe, (long long)(ms * khz));
    HIPCHK(c, hipGetLastError());
    return X265AMD_OK;
}

int x265amd_clock_probe_read(x265amd_ctx *c, double *shader_mhz)
{
    if (!c || !shader_mhz || !c->probe_khz) return fail(c, X265AMD_EINVAL, "no probe started");
    HIPCHK(c, hipSetDevice(c->device));
    long long out[2] = { 0, 0 };
    HIPCHK(c, hipMemcpyAsync(out, c->d_probe, sizeof(out), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (out[1] <= 0) return fail(c, X265AMD_EHIP, "probe did not run");
    *shader_mhz = (double)out[0] / (double)out[1] * c->probe_khz / 1000.0;
    return X265AMD_OK;
}

/* the 8 bytes of each 16-byte result record that go back to the host encoder */
__global__ __launch_bounds__(256) void pack_mv_cost_kernel(const uint2 *records, uint2 *out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = records[2 * i];
}

__global__ __launch_bounds__(256) void pack_mv_cost_indexed_kernel(const uint2 *records, const int *index, uint2 *out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = records[2 * index[i]];
}

int x265amd_pack_mv_cost_indexed(x265amd_ctx *c, const void *d_results, const int32_t *d_index, int n, void *d_out)
{
    if (!c || n < 0 || (n && (!d_results || !d_index || !d_out))) return fail(c, X265AMD_EINVAL, "null argument");
    if (!n) return X265AMD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(pack_mv_cost_indexed_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, (const uint2*)d_results, d_index, (uint2*)d_out, n);
    HIPCHK(c, hipGetLastError());
    return X265AMD_OK;
}

int x265amd_pack_mv_cost(x265amd_ctx *c, const void *d_results, int n, void *d_out)
{
    if (!c || n < 0 || (n && (!d_results || !d_out))) return fail(c, X265AMD_EINVAL, "null argument");
    if (!n) return X265AMD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(pack_mv_cost_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, (const uint2*)d_results, (uint2*)d_out, n);
    HIPCHK(c, hipGetLastError());
    return X265AMD_OK;
}

/* ---- explicit weighted prediction ---- */

static int weight_args_ok(int w0, int shift)
{
    return w0 >= -128 && w0 <= 255 && shift >= 0 && shift <= 20;
}

/* output plane of a weighting call: the caller's (dst->origin set, same geometry as src: the
 * reference keeps one weight buffer per reference and reuses it) or a fresh one */
static int weight_dst(x265amd_ctx *c, const x265amd_plane *src, x265amd_plane *dst)
{
    if (dst->origin)
    {
        if (dst->stride != src->stride || dst->width != src->width || dst->height != src->height ||
            dst->margin_x != src->margin_x || dst->margin_y != src->margin_y || dst->origin == src->origin)
            return fail(c, X265AMD_EINVAL, "output plane must be another plane of the source's geometry");
        return X265AMD_OK;
    }
    return plane_alloc(c, src->width, src->height, src->margin_x, src->margin_y, src->stride, dst);
}

int x265amd_weight_buffer(x265amd_ctx *c, const x265amd_plane *src, int w0, int round, int shift, int offset, x265amd_plane *dst)
{
    if (!c || !src || !src->origin || !dst) return fail(c, X265AMD_EINVAL, "null plane");
    if (!weight_args_ok(w0, shift)) return fail(c, X265AMD_EINVAL, "weight / shift out of range");
    int rc = weight_dst(c, src, dst);
    if (rc) return rc;
    const int bpp = c->depth > 8 ? 2 : 1;
    const size_t pad = ((size_t)src->margin_y * src->stride + src->margin_x) * bpp;
    const long nvec = (long)((plane_bytes(c, src->height, src->margin_y, src->stride) + 15) / 16);
    const uint4 *s = (const uint4*)((const char*)src->origin - pad);
    if ((uintptr_t)s & 15) return fail(c, X265AMD_EINVAL, "source buffer is not 16-byte aligned (planes made by this library are)");
    uint4 *d = (uint4*)((char*)dst->origin - pad);
    if (c->depth <= 8)
        hipLaunchKernelGGL(weight_buffer_kernel<uint8_t>, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, c->stream, s, d, nvec, w0, round, shift, offset, c->depth);
    else
        hipLaunchKernelGGL(weight_buffer_kernel<uint16_t>, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, c->stream, s, d, nvec, w0, round, shift, offset, c->depth);
    HIPCHK(c, hipGetLastError());
    return X265AMD_OK;
}

int x265amd_apply_weight(x265amd_ctx *c, const x265amd_plane *recon, int input_weight, int log2_denom, int input_offset, x265amd_plane *dst)
{
    if (!c || !recon || !recon->origin || !dst) return fail(c, X265AMD_EINVAL, "null plane");
    /* reference.cpp:60-63, :226-228 */
    const int offset = input_offset * (1 << (c->depth - 8));
    const int shift = log2_denom + (14 - c->depth);
    const int round = shift ? 1 << (shift - 1) : 0;
    if (log2_denom < 0 || !weight_args_ok(input_weight, shift)) return fail(c, X265AMD_EINVAL, "weight / denominator out of range");
    const int padwidth = (recon->width + 15) & ~15;
    if (padwidth - recon->width > recon->margin_x) return fail(c, X265AMD_EINVAL, "margin_x smaller than the 16-column rounding of the width");
    int rc = weight_dst(c, recon, dst);
    if (rc) return rc;
    if (recon->stride < 16) return fail(c, X265AMD_EINVAL, "stride too small");
    const int bpp = c->depth > 8 ? 2 : 1;
    const size_t pad = ((size_t)recon->margin_y * recon->stride + recon->margin_x) * bpp;
    const long nvec = (long)((plane_bytes(c, recon->height, recon->margin_y, recon->stride) + 15) / 16);
    const char *sb = (const char*)recon->origin - pad;
    char *db = (char*)dst->origin - pad;
    if (((uintptr_t)sb | (uintptr_t)db) & 15) return fail(c, X265AMD_EINVAL, "plane buffers are not 16-byte aligned (planes made by this library are)");
    const dim3 grd((unsigned)((nvec + 255) / 256)), blk(256);
    if (c->depth <= 8)
        hipLaunchKernelGGL(apply_weight_kernel<uint8_t>, grd, blk, 0, c->stream, (const uint8_t*)sb, (uint8_t*)db, nvec, recon->stride,
                           recon->width, recon->height, recon->margin_x, recon->margin_y, input_weight, round, shift, offset, c->depth);
    else
        hipLaunchKernelGGL(apply_weight_kernel<uint16_t>, grd, blk, 0, c->stream, (const uint16_t*)sb, (uint16_t*)db, nvec, recon->stride,
                           recon->width, recon->height, recon->margin_x, recon->margin_y, input_weight, round, shift, offset, c->depth);
    HIPCHK(c, hipGetLastError());
    return X265AMD_OK;
}

int x265amd_lookahead_weight_cost(x265amd_ctx *c, const x265amd_plane *fenc, const x265amd_plane *src, const int32_t *d_intra_cost,
                                  int wcu, int hcu, int use_wp, int scale, int denom, int offset, uint32_t *cost)
{
    if (!c || !fenc || !src || !fenc->origin || !src->origin || !d_intra_cost || !cost || wcu <= 0 || hcu <= 0)
        return fail(c, X265AMD_EINVAL, "null argument");
    if (fenc->width < wcu * 8 || fenc->height < hcu * 8 || src->width < wcu * 8 || src->height < hcu * 8)
        return fail(c, X265AMD_EINVAL, "planes smaller than the CU grid");
    /* slicetype.cpp:1230-1237 */
    const int correction = 14 - c->depth;
    const int shift = denom + correction, round = use_wp ? 1 << (denom - 1 + correction) : 0;
    if (use_wp && (denom < 0 || shift < 1 || !weight_args_ok(scale, shift))) return fail(c, X265AMD_EINVAL, "weight / denominator out of range");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(c->d_out, 0, sizeof(uint32_t), c->stream));
    PlaneDev f = { fenc->origin, fenc->stride }, s = { src->origin, src->stride };
    if (c->depth <= 8)
        hipLaunchKernelGGL(la_weight_cost_kernel<uint8_t>, dim3(hcu), dim3(LA_COST_THREADS), 0, c->stream, f, s, d_intra_cost, wcu,
                           use_wp, scale, round, shift, offset << (c->depth - 8), c->depth, (unsigned int*)c->d_out);
    else
        hipLaunchKernelGGL(la_weight_cost_kernel<uint16_t>, dim3(hcu), dim3(LA_COST_THREADS), 0, c->stream, f, s, d_intra_cost, wcu,
                           use_wp, scale, round, shift, offset << (c->depth - 8), c->depth, (unsigned int*)c->d_out);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(cost, c->d_out, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return X265AMD_OK;
}

/* CostEstimate::weightsAnalyse (slicetype.cpp:1258-1340): the float bookkeeping runs here on the
 * host exactly as the reference's, the two cost evaluations and the weighting of the four planes
 * on the device */
int x265amd_weights_analyse(x265amd_ctx *c, const x265amd_plane *fenc, const x265amd_plane ref[4], const int32_t *d_intra_cost,
                            int wcu, int hcu, const uint64_t stats[4], x265amd_weight *out, x265amd_plane weighted[4])
{
    if (!c || !fenc || !ref || !d_intra_cost || !stats || !out || !weighted) return fail(c, X265AMD_EINVAL, "null argument");
    const float epsilon = 1.f / 128.f;
    const int lines = hcu * 8, width = wcu * 8, depth = c->depth;
    memset(out, 0, sizeof(*out));
    out->orig_score = 1;
    float guessScale, fencMean, refMean;
    if (stats[0] && stats[2]) guessScale = sqrtf((float)stats[0] / stats[2]);
    else guessScale = 1.0f;
    fencMean = (float)stats[1] / (lines * width) / (1 << (depth - 8));
    refMean = (float)stats[3] / (lines * width) / (1 << (depth - 8));
    if (fabsf(refMean - fencMean) < 0.5f && fabsf(1.f - guessScale) < epsilon) return X265AMD_OK;

    /* wpScalingParam::setFromWeightAndOffset (TComSlice.h:1274-1286) */
    int mindenom = 7, minscale = (int)(guessScale * 128 + 0.5f), minoff = 0, found = 0;
    while (mindenom > 0 && minscale > 127) { mindenom--; minscale >>= 1; }
    if (minscale > 127) minscale = 127;

    uint32_t minscore = 0, origscore = 1, s = 0;
    int rc = x265amd_lookahead_weight_cost(c, fenc, &ref[0], d_intra_cost, wcu, hcu, 0, 0, 0, 0, &minscore);
    if (rc) return rc;
    origscore = minscore;
    out->orig_score = origscore; out->min_score = minscore;
    if (!minscore) return X265AMD_OK;

    int curScale = minscale;
    int curOffset = (int)(fencMean - refMean * curScale / (1 << mindenom) + 0.5f);
    if (curOffset < -128 || curOffset > 127)
    {
        curOffset = curOffset < -128 ? -128 : 127;
        curScale = (int)((1 << mindenom) * (fencMean - curOffset) / refMean + 0.5f);
        curScale = curScale < 0 ? 0 : (curScale > 127 ? 127 : curScale);
    }
    if ((rc = x265amd_lookahead_weight_cost(c, fenc, &ref[0], d_intra_cost, wcu, hcu, 1, curScale, mindenom, curOffset, &s))) return rc;
    if (s < minscore) { minscore = s; minscale = curScale; minoff = curOffset; found = 1; }
    while (mindenom > 0 && !(minscale & 1)) { mindenom--; minscale >>= 1; }
    out->min_score = minscore;
    if (!found || (minscale == 1 << mindenom && minoff == 0) || (float)minscore / origscore > 0.998f) return X265AMD_OK;

    const int correction = 14 - depth;
    for (int i = 0; i < 4; i++)
        if ((rc = x265amd_weight_buffer(c, &ref[i], minscale, 1 << (mindenom - 1 + correction), mindenom + correction,
                                        minoff << (depth - 8), &weighted[i])))
            return rc;
    out->weighted = 1; out->scale = minscale; out->denom = mindenom; out->offset = minoff;
    return X265AMD_OK;
}

/* ---- bi-prediction estimate ---- */


int x265amd_bipred_batch_dev(x265amd_ctx *c, const x265amd_plane *fenc, const x265amd_plane *ref0, const x265amd_plane *ref1,
                             const x265amd_bipred_job *d_jobs, int njobs, x265amd_bipred_result *d_results)
{
    if (!c || !fenc || !ref0 || !ref1 || !fenc->origin || !ref0->origin || !ref1->origin || !d_jobs || !d_results || njobs < 0)
        return fail(c, X265AMD_EINVAL, "null argument");
    if (!njobs) return X265AMD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    BipredArgs a;
    a.fenc.origin = fenc->origin; a.fenc.stride = fenc->stride;
    a.ref0.origin = ref0->origin; a.ref0.stride = ref0->stride;
    a.ref1.origin = ref1->origin; a.ref1.stride = ref1->stride;
    a.depth = c->depth; a.njobs = njobs; a.jobs = d_jobs; a.results = d_results;
    if (c->depth <= 8) hipLaunchKernelGGL(bipred_kernel<uint8_t>, dim3(njobs), dim3(64), 0, c->stream, a);
    else hipLaunchKernelGGL(bipred_kernel<uint16_t>, dim3(njobs), dim3(64), 0, c->stream, a);
    HIPCHK(c, hipGetLastError());
    return X265AMD_OK;
}

static int bipred_validate(x265amd_ctx *c, const x265amd_plane *fenc, const x265amd_plane *ref0, const x265amd_plane *ref1,
                           const x265amd_bipred_job *jobs, int njobs)
{
    if (!c || !fenc || !ref0 || !ref1 || !fenc->origin || !ref0->origin || !ref1->origin || !jobs || njobs < 0)
        return fail(c, X265AMD_EINVAL, "null argument");
    const x265amd_plane *refs[2] = { ref0, ref1 };
    for (int i = 0; i < njobs; i++)
    {
        const x265amd_bipred_job &j = jobs[i];
        if (x265amd_partition_from_sizes(j.w, j.h) < 0) return fail(c, X265AMD_EINVAL, "job: not a luma partition");
        if (j.pu_x < 0 || j.pu_y < 0 || j.pu_x + j.w > fenc->width + fenc->margin_x || j.pu_y + j.h > fenc->height + fenc->margin_y)
            return fail(c, X265AMD_EINVAL, "job: PU outside the source plane");
        for (int l = 0; l < 2; l++)
        {
            const int16_t *mv = l ? j.mv1 : j.mv0;
            /* the 8-tap footprint of the prediction: 3 samples before, 4 after, tiles read as whole quads */
            const int x0 = j.pu_x + (mv[0] >> 2) - 3, x1 = j.pu_x + j.w + (mv[0] >> 2) + 8;
            const int y0 = j.pu_y + (mv[1] >> 2) - 3, y1 = j.pu_y + j.h + (mv[1] >> 2) + 4;
            const x265amd_plane *r = refs[l];
            if (x0 < -r->margin_x || y0 < -r->margin_y || x1 > r->width + r->margin_x || y1 > r->height + r->margin_y)
                return fail(c, X265AMD_EINVAL, "job: prediction block leaves the padded reference plane");
        }
    }
    return X265AMD_OK;
}

struct x265amd_bipred_list
{
    void *d_blob;              /* jobs | first[njobs + 64] | chunk_job[nchunks] */
    size_t first_off, chunk_off;
    int njobs, nchunks;
};

/* zeroes d_out and launches the 64-tiles-per-wave kernel on the engine's stream */
static int bipred_flat_launch(x265amd_ctx *c, const x265amd_plane *fenc, const x265amd_plane *ref0, const x265amd_plane *ref1,
                              const void *d_blob, const x265amd_bipred_list *l, const x265amd_result *res0, const x265amd_result *res1,
                              x265amd_bipred_result *d_out)
{
    HIPCHK(c, hipMemsetAsync(d_out, 0, (size_t)l->njobs * sizeof(x265amd_bipred_result), c->stream));
    BipredFlatArgs fa;
    fa.b.fenc.origin = fenc->origin; fa.b.fenc.stride = fenc->stride;
    fa.b.ref0.origin = ref0->origin; fa.b.ref0.stride = ref0->stride;
    fa.b.ref1.origin = ref1->origin; fa.b.ref1.stride = ref1->stride;
    fa.b.depth = c->depth; fa.b.njobs = l->njobs;
    fa.b.jobs = (const x265amd_bipred_job*)d_blob; fa.b.results = d_out;
    fa.first = (const int32_t*)((const char*)d_blob + l->first_off);
    fa.chunk_job = (const int32_t*)((const char*)d_blob + l->chunk_off);
    fa.nchunks = l->nchunks;
    fa.res0 = res0; fa.res1 = res1;
    if (c->depth <= 8) hipLaunchKernelGGL(bipred_flat_kernel<uint8_t>, dim3(fa.nchunks), dim3(64), 0, c->stream, fa);
    else hipLaunchKernelGGL(bipred_flat_kernel<uint16_t>, dim3(fa.nchunks), dim3(64), 0, c->stream, fa);
    HIPCHK(c, hipGetLastError());
    return X265AMD_OK;
}

static void bipred_plan(const x265amd_bipred_job *jobs, int njobs, std::vector<char> &blob, x265amd_bipred_list *l)
{
    /* host job arrays show their sizes: tiles are numbered here and waves take 64 of them at a time, whatever PUs they
     * belong to (x265amd_bipred_batch_dev cannot see the sizes and runs one wave per PU) */
    std::vector<int> tiles(njobs);
    for (int i = 0; i < njobs; i++) tiles[i] = (jobs[i].w >> 2) * (jobs[i].h >> 2);
    l->njobs = njobs;
    flat_plan(jobs, sizeof(x265amd_bipred_job), njobs, tiles, blob, l->first_off, l->chunk_off, l->nchunks);
}

int x265amd_bipred_batch(x265amd_ctx *c, const x265amd_plane *fenc, const x265amd_plane *ref0, const x265amd_plane *ref1,
                         const x265amd_bipred_job *jobs, int njobs, x265amd_bipred_result *results)
{
    if (!results) return fail(c, X265AMD_EINVAL, "null argument");
    int rc = bipred_validate(c, fenc, ref0, ref1, jobs, njobs);
    if (rc) return rc;
    if (!njobs) return X265AMD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    x265amd_bipred_list l;
    std::vector<char> blob;
    bipred_plan(jobs, njobs, blob, &l);
    if ((rc = ensure(c, &c->d_jobs, &c->jobs_cap, blob.size()))) return rc;
    if ((rc = ensure(c, &c->d_results, &c->results_cap, (size_t)njobs * sizeof(x265amd_bipred_result)))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_jobs, blob.data(), blob.size(), hipMemcpyHostToDevice, c->stream));
    if ((rc = bipred_flat_launch(c, fenc, ref0, ref1, c->d_jobs, &l, 0, 0, (x265amd_bipred_result*)c->d_results))) return rc;
    HIPCHK(c, hipMemcpyAsync(results, c->d_results, (size_t)njobs * sizeof(x265amd_bipred_result), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return X265AMD_OK;
}

int x265amd_bipred_list_create(x265amd_ctx *c, const x265amd_plane *fenc, const x265amd_plane *ref0, const x265amd_plane *ref1,
                               const x265amd_bipred_job *jobs, int njobs, x265amd_bipred_list **out)
{
    if (!out) return fail(c, X265AMD_EINVAL, "null argument");
    int rc = bipred_validate(c, fenc, ref0, ref1, jobs, njobs);
    if (rc) return rc;
    x265amd_bipred_list *l = new x265amd_bipred_list();
    std::vector<char> blob;
    bipred_plan(jobs, njobs, blob, l);
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = hipMalloc(&l->d_blob, blob.size() + 16);
    if (e == hipSuccess) e = hipMemcpyAsync(l->d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess)
    {
        if (l->d_blob) (void)hipFree(l->d_blob);
        delete l;
        HIPCHK(c, e);
    }
    *out = l;
    return X265AMD_OK;
}

int x265amd_bipred_list_free(x265amd_ctx *c, x265amd_bipred_list *l)
{
    if (!c) return X265AMD_EINVAL;
    if (!l) return X265AMD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipError_t e = hipFree(l->d_blob);
    delete l;
    HIPCHK(c, e);
    return X265AMD_OK;
}

int x265amd_bipred_batch_list(x265amd_ctx *c, const x265amd_plane *fenc, const x265amd_plane *ref0, const x265amd_plane *ref1,
                              const x265amd_bipred_list *l, const x265amd_result *d_results0, const x265amd_result *d_results1,
                              x265amd_bipred_result *d_out)
{
    if (!c || !fenc || !ref0 || !ref1 || !fenc->origin || !ref0->origin || !ref1->origin || !l || !d_out)
        return fail(c, X265AMD_EINVAL, "null argument");
    if (!l->njobs) return X265AMD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    return bipred_flat_launch(c, fenc, ref0, ref1, l->d_blob, l, d_results0, d_results1, d_out);
}

/* ---- merge-candidate estimate ---- */

int x265amd_merge_sa8d_batch(x265amd_ctx *c, const x265amd_plane *fenc, const x265amd_plane *refs, int nrefs,
                             const x265amd_merge_job *jobs, int njobs, int32_t *sa8d)
{
    if (!c || !fenc || !fenc->origin || !refs || nrefs <= 0 || nrefs > 32 || !jobs || !sa8d || njobs < 0)
        return fail(c, X265AMD_EINVAL, "null argument");
    if (!njobs) return X265AMD_OK;
    for (int i = 0; i < njobs; i++)
    {
        const x265amd_merge_job &j = jobs[i];
        if (!(j.size == 8 || j.size == 16 || j.size == 32 || j.size == 64)) return fail(c, X265AMD_EINVAL, "job: CU size must be 8, 16, 32 or 64");
        if (j.dir < 1 || j.dir > 3 || j.ref0 >= nrefs || j.ref1 >= nrefs) return fail(c, X265AMD_EINVAL, "job: bad direction or reference index");
        if (j.pu_x < 0 || j.pu_y < 0 || j.pu_x + j.size > fenc->width + fenc->margin_x || j.pu_y + j.size > fenc->height + fenc->margin_y)
            return fail(c, X265AMD_EINVAL, "job: CU outside the source plane");
        for (int l = 0; l < 2; l++)
        {
            if (!(j.dir & (1 << l))) continue;
            const int16_t *mv = l ? j.mv1 : j.mv0;
            const x265amd_plane *r = &refs[l ? j.ref1 : j.ref0];
            /* a whole-sample MV of a single list reads the block itself (this is how the distortion of a finished
             * prediction picture, which has no margins, is taken: the picture as reference, MV 0); everything else
             * reads the 8-tap footprint, tiles as whole quads */
            const bool copy = j.dir != 3 && ((mv[0] | mv[1]) & 3) == 0;
            const int x0 = j.pu_x + (mv[0] >> 2) - (copy ? 0 : 3), x1 = j.pu_x + j.size + (mv[0] >> 2) + (copy ? 0 : 8);
            const int y0 = j.pu_y + (mv[1] >> 2) - (copy ? 0 : 3), y1 = j.pu_y + j.size + (mv[1] >> 2) + (copy ? 0 : 4);
            if (!r->origin || x0 < -r->margin_x || y0 < -r->margin_y || x1 > r->width + r->margin_x || y1 > r->height + r->margin_y)
                return fail(c, X265AMD_EINVAL, "job: prediction block leaves the padded reference plane");
        }
    }
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    PlaneDev hp[32];
    for (int i = 0; i < nrefs; i++) { hp[i].origin = refs[i].origin; hp[i].stride = refs[i].stride; }
    /* tiles numbered on the host (CU sizes are visible here); waves take 64 of them at a time */
    std::vector<int> tiles(njobs);
    for (int i = 0; i < njobs; i++) tiles[i] = (jobs[i].size >> 2) * (jobs[i].size >> 2);
    std::vector<char> blob;
    MergeFlatArgs fa;
    size_t first_off, chunk_off;
    flat_plan(jobs, sizeof(x265amd_merge_job), njobs, tiles, blob, first_off, chunk_off, fa.nchunks, 1);
    const size_t pb = sizeof(PlaneDev) * 32;
    if ((rc = ensure(c, &c->d_jobs, &c->jobs_cap, pb + blob.size()))) return rc;
    if ((rc = ensure(c, &c->d_results, &c->results_cap, (size_t)njobs * sizeof(int32_t)))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_jobs, hp, sizeof(PlaneDev) * nrefs, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync((char*)c->d_jobs + pb, blob.data(), blob.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_results, 0, (size_t)njobs * sizeof(int32_t), c->stream));
    MergeArgs &a = fa.m;
    a.fenc.origin = fenc->origin; a.fenc.stride = fenc->stride;
    a.refs = (const PlaneDev*)c->d_jobs; a.depth = c->depth; a.njobs = njobs;
    a.jobs = (const x265amd_merge_job*)((char*)c->d_jobs + pb); a.out = (int32_t*)c->d_results;
    fa.first = (const int32_t*)((const char*)a.jobs + first_off);
    fa.chunk_job = (const int32_t*)((const char*)a.jobs + chunk_off);
    if (c->depth <= 8) hipLaunchKernelGGL(merge_flat_kernel<uint8_t>, dim3(fa.nchunks), dim3(64), 0, c->stream, fa);
    else hipLaunchKernelGGL(merge_flat_kernel<uint16_t>, dim3(fa.nchunks), dim3(64), 0, c->stream, fa);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(sa8d, c->d_results, (size_t)njobs * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return X265AMD_OK;
}

/* ---- motion compensation ---- */

static int mc_ref_table(x265amd_ctx *c, const x265amd_picture *refs, const x265amd_mc_weight *weights, int nrefs, McRef *hp)
{
    memset(hp, 0, sizeof(McRef) * nrefs);
    for (int i = 0; i < nrefs; i++)
    {
        const x265amd_plane *pl[3] = { &refs[i].y, &refs[i].cb, &refs[i].cr };
        for (int k = 0; k < 3; k++)
        {
            if (!pl[k]->origin) return fail(c, X265AMD_EINVAL, "picture: null plane");
            hp[i].pl[k].origin = pl[k]->origin; hp[i].pl[k].stride = pl[k]->stride;
        }
        if (weights)
        {
            for (int k = 0; k < 3; k++)
                if (weights[i].log2denom[k] > 7 || weights[i].w[k] < -128 || weights[i].w[k] > 255 || weights[i].offset[k] < -128 ||
                    weights[i].offset[k] > 127)
                    return fail(c, X265AMD_EINVAL, "weight: log2denom is 0..7, weight -128..255, offset -128..127");
            hp[i].wp = weights[i];
        }
    }
    return X265AMD_OK;
}

int x265amd_pictures_upload(x265amd_ctx *c, const x265amd_picture *refs, const x265amd_mc_weight *weights, int nrefs, void **d_ref_planes)
{
    if (!c || !refs || nrefs <= 0 || nrefs > 32 || !d_ref_planes) return fail(c, X265AMD_EINVAL, "null argument");
    McRef hp[32];
    int rc = mc_ref_table(c, refs, weights, nrefs, hp);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    void *d = 0;
    HIPCHK(c, hipMalloc(&d, sizeof(McRef) * nrefs));
    hipError_t e = hipMemcpyAsync(d, hp, sizeof(McRef) * nrefs, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { (void)hipFree(d); HIPCHK(c, e); }
    *d_ref_planes = d;
    return X265AMD_OK;
}

struct x265amd_mc_list
{
    void *d_blob;              /* jobs | first[njobs + 64] | chunk_job[nchunks] */
    size_t first_off, chunk_off;
    int njobs, nchunks;
};

static int mc_validate(x265amd_ctx *c, const x265amd_picture *refs, const x265amd_mc_weight *weights, int nrefs,
                       const x265amd_picture *pred, const x265amd_mc_job *jobs, int njobs)
{
    if (!c || !refs || nrefs <= 0 || nrefs > 32 || !pred || !jobs || njobs < 0) return fail(c, X265AMD_EINVAL, "null argument");
    if (!pred->y.origin || !pred->cb.origin || !pred->cr.origin) return fail(c, X265AMD_EINVAL, "null prediction plane");
    for (int i = 0; i < nrefs; i++)
        if (!refs[i].y.origin || !refs[i].cb.origin || !refs[i].cr.origin) return fail(c, X265AMD_EINVAL, "picture: null plane");
    for (int i = 0; i < njobs; i++)
    {
        const x265amd_mc_job &j = jobs[i];
        if (x265amd_partition_from_sizes(j.w, j.h) < 1) return fail(c, X265AMD_EINVAL, "job: not a luma partition above 4x4");
        if (j.dir < 1 || j.dir > 3 || !(j.flags & 3) || (j.flags & ~7) || j.ref0 >= nrefs || j.ref1 >= nrefs)
            return fail(c, X265AMD_EINVAL, "job: bad direction, flags or reference index");
        if ((j.flags & 4) && !weights) return fail(c, X265AMD_EINVAL, "job: weighted, but no weights were given");
        if (j.pu_x < 0 || j.pu_y < 0 || (j.pu_x & 3) || (j.pu_y & 3)) return fail(c, X265AMD_EINVAL, "job: PU position must be a multiple of 4");
        if (((j.flags & 1) && (j.pu_x + j.w > pred->y.width || j.pu_y + j.h > pred->y.height)) ||
            ((j.flags & 2) && ((j.pu_x + j.w) / 2 > pred->cb.width || (j.pu_y + j.h) / 2 > pred->cb.height ||
                               (j.pu_x + j.w) / 2 > pred->cr.width || (j.pu_y + j.h) / 2 > pred->cr.height)))
            return fail(c, X265AMD_EINVAL, "job: PU outside the prediction picture");
        for (int l = 0; l < 2; l++)
        {
            if (!(j.dir & (1 << l))) continue;
            const int16_t *mv = l ? j.mv1 : j.mv0;
            const x265amd_picture *r = &refs[l ? j.ref1 : j.ref0];
            if (j.flags & 1)
            {
                /* 8-tap footprint, tiles read as whole quads */
                const int x0 = j.pu_x + (mv[0] >> 2) - 3, x1 = j.pu_x + j.w + (mv[0] >> 2) + 8;
                const int y0 = j.pu_y + (mv[1] >> 2) - 3, y1 = j.pu_y + j.h + (mv[1] >> 2) + 4;
                if (x0 < -r->y.margin_x || y0 < -r->y.margin_y || x1 > r->y.width + r->y.margin_x || y1 > r->y.height + r->y.margin_y)
                    return fail(c, X265AMD_EINVAL, "job: luma prediction block leaves the padded reference plane");
            }
            if (j.flags & 2)
                for (int k = 0; k < 2; k++)
                {
                    /* 4-tap footprint; a 2x2 block reads its rows as two quads from one sample to the left */
                    const x265amd_plane *p = k ? &r->cr : &r->cb;
                    const int x0 = j.pu_x / 2 + (mv[0] >> 3) - 1, x1 = (j.pu_x + j.w) / 2 + (mv[0] >> 3) + 5;
                    const int y0 = j.pu_y / 2 + (mv[1] >> 3) - 1, y1 = (j.pu_y + j.h) / 2 + (mv[1] >> 3) + 2;
                    if (x0 < -p->margin_x || y0 < -p->margin_y || x1 > p->width + p->margin_x || y1 > p->height + p->margin_y)
                        return fail(c, X265AMD_EINVAL, "job: chroma prediction block leaves the padded reference plane");
                }
        }
    }
    return X265AMD_OK;
}

/* Numbers the 4x4 tiles of `njobs` records (tiles[i] each) consecutively and lays records | first[] | chunk_job[] out in
 * `blob`: first[] has njobs + 64 entries (the tail repeats the total), chunk_job[k] is the record tile 64 k belongs to.
 * With align16 a record starts at a multiple of min(its tiles, 16); the tiles skipped belong to the record before. */
static void flat_plan(const void *jobs, size_t job_size, int njobs, const std::vector<int> &tiles, std::vector<char> &blob,
                      size_t &first_off, size_t &chunk_off, int &nchunks, int align16)
{
    std::vector<int32_t> start(njobs + 1);
    long run = 0;
    for (int i = 0; i < njobs; i++)
    {
        if (align16)
        {
            const int al = tiles[i] < 16 ? tiles[i] : 16;
            run = (run + al - 1) / al * al;
        }
        start[i] = (int32_t)run;
        run += tiles[i];
    }
    start[njobs] = (int32_t)run;
    nchunks = (int)((run + 63) / 64);
    first_off = ((size_t)njobs * job_size + 15) & ~(size_t)15;
    chunk_off = first_off + ((size_t)njobs + 64) * sizeof(int32_t);
    blob.assign(chunk_off + (size_t)nchunks * sizeof(int32_t), 0);
    memcpy(blob.data(), jobs, (size_t)njobs * job_size);
    int32_t *first = (int32_t*)(blob.data() + first_off), *chunk = (int32_t*)(blob.data() + chunk_off);
    for (int i = 0; i < njobs; i++)
    {
        first[i] = start[i];
        for (int k = (start[i] + 63) / 64; 64 * k < start[i + 1]; k++) chunk[k] = i;
    }
    for (int i = 0; i < 64; i++) first[njobs + i] = start[njobs];
}

int x265amd_tile_plan(const int32_t *tiles, int n, int align16, int32_t *first, int32_t *chunk_job, int *nchunks)
{
    if (!tiles || n < 0 || !first || !chunk_job || !nchunks) return X265AMD_EINVAL;
    std::vector<int> t(tiles, tiles + n);
    for (int i = 0; i < n; i++)
        if (t[i] < 1) return X265AMD_EINVAL;
    std::vector<char> blob, none(1);
    size_t fo, co;
    flat_plan(none.data(), 0, n, t, blob, fo, co, *nchunks, align16);
    memcpy(first, blob.data() + fo, ((size_t)n + 64) * sizeof(int32_t));
    memcpy(chunk_job, blob.data() + co, (size_t)*nchunks * sizeof(int32_t));
    return X265AMD_OK;
}

static void mc_plan(const x265amd_mc_job *jobs, int njobs, std::vector<char> &blob, x265amd_mc_list *l)
{
    std::vector<int> tiles(njobs);
    for (int i = 0; i < njobs; i++) tiles[i] = (jobs[i].w >> 2) * (jobs[i].h >> 2);
    l->njobs = njobs;
    flat_plan(jobs, sizeof(x265amd_mc_job), njobs, tiles, blob, l->first_off, l->chunk_off, l->nchunks);
}

static int mc_launch(x265amd_ctx *c, const void *d_ref_planes, const x265amd_picture *pred, const void *d_blob, const x265amd_mc_list *l,
                     const x265amd_result *res0 = 0, const x265amd_result *res1 = 0)
{
    McArgs a;
    a.res0 = res0; a.res1 = res1;
    a.refs = (const McRef*)d_ref_planes;
    a.pred[0].origin = pred->y.origin; a.pred[0].stride = pred->y.stride;
    a.pred[1].origin = pred->cb.origin; a.pred[1].stride = pred->cb.stride;
    a.pred[2].origin = pred->cr.origin; a.pred[2].stride = pred->cr.stride;
    a.depth = c->depth; a.njobs = l->njobs; a.nchunks = l->nchunks;
    a.jobs = (const x265amd_mc_job*)d_blob;
    a.first = (const int32_t*)((const char*)d_blob + l->first_off);
    a.chunk_job = (const int32_t*)((const char*)d_blob + l->chunk_off);
    if (c->depth <= 8) hipLaunchKernelGGL(mc_kernel<uint8_t>, dim3(l->nchunks), dim3(64), 0, c->stream, a);
    else hipLaunchKernelGGL(mc_kernel<uint16_t>, dim3(l->nchunks), dim3(64), 0, c->stream, a);
    HIPCHK(c, hipGetLastError());
    return X265AMD_OK;
}

int x265amd_mc_list_create(x265amd_ctx *c, const x265amd_picture *refs, const x265amd_mc_weight *weights, int nrefs,
                           const x265amd_picture *pred, const x265amd_mc_job *jobs, int njobs, x265amd_mc_list **out)
{
    if (!out) return fail(c, X265AMD_EINVAL, "null argument");
    int rc = mc_validate(c, refs, weights, nrefs, pred, jobs, njobs);
    if (rc) return rc;
    x265amd_mc_list *l = new x265amd_mc_list();
    std::vector<char> blob;
    mc_plan(jobs, njobs, blob, l);
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = hipMalloc(&l->d_blob, blob.size() + 16);
    if (e == hipSuccess) e = hipMemcpyAsync(l->d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess)
    {
        if (l->d_blob) (void)hipFree(l->d_blob);
        delete l;
        HIPCHK(c, e);
    }
    *out = l;
    return X265AMD_OK;
}

int x265amd_mc_list_free(x265amd_ctx *c, x265amd_mc_list *l)
{
    if (!c) return X265AMD_EINVAL;
    if (!l) return X265AMD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipError_t e = hipFree(l->d_blob);
    delete l;
    HIPCHK(c, e);
    return X265AMD_OK;
}

int x265amd_mc_batch_dev_mv(x265amd_ctx *c, const void *d_ref_planes, const x265amd_picture *pred, const x265amd_mc_list *l,
                            const x265amd_result *d_results0, const x265amd_result *d_results1)
{
    if (!c || !d_ref_planes || !pred || !pred->y.origin || !pred->cb.origin || !pred->cr.origin || !l)
        return fail(c, X265AMD_EINVAL, "null argument");
    if (!l->njobs) return X265AMD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    return mc_launch(c, d_ref_planes, pred, l->d_blob, l, d_results0, d_results1);
}

int x265amd_mc_batch_dev(x265amd_ctx *c, const void *d_ref_planes, const x265amd_picture *pred, const x265amd_mc_list *l)
{
    return x265amd_mc_batch_dev_mv(c, d_ref_planes, pred, l, 0, 0);
}

int x265amd_mc_batch(x265amd_ctx *c, const x265amd_picture *refs, const x265amd_mc_weight *weights, int nrefs,
                     const x265amd_picture *pred, const x265amd_mc_job *jobs, int njobs)
{
    int rc = mc_validate(c, refs, weights, nrefs, pred, jobs, njobs);
    if (rc) return rc;
    if (!njobs) return X265AMD_OK;
    McRef hp[32];
    if ((rc = mc_ref_table(c, refs, weights, nrefs, hp))) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    x265amd_mc_list l;
    std::vector<char> blob;
    mc_plan(jobs, njobs, blob, &l);
    const size_t pb = sizeof(McRef) * 32;
    if ((rc = ensure(c, &c->d_jobs, &c->jobs_cap, pb + blob.size()))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_jobs, hp, sizeof(McRef) * nrefs, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync((char*)c->d_jobs + pb, blob.data(), blob.size(), hipMemcpyHostToDevice, c->stream));
    if ((rc = mc_launch(c, c->d_jobs, pred, (char*)c->d_jobs + pb, &l))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return X265AMD_OK;
}

int x265amd_rk_ds_plane(x265amd_ctx *c, const x265amd_plane *orig, x265amd_plane *out)
{
    if (!c || !orig || !orig->origin || !out) return fail(c, X265AMD_EINVAL, "null plane");
    if (orig->margin_x != orig->margin_y || (orig->margin_x & 3))
        return fail(c, X265AMD_EINVAL, "RK downsampled plane needs equal margins that are multiples of 4");
    const int w4 = orig->width / 4, h4 = orig->height / 4, m4 = orig->margin_x / 4;
    const int stride = w4 + 2 * m4;
    void *d = 0;
    HIPCHK(c, hipSetDevice(c->device));
    /* + slack, as for every plane: window staging reads whole 16-byte chunks */
    const size_t ds_bytes = (size_t)(h4 + 2 * m4) * stride * sizeof(int16_t) + 256;
    HIPCHK(c, hipMalloc(&d, ds_bytes));
    HIPCHK(c, hipMemsetAsync(d, 0, ds_bytes, c->stream));
    int16_t *origin = (int16_t*)d + (size_t)m4 * stride + m4;
    c->plane_base[origin] = d;
    dim3 blk(64, 4), grd((w4 + 63) / 64, (h4 + 3) / 4), grd2((w4 + 2 * m4 + 63) / 64, (h4 + 2 * m4 + 3) / 4);
    if (c->depth <= 8)
        hipLaunchKernelGGL(rk_downsample_kernel<uint8_t>, grd, blk, 0, c->stream, (const uint8_t*)orig->origin, (long)orig->stride,
                           origin, (long)stride, w4, h4);
    else
        hipLaunchKernelGGL(rk_downsample_kernel<uint16_t>, grd, blk, 0, c->stream, (const uint16_t*)orig->origin, (long)orig->stride,
                           origin, (long)stride, w4, h4);
    hipLaunchKernelGGL(extend_border_kernel<uint16_t>, grd2, blk, 0, c->stream, (uint16_t*)origin, (long)stride, w4, h4, m4, m4);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    out->origin = origin; out->stride = stride; out->width = w4; out->height = h4; out->margin_x = m4; out->margin_y = m4;
    return X265AMD_OK;
}

int x265amd_rk_ctu_batch_dev(x265amd_ctx *c, const x265amd_plane *fenc, const x265amd_plane *recon, const x265amd_plane *ds,
                             int qp, const x265amd_rk_ctu *d_ctus, int nctus, x265amd_rk_result *d_results)
{
    if (!c || !fenc || !recon || !ds || !d_ctus || !d_results || nctus < 0) return fail(c, X265AMD_EINVAL, "null argument");
    /* the 4x4 sums of the coarse stage are int16 and the block surfaces of the refinement 16 bits wide, as the
     * reference's `short` buffers are: 16 * 1023 and 64 * 1023 fit, the sums of deeper samples do not */
    if (c->depth > 10) return fail(c, X265AMD_EINVAL, "the Rockchip path takes samples of up to 10 bits");
    int rc = x265amd_set_qp(c, qp);
    if (rc) return rc;
    if (!nctus) return X265AMD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    RkArgs a;
    a.fenc.origin = fenc->origin; a.fenc.stride = fenc->stride;
    a.recon.origin = recon->origin; a.recon.stride = recon->stride;
    a.ds = (const int16_t*)ds->origin; a.ds_stride = ds->stride;
    a.pic_w = fenc->width; a.pic_h = fenc->height; a.depth = c->depth;
    a.ctab = c->tables[qp] + 32768;
    a.ctus = d_ctus; a.results = d_results; a.nctus = nctus;
    /* window of the widest range the RK rules allow: (384/2/4*2 + 15) x (320/2/4*2 + 15) int16 */
    const size_t lds = (512 + 112 * 95 + 16) * sizeof(int16_t);
    if (c->depth <= 8)
    {
        if (!c->rime_lds_raised)
        {
            HIPCHK(c, hipFuncSetAttribute((const void*)rk_rime_kernel<uint8_t>, hipFuncAttributeMaxDynamicSharedMemorySize, RL_TOTAL));
            c->rime_lds_raised = true;
        }
        hipLaunchKernelGGL(rk_cime_kernel<uint8_t>, dim3(nctus), dim3(CIME_THREADS), lds, c->stream, a);
        hipLaunchKernelGGL(rk_rime_kernel<uint8_t>, dim3(nctus), dim3(RIME_THREADS), RL_TOTAL, c->stream, a);
        hipLaunchKernelGGL((rk_cu_kernel<uint8_t, true>), dim3(nctus * 85), dim3(64), L_FENC + rk_region_bytes<uint8_t>(64), c->stream, a);
    }
    else
    {
        /* RIME of the whole CTU from 8x8 block surfaces, as with 8-bit samples */
        if (!c->rime_lds_raised)
        {
            HIPCHK(c, hipFuncSetAttribute((const void*)rk_rime16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, R16_TOTAL));
            c->rime_lds_raised = true;
        }
        hipLaunchKernelGGL(rk_cime_kernel<uint16_t>, dim3(nctus), dim3(CIME_THREADS), lds, c->stream, a);
        hipLaunchKernelGGL(rk_rime16_kernel, dim3(nctus), dim3(RIME_THREADS), R16_TOTAL, c->stream, a);
        hipLaunchKernelGGL((rk_cu_kernel<uint16_t, true>), dim3(nctus * 85), dim3(64), L_FENC + rk_region_bytes<uint16_t>(64), c->stream, a);
    }
    HIPCHK(c, hipGetLastError());
    return X265AMD_OK;
}

int x265amd_rk_ctu_batch(x265amd_ctx *c, const x265amd_plane *fenc, const x265amd_plane *recon, const x265amd_plane *ds,
                         int qp, const x265amd_rk_ctu *ctus, int nctus, x265amd_rk_result *results)
{
    if (!c || !fenc || !recon || !ds || !ctus || !results || nctus < 0) return fail(c, X265AMD_EINVAL, "null argument");
    for (int i = 0; i < nctus; i++)
    {
        const x265amd_rk_ctu &j = ctus[i];
        if (j.ctu_x < 0 || j.ctu_y < 0 || (j.ctu_x & 63) || (j.ctu_y & 63) || j.ctu_x >= fenc->width || j.ctu_y >= fenc->height)
            return fail(c, X265AMD_EINVAL, "rk ctu: origin is not a CTU position inside the picture");
        if (j.valid_w < 8 || j.valid_w > 64 || j.valid_h < 8 || j.valid_h > 64 || (j.valid_w & 3) || (j.valid_h & 3))
            return fail(c, X265AMD_EINVAL, "rk ctu: bad valid width / height");
        if (j.mvmin[0] > j.mvmax[0] || j.mvmin[1] > j.mvmax[1] || j.mvmax[0] / 4 - j.mvmin[0] / 4 + 1 > 96 ||
            j.mvmax[1] / 4 - j.mvmin[1] / 4 + 1 > 80)
            return fail(c, X265AMD_EINVAL, "rk ctu: CIME range empty or wider than 96 x 80 downsampled positions");
        /* CIME window inside the downsampled plane; RIME/FME windows inside the reconstructed plane */
        if (j.ctu_x / 4 + j.mvmin[0] / 4 < -ds->margin_x || j.ctu_x / 4 + j.mvmax[0] / 4 + 16 > ds->width + ds->margin_x ||
            j.ctu_y / 4 + j.mvmin[1] / 4 < -ds->margin_y || j.ctu_y / 4 + j.mvmax[1] / 4 + 16 > ds->height + ds->margin_y)
            return fail(c, X265AMD_EINVAL, "rk ctu: CIME window leaves the downsampled plane");
        int lo[2] = { j.mvmin[0], j.mvmin[1] }, hi[2] = { j.mvmax[0], j.mvmax[1] };
        for (int k = 0; k < 2; k++)
            for (int d = 0; d < 2; d++) { if (j.pmv[k][d] < lo[d]) lo[d] = j.pmv[k][d]; if (j.pmv[k][d] > hi[d]) hi[d] = j.pmv[k][d]; }
        if (j.ctu_x + lo[0] - 8 - 6 < -recon->margin_x || j.ctu_x + 64 + hi[0] + 8 + 6 > recon->width + recon->margin_x ||
            j.ctu_y + lo[1] - 6 - 6 < -recon->margin_y || j.ctu_y + 64 + hi[1] + 6 + 6 > recon->height + recon->margin_y)
            return fail(c, X265AMD_EINVAL, "rk ctu: RIME / FME windows leave the padded reference plane");
    }
    if (!nctus) return X265AMD_OK;
    int rc;
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = ensure(c, &c->d_jobs, &c->jobs_cap, (size_t)nctus * sizeof(x265amd_rk_ctu)))) return rc;
    if ((rc = ensure(c, &c->d_results, &c->results_cap, (size_t)nctus * sizeof(x265amd_rk_result)))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_jobs, ctus, (size_t)nctus * sizeof(x265amd_rk_ctu), hipMemcpyHostToDevice, c->stream));
    rc = x265amd_rk_ctu_batch_dev(c, fenc, recon, ds, qp, (const x265amd_rk_ctu*)c->d_jobs, nctus, (x265amd_rk_result*)c->d_results);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(results, c->d_results, (size_t)nctus * sizeof(x265amd_rk_result), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return X265AMD_OK;
}

} /* extern "C" */

#include "me_primitives.inc"
