/* x265_amd_me.h -- C ABI of the MI355X motion-estimation engine.
 *
 * Drop-in boundary for the hot path behind x265's MotionEstimate class and the
 * EncoderPrimitives function table. Plain C, plain pointers and sizes; every entry
 * point returns 0 on success or a negative X265AMD_E* code (the reference path has
 * no error returns -- motion.cpp only asserts -- so errors are new and only cover
 * bad arguments and HIP failures; x265amd_last_error() gives the text).
 *
 * Each declaration cites the reference interface it replaces
 * (paths relative to the reference's source/ directory).
 */
#ifndef X265_AMD_ME_H
#define X265_AMD_ME_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define X265AMD_OK          0
#define X265AMD_EINVAL     -1   /* bad argument */
#define X265AMD_EHIP       -2   /* HIP runtime error */
#define X265AMD_ENOMEM     -3
#define X265AMD_EBUSY      -4   /* a resource that serves one call at a time is in use */

/* Incremented whenever an entry point that exists changes its arguments (2: x265amd_fields_upload takes the CU grid,
 * method and merange it checks the planes for). A binding compares x265amd_abi_version() with the header it was written
 * against before its first call: a stale binding would pass arguments in the wrong slots, which no linker notices. */
#define X265AMD_ABI_VERSION 2
int x265amd_abi_version(void);

/* search methods: X265_DIA_SEARCH .. X265_FULL_SEARCH, x265.h (searchMethod) */
enum { X265AMD_DIA = 0, X265AMD_HEX = 1, X265AMD_UMH = 2, X265AMD_STAR = 3, X265AMD_FULL = 4 };

typedef struct x265amd_ctx x265amd_ctx;

/* A padded picture plane resident in device memory. Replaces the
 * (pixel* fpelPlane, int lumaStride) pair of struct ReferencePlanes
 * (common/lowres.h:37-52) and MotionEstimate::setSourcePlane (encoder/motion.h:76).
 * `origin` is the DEVICE address of pixel (0,0); `margin_*` replicated border
 * pixels exist on every side (TComPicYuv.cpp:69-131). Samples are uint8_t for
 * bit depth 8 and uint16_t above. */
typedef struct
{
    void   *origin;
    int32_t stride;            /* in samples */
    int32_t width, height;
    int32_t margin_x, margin_y;
} x265amd_plane;

/* One PU search = one call of MotionEstimate::setSourcePU + motionEstimate
 * (encoder/motion.cpp:116, :549). MVs are {x, y}; mvmin/mvmax full-pel inclusive,
 * qmvp and mvc quarter-pel, as in the reference signature (encoder/motion.h:93). */
typedef struct
{
    int16_t pu_x, pu_y;        /* PU position in the picture: blockOffset = pu_y * stride + pu_x */
    uint8_t w, h;              /* an HEVC luma partition (common/primitives.h:65-73), not 4x4 */
    uint8_t numc;              /* numCandidates, 0..4 */
    uint8_t reserved;          /* 0; in the x265amd_search_batch_pairs calls: index of the job's (source, reference) pair */
    int16_t mvmin[2], mvmax[2];
    int16_t qmvp[2];
    int16_t mvc[4][2];
    int32_t merange;
} x265amd_job;

typedef struct
{
    int16_t  mv[2];            /* outQMv */
    int32_t  cost;             /* return value of motionEstimate */
    uint32_t sad_calls;        /* full-pel SAD evaluations made (x3/x4 groups count 3/4) */
    uint32_t subpel_calls;     /* subpelCompare / lowresQPelCost evaluations made */
} x265amd_result;

/* ---- context ---- */

/* Creates an engine bound to HIP device `device` for `bit_depth` 8, 10 or 12
 * (the reference fixes this at build time through HIGH_BIT_DEPTH / g_bitDepth). */
int x265amd_create(int device, int bit_depth, x265amd_ctx **out);
/* The same, with the engine's stream at the device's highest priority when high_priority > 0, at its lowest when
 * high_priority < 0 (normal priority otherwise, as x265amd_create): for a caller that runs several engines side by side
 * and wants one launch chain dispatched ahead of the others (or behind them). */
int x265amd_create_priority(int device, int bit_depth, int high_priority, x265amd_ctx **out);
void x265amd_destroy(x265amd_ctx *ctx);
const char *x265amd_last_error(const x265amd_ctx *ctx);
/* HIP stream (hipStream_t) the engine launches on; for event timing by callers */
void *x265amd_stream(x265amd_ctx *ctx);

/* BitCost::setQP (encoder/bitcost.cpp:34-59): builds (once) and selects the
 * lambda-scaled MVD cost table of `qp` (0..69); x265amd_cost_table copies the 65535
 * entries [-32767, 32767] of the device-resident table the kernels read to `out` (host). */
int x265amd_set_qp(x265amd_ctx *ctx, int qp);
int x265amd_cost_table(x265amd_ctx *ctx, int qp, uint16_t *out);

/* ---- planes ---- */

/* Uploads a padded host plane (`buf` holds (height + 2*margin_y) rows of `stride`
 * samples, pixel (0,0) at margin_y * stride + margin_x) and returns its descriptor.
 * Stands for the pictures the encoder already owns (TComPicYuv, Lowres::buffer). */
int x265amd_plane_upload(x265amd_ctx *ctx, const void *buf, int width, int height,
                         int margin_x, int margin_y, int stride, x265amd_plane *out);
/* Releases a plane made by this context: x265amd_plane_upload, x265amd_lowres_init, the weighting calls, and the
 * int16 plane of x265amd_rk_ds_plane alike (the allocation is looked up by `origin`, so the sample size does not
 * matter). Waits for the engine's stream first. X265AMD_EINVAL for a plane this context did not make. */
int x265amd_plane_free(x265amd_ctx *ctx, x265amd_plane *plane);
/* Lowres::init (common/lowres.cpp:121-166): frame_init_lowres_core + border
 * extension, on device. `src` is a full-resolution padded plane; produces the four
 * half-resolution planes (full-pel, H, V, centre half-pel) with `src`'s margins. */
int x265amd_lowres_init(x265amd_ctx *ctx, const x265amd_plane *src, x265amd_plane out[4]);
/* copies a whole padded plane (all rows incl. margins) back to the host */
int x265amd_plane_download(x265amd_ctx *ctx, const x265amd_plane *plane, void *buf);

/* raw device-to-host copy (after draining the engine's stream); for result arrays and
 * planes whose sample size differs from the context's (the int16 RK plane) */
int x265amd_download(x265amd_ctx *ctx, const void *d_src, void *dst, size_t bytes);

/* ---- motion search ---- */

/* MotionEstimate::motionEstimate for `njobs` PUs of one (source, reference) pair
 * (encoder/motion.cpp:549-1146; callers TEncSearch.cpp:4303, slicetype.cpp:1468).
 * `ref` points to 1 plane, or to the 4 lowres hpel planes when is_lowres
 * (ReferencePlanes::lowresPlane, lowres.h:44; lowres searches are 8x8 PUs only).
 * method / subme are setSearchMethod / setSubpelRefine (motion.h:70-72).
 * jobs and results are HOST arrays. */
int x265amd_search_batch(x265amd_ctx *ctx, const x265amd_plane *fenc, const x265amd_plane *ref, int is_lowres,
                         int qp, int method, int subme,
                         const x265amd_job *jobs, int njobs, x265amd_result *results);
/* Same, with jobs and results already in DEVICE memory; asynchronous on the
 * engine's stream (x265amd_sync waits). The jobs are not validated. `max_pu_dim` (8, 16, 32,
 * 64; 0 = 64) is the largest PU width / height in the batch: LDS per PU is reserved for that
 * size, so batches of small PUs keep more PUs resident per CU; a larger job in the batch gets
 * cost -1 and is not searched. With 8-bit samples the square PUs of exactly that size take the
 * fixed-shape kernels (several PUs per wave, or several waves per PU), so a caller after
 * throughput launches one batch per PU size; results do not depend on it. Other partitions in the batch (2NxN, Nx2N, the
 * asymmetric pairs, squares of other sizes) have fixed-shape kernels of their own for both stages: the small ones (8x4, 4x8,
 * 16x4, 4x16, 8x8, 16x8, 8x16) two to eight PUs per wave whatever the mix of shapes, the larger ones (16x12 .. 64x48) a
 * wave per PU; only squares of sizes the launch is not made for take the one-wave search. A batch for 64x64 PUs (samples of at most
 * 10 bits) takes its 32x32 PUs through the fixed-shape kernels as well, and a 64x64 PU that is directly followed in the list by
 * its four 32x32 children (raster order, same pictures) refines them together with itself where all five start the sub-pel
 * rounds from one MV -- a quarter of the children's work; the list order changes no result.
 * Planes must be readable 32 bytes past their last sample (planes made by this library are).
 * max_pu_dim | X265AMD_SQUARE_ONLY: the caller states that every PU of the list is a square of max_pu_dim samples (with
 * 64: of 64 or 32 samples), which is what a list made per PU size for throughput holds. An X265_STAR_SEARCH launch then
 * leaves out the pass that looks for PUs of other shapes in the list; such PUs, if the statement is false, are not
 * searched and their result records are meaningless. x265amd_search_batch sees its host job array and sets this itself. */
#define X265AMD_SQUARE_ONLY 0x100
int x265amd_search_batch_dev(x265amd_ctx *ctx, const x265amd_plane *fenc, const x265amd_plane *ref, int is_lowres,
                             int qp, int method, int subme,
                             const x265amd_job *d_jobs, int njobs, x265amd_result *d_results, int max_pu_dim);
int x265amd_sync(x265amd_ctx *ctx);
/* Measurement aid, no counterpart in the reference: how many PUs of this context's last launch through the fixed-shape
 * integer kernels (X265_STAR_SEARCH, X265_HEX_SEARCH, X265_DIA_SEARCH) did not end with what those kernels measure -- the
 * first three rounds of the star pattern (StarPatternSearch, encoder/motion.cpp:307-547: a point improved, so larger
 * distances, the two-point or raster refinement and re-centred patterns followed), the first hexagon ring (:673-707: it
 * moved the centre) or the first diamond round (:628-649) -- or were of a shape those kernels do not take; 0 for any other
 * launch. Waits for the engine's stream. */
int x265amd_marked_count(x265amd_ctx *ctx, int *marked);

/* The same searches over several (source, reference) pairs in ONE launch: job.reserved (0 .. npairs - 1) names the pair
 * a job searches. This is the shape of the caller's loops -- TEncSearch::predInterSearch runs motionEstimate for every
 * reference of both lists of a PU (Lib/TLibEncoder/TEncSearch.cpp:4281-4303), and the frame encoders of several frames
 * run at once (encoder/frameencoder.cpp:1087-1163) -- and what fills the GPU: a 64x64 PU is one long wave, so a launch
 * over one picture's 480 of them is bounded by that wave's latency, a launch over all pictures in flight by throughput.
 * Full-resolution planes only. x265amd_pairs_upload puts the table on the device (release with x265amd_device_free).
 * The host-array form checks every job against its own pair like x265amd_search_batch; the device-resident form does
 * not validate jobs (a pair index beyond the table is read as 0). */
typedef struct { x265amd_plane fenc, ref; } x265amd_pair;
int x265amd_pairs_upload(x265amd_ctx *ctx, const x265amd_pair *pairs, int npairs /* <= 256 */, void **d_pairs);
int x265amd_search_batch_pairs(x265amd_ctx *ctx, const x265amd_pair *pairs, int npairs, int qp, int method, int subme,
                               const x265amd_job *jobs, int njobs, x265amd_result *results);
int x265amd_search_batch_pairs_dev(x265amd_ctx *ctx, const void *d_pairs, int npairs, int qp, int method, int subme,
                                   const x265amd_job *d_jobs, int njobs, x265amd_result *d_results, int max_pu_dim);

/* An X265_STAR_SEARCH launch of the 16x16 PUs and of the 8x8 PUs of the pictures in flight as one: two calls of
 * x265amd_search_batch_pairs_dev (max_pu_dim 16 | X265AMD_SQUARE_ONLY, then 8 | X265AMD_SQUARE_ONLY) give the same results.
 * The lists hold only squares of their size. Where PU 4 k + q (q = 0 .. 3) of the children's list is quadrant q
 * (raster order) of PU k of the parents' list, searches the same pictures and starts its sub-pel rounds from the parent's
 * MV, the parent takes it through those rounds: SAD and SATD are sums over 4x4 tiles, so the child's candidates cost a few
 * instructions beside the parent's instead of a refinement of their own (encoder/motion.cpp:1071-1141 for both; the
 * quadtree of TEncCu::xCompressCU, Lib/TLibEncoder/TEncCu.cpp:2352, is why the callers have both). Children anywhere
 * else in their list, or whose decisions leave the parent's, are refined on their own: the order of the lists changes no
 * result. marked_count afterwards counts the parents' list. Samples of at most 10 bits. */
int x265amd_search_batch_pairs_family_dev(x265amd_ctx *ctx, const void *d_pairs, int npairs, int qp, int subme,
                                          const x265amd_job *d_parents, int nparents, x265amd_result *d_parent_results,
                                          const x265amd_job *d_children, int nchildren, x265amd_result *d_child_results);
/* The same launch for X265_HEX_SEARCH and X265_DIA_SEARCH as well (`method`; encoder/motion.cpp:628-723): their first
 * hexagon ring + closing square, or first diamond round, are measured by the fixed-shape kernels like the first three
 * rounds of the star pattern, and the PUs whose ring / round moves the centre finish their walk one per wave. */
int x265amd_search_batch_pairs_family_method_dev(x265amd_ctx *ctx, const void *d_pairs, int npairs, int qp, int method, int subme,
                                                 const x265amd_job *d_parents, int nparents, x265amd_result *d_parent_results,
                                                 const x265amd_job *d_children, int nchildren, x265amd_result *d_child_results);

/* The checks x265amd_search_batch makes on host job arrays (partition of motion.cpp:119, numc, range, PU inside the
 * source plane, search window + 8-tap apron inside the padded reference: what xSetSearchRange, TEncSearch.cpp:4188-4215,
 * and the caller's margins guarantee in the reference), for a DEVICE-resident job list, which the *_dev entry points run
 * as it is: a caller that builds or edits jobs on the device checks the list once and launches it as often as it likes.
 * X265AMD_OK when every job may run; otherwise X265AMD_EINVAL, the message names the first offending job and why,
 * *n_bad (optional) is how many there are and *first_bad (optional) the first one's index. Synchronous; no counterpart
 * in the reference, which asserts. The pairs form takes the HOST pair array x265amd_pairs_upload was given. */
int x265amd_check_jobs_dev(x265amd_ctx *ctx, const x265amd_plane *fenc, const x265amd_plane *ref, int is_lowres, int method,
                           const x265amd_job *d_jobs, int njobs, int *n_bad, int *first_bad);
int x265amd_check_jobs_pairs_dev(x265amd_ctx *ctx, const x265amd_pair *pairs, int npairs, int method,
                                 const x265amd_job *d_jobs, int njobs, int *n_bad, int *first_bad);

/* X265_FULL_SEARCH (encoder/motion.cpp:1021-1056) scans each PU's search range from a copy of
 * the reference window in LDS; ranges too large for the window are scanned in strips. `bytes`
 * (8192 .. 131072 with 8-bit samples, 16384 .. 131072 with deeper ones; unset, the library takes 36864, or 49152 / 65536
 * for launches of 16-bit PUs up to 32x32 / 64x64) is the LDS one PU may use for that copy when
 * the jobs are device-resident (x265amd_search_batch sizes it from the jobs it is given, up to
 * this limit): smaller windows keep more PUs resident per compute unit. Tuning only; results do
 * not depend on it. */
int x265amd_set_esa_window_bytes(x265amd_ctx *ctx, int bytes);

/* X265_FULL_SEARCH (encoder/motion.cpp:1021-1056) of every square PU of whole 64x64 CTUs: the quadtree of
 * TEncCu::xCompressCU (Lib/TLibEncoder/TEncCu.cpp:2352) has the encoder search the 64x64 PU, its four 32x32, sixteen
 * 16x16 and sixty-four 8x8 PUs, and the SAD of a PU at a displacement is the sum of its 8x8 blocks' -- the reference's
 * hardware model shares them the same way (hardwareC/inter.cpp:2521-2624). One scan per CTU measures
 * the 8x8 level over the bounding box of the PUs' ranges and adds the sums up the quadtree, every PU keeping the first
 * minimum of its own range with its own predictor's lambda * bits; results equal x265amd_search_batch with X265_FULL_SEARCH
 * job for job (a QP whose cost table can wrap runs the per-size scans). It pays where the
 * predictors of a CTU's PUs lie close together (the scan covers the box around all ranges); a caller sends CTUs whose
 * ranges lie far apart through x265amd_search_batch_dev per size instead.
 * jobs / results: 85 nctu records -- the 64 nctu 8x8 PUs (CTU-major; inside a CTU in Morton order: bit 0 of the index
 * = bit 0 of the block column, bit 1 = bit 0 of the block row, and so on), then the 16 nctu 16x16 PUs (likewise), the
 * 4 nctu 32x32 PUs (raster order inside a CTU), the nctu 64x64 PUs. A slot with w = h = 0 holds no PU (CTUs the picture's
 * edge cuts) and gets cost -1; its pu_x / pu_y still name its place, and slot 0 of a CTU's 8x8 PUs is the CTU's origin.
 * The host form checks every slot (place, size, and what x265amd_search_batch checks); the device-resident form does not. */
int x265amd_esa_ctu_batch(x265amd_ctx *ctx, const x265amd_plane *fenc, const x265amd_plane *ref, int qp, int subme,
                          const x265amd_job *jobs, int nctu, x265amd_result *results);
int x265amd_esa_ctu_batch_dev(x265amd_ctx *ctx, const x265amd_plane *fenc, const x265amd_plane *ref, int qp, int subme,
                              const x265amd_job *d_jobs, int nctu, x265amd_result *d_results);
/* The same over several (source, reference) pairs in one launch (x265amd_pairs_upload): slot 0 of a CTU's 8x8 PUs names
 * the CTU's pair in job.reserved, and so does every slot that holds a PU. */
int x265amd_esa_ctu_batch_pairs_dev(x265amd_ctx *ctx, const void *d_pairs, int npairs, int qp, int subme,
                                    const x265amd_job *d_jobs, int nctu, x265amd_result *d_results);

/* MotionEstimate::subpelCompare (encoder/motion.cpp:1651-1689) for njobs PUs:
 * job.qmvp is the qpel MV to evaluate, cmp 0 = sad, 1 = satd; results[i].cost. */
int x265amd_subpel_compare_batch(x265amd_ctx *ctx, const x265amd_plane *fenc, const x265amd_plane *ref,
                                 int cmp, const x265amd_job *jobs, int njobs, x265amd_result *results);

/* ---- lookahead ---- */

/* The motion-search part of EstimateRow::estimateCUCost over a whole lowres frame
 * for one reference list (encoder/slicetype.cpp:1401-1470 driven by
 * CostEstimate::processRow :1342): 8x8 CUs in reverse raster order, each predicted
 * from its right / below / below-left / below-right neighbours' final MVs.
 * Fills Lowres::lowresMvs[list][dist] and lowresMvCosts[list][dist]
 * (common/lowres.h:124-125). nfields independent fields run in one launch.
 * out_mvs: nfields * ncu * 2 int16, out_costs: nfields * ncu int32 (HOST). */
typedef struct
{
    x265amd_plane fenc;        /* lowresPlane[0] of the frame being searched */
    x265amd_plane ref[4];      /* the four hpel planes of the reference frame (one stride) */
} x265amd_field;
/* Geometry checked before anything is launched (X265AMD_EINVAL otherwise): every plane covers the CU grid, and the
 * reference planes have margins of at least 12 samples -- the range reaches 8 samples outside the grid
 * (slicetype.cpp:1427-1430), hexagon / square steps 3 more without a range check, the quarter-pel average one more.
 * X265AMD_UMH reads much further (motion.cpp:732, :301-303: its cross is measured from the start block, the vertical one
 * at x = omv.x + omv.y) and needs margin_x >= 32 + 8 * (height_in_cu - 1) + 6 * merange / 4 + 6 and
 * margin_y >= 30 + 6 * merange / 4; the reference's own lookahead only ever uses X265_HEX_SEARCH (slicetype.h:65-72).
 * Performance: a field's latency is its dependency chain of width_in_cu + 2 * (height_in_cu - 1) CU searches however many
 * fields a launch holds (about 1 ms for a 1080p field), so batch fields: launches of up to ~1000 waves (nfields *
 * height_in_cu / 2) run a form of the kernel with shorter steps, larger ones the form with the higher throughput. */
int x265amd_lookahead_fields(x265amd_ctx *ctx, const x265amd_field *fields, int nfields,
                             int width_in_cu, int height_in_cu, int qp, int method, int subme, int merange,
                             int16_t *out_mvs, int32_t *out_costs, x265amd_result *out_stats /* nfields*ncu or NULL */);

/* Device-resident variant: `d_fields` comes from x265amd_fields_upload (release it with
 * x265amd_device_free), outputs are DEVICE arrays of the sizes above; asynchronous on
 * the engine's stream. Same reference path as x265amd_lookahead_fields. x265amd_fields_upload makes the
 * geometry checks above for the grid, method and merange it is given and the context remembers them with the
 * array: x265amd_lookahead_fields_dev takes such an array of the same context (or a run of its fields: d_fields
 * + k * 80 bytes) and returns X265AMD_EINVAL for any other pointer, another grid, a larger merange, or UMH on an
 * array that was not checked for UMH. */
int x265amd_fields_upload(x265amd_ctx *ctx, const x265amd_field *fields, int nfields,
                          int width_in_cu, int height_in_cu, int method, int merange, void **d_fields);
int x265amd_device_free(x265amd_ctx *ctx, void *d_ptr);
/* Free and total memory of the context's GPU as the driver reports them (hipMemGetInfo); no reference counterpart: for
 * callers that size their resident picture sets, and for the test that x265amd_lookahead_fields_static takes none. */
int x265amd_device_mem_info(x265amd_ctx *ctx, size_t *free_bytes, size_t *total_bytes);
int x265amd_lookahead_fields_dev(x265amd_ctx *ctx, const void *d_fields, int nfields,
                                 int width_in_cu, int height_in_cu, int qp, int method, int subme, int merange,
                                 int16_t *d_mvs, int32_t *d_costs, x265amd_result *d_stats /* or NULL */);

/* x265amd_lookahead_fields for callers that cannot allocate device memory (tight serving loops, harnesses that forbid
 * hipMalloc). It has no counterpart in the reference. Same arguments and results as x265amd_lookahead_fields, but the
 * call allocates and frees no device memory (no hipMalloc / hipFree, here or in a library it calls): the field
 * descriptors, the MV / cost / statistics arrays the kernel writes and the row ticket of the kernel sit in a static device
 * array of X265AMD_STATIC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends. It is a
 * synchronous host call like its twin -- it copies from and to the caller's host arrays and waits for its stream -- so it
 * is not for use under stream capture; error paths build their message on the host heap. Limits, refused with
 * X265AMD_EINVAL: at most X265AMD_STATIC_MAX_FIELDS fields, and
 *   256 + 80 * nfields (rounded up to 256) + nfields * width_in_cu * height_in_cu * (8, or 24 with out_stats)
 * bytes must fit the array (e.g. 120 fields of a 1080p frame, 7 of an 8K frame). The cost table of `qp` must have been
 * built before by x265amd_set_qp (that call allocates; this one returns X265AMD_EINVAL if the table is missing).
 * All calls on one GPU share that GPU's array, one at a time: a call that finds another one (of any context or thread of
 * the process) still running on the same GPU returns X265AMD_EBUSY and touches nothing. */
#define X265AMD_STATIC_SCRATCH_BYTES (8 << 20)
#define X265AMD_STATIC_MAX_FIELDS 256
int x265amd_lookahead_fields_static(x265amd_ctx *ctx, const x265amd_field *fields, int nfields,
                                    int width_in_cu, int height_in_cu, int qp, int method, int subme, int merange,
                                    int16_t *out_mvs, int32_t *out_costs, x265amd_result *out_stats /* nfields*ncu or NULL */);

/* ---- lookahead frame cost ---- */

/* Everything EstimateRow::estimateCUCost computes for the 8x8 CUs of frame b besides the
 * searches (encoder/slicetype.cpp:1471-1581: the two bidirectional candidates through
 * ReferencePlanes::lowresMC, common/lowres.h:56-76; the 35-mode intra estimate; the list /
 * intra choice) and the sums CostEstimate::estimateFrameCost forms from its rows
 * (:1193-1204), for p0 = b - d0 (list 0) and p1 = b + d1 (list 1). d0 = d1 = 0 is the
 * intra-only call of :1145. The searches come first: mvs* / mv_costs* are the fields
 * x265amd_lookahead_fields_dev wrote for (b, p0) and (b, p1) -- Lowres::lowresMvs[l][d - 1]
 * and lowresMvCosts[l][d - 1] (common/lowres.h:124-125) -- whether searched now or kept from
 * an earlier call (bDoSearch, :1150-1151). All arrays are DEVICE memory; ncu = wcu * hcu.
 *   intra_calculated  Lowres::bIntraCalculated: 0 = compute intra_cost and row_satds_intra
 *                     here (8-bit pictures only: the reference gathers its intra edges with
 *                     byte counts, :1503-1512, so its result above 8 bits is undefined),
 *                     1 = intra_cost holds an earlier call's result
 *   inv_qscale        Lowres::invQscaleFactor or NULL (AQ off)
 *   intra_cost        Lowres::intraCost[ncu]
 *   lowres_costs      Lowres::lowresCosts[d0][d1][ncu]: min(cost, 16383) | lists used << 14
 *   row_satds_intra   Lowres::rowSatds[0][0][hcu]
 *   row_satds         Lowres::rowSatds[d0][d1][hcu] (unused when d0 = d1 = 0)
 *   totals[5]         sums over the rows of costEst, costEstAq, costIntra, costIntraAq,
 *                     intraMbs; the caller applies the B-frame scaling of :1208-1209 */
typedef struct
{
    x265amd_plane fenc;        /* lowresPlane[0] of frames[b] */
    x265amd_plane ref0[4];     /* hpel planes of frames[p0]; ignored when d0 == 0 */
    x265amd_plane ref1[4];     /* hpel planes of frames[p1]; ignored when d1 == 0 */
    int32_t d0, d1;
    int32_t intra_calculated;
    int32_t reserved;
    const int16_t *mvs0; const int32_t *mv_costs0;
    const int16_t *mvs1; const int32_t *mv_costs1;
    const int32_t *inv_qscale;
    int32_t  *intra_cost;
    uint16_t *lowres_costs;
    int32_t  *row_satds_intra;
    int32_t  *row_satds;
    int64_t  *totals;
} x265amd_frame_cost;

/* validates `nframes` descriptors and copies them to the device (release with
 * x265amd_device_free); x265amd_lookahead_frame_costs_dev then evaluates them all in one
 * go, asynchronously on the engine's stream (after the searches queued before it) */
int x265amd_frame_costs_upload(x265amd_ctx *ctx, const x265amd_frame_cost *frames, int nframes,
                               int width_in_cu, int height_in_cu, void **d_frames);
int x265amd_lookahead_frame_costs_dev(x265amd_ctx *ctx, const void *d_frames, int nframes,
                                      int width_in_cu, int height_in_cu);

/* zero-filled device memory for the arrays above (release with x265amd_device_free), and a
 * host-to-device copy that completes before it returns; stand for the buffers the encoder's
 * Lowres::create allocates (common/lowres.cpp:59-96) */
int x265amd_device_alloc(x265amd_ctx *ctx, size_t bytes, void **d_ptr);
int x265amd_upload(x265amd_ctx *ctx, void *d_dst, const void *src, size_t bytes);

/* out[i] = the first 8 bytes of results[i] (x265amd_mv mv; int32 cost): what motionEstimate hands back to its caller
 * (outQMv and the return value, encoder/motion.cpp:549, 1144-1145) without the two visit counters, packed for the
 * copy to the host or the gather to the rank that feeds the encoder. Device pointers; asynchronous on the engine's
 * stream, after the searches launched before it. */
int x265amd_pack_mv_cost(x265amd_ctx *ctx, const void *d_results, int n, void *d_out);
/* The same with the records picked by index: out[i] = the MV + cost half of record d_index[i] (a launch whose list order
 * is not the order the host wants the records in, e.g. the CTU order of x265amd_esa_ctu_batch_dev). */
int x265amd_pack_mv_cost_indexed(x265amd_ctx *ctx, const void *d_results, const int32_t *d_index, int n, void *d_out);

/* Measurement aid, no counterpart in the reference: the shader clock while the chip is busy. _start queues one wave on
 * the engine's stream that sleeps for `ms` milliseconds of the device's constant-rate clock and counts shader cycles
 * meanwhile (launch it on a stream of its own next to the work of interest); _read waits for it and gives the average
 * shader clock over that time in MHz. The issue-rate bounds reported next to the kernels (VALU instructions x 4 cycles per
 * SIMD) are priced at this clock, which a loaded MI355X holds well below its nominal 2.4 GHz. */
int x265amd_clock_probe_start(x265amd_ctx *ctx, double ms);
int x265amd_clock_probe_read(x265amd_ctx *ctx, double *shader_mhz);

/* ---- explicit weighted prediction ---- */

/* primitives.weight_pp (common/pixel.cpp:512-529) over a whole padded buffer, the way the
 * lookahead weights Lowres::buffer[i] end to end (encoder/slicetype.cpp:1333-1337): every sample
 * becomes clip(((w0 * (int16)(v << (14 - depth)) + round) >> shift) + offset). If dst->origin
 * is NULL the output plane is allocated here with `src`'s geometry (release with
 * x265amd_plane_free); otherwise `dst` must be another plane of that geometry and is overwritten
 * (the reference keeps one weight buffer per reference and reuses it). The same holds for
 * x265amd_apply_weight and for the planes of x265amd_weights_analyse. Asynchronous on the
 * engine's stream. */
int x265amd_weight_buffer(x265amd_ctx *ctx, const x265amd_plane *src, int w0, int round, int shift, int offset,
                          x265amd_plane *dst);

/* MotionReference::init(pic, wp) + applyWeight over all CTU rows (encoder/reference.cpp:39-71,
 * :211-256): the weighted copy of a reconstructed reference that motionEstimate then searches
 * (weight = wp->inputWeight, shift = log2WeightDenom, offset = inputOffset << (depth - 8)), with
 * its borders replicated from the weighted picture. Asynchronous. */
int x265amd_apply_weight(x265amd_ctx *ctx, const x265amd_plane *recon, int input_weight, int log2_denom,
                         int input_offset, x265amd_plane *dst);

/* CostEstimate::weightCostLuma (encoder/slicetype.cpp:1223-1256): sum over the 8x8 CUs of
 * min(satd(src block [weighted by (scale, denom, offset) when use_wp], fenc block), intraCost).
 * fenc / src: lowres full-pel planes; d_intra_cost: DEVICE Lowres::intraCost; *cost: HOST. */
int x265amd_lookahead_weight_cost(x265amd_ctx *ctx, const x265amd_plane *fenc, const x265amd_plane *src,
                                  const int32_t *d_intra_cost, int width_in_cu, int height_in_cu,
                                  int use_wp, int scale, int denom, int offset, uint32_t *cost);

/* CostEstimate::weightsAnalyse (encoder/slicetype.cpp:1258-1340) for frame b (fenc = its
 * lowresPlane[0]) against frame p0 (ref = its four hpel planes). stats = { fenc wp_ssd[0], fenc
 * wp_sum[0], ref wp_ssd[0], ref wp_sum[0] } (encoder/ratecontrol.cpp:44-56, :185-192). When a
 * weight is chosen (out->weighted, CostEstimate::weightedRef.isWeighted) `weighted` receives the
 * four weighted planes (CostEstimate::wbuffer), to be passed as the list-0 reference of the
 * lookahead searches and frame cost of (b, p0). Synchronous (two cost evaluations are read back). */
typedef struct
{
    int32_t  weighted;          /* 0: the reference is used as it is */
    int32_t  scale, denom, offset;  /* w.inputWeight, w.log2WeightDenom, w.inputOffset */
    uint32_t orig_score, min_score;
} x265amd_weight;
int x265amd_weights_analyse(x265amd_ctx *ctx, const x265amd_plane *fenc, const x265amd_plane ref[4],
                            const int32_t *d_intra_cost, int width_in_cu, int height_in_cu,
                            const uint64_t stats[4], x265amd_weight *out, x265amd_plane weighted[4]);

/* ---- bi-prediction estimate ---- */

/* The distortion part of the bi-directional check of TEncSearch::predInterSearch
 * (Lib/TLibEncoder/TEncSearch.cpp:4346-4368) for a batch of PUs: each PU is predicted from both
 * references at its best list-0 / list-1 quarter-pel MVs (TComPrediction::xPredInterLumaBlk,
 * Lib/TLibCommon/TComPrediction.cpp:405-438), the predictions are averaged (pixelavg_pp) and
 * compared with the source by SATD; satd_zero is the same for the two co-located blocks
 * (:4362-4368). The bit costs around it (:4359-4360, :4370-4378) stay with the caller. */
typedef struct
{
    int16_t pu_x, pu_y;
    uint8_t w, h;              /* an HEVC luma partition */
    uint8_t reserved[2];
    int16_t mv0[2], mv1[2];    /* quarter-pel MVs into ref0 / ref1 */
} x265amd_bipred_job;
typedef struct { int32_t satd, satd_zero; } x265amd_bipred_result;
/* jobs / results are HOST arrays */
int x265amd_bipred_batch(x265amd_ctx *ctx, const x265amd_plane *fenc, const x265amd_plane *ref0, const x265amd_plane *ref1,
                         const x265amd_bipred_job *jobs, int njobs, x265amd_bipred_result *results);
/* device-resident, asynchronous on the engine's stream; the jobs are not validated */
int x265amd_bipred_batch_dev(x265amd_ctx *ctx, const x265amd_plane *fenc, const x265amd_plane *ref0, const x265amd_plane *ref1,
                             const x265amd_bipred_job *d_jobs, int njobs, x265amd_bipred_result *d_results);

/* The same estimate for a list of PUs that is checked and laid out once and launched asynchronously on the engine's
 * stream (x265amd_bipred_batch_dev cannot see the PU sizes and runs one wave per PU; a list knows them and fills its
 * waves with 4x4 tiles of neighbouring PUs). d_results0 / d_results1, when not NULL, are device-resident outputs of
 * x265amd_search_batch_dev for the same PUs in the same order, launched earlier on the stream: PU i then takes its
 * list-0 / list-1 MV from them instead of from its job record (predInterSearch's sequence, TEncSearch.cpp:4303 then
 * :4346-4368, without a host round trip; such MVs are not validated). d_out: njobs device records. */
typedef struct x265amd_bipred_list x265amd_bipred_list;
int x265amd_bipred_list_create(x265amd_ctx *ctx, const x265amd_plane *fenc, const x265amd_plane *ref0, const x265amd_plane *ref1,
                               const x265amd_bipred_job *jobs, int njobs, x265amd_bipred_list **out);
int x265amd_bipred_list_free(x265amd_ctx *ctx, x265amd_bipred_list *list);
int x265amd_bipred_batch_list(x265amd_ctx *ctx, const x265amd_plane *fenc, const x265amd_plane *ref0, const x265amd_plane *ref1,
                              const x265amd_bipred_list *list, const x265amd_result *d_results0, const x265amd_result *d_results1,
                              x265amd_bipred_result *d_out);

/* ---- merge-candidate estimate ---- */

/* The distortion of the merge candidates of a 2Nx2N CU (encoder/compress.cpp:248-271): luma motion
 * compensation of the candidate (TComPrediction::motionCompensation with bLuma only,
 * Lib/TLibCommon/TComPrediction.cpp:214-283 -- dir 1 / 2: one list through xPredInterLumaBlk's
 * pixel path, :405-438; dir 3: both lists through its 14-bit path, :441-476, and TComYuv::addAvg,
 * Lib/TLibCommon/TComYuv.cpp:573-612) and primitives.sa8d against the source CU (:265). The
 * caller has clipped the MVs (TComDataCU::clipMv) and turned a candidate whose two lists are the
 * same picture and MV into dir 1 (xCheckIdenticalMotion, TComPrediction.cpp:276). Weighted
 * prediction flavours (:252-268) are not covered here (x265amd_mc_batch has them).
 * The distortion of an inter mode whose prediction already exists (TEncCu::xComputeCostInter,
 * encoder/compress.cpp:209-212: sa8d of the source CU against predYuv) is the same call with the
 * prediction picture's luma plane as the only reference, dir 1 and MV (0, 0): a whole-sample MV of a
 * single list reads exactly the block, so the plane needs no margins. */
typedef struct
{
    int16_t pu_x, pu_y;
    uint8_t size;              /* 8, 16, 32 or 64 */
    uint8_t dir;               /* interDir: 1 = list 0, 2 = list 1, 3 = both */
    uint8_t ref0, ref1;        /* indices into `refs` */
    int16_t mv0[2], mv1[2];    /* quarter-pel */
} x265amd_merge_job;
/* refs: nrefs (<= 32) reconstructed reference planes; jobs / sa8d: HOST arrays */
int x265amd_merge_sa8d_batch(x265amd_ctx *ctx, const x265amd_plane *fenc, const x265amd_plane *refs, int nrefs,
                             const x265amd_merge_job *jobs, int njobs, int32_t *sa8d);

/* ---- motion compensation ---- */

/* TComPrediction::motionCompensation without weighted prediction (Lib/TLibCommon/TComPrediction.cpp:
 * 201-286) for a batch of PUs, luma and 4:2:0 chroma, written into a prediction picture at each PU's
 * own position (the reference writes the CU-local TComYuv predYuv; one picture-sized buffer stands
 * for all of them). dir 1 / 2: one list through the pixel flavours of xPredInterLumaBlk (:405-438)
 * and xPredInterChromaBlk (:491-540); dir 3: both lists through the 14-bit flavours (:441-478,
 * :543-593) and TComYuv::addAvg (Lib/TLibCommon/TComYuv.cpp:573-650). flags bit 0 = bLuma, bit 1 =
 * bChroma. The luma quarter-pel MV addresses chroma in eighth-pel units (:496, :504-505). The caller
 * has clipped the MVs (TComDataCU::clipMv) and turned identical motion of both lists into dir 1
 * (xCheckIdenticalMotion, :183-198). PUs of one call must not overlap (they never do inside a frame's
 * final partitioning).
 * flags bit 2 = explicit weighted prediction for this PU (PPS getUseWP / getWPBiPred): every used list
 * goes through the 14-bit flavour, then xWeightedPredictionUni (:212-226, :357-376; addWeightUni with
 * weight_sp, Lib/TLibCommon/TComWeightPrediction.cpp:433-485) or xWeightedPredictionBi (:334-356;
 * addWeightBi, TComWeightPrediction.cpp:203-313) with the parameters getWpScaling derives (:534-560)
 * from the `weights` entry of each reference picture. */
typedef struct
{
    int16_t pu_x, pu_y;        /* luma position, a multiple of 4 */
    uint8_t w, h;              /* an HEVC luma partition, not 4x4 */
    uint8_t dir;               /* 1 = list 0, 2 = list 1, 3 = both */
    uint8_t flags;             /* 1 = luma, 2 = chroma, 3 = both; + 4 = weighted */
    uint8_t ref0, ref1;        /* indices into `refs` */
    uint8_t reserved[2];
    int16_t mv0[2], mv1[2];    /* quarter-pel */
} x265amd_mc_job;
/* A 4:2:0 picture resident in device memory (TComPicYuv): cb / cr are half size in both directions */
typedef struct { x265amd_plane y, cb, cr; } x265amd_picture;
/* wpScalingParam of one reference picture per component Y, Cb, Cr (Lib/TLibCommon/TComSlice.h:
 * inputWeight, inputOffset at 8-bit scale, log2WeightDenom 0..7). A picture used with different weights
 * in the two lists is listed twice in `refs`. */
typedef struct
{
    int16_t w[3], offset[3];
    uint8_t log2denom[3], reserved;
} x265amd_mc_weight;
/* refs: nrefs (<= 32) reconstructed reference pictures; weights: nrefs entries, or NULL when no job is
 * weighted; pred: the output picture (margins may be 0); jobs: HOST array. Returns when the prediction
 * samples are in `pred`. */
int x265amd_mc_batch(x265amd_ctx *ctx, const x265amd_picture *refs, const x265amd_mc_weight *weights, int nrefs,
                     const x265amd_picture *pred, const x265amd_mc_job *jobs, int njobs);
/* Device-resident variant, asynchronous on the engine's stream. x265amd_pictures_upload puts the plane
 * descriptors and weights of `refs` on the device (free with x265amd_device_free); x265amd_mc_list_create checks the
 * jobs as x265amd_mc_batch does, numbers their 4x4 tiles for the launch and uploads both (a list can be
 * launched any number of times, e.g. one list per frame); x265amd_mc_batch_dev only launches: the pictures it is
 * given must have the geometry the list was checked against. */
typedef struct x265amd_mc_list x265amd_mc_list;
int x265amd_pictures_upload(x265amd_ctx *ctx, const x265amd_picture *refs, const x265amd_mc_weight *weights, int nrefs,
                            void **d_ref_planes);
int x265amd_mc_list_create(x265amd_ctx *ctx, const x265amd_picture *refs, const x265amd_mc_weight *weights, int nrefs,
                           const x265amd_picture *pred, const x265amd_mc_job *jobs, int njobs, x265amd_mc_list **out);
int x265amd_mc_list_free(x265amd_ctx *ctx, x265amd_mc_list *list);
int x265amd_mc_batch_dev(x265amd_ctx *ctx, const void *d_ref_planes, const x265amd_picture *pred, const x265amd_mc_list *list);
/* Prediction straight from search results: as x265amd_mc_batch_dev, but job i takes its list-0 MV from
 * d_results0[i].mv and its list-1 MV from d_results1[i].mv -- device-resident outputs of
 * x265amd_search_batch_dev for the same PUs in the same order, launched earlier on the engine's stream --
 * instead of from the job record (the encoder's motionEstimate -> motionCompensation sequence,
 * Lib/TLibEncoder/TEncSearch.cpp:4303 then :4790-4803, without a host round trip). Either pointer may be
 * NULL: that list keeps the job's MV. These MVs are not validated; the search ranges must keep the 8-tap /
 * 4-tap footprints inside the padded reference planes. */
int x265amd_mc_batch_dev_mv(x265amd_ctx *ctx, const void *d_ref_planes, const x265amd_picture *pred, const x265amd_mc_list *list,
                            const x265amd_result *d_results0, const x265amd_result *d_results1);

/* The launch plan behind x265amd_mc_*, x265amd_bipred_batch / _list and x265amd_merge_sa8d_batch, as a pure host
 * function (no counterpart in the reference; exported so that it can be checked without a GPU): the 4x4 tiles of n
 * records with tiles[i] tiles each are numbered consecutively; first[i] (n + 64 entries, the tail repeating the total)
 * is the first tile of record i, chunk_job[k] (*nchunks entries, capacity total / 64 + n / 4 + 2 is always enough) the
 * record that tile 64 k belongs to. With align16 a record starts at a multiple of min(tiles[i], 16) and the tiles
 * skipped belong to the record before. Needs no context; returns X265AMD_EINVAL for a tile count below 1. */
int x265amd_tile_plan(const int32_t *tiles, int n, int align16, int32_t *first, int32_t *chunk_job, int *nchunks);

/* ---- Rockchip path (rd >= 5): CIME + RIME + FME ---- */

/* One 64x64 CTU against one reference picture: everything the 15-argument
 * MotionEstimate::motionEstimate overload computes for the CTU's 85 CUs
 * (encoder/motion.cpp:1250-1410, caller TEncSearch.cpp:3704-3822; hardware C-model
 * twins hardwareC/inter.cpp:679-734 CIME and :2521-2624 RimeAndFme, which
 * hardwareC.cpp:69-155 asserts equal). The overload keeps the CTU's PMVs in
 * function-static storage; here they are explicit:
 *   pmv[0..1]  = Pmv[ref][1..2], the de-duplicated neighbour PMVs (full-pel, even) that
 *                centre RIME windows 1 and 2 (window 0 is the CIME result);
 *   mvp[0..2]  = mvNeighBor[7..9], the predictors the MVD cost is taken against
 *                (TEncSearch.cpp:3788-3805); have_mvd = isHaveMvd.
 * mvmin/mvmax come from the RK xSetSearchRange (TEncSearch.cpp:4188-4215);
 * valid_w/h = nValidCtuWidth/Height (:3811-3818). */
typedef struct
{
    int16_t ctu_x, ctu_y;
    int16_t valid_w, valid_h;
    int16_t mvmin[2], mvmax[2];
    int16_t pmv[2][2];
    int16_t mvp[3][2];
    uint8_t have_mvd, reserved[3];
} x265amd_rk_ctu;

typedef struct { int16_t mv[2]; int32_t cost; } x265amd_rk_mv;

/* cime = g_leftPMV (full-pel), rime[i] = g_RimeMv[ref][i] (full-pel), fme[i] = g_FmeMv[ref][i]
 * = outQMv / return value (quarter-pel), i = offsIdx of OffsFromCtu64 (hardwareC/inter.cpp:23-38):
 * 0-63 8x8, 64-79 16x16, 80-83 32x32, 84 64x64. CUs outside the picture have cost -1
 * (TEncCu::isOutPic, TEncCu.cpp:475-493). */
typedef struct
{
    x265amd_rk_mv cime;
    x265amd_rk_mv rime[85];
    x265amd_rk_mv fme[85];
} x265amd_rk_result;

/* MotionReference::init of the RK build (encoder/reference.cpp:161-178): the 4x4-sum
 * plane of an ORIGINAL picture (downSampleAndSum :143) with replicated borders
 * (extendEdge :73). `out` describes an int16 plane of (width/4 x height/4) samples with
 * margins margin/4; `orig` must have margin_x == margin_y. */
int x265amd_rk_ds_plane(x265amd_ctx *ctx, const x265amd_plane *orig, x265amd_plane *out);

/* fenc: source picture; recon: reconstructed reference (fpelPlane); ds: x265amd_rk_ds_plane
 * of the reference's original picture (fpelPlaneOrigDS). ctus / results are HOST arrays.
 * Contexts of more than 10 bits per sample are refused (X265AMD_EINVAL): the 4x4 sums and the
 * 8x8 block SADs of this path are 16-bit quantities, as the reference's `short` buffers are. */
int x265amd_rk_ctu_batch(x265amd_ctx *ctx, const x265amd_plane *fenc, const x265amd_plane *recon,
                         const x265amd_plane *ds, int qp, const x265amd_rk_ctu *ctus, int nctus,
                         x265amd_rk_result *results);
/* device-resident, asynchronous on the engine's stream */
int x265amd_rk_ctu_batch_dev(x265amd_ctx *ctx, const x265amd_plane *fenc, const x265amd_plane *recon,
                             const x265amd_plane *ds, int qp, const x265amd_rk_ctu *d_ctus, int nctus,
                             x265amd_rk_result *d_results);

/* ---- primitives table ---- */

/* Mirror of the ME-relevant slice of struct EncoderPrimitives
 * (common/primitives.h:209-288) with the reference's typedefs (:145-204). All
 * pointers are HOST pointers; every call stages its operands on the device, runs
 * the HIP kernel and returns the result, so the table can be checked call by call
 * against the C primitives the way test/pixelharness.cpp and
 * test/ipfilterharness.cpp do. Indexed by the LumaPartitions enum (:65-73).
 * `pixel` is uint8_t or uint16_t according to the context's bit depth.
 * One table is live per process: its entries serve the context of the latest x265amd_setup_primitives call
 * (depth, stream, staging buffer), are not thread-safe, and abort with a message once that context has been
 * destroyed. */
#define X265AMD_NUM_LUMA_PARTITIONS 25
typedef int  (*x265amd_pixelcmp_t)(const void *fenc, intptr_t fencstride, const void *fref, intptr_t frefstride);
typedef void (*x265amd_pixelcmp_x3_t)(const void *fenc, const void *fref0, const void *fref1, const void *fref2,
                                      intptr_t frefstride, int32_t *res);
typedef void (*x265amd_pixelcmp_x4_t)(const void *fenc, const void *fref0, const void *fref1, const void *fref2,
                                      const void *fref3, intptr_t frefstride, int32_t *res);
typedef void (*x265amd_filter_pp_t)(const void *src, intptr_t srcStride, void *dst, intptr_t dstStride, int coeffIdx);
typedef void (*x265amd_filter_hps_t)(const void *src, intptr_t srcStride, int16_t *dst, intptr_t dstStride,
                                     int coeffIdx, int isRowExt);
typedef void (*x265amd_filter_sp_t)(const int16_t *src, intptr_t srcStride, void *dst, intptr_t dstStride, int coeffIdx);
typedef void (*x265amd_pixelavg_pp_t)(void *dst, intptr_t dstride, const void *src0, intptr_t sstride0,
                                      const void *src1, intptr_t sstride1, int weight);
typedef void (*x265amd_downscale_t)(const void *src0, void *dstf, void *dsth, void *dstv, void *dstc,
                                    intptr_t src_stride, intptr_t dst_stride, int width, int height);
typedef struct
{
    x265amd_pixelcmp_t    sad[X265AMD_NUM_LUMA_PARTITIONS];
    x265amd_pixelcmp_x3_t sad_x3[X265AMD_NUM_LUMA_PARTITIONS];
    x265amd_pixelcmp_x4_t sad_x4[X265AMD_NUM_LUMA_PARTITIONS];
    x265amd_pixelcmp_t    satd[X265AMD_NUM_LUMA_PARTITIONS];
    x265amd_pixelcmp_t    sa8d_inter[X265AMD_NUM_LUMA_PARTITIONS];
    x265amd_filter_pp_t   luma_hpp[X265AMD_NUM_LUMA_PARTITIONS];
    x265amd_filter_hps_t  luma_hps[X265AMD_NUM_LUMA_PARTITIONS];
    x265amd_filter_pp_t   luma_vpp[X265AMD_NUM_LUMA_PARTITIONS];
    x265amd_filter_sp_t   luma_vsp[X265AMD_NUM_LUMA_PARTITIONS];
    x265amd_pixelavg_pp_t pixelavg_pp[X265AMD_NUM_LUMA_PARTITIONS];
    x265amd_downscale_t   frame_init_lowres_core;
} x265amd_primitives;

/* x265_setup_primitives (x265.h:665, common/primitives.cpp:82-149): fills `p` with
 * entry points served by `ctx` (which becomes the process-wide engine for the
 * table, as the reference's `primitives` global is process-wide). */
int x265amd_setup_primitives(x265amd_ctx *ctx, x265amd_primitives *p);

/* partitionFromSizes (common/primitives.h:134-143); -1 if not a luma partition */
int x265amd_partition_from_sizes(int width, int height);

#ifdef __cplusplus
}
#endif
#endif /* X265_AMD_ME_H */
