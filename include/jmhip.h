/*
 * jmhip.h -- C ABI of libjmhip.so: the JM lencod per-macroblock hot path on MI355X (gfx950).
 *
 * Boundary: plain C, pointers and sizes only. Every entry point names the reference interface it
 * replaces (paths relative to the reference repo's lencod/). The library is HIP-only: there is no CPU
 * fallback, a missing device or a failed launch is an error code (never different numerics).
 *
 * Model. JM runs the path one call per block under a serial macroblock loop; the device runs it one
 * launch per FRAME stage. A context owns device-resident pictures:
 *   reference slots  -- integer-pel recon + the 16 quarter-pel luma planes + the 8x8 (4:2:0) / 4x8 (4:2:2)
 *                       / 4x4 (4:4:4) eighth-pel chroma planes, exactly the arrays JM's StorablePicture
 *                       holds after UnifiedOneForthPix (src/image.c:1601), as 8-bit samples;
 *   the current picture (source samples);
 *   per-macroblock job/result arrays of the stage calls below.
 * Calls enqueue on the context's HIP stream and return; jmhip_sync() (or any download) completes them.
 * From JM's single thread every exported function is therefore synchronous once followed by a download.
 *
 * Sample type at the boundary: JM's `imgpel` is unsigned short even for 8-bit video
 * (inc/global.h:46-52). Upload/download take either 1- or 2-byte samples (pel_bytes); on the device
 * 8-bit video is stored packed (BitDepth > 8 is rejected with JMHIP_ERR_UNSUPPORTED in this version).
 */
#ifndef JMHIP_H
#define JMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JMHIP_ABI_VERSION 1

/* One context per GPU and per encoder instance. A context is NOT thread-safe (JM is single-threaded and non-reentrant; every
 * call is issued from the thread that owns the encoder); different contexts may be used from different threads or processes.
 * Calls are asynchronous on the context's stream unless they return data to the host; errors are returned, never hidden:
 * there is no CPU fallback anywhere behind this interface. */
typedef struct jmhip_ctx jmhip_ctx;

enum {
  JMHIP_OK = 0,
  JMHIP_ERR_ARG = 1,          /* bad argument (NULL, size mismatch, out of range)            */
  JMHIP_ERR_DEVICE = 2,       /* HIP call failed / no gfx950 device                           */
  JMHIP_ERR_UNSUPPORTED = 3,  /* valid JM configuration this version does not implement       */
  JMHIP_ERR_NOMEM = 4
};

enum { JMHIP_YUV400 = 0, JMHIP_YUV420 = 1, JMHIP_YUV422 = 2, JMHIP_YUV444 = 3 };   /* img->yuv_format */

#define JMHIP_PAD 20                      /* IMG_PAD_SIZE, inc/defines.h:107                            */
#define JMHIP_NPART 41                    /* partitions of one MB over block types 1..7                 */

/* ------------------------------------------------------------------ context */

typedef struct {
  int device;              /* HIP device ordinal                                                     */
  int width, height;       /* coded luma size (multiples of 16; JM pads 1080 -> 1088, configfile.c:997) */
  int yuv_format;          /* JMHIP_YUV4xx                                                           */
  int bit_depth;           /* 8                                                                      */
  int max_refs;            /* number of reference slots                                              */
  int search_range;        /* input->search_range: sizes the per-MB search window                    */
} jmhip_config;

int  jmhip_ctx_create(const jmhip_config *cfg, jmhip_ctx **out);
void jmhip_ctx_destroy(jmhip_ctx *ctx);
int  jmhip_sync(jmhip_ctx *ctx);
/* The context's HIP stream (hipStream_t) for callers that enqueue their own work -- a collective, a copy -- in order with the
 * context's kernels instead of synchronising the host (bench.py wraps it as torch.cuda.ExternalStream for the RCCL all-gather). */
void *jmhip_stream_handle(jmhip_ctx *ctx);
const char *jmhip_last_error(jmhip_ctx *ctx);      /* text of the last failure on this context */
const char *jmhip_strerror(int code);
int  jmhip_abi_version(void);

/* Kernel timing on the context's stream (hipEvents recorded around each stage launch).
 * jmhip_timing_enable(ctx,1) makes every stage call record start/stop events; jmhip_timing_read returns
 * the accumulated milliseconds and launch count per stage since the last reset and resets them. */
enum { JMHIP_STAGE_INTERP_LUMA = 0, JMHIP_STAGE_INTERP_CHROMA, JMHIP_STAGE_ME_INT, JMHIP_STAGE_ME_SUB,
       JMHIP_STAGE_MC, JMHIP_STAGE_TQ, JMHIP_STAGE_DEBLOCK, JMHIP_STAGE_COUNT };
int jmhip_timing_enable(jmhip_ctx *ctx, int on);
int jmhip_timing_read(jmhip_ctx *ctx, double ms[JMHIP_STAGE_COUNT], int launches[JMHIP_STAGE_COUNT]);
/* Only the stages whose bit (1 << JMHIP_STAGE_x) is set record events while timing is on (default: all). */
int jmhip_timing_select(jmhip_ctx *ctx, unsigned stage_mask);

/* ------------------------------------------------------------------ pictures */

/* Upload the reconstructed integer-pel picture of a reference into slot `ref` (what JM holds in
 * StorablePicture.imgY / imgUV[2], inc/mbuffer.h:20-95). Strides in samples. U/V may be NULL for 4:0:0.
 * `device_ptrs` != 0: Y/U/V are device pointers (8-bit packed only), copied device-to-device. */
int jmhip_ref_upload(jmhip_ctx *ctx, int ref, const void *Y, const void *U, const void *V,
                     int pel_bytes, int stride_y, int stride_c, int device_ptrs);

/* getSubImagesLuma (src/img_luma.c:45, proto inc/img_luma.h:20): builds the 16 quarter-pel planes of slot
 * `ref` from its uploaded integer picture. Layout on device: plane (y&3)*4+(x&3), each
 * (height+40) x (width+40) samples, JM's imgY_sub[4][4][H+40][W+40]. */
int jmhip_interp_luma(jmhip_ctx *ctx, int ref);

/* getSubImagesChroma (src/img_chroma.c:374, proto inc/img_chroma.h:20): imgUV_sub[2][sy][sx][..][..]. */
int jmhip_interp_chroma(jmhip_ctx *ctx, int ref);
/* Luma AND chroma sub-pel planes restricted to the picture's luma rows [row0, row1) (slice-parallel ranks: own band +- search
 * reach; rows outside keep what an earlier call left). Values inside the range are identical to the full-plane calls'. */
int jmhip_interp_rows(jmhip_ctx *ctx, int ref, int row0, int row1);
/* The luma part of jmhip_interp_rows alone: the frame stage predicts chroma without the eighth-pel planes when they are not
 * built (jmhip_residual_frame). */
int jmhip_interp_luma_rows(jmhip_ctx *ctx, int ref, int row0, int row1);

/* Test/diagnostic: copy sub-pel planes back. `out` holds 16 (luma) or sub_y*sub_x (chroma, one component
 * uv = 0/1) planes of padded size, contiguous, pel_bytes per sample. */
int jmhip_ref_download_luma(jmhip_ctx *ctx, int ref, void *out, int pel_bytes);
int jmhip_ref_download_chroma(jmhip_ctx *ctx, int ref, int uv, void *out, int pel_bytes);
/* The same planes delivered the way JM stores them (imgY_sub[4][4][row], imgUV_sub[uv][sy][sx][row]: arrays of ROW pointers,
 * inc/global.h StorablePicture): rows[plane * padded_height + row] points to padded_width samples of pel_bytes each. This is what the
 * JM binding calls after jmhip_interp_luma / _chroma, because JM's own motion compensation (LumaPrediction, src/macroblock.c:836) reads
 * the planes on the host: packed bytes cross the link into a page-locked staging buffer plane by plane, and the arrived planes are
 * widened into the caller's rows while the rest are still in flight (by four host threads; JMHIP_HOST_THREADS=n sets another count, 1 = the
 * calling thread alone). A plane whose FIRST row pointer is NULL is skipped -- a binding that answers JM's prediction from the device only fetches
 * the planes a forwarded call is about to read. */
int jmhip_ref_download_luma_rows(jmhip_ctx *ctx, int ref, void *const *rows, int pel_bytes);
int jmhip_ref_download_chroma_rows(jmhip_ctx *ctx, int ref, int uv, void *const *rows, int pel_bytes);

/* Device addresses of a slot's integer-pel picture (for device-to-device exchange, e.g. the RCCL
 * all-gather of reconstructed slice bands): pitch in bytes. */
int jmhip_ref_device_planes(jmhip_ctx *ctx, int ref, void **Y, void **U, void **V, int *pitch_y, int *pitch_c);

/* Read-only view of a slot's picture planes (does not touch the slot's state), and a stream-ordered device-to-host copy: for
 * checks and checksums (bench.py's ref_checksum). */
int jmhip_ref_planes_peek(jmhip_ctx *ctx, int ref, void **Y, void **U, void **V, int *pitch_y, int *pitch_c);
int jmhip_copy_from_device(jmhip_ctx *ctx, const void *device_src, void *host_dst, size_t bytes);

/* Upload the current (source) picture: pCurImg / imgUV_org (inc/global.h). */
int jmhip_cur_upload(jmhip_ctx *ctx, const void *Y, const void *U, const void *V,
                     int pel_bytes, int stride_y, int stride_c, int device_ptrs);

/* Use the caller's DEVICE planes as the current picture without copying: 8-bit, tight pitch (width / chroma width), 4-byte
 * aligned, valid until the next jmhip_cur_upload / jmhip_cur_bind. U/V may be NULL for 4:0:0. */
int jmhip_cur_bind(jmhip_ctx *ctx, const void *Y, const void *U, const void *V);

/* ------------------------------------------------------------------ motion estimation */

/* Partition order inside a macroblock (index 0..40), JM's PartitionMotionSearch order
 * (src/mv-search.c:1378): type 1: 16x16; 2: 16x8 top,bottom; 3: 8x16 left,right; 4: 8x8 b8=0..3;
 * 5: 8x4 per b8 (top,bottom); 6: 4x8 per b8 (left,right); 7: 4x4 per b8 (raster in the 8x8).
 * jmhip_partition_info() returns blocktype and the 4x4-unit rectangle of partition p. */
void jmhip_partition_info(int p, int *blocktype, int *x4, int *y4, int *w4, int *h4);

enum { JMHIP_SEARCH_FULL = -1, JMHIP_SEARCH_FASTFULL = 0 };   /* input->SearchMode */

/* Slice/frame-level inputs of the search (JM: input->, img->, active_pps->). */
typedef struct {
  int search_mode;         /* JMHIP_SEARCH_FULL: FullPelBlockMotionSearch (src/me_fullsearch.c:47), centre per
                              partition from its own predictor (src/mv-search.c:752-762);
                              JMHIP_SEARCH_FASTFULL: FastFullPelBlockMotionSearch (src/me_fullfast.c:833), one
                              centre per MB from the 16x16 predictor (:550-566)                        */
  int search_range;        /* <= cfg.search_range (<= 64); beyond 44 the general kernel searches (64-bit keys) */
  int rdopt;               /* input->rdopt                                                           */
  int is_b_slice;          /* img->type == B_SLICE                                                   */
  int level_mv_min, level_mv_max;   /* LEVELMVLIMIT[img->LevelIndex][0..1], inc/mv-search.h:35-54    */
  int lambda[3];           /* lambda_factor[F_PEL,H_PEL,Q_PEL] (src/slice.c:1329-1358)                */
  int transform8x8_mode;   /* input->Transform8x8Mode: SATD uses the 8x8 Hadamard for block types <= 4 */
  int subpel;              /* !input->DisableSubpelME                                                 */
  uint64_t partition_mask; /* bit p set: partition p is searched (all 41: (1<<41)-1)                  */
  /* Weighted reference ME (input->UseWeightedReferenceME with explicit / implicit weights, src/mv-search.c:640-668): every reference
   * sample becomes iClip1(255, ((weight * p + wp_round) >> wp_denom) + offset) before SAD / SATD (computeSADWP src/me_distortion.c:413,
   * computeSATDWP :734, the weighted line of SetupFastFullPelSearch src/me_fullfast.c:600-640). Per reference SLOT: wp_weight[0] /
   * wp_offset[0] of JM's [list][ref][0] entry for the picture in that slot. wp_enable = 0: plain search. */
  int32_t wp_enable, wp_round, wp_denom;
  int16_t wp_weight[16], wp_offset[16];
  /* Error metrics and the chroma term. metric_set = 0: JM's defaults -- MEDistortionFPel / HPel / QPel = SAD / Hadamard SAD / Hadamard
   * SAD, luma only -- the fields below are ignored (the fast kernels). metric_set = 1: metric[F_PEL, H_PEL, Q_PEL] =
   * input->MEErrorMetric[] (0 SAD, 1 SSE, 2 Hadamard SAD; the computeUniPred dispatch of src/mv-search.c:400-424, incl. the carried
   * minimum when two levels share a metric, :396-397, and SetupFastFullPelSearch's squared error for every metric but SAD,
   * src/me_fullfast.c:512); chroma_me = input->ChromaMEEnable (0; 1: Cb / Cr term at integer positions; 2: at sub-pel positions
   * too; src/me_distortion.c:376-402, :1072-1098), chroma_me_weight = input->ChromaMEWeight: needs the current picture's chroma
   * planes and jmhip_interp_chroma on every reference used. With wp_enable the chroma term is weighted per slot and component
   * (wp_weight_cr[slot][uv], computeSADWP :443-470, computeSSEWP :1107). */
  int32_t metric_set, metric[3], chroma_me, chroma_me_weight;
  int32_t wp_chroma_round, wp_chroma_denom;
  int16_t wp_weight_cr[16][2], wp_offset_cr[16][2];
} jmhip_me_params;

/* Per-macroblock inputs. pred_mv: motion-vector predictor per partition in quarter-pel units, what
 * SetMotionVectorPredictor (src/mv-search.c:87) yields for that partition. ref: reference slot. */
typedef struct {
  int16_t mb_x, mb_y;                    /* macroblock position in MB units                          */
  int16_t ref;                           /* reference slot; ref_is_0 = (JM ref index == 0)            */
  int16_t ref_is_0;
  int16_t pred_mv[JMHIP_NPART][2];
} jmhip_me_mb;

typedef struct {
  int16_t mv[JMHIP_NPART][2];            /* final MV, quarter-pel (img->all_mv value)                 */
  int32_t cost[JMHIP_NPART];             /* min_mcost returned by BlockMotionSearch's search chain     */
  int16_t mv_int[JMHIP_NPART][2];        /* integer-search result in pel units (before << 2)          */
  int32_t cost_int[JMHIP_NPART];         /* min_mcost after the integer search                         */
} jmhip_me_result;

/* One frame stage: for every MB in `mbs` runs, per partition, the chain of BlockMotionSearch
 * (src/mv-search.c:560) from the predictor on: search centre, integer full search over
 * (2R+1)^2 candidates with SAD + MV_COST_SMP (inc/defines.h:128) and lowest-spiral-index tie-break,
 * then half- and quarter-pel refinement with SATD (src/me_fullsearch.c:341). Uses the current picture
 * and the reference slots. mbs/results are host arrays of n entries. */
int jmhip_me_frame(jmhip_ctx *ctx, const jmhip_me_params *prm, const jmhip_me_mb *mbs, int n,
                   jmhip_me_result *results);

/* Same, split for device-resident pipelines: enqueue, leave results on device. mbs == NULL re-runs the job array
 * the previous call uploaded (same n and search geometry): nothing crosses PCIe in steady state. */
int jmhip_me_frame_async(jmhip_ctx *ctx, const jmhip_me_params *prm, const jmhip_me_mb *mbs, int n);
int jmhip_me_results_download(jmhip_ctx *ctx, jmhip_me_result *results, int n);

/* Single distortion evaluations with JM's signatures' meaning, batched (tests, EPZS/UMHex host loops):
 * computeSAD / computeSATD (src/me_distortion.c:351,657; protos inc/me_distortion.h:43-54) with
 * min_mcost = INT_MAX, UMV access. cand_x/cand_y are quarter-pel INCLUDING the +80 pad offset. */
typedef struct {
  int16_t pic_x, pic_y;     /* block origin in the current picture (pel)                              */
  int16_t bsx, bsy;         /* block size (4, 8 or 16 each)                                           */
  int32_t cand_x, cand_y;
  int16_t ref, use_satd;    /* use_satd: 0 SAD, 1 SATD 4x4, 2 SATD 8x8 (test8x8transform)             */
  int16_t umv, wp;          /* ref_access_method (1 = UMV_ACCESS); wp: computeSADWP / computeSATDWP     */
  int16_t weight, offset;   /* weight_luma, offset_luma (src/me_distortion.c:53)                       */
  int16_t wp_round, wp_denom; /* wp_luma_round, luma_log_weight_denom                                  */
} jmhip_dist_job;
/* out[i] = the FULL distortion (no early exit: what JM returns for min_mcost = INT_MAX). Luma only. */
int jmhip_distortion_batch(jmhip_ctx *ctx, const jmhip_dist_job *jobs, int n, int32_t *out);

/* Distortion SURFACES for host-driven integer-pel walks (EPZSPelBlockMotionSearch src/me_epzs.c:1500,
 * UMHEXIntegerPelBlockMotionSearch src/me_umhex.c:229): every integer displacement (cx-R..cx+R, cy-R..cy+R) of one
 * macroblock against one reference, raster order (dy outer), in the granularity at which JM's kernels can leave early, so
 * that the host reproduces computeSAD / computeSATD return values exactly, partial sums included:
 *   JMHIP_SURFACE_SAD_ROWS     64 uint16 per displacement: [row 0..15][4-sample group 0..3] = the inner-loop terms of
 *                              computeSAD (src/me_distortion.c:364-375, row-wise exit :373)
 *   JMHIP_SURFACE_SATD_BLOCKS  20 uint16: [0..15] HadamardSAD4x4 of the 4x4 blocks (raster), [16..19] HadamardSAD8x8 of
 *                              the 8x8 blocks (computeSATD, :657-731, block-wise exit :690/:720)
 * Samples outside the picture follow UMV access (= per-sample clamp on the integer plane). */
enum { JMHIP_SURFACE_SAD_ROWS = 0, JMHIP_SURFACE_SATD_BLOCKS = 1 };
typedef struct {
  int16_t mb_x, mb_y, ref, R; int16_t cx, cy;      /* centre (cx, cy) in pels                                         */
  int16_t wp, weight, offset, wp_round, wp_denom;   /* wp != 0: computeSADWP / computeSATDWP (:413, :734): every reference */
  int16_t pad;                                      /* sample becomes clip1(((weight*s + wp_round) >> wp_denom) + offset)  */
} jmhip_surface_job;
int jmhip_distortion_surface(jmhip_ctx *ctx, int kind, const jmhip_surface_job *jobs, int n, uint16_t *out);

/* Bi-predictive search of the 16x16 block (JM runs it for block type 1 only, src/mv-search.c:864):
 * stage 0 = FullPelBlockMotionBiPred (src/me_fullsearch.c:164): the block at the FIXED vector s_mv of picture ref1 is
 *   averaged (computeBiPredSAD1, src/me_distortion.c:482) or weight-combined (computeBiPredSAD2 :556) with every candidate
 *   of picture ref2 in the +-search_range window round mv; cost = MV_COST(s_mv, pred1) + MV_COST(cand, pred2) + SAD.
 * stage 1 = SubPelBlockSearchBiPred (:520): half- then quarter-pel refinement of mv (9 + 8 positions) with
 *   computeBiPredSATD1/2 (:824/:907), cost = MV_COST(cand, pred2) + SATD; s_mv and mv in quarter-pel units.
 * JM's naming is kept: "1" = fixed block, read from listX[list][ref]; "2" = swept candidate, read from listX[list^1][0].
 * min_mcost is the carried minimum: a candidate is accepted only below it, otherwise mv and min_mcost come back. */
typedef struct {
  int16_t mb_x, mb_y;
  int16_t ref1, ref2;        /* reference slots of the fixed / swept picture                                   */
  int16_t s_mv[2], mv[2];    /* stage 0: pel units (mv = search centre); stage 1: quarter-pel                   */
  int16_t pred1[2], pred2[2];/* predictors, quarter-pel (pred1 is only read by stage 0)                        */
  int32_t min_mcost;
  int16_t search_range, stage;
} jmhip_bipred_job;
typedef struct { int16_t mv[2]; int32_t cost; } jmhip_bipred_result;
typedef struct {
  int lambda[3];             /* lambda_factor[F_PEL, H_PEL, Q_PEL]                                              */
  int transform8x8_mode;     /* 8x8 Hadamard in stage 1 (test8x8transform for block type 1)                     */
  int apply_weights;         /* active_pps->weighted_bipred_idc > 0                                             */
  int weight1, weight2, offset_bi, wp_luma_round, luma_log_weight_denom;
} jmhip_bipred_params;
int jmhip_bipred_search(jmhip_ctx *ctx, const jmhip_bipred_params *prm, const jmhip_bipred_job *jobs, int n, jmhip_bipred_result *results);

/* SubPelBlockMotionSearch alone (src/me_fullsearch.c:341): results[i].mv_int[p] is the INPUT (integer vector in pel
 * units, as FullPel/FastFull left it), results[i].mv/cost[p] the output. Same params/jobs as jmhip_me_frame. With
 * metric_set and equal metrics at the integer and half-pel level, results[i].cost_int[p] is an input too: the min_mcost the
 * refinement starts from (src/mv-search.c:785-788). */
int jmhip_me_subpel(jmhip_ctx *ctx, const jmhip_me_params *prm, const jmhip_me_mb *mbs, int n, jmhip_me_result *results);


/* ------------------------------------------------------------------ P-slice motion search + low-complexity inter decision, whole slice on the device
 *
 * One call does, for every macroblock of a P slice, what JM does between start_macroblock and the residual coding when
 * RDOptimization = 0 and intra modes are off in inter slices (DisableIntraInInter = 1), with the 4x4 transform:
 *   encode_one_macroblock_low   src/md_low.c:46        (modes 1..3 :112-188, P8x8 :190-330, final parameters :543-636)
 *   PartitionMotionSearch       src/mv-search.c:1378   BlockMotionSearch  src/mv-search.c:560
 *   SetMotionVectorPredictor    src/mv-search.c:87     UMHEXSetMotionVectorPredictor src/me_umhex.c:1298 (dynamic search range)
 *   the integer search of the configured SearchMode:
 *     -1 FullPelBlockMotionSearch src/me_fullsearch.c:47      0 FastFullPelBlockMotionSearch src/me_fullfast.c:833 (+ Setup :491)
 *      1 UMHEXIntegerPelBlockMotionSearch src/me_umhex.c:229  3 EPZSPelBlockMotionSearch src/me_epzs.c:1500
 *      2 smpUMHEXIntegerPelBlockMotionSearch src/me_umhexsmp.c:152 (the simplified UMHexagonS; stateless between calls)
 *   the sub-pel search: SubPelBlockMotionSearch src/me_fullsearch.c:341, UMHEXSubPelBlockMotionSearch src/me_umhex.c:562 (block
 *     types > 3), EPZSSubPelBlockMotionSearch src/me_epzs.c:2390, smpUMHEXSubPelBlockMotionSearch src/me_umhexsmp.c:616 (block types > 1) /
 *     smpUMHEXFullSubPelBlockMotionSearch :422 (the 16x16 block)
 *   the skip shortcut of the 16x16 block (FindSkipModeMotionVector src/mv-search.c:1189, GetSkipCostMB :1136, :829-849)
 *   list_prediction_cost src/mode_decision.c:255, submacroblock_mode_decision :530 (rdopt = 0 path), SetRefAndMotionVectors /
 *     SetModesAndRefframeForBlocks / SetMotionVectorsMB (src/rdopt.c:2777 / :1262 / :1845).
 * Macroblocks run as a 2:1 wavefront (one workgroup per macroblock row; a macroblock starts when its left neighbour and the
 * upper-right one are done), which keeps every neighbour dependency of JM's raster order: predictors, the EPZS row memories
 * (EPZSDistortion, EPZSMotion), the UMHexagonS cost maps. The one dependency that is NOT a neighbour's -- EPZS reads img->all_mv of
 * the PREVIOUS macroblock in coding order (src/me_epzs.c:1433-1471 on vectors the current macroblock has not searched yet), i.e. the
 * last macroblock of the row above for the first of a row -- is speculated per row and verified: the call re-runs the slice until
 * every row started from what the row above really left (at most rows + 1 passes; jmhip_slice_result_info reports the count).
 * EPZS's visited map is JM's: never cleared, a cell counts as visited when it holds the 16-bit EPZSBlkCount of the running search
 * (src/me_epzs.c:49,1550,1598,1757,1840) -- so also when the search 65536 k calls earlier stamped it last, or when the counter passes
 * the zero the map was allocated with. The map and the counter live in the context across slices and pictures (jmhip_slice_state_reset
 * = a freshly started encoder); a call finds the tests an old stamp answers by a scan over its macroblocks' touch records and searches
 * those macroblocks again with the cells pre-marked, until the set is stable (jmhip_epzs_map_info reports how many there were).
 * Frame pictures, luma-only motion estimation (ChromaMEEnable 0), list 0 only, up to JMHIP_SLICE_REFS references. */
#define JMHIP_SLICE_REFS 5                /* every cfg the reference ships has NumberReferenceFrames = 5 (bin/encoder_baseline.cfg:53) */
enum { JMHIP_SEARCH_UMHEX = 1, JMHIP_SEARCH_UMHEX_SIMPLE = 2, JMHIP_SEARCH_EPZS = 3 };
typedef struct jmhip_slice_params {
  int32_t search_mode;                     /* input->SearchMode: -1, 0, 1, 2, 3 */
  int32_t search_range;                    /* input->search_range */
  int32_t full_search;                     /* input->full_search (RestrictSearchRange): range per reference / block type, mv-search.c:1411-1416 */
  int32_t num_refs;                        /* listXsize[LIST_0] */
  int32_t ref_slot[JMHIP_SLICE_REFS];      /* list-0 index -> reference slot of the context */
  int32_t valid[8];                        /* enc_mb.valid[1..7] (input->InterSearch[0][..]); [1] must be set */
  int32_t lambda_mf[3];                    /* enc_mb.lambda_mf[F_PEL, H_PEL, Q_PEL] */
  int32_t ref_cost1;                       /* (int)(2 * enc_mb.lambda_me[Q_PEL]): cost of ref > 0 when rdopt = 0 (mode_decision.c:276) */
  int32_t md_metric;                       /* input->ModeDecisionMetric (skip cost): 0 SAD, 2 SATD */
  int32_t metric[3];                       /* input->MEErrorMetric[F_PEL, H_PEL, Q_PEL]: 0 SAD, 2 SATD */
  int32_t level_mv_min, level_mv_max;      /* LEVELMVLIMIT[img->LevelIndex][0..1] */
  int32_t wp_me, wp_pred;                  /* explicit weights in the searches (UseWeightedReferenceME) / in LumaPrediction (skip cost) */
  int32_t wp_round, wp_denom;              /* wp_luma_round, luma_log_weight_denom */
  int16_t wp_weight[JMHIP_SLICE_REFS], wp_offset[JMHIP_SLICE_REFS];     /* wp_weight[0][ref][0], wp_offset[0][ref][0] */
  int32_t mb_first, mb_count;              /* the slice: macroblock addresses [mb_first, mb_first + mb_count) */
  /* EPZS (SearchMode 3): input->EPZS*; thresholds and window predictors as EPZSInit builds them (jmhip_epzs_setup); the POC-distance
   * scales mv_scale[LIST_0][i][k] of EPZSSliceInit (src/me_epzs.c:516-547; jmhip_epzs_scales) */
  int32_t epzs_pattern, epzs_dual, epzs_fixed, epzs_temporal, epzs_spatial_mem, epzs_subpel_me;
  int32_t epzs_thres[4][8];                /* minthres, medthres, maxthres, subthres */
  int32_t epzs_nwin, epzs_nwin_ext;
  int16_t epzs_win[40][2], epzs_win_ext[100][2];
  int32_t epzs_mv_scale[JMHIP_SLICE_REFS][JMHIP_SLICE_REFS];
  /* UMHexagonS (SearchMode 1): input->UMHexDSR and the thresholds of UMHEX_DefineThreshold(MB) (src/me_umhex.c:78-146; jmhip_umhex_setup) */
  int32_t umhex_dsr;
  int32_t umhex_thres[4][8];               /* Median_Pred_Thd_MB, Big_Hexagon_Thd_MB, Multi_Ref_Thd_MB, Threshold_DSR_MB */
  float   umhex_bsize[8], umhex_alpha1[8], umhex_alpha2[8];
  /* input->Transform8x8Mode (round 2): 0 = 4x4 transform only. 1 = both sizes compete: block types 1..4 are searched with the 8x8 Hadamard
   * (src/mv-search.c:640), TransformDecision (src/macroblock.c:1458) follows each of modes 1..3 (src/md_low.c:183-188, incl. the references
   * SetModesAndRefframeForBlocks writes into the picture array, src/rdopt.c:1470-1500), P8x8 runs an 8x8-transform pass (sub-mode 4 only)
   * before the 4x4-transform pass (:203-262), the cheaper pass wins (:281-326, ties: GetBestTransformP8x8 src/rdopt.c:3262) -- and an
   * 8x8-transform winner whose four blocks quantise to nothing falls back to the 4x4 pass's partitioning (:547-548): that needs the inter 8x8
   * luma quantiser of the slice below (LumaResidualCoding8x8 src/macroblock.c:1009, dct_8x8 src/transform8x8.c:1452, flat or scaling-matrix
   * tables as JM built them, AdaptiveRounding off, frame scan). 2 = 8x8 only (no 4x4 pass, no fallback, tables unused).
   * valid[4] must be set with 1 or 2. The skip cost takes distortion8x8 (src/mv-search.c:1171). */
  int32_t transform8x8_mode;
  int32_t t8_qp, t8_cavlc, t8_disthres;                /* currMB->qp_scaled[0], input->symbol_mode == CAVLC, input->disthres */
  int32_t t8_levelscale[64], t8_leveloffset[64];       /* LevelScale8x8Luma_Inter[qp % 6], LevelOffset8x8Luma_Inter[qp / 6], row-major [j][i] */
  /* Several slices in ONE call (input->slice_mode 1, fixed macroblock count: BASELINE config 4's eight slices): slice_mbs > 0 cuts
   * [mb_first, mb_first + mb_count) into slices of slice_mbs macroblocks (the last one shorter). Neighbours across a slice boundary are
   * unavailable (src/mb_access.c:30-36); img->all_mv runs on from one slice's last macroblock to the next slice's first as in JM. The slices
   * relax together: the sweeps a picture needs are those of its slowest slice, not their sum. All search modes: the memories of the EPZS and
   * UMHexagonS walkers are picture-level arrays that JM carries on from slice to slice in coding order, which is what the stored rows of the
   * relaxation hold. 0 = one slice. */
  int32_t slice_mbs;
  /* input->rdopt as BlockMotionSearch sees it: 0 (the configuration whose decision this call reproduces); != 0 drops what JM does only with
   * RDOptimization off -- the clamp of the search centre to the range (src/mv-search.c:755), check_for_00 / check_position0
   * (src/me_fullsearch.c:75, :361), the pos_00 pre-check of FastFullSearch (src/me_fullfast.c:867), the skip shortcut (src/mv-search.c:826-849) --
   * so that every BlockMotionSearch record is what JM's call WITH THAT PREDICTOR returns in the high-complexity modes too. The decision between
   * the calls stays the low-complexity one: with rdopt != 0 it is a GUESS of JM's rate-distortion decision, and a caller answers a call from a
   * record only when its own predictor equals the recorded one (the JM binding's speculative mode). Search modes -1, 0, 2 (pure functions of
   * the predictor). */
  int32_t rdopt;
} jmhip_slice_params;

/* what JM knows of one macroblock after the decision + the outcome of each of its BlockMotionSearch calls */
typedef struct jmhip_mb_inter {
  int32_t best_mode;                       /* 1, 2, 3 or 8 (P8x8) */
  int32_t min_cost;
  int32_t b8mode[4], b8ref[4];
  int16_t final_mv[16][2];                 /* enc_picture->mv[LIST_0] of the macroblock, 4x4 blocks in raster order */
  int16_t skip_mv[2];                      /* all_mv[..][LIST_0][0][0] (FindSkipModeMotionVector) */
  int16_t pred[JMHIP_SLICE_REFS][JMHIP_NPART][2], mv_int[JMHIP_SLICE_REFS][JMHIP_NPART][2], mv[JMHIP_SLICE_REFS][JMHIP_NPART][2];
  int32_t cost_int[JMHIP_SLICE_REFS][JMHIP_NPART], cost[JMHIP_SLICE_REFS][JMHIP_NPART];
  /* Transform8x8Mode: the BlockMotionSearch calls of the 8x8-transform P8x8 pass (block type 4, per 8x8 block; with mode 1 JM searches the
   * same blocks again in the 4x4-transform pass: those calls are in the arrays above), the transform size of the chosen mode as the decision
   * leaves it (before the residual coder's cbp == 0 reset), and the coded-block pattern of the 8x8-transform pass where the decision needed it
   * (else -1) */
  int16_t pred8ts[JMHIP_SLICE_REFS][4][2], mv_int8ts[JMHIP_SLICE_REFS][4][2], mv8ts[JMHIP_SLICE_REFS][4][2];
  int32_t cost_int8ts[JMHIP_SLICE_REFS][4], cost8ts[JMHIP_SLICE_REFS][4];
  int32_t transform8x8_flag, cbp8ts;
  /* the P8x8 CANDIDATE as submacroblock_mode_decision (src/mode_decision.c:531) leaves it, whatever mode wins: sub-mode (4..7; 0 when the 8x8 modes
   * are disabled) and list-0 reference of each 8x8 block. JM predicts and transforms it per 8x8 block before the decision (:874) -- a slice-level
   * binding that also answers those calls codes it with jmhip_slice_to_frame_candidates. */
  int32_t p8mode[4], p8ref[4];
} jmhip_mb_inter;

/* Host helpers that fill the search-mode blocks of jmhip_slice_params exactly as JM's initialisation does. */
void jmhip_epzs_setup(jmhip_slice_params *p, int search_range, int pattern, int dual, int fixed, int temporal, int spatial_mem, int subpel_me,
                      int min_scale, int med_scale, int max_scale, int subpel_scale);          /* EPZSInit, src/me_epzs.c:333 (8-bit, no chroma ME) */
void jmhip_epzs_scales(jmhip_slice_params *p, int poc, const int *list0_poc, int num_refs);   /* EPZSSliceInit, src/me_epzs.c:516-547 */
void jmhip_umhex_setup(jmhip_slice_params *p, int dsr, int scale, int qp_n, int width);       /* UMHEX_DefineThreshold, src/me_umhex.c:78 */

/* State that outlives a slice (JM's file-static / global arrays): EPZSDistortion, EPZSMotion, img->all_mv of the last macroblock coded,
 * the UMHexagonS cost maps. jmhip_slice_state_reset zeroes it (start of a sequence). The picture-level LIST_0 arrays
 * (enc_picture->ref_idx / mv) are NOT reset between pictures: a macroblock only ever reads entries of macroblocks coded before it in the
 * same slice, and the previous picture's field is the first guess of the relaxation schedule below. */
int jmhip_slice_state_reset(jmhip_ctx *ctx);
/* EPZS temporal predictors: the scaled co-located field EPZSSliceInit builds (EPZSCo_located->mv[LIST_0], src/me_epzs.c:986-1030),
 * [H/4][W/4][2] int16 quarter-pel, host array. Needed before jmhip_p_slice_search when epzs_temporal is set. */
int jmhip_epzs_colocated_upload(jmhip_ctx *ctx, const int16_t *col_mv);
/* Run one P slice on the current picture. results: host array of mb_count records (may be NULL: results stay on the device for
 * jmhip_slice_results_download / jmhip_slice_field_download).
 * Schedule: JM codes the macroblocks in raster order and each reads what its predecessors left (predictors, row memories). The device
 * evaluates every macroblock of the slice at once and repeats until a sweep changes nothing; the dependencies are acyclic, so that fixpoint
 * is unique and is JM's result. If the sweeps do not settle within a cap the coding-order wavefront finishes; environment
 * JMHIP_SLICE_SCHED=wave uses the wavefront only, JMHIP_SLICE_SWEEPS=n sets the cap, JMHIP_SLICE_TRACE=1 prints the sweeps.
 * One slice search at a time per device and process (the parameter block lives in __constant__ memory); the call synchronises. */
int jmhip_p_slice_search(jmhip_ctx *ctx, const jmhip_slice_params *prm, jmhip_mb_inter *results);
int jmhip_slice_results_download(jmhip_ctx *ctx, jmhip_mb_inter *results, int mb_first, int mb_count);
/* The picture-level LIST_0 arrays: ref_idx [H/4][W/4] int8, mv [H/4][W/4][2] int16 (either may be NULL). */
int jmhip_slice_field_download(jmhip_ctx *ctx, int8_t *ref_idx, int16_t *mv);
/* sweeps (relaxation) plus passes (wavefront) the last jmhip_p_slice_search needed */
int jmhip_slice_result_info(jmhip_ctx *ctx, int *passes);
/* EPZS: map tests of the last jmhip_p_slice_search that JM answers "visited" from a stamp this search did not write (see above), and the
 * integer searches since jmhip_slice_state_reset (its low 16 bits are EPZSBlkCount, src/me_epzs.c:49). Either pointer may be NULL. */
int jmhip_epzs_map_info(jmhip_ctx *ctx, int *aliased_tests, uint32_t *searches);
/* EPZSMap ([2 search_range + 1]^2 stamps, row-major as JM indexes it; NULL = all zero) and EPZSBlkCount of an encoder that is already
 * running (src/me_epzs.c:49,92): what a binding hands over when it attaches mid-stream, or after JM ran EPZS searches of its own. */
int jmhip_epzs_map_upload(jmhip_ctx *ctx, const int16_t *map, int search_range, int blk_count);
/* Hand the searched picture (every macroblock: all its slices searched) to the frame stage: the decided modes, the vector and the
 * reference slot of every 8x8 block become the inputs of jmhip_residual_frame(modes = NULL), which then runs LumaResidualCoding /
 * ChromaResidualCoding on them (per-8x8 reference pictures: macroblock.c:1009-1110 with SetModesAndRefframe), and of jmhip_deblock_recon.
 * ref_slot: list-0 index -> reference slot, as in the slice calls. */
int jmhip_slice_to_frame(jmhip_ctx *ctx, const int32_t *ref_slot, int num_refs);     /* reference slots 0..7 */
/* The same for the macroblocks [mb_first, mb_first + mb_count) alone -- a rank of the slice-parallel layout (one slice per GPU,
 * src/slice.c:214 with SliceMode 1) hands ITS slice to the frame stage: job i of jmhip_residual_frame is macroblock mb_first + i, the
 * reconstruction covers those macroblocks' rows (jmhip_recon_pack_band). The range must have been searched in the current picture. */
int jmhip_slice_to_frame_band(jmhip_ctx *ctx, const int32_t *ref_slot, int num_refs, int mb_first, int mb_count);
/* The same hand-over with every macroblock in its P8x8 CANDIDATE form (jmhip_mb_inter.p8mode / p8ref, the vectors of those sub-modes) instead of the
 * decided mode: jmhip_residual_frame then yields what LumaResidualCoding8x8 computes for the candidate inside submacroblock_mode_decision
 * (src/mode_decision.c:874). The reconstruction of that pass is not a picture: call it BEFORE the hand-over of the decision. */
int jmhip_slice_to_frame_candidates(jmhip_ctx *ctx, const int32_t *ref_slot, int num_refs, int mb_first, int mb_count);

/* ------------------------------------------------------------------ low-complexity (rdopt off) mode-decision costs */

/* The residual distortion of an inter-predicted macroblock, the quantity JM's RD-off decision compares:
 *   TransformDecision (src/macroblock.c:1458): cost4x4 = sum of distortion4x4 over the sixteen 4x4 blocks, cost8x8 = sum of
 *     distortion8x8 over the four 8x8 blocks -- with JM's diff64 layout (the 8x8 block's four 4x4 residual blocks stored one after
 *     the other and read back as 8 rows of 8, :1496-1505): layout = JMHIP_DIFF64_SEQUENTIAL;
 *   GetSkipCostMB (src/mv-search.c:1136): the same sums on the true 8x8 raster: layout = JMHIP_DIFF64_RASTER.
 * Prediction per 4x4 block as LumaPrediction(.., 4, 4, ..): its own vector(s) and reference slot(s), UMV origin clamp per 4x4 block. metric: input->ModeDecisionMetric, 0 = SAD, 2 = SATD (src/me_distortion.c:76, :110). out[i] = {cost4x4, cost8x8}.
 * (The two layouts yield equal sums: the sequential one permutes the index bits of the 8x8 block and the Hadamard magnitudes' sum is
 * invariant under that; both are implemented literally and tested.) */
enum { JMHIP_DIFF64_SEQUENTIAL = 0, JMHIP_DIFF64_RASTER = 1 };
/* One macroblock's prediction as LumaPrediction forms it (src/macroblock.c:836-945), 4x4 blocks in raster order y*4+x:
 *   bi[b] == 0: list-X prediction from (mv[b], ref[b]);  bi[b] != 0: p_dir 2, the second operand is (mv1[b], ref1[b]).
 *   weighted == 0: the sample or (a + b + 1) >> 1; weighted != 0: clip1(((w0*a + wp_round) >> denom) + off) (uni) or
 *   clip1(((w0*a + w1*b + 2*wp_round) >> (denom + 1)) + off) (bi), per block (the weights depend on the references).
 * blocks: bit b set = 4x4 block b counts in cost4x4; an 8x8 block counts in cost8x8 when all four of its 4x4 blocks are set
 * (BIDPartitionCost src/mv-search.c:1050 evaluates one partition; TransformDecision / GetSkipCostMB the whole macroblock: 0xffff). */
typedef struct {
  int16_t mb_x, mb_y;
  uint16_t blocks;
  int16_t weighted, wp_round, wp_denom;
  int16_t mv[16][2], mv1[16][2];
  int8_t  ref[16], ref1[16], bi[16];
  int16_t w0[16], w1[16], off[16];
} jmhip_predcost_job;
int jmhip_pred_cost_batch(jmhip_ctx *ctx, const jmhip_predcost_job *jobs, int n, int metric, int layout, int32_t (*out)[2]);

/* ------------------------------------------------------------------ transform / quant / recon */

/* Quantiser of one block (JM: levelscale/invlevelscale/leveloffset selected at src/block.c:874-877,
 * src/transform8x8.c:1487-1489). Tables row-major [j][i]. */
typedef struct {
  int32_t qp;                    /* currMB->qp_scaled[pl] or chroma qp                               */
  int32_t adaptive_rounding, adapt_rnd_weight;
  int32_t field_scan, disthres, max_val, cavlc, img_qp, transform8x8_flag;
  int32_t levelscale[64], invlevelscale[64], leveloffset[64];     /* 16 used for 4x4 kinds          */
} jmhip_quant;

enum { JMHIP_TQ_LUMA4x4 = 0,     /* dct_4x4   src/block.c:843, pointer pDCT_4x4 inc/global.h:1376      */
       JMHIP_TQ_LUMA8x8 = 1,     /* dct_8x8   src/transform8x8.c:1452, proto inc/transform8x8.h:23     */
       JMHIP_TQ_LUMA16x16 = 2,   /* dct_16x16 src/block.c:564, proto inc/global.h:1378                 */
       JMHIP_TQ_CHROMA = 3 };    /* dct_chroma src/block.c:1051, proto inc/global.h:1379               */

/* One macroblock-component job. For LUMA4x4 one job covers the 16 4x4 blocks of the MB (each an
 * independent dct_4x4 call), LUMA8x8 the four 8x8 blocks. src = original samples, pred = prediction
 * (img->mpr); the residual img->m7 = src - pred is formed on the device. */
typedef struct {
  uint8_t  src[16][16];
  uint8_t  pred[16][16];
  int32_t  quant;                /* index into the quant table array passed to the batch call         */
  int32_t  quant_dc;             /* 4:2:2 chroma DC quantiser (qp+3), else ignored                     */
  int32_t  uv, cr_cbp_in;        /* CHROMA: component and incoming cr_cbp                              */
  int32_t  intra16_unused;
} jmhip_tq_job;

typedef struct {
  int32_t  levels[16][17];       /* (level) lists per 4x4 block in JM block order b8*4+b4, 0-terminated; entries after the
                                    terminator are not written (JM never clears img->cofAC either);
                                    LUMA8x8: rows 4*b8 .. 4*b8+3 hold cofAC[b8][0..3] (17 entries each
                                    for CAVLC interleave) -- see levels8 for the 64-entry CABAC list  */
  int32_t  runs[16][17];
  int32_t  levels8[4][65], runs8[4][65];  /* LUMA8x8 non-interleaved (CABAC / flag off) lists        */
  int32_t  dc_levels[17], dc_runs[17];    /* LUMA16x16 / CHROMA DC lists                               */
  uint8_t  recon[16][16];
  int32_t  fadjust[16][16];
  int32_t  coeff_cost[16];       /* per 4x4 (LUMA4x4) / per 8x8 in [0..3] (LUMA8x8)                    */
  int32_t  nonzero[16];          /* return value of dct_4x4 / dct_8x8 per block                        */
  int32_t  ret;                  /* dct_16x16: ac_coef; dct_chroma: cr_cbp                             */
  int64_t  cbp_blk;              /* dct_chroma: bits to OR into currMB->cbp_blk ...                     */
  int64_t  cbp_clear;            /* ... after clearing these (block.c:1399): cbp = (cbp & ~clear) | set  */
} jmhip_tq_result;

int jmhip_tq_batch(jmhip_ctx *ctx, int kind, int yuv_format, const jmhip_quant *quants, int nquants,
                   const jmhip_tq_job *jobs, int n, jmhip_tq_result *results);

/* ------------------------------------------------------------------ frame stage: MC prediction -> residual -> TQ -> recon */

/* Explicit weighted prediction in the frame stage's motion compensation (P slices with active_pps->weighted_pred_flag): every predicted sample
 * becomes iClip1(255, ((weight * p + round) >> denom) + offset) -- LumaPrediction src/macroblock.c:891-900 with wp_weight[0][ref][0] / wp_offset[0][ref][0],
 * wp_luma_round, luma_log_weight_denom; ChromaPrediction4x4 :1895-1903 with components 1, 2, wp_chroma_round, chroma_log_weight_denom.
 * Indexed by reference SLOT; [k][0] luma, [k][1] Cb, [k][2] Cr. wp == NULL or enable == 0: plain prediction. Stays set until changed. */
typedef struct jmhip_frame_wp {
  int32_t enable, luma_round, luma_denom, chroma_round, chroma_denom;
  int16_t weight[16][3], offset[16][3];
} jmhip_frame_wp;
int jmhip_frame_wp_set(jmhip_ctx *ctx, const jmhip_frame_wp *wp);

/* B macroblocks in the frame stage: the second list of LumaPrediction (src/macroblock.c:836: p_dir 0 / 1 / 2 at :862-876, the mixes
 * :880-940) and ChromaPrediction4x4 (:1768-1830). Per macroblock of the job list: per 8x8 block the prediction direction (0 list 0 only --
 * the P case --, 1 list 1 only, 2 both), the list-1 reference SLOT, and per 4x4 block (raster) the list-1 vector in quarter-pel units. List 0
 * stays what the search stage / jmhip_slice_to_frame left. bi == NULL switches back to P macroblocks. Host array, copied. */
typedef struct jmhip_mb_bipred { int8_t pdir[4]; int8_t ref1[4]; int16_t mv1[16][2]; } jmhip_mb_bipred;
/* Weights of the second list, used when jmhip_frame_wp_set has weighting on (its rounding / denominators apply; JM's weighted_bipred_idc):
 * by reference SLOT (0..3) and component (Y, Cb, Cr): bi-predictive pair weights wbp_weight[0 / 1][ref0][ref1][c], list-1 uni-directional
 * weight wp_weight[1][ref1][c] and offset wp_offset[1][ref1][c]. JM quirk mirrored: the bi-predictive chroma mix shifts by the LUMA
 * denominator + 1 (src/macroblock.c:1781). */
typedef struct jmhip_frame_bw { int16_t w0[4][4][3], w1[4][4][3], weight1[4][3], offset1[4][3]; } jmhip_frame_bw;
int jmhip_frame_bipred_set(jmhip_ctx *ctx, const jmhip_mb_bipred *bi, int n, const jmhip_frame_bw *bw);

/* Inter mode of one macroblock as JM's mode decision fixed it (the decision itself stays on the host):
 * mode 1 = 16x16, 2 = 16x8, 3 = 8x16, 8 = P8x8 with b8mode[b] in {4,5,6,7} (8x8, 8x4, 4x8, 4x4). */
typedef struct { int8_t mode; int8_t b8mode[4]; int8_t pad[3]; } jmhip_mb_mode;   /* pad[0] = luma_transform_size_8x8_flag (0/1) */

/* For the n macroblocks of the last jmhip_me_frame(_async) call, in the same order:
 *   LumaPrediction / OneComponentLumaPrediction (src/macroblock.c:836, :807) per 4x4 block with the motion vectors
 *   the search found for the macroblock's mode, chroma prediction from the eighth-pel planes
 *   (OneComponentChromaPrediction4x4_retrieve, src/macroblock.c:1593), residual m7 = org - mpr
 *   (src/macroblock.c:1059-1068), dct_4x4 x16 and dct_chroma x2 (inter tables), reconstruction into the context's
 *   recon picture. modes == NULL: the device takes, per macroblock, the partitioning with the smallest summed
 *   motion cost (ties to the lower mode number) -- a stand-in for the host's mode decision used by bench.py.
 *   quants[0] = luma inter quantiser, quants[1] = chroma quantiser, quants[2] = 4:2:2 chroma DC (qp+3) quantiser.
 * Results stay on the device until downloaded: luma[n], chroma[2n] (index 2*i+uv), modes_out[n], cbp[n], cbp_blk[n]. */
/* Chroma prediction (src/macroblock.c:1593 ..._retrieve) reads the eighth-pel planes of jmhip_interp_chroma when every used
 * reference has them; when they were not built (reference slots 0..3) the kernel computes the very samples those planes would
 * hold from the integer chroma picture -- identical results, no 64x chroma planes in HBM. */
int jmhip_residual_frame(jmhip_ctx *ctx, const jmhip_mb_mode *modes, const jmhip_quant quants[3]);
/* Same with nquants = 4: quants[3] is the 8x8 luma quantiser (transform8x8_flag = 1, 64-entry tables). Macroblocks whose mode has
 * pad[0] = 1 (JM's TransformDecision outcome, src/macroblock.c:1458; only legal without partitions below 8x8) take the dct_8x8
 * branch of LumaResidualCoding8x8 (src/macroblock.c:1131-1188): prediction and UMV clamp per 8x8 block, dct_8x8, cbp_blk bits
 * 51 << (4*b8 - 2*(b8&1)), the same coefficient-cost thresholds. */
int jmhip_residual_frame_q(jmhip_ctx *ctx, const jmhip_mb_mode *modes, const jmhip_quant *quants, int nquants);
/* cbp / cbp_blk: currMB->cbp and currMB->cbp_blk after the _LUMA_COEFF_COST_ (8x8) and _LUMA_MB_COEFF_COST_ (MB)
 * thresholding of src/macroblock.c:1236-1258, :1386-1392 and the chroma cr_cbp. Any output pointer may be NULL. */
int jmhip_residual_download(jmhip_ctx *ctx, jmhip_tq_result *luma, jmhip_tq_result *chroma, jmhip_mb_mode *modes_out,
                            int32_t *cbp, int64_t *cbp_blk, int n);
/* The dense per-macroblock record of the fused 4:2:0 frame stage (4x4 transform): what JM's dct_4x4 x16 (src/block.c:843) and dct_chroma x2
 * (:1051) leave behind for one macroblock. jmhip_residual_records_download copies the records of the last jmhip_residual_frame as they are
 * (2.4 KB each; jmhip_residual_download expands them into three 5.8 KB jmhip_tq_result) -- the form a slice-level binding answers JM's
 * dct_4x4 / dct_chroma calls from. JMHIP_ERR_UNSUPPORTED when the last frame stage took the separate kernels (4:2:2, 4:0:0, 8x8 transform). */
typedef struct jmhip_mb_residual {
  int16_t lev[24][16];           /* (level) lists in scan order: luma blocks 0..15 (JM order b8*4+b4), Cb 16..19, Cr 20..23 (AC) */
  uint8_t run[24][16];
  uint8_t cnt[24];               /* entries of each list; JM's 0 terminator follows them */
  int16_t dc_lev[2][4];          /* chroma DC lists */
  uint8_t dc_run[2][4];
  uint8_t dc_cnt[2];
  uint8_t ac_zeroed[2];          /* _CHROMA_COEFF_COST_ thresholding hit: the AC levels of the component read 0, the runs stay (block.c:1384-1410) */
  uint8_t pad0[4];
  int32_t coeff_cost[16];        /* luma, per 4x4 block: what dct_4x4 adds to *coeff_cost */
  int32_t ret[2];                /* dct_chroma's return value (cr_cbp) per component */
  uint16_t nonzero;              /* luma: bit blk = dct_4x4's return value */
  uint16_t pad1[3];
  int64_t cbp_blk[2], cbp_clear[2];   /* dct_chroma: currMB->cbp_blk = (cbp_blk & ~cbp_clear) | cbp_blk */
  int16_t fadj_y[16][16];        /* adaptive rounding only: img->fadjust4x4 / fadjust4x4Cr */
  int16_t fadj_c[2][8][8];
  uint8_t recon_y[16][16];       /* the transform path's reconstruction, as dct_4x4 writes it (before the caller's coefficient-cost decision) */
  uint8_t recon_c[2][8][8];
  uint8_t pad2[8];
} jmhip_mb_residual;
int jmhip_residual_records_download(jmhip_ctx *ctx, jmhip_mb_residual *records, int n);
/* Keep the prediction picture -- img->mpr of every macroblock of jmhip_residual_frame, luma and chroma (src/macroblock.c:836, :1593) -- beside
 * the recon picture (fused 4:2:0 stage only), and copy it to the host (pel_bytes 1 or 2): a binding that answers JM's LumaPrediction /
 * ChromaPrediction4x4 from the device reads it, and checks its dct inputs against it. */
int jmhip_frame_keep_prediction(jmhip_ctx *ctx, int on);
int jmhip_pred_download(jmhip_ctx *ctx, void *Y, void *U, void *V, int pel_bytes);
/* Make the recon picture the integer-pel picture of reference slot `ref`, e.g. for the next frame. Nothing is copied: the
 * slot's planes and the recon planes TRADE PLACES. Consequences: (1) device pointers handed out earlier by
 * jmhip_ref_device_planes / jmhip_ref_planes_peek for this slot are stale -- ask again; (2) the recon picture is INVALID
 * afterwards (it holds the slot's old picture): jmhip_recon_download / _copy_band / _pack_band / jmhip_deblock_* and a second
 * jmhip_recon_to_ref return JMHIP_ERR_ARG until jmhip_residual_frame or jmhip_recon_upload has produced a new one. */
int jmhip_recon_to_ref(jmhip_ctx *ctx, int ref);
/* Copy the band of macroblock rows [mb_row0, mb_row0+mb_rows) of the recon picture into caller-provided DEVICE buffers
 * (tightly packed rows): the send buffer of the per-frame all-gather of reconstructed slice bands. */
int jmhip_recon_copy_band(jmhip_ctx *ctx, void *Y, void *U, void *V, int mb_row0, int mb_rows);
/* Copy the recon picture to the host (pel_bytes 1, or 2 for JM's `imgpel` rows). */
int jmhip_recon_download(jmhip_ctx *ctx, void *Y, void *U, void *V, int pel_bytes);
/* Load the recon picture from the host (tightly packed rows; pel_bytes 1 or 2): for callers that reconstruct elsewhere
 * (the JM binding) and only want jmhip_deblock_frame. */
int jmhip_recon_upload(jmhip_ctx *ctx, const void *Y, const void *U, const void *V, int pel_bytes);

/* ---- in-loop deblocking filter: DeblockFrame, lencod/src/loopFilter.c:87 (DeblockMb :128, GetStrengthNormal :263,
 * EdgeLoopLumaNormal :529, EdgeLoopChromaNormal :815). Frame pictures without MBAFF; not for SP/SI slices. */
typedef struct jmhip_deblock_mb {      /* what the filter reads of img->mb_data[i], raster order */
  uint8_t intra;                       /* mb_type is I4MB, I8MB, I16MB or IPCM (ANY_INTRA, loopFilter.c:261) */
  uint8_t qp, qpc[2];                  /* IPCM: 0 (DeblockFrame :105-113) */
  uint8_t disable_idc;                 /* LFDisableIdc: 0 filter, 1 off, 2 not across slice borders */
  int8_t alpha_c0_offset, beta_offset; /* LFAlphaC0Offset, LFBetaOffset */
  uint8_t transform_8x8;               /* luma_transform_size_8x8_flag: luma edges 1 and 3 are skipped (:153) */
  uint8_t avail_a, avail_b;            /* mbAvailA / mbAvailB as the encoder left them (slice-aware); read when disable_idc == 2 (:163-169) */
  uint16_t cbp_blk;                    /* the 16 luma bits of cbp_blk */
} jmhip_deblock_mb;
typedef struct jmhip_deblock_blk {     /* one 4x4 block, raster order over the picture (width/4 per row) */
  int16_t mv[2][2];                    /* enc_picture->mv[list][by][bx][0..1] */
  int64_t ref_id[2];                   /* enc_picture->ref_pic_id[list][by][bx], INT64_MIN where ref_idx[list] < 0 (:334-337) */
} jmhip_deblock_blk;
/* Filters the recon picture in place, macroblock rows [mb_row0, mb_row0 + mb_rows) (mb_rows <= 0: the whole picture; a band is only
 * self-contained when its first row's top edge is not filtered, i.e. slices with disable_idc 2). mbs: mbw*mbh entries, blks:
 * 16*mbw*mbh entries, HOST arrays borrowed for the call. mvlimit: 4 (frame pictures). Results are identical to JM's macroblock-
 * order filter: the kernel keeps that order through a 2:1 wavefront (deblock.hip). */
int jmhip_deblock_frame(jmhip_ctx *ctx, const jmhip_deblock_mb *mbs, const jmhip_deblock_blk *blks, int mvlimit, int mb_row0, int mb_rows);

/* The same filter fed from what the frame stage left on the device: the vectors of the last jmhip_me_frame, the modes and coded-block
 * bits of the last jmhip_residual_frame (same macroblock list; P macroblocks, one reference slot per macroblock) -- the state JM
 * would have stored in img->mb_data[] / enc_picture before image.c:331 calls DeblockFrame. Nothing crosses PCIe. The picture-level
 * values come from the caller: the (uniform) quantisers as MbQ->qp / qpc[0..1], the slice header's filter fields, and the slice
 * height in macroblock rows (0 = one slice) from which mbAvailB follows for disable_idc 2. Rows outside the list (other ranks' bands)
 * are not known: restrict [mb_row0, mb_row0 + mb_rows) to listed rows whose top edge is off. */
typedef struct jmhip_deblock_params {
  int32_t qp, qpc[2];
  int32_t disable_idc, alpha_c0_offset, beta_offset;
  int32_t slice_rows;
  int32_t mvlimit;               /* 0 or 4: frame pictures */
  int32_t mb_row0, mb_rows;      /* mb_rows <= 0: the whole picture */
} jmhip_deblock_params;
int jmhip_deblock_recon(jmhip_ctx *ctx, const jmhip_deblock_params *prm);
/* One-buffer band exchange (slice-parallel ranks): rank `rank`'s reconstructed band of `band_rows` macroblock rows
 * as ONE device chunk [Y rows | U rows | V rows] (jmhip_band_chunk_bytes), so that one all-gather moves the frame;
 * jmhip_ref_unpack_bands scatters the `world` gathered chunks into reference slot `ref` (rows below the picture are padding). */
size_t jmhip_band_chunk_bytes(jmhip_ctx *ctx, int band_rows);
int jmhip_recon_pack_band(jmhip_ctx *ctx, void *chunk_device, int rank, int band_rows);
int jmhip_ref_unpack_bands(jmhip_ctx *ctx, int ref, const void *chunks_device, int world, int band_rows);

/* sizeof() of the ABI structs, for language bindings to verify their layout: 0 jmhip_me_mb, 1 jmhip_me_result,
 * 2 jmhip_quant, 3 jmhip_tq_job, 4 jmhip_tq_result, 5 jmhip_dist_job, 6 jmhip_me_params, 7 jmhip_config,
 * 8 jmhip_mb_mode, 9 jmhip_surface_job, 10 jmhip_bipred_job, 11 jmhip_bipred_result, 12 jmhip_bipred_params, 13 jmhip_predcost_job,
 * 14 jmhip_deblock_mb, 15 jmhip_deblock_blk, 16 jmhip_deblock_params, 17 jmhip_slice_params, 18 jmhip_mb_inter, 19 jmhip_frame_wp,
 * 20 jmhip_mb_bipred, 21 jmhip_frame_bw, 22 jmhip_mb_residual. */
int jmhip_sizeof(int which);

/* Flat (no scaling matrix) tables: CalculateQuantParam / CalculateQuant8Param (src/q_matrix.c:451,590) and
 * default CalculateOffsetParam / CalculateOffset8Param (src/q_offsets.c:491,629); offset11 = 682 or 342. */
void jmhip_flat_quant(jmhip_quant *q, int qp, int offset11, int is8x8);

#ifdef __cplusplus
}
#endif
#endif
