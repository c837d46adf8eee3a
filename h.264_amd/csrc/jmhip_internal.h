// jmhip_internal.h -- shared declarations of libjmhip.so (C++/HIP side; the public ABI is include/jmhip.h)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "jmhip.h"

struct ChromaGeom {           // chroma_mc_setup, lencod/src/lencod.c:2851-2884
  int sub_x, sub_y, pad_x, pad_y, shift_x, shift_y, mask_x, mask_y, mul_x, mul_y, mb_w, mb_h;
};

// what the frame stage keeps per macroblock after thresholding (frame.hip finalize_kernel); read by deblock.hip
struct JmMbCoded { int32_t cbp; int32_t pad; int64_t cbp_blk; };

struct RefSlot {
  uint8_t *y = nullptr, *u = nullptr, *v = nullptr;   // integer-pel recon, pitch = W / Wc
  uint8_t *luma_sub = nullptr;                        // [16][Hp][Wp]
  uint8_t *cr_sub[2] = {nullptr, nullptr};            // [sub_y*sub_x][Hcp][Wcp]
  bool has_pic = false, has_luma_sub = false, has_cr_sub = false;
};

struct jmhip_ctx {
  jmhip_config cfg{};
  hipStream_t stream = nullptr;
  int W = 0, H = 0, Wp = 0, Hp = 0;                   // luma and padded luma size
  int Wc = 0, Hc = 0, Wcp = 0, Hcp = 0;               // chroma and padded chroma size
  int mbw = 0, mbh = 0;
  ChromaGeom cg{};
  std::vector<RefSlot> refs;
  uint8_t *cur_y = nullptr, *cur_u = nullptr, *cur_v = nullptr;            // the active current picture (owned or bound)
  uint8_t *cur_own[3] = {nullptr, nullptr, nullptr};                          // the context's own planes (jmhip_cur_upload target)
  bool has_cur = false;
  // staging for 16-bit sample conversion
  void *stage_dev = nullptr; size_t stage_bytes = 0;
  // page-locked host staging of the row-pointer downloads (jmhip_ref_download_*_rows) and one event per plane in flight
  uint8_t *pin_host = nullptr; size_t pin_bytes = 0;
  std::vector<hipEvent_t> pin_evt;
  // ME job/result arrays
  void *me_jobs_dev = nullptr; void *me_res_dev = nullptr; int me_capacity = 0; int me_n = 0;
  int me_max_uw = 0, me_max_uh = 0, me_last_mode = 0, me_last_R = 0, me_last_rdopt = 0, me_last_lvl[2] = {0, 0};
  unsigned long long me_last_mask = 0;
  bool me_last_metric_path = false;
  unsigned me_ref_mask = 0;
  void *me_idx_dev = nullptr;                         // macroblock indices: fast-path list then generic list
  std::vector<int> me_fast_idx, me_gen_idx;                           // reference slots used by the last ME call
  void *surf_dev = nullptr, *surf_jobs_dev = nullptr; size_t surf_cap = 0, surf_jobs_cap = 0;   // jmhip_distortion_surface
  void *ref_ptrs_dev = nullptr;                       // [0..31] integer recon, [32..63] quarter-pel plane stacks
  // jmhip_recon_to_ref swaps plane pointers; the one table entry that changes is written by the next kernel that can carry it
  // (interp_luma, which every pipeline launches next) instead of a launch of its own; jm_ensure_ref_table flushes it otherwise
  struct TableFix { int idx = -1; const uint8_t *ptr = nullptr; } table_fix;
  // frame pipeline (MC -> residual -> TQ -> recon): per-MB luma job/result, 2 chroma jobs/results, recon picture
  void *fr_jobs_y = nullptr, *fr_jobs_c = nullptr, *fr_res_y = nullptr, *fr_res_c = nullptr, *fr_quant = nullptr, *fr_modes = nullptr;
  int fr_capacity = 0, fr_n = 0;
  void *fr_rec = nullptr;                             // fused frame stage: one JmMbRes record per macroblock (frame_common.h)
  bool fr_fused = false;                              // the last jmhip_residual_frame took the fused kernel: results live in fr_rec
  void *fr_blk_ref = nullptr;                         // [n][4] reference slot per 8x8 block (frame stage fed from the slice search)
  bool fr_slices_t8 = false;                          // ... and some of them may carry the 8x8-transform flag (Transform8x8Mode in the slice search)
  bool fr_from_slices = false;                        // modes + per-block references of the frame stage were left on the device by jmhip_slice_to_frame
  void *fr_bi = nullptr; int fr_bi_n = 0, fr_bi_capacity = 0; unsigned fr_bi_mask = 0;   // second list of B macroblocks (jmhip_frame_bipred_set), device array
  jmhip_frame_bw fr_bw{};
  jmhip_frame_wp fr_wp{};                             // explicit weighted prediction of the frame stage (enable = 0: off)
  jmhip_quant fr_quant_host[4];
  uint8_t *rec_y = nullptr, *rec_u = nullptr, *rec_v = nullptr;
  uint8_t *pred_y = nullptr, *pred_u = nullptr, *pred_v = nullptr;   // jmhip_frame_keep_prediction: the prediction picture of the last fused frame stage
  bool keep_pred = false, pred_valid = false;
  bool rec_valid = false;                             // the recon planes hold a reconstruction (cleared by jmhip_recon_to_ref's plane swap)
  bool rec_has_pic = false;                           // recon planes loaded by jmhip_recon_upload
  void *dbk_dev = nullptr; size_t dbk_cap = 0;        // deblocking: macroblock / block / edge arrays
  void *dbr_dev = nullptr; size_t dbr_cap = 0; int dbk_sweeps = 0;   // deblocking, relaxation schedule: per-macroblock records, change flags, counters
  // TQ arrays
  void *tq_jobs_dev = nullptr, *tq_res_dev = nullptr, *tq_quant_dev = nullptr; int tq_capacity = 0, tq_qcap = 0;
  void *slice_state = nullptr;                         // me_wave.hip: the P-slice search state (field arrays, EPZS / UMHexagonS memories)
  void *xslice_state = nullptr;                        // me_xslice.hip: call records and work lists of the exhaustive searches' sweeps
  // timing
  bool timing = false;
  unsigned timing_mask = ~0u;                        // stages that record events while timing is on (jmhip_timing_select)
  double stage_ms[JMHIP_STAGE_COUNT] = {0};
  int stage_launches[JMHIP_STAGE_COUNT] = {0};
  struct PendingEvt { int stage; hipEvent_t a, b; };
  std::vector<PendingEvt> pending;
  std::vector<hipEvent_t> evt_pool;
  std::string err;
};

#define JM_HIP_CHECK(ctx, call)                                                              \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                        \
      return JMHIP_ERR_DEVICE;                                                               \
    }                                                                                        \
  } while (0)

static inline int jm_fail(jmhip_ctx *ctx, int code, const char *msg)
{
  if (ctx) ctx->err = msg;
  return code;
}

// RAII-less stage timer: call begin before the launches of a stage and end after them.
void jm_stage_begin(jmhip_ctx *ctx, int stage);
void jm_stage_end(jmhip_ctx *ctx, int stage);

int jm_upload_plane(jmhip_ctx *c, uint8_t *dst, const void *src, int w, int h, int pel_bytes, int stride, int device_ptrs);
int jm_download_planes(jmhip_ctx *c, const uint8_t *src, size_t n, void *out, int pel_bytes);

// kernels (one translation unit each)
int jm_launch_interp_luma(jmhip_ctx *ctx, int ref, int prow0 = 0, int prow1 = 0);
int jm_launch_interp_chroma(jmhip_ctx *ctx, int ref, int prow0 = 0, int prow1 = 0);
constexpr int JMHIP_TQ_SELECT = 0x100;   // frame stage: each luma kernel takes only the macroblocks of its transform size
int jm_launch_tq(jmhip_ctx *ctx, int kind, int yuv_format, const void *jobs, const void *quants, void *results, int n);
int jm_ensure_ref_table(jmhip_ctx *ctx);
int jm_ensure_recon(jmhip_ctx *ctx);                                                       // jmhip_ctx.hip: all three recon planes or none
int jm_flush_table_fix(jmhip_ctx *ctx);
void jm_slice_state_free(jmhip_ctx *ctx);
void jm_xslice_free(jmhip_ctx *ctx);                                                       // me_xslice.hip
bool jm_xslice_covers(const jmhip_slice_params *prm);
int jm_xslice_run(jmhip_ctx *ctx, const jmhip_slice_params *prm, int8_t *ref_idx, short *mv, jmhip_mb_inter *out, int *passes, int *settled);
int jm_me_arrays_ensure(jmhip_ctx *ctx, int n);                                            // me_int.hip
int jm_frame_buffers_ensure(jmhip_ctx *ctx, int n);                                        // frame.hip                                                   // me_wave.hip                                                    // frame.hip
struct MeDev;
int jm_me_sub_tables(jmhip_ctx *ctx);                                                   // me_sub.hip
void jm_launch_me_sub(jmhip_ctx *ctx, const MeDev &P, const jmhip_me_mb *jobs_dev, jmhip_me_result *res_dev, int n);
// device-resident work lists (me_xslice.hip): the pair-lane integer search over items, the refinement over (job index, partition mask) pairs
int jm_me_pair_geometry(jmhip_ctx *ctx, int R, MeDev *P, size_t *lds);                     // me_int.hip
void jm_launch_me_pair_list(jmhip_ctx *ctx, const MeDev &P, const MeDev *P_dev, size_t lds, const jmhip_me_mb *jobs_dev, const int *idx_dev, jmhip_me_result *res_dev, const int *cnt_dev, int cap, int grid);
void jm_launch_me_sub_list(jmhip_ctx *ctx, const MeDev &P, const jmhip_me_mb *jobs_dev, jmhip_me_result *res_dev, const int *list_dev, const unsigned long long *masks_dev, const int *cnt_dev, int cap, int grid);
// me_metric.hip: the search chain for every error metric / the chroma term (jmhip_me_params.metric_set)
static inline bool jm_me_metric_path(const jmhip_me_params *prm)
{ return prm->metric_set && (prm->metric[0] != 0 || prm->metric[1] != 2 || prm->metric[2] != 2 || prm->chroma_me != 0); }
int jm_me_metric_check(jmhip_ctx *ctx, const jmhip_me_params *prm, unsigned ref_mask, const char *who);
int jm_launch_me_metric(jmhip_ctx *ctx, const jmhip_me_params *prm, const MeDev &P, const jmhip_me_mb *jobs_dev, const int *idx_dev,
                        jmhip_me_result *res_dev, int n, size_t lds, int skip_int);

// Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8) and every XCD has its own L2. Kernels whose neighbouring
// work items share data (adjacent macroblocks: overlapping reference windows) take their item through this mapping, which
// hands each XCD one contiguous run of items. Launch jm_xcd_grid(n) workgroups; blocks past the end get -1 and leave.
__host__ __device__ static inline int jm_xcd_grid(int n) { return ((n + 7) / 8) * 8; }
__device__ __forceinline__ int jm_xcd_item_of(int block, int n)
{
  const int per = (n + 7) >> 3;
  const int i = (block & 7) * per + (block >> 3);
  return i < n ? i : -1;
}
__device__ __forceinline__ int jm_xcd_item(int n) { return jm_xcd_item_of((int)blockIdx.x, n); }

// Device-built work lists in JM_SHARDS shards (me_xslice.hip): producers append to the shard of their macroblock address, so that thousands of
// workgroups do not queue on one counter (one word takes ~88 atomics per microsecond: 8160 producers x 6 counters cost 0.5 ms). Shard s keeps
// its count at cnt[s * JM_SHARD_STRIDE] (a cache line of its own) and its entries at list[s * cap + 0 ..]. A consumer walks VIRTUAL slots
// v = offset * JM_SHARDS + shard, 0 <= v < JM_SHARDS * max count; jm_shard_slots() is that bound (every lane of the calling wave gets it).
constexpr int JM_SHARDS = 64, JM_SHARD_STRIDE = 16;
__device__ __forceinline__ int jm_shard_slots(const int *cnt)
{
  int m = cnt[(threadIdx.x & 63) * JM_SHARD_STRIDE];
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) m = max(m, __shfl_xor(m, o));
  return __builtin_amdgcn_readfirstlane(m) * JM_SHARDS;
}
// the list entry of virtual slot v, or -1 when the shard has no such entry
__device__ __forceinline__ int jm_shard_entry(const int *cnt, const int *list, int cap, int v)
{
  const int s = v & (JM_SHARDS - 1), o = v / JM_SHARDS;
  return o < cnt[s * JM_SHARD_STRIDE] ? list[(size_t)s * cap + o] : -1;
}
