// me_int.hip -- block motion estimation, integer-pel stage (SAD full search) and the jmhip_me_frame entry points.
// (Sub-pel refinement: me_sub.hip. Distortion batch / surfaces / bi-pred / decision costs: me_aux.hip. Shared helpers: me_common.h.)
//
// Replaces, per partition, the chain BlockMotionSearch runs from the predictor on (lencod/src/mv-search.c:751-826):
//   search centre                      mv-search.c:752-762 (FullSearch) / me_fullfast.c:550-566 (FastFullSearch)
//   FullPelBlockMotionSearch           me_fullsearch.c:47-155     (SearchMode -1)
//   FastFullPelBlockMotionSearch       me_fullfast.c:833-903 on the SAD surface of SetupFastFullPelSearch :491 (SearchMode 0)
//   computeSAD / computeSATD           me_distortion.c:351 / :657 (HadamardSAD4x4 :182, HadamardSAD8x8 :272)
//   SubPelBlockMotionSearch            me_fullsearch.c:341-511
//   MV_COST_SMP / mvbits / spiral      defines.h:126-128, mv-search.c:333-341, :366-393
//
// Bit-exactness notes (each checked against the oracle, which is pinned by the real JM):
//  * JM's candidate loop is sequential with strict '<' updates, so the result is argmin over (cost, spiral index):
//    the kernel reduces packed keys (cost << 13 | tie) with the spiral index as the tie field. The early exits
//    (me_distortion.c:373, :690) return partial sums that are never accepted, so full evaluation is identical.
//  * Integer-pel reference samples: UMV access clamps the BLOCK ORIGIN into the 20-pel replicated ring
//    (refbuf.c:37-43); for the integer plane that equals clamping every sample coordinate to the picture, so the
//    search window is staged from the unpadded recon with per-sample clamping (no padded plane needed).
//  * check_for_00 compares a quarter-pel coordinate with a pel coordinate (me_fullsearch.c:129): mirrored.
//  * FastFullSearch evaluates the (0,0) vector first when !rdopt (me_fullfast.c:867-876): it wins all ties.
//  * Sub-pel: 4x4/8x8 sub-block origins are clamped one by one under UMV access (me_distortion.c:678); FAST vs
//    UMV is decided per phase as me_fullsearch.c:412-420, :468-476.
//
// Kernels of this file:
//   me_int_kernel        generic integer search: any partition mask, any spread of centres (union window in LDS, lane <-> candidate)
//   me_int_fast_kernel   one lane per candidate column, rolling register window, 2R+1 >= 64 (JMHIP_ME_KERNEL=single)
//   me_int_pair_kernel   DEFAULT: a candidate's 16x16 block split across a lane pair, 3 waves/SIMD, 2R+1 >= 32
// Roofline: ~10^3 integer ops per byte of compulsory traffic -> integer-VALU bound, not HBM bound.
#include "me_common.h"

namespace {

// ------------------------------------------------------------------------------------------------ integer search

// JM hands computeSAD the bound  min_mcost - mcost  (me_fullsearch.c:138). At spiral position 0 min_mcost is still INT_MAX,
// so a NEGATIVE mcost -- possible only through the check_for_00 bonus, i.e. for the zero vector of the 16x16 block at picture
// position (0,0) when it is the search centre -- wraps the bound below zero, and computeSAD leaves after its first row
// (me_distortion.c:364-375). That candidate then competes with  mv cost - bonus + SAD(row 0)  and, being first in the
// spiral, keeps ties. Found by running the real encoder on the device path (tests/test_jm_shim.py); mirrored here.
__device__ __forceinline__ void wrapped_bound_00v(const MeDev &P, int mb_x, int mb_y, int ref, int ref_is_0, int pmx0, int pmy0, int cx, int cy, int *mvx, int *mvy, int *cost)
{
  if (P.mode != JMHIP_SEARCH_FULL || P.rdopt || P.is_b || !ref_is_0 || mb_x || mb_y || cx || cy) return;
  const int c0 = mv_cost(P.lam_f, -pmx0, -pmy0) - ((P.lam_f * 16) >> 16);
  if (c0 >= 0) return;
  const uint8_t *ref_y = P.ref_y[ref];
  int row = 0;
  for (int x = 0; x < 16; x++) {
    int v = ref_y[x];
    if (P.wp_on) v = min(max(((P.wp_w[ref] * v + P.wp_round) >> P.wp_denom) + P.wp_o[ref], 0), 255);     // computeSADWP's row (me_distortion.c:431)
    row += iabs((int)P.cur[x] - v);
  }
  if (c0 + row <= *cost) { *mvx = 0; *mvy = 0; *cost = c0 + row; }
}
__device__ __forceinline__ void wrapped_bound_00(const MeDev &P, const jmhip_me_mb &job, int cx, int cy, int *mvx, int *mvy, int *cost)
{
  if (P.mode != JMHIP_SEARCH_FULL) return;
  wrapped_bound_00v(P, job.mb_x, job.mb_y, job.ref, job.ref_is_0, job.pred_mv[0][0], job.pred_mv[0][1], cx, cy, mvx, mvy, cost);
}

__global__ __launch_bounds__(256) void me_int_kernel(MeDev P, const jmhip_me_mb *__restrict__ jobs, const int *__restrict__ job_index,
                                                     jmhip_me_result *__restrict__ res_all, int n_items)
{
  const int item = jm_xcd_item(n_items);
  if (item < 0) return;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ int s_cx[JMHIP_NPART], s_cy[JMHIP_NPART], s_px[JMHIP_NPART], s_py[JMHIP_NPART];
  __shared__ int s_u[8];                           // umin_x, umin_y, UW, UH, uniform
  __shared__ uint32_t s_cur[64];
  __shared__ unsigned s_red[JMHIP_NPART][4];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mbi = job_index[item];
  const jmhip_me_mb &job = jobs[mbi];
  jmhip_me_result *res = res_all + mbi;
  const int mbx = job.mb_x, mby = job.mb_y;
  const unsigned long long mask = P.mask;

  if (tid < JMHIP_NPART) {
    const int src = (P.mode == JMHIP_SEARCH_FASTFULL) ? 0 : tid;
    int cx, cy;
    search_center(P, job.pred_mv[src][0], job.pred_mv[src][1], &cx, &cy);
    s_cx[tid] = cx; s_cy[tid] = cy;
    s_px[tid] = job.pred_mv[tid][0]; s_py[tid] = job.pred_mv[tid][1];
  }
  if (tid < 64) {
    const int r = tid >> 2, k = tid & 3;
    s_cur[tid] = *reinterpret_cast<const uint32_t *>(P.cur + (size_t)(mby * 16 + r) * P.W + mbx * 16 + k * 4);
  }
  __syncthreads();
  if (tid == 0) {
    int x0 = 1 << 30, x1 = -(1 << 30), y0 = 1 << 30, y1 = -(1 << 30), uni = 1, fx = 0, fy = 0, first = 1;
    for (int p = 0; p < JMHIP_NPART; p++) if ((mask >> p) & 1) {
      x0 = min(x0, s_cx[p]); x1 = max(x1, s_cx[p]); y0 = min(y0, s_cy[p]); y1 = max(y1, s_cy[p]);
      if (first) { fx = s_cx[p]; fy = s_cy[p]; first = 0; } else if (s_cx[p] != fx || s_cy[p] != fy) uni = 0;
    }
    s_u[0] = x0 - P.R; s_u[1] = y0 - P.R; s_u[2] = x1 - x0 + 2 * P.R + 1; s_u[3] = y1 - y0 + 2 * P.R + 1; s_u[4] = uni;
    s_u[5] = fx; s_u[6] = fy;
  }
  __syncthreads();
  const int umin_x = s_u[0], umin_y = s_u[1], UW = s_u[2], UH = s_u[3], uniform = s_u[4];
  const int pitch = P.win_pitch;
  // the host sized the window for the worst MB of the launch; a larger one would overflow LDS
  if (UW + 15 > pitch || UH + 15 > P.win_rows) { if (tid < JMHIP_NPART) res[0].cost_int[tid] = -2; return; }

  // ---- stage the reference window (integer recon, per-sample clamp == JM's padded plane + UMV origin clamp)
  {
    const uint8_t *ref = P.ref_y[job.ref];
    const int bx = mbx * 16 + umin_x, by = mby * 16 + umin_y;
    const int nw = (UW + 15 + 3) >> 2;             // dwords per row
    for (int d = tid; d < nw * (UH + 15); d += 256) {
      const int y = d / nw, xw = d - y * nw;
      const uint8_t *row = ref + (size_t)clampi(by + y, 0, P.H - 1) * P.W;
      uint32_t v = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) v |= (uint32_t)row[clampi(bx + xw * 4 + k, 0, P.W - 1)] << (8 * k);
      if (P.wp_on) v = wp_apply4(v, P.wp_w[job.ref], P.wp_o[job.ref], P.wp_round, P.wp_denom);      // the weighted window: every SAD below sees weighted samples
      *reinterpret_cast<uint32_t *>(smem + (size_t)y * pitch + xw * 4) = v;
    }
  }
  __syncthreads();

  unsigned best[JMHIP_NPART];
#pragma unroll
  for (int p = 0; p < JMHIP_NPART; p++) best[p] = KEY_INVALID;

  const int w16 = (P.lam_f * 16) >> 16;            // WEIGHTED_COST(lambda,16)
  const int quirk00 = (P.mode == JMHIP_SEARCH_FULL) && !P.rdopt && !P.is_b && job.ref_is_0;   // check_for_00, :75
  const int ff00 = (P.mode == JMHIP_SEARCH_FASTFULL) && !P.rdopt;                            // pos_00 pre-check
  const int ucx = s_u[5], ucy = s_u[6];            // the common centre when all active partitions agree

  for (int c = tid; c < UW * UH; c += 256) {
    const int ay = c / UW, ax = c - ay * UW;
    const int mvx = umin_x + ax, mvy = umin_y + ay;       // candidate motion vector (pel)
    // ---- sixteen 4x4 SADs of the 16x16 block at this displacement
    unsigned sad[16];
#pragma unroll
    for (int b = 0; b < 16; b++) sad[b] = 0;
    const uint8_t *wrow = smem + (size_t)ay * pitch + (ax & ~3);
    const unsigned sh = ax & 3;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const uint32_t *wp = reinterpret_cast<const uint32_t *>(wrow + (size_t)r * pitch);
      const uint32_t d0 = wp[0], d1 = wp[1], d2 = wp[2], d3 = wp[3], d4 = wp[4];
      const uint32_t r0 = __builtin_amdgcn_alignbyte(d1, d0, sh), r1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
      const uint32_t r2 = __builtin_amdgcn_alignbyte(d3, d2, sh), r3 = __builtin_amdgcn_alignbyte(d4, d3, sh);
      const int b = (r >> 2) * 4;
      sad[b + 0] = __builtin_amdgcn_sad_u8(r0, s_cur[r * 4 + 0], sad[b + 0]);
      sad[b + 1] = __builtin_amdgcn_sad_u8(r1, s_cur[r * 4 + 1], sad[b + 1]);
      sad[b + 2] = __builtin_amdgcn_sad_u8(r2, s_cur[r * 4 + 2], sad[b + 2]);
      sad[b + 3] = __builtin_amdgcn_sad_u8(r3, s_cur[r * 4 + 3], sad[b + 3]);
    }
    // ---- SetupLargerBlocks (me_fullfast.c:210-273): partition SADs in partition-table order
    unsigned ps[JMHIP_NPART];
#pragma unroll
    for (int b8 = 0; b8 < 4; b8++) {
      const int o = 8 * (b8 >> 1) + 2 * (b8 & 1);         // top-left 4x4 index of the 8x8
      const unsigned a = sad[o], b = sad[o + 1], cc = sad[o + 4], d = sad[o + 5];
      ps[25 + 4 * b8 + 0] = a; ps[25 + 4 * b8 + 1] = b; ps[25 + 4 * b8 + 2] = cc; ps[25 + 4 * b8 + 3] = d;   // 4x4
      ps[9 + 2 * b8 + 0] = a + b; ps[9 + 2 * b8 + 1] = cc + d;                                            // 8x4
      ps[17 + 2 * b8 + 0] = a + cc; ps[17 + 2 * b8 + 1] = b + d;                                          // 4x8
      ps[5 + b8] = a + b + cc + d;                                                                       // 8x8
    }
    ps[1] = ps[5] + ps[6]; ps[2] = ps[7] + ps[8];          // 16x8 top, bottom
    ps[3] = ps[5] + ps[7]; ps[4] = ps[6] + ps[8];          // 8x16 left, right
    ps[0] = ps[1] + ps[2];                                 // 16x16

    int upos = 0;
    if (uniform) {
      const int dx = mvx - ucx, dy = mvy - ucy;           // always inside: the union IS the window
      upos = spiral_pos(dx, dy) + 1;
      if (ff00 && mvx == 0 && mvy == 0) upos = 0;         // pos_00 is evaluated first: wins every tie
    }
#pragma unroll
    for (int p = 0; p < JMHIP_NPART; p++) {
      if (!((mask >> p) & 1)) continue;
      int tie = upos;
      bool valid = true;
      if (!uniform) {
        const int dx = mvx - s_cx[p], dy = mvy - s_cy[p];
        valid = (iabs(dx) <= P.R) && (iabs(dy) <= P.R);
        tie = spiral_pos(dx, dy) + 1;
        if (ff00 && mvx == 0 && mvy == 0) tie = 0;
      }
      int cost = mv_cost(P.lam_f, 4 * mvx - s_px[p], 4 * mvy - s_py[p]) + (int)ps[p];
      if (p == 0) {
        // check_for_00 (me_fullsearch.c:129-132): cand_x (quarter-pel, absolute) == pic_pix_x (pel); keys are biased by w16
        if (quirk00 && 4 * (mbx * 16 + mvx) == mbx * 16 && 4 * (mby * 16 + mvy) == mby * 16) cost -= w16;
        cost += w16;
      }
      const unsigned key = valid ? (((unsigned)cost << TIE_BITS) | (unsigned)tie) : KEY_INVALID;
      best[p] = min(best[p], key);
    }
  }

  // ---- reduce the 41 minima over the workgroup
#pragma unroll
  for (int p = 0; p < JMHIP_NPART; p++) {
    unsigned v = best[p];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = min(v, (unsigned)__shfl_xor((int)v, off, 64));
    if (lane == 0) s_red[p][wave] = v;
  }
  __syncthreads();
  if (tid < JMHIP_NPART) {
    const int p = tid;
    jmhip_me_result &o = res[0];
    if ((mask >> p) & 1) {
      const unsigned k = min(min(s_red[p][0], s_red[p][1]), min(s_red[p][2], s_red[p][3]));
      int cost = (int)(k >> TIE_BITS), tie = (int)(k & ((1u << TIE_BITS) - 1));
      if (p == 0) cost -= w16;
      int mvx, mvy;
      if (tie == 0) { mvx = 0; mvy = 0; }               // FastFull pos_00
      else { int dx, dy; spiral_offset(tie - 1, &dx, &dy); mvx = s_cx[p] + dx; mvy = s_cy[p] + dy; }
      if (p == 0) wrapped_bound_00(P, job, s_cx[0], s_cy[0], &mvx, &mvy, &cost);
      o.mv_int[p][0] = (int16_t)mvx; o.mv_int[p][1] = (int16_t)mvy; o.cost_int[p] = cost;
      if (!P.subpel) { o.mv[p][0] = (int16_t)(mvx << 2); o.mv[p][1] = (int16_t)(mvy << 2); o.cost[p] = cost; }
    } else {
      o.mv_int[p][0] = o.mv_int[p][1] = 0; o.cost_int[p] = -1;
      o.mv[p][0] = o.mv[p][1] = 0; o.cost[p] = -1;
    }
  }
}


// ------------------------------------------------------------------------------------------------ integer search, fast path
//
// All 41 partitions share ONE search centre (always true for FastFullSearch; for FullSearch whenever the partition
// predictors agree to the pel) and all partitions are searched. Mapping:
//   lane  <-> candidate column (64 columns per wave-row; columns >= 64 take the per-candidate path below)
//   wave  <-> band of candidate rows; a lane walks DOWN its column, so consecutive candidates share 15 of their 16
//             reference rows: the window lives in a rolling 16-row register file (static indices through a 16x
//             unrolled loop) and only one new row (4 dwords) is fetched from LDS per candidate
//   LDS   <-> four byte-shifted copies of the window, so lanes read dword-aligned words whatever their column
//             (copy stride = 8 mod 32 dwords: the 32 lanes of a read group hit 32 different banks)
//   SAD   <-> v_sad_hi_u8: the 16 4x4 SADs come out already shifted by 16, so a partition key is
//             (sad << 16) + (mvcost << 16) + tie = ONE v_add3_u32, and the running minimum ONE v_min_u32
//   MV cost per (lane, partition) = lambda*(bits_x[lane] + bits_y[row]) >> 16 is cached in a register and only
//             recomputed on rows where bits_y changes for that partition (a wave-uniform test against a per-row
//             bitmask prepared in LDS): mvbits is a step function of log2|d|
// The 16x16 partition keeps a 64-bit key (cost up to 65280 + mvcost does not fit 16 bits).
constexpr int FAST_TIE_BITS = 16;

#ifdef JMHIP_STAMPS
#define STAMP(k) do { if (P.stamps && blockIdx.x < 512 && lane == 0) P.stamps[(blockIdx.x * 4 + wave) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

// min over the 64 lanes of a wave without touching LDS: xor-butterfly inside each row of 16 lanes with DPP
// (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror), then the four row results through v_readlane
__device__ __forceinline__ unsigned row16_min_u32(unsigned v)
{
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false));
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false));
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false));
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, false));
  return v;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
  v = row16_min_u32(v);
  const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return min(min(a, b), min(c, d));
}
__device__ __forceinline__ unsigned long long min_u64(unsigned long long a, unsigned long long b) { return b < a ? b : a; }
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v, int ctrl_sel)
{
  unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
  switch (ctrl_sel) {
  case 0: lo = __builtin_amdgcn_update_dpp((int)lo, (int)lo, 0xB1, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp((int)hi, (int)hi, 0xB1, 0xf, 0xf, false); break;
  case 1: lo = __builtin_amdgcn_update_dpp((int)lo, (int)lo, 0x4E, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp((int)hi, (int)hi, 0x4E, 0xf, 0xf, false); break;
  case 2: lo = __builtin_amdgcn_update_dpp((int)lo, (int)lo, 0x141, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp((int)hi, (int)hi, 0x141, 0xf, 0xf, false); break;
  default: lo = __builtin_amdgcn_update_dpp((int)lo, (int)lo, 0x140, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp((int)hi, (int)hi, 0x140, 0xf, 0xf, false); break;
  }
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#pragma unroll
  for (int k = 0; k < 4; k++) v = min_u64(v, dpp_u64(v, k));
  unsigned long long r = ~0ull;
#pragma unroll
  for (int l = 0; l < 64; l += 16) {
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, l), hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), l);
    r = min_u64(r, ((unsigned long long)hi << 32) | lo);
  }
  return r;
}

struct FastShared {
  int cx, cy, px[JMHIP_NPART], py[JMHIP_NPART];
  uint32_t cur[64];
  unsigned chg[160];                            // per candidate row: does any partition's vertical mv bits differ from the row above
  uint8_t bytab[160][48] __attribute__((aligned(16)));   // vertical mv bits per candidate row and partition (48: three 16-byte reads)
  uint8_t bxtab[84][44];                        // horizontal mv bits per candidate column and partition (44: dword rows); 64 main-grid columns + the rest (2R+1 <= 81)
  unsigned part[JMHIP_NPART][4];                // partial minima of the final reduction
  unsigned long long part0[4];
};

__device__ __forceinline__ int spiral_base_A(int dy)   // + 2*dx gives pos when |dy| > |dx|
{
  const int l = iabs(dy);
  return (2 * l - 1) * (2 * l - 1) + 2 * (l - 1) + (dy > 0 ? 1 : 0);
}
__device__ __forceinline__ int spiral_base_B(int dx)   // + 2*dy gives pos when |dx| >= |dy|
{
  const int l = iabs(dx);
  if (l == 0) return 0;
  return (2 * l - 1) * (2 * l - 1) + 2 * (2 * l - 1) + 2 * l + (dx > 0 ? 1 : 0);
}

__global__ __launch_bounds__(256, 2) void me_int_fast_kernel(MeDev P, const jmhip_me_mb *__restrict__ jobs, const int *__restrict__ job_index,
                                                            jmhip_me_result *__restrict__ res, int n_items)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t swin[];      // 4 shifted window copies
  __shared__ FastShared S;
  const int item = jm_xcd_item(n_items);
  if (item < 0) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  STAMP(0);
  // item = macroblock | representative partition << 24: one item per DISTINCT search centre of the macroblock. All 41
  // partitions are evaluated around that centre; only those whose own centre it is are written (FastFull: all of them).
  const int mbi = job_index[item] & 0xffffff, rep = (job_index[item] >> 24) & 63;
  const jmhip_me_mb &job = jobs[mbi];
  const int mbx = job.mb_x, mby = job.mb_y;
  const int R = P.R, UW = 2 * R + 1, UH = UW;
  const int PITCH = P.win_pitch >> 2;            // dwords per window row
  const int CS = P.win_copy_stride;              // dwords between shifted copies (== 8 mod 32)
  const int WROWS = UH + 15;

  // every lane derives the centre itself (uniform scalar work): no LDS round trip before the window loads can start
  int ucx, ucy;
  search_center(P, job.pred_mv[rep][0], job.pred_mv[rep][1], &ucx, &ucy);
  const int umin_x = ucx - R, umin_y = ucy - R;
  if (tid < JMHIP_NPART) { S.px[tid] = job.pred_mv[tid][0]; S.py[tid] = job.pred_mv[tid][1]; }

  // ---- reference window: copy 0 plus three byte-shifted copies, staged from the integer recon with per-sample clamping.
  //      Lane slots are (row = tid/32 + 8*i, dword column = tid%32): no runtime divisions. Away from the left/right picture
  //      edge the lane fetches three aligned dwords per slot and builds all four copies in registers; the loads are issued
  //      first and the mv-bits tables are computed while they are in flight.
  const uint8_t *ref = P.ref_y[job.ref];
  const int wpw = P.wp_w[job.ref], wpo = P.wp_o[job.ref];      // weighted reference ME (P.wp_on)
  const int bx = mbx * 16 + umin_x, by = mby * 16 + umin_y;
  const int xw = tid & 31, y0 = tid >> 5;
  const bool inside = bx >= 4 && bx + PITCH * 4 + 8 <= P.W;       // no horizontal clamping anywhere in the window (+ the third dword)
  constexpr int NB = 12;                                           // row slots per lane: 8 * 12 = 96 >= window rows (R <= 40)
  uint32_t q0[NB], q1[NB], q2[NB];
  if (inside && xw < PITCH) {
#pragma unroll
    for (int u = 0; u < NB; u++) {
      const int y = min(y0 + 8 * u, WROWS - 1);
      const uintptr_t a = reinterpret_cast<uintptr_t>(ref + (size_t)clampi(by + y, 0, P.H - 1) * P.W + bx + xw * 4);
      const uint32_t *q = reinterpret_cast<const uint32_t *>(a & ~uintptr_t(3));
      q0[u] = q[0]; q1[u] = q[1]; q2[u] = q[2];
    }
  }
  // mv bits tables: vertical [row][partition], horizontal [column][partition]; lane slots (index = tid/64 + 4*i, partition = tid%64)
  {
    const int p = tid & 63, i0 = tid >> 6;
    if (p < 44) {
      const int py = p < JMHIP_NPART ? job.pred_mv[p][1] : 0, px = p < JMHIP_NPART ? job.pred_mv[p][0] : 0;
      for (int row = i0; row < UH; row += 4) S.bytab[row][p] = p < JMHIP_NPART ? (uint8_t)mvbits(4 * (umin_y + row) - py) : 0;
      for (int c = i0; c < UW; c += 4) S.bxtab[c][p] = p < JMHIP_NPART ? (uint8_t)mvbits(4 * (umin_x + c) - px) : 0;
    }
  }
  if (xw < PITCH) {
    if (inside) {
      const unsigned sh = (unsigned)(bx & 3);
#pragma unroll
      for (int u = 0; u < NB; u++)
        if (y0 + 8 * u < WROWS) {
          uint32_t a = __builtin_amdgcn_alignbyte(q1[u], q0[u], sh), b = __builtin_amdgcn_alignbyte(q2[u], q1[u], sh);
          if (P.wp_on) { a = wp_apply4(a, wpw, wpo, P.wp_round, P.wp_denom); b = wp_apply4(b, wpw, wpo, P.wp_round, P.wp_denom); }
          uint32_t *w = swin + (y0 + 8 * u) * PITCH + xw;
          w[0] = a;
          w[1 * CS] = __builtin_amdgcn_alignbyte(b, a, 1u);
          w[2 * CS] = __builtin_amdgcn_alignbyte(b, a, 2u);
          w[3 * CS] = __builtin_amdgcn_alignbyte(b, a, 3u);
        }
    } else {
#pragma unroll
      for (int u = 0; u < NB; u++)
        if (y0 + 8 * u < WROWS) {
          const uint8_t *row = ref + (size_t)clampi(by + y0 + 8 * u, 0, P.H - 1) * P.W;
          uint32_t v = 0;
#pragma unroll
          for (int k = 0; k < 4; k++) v |= (uint32_t)row[clampi(bx + xw * 4 + k, 0, P.W - 1)] << (8 * k);
          if (P.wp_on) v = wp_apply4(v, wpw, wpo, P.wp_round, P.wp_denom);
          swin[(y0 + 8 * u) * PITCH + xw] = v;
        }
    }
  }
  __syncthreads();
  if (!inside && xw < PITCH - 1)                     // picture-edge macroblocks: shifted copies from copy 0
    for (int y = y0; y < WROWS; y += 8) {
      const uint32_t a = swin[y * PITCH + xw], b = swin[y * PITCH + xw + 1];
      swin[1 * CS + y * PITCH + xw] = __builtin_amdgcn_alignbyte(b, a, 1u);
      swin[2 * CS + y * PITCH + xw] = __builtin_amdgcn_alignbyte(b, a, 2u);
      swin[3 * CS + y * PITCH + xw] = __builtin_amdgcn_alignbyte(b, a, 3u);
    }
  if (tid < UH) {
    unsigned m = (tid == 0);
    if (tid) {
      const uint32_t *a = reinterpret_cast<const uint32_t *>(S.bytab[tid]), *b = reinterpret_cast<const uint32_t *>(S.bytab[tid - 1]);
#pragma unroll
      for (int g = 0; g < 11; g++) m |= (a[g] != b[g]);
    }
    S.chg[tid] = m;
  }
  __syncthreads();

  STAMP(1);
  // ---- per-lane constants
  const int col = lane;                              // candidate column of the main grid
  const int mvx = umin_x + col;
  const int dx = mvx - ucx, adx = iabs(dx);
  const int tieB = spiral_base_B(dx) + 1;            // tie = spiral index + 1 (0 is reserved for FastFull's pos_00)
  const int twodx = 2 * dx;
  uint32_t bxp[11];                                  // horizontal mv bits of the 41 partitions, 4 per dword
#pragma unroll
  for (int g = 0; g < 11; g++) bxp[g] = reinterpret_cast<const uint32_t *>(S.bxtab[col])[g];
  const uint32_t *lbase = swin + (col & 3) * CS + (col >> 2);

  const int lam = P.lam_f;
  const int w16 = (lam * 16) >> 16;
  const int quirk00 = (P.mode == JMHIP_SEARCH_FULL) && !P.rdopt && !P.is_b && job.ref_is_0;
  const int ff00 = (P.mode == JMHIP_SEARCH_FASTFULL) && !P.rdopt;
  const bool qx = quirk00 && (4 * (mbx * 16 + mvx) == mbx * 16);       // check_for_00 column test (me_fullsearch.c:129)

  unsigned best[JMHIP_NPART];                        // best[0] unused (64-bit key below)
#pragma unroll
  for (int p = 0; p < JMHIP_NPART; p++) best[p] = KEY_INVALID;
  unsigned long long best0 = ~0ull;
  unsigned mvc[JMHIP_NPART];                         // cached (mv cost << 16) per partition; mvc[0] unshifted + bias

  const int r0 = (UH * wave) >> 2, r1 = (UH * (wave + 1)) >> 2, nrows = r1 - r0;
  uint32_t win[16][4];
#pragma unroll
  for (int j = 0; j < 15; j++) {
    const uint32_t *wp = lbase + (r0 + j) * PITCH;
    win[j][0] = wp[0]; win[j][1] = wp[1]; win[j][2] = wp[2]; win[j][3] = wp[3];
  }

  // cur MB in scalar registers: v_sad_hi_u8 takes an SGPR operand
  uint32_t curs[64];
  {
    const uint32_t *cm = reinterpret_cast<const uint32_t *>(P.cur + (size_t)(mby * 16) * P.W + mbx * 16);   // uniform address: s_load
    const int cw = P.W >> 2;
#pragma unroll
    for (int i = 0; i < 64; i++) curs[i] = __builtin_amdgcn_readfirstlane(cm[(i >> 2) * cw + (i & 3)]);
  }

  unsigned mnext = 0;
  // one candidate (window row offset TT inside the current 16-row chunk): TT is a compile-time constant so that the
  // rolling window slots (TT + r) & 15 are static register names
  auto step = [&](auto ttc, int t0) __attribute__((always_inline)) {
    constexpr int tt = decltype(ttc)::value;
    const int t = t0 + tt;
    if (t >= nrows) return;
    const int row = r0 + t;
    {                                                // the one new window row of this candidate
      const uint32_t *wp = lbase + (row + 15) * PITCH;
      constexpr int sl = (tt + 15) & 15;
      win[sl][0] = wp[0]; win[sl][1] = wp[1]; win[sl][2] = wp[2]; win[sl][3] = wp[3];
    }
    const int mvy = umin_y + row, dy = mvy - ucy;
    // ---- refresh the cached mv costs on rows where some partition's vertical bits change (wave-uniform test,
    //      flag fetched one step ahead): packed-byte add of the horizontal and vertical bits of 4 partitions at a time
    {
      const unsigned m = (t == 0) ? 1u : mnext;
      mnext = S.chg[min(row + 1, UH - 1)];
      if (__builtin_amdgcn_readfirstlane(m)) {
        const uint4 *bt = reinterpret_cast<const uint4 *>(S.bytab[row]);
        const uint4 b0 = bt[0], b1 = bt[1], b2 = bt[2];
        const uint32_t byp[11] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z};
#pragma unroll
        for (int g = 0; g < 11; g++) {
          const uint32_t sum4 = bxp[g] + byp[g];             // no carries: bits <= 2*25
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int p = 4 * g + k;
            if (p < JMHIP_NPART) {
              const unsigned prod = __umul24((unsigned)lam, (sum4 >> (8 * k)) & 255u);
              mvc[p] = p ? (prod & 0xffff0000u) : ((prod >> 16) + (unsigned)w16);
            }
          }
        }
      }
    }
    // ---- tie = spiral index + 1 from the per-lane / per-row halves. It SEEDS the sixteen 4x4 accumulators (free: the
    //      first v_sad_hi_u8 of a block takes it as its addend), so a partition made of k 4x4 blocks carries k*tie in the
    //      low 16 bits of its sum: still ordered by tie, k*tie <= 8*(81*81+1) < 2^16, and the key needs no tie add.
    const int ady = iabs(dy);
    unsigned tie = (ady > adx) ? (unsigned)(spiral_base_A(dy) + 1 + twodx) : (unsigned)(tieB + 2 * dy);
    if (ff00 && mvx == 0 && mvy == 0) tie = 0;
    // ---- sixteen 4x4 SADs, pre-shifted by 16 (v_sad_hi_u8)
    unsigned sad[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int sl = (tt + r) & 15, b = (r >> 2) * 4;
#pragma unroll
      for (int k = 0; k < 4; k++)
        sad[b + k] = __builtin_amdgcn_sad_hi_u8(win[sl][k], curs[r * 4 + k], (r & 3) ? sad[b + k] : tie);
    }
    unsigned ps[JMHIP_NPART];
#pragma unroll
    for (int b8 = 0; b8 < 4; b8++) {
      const int o = 8 * (b8 >> 1) + 2 * (b8 & 1);
      const unsigned a = sad[o], b = sad[o + 1], cc = sad[o + 4], d = sad[o + 5];
      ps[25 + 4 * b8 + 0] = a; ps[25 + 4 * b8 + 1] = b; ps[25 + 4 * b8 + 2] = cc; ps[25 + 4 * b8 + 3] = d;
      ps[9 + 2 * b8 + 0] = a + b; ps[9 + 2 * b8 + 1] = cc + d;
      ps[17 + 2 * b8 + 0] = a + cc; ps[17 + 2 * b8 + 1] = b + d;
      ps[5 + b8] = a + b + cc + d;
    }
    ps[1] = ps[5] + ps[6]; ps[2] = ps[7] + ps[8];
    ps[3] = ps[5] + ps[7]; ps[4] = ps[6] + ps[8];
#pragma unroll
    for (int p = 1; p < JMHIP_NPART; p++) best[p] = min(best[p], ps[p] + mvc[p]);
    {
      unsigned c0 = ((ps[1] >> 16) + (ps[2] >> 16)) + mvc[0];
      if (qx && 4 * (mby * 16 + mvy) == mby * 16) c0 -= (unsigned)w16;
      const unsigned long long k0 = ((unsigned long long)c0 << 32) | tie;
      best0 = k0 < best0 ? k0 : best0;
    }
  };
  STAMP(2);
  for (int t0 = 0; t0 < nrows; t0 += 16) {
    step(std::integral_constant<int, 0>{}, t0);  step(std::integral_constant<int, 1>{}, t0);
    step(std::integral_constant<int, 2>{}, t0);  step(std::integral_constant<int, 3>{}, t0);
    step(std::integral_constant<int, 4>{}, t0);  step(std::integral_constant<int, 5>{}, t0);
    step(std::integral_constant<int, 6>{}, t0);  step(std::integral_constant<int, 7>{}, t0);
    step(std::integral_constant<int, 8>{}, t0);  step(std::integral_constant<int, 9>{}, t0);
    step(std::integral_constant<int, 10>{}, t0); step(std::integral_constant<int, 11>{}, t0);
    step(std::integral_constant<int, 12>{}, t0); step(std::integral_constant<int, 13>{}, t0);
    step(std::integral_constant<int, 14>{}, t0); step(std::integral_constant<int, 15>{}, t0);
  }

  STAMP(3);
  // ---- columns beyond the 64 of the main grid: one candidate per lane, straightforward evaluation
  {
    const int nrest = (UW - 64) * UH;
    // hand the extra wave-iterations to the waves with the shortest bands first
    for (int c = tid; c < nrest; c += 256) {
      const int ay = c / (UW - 64), ax = 64 + (c - ay * (UW - 64));
      const int cmx = umin_x + ax, cmy = umin_y + ay;
      unsigned tie = (unsigned)spiral_pos(cmx - ucx, cmy - ucy) + 1;
      if (ff00 && cmx == 0 && cmy == 0) tie = 0;
      unsigned sad[16];
#pragma unroll
      for (int b = 0; b < 16; b++) sad[b] = tie;          // seeded like the main grid
      const uint32_t *wrow = swin + (ax & 3) * CS + ay * PITCH + (ax >> 2);
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const uint32_t *wp = wrow + r * PITCH;
        const int b = (r >> 2) * 4;
#pragma unroll
        for (int k = 0; k < 4; k++) sad[b + k] = __builtin_amdgcn_sad_hi_u8(wp[k], curs[r * 4 + k], sad[b + k]);
      }
      unsigned ps[JMHIP_NPART];
#pragma unroll
      for (int b8 = 0; b8 < 4; b8++) {
        const int o = 8 * (b8 >> 1) + 2 * (b8 & 1);
        const unsigned a = sad[o], b = sad[o + 1], cc = sad[o + 4], d = sad[o + 5];
        ps[25 + 4 * b8 + 0] = a; ps[25 + 4 * b8 + 1] = b; ps[25 + 4 * b8 + 2] = cc; ps[25 + 4 * b8 + 3] = d;
        ps[9 + 2 * b8 + 0] = a + b; ps[9 + 2 * b8 + 1] = cc + d;
        ps[17 + 2 * b8 + 0] = a + cc; ps[17 + 2 * b8 + 1] = b + d;
        ps[5 + b8] = a + b + cc + d;
      }
      ps[1] = ps[5] + ps[6]; ps[2] = ps[7] + ps[8];
      ps[3] = ps[5] + ps[7]; ps[4] = ps[6] + ps[8];
      // mv costs from the bits tables, four partitions per dword (as the main grid's refresh)
      unsigned mc0 = 0;
      {
        const uint32_t *bxr = reinterpret_cast<const uint32_t *>(S.bxtab[ax]), *byr = reinterpret_cast<const uint32_t *>(S.bytab[ay]);
#pragma unroll
        for (int g = 0; g < 11; g++) {
          const uint32_t sum4 = bxr[g] + byr[g];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int p = 4 * g + k;
            if (p < JMHIP_NPART) {
              const unsigned prod = __umul24((unsigned)lam, (sum4 >> (8 * k)) & 255u);
              if (p) best[p] = min(best[p], ps[p] + (prod & 0xffff0000u)); else mc0 = prod >> 16;
            }
          }
        }
      }
      unsigned c0 = ((ps[1] >> 16) + (ps[2] >> 16)) + mc0 + (unsigned)w16;
      if (quirk00 && 4 * (mbx * 16 + cmx) == mbx * 16 && 4 * (mby * 16 + cmy) == mby * 16) c0 -= (unsigned)w16;
      const unsigned long long k0 = ((unsigned long long)c0 << 32) | tie;
      best0 = k0 < best0 ? k0 : best0;
    }
  }

  STAMP(4);
  // ---- reduce over the workgroup through an LDS transpose (the window is dead by now): every lane stores its 41 keys
  //      as red[p][tid] (+ the low half of the 64-bit 16x16 key as row 41); then (partition, quarter) slots of 64 keys
  //      are reduced with 16-byte reads
  __syncthreads();                                   // all waves are done reading the window
  {
    uint32_t *red = swin;
#pragma unroll
    for (int p = 1; p < JMHIP_NPART; p++) red[p * 256 + tid] = best[p];
    red[0 * 256 + tid] = (unsigned)(best0 >> 32);
    red[JMHIP_NPART * 256 + tid] = (unsigned)best0;
  }
  __syncthreads();
  if (tid < 4 * JMHIP_NPART) {
    const int p = tid >> 2, qt = tid & 3;
    const uint4 *src = reinterpret_cast<const uint4 *>(swin + p * 256 + qt * 64);
    if (p) {
      unsigned m = KEY_INVALID;
#pragma unroll
      for (int k = 0; k < 16; k++) { const uint4 v = src[k]; m = min(min(m, v.x), min(min(v.y, v.z), v.w)); }
      S.part[p][qt] = m;
    } else {
      const uint4 *lo = reinterpret_cast<const uint4 *>(swin + JMHIP_NPART * 256 + qt * 64);
      unsigned long long m = ~0ull;
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const uint4 h = src[k], l = lo[k];
        m = min_u64(m, ((unsigned long long)h.x << 32) | l.x); m = min_u64(m, ((unsigned long long)h.y << 32) | l.y);
        m = min_u64(m, ((unsigned long long)h.z << 32) | l.z); m = min_u64(m, ((unsigned long long)h.w << 32) | l.w);
      }
      S.part0[qt] = m;
    }
  }
  STAMP(5);
  __syncthreads();
  if (tid < JMHIP_NPART) {
    const int p = tid;
    jmhip_me_result &o = res[mbi];
    int cost, tie;
    if (p == 0) {
      const unsigned long long k = min_u64(min_u64(S.part0[0], S.part0[1]), min_u64(S.part0[2], S.part0[3]));
      cost = (int)(unsigned)(k >> 32) - w16; tie = (int)(unsigned)k;
    } else {
      const unsigned k = min(min(S.part[p][0], S.part[p][1]), min(S.part[p][2], S.part[p][3]));
      const PartInfo q = c_part[p];
      cost = (int)(k >> FAST_TIE_BITS); tie = (int)(k & 0xffffu) / (q.w4 * q.h4);      // low half holds (4x4 blocks) * tie
    }
    int rx, ry;
    if (tie == 0) { rx = 0; ry = 0; }
    else { int ddx, ddy; spiral_offset(tie - 1, &ddx, &ddy); rx = ucx + ddx; ry = ucy + ddy; }
    if (p == 0) wrapped_bound_00(P, jobs[mbi], ucx, ucy, &rx, &ry, &cost);
    bool mine = true;
    if (P.mode == JMHIP_SEARCH_FULL) { int pcx, pcy; search_center(P, S.px[p], S.py[p], &pcx, &pcy); mine = (pcx == ucx && pcy == ucy); }
    if (mine) {
      o.mv_int[p][0] = (int16_t)rx; o.mv_int[p][1] = (int16_t)ry; o.cost_int[p] = cost;
      if (!P.subpel) { o.mv[p][0] = (int16_t)(rx << 2); o.mv[p][1] = (int16_t)(ry << 2); o.cost[p] = cost; }
    }
  }
  STAMP(6);
}

// ------------------------------------------------------------------------------------------------ pair-lane integer search
//
// Same algorithm as me_int_fast_kernel with the 16x16 block of a candidate split between TWO lanes (left / right 8 columns).
// Why: the one-lane form needs ~200 VGPRs (2 waves/SIMD) and 45 % of a workgroup's life is spent in latency phases (window
// staging, tables, reduction) that two resident waves per SIMD cannot hide. A half candidate needs a 16 x 8-byte rolling
// window (32 VGPRs), 21 + 1 running minima and cached mv costs: ~146 VGPRs, 3 waves/SIMD, 3 workgroups per CU.
// Each half owns the 19 partitions that lie inside it (8x16, two 8x8, four 8x4, four 4x8, eight 4x4); the three that span
// both halves (16x8 top/bottom, 16x16) take the partner's 8x8 sums through a DPP quad swap and are tracked by both lanes.
// Lane pair (2q, 2q+1) <-> candidate column; wave <-> (column group of 32, band of the candidate rows): two groups x two
// bands for 2R+1 >= 64, one group x four bands for 32 <= 2R+1 < 64; the remaining columns go pairwise through the same code.

constexpr int PAIR_NK = 22;                          // local partitions per half: 0..18 own, 19 = 16x8 top, 20 = 16x8 bottom, 21 = 16x16
__constant__ int8_t c_pair_g[2][24];                // global partition of (half, local index)
__constant__ int8_t c_pair_slot[JMHIP_NPART];       // partition -> local * 2 + half (spanning ones: half 0)
int8_t h_pair_g[2][24], h_pair_slot[JMHIP_NPART];

void build_pair_tables()
{
  build_part_table();
  auto find = [](int bt, int x4, int y4) { for (int p = 0; p < JMHIP_NPART; p++) if (h_part[p].bt == bt && h_part[p].x4 == x4 && h_part[p].y4 == y4) return p; return -1; };
  for (int h = 0; h < 2; h++) {
    int8_t *g = h_pair_g[h];
    for (int j = 0; j < 24; j++) g[j] = -1;
    for (int rg = 0; rg < 4; rg++) for (int c = 0; c < 2; c++) g[rg * 2 + c] = (int8_t)find(7, 2 * h + c, rg);         // 4x4 leaves
    for (int rg = 0; rg < 4; rg++) g[8 + rg] = (int8_t)find(5, 2 * h, rg);                                              // 8x4
    for (int k = 0; k < 2; k++) for (int c = 0; c < 2; c++) g[12 + k * 2 + c] = (int8_t)find(6, 2 * h + c, 2 * k);      // 4x8
    for (int k = 0; k < 2; k++) g[16 + k] = (int8_t)find(4, 2 * h, 2 * k);                                              // 8x8
    g[18] = (int8_t)find(3, 2 * h, 0);                                                                                  // 8x16
    g[19] = (int8_t)find(2, 0, 0); g[20] = (int8_t)find(2, 0, 2); g[21] = 0;                                            // spanning
  }
  for (int p = 0; p < JMHIP_NPART; p++) h_pair_slot[p] = -1;
  for (int h = 1; h >= 0; h--) for (int j = 0; j < PAIR_NK; j++) h_pair_slot[h_pair_g[h][j]] = (int8_t)(j * 2 + h);    // half 0 wins for the spanning ones
}

constexpr int SLOT_UNUSED = 0x7fffffff;
struct PairShared {                                   // 2R+1 <= 81 on this path (host check)
  int px[JMHIP_NPART], py[JMHIP_NPART];
  int spx[48], spy[48];                               // predictor of table slot half * 24 + local (SLOT_UNUSED: none)
  unsigned chg[96];
  uint8_t bytab[96][48] __attribute__((aligned(16)));    // [row][half * 24 + local]: vertical mv bits
  uint8_t bxtab[84][48] __attribute__((aligned(16)));    // [column][half * 24 + local]: horizontal mv bits
  uint32_t cur[64];
  unsigned part[44][2];
};

__device__ __forceinline__ unsigned quad_swap_add(unsigned v)
{
  return v + (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]: the partner lane
}

// LIST: the work items are a device-resident list whose LENGTH is device-resident too (*n_items_dev; the slice search's relaxation sweeps,
// me_xslice.hip, build it on the device and must not wait for the host): a fixed grid strides over it, one item per trip.
#ifndef JMHIP_PAIR_LIST_WAVES
#define JMHIP_PAIR_LIST_WAVES 2        /* list form: two waves per SIMD without spills measured 5 % faster than three with 55 spilled dwords */
#endif
template <bool LIST>
__global__ __launch_bounds__(256, LIST ? JMHIP_PAIR_LIST_WAVES : 3) void me_int_pair_kernel(MeDev P0, const jmhip_me_mb *__restrict__ jobs, const int *__restrict__ job_index,
                                                            jmhip_me_result *__restrict__ res, int n_items, const int *__restrict__ n_items_dev, int list_cap,
                                                            const MeDev *__restrict__ P_dev)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t swin[];      // 4 shifted window copies
  __shared__ PairShared S;
  if (LIST) n_items = jm_shard_slots(n_items_dev);                     // sharded list (jmhip_internal.h): virtual slots
  for (int vblock = blockIdx.x; vblock < jm_xcd_grid(n_items); vblock += gridDim.x) {
  if (LIST && vblock != (int)blockIdx.x) __syncthreads();              // the previous trip's last readers of S / swin
  // the thread index is re-read (opaquely) in every trip: otherwise everything derived from it is hoisted out of the loop and kept alive across
  // the whole body, which is already at the register budget of three waves per SIMD (93 dwords per lane went to scratch)
  int tid = threadIdx.x;
  if (LIST) asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = tid >> 6;
  // ... and the parameter block is re-read from memory in every trip (LIST): as a by-value argument its forty-odd scalars stay live across the loop
  // and the scalar file overflows into vector registers
  const MeDev *pp = P_dev;
  if (LIST) asm volatile("" : "+s"(pp));
  const MeDev &P = LIST ? *pp : P0;
  const int item = jm_xcd_item_of(vblock, n_items);
  if (item < 0) { if (LIST) continue; return; }
  const int word = LIST ? jm_shard_entry(n_items_dev, job_index, list_cap, item) : job_index[item];
  if (LIST && word < 0) continue;

  STAMP(0);
  const int mbi = word & 0xffffff, rep = (word >> 24) & 63;
  const int uni = __builtin_amdgcn_readfirstlane((word >> 30) & 1);      // one predictor for all 41 partitions
  const jmhip_me_mb &job = jobs[mbi];
  const int mbx = job.mb_x, mby = job.mb_y;
  const int R = P.R, UW = 2 * R + 1, UH = UW;
  const int PITCH = P.win_pitch >> 2, CS = P.win_copy_stride, WROWS = UH + 15;

  int ucx, ucy;
  search_center(P, job.pred_mv[rep][0], job.pred_mv[rep][1], &ucx, &ucy);
  const int umin_x = ucx - R, umin_y = ucy - R;
  if (tid < JMHIP_NPART) { S.px[tid] = job.pred_mv[tid][0]; S.py[tid] = job.pred_mv[tid][1]; }
  if (tid >= 64 && tid < 112) {                       // predictors in table-slot order, for the table build below
    const int s = tid - 64, g = c_pair_g[s / 24][s % 24];
    S.spx[s] = g >= 0 ? job.pred_mv[g][0] : SLOT_UNUSED;
    S.spy[s] = g >= 0 ? job.pred_mv[g][1] : SLOT_UNUSED;
  }
  __syncthreads();                                    // before the window loads are issued: every lane waits for the job anyway
  if (tid < 64) S.cur[tid] = *reinterpret_cast<const uint32_t *>(P.cur + (size_t)(mby * 16 + (tid >> 2)) * P.W + mbx * 16 + (tid & 3) * 4);

  // ---- reference window (as me_int_fast_kernel): loads first, tables while they fly, four copies from registers
  const uint8_t *ref = P.ref_y[job.ref];
  const int wpw = P.wp_w[job.ref], wpo = P.wp_o[job.ref];      // weighted reference ME (P.wp_on)
  const int bx = mbx * 16 + umin_x, by = mby * 16 + umin_y;
  const int xw = tid & 31, y0 = tid >> 5;
  const bool inside = bx >= 4 && bx + PITCH * 4 + 8 <= P.W;
  constexpr int NB = 12;
  uint32_t q0[NB], q1[NB], q2[NB];
  if (inside && xw < PITCH) {
#pragma unroll
    for (int u = 0; u < NB; u++) {
      const int y = min(y0 + 8 * u, WROWS - 1);
      const uintptr_t a = reinterpret_cast<uintptr_t>(ref + (size_t)clampi(by + y, 0, P.H - 1) * P.W + bx + xw * 4);
      const uint32_t *q = reinterpret_cast<const uint32_t *>(a & ~uintptr_t(3));
      q0[u] = q[0]; q1[u] = q[1]; q2[u] = q[2];
    }
  }
  // mv-bit tables, four slots (one dword) per item so that all 256 lanes work and a line of 48 slots is 12 stores:
  // 12 x (UH + UW) items; slot = half * 24 + local, unused slots hold 0
  for (int e = tid; e < 12 * (UH + UW); e += 256) {
    const bool isx = e >= 12 * UH;
    const int e2 = isx ? e - 12 * UH : e, line = e2 / 12, q = e2 - line * 12;
    const int v4 = 4 * ((isx ? umin_x : umin_y) + line);
    const int *sp = isx ? S.spx : S.spy;
    uint32_t w = 0;
    if (uni) {                                          // every used slot holds the same count
      const uint32_t b = (uint32_t)mvbits(v4 - sp[0]) * 0x01010101u;
#pragma unroll
      for (int k = 0; k < 4; k++) w |= sp[4 * q + k] == SLOT_UNUSED ? 0u : (b & (0xffu << (8 * k)));
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int p = sp[4 * q + k];
        w |= (p == SLOT_UNUSED ? 0u : (uint32_t)mvbits(v4 - p)) << (8 * k);
      }
    }
    reinterpret_cast<uint32_t *>(isx ? S.bxtab[line] : S.bytab[line])[q] = w;
  }
  if (xw < PITCH) {
    if (inside) {
      const unsigned sh = (unsigned)(bx & 3);
#pragma unroll
      for (int u = 0; u < NB; u++)
        if (y0 + 8 * u < WROWS) {
          uint32_t a = __builtin_amdgcn_alignbyte(q1[u], q0[u], sh), b = __builtin_amdgcn_alignbyte(q2[u], q1[u], sh);
          if (P.wp_on) { a = wp_apply4(a, wpw, wpo, P.wp_round, P.wp_denom); b = wp_apply4(b, wpw, wpo, P.wp_round, P.wp_denom); }
          uint32_t *w = swin + (y0 + 8 * u) * PITCH + xw;
          w[0] = a;
          w[1 * CS] = __builtin_amdgcn_alignbyte(b, a, 1u);
          w[2 * CS] = __builtin_amdgcn_alignbyte(b, a, 2u);
          w[3 * CS] = __builtin_amdgcn_alignbyte(b, a, 3u);
        }
    } else {
#pragma unroll
      for (int u = 0; u < NB; u++)
        if (y0 + 8 * u < WROWS) {
          const uint8_t *row = ref + (size_t)clampi(by + y0 + 8 * u, 0, P.H - 1) * P.W;
          uint32_t v = 0;
#pragma unroll
          for (int k = 0; k < 4; k++) v |= (uint32_t)row[clampi(bx + xw * 4 + k, 0, P.W - 1)] << (8 * k);
          if (P.wp_on) v = wp_apply4(v, wpw, wpo, P.wp_round, P.wp_denom);
          swin[(y0 + 8 * u) * PITCH + xw] = v;
        }
    }
  }
  __syncthreads();
  if (!inside && xw < PITCH - 1)
    for (int y = y0; y < WROWS; y += 8) {
      const uint32_t a = swin[y * PITCH + xw], b = swin[y * PITCH + xw + 1];
      swin[1 * CS + y * PITCH + xw] = __builtin_amdgcn_alignbyte(b, a, 1u);
      swin[2 * CS + y * PITCH + xw] = __builtin_amdgcn_alignbyte(b, a, 2u);
      swin[3 * CS + y * PITCH + xw] = __builtin_amdgcn_alignbyte(b, a, 3u);
    }
  if (tid < UH) {
    unsigned m = (tid == 0);
    if (tid) {
      const uint32_t *a = reinterpret_cast<const uint32_t *>(S.bytab[tid]), *b = reinterpret_cast<const uint32_t *>(S.bytab[tid - 1]);
#pragma unroll
      for (int g = 0; g < 12; g++) m |= (a[g] != b[g]);
    }
    S.chg[tid] = m;
  }
  __syncthreads();
  STAMP(1);

  // ---- per-lane constants
  const int half = lane & 1;
  const int NG = UW >= 64 ? 2 : 1;                   // column groups of 32 lane pairs (2R+1 >= 32 on this path)
  const int lam = P.lam_f;
  const int w16 = (lam * 16) >> 16;
  const int quirk00 = (P.mode == JMHIP_SEARCH_FULL) && !P.rdopt && !P.is_b && job.ref_is_0;
  const int ff00 = (P.mode == JMHIP_SEARCH_FASTFULL) && !P.rdopt;
  uint32_t cur[16][2];                               // this half's 8 columns of the current macroblock
#pragma unroll
  for (int r = 0; r < 16; r++) { cur[r][0] = S.cur[r * 4 + 2 * half]; cur[r][1] = S.cur[r * 4 + 2 * half + 1]; }

  // best[21]: the 16x16 key. Its cost can exceed 16 bits and goes negative through the 00-bonus, so it is kept as
  // (cost + bias) << 13 | tie: cost + bias < 2^17 (lambda is bounded on the host), tie < 2^13 (R <= 44).
  unsigned best[PAIR_NK];
#pragma unroll
  for (int j = 0; j < PAIR_NK; j++) best[j] = KEY_INVALID;
  unsigned mvc[PAIR_NK];                             // cached (mv cost << 16); [21] (16x16) unshifted + bias

  // one half candidate: sads of the 8 leaves (tie seeded), tree, partner exchange, 21 + 1 keys
  // mode 0: keep the keys in hold[] (even step of the row walk); 1: best = min3(best, hold, key) (odd step: one v_min3_u32 per two
  // candidates instead of two v_min_u32, both quarter-rate); 2: plain minimum (candidates outside the walk)
  unsigned hold[PAIR_NK];
#pragma unroll
  for (int j = 0; j < PAIR_NK; j++) hold[j] = KEY_INVALID;
  auto evaluate = [&](auto modec, const uint32_t (&w)[16][2], int base, unsigned tie, bool zero_bonus) __attribute__((always_inline)) {
    constexpr int mode = decltype(modec)::value;
    unsigned sad[8];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int sl = (base + r) & 15, b = (r >> 2) * 2;
      sad[b + 0] = __builtin_amdgcn_sad_hi_u8(w[sl][0], cur[r][0], (r & 3) ? sad[b + 0] : tie);
      sad[b + 1] = __builtin_amdgcn_sad_hi_u8(w[sl][1], cur[r][1], (r & 3) ? sad[b + 1] : tie);
    }
    unsigned ps[PAIR_NK - 1];
#pragma unroll
    for (int l = 0; l < 8; l++) ps[l] = sad[l];
#pragma unroll
    for (int rg = 0; rg < 4; rg++) ps[8 + rg] = sad[rg * 2] + sad[rg * 2 + 1];
#pragma unroll
    for (int k = 0; k < 2; k++) { ps[12 + 2 * k] = sad[4 * k] + sad[4 * k + 2]; ps[13 + 2 * k] = sad[4 * k + 1] + sad[4 * k + 3]; }
    ps[16] = ps[8] + ps[9]; ps[17] = ps[10] + ps[11];
    ps[18] = ps[16] + ps[17];
    ps[19] = quad_swap_add(ps[16]); ps[20] = quad_swap_add(ps[17]);
    unsigned c0 = ((ps[19] >> 16) + (ps[20] >> 16)) + mvc[21];
    if (zero_bonus) c0 -= (unsigned)w16;
    unsigned key[PAIR_NK];
#pragma unroll
    for (int j = 0; j < PAIR_NK - 1; j++) key[j] = ps[j] + mvc[j];
    key[21] = (c0 << TIE_BITS) + tie;
#pragma unroll
    for (int j = 0; j < PAIR_NK; j++) {
      if (mode == 0) hold[j] = key[j];
      else if (mode == 1) best[j] = min(min(best[j], hold[j]), key[j]);
      else best[j] = min(best[j], key[j]);
    }
  };
  auto load_mvc = [&](int colx, int row) __attribute__((always_inline)) {
    if (uni) {                                          // slot 0 of a half is a 4x4 leaf, always used
      const unsigned prod = __umul24((unsigned)lam, (unsigned)S.bytab[row][half * 24] + (unsigned)S.bxtab[colx][half * 24]);
      const unsigned hi = prod & 0xffff0000u;
#pragma unroll
      for (int j = 0; j < PAIR_NK - 1; j++) mvc[j] = hi;
      mvc[PAIR_NK - 1] = (prod >> 16) + (unsigned)w16;
      return;
    }
    const uint2 *bt = reinterpret_cast<const uint2 *>(&S.bytab[row][half * 24]), *bxq = reinterpret_cast<const uint2 *>(&S.bxtab[colx][half * 24]);
    const uint2 b0 = bt[0], b1 = bt[1], b2 = bt[2], x0 = bxq[0], x1 = bxq[1], x2 = bxq[2];
    const uint32_t byp[6] = {b0.x, b0.y, b1.x, b1.y, b2.x, b2.y}, bxp[6] = {x0.x, x0.y, x1.x, x1.y, x2.x, x2.y};
#pragma unroll
    for (int g = 0; g < 6; g++) {
      const uint32_t sum4 = bxp[g] + byp[g];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int j = 4 * g + k;
        if (j < PAIR_NK) {
          const unsigned prod = __umul24((unsigned)lam, (sum4 >> (8 * k)) & 255u);
          mvc[j] = j < PAIR_NK - 1 ? (prod & 0xffff0000u) : ((prod >> 16) + (unsigned)w16);
        }
      }
    }
  };

  STAMP(2);
  // ---- main grid: 64 columns (two groups of 32 lane pairs) x all rows (two halves)
  {
    const int col = (wave % NG) * 32 + (lane >> 1);
    const int mvx = umin_x + col, dx = mvx - ucx, adx = iabs(dx);
    const int tieB = spiral_base_B(dx) + 1, twodx = 2 * dx;
    const bool qx = quirk00 && (4 * (mbx * 16 + mvx) == mbx * 16);
    const uint32_t *lbase = swin + (col & 3) * CS + (col >> 2) + 2 * half;
    const int nparts = 4 / NG, part = wave / NG;      // row bands: 2 (two column groups) or 4 (one)
    // wave-uniform by construction; telling the compiler keeps the per-row arithmetic (spiral ring of the row, bonus test) scalar
    const int r0 = __builtin_amdgcn_readfirstlane((UH * part) / nparts), r1 = __builtin_amdgcn_readfirstlane((UH * (part + 1)) / nparts), nrows = r1 - r0;
    uint32_t win[16][2];
#pragma unroll
    for (int j = 0; j < 15; j++) { const uint32_t *wp = lbase + (r0 + j) * PITCH; win[j][0] = wp[0]; win[j][1] = wp[1]; }
    unsigned mnext = 0;
    auto step = [&](auto ttc, int t0) __attribute__((always_inline)) {
      constexpr int tt = decltype(ttc)::value;
      const int t = t0 + tt;
      if (t >= nrows) return;
      const int row = r0 + t;
      { const uint32_t *wp = lbase + (row + 15) * PITCH; constexpr int sl = (tt + 15) & 15; win[sl][0] = wp[0]; win[sl][1] = wp[1]; }
      const int mvy = umin_y + row, dy = mvy - ucy;
      {
        const unsigned m = (t == 0) ? 1u : mnext;
        mnext = S.chg[min(row + 1, UH - 1)];
        if (__builtin_amdgcn_readfirstlane(m)) load_mvc(col, row);
      }
      const int ady = iabs(dy);
      unsigned tie = (ady > adx) ? (unsigned)(spiral_base_A(dy) + 1 + twodx) : (unsigned)(tieB + 2 * dy);
      if (ff00 && mvx == 0 && mvy == 0) tie = 0;
      evaluate(std::integral_constant<int, tt & 1>{}, win, tt, tie, qx && 4 * (mby * 16 + mvy) == mby * 16);
    };
    for (int t0 = 0; t0 < nrows; t0 += 16) {
      step(std::integral_constant<int, 0>{}, t0);  step(std::integral_constant<int, 1>{}, t0);
      step(std::integral_constant<int, 2>{}, t0);  step(std::integral_constant<int, 3>{}, t0);
      step(std::integral_constant<int, 4>{}, t0);  step(std::integral_constant<int, 5>{}, t0);
      step(std::integral_constant<int, 6>{}, t0);  step(std::integral_constant<int, 7>{}, t0);
      step(std::integral_constant<int, 8>{}, t0);  step(std::integral_constant<int, 9>{}, t0);
      step(std::integral_constant<int, 10>{}, t0); step(std::integral_constant<int, 11>{}, t0);
      step(std::integral_constant<int, 12>{}, t0); step(std::integral_constant<int, 13>{}, t0);
      step(std::integral_constant<int, 14>{}, t0); step(std::integral_constant<int, 15>{}, t0);
    }
    if (nrows & 1) {                                  // the last row of an odd band is still on hold
#pragma unroll
      for (int j = 0; j < PAIR_NK; j++) best[j] = min(best[j], hold[j]);
    }
  }

  STAMP(3);
  // ---- columns beyond 64: one candidate per lane pair, fresh window rows, mv costs straight from the tables
  {
    const int nrc = UW - 32 * NG, nrest = nrc * UH;
    for (int c = tid >> 1; c < nrest; c += 128) {
      const int ay = c / nrc, ax = 32 * NG + (c - ay * nrc);
      const int cmx = umin_x + ax, cmy = umin_y + ay;
      unsigned tie = (unsigned)spiral_pos(cmx - ucx, cmy - ucy) + 1;
      if (ff00 && cmx == 0 && cmy == 0) tie = 0;
      const uint32_t *wrow = swin + (ax & 3) * CS + ay * PITCH + (ax >> 2) + 2 * half;
      uint32_t w[16][2];
#pragma unroll
      for (int r = 0; r < 16; r++) { w[r][0] = wrow[r * PITCH]; w[r][1] = wrow[r * PITCH + 1]; }
      load_mvc(ax, ay);
      evaluate(std::integral_constant<int, 2>{}, w, 0, tie, quirk00 && 4 * (mbx * 16 + cmx) == mbx * 16 && 4 * (mby * 16 + cmy) == mby * 16);
    }
  }

  STAMP(4);
  // ---- reduce: keys to LDS as red[slot = local * 2 + half][pair index], 64-bit key as two slot rows (42+half hi, 44+half lo)
  __syncthreads();
  {
    uint32_t *red = swin;
    const int q = tid >> 1;
#pragma unroll
    for (int j = 0; j < PAIR_NK; j++) red[(j * 2 + half) * 128 + q] = best[j];
  }
  __syncthreads();
  if (tid < 88) {                                     // 44 slots x 2 halves of 64 entries
    const int s = tid >> 1, hq = tid & 1;
    const uint4 *src = reinterpret_cast<const uint4 *>(swin + s * 128 + hq * 64);
    unsigned mm = KEY_INVALID;
#pragma unroll
    for (int k = 0; k < 16; k++) { const uint4 v = src[k]; mm = min(min(mm, v.x), min(min(v.y, v.z), v.w)); }
    S.part[s][hq] = mm;
  }
  __syncthreads();
  STAMP(5);
  if (tid < JMHIP_NPART) {
    const int p = tid, s = c_pair_slot[p];
    jmhip_me_result &o = res[mbi];
    int cost, tie;
    if (p == 0) {
      const unsigned k = min(S.part[s][0], S.part[s][1]);
      cost = (int)(k >> TIE_BITS) - w16; tie = (int)(k & ((1u << TIE_BITS) - 1));
    } else {
      const unsigned k = min(S.part[s][0], S.part[s][1]);
      const PartInfo q = c_part[p];
      cost = (int)(k >> FAST_TIE_BITS); tie = (int)(k & 0xffffu) / (q.w4 * q.h4);
    }
    int rx, ry;
    if (tie == 0) { rx = 0; ry = 0; }
    else { int ddx, ddy; spiral_offset(tie - 1, &ddx, &ddy); rx = ucx + ddx; ry = ucy + ddy; }
    if (p == 0) wrapped_bound_00(P, jobs[mbi], ucx, ucy, &rx, &ry, &cost);
    bool mine = true;
    if (P.mode == JMHIP_SEARCH_FULL) { int pcx, pcy; search_center(P, S.px[p], S.py[p], &pcx, &pcy); mine = (pcx == ucx && pcy == ucy); }
    if (mine) {
      o.mv_int[p][0] = (int16_t)rx; o.mv_int[p][1] = (int16_t)ry; o.cost_int[p] = cost;
      if (!P.subpel) { o.mv[p][0] = (int16_t)(rx << 2); o.mv[p][1] = (int16_t)(ry << 2); o.cost[p] = cost; }
    }
  }
  STAMP(6);
  if (!LIST) break;
  }
}

// ------------------------------------------------------------------------------------------------ persistent pair-lane search
//
// EXPERIMENT, not the default (JMHIP_ME_KERNEL=pers): measured 0.303 ms against the pair kernel's 0.267 ms at 1080p -- see DESIGN.md section 3.
// me_int_pair_kernel's walk (the same lane pairs, keys and tables) inside a PERSISTENT workgroup: three per CU, each taking every
// (grid/8)-th item of its XCD's contiguous run. What changes is everything around the walk -- the part that kept the VALU idle
// (setup 25 %, reduction 6 %, decode 5 % of a workgroup's life, all latency):
//  * the NEXT item is staged while the current one is walked: its job record, current macroblock and reference window go from HBM/L2
//    straight into a second LDS buffer with global_load_lds_dword (no VGPRs, no wait until the barrier that ends the item); its mv-bit
//    tables are built into a second table set from the job record that arrived one item earlier;
//  * the window is read from plane 0 of the reference's quarter-pel stack (the integer picture with its replicated ring): a dword that
//    lies outside the picture is a dword of the ring -- four equal bytes -- so clamping the DWORD address is exact and every load is an
//    aligned dword; one raw copy in LDS, v_alignbyte in the walk (two more VALU per 105-instruction step);
//  * the 22 keys of a wave are folded with three DPP row shifts each instead of an LDS transpose, and the last 16-way minimum, the
//    decode and the store of item k run at the top of item k+1's iteration: ONE barrier per item instead of six.
// Exit condition: a static count of items per workgroup (no queue, no flag): every wave leaves after its last barrier.

typedef __attribute__((address_space(3))) void jm_lds_void;
typedef const __attribute__((address_space(1))) void jm_glb_void;
__device__ __forceinline__ void dma_dword(const void *src, void *lds_wave_base)     // LDS address = wave-uniform base + 4 * lane
{
  __builtin_amdgcn_global_load_lds((jm_glb_void *)src, (jm_lds_void *)lds_wave_base, 4, 0, 0);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

struct PersShared {                                   // 2R+1 <= 81 on this path (host check)
  uint32_t job[4][64];                                // jmhip_me_mb records (43 dwords) of items k-1 .. k+2
  int idx[4];                                         // their job_index words
  uint32_t cur[2][64];
  uint32_t gslot[12];                                 // c_pair_g as bytes, table-slot order (half * 24 + local)
  uint8_t bytab[2][84][48] __attribute__((aligned(16)));
  uint8_t bxtab[2][84][48] __attribute__((aligned(16)));
  unsigned red[2][4][4][2][PAIR_NK];                  // [buffer][wave][row of 16 lanes][half][local key]
};

template <unsigned CTRL> __device__ __forceinline__ unsigned dpp_min(unsigned v)
{
  return min(v, (unsigned)__builtin_amdgcn_update_dpp((int)KEY_INVALID, (int)v, CTRL, 0xf, 0xf, false));
}

__global__ __launch_bounds__(256, 3) void me_int_pers_kernel(MeDev P, const jmhip_me_mb *__restrict__ jobs, const int *__restrict__ job_index,
                                                            jmhip_me_result *__restrict__ res, int n_items)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t swin[];      // two raw windows, NWP dwords apart
  __shared__ PersShared S;
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6), half = lane & 1;
  const int R = P.R, UW = 2 * R + 1, UH = UW, PITCH = P.win_pitch, WROWS = UH + 15, NW = WROWS * PITCH, NWP = P.win_copy_stride;
  const unsigned pitch_inv = ((1u << 20) + PITCH - 1) / PITCH;          // e / PITCH == (e * pitch_inv) >> 20 for e < 4096, PITCH <= 25

  // static schedule: XCD x takes the contiguous run [x * per, (x + 1) * per) (as jm_xcd_item); workgroup `slot` of the XCD every nslots-th item of it
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int per = (n_items + 7) >> 3;
  const int run = min(per, n_items - xcd * per);
  const int niter = run > slot ? (run - slot + nslots - 1) / nslots : 0;
  if (niter == 0) return;
  const int item0 = xcd * per + slot;

  const int lam = P.lam_f;
  const int w16 = (lam * 16) >> 16;
  const int ff00 = (P.mode == JMHIP_SEARCH_FASTFULL) && !P.rdopt;
  const int NG = UW >= 64 ? 2 : 1;
  const int my_slot = tid < JMHIP_NPART ? c_pair_slot[tid] : 0;
  const int my_area = tid < JMHIP_NPART ? c_part[tid].w4 * c_part[tid].h4 : 1;
  if (tid < 12) {
    uint32_t w = 0;
    for (int k = 0; k < 4; k++) { const int sl = 4 * tid + k; w |= (uint32_t)(uint8_t)c_pair_g[sl / 24][sl % 24] << (8 * k); }
    S.gslot[tid] = w;
  }

  auto dma_job = [&](int k, int idx) {                 // wave 0
    if (lane < 43) dma_dword(reinterpret_cast<const uint32_t *>(jobs + (idx & 0xffffff)) + lane, &S.job[k & 3][0]);
    if (lane == 0) S.idx[k & 3] = idx;
  };
  // header of an item whose job record is in LDS: position, centre, window origin
  struct Head { int mbx, mby, ref, ref_is_0, rep, mbi, ucx, ucy; };
  auto head_of = [&](int k) {
    const uint32_t *jb = S.job[k & 3];
    const int w0 = uni((int)jb[0]), w1 = uni((int)jb[1]), idx = uni(S.idx[k & 3]);
    Head h;
    h.mbx = (short)(w0 & 0xffff); h.mby = w0 >> 16; h.ref = (short)(w1 & 0xffff); h.ref_is_0 = w1 >> 16;
    h.mbi = idx & 0xffffff; h.rep = (idx >> 24) & 63;
    const int pm = uni((int)jb[2 + h.rep]);
    search_center(P, (short)(pm & 0xffff), pm >> 16, &h.ucx, &h.ucy);
    return h;
  };

#ifdef JMHIP_STAMPS
  unsigned long long t_prev = __builtin_amdgcn_s_memtime(), t_acc[6] = {0, 0, 0, 0, 0, 0};
#define PSTAMP(q) do { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); t_acc[q] += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define PSTAMP(q) do { } while (0)
#endif
  int idx_pipe = job_index[item0];                    // job_index word of item k + 2 at the top of iteration k
  if (wave == 0) dma_job(0, idx_pipe);
  idx_pipe = niter > 1 ? job_index[item0 + nslots] : 0;
  __syncthreads();

  for (int k = -1; k <= niter; k++) {
    // ================================================================ A: stage item k + 1 (job record arrived during iteration k - 1)
    if (k + 1 < niter) {
      const int k1 = k + 1;
      if (wave == 0 && k + 2 < niter) dma_job(k + 2, idx_pipe);
      if (k + 3 < niter) idx_pipe = job_index[item0 + (k + 3) * nslots];
      const Head h = head_of(k1);
      const int umin_x = h.ucx - R, umin_y = h.ucy - R;
      const int bx = h.mbx * 16 + umin_x, by = h.mby * 16 + umin_y, bxa = bx & ~3;
      const uint8_t *plane = P.ref_sub[h.ref];          // plane 0 of the quarter-pel stack: integer samples, JMHIP_PAD-pel ring, stride Wp
      uint32_t *wdst = swin + (k1 & 1) * NWP;
      for (int e0 = 0; e0 < NW; e0 += 256) {
        const int e = e0 + tid;
        if (e < NW) {
          const int row = (int)(((unsigned)e * pitch_inv) >> 20), xw = e - row * PITCH;
          const int y = clampi(by + row, 0, P.H - 1) + JMHIP_PAD, x = clampi(bxa + 4 * xw, -4, P.W) + JMHIP_PAD;
          dma_dword(plane + (size_t)y * P.Wp + x, wdst + e0 + wave * 64);
        }
      }
      if (wave == 1) dma_dword(P.cur + (size_t)(h.mby * 16 + (lane >> 2)) * P.W + h.mbx * 16 + (lane & 3) * 4, &S.cur[k1 & 1][0]);
      // mv-bit tables, four slots (one dword) per element: 12 x (UH + UW) elements; slot = half * 24 + local, unused slots hold 0
      const uint32_t *jb = S.job[k1 & 3];
      for (int e = tid; e < 12 * (UH + UW); e += 256) {
        const bool isx = e >= 12 * UH;
        const int e2 = isx ? e - 12 * UH : e, line = e2 / 12, q = e2 - line * 12;
        const int v4 = 4 * ((isx ? umin_x : umin_y) + line);
        const uint32_t g4 = S.gslot[q];
        uint32_t w = 0;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          const int g = (int)(int8_t)(g4 >> (8 * kk));
          const int pm = (int)jb[2 + max(g, 0)];
          const int pv = isx ? (int)(short)(pm & 0xffff) : (pm >> 16);
          w |= (g < 0 ? 0u : (uint32_t)mvbits(v4 - pv)) << (8 * kk);
        }
        reinterpret_cast<uint32_t *>(isx ? S.bxtab[k1 & 1][line] : S.bytab[k1 & 1][line])[q] = w;
      }
    }

    PSTAMP(0);
    // ================================================================ B: finish item k - 1 (16-way minimum of the waves' row results, decode, store)
    if (k >= 1 && tid < JMHIP_NPART) {
      const int kp = k - 1, p = tid, local = my_slot >> 1, hf = my_slot & 1;
      const Head h = head_of(kp);
      unsigned key = KEY_INVALID;
#pragma unroll
      for (int w = 0; w < 4; w++)
#pragma unroll
        for (int r = 0; r < 4; r++) key = min(key, S.red[kp & 1][w][r][hf][local]);
      jmhip_me_result &o = res[h.mbi];
      int cost, tie;
      if (p == 0) { cost = (int)(key >> TIE_BITS) - w16; tie = (int)(key & ((1u << TIE_BITS) - 1)); }
      else { cost = (int)(key >> FAST_TIE_BITS); tie = (int)(key & 0xffffu) / my_area; }
      int rx, ry;
      if (tie == 0) { rx = 0; ry = 0; }
      else { int ddx, ddy; spiral_offset(tie - 1, &ddx, &ddy); rx = h.ucx + ddx; ry = h.ucy + ddy; }
      if (p == 0) { const int pm0 = (int)S.job[kp & 3][2]; wrapped_bound_00v(P, h.mbx, h.mby, h.ref, h.ref_is_0, (short)(pm0 & 0xffff), pm0 >> 16, h.ucx, h.ucy, &rx, &ry, &cost); }
      bool mine = true;
      if (P.mode == JMHIP_SEARCH_FULL) {
        const int pm = (int)S.job[kp & 3][2 + p];
        int pcx, pcy; search_center(P, (short)(pm & 0xffff), pm >> 16, &pcx, &pcy); mine = (pcx == h.ucx && pcy == h.ucy);
      }
      if (mine) {
        o.mv_int[p][0] = (int16_t)rx; o.mv_int[p][1] = (int16_t)ry; o.cost_int[p] = cost;
        if (!P.subpel) { o.mv[p][0] = (int16_t)(rx << 2); o.mv[p][1] = (int16_t)(ry << 2); o.cost[p] = cost; }
      }
    }

    PSTAMP(1);
    // ================================================================ C: walk item k
    if (k >= 0 && k < niter) {
      const Head h = head_of(k);
      const int mbx = h.mbx, mby = h.mby, ucx = h.ucx, ucy = h.ucy;
      const int umin_x = ucx - R, umin_y = ucy - R;
      const int xoff = (mbx * 16 + umin_x) & 3;
      const int quirk00 = (P.mode == JMHIP_SEARCH_FULL) && !P.rdopt && !P.is_b && h.ref_is_0;
      const uint32_t *W = swin + (k & 1) * NWP;
      const uint8_t (*bytab)[48] = S.bytab[k & 1];
      const uint8_t (*bxtab)[48] = S.bxtab[k & 1];
      uint32_t cur[16][2];
#pragma unroll
      for (int r = 0; r < 16; r++) { cur[r][0] = S.cur[k & 1][r * 4 + 2 * half]; cur[r][1] = S.cur[k & 1][r * 4 + 2 * half + 1]; }

      unsigned best[PAIR_NK], hold[PAIR_NK], mvc[PAIR_NK];
#pragma unroll
      for (int j = 0; j < PAIR_NK; j++) { best[j] = KEY_INVALID; hold[j] = KEY_INVALID; }
      auto evaluate = [&](auto modec, const uint32_t (&w)[16][2], int base, unsigned tie, bool zero_bonus) __attribute__((always_inline)) {
        constexpr int mode = decltype(modec)::value;
        unsigned sad[8];
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int sl = (base + r) & 15, b = (r >> 2) * 2;
          sad[b + 0] = __builtin_amdgcn_sad_hi_u8(w[sl][0], cur[r][0], (r & 3) ? sad[b + 0] : tie);
          sad[b + 1] = __builtin_amdgcn_sad_hi_u8(w[sl][1], cur[r][1], (r & 3) ? sad[b + 1] : tie);
        }
        unsigned ps[PAIR_NK - 1];
#pragma unroll
        for (int l = 0; l < 8; l++) ps[l] = sad[l];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) ps[8 + rg] = sad[rg * 2] + sad[rg * 2 + 1];
#pragma unroll
        for (int kk = 0; kk < 2; kk++) { ps[12 + 2 * kk] = sad[4 * kk] + sad[4 * kk + 2]; ps[13 + 2 * kk] = sad[4 * kk + 1] + sad[4 * kk + 3]; }
        ps[16] = ps[8] + ps[9]; ps[17] = ps[10] + ps[11];
        ps[18] = ps[16] + ps[17];
        ps[19] = quad_swap_add(ps[16]); ps[20] = quad_swap_add(ps[17]);
        unsigned c0 = ((ps[19] >> 16) + (ps[20] >> 16)) + mvc[21];
        if (zero_bonus) c0 -= (unsigned)w16;
        unsigned key[PAIR_NK];
#pragma unroll
        for (int j = 0; j < PAIR_NK - 1; j++) key[j] = ps[j] + mvc[j];
        key[21] = (c0 << TIE_BITS) + tie;
#pragma unroll
        for (int j = 0; j < PAIR_NK; j++) {
          if (mode == 0) hold[j] = key[j];
          else if (mode == 1) best[j] = min(min(best[j], hold[j]), key[j]);
          else best[j] = min(best[j], key[j]);
        }
      };
      auto load_mvc = [&](int colx, int row) __attribute__((always_inline)) {
        const uint2 *bt = reinterpret_cast<const uint2 *>(&bytab[row][half * 24]), *bxq = reinterpret_cast<const uint2 *>(&bxtab[colx][half * 24]);
        const uint2 b0 = bt[0], b1 = bt[1], b2 = bt[2], x0 = bxq[0], x1 = bxq[1], x2 = bxq[2];
        const uint32_t byp[6] = {b0.x, b0.y, b1.x, b1.y, b2.x, b2.y}, bxp[6] = {x0.x, x0.y, x1.x, x1.y, x2.x, x2.y};
#pragma unroll
        for (int g = 0; g < 6; g++) {
          const uint32_t sum4 = bxp[g] + byp[g];
#pragma unroll
          for (int kk = 0; kk < 4; kk++) {
            const int j = 4 * g + kk;
            if (j < PAIR_NK) {
              const unsigned prod = __umul24((unsigned)lam, (sum4 >> (8 * kk)) & 255u);
              mvc[j] = j < PAIR_NK - 1 ? (prod & 0xffff0000u) : ((prod >> 16) + (unsigned)w16);
            }
          }
        }
      };

      // ---- main grid: 64 columns (two groups of 32 lane pairs) x all rows (two bands), or 32 columns x four bands
      {
        const int col = (wave % NG) * 32 + (lane >> 1);
        const int mvx = umin_x + col, dx = mvx - ucx, adx = iabs(dx);
        const int tieB = spiral_base_B(dx) + 1, twodx = 2 * dx;
        const bool qx = quirk00 && (4 * (mbx * 16 + mvx) == mbx * 16);
        const int bcol = xoff + col + 8 * half;          // byte column of this half's first sample in the raw window
        const unsigned sh = (unsigned)(bcol & 3);
        const uint32_t *lbase = W + (bcol >> 2);
        const int nparts = 4 / NG, part = wave / NG;
        const int r0 = uni((UH * part) / nparts), r1 = uni((UH * (part + 1)) / nparts), nrows = r1 - r0;
        // rows of the band whose vertical mv bits differ from the row before (any slot): those steps refresh the cached mv costs
        unsigned long long chg;
        {
          bool f = false;
          if (lane > 0 && lane < nrows) {
            const uint4 *a = reinterpret_cast<const uint4 *>(bytab[r0 + lane]), *b = reinterpret_cast<const uint4 *>(bytab[r0 + lane - 1]);
#pragma unroll
            for (int g = 0; g < 3; g++) { const uint4 u = a[g], v = b[g]; f = f || u.x != v.x || u.y != v.y || u.z != v.z || u.w != v.w; }
          }
          chg = __ballot(f) | 1ull;
        }
        uint32_t win[16][2];
#pragma unroll
        for (int j = 0; j < 15; j++) {
          const uint32_t *wp = lbase + (r0 + j) * PITCH;
          const uint32_t d0 = wp[0], d1 = wp[1], d2 = wp[2];
          win[j][0] = __builtin_amdgcn_alignbyte(d1, d0, sh); win[j][1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
        }
        auto step = [&](auto ttc, int t0) __attribute__((always_inline)) {
          constexpr int tt = decltype(ttc)::value;
          const int t = t0 + tt;
          if (t >= nrows) return;
          const int row = r0 + t;
          {
            const uint32_t *wp = lbase + (row + 15) * PITCH;
            constexpr int sl = (tt + 15) & 15;
            const uint32_t d0 = wp[0], d1 = wp[1], d2 = wp[2];
            win[sl][0] = __builtin_amdgcn_alignbyte(d1, d0, sh); win[sl][1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
          }
          const int mvy = umin_y + row, dy = mvy - ucy;
          if ((chg >> t) & 1ull) load_mvc(col, row);
          const int ady = iabs(dy);
          unsigned tie = (ady > adx) ? (unsigned)(spiral_base_A(dy) + 1 + twodx) : (unsigned)(tieB + 2 * dy);
          if (ff00 && mvx == 0 && mvy == 0) tie = 0;
          evaluate(std::integral_constant<int, tt & 1>{}, win, tt, tie, qx && 4 * (mby * 16 + mvy) == mby * 16);
        };
        for (int t0 = 0; t0 < nrows; t0 += 16) {
          step(std::integral_constant<int, 0>{}, t0);  step(std::integral_constant<int, 1>{}, t0);
          step(std::integral_constant<int, 2>{}, t0);  step(std::integral_constant<int, 3>{}, t0);
          step(std::integral_constant<int, 4>{}, t0);  step(std::integral_constant<int, 5>{}, t0);
          step(std::integral_constant<int, 6>{}, t0);  step(std::integral_constant<int, 7>{}, t0);
          step(std::integral_constant<int, 8>{}, t0);  step(std::integral_constant<int, 9>{}, t0);
          step(std::integral_constant<int, 10>{}, t0); step(std::integral_constant<int, 11>{}, t0);
          step(std::integral_constant<int, 12>{}, t0); step(std::integral_constant<int, 13>{}, t0);
          step(std::integral_constant<int, 14>{}, t0); step(std::integral_constant<int, 15>{}, t0);
        }
        if (nrows & 1) {
#pragma unroll
          for (int j = 0; j < PAIR_NK; j++) best[j] = min(best[j], hold[j]);
        }
      }

      PSTAMP(2);
      // ---- columns beyond the main grid: one candidate per lane pair, fresh window rows
      {
        const int nrc = UW - 32 * NG, nrest = nrc * UH;
        for (int c = tid >> 1; c < nrest; c += 128) {
          const int ay = c / nrc, ax = 32 * NG + (c - ay * nrc);
          const int cmx = umin_x + ax, cmy = umin_y + ay;
          unsigned tie = (unsigned)spiral_pos(cmx - ucx, cmy - ucy) + 1;
          if (ff00 && cmx == 0 && cmy == 0) tie = 0;
          const int bcol = xoff + ax + 8 * half;
          const unsigned sh = (unsigned)(bcol & 3);
          const uint32_t *wrow = W + ay * PITCH + (bcol >> 2);
          uint32_t w[16][2];
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const uint32_t d0 = wrow[r * PITCH], d1 = wrow[r * PITCH + 1], d2 = wrow[r * PITCH + 2];
            w[r][0] = __builtin_amdgcn_alignbyte(d1, d0, sh); w[r][1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
          }
          load_mvc(ax, ay);
          evaluate(std::integral_constant<int, 2>{}, w, 0, tie, quirk00 && 4 * (mbx * 16 + cmx) == mbx * 16 && 4 * (mby * 16 + cmy) == mby * 16);
        }
      }

      PSTAMP(3);
      // ---- fold the wave's keys over the 32 lanes of each half: three row shifts leave a row's minima in its lanes 14 (left) / 15 (right)
      {
        unsigned *rd = &S.red[k & 1][wave][lane >> 4][half][0];
        const bool writer = (lane & 15) >= 14;
#pragma unroll
        for (int j = 0; j < PAIR_NK; j++) {
          unsigned v = best[j];
          v = dpp_min<0x112>(v); v = dpp_min<0x114>(v); v = dpp_min<0x118>(v);
          if (writer) rd[j] = v;
        }
      }
    }
    PSTAMP(4);
    __syncthreads();                                  // (the compiler waits for the staged loads, vmcnt(0), before it)
    PSTAMP(5);
  }
#ifdef JMHIP_STAMPS
  if (P.stamps && blockIdx.x < 512 && lane == 0) {
    unsigned long long cum = 0;
    P.stamps[(blockIdx.x * 4 + wave) * 8] = 0;
    for (int q = 0; q < 6; q++) { cum += t_acc[q]; P.stamps[(blockIdx.x * 4 + wave) * 8 + q + 1] = cum; }
  }
#endif
}

int ensure_tables(jmhip_ctx *c)
{
  static bool uploaded[64] = {false};
  const int dev = c->cfg.device;
  if (dev < 64 && uploaded[dev]) return JMHIP_OK;
  build_part_table();
  JM_HIP_CHECK(c, hipMemcpyToSymbol(HIP_SYMBOL(c_part), h_part, sizeof(h_part)));
  build_pair_tables();
  JM_HIP_CHECK(c, hipMemcpyToSymbol(HIP_SYMBOL(c_pair_g), h_pair_g, sizeof(h_pair_g)));
  JM_HIP_CHECK(c, hipMemcpyToSymbol(HIP_SYMBOL(c_pair_slot), h_pair_slot, sizeof(h_pair_slot)));
  if (dev < 64) uploaded[dev] = true;
  return JMHIP_OK;
}

}  // namespace

// device table of reference plane pointers: [0..31] integer recon, [32..63] luma quarter-pel stacks,
// [64..95] Cb eighth-pel stacks, [96..127] Cr
int jm_ensure_ref_table(jmhip_ctx *c)
{
  if (c->ref_ptrs_dev) return jm_flush_table_fix(c);
  std::vector<const uint8_t *> tab(128, nullptr);
  for (size_t i = 0; i < c->refs.size(); i++) {
    tab[i] = c->refs[i].y; tab[32 + i] = c->refs[i].luma_sub; tab[64 + i] = c->refs[i].cr_sub[0]; tab[96 + i] = c->refs[i].cr_sub[1];
  }
  if (hipMalloc(&c->ref_ptrs_dev, sizeof(void *) * 128) != hipSuccess) return jm_fail(c, JMHIP_ERR_NOMEM, "reference pointer table");
  JM_HIP_CHECK(c, hipMemcpy(c->ref_ptrs_dev, tab.data(), sizeof(void *) * 128, hipMemcpyHostToDevice));
  return JMHIP_OK;
}

extern "C" void jmhip_partition_info(int p, int *blocktype, int *x4, int *y4, int *w4, int *h4)
{
  build_part_table();
  if (p < 0 || p >= JMHIP_NPART) return;
  if (blocktype) *blocktype = h_part[p].bt;
  if (x4) *x4 = h_part[p].x4;
  if (y4) *y4 = h_part[p].y4;
  if (w4) *w4 = h_part[p].w4;
  if (h4) *h4 = h_part[p].h4;
}

// The pair-lane search over a device-resident item list (me_xslice.hip): window geometry of the fast kernels for range R, and the launch.
// Items: job index | representative partition << 24 | (all 41 predictors equal) << 30, as jmhip_me_frame builds them on the host.
int jm_me_pair_geometry(jmhip_ctx *c, int R, MeDev *P, size_t *lds)
{
  if (2 * R + 1 < 32 || 2 * R + 1 > 81 || (2 * R + 1) * (2 * R + 1) + 1 >= (1 << TIE_BITS)) return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "pair-lane search: 16 <= search_range <= 40");
  const int fpitch_dw = ((2 * R + 1 + 15 + 3) >> 2) + 2, frows = 2 * R + 1 + 15;
  int fcs = fpitch_dw * frows;
  while ((fcs & 31) != 8) fcs++;
  size_t plds = ((size_t)3 * fcs + (size_t)fpitch_dw * frows) * 4;
  if (plds < (size_t)44 * 128 * 4) plds = (size_t)44 * 128 * 4;
  P->win_pitch = fpitch_dw * 4; P->win_rows = frows; P->win_copy_stride = fcs;
  *lds = plds;
  return ensure_tables(c);
}
void jm_launch_me_pair_list(jmhip_ctx *c, const MeDev &P, const MeDev *P_dev, size_t lds, const jmhip_me_mb *jobs_dev, const int *idx_dev, jmhip_me_result *res_dev, const int *cnt_dev, int cap, int grid)
{
  me_int_pair_kernel<true><<<grid, 256, lds, c->stream>>>(P, jobs_dev, idx_dev, res_dev, 0, cnt_dev, cap, P_dev);
}

// host mirror of search_center (mv-search.c:752-762) to size the LDS window
constexpr int FAST_MAX_CENTRES = JMHIP_NPART;  // one walk per distinct centre always: a single macroblock in the union-window kernel is a 0.5 ms serial chain

// JMHIP_ME_KERNEL=single selects the one-lane-per-candidate kernel (2R+1 >= 64 only); default: the pair-lane kernel (2R+1 >= 32)
static int me_use_pair_kernel()
{
  const char *e = getenv("JMHIP_ME_KERNEL");         // read per call: tests switch it
  return e && !strcmp(e, "single") ? 0 : 1;
}
// JMHIP_ME_KERNEL=pers selects the persistent, double-buffered form of the pair-lane kernel (me_int_pers_kernel): bit-exact, measured
// 13 % SLOWER than the default at 1080p (0.303 vs 0.267 ms; DESIGN.md section 3 has the counters that say why) -- kept as the measured experiment
static int me_use_pers_kernel()
{
  const char *e = getenv("JMHIP_ME_KERNEL");
  return e && !strcmp(e, "pers") ? 1 : 0;
}

static void host_center(const jmhip_me_params *prm, int pmx, int pmy, int *cx, int *cy)
{
  const int R = prm->search_range;
  auto clip = [](int lo, int hi, int x) { return x < lo ? lo : (x > hi ? hi : x); };
  int mx = pmx / 4, my = pmy / 4;
  if (!prm->rdopt) { mx = clip(-R, R, mx); my = clip(-R, R, my); }
  mx = clip(-2047 + R, 2047 - R, mx);
  my = clip(prm->level_mv_min + R, prm->level_mv_max - R, my);
  *cx = mx; *cy = my;
}

// job / result / index arrays of the search stage for n macroblocks (also filled by jmhip_slice_to_frame, me_wave.hip)
int jm_me_arrays_ensure(jmhip_ctx *c, int n)
{
  if (c->me_capacity >= n) return JMHIP_OK;
  if (c->me_jobs_dev) JM_HIP_CHECK(c, hipFree(c->me_jobs_dev));
  if (c->me_res_dev) JM_HIP_CHECK(c, hipFree(c->me_res_dev));
  if (c->me_idx_dev) JM_HIP_CHECK(c, hipFree(c->me_idx_dev));
  c->me_jobs_dev = c->me_res_dev = c->me_idx_dev = nullptr; c->me_capacity = 0; c->me_n = 0;
  if (hipMalloc(&c->me_jobs_dev, sizeof(jmhip_me_mb) * (size_t)n) != hipSuccess) return jm_fail(c, JMHIP_ERR_NOMEM, "ME job array");
  if (hipMalloc(&c->me_res_dev, sizeof(jmhip_me_result) * (size_t)n) != hipSuccess) return jm_fail(c, JMHIP_ERR_NOMEM, "ME result array");
  if (hipMalloc(&c->me_idx_dev, sizeof(int) * (size_t)n * FAST_MAX_CENTRES) != hipSuccess) return jm_fail(c, JMHIP_ERR_NOMEM, "ME index array");
  c->me_capacity = n;
  return JMHIP_OK;
}

extern "C" int jmhip_me_frame_async(jmhip_ctx *c, const jmhip_me_params *prm, const jmhip_me_mb *mbs, int n)
{
  if (!c || !prm || n <= 0) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_me_frame: NULL/empty arguments") : JMHIP_ERR_ARG;
  const bool resident = (mbs == nullptr);            // reuse the job array uploaded (and validated) by the previous call
  if (resident && (n != c->me_n || !c->me_jobs_dev)) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_me_frame: mbs == NULL needs a previous call with the same n");
  if (prm->search_mode != JMHIP_SEARCH_FULL && prm->search_mode != JMHIP_SEARCH_FASTFULL)
    return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_me_frame: search_mode must be -1 (FullSearch) or 0 (FastFullSearch); EPZS/UMHex run on the host over jmhip_distortion_batch");
  const int R = prm->search_range;
  if (R < 0 || R > c->cfg.search_range) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_me_frame: search_range exceeds the context's");
  // search_range > 44: the spiral index does not fit the 13-bit tie field of the fast kernels' packed 32-bit keys; the general kernel
  // (me_metric.hip, 64-bit keys) searches instead
  const bool wide = (2 * R + 1) * (2 * R + 1) + 1 >= (1 << TIE_BITS);
  jmhip_me_params eff = *prm;
  if (!eff.metric_set) { eff.metric[0] = 0; eff.metric[1] = eff.metric[2] = 2; eff.chroma_me = 0; eff.metric_set = 1; }
  for (int k = 0; k < 3; k++)
    if (prm->lambda[k] < 0 || prm->lambda[k] >= (1 << 24)) return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_me_frame: lambda factor must be below 2^24 (the kernels multiply it with __umul24; JM's largest, QP 51, is 5.5e6)");
  if (!c->has_cur) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_me_frame: current picture not uploaded");
  if (!(prm->partition_mask & ((1ull << JMHIP_NPART) - 1))) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_me_frame: empty partition mask");
  if (prm->wp_enable && (prm->wp_denom < 0 || prm->wp_denom > 7 || prm->wp_round < 0 || prm->wp_round > 64)) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_me_frame: weighted-prediction denominator / rounding out of range");
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  int rc = ensure_tables(c);
  if (rc) return rc;

  // validate jobs and size the LDS window from the worst spread of search centres
  int max_uw = c->me_max_uw, max_uh = c->me_max_uh;
  unsigned ref_mask = c->me_ref_mask;
  if (resident) {
    if (prm->search_mode != c->me_last_mode || prm->search_range != c->me_last_R || prm->rdopt != c->me_last_rdopt ||
        prm->partition_mask != c->me_last_mask || prm->level_mv_min != c->me_last_lvl[0] || prm->level_mv_max != c->me_last_lvl[1])
      return jm_fail(c, JMHIP_ERR_ARG, "jmhip_me_frame: resident jobs need the search geometry of the call that uploaded them");
    for (size_t k = 0; k < c->refs.size(); k++)
      if (((ref_mask >> k) & 1) && (!c->refs[k].has_pic || (prm->subpel && !c->refs[k].has_luma_sub)))
        return jm_fail(c, JMHIP_ERR_ARG, "jmhip_me_frame: reference slot of the resident jobs is not ready");
  } else {
    // a new upload invalidates the resident job set until it has been validated AND copied: an early return below must not
    // leave me_n describing the previous upload next to half-built index lists (a later resident call would launch on them)
    c->me_n = 0;
    c->fr_from_slices = false;                       // the frame stage's inputs come from this search again
    max_uw = max_uh = 0; ref_mask = 0; c->me_fast_idx.clear(); c->me_gen_idx.clear();
  }
  const bool full_mask = (prm->partition_mask & ((1ull << JMHIP_NPART) - 1)) == ((1ull << JMHIP_NPART) - 1);
  const bool metric_any = jm_me_metric_path(prm) || wide;      // other metrics / chroma term: me_metric.hip
  // its integer stage is only needed when the integer level itself differs from JM's default (SAD, luma only): otherwise the fast
  // kernels search and me_metric.hip refines
  const bool metric_path = metric_any && (eff.metric[0] != 0 || eff.chroma_me != 0 || wide);
  if (resident && metric_path != c->me_last_metric_path) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_me_frame: resident jobs need the metric path of the call that uploaded them");
  for (int i = 0; i < n && !resident; i++) {
    const jmhip_me_mb &m = mbs[i];
    if (m.mb_x < 0 || m.mb_x >= c->mbw || m.mb_y < 0 || m.mb_y >= c->mbh) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_me_frame: macroblock outside the picture");
    if (m.ref < 0 || m.ref >= (int)c->refs.size() || !c->refs[m.ref].has_pic) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_me_frame: reference slot not uploaded");
    if (prm->subpel && !c->refs[m.ref].has_luma_sub) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_me_frame: sub-pel planes of the reference not built (jmhip_interp_luma)");
    ref_mask |= 1u << m.ref;
    int x0 = 1 << 30, x1 = -(1 << 30), y0 = 1 << 30, y1 = -(1 << 30);
    for (int p = 0; p < JMHIP_NPART; p++) if ((prm->partition_mask >> p) & 1) {
      int cx, cy;
      const int s = prm->search_mode == JMHIP_SEARCH_FASTFULL ? 0 : p;
      host_center(prm, m.pred_mv[s][0], m.pred_mv[s][1], &cx, &cy);
      x0 = cx < x0 ? cx : x0; x1 = cx > x1 ? cx : x1; y0 = cy < y0 ? cy : y0; y1 = cy > y1 ? cy : y1;
    }
    const int uw = x1 - x0 + 2 * R + 1, uh = y1 - y0 + 2 * R + 1;
    // fast path: every partition searched, at least 64 candidate columns, and few DISTINCT centres: one work item per
    // distinct centre (FastFull and single-predictor macroblocks have one; JM's FullSearch typically a handful, the
    // neighbouring predictors being close). Beyond FAST_MAX_CENTRES the union-window kernel is cheaper.
    int reps[FAST_MAX_CENTRES], rcx[FAST_MAX_CENTRES], rcy[FAST_MAX_CENTRES], ng = 0;
    bool fast = !metric_path && full_mask && (2 * R + 1 >= (me_use_pair_kernel() ? 32 : 64)) && (2 * R + 1 + 15 <= 96) && i < (1 << 24);
    for (int p = 0; p < JMHIP_NPART && fast; p++) {
      int cx, cy, g;
      const int s = prm->search_mode == JMHIP_SEARCH_FASTFULL ? 0 : p;
      host_center(prm, m.pred_mv[s][0], m.pred_mv[s][1], &cx, &cy);
      for (g = 0; g < ng; g++) if (rcx[g] == cx && rcy[g] == cy) break;
      if (g < ng) continue;
      if (ng == FAST_MAX_CENTRES) { fast = false; break; }
      reps[ng] = p; rcx[ng] = cx; rcy[ng] = cy; ng++;
    }
    if (fast) {
      // bit 30: all 41 predictors of the macroblock are equal (the metric's workload; smooth motion in JM's own fields): one mv cost serves every partition
      bool uni = true;
      for (int p = 1; p < JMHIP_NPART && uni; p++) uni = m.pred_mv[p][0] == m.pred_mv[0][0] && m.pred_mv[p][1] == m.pred_mv[0][1];
      for (int g = 0; g < ng; g++) c->me_fast_idx.push_back(i | (reps[g] << 24) | (uni ? 1 << 30 : 0));
    }
    else {
      c->me_gen_idx.push_back(i);
      max_uw = uw > max_uw ? uw : max_uw; max_uh = uh > max_uh ? uh : max_uh;
    }
  }
  const int pitch = ((max_uw + 15 + 3) & ~3) + 8;       // +8: the 5th dword read of the last candidate column
  const int rows = max_uh + 15;
  const size_t lds = (size_t)pitch * rows + 16;
  // fast path window: 4 byte-shifted copies, copy stride == 8 (mod 32) dwords
  const int fpitch_dw = ((2 * R + 1 + 15 + 3) >> 2) + 2, frows = 2 * R + 1 + 15;
  int fcs = fpitch_dw * frows;
  while ((fcs & 31) != 8) fcs++;
  size_t flds = ((size_t)3 * fcs + (size_t)fpitch_dw * frows) * 4;
  if (flds < (size_t)(JMHIP_NPART + 1) * 256 * 4) flds = (size_t)(JMHIP_NPART + 1) * 256 * 4;     // reused by the final key transpose
  if (!c->me_fast_idx.empty() && flds > 60 * 1024) return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_me_frame: search range too large for the fast-path LDS window");
  if (lds > 60 * 1024) return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_me_frame: search centres of one macroblock are too far apart for one LDS window");

  if ((rc = jm_me_arrays_ensure(c, n))) return rc;
  if ((rc = jm_ensure_ref_table(c))) return rc;
  if (!resident) {
    JM_HIP_CHECK(c, hipMemcpyAsync(c->me_jobs_dev, mbs, sizeof(jmhip_me_mb) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    // index lists of the two integer-search kernels: [0, nfast) fast, [nfast, n) generic
    std::vector<int> idx(c->me_fast_idx);
    idx.insert(idx.end(), c->me_gen_idx.begin(), c->me_gen_idx.end());
    JM_HIP_CHECK(c, hipMemcpyAsync(c->me_idx_dev, idx.data(), sizeof(int) * idx.size(), hipMemcpyHostToDevice, c->stream));
    JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));     // idx is a stack vector
  }
  c->me_n = n; c->me_ref_mask = ref_mask; c->me_max_uw = max_uw; c->me_max_uh = max_uh;
  c->me_last_mode = prm->search_mode; c->me_last_R = prm->search_range; c->me_last_rdopt = prm->rdopt; c->me_last_mask = prm->partition_mask;
  c->me_last_lvl[0] = prm->level_mv_min; c->me_last_lvl[1] = prm->level_mv_max;
  c->me_last_metric_path = metric_path;
  if (metric_any && (rc = jm_me_metric_check(c, &eff, ref_mask, "jmhip_me_frame"))) return rc;

  MeDev P{};
  P.mode = prm->search_mode; P.R = R; P.rdopt = prm->rdopt; P.is_b = prm->is_b_slice;
  P.lvl_min = prm->level_mv_min; P.lvl_max = prm->level_mv_max;
  P.lam_f = prm->lambda[0]; P.lam_h = prm->lambda[1]; P.lam_q = prm->lambda[2];
  P.t8x8 = prm->transform8x8_mode ? 1 : 0; P.subpel = prm->subpel ? 1 : 0;
  P.wp_on = prm->wp_enable ? 1 : 0; P.wp_round = prm->wp_round; P.wp_denom = prm->wp_denom;
  for (int k = 0; k < 16; k++) { P.wp_w[k] = prm->wp_weight[k]; P.wp_o[k] = prm->wp_offset[k]; }
  P.mask = prm->partition_mask & ((1ull << JMHIP_NPART) - 1);
  P.W = c->W; P.H = c->H; P.Wp = c->Wp; P.Hp = c->Hp;
  P.win_pitch = pitch; P.win_rows = rows;
  P.cur = c->cur_y;
  P.ref_y = reinterpret_cast<const uint8_t *const *>(c->ref_ptrs_dev);
  P.ref_sub = reinterpret_cast<const uint8_t *const *>(c->ref_ptrs_dev) + 32;

  const int nfast = (int)c->me_fast_idx.size(), ngen = (int)c->me_gen_idx.size();
#ifdef JMHIP_STAMPS
  static unsigned long long *stamps_dev = nullptr;
  if (!stamps_dev) { (void)hipMalloc((void **)&stamps_dev, 512 * 4 * 8 * 8); }
  (void)hipMemsetAsync(stamps_dev, 0, 512 * 4 * 8 * 8, c->stream);
  P.stamps = stamps_dev;
#endif
  if (metric_path) {                                    // integer and sub-pel stage in one kernel
    jm_stage_begin(c, JMHIP_STAGE_ME_INT);
    rc = jm_launch_me_metric(c, &eff, P, (const jmhip_me_mb *)c->me_jobs_dev, (const int *)c->me_idx_dev, (jmhip_me_result *)c->me_res_dev, ngen, lds, 0);
    jm_stage_end(c, JMHIP_STAGE_ME_INT);
    JM_HIP_CHECK(c, hipGetLastError());
    return rc;
  }
  jm_stage_begin(c, JMHIP_STAGE_ME_INT);
  if (nfast) {
    MeDev PF = P;
    PF.win_pitch = fpitch_dw * 4; PF.win_rows = frows; PF.win_copy_stride = fcs;
    const int use_pair = me_use_pair_kernel();
    size_t plds = ((size_t)3 * fcs + (size_t)fpitch_dw * frows) * 4;          // the window; its memory is reused by the 44 x 128 key transpose
    if (plds < (size_t)44 * 128 * 4) plds = (size_t)44 * 128 * 4;
    // persistent form: needs the padded integer plane (plane 0 of the quarter-pel stack) of every used reference and no weighted reference ME
    bool pers = use_pair && me_use_pers_kernel() && !P.wp_on;
    for (size_t k = 0; k < c->refs.size() && pers; k++) if (((ref_mask >> k) & 1) && !c->refs[k].has_luma_sub) pers = false;
    if (pers) {
      const int ppitch = (2 * R + 1 + 17) / 4 + 1, nw = ppitch * frows, nwp = ((nw + 63) & ~63) + 64;
      const size_t wlds = (size_t)2 * nwp * 4;
      static int per_cu[64] = {0}, ncu[64] = {0};
      const int dev = c->cfg.device & 63;
      if (!ncu[dev]) {
        hipDeviceProp_t prop;
        JM_HIP_CHECK(c, hipGetDeviceProperties(&prop, c->cfg.device));
        int nb = 0;
        JM_HIP_CHECK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, me_int_pers_kernel, 256, 20480));
        per_cu[dev] = nb < 1 ? 1 : nb; ncu[dev] = prop.multiProcessorCount;
      }
      int grid = (per_cu[dev] * ncu[dev]) & ~7;
      if (grid < 8) grid = 8;
      if (const char *e = getenv("JMHIP_ME_PERS_GRID")) { const int g = atoi(e) & ~7; if (g >= 8 && g < grid) grid = g; }     // tests: few workgroups, many items each
      if (grid > jm_xcd_grid(nfast)) grid = jm_xcd_grid(nfast);
      PF.win_pitch = ppitch; PF.win_copy_stride = nwp;
      me_int_pers_kernel<<<grid, 256, wlds, c->stream>>>(PF, (const jmhip_me_mb *)c->me_jobs_dev, (const int *)c->me_idx_dev, (jmhip_me_result *)c->me_res_dev, nfast);
    }
    else if (use_pair) me_int_pair_kernel<false><<<jm_xcd_grid(nfast), 256, plds, c->stream>>>(PF, (const jmhip_me_mb *)c->me_jobs_dev, (const int *)c->me_idx_dev, (jmhip_me_result *)c->me_res_dev, nfast, nullptr, 0, nullptr);
    else me_int_fast_kernel<<<jm_xcd_grid(nfast), 256, flds, c->stream>>>(PF, (const jmhip_me_mb *)c->me_jobs_dev, (const int *)c->me_idx_dev, (jmhip_me_result *)c->me_res_dev, nfast);
  }
  if (ngen)
    me_int_kernel<<<jm_xcd_grid(ngen), 256, lds, c->stream>>>(P, (const jmhip_me_mb *)c->me_jobs_dev, (const int *)c->me_idx_dev + nfast, (jmhip_me_result *)c->me_res_dev, ngen);
  jm_stage_end(c, JMHIP_STAGE_ME_INT);
  JM_HIP_CHECK(c, hipGetLastError());
#ifdef JMHIP_STAMPS
  {
    static int shots = 0;
    if (++shots == 3) {
      std::vector<unsigned long long> h(512 * 4 * 8);
      (void)hipStreamSynchronize(c->stream);
      (void)hipMemcpy(h.data(), stamps_dev, h.size() * 8, hipMemcpyDeviceToHost);
      double acc[4][7] = {{0}};
      int cnt = 0;
      for (int b = 0; b < 512 && b < nfast; b++) { cnt++; for (int w = 0; w < 4; w++) for (int k = 1; k < 7; k++) acc[w][k] += (double)(h[(b * 4 + w) * 8 + k] - h[(b * 4 + w) * 8 + k - 1]); }
      const char *names[7] = {"", "setup", "lane-const", "main", "rest", "reduce", "final"};
      for (int w = 0; w < 4; w++) { fprintf(stderr, "STAMPS wave %d:", w); for (int k = 1; k < 7; k++) fprintf(stderr, " %s=%.0f", names[k], acc[w][k] / cnt); fprintf(stderr, "\n"); }
    }
  }
#endif
  if (P.subpel && metric_any) {                        // the refinement with the configured metrics, from the integer vectors and costs just left
    jm_stage_begin(c, JMHIP_STAGE_ME_SUB);
    rc = jm_launch_me_metric(c, &eff, P, (const jmhip_me_mb *)c->me_jobs_dev, nullptr, (jmhip_me_result *)c->me_res_dev, n, 0, 1);
    jm_stage_end(c, JMHIP_STAGE_ME_SUB);
    JM_HIP_CHECK(c, hipGetLastError());
    return rc;
  }
  if (P.subpel) {
#ifdef JMHIP_STAMPS
    (void)hipMemsetAsync(P.stamps, 0, 512 * 4 * 8 * 8, c->stream);
#endif
    if ((rc = jm_me_sub_tables(c))) return rc;
    jm_stage_begin(c, JMHIP_STAGE_ME_SUB);
    jm_launch_me_sub(c, P, (const jmhip_me_mb *)c->me_jobs_dev, (jmhip_me_result *)c->me_res_dev, n);
    jm_stage_end(c, JMHIP_STAGE_ME_SUB);
    JM_HIP_CHECK(c, hipGetLastError());
#ifdef JMHIP_STAMPS
    {
      static int sshots = 0;
      if (++sshots == 3) {
        std::vector<unsigned long long> h(512 * 32);
        (void)hipStreamSynchronize(c->stream);
        (void)hipMemcpy(h.data(), P.stamps, h.size() * 8, hipMemcpyDeviceToHost);
        double acc[12] = {0};
        for (int b = 0; b < 512; b++) for (int k = 1; k < 12; k++) acc[k] += (double)(h[b * 32 + k] - h[b * 32 + k - 1]);
        const char *nm[12] = {"", "setup", "h:leaders", "h:satd", "h:sum", "h:decide", "q:setup", "q:leaders", "q:satd", "q:sum", "q:decide", "store"};
        fprintf(stderr, "SUB STAMPS (cycles, avg over 512 blocks):");
        for (int k = 1; k < 12; k++) fprintf(stderr, " %s=%.0f", nm[k], acc[k] / 512);
        fprintf(stderr, "\n");
      }
    }
#endif
  }
  return JMHIP_OK;
}

extern "C" int jmhip_me_results_download(jmhip_ctx *c, jmhip_me_result *results, int n)
{
  if (!c || !results || n <= 0) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_me_results_download: NULL/empty arguments") : JMHIP_ERR_ARG;
  if (n > c->me_n) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_me_results_download: more results requested than jobs enqueued");
  JM_HIP_CHECK(c, hipMemcpyAsync(results, c->me_res_dev, sizeof(jmhip_me_result) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < n; i++)
    for (int p = 0; p < JMHIP_NPART; p++)
      if (results[i].cost_int[p] == -2) return jm_fail(c, JMHIP_ERR_DEVICE, "me_int_kernel: LDS window smaller than a macroblock's search area (internal sizing error)");
  return JMHIP_OK;
}

extern "C" int jmhip_me_frame(jmhip_ctx *c, const jmhip_me_params *prm, const jmhip_me_mb *mbs, int n, jmhip_me_result *results)
{
  int rc = jmhip_me_frame_async(c, prm, mbs, n);
  if (rc) return rc;
  return jmhip_me_results_download(c, results, n);
}
