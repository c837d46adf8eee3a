// me_xslice.hip -- the exhaustive searches of a P slice (FullSearch, FastFullSearch) as relaxation sweeps over the FAST frame kernels.
//
// jmhip_p_slice_search keeps JM's raster-order dependencies (predictors from the neighbours' final vectors, the field writes between the
// partitions of a macroblock, the low-complexity decision) by iterating to the unique fixpoint of "every macroblock = f(its predecessors)"
// (me_wave.hip). me_wave.hip runs one wave per macroblock that searches as it goes; for the exhaustive modes that wave
// spent its life building and re-reading SAD surfaces. Here the two halves are separated:
//
//   * a BlockMotionSearch call of FullSearch is a pure function of (macroblock, reference, partition, predictor) -- FastFullSearch: and of the
//     reference's 16x16 predictor, the window centre (src/mv-search.c:560, me_fullsearch.c:47, me_fullfast.c:491/:833). Its result is kept as a
//     RECORD: jobs[ref][mb].pred_mv[p] -> res[ref][mb].{mv_int, cost_int, mv, cost}[p], with a valid bit per record;
//   * x_sim_kernel replays encode_one_macroblock_low (src/md_low.c:46) for a macroblock WITHOUT searching: it walks the partitions in JM's
//     order, forms each predictor from the field as the earlier partitions left it (SetMotionVectorPredictor, src/mv-search.c:87), and takes the
//     record when its predictor is the one just formed; otherwise it stores the new predictor, marks the record as needed and carries on with
//     the stale result as its guess. It ends with the decision and hands the macroblock's sixteen field entries on;
//   * the needed records are computed by the frame kernels over device-resident work lists: me_int_pair_kernel (one item per macroblock,
//     reference and distinct window centre), me_sub_kernel (per macroblock and reference, only the needed partitions), x_skip_kernel (the
//     skip cost of GetSkipCostMB, src/mv-search.c:1136, per distinct skip vector);
//   * a sweep = sim -> integer search -> refinement -> skip costs; macroblocks are re-simulated when a neighbour changed what it hands on or
//     when records of their own arrived. A sweep with no needed record and no changed macroblock is the fixpoint: every macroblock then
//     replayed JM's algorithm on final inputs with exact records, which is what JM's sequential loop computes.
// Nothing here waits for the host inside a sweep: list lengths stay on the device; the host looks at the counters every few sweeps.
//
// Covered: search modes -1 / 0, 16 <= search_range <= 40 (the pair-lane kernel), SAD at integer and Hadamard SAD at sub-pel positions (JM's
// defaults), RestrictSearchRange 2, the 4x4 transform, explicit weights, several slices per call, rdopt-aware records. Everything else stays
// with me_wave.hip's kernels (jmhip_p_slice_search decides).
#include "me_common.h"
#include <time.h>

namespace {

constexpr int XR = JMHIP_SLICE_REFS;
constexpr int X_MAX_SWEEPS = 1024;               // counter slots (one per sweep) of a call
constexpr int X_CNT_LIST = JM_SHARDS * JM_SHARD_STRIDE;      // ints per shard-counter block
constexpr int X_CNT_FLAGS = 3 * X_CNT_LIST;                  // [+0] some macroblock changed what it hands on [+1] some macroblock asked for records [+2..4] statistics: changed, needing, simulated
constexpr int X_CNT_SWEEP = X_CNT_FLAGS + 16;                // ints per sweep

struct XSkip { short mvx, mvy; int cost; int state; int pad; };      // state 1: cost is GetSkipCostMB at (mvx, mvy) on this picture

struct XDev {
  int search_mode, R, num_refs, rdopt;
  int valid[8];
  int lambda_mf[3], ref_cost1, md_metric;
  int lvl_min, lvl_max;
  int mb_first, mb_count, slice_mbs;
  int wp_pred, wp_round, wp_denom;
  short wp_weight0, wp_offset0;
  int wp_me; short wp_weight[XR], wp_offset[XR];     // explicit weights in the searches (per list-0 index)
  int ref_slot[XR];
  int W, H, Wp, Hp, mbw, mbh, w4, nmb;
  int first_sweep;
  const uint8_t *cur;
  const uint8_t *const *ref_sub;
  int8_t *ref_idx; short *mv;                    // enc_picture->ref_idx / mv [LIST_0], the slice search's picture arrays (me_wave.hip)
  jmhip_mb_inter *out;
  jmhip_me_mb *jobs; jmhip_me_result *res;       // [ref][macroblock]
  unsigned long long *valid_rec, *need_rec;     // [ref][macroblock]: bit p
  XSkip *skip;
  uint8_t *pending;                              // records of this macroblock were asked for in the last sweep
  const uint8_t *chg_prev; uint8_t *chg_next;    // which macroblocks changed what they hand on: last sweep / this sweep
  int *items, *sub_list, *skip_list;             // sharded lists (jmhip_internal.h): shard = macroblock address % JM_SHARDS
  int cap_items, cap_sub, cap_skip;              // entries per shard
  int *cnt;                                      // this sweep's counters: three shard-counter blocks (items, refinements, skip costs), then flags / statistics
  int debug;                                     // JMHIP_X_DEBUG (timing experiments only, results are WRONG): 1 no refinement loads, 2 no integer scan, 4 no window staging
  int stats;                                     // count simulated / changed / needing macroblocks exactly (JMHIP_SLICE_TRACE); else flags only
  int chain_budget;                              // CHAIN, FullSearch: records a macroblock searches in place per sweep; what it misses beyond that goes to the work lists
};

// static description of partition p for the replay: block type, rectangle in 4x4 units, the ring cells of its neighbours A, B, C, D
// (mb_access.c getLuma4x4Neighbour via mv-search.c:100-127), whether C lies in a part of the macroblock coded later (:108-127), and the
// directional rule of 16x8 / 8x16 blocks (:181-210)
// Two aligned dwords per partition, decoded with shifts. As a struct of twelve int8 fields hipcc (ROCm 7.2) fetched neighbouring fields with ONE
// vector load at an odd address (global_load_ushort offset:5, global_load_dword offset:1 from the __constant__ array), and on MI355X those
// loads returned other bytes than the fields: the replay then read a wrong C cell for some partitions -- found with a per-step dump against
// an optnone build of the same source.
struct alignas(8) XPart {
  uint32_t g, n;                                 // g: bt | x4 << 4 | y4 << 8 | w4 << 12 | h4 << 16 | cblk << 20 | dir << 24;  n: a | b << 8 | c << 16 | d << 24
  __host__ __device__ int bt() const { return g & 15; }
  __host__ __device__ int x4() const { return (g >> 4) & 15; }
  __host__ __device__ int y4() const { return (g >> 8) & 15; }
  __host__ __device__ int w4() const { return (g >> 12) & 15; }
  __host__ __device__ int h4() const { return (g >> 16) & 15; }
  __host__ __device__ int cblk() const { return (g >> 20) & 1; }
  __host__ __device__ int dir() const { return (g >> 24) & 7; }
  __host__ __device__ int a() const { return n & 255; }
  __host__ __device__ int b() const { return (n >> 8) & 255; }
  __host__ __device__ int c() const { return (n >> 16) & 255; }
  __host__ __device__ int d() const { return n >> 24; }
};
__constant__ XPart c_xpart[JMHIP_NPART];
XPart h_xpart[JMHIP_NPART];

void build_xparts()
{
  build_part_table();
  for (int p = 0; p < JMHIP_NPART; p++) {
    const PartInfo &q = h_part[p];
    const int a = (q.y4 + 1) * 6 + q.x4, b = q.y4 * 6 + q.x4 + 1, c = q.y4 * 6 + q.x4 + q.w4 + 1, d = q.y4 * 6 + q.x4;
    const int mb_x = 4 * q.x4, mb_y = 4 * q.y4, bsx = 4 * q.w4, bsy = 4 * q.h4;
    int cblk = 0;
    if (mb_y > 0) {
      if (mb_x < 8) { if (mb_y == 8) { if (bsx == 16) cblk = 1; } else if (mb_x + bsx == 8) cblk = 1; }
      else if (mb_x + bsx == 16) cblk = 1;
    }
    int dir = 0;
    if (bsx == 8 && bsy == 16) dir = mb_x == 0 ? 1 : 2;
    else if (bsx == 16 && bsy == 8) dir = mb_y == 0 ? 3 : 4;
    h_xpart[p].g = (uint32_t)(q.bt | (q.x4 << 4) | (q.y4 << 8) | (q.w4 << 12) | (q.h4 << 16) | (cblk << 20) | (dir << 24));
    h_xpart[p].n = (uint32_t)(a | (b << 8) | (c << 16) | (d << 24));
  }
}

// a cell of the 5 x 6 ring of 4x4 blocks round (and inside) the macroblock: reference + 2 in the top nibble (0: the block is not available),
// the vector as two 14-bit fields
__device__ __forceinline__ uint32_t pk_cell(int ref, int mvx, int mvy) { return ((uint32_t)(ref + 2) << 28) | (((uint32_t)mvy & 0x3fffu) << 14) | ((uint32_t)mvx & 0x3fffu); }
__device__ __forceinline__ int cell_mvx(uint32_t c) { return ((int)(c << 18)) >> 18; }
__device__ __forceinline__ int cell_mvy(uint32_t c) { return ((int)(c << 4)) >> 18; }
__device__ __forceinline__ int mv_x(uint32_t v) { return (int)(short)(v & 0xffffu); }
__device__ __forceinline__ int mv_y(uint32_t v) { return (int)(short)(v >> 16); }
__device__ __forceinline__ uint32_t pk_mv(int x, int y) { return ((uint32_t)y << 16) | ((uint32_t)x & 0xffffu); }
__device__ __forceinline__ int part_of(int bt, int bx, int by)
{
  const int b8 = (by >> 1) * 2 + (bx >> 1);
  switch (bt) {
  case 1: return 0;
  case 2: return 1 + (by >> 1);
  case 3: return 3 + (bx >> 1);
  case 4: return 5 + b8;
  case 5: return 9 + b8 * 2 + (by & 1);
  case 6: return 17 + b8 * 2 + (bx & 1);
  default: return 25 + b8 * 4 + (by & 1) * 2 + (bx & 1);
  }
}

// is macroblock (nx, ny) an available neighbour of macroblock `cur` (address): inside the picture, inside [mb_first, mb_first + mb_count) and
// in the same slice (src/mb_access.c:30-36)
__device__ __forceinline__ bool mb_avail(const XDev &D, int cur, int nx, int ny)
{
  if (nx < 0 || ny < 0 || nx >= D.mbw || ny >= D.mbh) return false;
  const int a = ny * D.mbw + nx;
  if (a < D.mb_first || a >= D.mb_first + D.mb_count) return false;
  if (D.slice_mbs > 0) {
    const int lo = D.mb_first + ((cur - D.mb_first) / D.slice_mbs) * D.slice_mbs;
    if (a < lo) return false;
  }
  return true;
}

// SetMotionVectorPredictor (src/mv-search.c:87) for reference `ref` from the neighbour cells A, B, C, D of a partition (C already void where
// it lies in a part of the macroblock coded later), dir: the directional rule of 16x8 / 8x16 blocks. Returns the predictor as an int16 pair.
// A pure function of six scalars, kept out of line (one copy instead of four inlined ones in the replay's loops).
__device__ __noinline__ uint32_t x_median_pred(uint32_t ca, uint32_t cb, uint32_t cc, uint32_t cd, int dir, int ref)
{
  if ((cc >> 28) == 0) cc = cd;
  const int avA = (ca >> 28) != 0, avB = (cb >> 28) != 0, avC = (cc >> 28) != 0;
  const int rL = avA ? (int)(ca >> 28) - 2 : -1, rU = avB ? (int)(cb >> 28) - 2 : -1, rUR = avC ? (int)(cc >> 28) - 2 : -1;
  int type = 0;
  if (rL == ref && rU != ref && rUR != ref) type = 1;
  else if (rL != ref && rU == ref && rUR != ref) type = 2;
  else if (rL != ref && rU != ref && rUR == ref) type = 3;
  if (dir == 1) { if (rL == ref) type = 1; }
  else if (dir == 2) { if (rUR == ref) type = 3; }
  else if (dir == 3) { if (rU == ref) type = 2; }
  else if (dir == 4) { if (rL == ref) type = 1; }
  int pv[2];
#pragma unroll
  for (int hv = 0; hv < 2; hv++) {
    const int a = avA ? (hv ? cell_mvy(ca) : cell_mvx(ca)) : 0, b = avB ? (hv ? cell_mvy(cb) : cell_mvx(cb)) : 0, c = avC ? (hv ? cell_mvy(cc) : cell_mvx(cc)) : 0;
    if (type == 0) pv[hv] = !(avB || avC) ? a : max(min(a, b), min(max(a, b), c));        // the median of the three
    else pv[hv] = type == 1 ? a : (type == 2 ? b : c);
  }
  return pk_mv(pv[0], pv[1]);
}
// search centre from the predictor (src/mv-search.c:752-762 / me_fullfast.c:552-563), as search_center() of the frame kernels
__device__ __forceinline__ uint32_t x_centre(const XDev &D, int pmx, int pmy)
{
  int mx = pmx / 4, my = pmy / 4;
  if (!D.rdopt) { mx = clampi(mx, -D.R, D.R); my = clampi(my, -D.R, D.R); }
  mx = clampi(mx, -2047 + D.R, 2047 - D.R);
  my = clampi(my, D.lvl_min + D.R, D.lvl_max - D.R);
  return pk_mv(mx, my);
}

__device__ __forceinline__ int x_refbits(int r) { return r == 0 ? 1 : (r < 3 ? 3 : 5); }       // src/mv-search.c:344-352

// x_sim_kernel: the replay of encode_one_macroblock_low (src/md_low.c:46) on records, one macroblock per workgroup. The control flow is uniform and
// everything the replay touches lives in registers: lane L < 30 holds ring cell L, lane p < 41 holds partition p's descriptor and its records for
// every reference (NR is a template parameter so that those are plain registers); a cell or record is read with v_readlane (its index is
// uniform), a field write is a per-lane select.
//   CHAIN = false (64 threads): a record whose predictor is not the one just formed is ASKED FOR (need mask -> work lists for the frame kernels)
//     and the replay carries on with the stale result as its guess -- sweep 0 of a picture, where every record of every macroblock is wanted and
//     the frame kernels compute them at their full rate;
//   CHAIN = true (256 threads, the four waves replaying in lockstep on equal registers): such a record is COMPUTED on the spot by the whole
//     workgroup -- the partition's integer search over its own window (FullPelBlockMotionSearch src/me_fullsearch.c:47 / the partition's scan of
//     FastFullPelBlockMotionSearch src/me_fullfast.c:833, from a reference window staged in LDS) and its refinement (SubPelBlockMotionSearch
//     src/me_fullsearch.c:341) -- and the replay continues with the exact result: one pass leaves the macroblock consistent with what its
//     neighbours handed on, however long its internal chain of predictors is. The sweeps after the first are of this kind: their number is
//     the length of the chains ACROSS macroblocks only.
// SAD of a (4 RW) x BH block: reference rows from the staged window (dword pointer q at the block's first row, the block starting `sh` bytes into
// q[0]; wpdw dwords per window row), current rows from LDS (16-byte rows). Four rows' loads are in flight together. row0: the first row's SAD.
template <int RW, int BH>
__device__ __forceinline__ void x_block_sad(const uint32_t *q, int wpdw, unsigned sh, const uint8_t *cur, unsigned &v, unsigned &row0)
{
#pragma unroll
  for (int y0 = 0; y0 < BH; y0 += 4) {
    uint32_t d[4][RW + 1], c[4][RW];
#pragma unroll
    for (int y = 0; y < 4; y++) {
#pragma unroll
      for (int i = 0; i <= RW; i++) d[y][i] = q[(y0 + y) * wpdw + i];
#pragma unroll
      for (int i = 0; i < RW; i++) c[y][i] = reinterpret_cast<const uint32_t *>(cur + (y0 + y) * 16)[i];
    }
#pragma unroll
    for (int y = 0; y < 4; y++) {
#pragma unroll
      for (int i = 0; i < RW; i++) v = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(d[y][i + 1], d[y][i], sh), c[y][i], v);
      if (y0 + y == 0) row0 = v;
    }
  }
}

#ifdef JMHIP_X_PROF
__device__ unsigned long long g_xprof[16];
__device__ unsigned long long g_xprof_n;
#define XPROF_T0 unsigned long long xp_t = __builtin_amdgcn_s_memtime()
#define XPROF(k) do { const unsigned long long xp_n = __builtin_amdgcn_s_memtime(); if (tid == 0) atomicAdd(&g_xprof[k], xp_n - xp_t); xp_t = xp_n; } while (0)
#else
#define XPROF_T0 do { } while (0)
#define XPROF(k) do { } while (0)
#endif

constexpr int X_WIN_MARGIN = 4;                  // CHAIN: the staged window reaches this far beyond the range round its centre (FullSearch centres differ by a pel or two)

template <bool FFS, int NR, bool CHAIN>
__global__ __launch_bounds__(CHAIN ? 256 : 64) void x_sim_kernel(XDev D)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t x_win[];           // CHAIN: reference window, wside x wside bytes
  __shared__ uint32_t s_ring[32];
  __shared__ __attribute__((aligned(16))) uint8_t s_cur[16][16];
  __shared__ unsigned s_red[8];
  __shared__ int s_dist[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int addr = D.mb_first + (int)blockIdx.x;
  const int mbx = addr % D.mbw, mby = addr / D.mbw;
  // ---- is there anything new for this macroblock? (the four neighbouring macroblocks' availability: uniform, computed once)
  const bool aA = mb_avail(D, addr, mbx - 1, mby), aB = mb_avail(D, addr, mbx, mby - 1), aC = mb_avail(D, addr, mbx + 1, mby - 1), aD = mb_avail(D, addr, mbx - 1, mby - 1);
  bool active = D.first_sweep != 0 || D.pending[addr] != 0;
  if (!active) active = (aA && D.chg_prev[addr - 1]) || (aB && D.chg_prev[addr - D.mbw]) || (aC && D.chg_prev[addr - D.mbw + 1]) || (aD && D.chg_prev[addr - D.mbw - 1]);
  if (!active) { if (tid == 0) D.chg_next[addr] = 0; return; }

#ifdef JMHIP_X_PROF
  const unsigned long long xp_kernel_t0 = __builtin_amdgcn_s_memtime();
#endif
  // ---- stage: the ring of the motion field (lanes 0..29), descriptors and records (lanes 0..40). CHAIN: neighbours may be handing on new
  // entries at this very moment, and the four waves must replay the SAME ring: wave 0 reads it, the others take it from LDS
  const int gy = lane / 6, gx = lane - gy * 6, lx = gx - 1, ly = gy - 1;       // this lane's cell; (lx, ly) inside the macroblock for its own sixteen
  const bool own = lane < 30 && gy >= 1 && gx >= 1 && gx <= 4;
  uint32_t cell = 0u;
  if (lane < 30 && (!CHAIN || wave == 0)) {
    const bool av = gy == 0 ? (gx == 0 ? aD : (gx == 5 ? aC : aB)) : (gx == 0 && aA);
    cell = own ? pk_cell(-1, 0, 0) : 0u;                                    // the macroblock's own blocks: nothing written yet
    if (!own && av) {
      const size_t at = (size_t)(4 * mby - 1 + gy) * D.w4 + (4 * mbx - 1 + gx);
      cell = pk_cell(D.ref_idx[at], D.mv[at * 2], D.mv[at * 2 + 1]);
    }
    if (CHAIN) s_ring[lane] = cell;
  }
  if (CHAIN) {
    if (tid < 64) *reinterpret_cast<uint32_t *>(&s_cur[tid >> 2][(tid & 3) * 4]) = *reinterpret_cast<const uint32_t *>(D.cur + (size_t)(mby * 16 + (tid >> 2)) * D.W + mbx * 16 + (tid & 3) * 4);
    __syncthreads();
    cell = lane < 30 ? s_ring[lane] : 0u;
  }
  const int pl = min(lane, JMHIP_NPART - 1);
  const uint32_t dg = c_xpart[pl].g, dn = c_xpart[pl].n;
  uint32_t pm[NR], rm[NR]; int rc[NR];
  unsigned long long val[NR], need[NR];
  uint32_t mv16[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const size_t j = (size_t)r * D.nmb + addr;
    pm[r] = *reinterpret_cast<const uint32_t *>(D.jobs[j].pred_mv[pl]);
    rm[r] = *reinterpret_cast<const uint32_t *>(D.res[j].mv[pl]);
    rc[r] = D.res[j].cost[pl];
    // the records asked for in the last sweep have been computed since: they are valid now
    val[r] = D.first_sweep ? 0ull : (D.valid_rec[j] | D.need_rec[j]);
    need[r] = 0ull;
    mv16[r] = 0u;
  }
  XSkip sk = D.skip[addr];
  if (D.first_sweep) sk.state = 0;

  auto rl = [&](uint32_t v, int k) __attribute__((always_inline)) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)v, k); };
  // SetMotionVectorPredictor of the partition with descriptor words (g, n) for reference ref, from the ring in `cell`
  auto predict = [&](uint32_t g, uint32_t n, int ref) __attribute__((always_inline)) -> uint32_t {
    const uint32_t ca = rl(cell, n & 255), cb = rl(cell, (n >> 8) & 255), cd = rl(cell, n >> 24);
    uint32_t cc = rl(cell, (n >> 16) & 255);
    if ((g >> 20) & 1) cc = 0u;
    return x_median_pred(ca, cb, cc, cd, (g >> 24) & 7, ref);
  };

  // ---- CHAIN: the searches of one partition by the whole workgroup --------------------------------------------------------------------------
  const int wside = 2 * (D.R + X_WIN_MARGIN) + 1 + 15;                       // staged window: wside x wside samples, pitch wside rounded up to a dword
  const int wpitch = (wside + 3) & ~3;
  int win_ref = -1, win_cx = 0, win_cy = 0;                                  // what the window holds: reference, centre (pels)
  const size_t psz = (size_t)D.Wp * D.Hp;
  // stage the integer-pel window of reference r round (scx, scy): plane 0 of the quarter-pel stack with per-sample clamping (its ring replicates the
  // picture's edge, so that equals JM's clamp of the block origin, src/refbuf.c:37), weighted where the search is (computeSADWP src/me_distortion.c:413)
  auto stage_window = [&](int r, int scx, int scy) __attribute__((always_inline)) {
    const uint8_t *pl0 = D.ref_sub[D.ref_slot[r]];
    const int Rs = D.R + X_WIN_MARGIN, ox = mbx * 16 + scx - Rs + JMHIP_PAD, oy = mby * 16 + scy - Rs + JMHIP_PAD;
    __syncthreads();
    for (int i = tid; i < wside * (wpitch >> 2); i += 256) {
      const int wy = i / (wpitch >> 2), wx = (i - wy * (wpitch >> 2)) * 4;
      const uint8_t *row = pl0 + (size_t)clampi(oy + wy, 0, D.Hp - 1) * D.Wp;
      uint32_t v = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) v |= (uint32_t)row[clampi(ox + wx + k, 0, D.Wp - 1)] << (8 * k);
      if (D.wp_me) v = wp_apply4(v, D.wp_weight[r], D.wp_offset[r], D.wp_round, D.wp_denom);
      *reinterpret_cast<uint32_t *>(x_win + (size_t)wy * wpitch + wx) = v;
    }
    __syncthreads();
    win_ref = r; win_cx = scx; win_cy = scy;
  };
  // the minimum of a 32-bit key over the workgroup
  auto wg_min = [&](unsigned key) __attribute__((always_inline)) -> unsigned {
    for (int o = 1; o < 64; o <<= 1) key = min(key, (unsigned)__shfl_xor((int)key, o));
    __syncthreads();
    if (lane == 0) s_red[wave] = key;
    __syncthreads();
    return min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
  };
  // BlockMotionSearch of partition p (descriptor g) for reference r with predictor (pmx, pmy): integer search round (cx, cy), then sub-pel.
  // Returns the cost; *mv_out: the vector (int16 pair). Also leaves the integer result in the result record (the ABI's mv_int / cost_int).
  auto search_partition = [&](int r, int p, uint32_t g, int pmx, int pmy, int cx, int cy, uint32_t *mv_out) __attribute__((always_inline)) -> int {
    const int bt = g & 15, x4 = (g >> 4) & 15, y4 = (g >> 8) & 15, bsx = ((g >> 12) & 15) * 4, bsy = ((g >> 16) & 15) * 4;
    const int pic_x = mbx * 16 + 4 * x4, pic_y = mby * 16 + 4 * y4, R = D.R, side = 2 * R + 1, npos = side * side, lam = D.lambda_mf[0];
    XPROF_T0;
    if ((win_ref != r || iabs(cx - win_cx) > X_WIN_MARGIN || iabs(cy - win_cy) > X_WIN_MARGIN) && !(D.debug & 4)) stage_window(r, cx, cy);
    XPROF(0);
    if (D.debug & 4) { win_cx = cx; win_cy = cy; }
    const int Rs = R + X_WIN_MARGIN;
    // window coordinates of the block at displacement (0, 0) from the centre, minus the range: candidate (dx, dy) reads from (wx0 + dx + R, wy0 + dy + R)
    const int wx0 = 4 * x4 + cx - win_cx + Rs - R, wy0 = 4 * y4 + cy - win_cy + Rs - R;
    const int check00 = (!FFS && !D.rdopt && bt == 1 && r == 0), w16 = (lam * 16) >> 16;
    unsigned best = 0xffffffffu;
    const float rside = 1.0f / (float)side;
    for (int k = tid; k < ((D.debug & 2) ? 256 : npos); k += 256) {
      const int row = (int)(((float)k + 0.5f) * rside), dy = row - R, dx = k - row * side - R;       // k / side: exact for k < 2^13 (me_wave.hip surface_search)
      // (offsets, not pointer arithmetic through integers: a pointer rebuilt from uintptr_t is a GENERIC pointer and every read a flat_load)
      const int woff = (wy0 + dy + R) * wpitch + (wx0 + dx + R);
      const unsigned sh = (unsigned)(woff & 3);
      const uint32_t *q = reinterpret_cast<const uint32_t *>(x_win) + (woff >> 2);
      unsigned v = 0, row0 = 0;
      const uint8_t *cb = &s_cur[4 * y4][4 * x4];
      const int wpdw = wpitch >> 2;
      switch (bt) {                                                          // (uniform)
      case 1: x_block_sad<4, 16>(q, wpdw, sh, cb, v, row0); break;
      case 2: x_block_sad<4, 8>(q, wpdw, sh, cb, v, row0); break;
      case 3: x_block_sad<2, 16>(q, wpdw, sh, cb, v, row0); break;
      case 4: x_block_sad<2, 8>(q, wpdw, sh, cb, v, row0); break;
      case 5: x_block_sad<2, 4>(q, wpdw, sh, cb, v, row0); break;
      case 6: x_block_sad<1, 8>(q, wpdw, sh, cb, v, row0); break;
      default: x_block_sad<1, 4>(q, wpdw, sh, cb, v, row0); break;
      }
      int mc = mv_cost(lam, ((cx + dx) << 2) - pmx, ((cy + dy) << 2) - pmy);
      int tie = spiral_pos(dx, dy) + 1;
      int sad = (int)v;
      if (check00 && ((pic_x + cx + dx) << 2) == pic_x && ((pic_y + cy + dy) << 2) == pic_y) {       // check_for_00 compares quarter-pel with pel units, src/me_fullsearch.c:129
        mc -= w16;
        if (dx == 0 && dy == 0 && mc < 0) sad = (int)row0;                  // INT_MAX - (negative) wraps: computeSAD leaves after row 0 (me_int.hip wrapped_bound_00)
      }
      if (FFS && !D.rdopt && cx + dx == 0 && cy + dy == 0) tie = 0;          // pos_00 is tried first and keeps ties, src/me_fullfast.c:867
      best = min(best, ((unsigned)(mc + sad + 4096) << TIE_BITS) | (unsigned)tie);
    }
    XPROF(1);
    best = wg_min(best);
    XPROF(2);
    int icost = (int)(best >> TIE_BITS) - 4096, mvx, mvy;
    {
      const int tie = (int)(best & ((1u << TIE_BITS) - 1));
      if (tie == 0) { mvx = 0; mvy = 0; }
      else { int ddx, ddy; spiral_offset(tie - 1, &ddx, &ddy); mvx = cx + ddx; mvy = cy + ddy; }
    }
    if (tid == 0) {
      jmhip_me_result &o = D.res[(size_t)r * D.nmb + addr];
      o.mv_int[p][0] = (int16_t)mvx; o.mv_int[p][1] = (int16_t)mvy; o.cost_int[p] = icost;
    }
    mvx <<= 2; mvy <<= 2;
    // ---- SubPelBlockMotionSearch (src/me_fullsearch.c:341): Hadamard SAD at the 9 half-pel then 8 quarter-pel positions; the integer cost is not
    // carried (the metrics differ), the half-pel minimum is (mv-search.c:785-788, :396-397)
    const uint8_t *planes = D.ref_sub[D.ref_slot[r]];
    const int wpadx = D.Wp - 17, hpady = D.Hp - 17;
    const int max_x4 = (D.W - bsx + 2 * JMHIP_PAD) << 2, max_y4 = (D.H - bsy + 2 * JMHIP_PAD) << 2;
    const int check0 = (!D.rdopt && r == 0 && bt == 1 && mvx == 0 && mvy == 0);       // check_position0 (:361)
    const int nsx = bsx >> 2, nsub = nsx * (bsy >> 2);
    int min_mcost = INT_MAX;
    for (int phase = 0; phase < 2; phase++) {
      const int step = phase ? 1 : 2, first = phase ? 1 : 0, lamq = D.lambda_mf[phase ? 2 : 1], m = phase ? 0 : 1;
      const int p4x = ((pic_x + JMHIP_PAD) << 2) + mvx, p4y = ((pic_y + JMHIP_PAD) << 2) + mvy;
      const int umv = !((p4x > m) && (p4x < max_x4 - m) && (p4y > m) && (p4y < max_y4 - m));
      // (position, 4x4 sub-block) per thread: 9 x 16 at most
      const int pos = first + tid / nsub, sb = tid - (tid / nsub) * nsub, sy = sb / nsx, sx = sb - sy * nsx;
      int v = 0;
      if (pos < 9 && !(D.debug & 1)) {
        const int ox = p4x + c_s9x[pos] * step + sx * 16, oy = p4y + c_s9y[pos] * step + sy * 16;
        int ix = ox >> 2, iy = oy >> 2;
        if (umv) { ix = clampi(ix, 0, wpadx); iy = clampi(iy, 0, hpady); }     // origin clamp per 4x4 sub-block (computeSATD, me_distortion.c:678)
        const uint8_t *pp = planes + psz * ((oy & 3) * 4 + (ox & 3)) + (size_t)iy * D.Wp + ix;
        int df[4][4];
#pragma unroll
        for (int y = 0; y < 4; y++) {
          uint32_t q, hi;
          fetch_row(pp + (size_t)y * D.Wp, 4, &q, &hi);
          if (D.wp_me) q = wp_apply4(q, D.wp_weight[r], D.wp_offset[r], D.wp_round, D.wp_denom);
          const uint32_t cv = *reinterpret_cast<const uint32_t *>(&s_cur[4 * y4 + sy * 4 + y][4 * x4 + sx * 4]);
#pragma unroll
          for (int c = 0; c < 4; c++) df[y][c] = (int)((cv >> (8 * c)) & 255u) - (int)((q >> (8 * c)) & 255u);
        }
        v = satd4x4(df);
      }
      for (int o = 1; o < nsub; o <<= 1) v += __shfl_xor(v, o);             // nsub is a power of two <= 16: a position's sub-blocks sit in one wave
      __syncthreads();
      if (pos < 9 && sb == 0) s_dist[pos] = v;
      __syncthreads();
      int bestp = 0;
      for (int ps = first; ps < 9; ps++) {                                  // JM's scan: strict <, positions whose vector cost alone reaches the minimum are skipped
        int mcost = mv_cost(lamq, mvx + c_s9x[ps] * step - pmx, mvy + c_s9y[ps] * step - pmy);
        if (mcost >= min_mcost) continue;
        mcost += s_dist[ps];
        if (phase == 0 && ps == 0 && check0) mcost -= (lamq * 16) >> 16;
        if (mcost < min_mcost) { min_mcost = mcost; bestp = ps; }
      }
      if (bestp) { mvx += c_s9x[bestp] * step; mvy += c_s9y[bestp] * step; }
    }
    XPROF(3);
#ifdef JMHIP_X_PROF
    if (tid == 0) atomicAdd(&g_xprof_n, 1ull);
#endif
    *mv_out = pk_mv(mvx, mvy);
    return min_mcost;
  };

  // ---- the skip vector (FindSkipModeMotionVector, src/mv-search.c:1189): it depends on the ring only
  int skx = 0, sky = 0;
  {
    const uint32_t ca = rl(cell, 6), cb = rl(cell, 1);
    const int zl = (ca >> 28) == 0 || ca == pk_cell(0, 0, 0), za = (cb >> 28) == 0 || cb == pk_cell(0, 0, 0);
    if (!(za || zl)) { const uint32_t r = predict(rl(dg, 0), rl(dn, 0), 0); skx = mv_x(r); sky = mv_y(r); }
  }
  int skip_need = 0, skip_cost = sk.cost;
  if (!D.rdopt && !(sk.state == 1 && sk.mvx == skx && sk.mvy == sky)) skip_need = 1;      // (CHAIN = false: the stale cost stays the guess)
  if (CHAIN && skip_need) {
    // GetSkipCostMB (src/mv-search.c:1136): LumaPrediction of each 4x4 block at the skip vector from reference 0, distortion4x4 of the decision metric
    int v = 0;
    if (tid < 16) {
      const uint8_t *planes = D.ref_sub[D.ref_slot[0]];
      const int bx = (tid & 3) * 4, by = (tid >> 2) * 4;
      const int qx = ((mbx * 16 + bx + JMHIP_PAD) << 2) + skx, qy = ((mby * 16 + by + JMHIP_PAD) << 2) + sky;
      const int ix = clampi(qx >> 2, 0, D.Wp - 17), iy = clampi(qy >> 2, 0, D.Hp - 17);
      const uint8_t *pp = planes + psz * ((qy & 3) * 4 + (qx & 3)) + (size_t)iy * D.Wp + ix;
      int df[4][4];
#pragma unroll
      for (int y = 0; y < 4; y++) {
        uint32_t q, hi;
        fetch_row(pp + (size_t)y * D.Wp, 4, &q, &hi);
        if (D.wp_pred) q = wp_apply4(q, D.wp_weight0, D.wp_offset0, D.wp_round, D.wp_denom);
        const uint32_t cv = *reinterpret_cast<const uint32_t *>(&s_cur[by + y][bx]);
#pragma unroll
        for (int c = 0; c < 4; c++) df[y][c] = (int)((cv >> (8 * c)) & 255u) - (int)((q >> (8 * c)) & 255u);
      }
      if (D.md_metric == 2) v = satd4x4(df);
      else {
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
          for (int c = 0; c < 4; c++) v += iabs(df[y][c]);
      }
    }
    for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if (tid == 0) s_dist[0] = v;
    __syncthreads();
    sk.mvx = (short)skx; sk.mvy = (short)sky; sk.cost = s_dist[0]; sk.state = 1;
    skip_cost = sk.cost;
    __syncthreads();
  }
  skip_cost -= (D.lambda_mf[2] + 4096) >> 13;

  // a field write with ONE vector: the own cells inside rectangle (x4, y4, w4, h4) take (ref, mv)
  auto field_rect = [&](int x4, int y4, int w4, int h4, int ref, uint32_t m) __attribute__((always_inline)) {
    const bool in = own && lx >= x4 && lx < x4 + w4 && ly >= y4 && ly < y4 + h4;
    cell = in ? pk_cell(ref, mv_x(m), mv_y(m)) : cell;
  };
  // a field write after a decision: every own cell inside the rectangle takes reference `ref` and the vector block type `bt` left on it
  auto field_mode = [&](int x4, int y4, int w4, int h4, int ref, int bt) __attribute__((always_inline)) {
    uint32_t src = rm[0], m16 = mv16[0];
#pragma unroll
    for (int r = 1; r < NR; r++) if (ref == r) { src = rm[r]; m16 = mv16[r]; }
    const uint32_t m = bt == 1 ? m16 : (uint32_t)__shfl((int)src, own ? part_of(bt, lx, ly) : 0);
    const bool in = own && lx >= x4 && lx < x4 + w4 && ly >= y4 && ly < y4 + h4;
    cell = in ? pk_cell(ref, mv_x(m), mv_y(m)) : cell;
  };
  auto ref_set = [&](int x4, int y4, int ref) __attribute__((always_inline)) {          // the reference alone of an 8x8 block (vectors stay)
    const bool in = own && lx >= x4 && lx < x4 + 2 && ly >= y4 && ly < y4 + 2;
    cell = in ? ((cell & 0x0fffffffu) | ((uint32_t)(ref + 2) << 28)) : cell;
  };

  // per-reference registers with a run-time reference index (the loops over references stay rolled: ONE copy of the search code)
  auto sel32 = [&](const uint32_t (&a)[NR], int r) __attribute__((always_inline)) -> uint32_t { uint32_t v = a[0];
#pragma unroll
    for (int k = 1; k < NR; k++) v = r == k ? a[k] : v;
    return v; };
  auto put32 = [&](uint32_t (&a)[NR], int r, uint32_t v, bool at) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NR; k++) a[k] = (at && r == k) ? v : a[k]; };
  auto sel64 = [&](const unsigned long long (&a)[NR], int r) __attribute__((always_inline)) -> unsigned long long { unsigned long long v = a[0];
#pragma unroll
    for (int k = 1; k < NR; k++) v = r == k ? a[k] : v;
    return v; };
  auto put64 = [&](unsigned long long (&a)[NR], int r, unsigned long long v) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NR; k++) a[k] = r == k ? v : a[k]; };

  // PartitionMotionSearch (src/mv-search.c:1378) + list_prediction_cost (src/mode_decision.c:255) on records: every reference in turn, each
  // sub-partition's BlockMotionSearch (src/mv-search.c:560) answered from its record when the predictor just formed is the record's, its result
  // written into the field before the next one's predictor is formed; returns the cheapest reference's cost
  uint32_t ctr16[NR];                                                       // FastFullSearch: the window centre of each reference (its 16x16 predictor's)
#pragma unroll
  for (int r = 0; r < NR; r++) ctr16[r] = 0u;
  unsigned lists = 0u;                                                      // CHAIN: references whose missing records go to the work lists
  int budget = D.chain_budget;                                              // CHAIN: in-place searches left (a macroblock that misses everything would hold its sweep for 41 searches)
  unsigned long long inplace[NR];                                           // CHAIN: records computed in place (to write back)
#pragma unroll
  for (int r = 0; r < NR; r++) inplace[r] = 0ull;
  uint32_t rcu[NR];                                                         // (the costs as unsigned words, for the helpers above)
#pragma unroll
  for (int r = 0; r < NR; r++) rcu[r] = (uint32_t)rc[r];
  auto partition_search = [&](int bt, int block8, int *best_ref) __attribute__((always_inline)) -> int {
    const int base = bt == 1 ? 0 : bt == 2 ? 1 : bt == 3 ? 3 : bt == 4 ? 5 : bt == 5 ? 9 : bt == 6 ? 17 : 25;
    const int cnt = bt < 5 ? 1 : (bt < 7 ? 2 : 4);
    int best = INT_MAX;
    for (int r = 0; r < NR; r++) {
      int mc = 0;
      for (int k = 0; k < cnt; k++) {
        const int p = base + block8 * cnt + k;
        const uint32_t g = rl(dg, p), n = rl(dn, p);
        const uint32_t pk = predict(g, n, r), old = rl(sel32(pm, r), p);
        unsigned long long v = sel64(val, r), nd = sel64(need, r);
        const bool ok = old == pk && ((v >> p) & 1ull);
        if (FFS && p == 0) {
          const uint32_t c16 = x_centre(D, mv_x(pk), mv_y(pk));
          put32(ctr16, r, c16, true);
          if (old != pk && c16 != x_centre(D, mv_x(old), mv_y(old))) v = 0ull;      // a new window centre voids every record of the reference
        }
        int cost; uint32_t m;
        // CHAIN searches a missing record in place -- except FastFullSearch records of a reference whose window moved: all 41 are void then, and one
        // walk of the frame kernel (work lists, as in sweep 0) is cheaper than 41 scans
        const bool in_place = CHAIN && !(FFS && ((lists >> r) & 1u)) && budget > 0;
        if (FFS && CHAIN && p == 0 && !ok && v == 0ull) lists |= 1u << r;
        if (!ok && in_place && !(FFS && ((lists >> r) & 1u))) {
          budget--;
          const uint32_t ctr = FFS ? sel32(ctr16, r) : x_centre(D, mv_x(pk), mv_y(pk));
          cost = search_partition(r, p, g, mv_x(pk), mv_y(pk), mv_x(ctr), mv_y(ctr), &m);
          v |= 1ull << p;
          put64(inplace, r, sel64(inplace, r) | (1ull << p));
          put32(pm, r, pk, lane == p); put32(rm, r, m, lane == p); put32(rcu, r, (uint32_t)cost, lane == p);
        } else {
          if (!ok) { v &= ~(1ull << p); nd |= 1ull << p; put32(pm, r, pk, lane == p); }
          cost = (int)rl(sel32(rcu, r), p);
          m = rl(sel32(rm, r), p);
        }
        put64(val, r, v); put64(need, r, nd);
        if (p == 0) {
          if (!D.rdopt && skip_cost < cost) { cost = skip_cost; m = pk_mv(skx, sky); }      // the skip shortcut, src/mv-search.c:826-849 (every reference)
          put32(mv16, r, m, true);
        }
        mc += cost;
        field_rect((g >> 4) & 15, (g >> 8) & 15, (g >> 12) & 15, (g >> 16) & 15, r, m);
      }
      const int c = (r ? D.ref_cost1 : 0) + mc;
      if (c < best) { best = c; *best_ref = r; }
    }
    return best;
  };

  // ---- encode_one_macroblock_low, inter part (src/md_low.c:112-330), as ONE loop over its 21 partition searches: modes 1..3 (their blocks), then
  // for each 8x8 block the sub-modes 4..7. l0: the best reference per (mode, 8x8 block), 3 bits each; b8: the sub-mode per 8x8 block
  int best_mode = 1, min_cost = INT_MAX;
  unsigned long long l0 = 0ull;
  unsigned b8 = 0u;
  auto l0_set = [&](int mode, int k8, int r) __attribute__((always_inline)) { const int sh = (mode * 4 + k8) * 3; l0 = (l0 & ~(7ull << sh)) | ((unsigned long long)r << sh); };
  auto l0_get = [&](int mode, int k8) __attribute__((always_inline)) -> int { return (int)((l0 >> ((mode * 4 + k8) * 3)) & 7ull); };
  const bool any8 = D.valid[4] || D.valid[5] || D.valid[6] || D.valid[7];
  int acc = 0, mc8 = INT_MAX, bm = 0, bref = 0, cost8x8 = 0;
  for (int grp = 0; grp < 21; grp++) {
    const int mode = grp == 0 ? 1 : grp < 3 ? 2 : grp < 5 ? 3 : 4 + ((grp - 5) & 3);
    const int block = grp == 0 ? 0 : grp < 3 ? grp - 1 : grp < 5 ? grp - 3 : (grp - 5) >> 2;
    if (mode >= 4 && !any8) break;
    if (D.valid[mode]) {
      int best_ref = 0;
      int cost = partition_search(mode, block, &best_ref);
      if (mode < 4) {
        if (block == 0) acc = 0;
        acc += cost;
        if (mode == 1) { field_mode(0, 0, 4, 4, best_ref, 1); for (int k = 0; k < 4; k++) l0_set(1, k, best_ref); }
        else if (mode == 2) { l0_set(2, 2 * block, best_ref); l0_set(2, 2 * block + 1, best_ref); if (block == 0) field_mode(0, 0, 4, 2, best_ref, 2); }
        else { l0_set(3, block, best_ref); l0_set(3, block + 2, best_ref); if (block == 0) field_mode(0, 0, 2, 4, best_ref, 3); }
        if ((mode == 1 || block == 1) && acc < min_cost) { best_mode = mode; min_cost = acc; }
      } else {
        ref_set((block & 1) * 2, block & 2, best_ref);
        if (cost != INT_MAX) cost += ((D.lambda_mf[2] * (NR <= 1 ? 0 : x_refbits(mode - 4))) >> 16) - 1;
        if (cost < mc8) { mc8 = cost; bm = mode; bref = best_ref; }
      }
    }
    if (mode == 7) {                                                        // the 8x8 block is decided (src/mode_decision.c:531)
      cost8x8 += mc8;
      b8 = (b8 & ~(7u << (3 * block))) | ((unsigned)bm << (3 * block));
      l0_set(4, block, bref);
      field_mode((block & 1) * 2, block & 2, 2, 2, bref, bm);
      mc8 = INT_MAX; bm = 0; bref = 0;
      if (block == 3 && cost8x8 < min_cost) { best_mode = 8; min_cost = cost8x8; }
    }
  }
  // the final field of the macroblock
  for (int k8 = 0; k8 < 4; k8++)
    field_mode((k8 & 1) * 2, k8 & 2, 2, 2, l0_get(best_mode == 8 ? 4 : best_mode, k8), best_mode == 8 ? (int)((b8 >> (3 * k8)) & 7u) : best_mode);
#pragma unroll
  for (int r = 0; r < NR; r++) rc[r] = (int)rcu[r];

#ifdef JMHIP_X_PROF
  if (CHAIN && tid == 0) { atomicAdd(&g_xprof[8], __builtin_amdgcn_s_memtime() - xp_kernel_t0); atomicAdd(&g_xprof[9], 1ull); }
#endif
  if (CHAIN && wave != 0) return;                                           // wave 0 hands on (the other waves hold the same values)
  // ---- hand on: the sixteen field entries (and whether they changed), the decision, the new predictors and the work they ask for
  bool diff = false;
  if (own) {
    const size_t at = (size_t)(4 * mby + ly) * D.w4 + 4 * mbx + lx;
    const int8_t r = (int8_t)((int)(cell >> 28) - 2);
    const short vx = (short)cell_mvx(cell), vy = (short)cell_mvy(cell);
    diff = D.ref_idx[at] != r || D.mv[at * 2] != vx || D.mv[at * 2 + 1] != vy;
    D.ref_idx[at] = r; D.mv[at * 2] = vx; D.mv[at * 2 + 1] = vy;
    jmhip_mb_inter &o = D.out[addr];
    o.final_mv[ly * 4 + lx][0] = vx; o.final_mv[ly * 4 + lx][1] = vy;
  }
  const int changed = __ballot(diff) != 0ull;
  int any_need = CHAIN ? 0 : skip_need;
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const size_t j = (size_t)r * D.nmb + addr;
    const unsigned long long nd = need[r];
    if (lane == 0) { D.valid_rec[j] = val[r]; D.need_rec[j] = nd; }
    if (CHAIN && inplace[r]) {                                              // the records computed in this pass
      const bool done = lane < JMHIP_NPART && ((inplace[r] >> lane) & 1ull);
      if (done) { *reinterpret_cast<uint32_t *>(D.jobs[j].pred_mv[lane]) = pm[r]; *reinterpret_cast<uint32_t *>(D.res[j].mv[lane]) = rm[r]; D.res[j].cost[lane] = rc[r]; }
    }
    if (!nd) continue;
    const bool mine = lane < JMHIP_NPART && ((nd >> lane) & 1ull);
    if (mine) *reinterpret_cast<uint32_t *>(D.jobs[j].pred_mv[lane]) = pm[r];
    any_need = 1;
    const uint32_t pm0 = rl(pm[r], 0);
    const int uni = __ballot(lane < JMHIP_NPART && pm[r] != pm0) == 0ull;       // one predictor for all 41 partitions: the kernel's cheap mv-cost path
    const int sh = addr & (JM_SHARDS - 1);
    if (FFS) {
      if (lane == 0) {
        D.items[(size_t)sh * D.cap_items + atomicAdd(&D.cnt[sh * JM_SHARD_STRIDE], 1)] = (int)j | (uni << 30);
        D.sub_list[(size_t)sh * D.cap_sub + atomicAdd(&D.cnt[X_CNT_LIST + sh * JM_SHARD_STRIDE], 1)] = (int)j;
      }
    } else {
      // one item per distinct window centre among the needed partitions (the kernel writes every partition whose centre it is)
      const uint32_t ctr = mine ? x_centre(D, mv_x(pm[r]), mv_y(pm[r])) : 0u;
      bool lead = mine;
      for (int k = 0; k < JMHIP_NPART; k++) {
        const uint32_t ck = rl(ctr, k);
        if (k < lane && ((nd >> k) & 1ull) && ck == ctr) lead = false;
      }
      const unsigned long long lb = __ballot(lead);
      int base = 0;
      if (lane == 0) {
        base = atomicAdd(&D.cnt[sh * JM_SHARD_STRIDE], __popcll(lb));
        D.sub_list[(size_t)sh * D.cap_sub + atomicAdd(&D.cnt[X_CNT_LIST + sh * JM_SHARD_STRIDE], 1)] = (int)j;
      }
      base = __builtin_amdgcn_readfirstlane(base);
      if (lead) D.items[(size_t)sh * D.cap_items + base + __popcll(lb & ((1ull << lane) - 1ull))] = (int)j | (lane << 24) | (uni << 30);
    }
  }
  if (lane == 0) {
    const int sh = addr & (JM_SHARDS - 1);
    if (CHAIN) D.skip[addr] = sk;
    else if (skip_need) {
      XSkip n; n.mvx = (short)skx; n.mvy = (short)sky; n.cost = sk.cost; n.state = 0; n.pad = 0; D.skip[addr] = n;
      D.skip_list[(size_t)sh * D.cap_skip + atomicAdd(&D.cnt[2 * X_CNT_LIST + sh * JM_SHARD_STRIDE], 1)] = addr;
    }
    D.pending[addr] = (uint8_t)any_need;
    D.chg_next[addr] = (uint8_t)changed;
    // what the host looks at: flags, plain stores (thousands of macroblocks on one atomic counter would queue for longer than the replay takes)
    if (changed) D.cnt[X_CNT_FLAGS] = 1;
    if (any_need) D.cnt[X_CNT_FLAGS + 1] = 1;
    if (D.stats) { if (changed) atomicAdd(&D.cnt[X_CNT_FLAGS + 2], 1); if (any_need) atomicAdd(&D.cnt[X_CNT_FLAGS + 3], 1); atomicAdd(&D.cnt[X_CNT_FLAGS + 4], 1);
                   if (CHAIN) { int ns = 0; for (int r = 0; r < NR; r++) ns += __popcll(inplace[r]); atomicAdd(&D.cnt[X_CNT_FLAGS + 5], ns); } }
    jmhip_mb_inter &o = D.out[addr];
    o.best_mode = best_mode; o.min_cost = min_cost;
    for (int k = 0; k < 4; k++) { o.b8mode[k] = best_mode == 8 ? (int)((b8 >> (3 * k)) & 7u) : best_mode; o.b8ref[k] = l0_get(best_mode == 8 ? 4 : best_mode, k); }
    for (int k = 0; k < 4; k++) { o.p8mode[k] = (int)((b8 >> (3 * k)) & 7u); o.p8ref[k] = o.p8mode[k] ? l0_get(4, k) : 0; }      // the P8x8 candidate, whatever mode wins
    o.skip_mv[0] = (int16_t)skx; o.skip_mv[1] = (int16_t)sky;
    o.transform8x8_flag = 0; o.cbp8ts = -1;
  }
}

// GetSkipCostMB (src/mv-search.c:1136) for the macroblocks of the list: LumaPrediction of each 4x4 block at the skip vector from reference 0
// (block-origin clamp, explicit weights), distortion4x4 of the mode-decision metric (SAD, or HadamardSAD4x4 src/me_distortion.c:182)
__global__ __launch_bounds__(64) void x_skip_kernel(XDev D)
{
  const int *scnt = D.cnt + 2 * X_CNT_LIST;
  const int n = jm_shard_slots(scnt);                                       // virtual slots of the sharded list
  const int lane = threadIdx.x, sub = lane & 15, slot = lane >> 4;           // four macroblocks per wave, sixteen 4x4 blocks each
  const uint8_t *planes = D.ref_sub[D.ref_slot[0]];
  const size_t psz = (size_t)D.Wp * D.Hp;
  for (int i0 = blockIdx.x * 4; i0 < n; i0 += gridDim.x * 4) {
    const int i = i0 + slot;
    int v = 0, addr = i < n ? jm_shard_entry(scnt, D.skip_list, D.cap_skip, i) : -1;
    if (addr >= 0) {
      const XSkip sk = D.skip[addr];
      const int mbx = addr % D.mbw, mby = addr / D.mbw, bx = (sub & 3) * 4, by = (sub >> 2) * 4;
      const int cx = ((mbx * 16 + bx + JMHIP_PAD) << 2) + sk.mvx, cy = ((mby * 16 + by + JMHIP_PAD) << 2) + sk.mvy;
      const int ix = clampi(cx >> 2, 0, D.Wp - 17), iy = clampi(cy >> 2, 0, D.Hp - 17);
      const uint8_t *p = planes + psz * ((cy & 3) * 4 + (cx & 3)) + (size_t)iy * D.Wp + ix;
      int df[4][4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        uint32_t q, hi;
        fetch_row(p + (size_t)r * D.Wp, 4, &q, &hi);
        if (D.wp_pred) q = wp_apply4(q, D.wp_weight0, D.wp_offset0, D.wp_round, D.wp_denom);
        const uint32_t cv = *reinterpret_cast<const uint32_t *>(D.cur + (size_t)(mby * 16 + by + r) * D.W + mbx * 16 + bx);
#pragma unroll
        for (int c = 0; c < 4; c++) df[r][c] = (int)((cv >> (8 * c)) & 255u) - (int)((q >> (8 * c)) & 255u);
      }
      if (D.md_metric == 2) v = satd4x4(df);
      else {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int c = 0; c < 4; c++) v += iabs(df[r][c]);
      }
    }
    for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
    if (addr >= 0 && sub == 0) { D.skip[addr].cost = v; D.skip[addr].state = 1; }
  }
}

// the records of the slice's macroblocks in the ABI's layout (jmhip_mb_inter): what each BlockMotionSearch call returned
__global__ __launch_bounds__(64) void x_out_kernel(XDev D)
{
  const int lane = threadIdx.x, addr = D.mb_first + (int)blockIdx.x;
  jmhip_mb_inter &o = D.out[addr];
  const XSkip sk = D.skip[addr];
  const int skip_cost = sk.cost - ((D.lambda_mf[2] + 4096) >> 13);
  for (int r = 0; r < D.num_refs; r++) {
    const size_t j = (size_t)r * D.nmb + addr;
    if (lane < JMHIP_NPART) {
      const jmhip_me_result &q = D.res[j];
      int mvx = q.mv[lane][0], mvy = q.mv[lane][1], cost = q.cost[lane];
      if (lane == 0 && !D.rdopt && skip_cost < cost) { cost = skip_cost; mvx = sk.mvx; mvy = sk.mvy; }
      o.pred[r][lane][0] = D.jobs[j].pred_mv[lane][0]; o.pred[r][lane][1] = D.jobs[j].pred_mv[lane][1];
      o.mv_int[r][lane][0] = q.mv_int[lane][0]; o.mv_int[r][lane][1] = q.mv_int[lane][1]; o.cost_int[r][lane] = q.cost_int[lane];
      o.mv[r][lane][0] = (int16_t)mvx; o.mv[r][lane][1] = (int16_t)mvy; o.cost[r][lane] = cost;
    }
    if (lane < 4) {
      o.pred8ts[r][lane][0] = o.pred8ts[r][lane][1] = o.mv_int8ts[r][lane][0] = o.mv_int8ts[r][lane][1] = o.mv8ts[r][lane][0] = o.mv8ts[r][lane][1] = 0;
      o.cost_int8ts[r][lane] = o.cost8ts[r][lane] = 0;
    }
  }
}

// first call on a context: every job record knows its macroblock and reference slot
__global__ void x_jobs_init_kernel(XDev D)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D.num_refs * D.nmb) return;
  const int r = i / D.nmb, a = i - r * D.nmb;
  jmhip_me_mb &j = D.jobs[i];
  j.mb_x = (int16_t)(a % D.mbw); j.mb_y = (int16_t)(a / D.mbw); j.ref = (int16_t)D.ref_slot[r]; j.ref_is_0 = (int16_t)(r == 0);
}

template <bool FFS, bool CHAIN> void launch_sim_refs(int nr, int grid, size_t lds, hipStream_t st, const XDev &D)
{
  static_assert(XR == 5, "one instantiation per reference count");
  constexpr int NT = CHAIN ? 256 : 64;
  switch (nr) {
  case 1: x_sim_kernel<FFS, 1, CHAIN><<<grid, NT, lds, st>>>(D); break;
  case 2: x_sim_kernel<FFS, 2, CHAIN><<<grid, NT, lds, st>>>(D); break;
  case 3: x_sim_kernel<FFS, 3, CHAIN><<<grid, NT, lds, st>>>(D); break;
  case 4: x_sim_kernel<FFS, 4, CHAIN><<<grid, NT, lds, st>>>(D); break;
  default: x_sim_kernel<FFS, 5, CHAIN><<<grid, NT, lds, st>>>(D); break;
  }
}
void launch_sim(bool ffs, bool chain, int nr, int grid, hipStream_t st, const XDev &D)
{
  const int wside = 2 * (D.R + X_WIN_MARGIN) + 1 + 15;
  const size_t lds = chain ? (size_t)wside * ((wside + 3) & ~3) + 16 : 0;
  if (ffs) { if (chain) launch_sim_refs<true, true>(nr, grid, lds, st, D); else launch_sim_refs<true, false>(nr, grid, lds, st, D); }
  else { if (chain) launch_sim_refs<false, true>(nr, grid, lds, st, D); else launch_sim_refs<false, false>(nr, grid, lds, st, D); }
}

struct XState {
  jmhip_me_mb *jobs = nullptr; jmhip_me_result *res = nullptr;
  unsigned long long *valid_rec = nullptr, *need_rec = nullptr;
  XSkip *skip = nullptr;
  uint8_t *pending = nullptr, *chg[2] = {nullptr, nullptr};
  int *items = nullptr, *sub_list = nullptr, *skip_list = nullptr, *cnt = nullptr;
  MeDev *medev = nullptr;                        // the frame kernels' parameter block in device memory (their list form re-reads it per item)
  // the sweeps' flag words arrive in page-locked host memory one sweep behind the launches (no host wait between sweeps): two ints per sweep, an event each
  int *h_flags = nullptr; hipEvent_t evt[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int nmb = 0, refs = 0;
};

int x_cap(int nmb) { return (nmb + JM_SHARDS - 1) / JM_SHARDS; }      // macroblocks per shard

void x_release(XState *x)
{
  void *bufs[] = {x->jobs, x->res, x->valid_rec, x->need_rec, x->skip, x->pending, x->chg[0], x->chg[1], x->items, x->sub_list, x->skip_list, x->cnt, x->medev};
  for (void *b : bufs) if (b) (void)hipFree(b);
  if (x->h_flags) (void)hipHostFree(x->h_flags);
  for (hipEvent_t e : x->evt) if (e) (void)hipEventDestroy(e);
  delete x;
}

}  // namespace

void jm_xslice_free(jmhip_ctx *c)
{
  if (!c->xslice_state) return;
  x_release(static_cast<XState *>(c->xslice_state));
  c->xslice_state = nullptr;
}

// can the sweeps over the frame kernels take this slice? (jmhip_p_slice_search asks; the rest stays with me_wave.hip's kernels)
bool jm_xslice_covers(const jmhip_slice_params *p)
{
  if (const char *e = getenv("JMHIP_SLICE_X")) if (!atoi(e)) return false;
  return (p->search_mode == JMHIP_SEARCH_FULL || p->search_mode == JMHIP_SEARCH_FASTFULL) && p->search_range >= 16 && p->search_range <= 40 &&
         p->full_search == 2 && p->transform8x8_mode == 0 && p->metric[0] == 0 && p->metric[1] == 2 && p->metric[2] == 2 &&
         (p->md_metric == 0 || p->md_metric == 2) && p->num_refs <= XR;
}

// Runs the sweeps. *settled = 0 when they did not reach the fixpoint within the cap (the caller's own schedule then finishes from the field they
// left -- any state is a legal first guess); passes: sweeps run.
int jm_xslice_run(jmhip_ctx *c, const jmhip_slice_params *prm, int8_t *ref_idx, short *mv, jmhip_mb_inter *out, int *passes, int *settled)
{
  const int nmb = c->mbw * c->mbh, nr = prm->num_refs;
  *settled = 0;
  XState *x = static_cast<XState *>(c->xslice_state);
  if (x && (x->nmb != nmb || x->refs < nr)) { jm_xslice_free(c); x = nullptr; }
  bool fresh = false;
  if (!x) {
    x = new XState();
    x->nmb = nmb; x->refs = XR;
    const size_t nj = (size_t)XR * nmb;
    bool ok = hipMalloc((void **)&x->jobs, sizeof(jmhip_me_mb) * nj) == hipSuccess && hipMalloc((void **)&x->res, sizeof(jmhip_me_result) * nj) == hipSuccess &&
              hipMalloc((void **)&x->valid_rec, 8 * nj) == hipSuccess && hipMalloc((void **)&x->need_rec, 8 * nj) == hipSuccess &&
              hipMalloc((void **)&x->skip, sizeof(XSkip) * nmb) == hipSuccess && hipMalloc((void **)&x->pending, nmb) == hipSuccess &&
              hipMalloc((void **)&x->chg[0], nmb) == hipSuccess && hipMalloc((void **)&x->chg[1], nmb) == hipSuccess &&
              hipMalloc((void **)&x->items, sizeof(int) * (size_t)JM_SHARDS * x_cap(nmb) * XR * JMHIP_NPART) == hipSuccess &&
              hipMalloc((void **)&x->sub_list, sizeof(int) * (size_t)JM_SHARDS * x_cap(nmb) * XR) == hipSuccess &&
              hipMalloc((void **)&x->skip_list, sizeof(int) * (size_t)JM_SHARDS * x_cap(nmb)) == hipSuccess &&
              hipMalloc((void **)&x->cnt, sizeof(int) * (size_t)X_CNT_SWEEP * (X_MAX_SWEEPS + 1)) == hipSuccess &&
              hipMalloc((void **)&x->medev, sizeof(MeDev)) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&x->h_flags, sizeof(int) * 2 * (X_MAX_SWEEPS + 1), hipHostMallocDefault) == hipSuccess;
    for (int k = 0; ok && k < 8; k++) ok = hipEventCreateWithFlags(&x->evt[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) { x_release(x); return jm_fail(c, JMHIP_ERR_NOMEM, "record arrays of the exhaustive slice search"); }
    JM_HIP_CHECK(c, hipMemsetAsync(x->jobs, 0, sizeof(jmhip_me_mb) * nj, c->stream));
    JM_HIP_CHECK(c, hipMemsetAsync(x->res, 0, sizeof(jmhip_me_result) * nj, c->stream));
    JM_HIP_CHECK(c, hipMemsetAsync(x->skip, 0, sizeof(XSkip) * nmb, c->stream));
    JM_HIP_CHECK(c, hipMemsetAsync(x->pending, 0, nmb, c->stream));
    JM_HIP_CHECK(c, hipMemsetAsync(x->chg[0], 0, nmb, c->stream));
    JM_HIP_CHECK(c, hipMemsetAsync(x->chg[1], 0, nmb, c->stream));
    c->xslice_state = x;
    fresh = true;
  }
  (void)fresh;
  {
    static bool uploaded[64] = {false};
    const int dev = c->cfg.device;
    if (dev < 0 || dev >= 64 || !uploaded[dev]) {
      build_xparts();
      JM_HIP_CHECK(c, hipMemcpyToSymbol(HIP_SYMBOL(c_xpart), h_xpart, sizeof(h_xpart)));
      if (dev >= 0 && dev < 64) uploaded[dev] = true;
    }
  }
  int rc = jm_me_sub_tables(c);
  if (rc) return rc;
  MeDev P{};
  size_t plds = 0;
  P.mode = prm->search_mode; P.R = prm->search_range; P.rdopt = prm->rdopt; P.is_b = 0;
  P.lvl_min = prm->level_mv_min; P.lvl_max = prm->level_mv_max;
  P.lam_f = prm->lambda_mf[0]; P.lam_h = prm->lambda_mf[1]; P.lam_q = prm->lambda_mf[2];
  P.t8x8 = 0; P.subpel = 1;
  P.wp_on = prm->wp_me ? 1 : 0; P.wp_round = prm->wp_round; P.wp_denom = prm->wp_denom;
  for (int r = 0; r < nr; r++) { P.wp_w[prm->ref_slot[r]] = prm->wp_weight[r]; P.wp_o[prm->ref_slot[r]] = prm->wp_offset[r]; }
  P.mask = (1ull << JMHIP_NPART) - 1;
  P.W = c->W; P.H = c->H; P.Wp = c->Wp; P.Hp = c->Hp;
  P.cur = c->cur_y;
  P.ref_y = reinterpret_cast<const uint8_t *const *>(c->ref_ptrs_dev);
  P.ref_sub = reinterpret_cast<const uint8_t *const *>(c->ref_ptrs_dev) + 32;
  if ((rc = jm_me_pair_geometry(c, prm->search_range, &P, &plds))) return rc;
  JM_HIP_CHECK(c, hipMemcpyAsync(x->medev, &P, sizeof(MeDev), hipMemcpyHostToDevice, c->stream));
  JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));      // (P is a stack object)

  XDev D{};
  D.search_mode = prm->search_mode; D.R = prm->search_range; D.num_refs = nr; D.rdopt = prm->rdopt;
  for (int m = 0; m < 8; m++) D.valid[m] = prm->valid[m];
  for (int k = 0; k < 3; k++) D.lambda_mf[k] = prm->lambda_mf[k];
  D.ref_cost1 = prm->ref_cost1; D.md_metric = prm->md_metric; D.lvl_min = prm->level_mv_min; D.lvl_max = prm->level_mv_max;
  D.mb_first = prm->mb_first; D.mb_count = prm->mb_count; D.slice_mbs = prm->slice_mbs;
  D.wp_pred = prm->wp_pred; D.wp_round = prm->wp_round; D.wp_denom = prm->wp_denom; D.wp_weight0 = prm->wp_weight[0]; D.wp_offset0 = prm->wp_offset[0];
  for (int r = 0; r < nr; r++) { D.ref_slot[r] = prm->ref_slot[r]; D.wp_weight[r] = prm->wp_weight[r]; D.wp_offset[r] = prm->wp_offset[r]; }
  D.wp_me = prm->wp_me ? 1 : 0;
  D.W = c->W; D.H = c->H; D.Wp = c->Wp; D.Hp = c->Hp; D.mbw = c->mbw; D.mbh = c->mbh; D.w4 = c->W / 4; D.nmb = nmb;
  D.cur = c->cur_y; D.ref_sub = P.ref_sub;
  D.ref_idx = ref_idx; D.mv = mv; D.out = out;
  D.jobs = x->jobs; D.res = x->res; D.valid_rec = x->valid_rec; D.need_rec = x->need_rec; D.skip = x->skip; D.pending = x->pending;
  D.items = x->items; D.sub_list = x->sub_list; D.skip_list = x->skip_list; D.cnt = x->cnt;
  D.cap_items = x_cap(nmb) * XR * JMHIP_NPART; D.cap_sub = x_cap(nmb) * XR; D.cap_skip = x_cap(nmb);

  x_jobs_init_kernel<<<(nr * nmb + 255) / 256, 256, 0, c->stream>>>(D);       // (the reference slots of a list index may differ from call to call)
  const int cap = std::min(X_MAX_SWEEPS, getenv("JMHIP_SLICE_SWEEPS") ? atoi(getenv("JMHIP_SLICE_SWEEPS")) : 400);
  JM_HIP_CHECK(c, hipMemsetAsync(x->cnt, 0, sizeof(int) * (size_t)X_CNT_SWEEP * (cap + 1), c->stream));      // every sweep has its own counters: one clear per call
  const int check_every = getenv("JMHIP_SLICE_CHECK") ? std::max(1, atoi(getenv("JMHIP_SLICE_CHECK"))) : 1;
  const bool trace = getenv("JMHIP_SLICE_TRACE") != nullptr;
  const bool ffs = prm->search_mode == JMHIP_SEARCH_FASTFULL;
  int sweep = 0, quiet = 0;
  // FullSearch: five list sweeps, then the replay searches in place (measured on 1080p pictures: the tail of single macroblocks' chains is what the
  // list sweeps are slow at). FastFullSearch stays with the lists: its records die in whole references (the window moves) and its in-place
  // sweeps were slower than the list sweeps they replaced.
  // ... up to `chain_budget` records per macroblock and sweep: a macroblock that misses all 41 would hold its sweep for 41 serial searches (0.5 ms) while
  // the rest of the GPU idles; beyond the budget it asks through the work lists like an early sweep does, so the list kernels run in every sweep
  // (measured, 1080p bench clip: one reference 8.7 / 6.4 / 13.8 ms per picture without a budget, 8.1 / 6.1 / 15.6 with 10, 9.4 / 7.8 / 20.4 with 3 -- more, cheaper sweeps, no gain; two
  // references 17.1 / 18.5 ms without, 11.8 / 12.0 with 10, 10.3 / 10.5 with 3: a macroblock's chain is twice as long there)
  const int chain_budget = getenv("JMHIP_X_CHAIN_BUDGET") ? std::max(1, atoi(getenv("JMHIP_X_CHAIN_BUDGET"))) : (nr == 1 ? JMHIP_NPART : 4);
  const int chain_from = getenv("JMHIP_X_CHAIN_FROM") ? atoi(getenv("JMHIP_X_CHAIN_FROM")) : (ffs ? (1 << 30) : 5);
  for (; sweep < cap; sweep++) {
    int *cnt = x->cnt + (size_t)X_CNT_SWEEP * sweep;
    D.cnt = cnt; D.stats = trace; D.debug = getenv("JMHIP_X_DEBUG") ? atoi(getenv("JMHIP_X_DEBUG")) : 0;
    D.first_sweep = sweep == 0; D.chg_prev = x->chg[sweep & 1]; D.chg_next = x->chg[(sweep + 1) & 1];
    D.chain_budget = chain_budget;
    // The first sweeps ask for nearly every record of nearly every macroblock: the frame kernels compute them at their full rate over work lists.
    // Once few macroblocks still ask (chain_from), the replay computes what it misses in place and a sweep resolves a macroblock's whole chain;
    // what it still sends to the lists (FastFullSearch windows that moved) is a trickle: small grids.
    const bool chain = sweep >= chain_from;
    launch_sim(ffs, chain, nr, prm->mb_count, c->stream, D);
    {
      // list lengths are on the device: the grid is sized for the whole slice while it may still be wanted
      int big = jm_xcd_grid(std::min(nr * prm->mb_count * 2, 65536)), small = 1024;
      if (const char *e = getenv("JMHIP_X_GRID")) big = small = std::max(8, atoi(e) & ~7);        // experiments: few workgroups, many trips each
      const bool wide = !chain && sweep < 3;
      {
        jm_launch_me_pair_list(c, P, x->medev, plds, x->jobs, x->items, x->res, cnt, D.cap_items, wide ? big : small);
        jm_launch_me_sub_list(c, P, x->jobs, x->res, x->sub_list, x->need_rec, cnt + X_CNT_LIST, D.cap_sub, wide ? std::min(big, jm_xcd_grid(nr * prm->mb_count)) : small);
      }
      if (!chain && !prm->rdopt) x_skip_kernel<<<wide ? std::max(1, prm->mb_count / 4) : 256, 64, 0, c->stream>>>(D);
    }
    JM_HIP_CHECK(c, hipGetLastError());
    if (!trace) {
      // The sweep's two flag words (anything changed, anything needed) follow it into page-locked host memory; the host looks at the PREVIOUS sweep's
      // while this one runs, so the device never waits for the host between sweeps. A settled sweep is therefore followed by one sweep that finds nothing
      // to do (its lists are empty, no macroblock is due): ~30 us of empty launches instead of a host round trip per sweep.
      JM_HIP_CHECK(c, hipMemcpyAsync(x->h_flags + 2 * sweep, cnt + X_CNT_FLAGS, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
      JM_HIP_CHECK(c, hipEventRecord(x->evt[sweep & 7], c->stream));
      if (sweep >= 1) {
        JM_HIP_CHECK(c, hipEventSynchronize(x->evt[(sweep - 1) & 7]));
        if (x->h_flags[2 * (sweep - 1)] == 0 && x->h_flags[2 * (sweep - 1) + 1] == 0) { quiet = 1; break; }      // settled one sweep ago: `sweep` sweeps did the work
      }
      continue;
    }
    {
      int h[8];
      JM_HIP_CHECK(c, hipMemcpyAsync(h, cnt + X_CNT_FLAGS, sizeof(h), hipMemcpyDeviceToHost, c->stream));
      JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
      if (trace) {
        std::vector<int> hc(X_CNT_FLAGS);
        JM_HIP_CHECK(c, hipMemcpy(hc.data(), cnt, sizeof(int) * X_CNT_FLAGS, hipMemcpyDeviceToHost));
        int tot[3] = {0, 0, 0};
        for (int l = 0; l < 3; l++) for (int sh = 0; sh < JM_SHARDS; sh++) tot[l] += hc[l * X_CNT_LIST + sh * JM_SHARD_STRIDE];
        static double t_last = 0.0;
        struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
        const double t_now = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
        fprintf(stderr, "x sweep %d (%s): %d simulated, %d changed what they hand on, %d with needed records: %d search items, %d refinements, %d skip costs; %d records searched in place (%.3f ms since the previous line)\n",
                sweep, chain ? "chain" : "lists", h[4], h[2], h[3], tot[0], tot[1], tot[2], h[5], t_now - t_last);
        t_last = t_now;
      }
#ifdef JMHIP_X_PROF
      {
        unsigned long long hp[16], hn = 0, z[16] = {0}, zn = 0;
        (void)hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_xprof), sizeof(hp)); (void)hipMemcpyFromSymbol(&hn, HIP_SYMBOL(g_xprof_n), sizeof(hn));
        if (hn) fprintf(stderr, "  X PROF: %llu records: cycles per record: staging %.0f scan %.0f min %.0f subpel %.0f | %llu macroblocks, %.0f cycles each\n", hn, (double)hp[0] / hn, (double)hp[1] / hn, (double)hp[2] / hn, (double)hp[3] / hn, hp[9], hp[9] ? (double)hp[8] / hp[9] : 0.0);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_xprof), z, sizeof(z)); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_xprof_n), &zn, sizeof(zn));
      }
#endif
      if (h[0] == 0 && h[1] == 0) { quiet = 1; sweep++; break; }
    }
  }
  *passes = sweep;
  if (!quiet) return JMHIP_OK;
  x_out_kernel<<<prm->mb_count, 64, 0, c->stream>>>(D);
  JM_HIP_CHECK(c, hipGetLastError());
  *settled = 1;
  return JMHIP_OK;
}
