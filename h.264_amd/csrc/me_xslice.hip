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

#ifdef X_VAR_V2
#define XSYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)
#else
#define XSYNC() __syncthreads()
#endif

namespace {

constexpr int XR = JMHIP_SLICE_REFS;

struct XSkip { short mvx, mvy; int cost; int state; int pad; };      // state 1: cost is GetSkipCostMB at (mvx, mvy) on this picture

struct XDev {
  int search_mode, R, num_refs, rdopt;
  int valid[8];
  int lambda_mf[3], ref_cost1, md_metric;
  int lvl_min, lvl_max;
  int mb_first, mb_count, slice_mbs;
  int wp_pred, wp_round, wp_denom;
  short wp_weight0, wp_offset0;
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
  int *items, *sub_list, *skip_list;
  uint32_t *dbg;                                 // development aid (JMHIP_X_DUMP): macroblock mb_first's ring after every field write of its last replay
  int *cnt;                                      // [0] items [1] refinement jobs [2] skip jobs [3] macroblocks that changed [4] macroblocks with needs [5] simulated
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

struct XShared {
  uint32_t cell[32];
  uint32_t pm[XR][48], rm[XR][48];               // predictor / final vector of (reference, partition), int16 pairs
  int rc[XR][48];                                // final cost
  uint32_t mv16[XR];                             // the 16x16 vector after the skip shortcut (img->all_mv[..][ref][1])
};

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
__device__ __forceinline__ void x_predict(const uint32_t *cell, const XPart q, int ref, int *pmx, int *pmy)
{
  const uint32_t r = x_median_pred(cell[q.a()], cell[q.b()], q.cblk() ? 0u : cell[q.c()], cell[q.d()], q.dir(), ref);
  *pmx = mv_x(r); *pmy = mv_y(r);
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

// One wave per macroblock of the slice: replay encode_one_macroblock_low (src/md_low.c:46) on records. All lanes run the same control flow on
// values read from LDS (broadcast reads); the lanes are used for staging, for the sixteen cells of a field write, and for the work lists.
#ifdef X_VAR_V3
#define X_OPT __attribute__((optnone))
#else
#define X_OPT
#endif
template <bool FFS>
__global__ __launch_bounds__(64) X_OPT void x_sim_kernel(XDev D)
{
  __shared__ XShared S;
  __shared__ unsigned long long s_v[XR], s_n[XR];
  __shared__ int s_l0ref[5][4], s_b8m[4];        // best reference per mode and 8x8 block; sub-mode per 8x8 block
  const int lane = threadIdx.x;
  const int addr = D.mb_first + (int)blockIdx.x;
  const int mbx = addr % D.mbw, mby = addr / D.mbw;
  // ---- is there anything new for this macroblock? (the four neighbouring macroblocks' availability: wave-uniform, computed once)
  const bool aA = mb_avail(D, addr, mbx - 1, mby), aB = mb_avail(D, addr, mbx, mby - 1), aC = mb_avail(D, addr, mbx + 1, mby - 1), aD = mb_avail(D, addr, mbx - 1, mby - 1);
  bool active = D.first_sweep != 0 || D.pending[addr] != 0;
  if (!active) active = (aA && D.chg_prev[addr - 1]) || (aB && D.chg_prev[addr - D.mbw]) || (aC && D.chg_prev[addr - D.mbw + 1]) || (aD && D.chg_prev[addr - D.mbw - 1]);
  if (!active) { if (lane == 0) D.chg_next[addr] = 0; return; }
  const int nr = D.num_refs;

  // ---- stage: the ring of the motion field, the records of every reference
  if (lane < 30) {
    const int gy = lane / 6, gx = lane - gy * 6;
    const bool own = gy >= 1 && gx >= 1 && gx <= 4;                         // the macroblock's own blocks: nothing written yet
    const bool av = gy == 0 ? (gx == 0 ? aD : (gx == 5 ? aC : aB)) : (gx == 0 && aA);
    uint32_t v = own ? pk_cell(-1, 0, 0) : 0u;
    if (!own && av) {
      const size_t at = (size_t)(4 * mby - 1 + gy) * D.w4 + (4 * mbx - 1 + gx);
      v = pk_cell(D.ref_idx[at], D.mv[at * 2], D.mv[at * 2 + 1]);
    }
    S.cell[lane] = v;
  }
  for (int r = 0; r < nr; r++) {
    const size_t j = (size_t)r * D.nmb + addr;
    if (lane < JMHIP_NPART) {
      S.pm[r][lane] = *reinterpret_cast<const uint32_t *>(D.jobs[j].pred_mv[lane]);
      S.rm[r][lane] = *reinterpret_cast<const uint32_t *>(D.res[j].mv[lane]);
      S.rc[r][lane] = D.res[j].cost[lane];
    }
    if (lane == 0) {
      // the records asked for in the last sweep have been computed since: they are valid now
      s_v[r] = D.first_sweep ? 0ull : (D.valid_rec[j] | D.need_rec[j]);
      s_n[r] = 0ull;
    }
  }
  XSkip sk = D.skip[addr];
  if (D.first_sweep) sk.state = 0;
  XSYNC();

  // ---- the skip vector (FindSkipModeMotionVector, src/mv-search.c:1189): it depends on the ring only
  int skx = 0, sky = 0;
  {
    const uint32_t ca = S.cell[6], cb = S.cell[1];
    const int zl = (ca >> 28) == 0 || ca == pk_cell(0, 0, 0), za = (cb >> 28) == 0 || cb == pk_cell(0, 0, 0);
    if (!(za || zl)) x_predict(S.cell, c_xpart[0], 0, &skx, &sky);
  }
  int skip_need = 0, skip_cost = sk.cost;
  if (!D.rdopt && !(sk.state == 1 && sk.mvx == skx && sk.mvy == sky)) skip_need = 1;      // (the stale cost stays the guess)
  skip_cost -= (D.lambda_mf[2] + 4096) >> 13;

  int dbg_step = 0;
  // a field write: the cells of rectangle (x4, y4, w4, h4) take reference `ref` and, per cell, the vector block type `bt` left there
  auto field_set = [&](int x4, int y4, int w4, int h4, int ref, int bt) __attribute__((always_inline)) {
    XSYNC();
    if (lane < 16) {
      const int lx = lane & 3, ly = lane >> 2;
      if (lx >= x4 && lx < x4 + w4 && ly >= y4 && ly < y4 + h4) {
        const uint32_t m = bt == 1 ? S.mv16[ref] : S.rm[ref][part_of(bt, lx, ly)];
        S.cell[(ly + 1) * 6 + lx + 1] = pk_cell(ref, mv_x(m), mv_y(m));
      }
    }
    XSYNC();
    if (D.dbg && blockIdx.x == 0) {
      if (lane < 30) D.dbg[dbg_step * 40 + lane] = S.cell[lane];
      if (lane == 30) { D.dbg[dbg_step * 40 + 30] = (uint32_t)(x4 | (y4 << 4) | (w4 << 8) | (h4 << 12) | (ref << 16) | (bt << 20)); D.dbg[dbg_step * 40 + 31] = 0x5e7u; }
      dbg_step++;
    }
  };
  auto ref_set = [&](int x4, int y4, int ref) __attribute__((always_inline)) {          // the reference alone of an 8x8 block (vectors stay)
    XSYNC();
    if (lane < 16) {
      const int lx = lane & 3, ly = lane >> 2;
      if (lx >= x4 && lx < x4 + 2 && ly >= y4 && ly < y4 + 2) {
        uint32_t *c = &S.cell[(ly + 1) * 6 + lx + 1];
        *c = (*c & 0x0fffffffu) | ((uint32_t)(ref + 2) << 28);
      }
    }
    XSYNC();
  };

  // BlockMotionSearch (src/mv-search.c:560) on records: returns the cost, leaves the vector in S.rm / S.mv16
  auto block_search = [&](int r, int p) __attribute__((always_inline)) -> int {
    const XPart q = c_xpart[p];
    int pmx, pmy;
    x_predict(S.cell, q, r, &pmx, &pmy);
    const uint32_t pk = pk_mv(pmx, pmy), old = S.pm[r][p];
    unsigned long long v = s_v[r];
    if (D.dbg && blockIdx.x == 0) {
      if (lane < 30) D.dbg[dbg_step * 40 + lane] = S.cell[lane];
      if (lane == 30) { D.dbg[dbg_step * 40 + 30] = (uint32_t)p | ((uint32_t)r << 8); D.dbg[dbg_step * 40 + 31] = 0xb5u; D.dbg[dbg_step * 40 + 32] = pk; D.dbg[dbg_step * 40 + 33] = old; D.dbg[dbg_step * 40 + 34] = S.rm[r][p]; }
      dbg_step++;
    }
#ifdef JMHIP_X_DEBUG
    if (addr == JMHIP_X_DEBUG && lane == 0)
      printf("mb %d ref %d p %d: cells a %08x b %08x c %08x d %08x cblk %d -> pred (%d,%d) old (%d,%d) valid %d  rec mv (%d,%d) cost %d\n", addr, r, p, S.cell[q.a()], S.cell[q.b()], S.cell[q.c()], S.cell[q.d()], q.cblk(),
             pmx, pmy, mv_x(old), mv_y(old), (int)((v >> p) & 1ull), mv_x(S.rm[r][p]), mv_y(S.rm[r][p]), S.rc[r][p]);
#endif
    bool ok = old == pk && ((v >> p) & 1ull);
    if (FFS && p == 0 && old != pk && x_centre(D, pmx, pmy) != x_centre(D, mv_x(old), mv_y(old))) { v = 0ull; }   // a new window centre voids every record of the reference
    if (!ok) {
      v &= ~(1ull << p);
      if (lane == 0) { S.pm[r][p] = pk; s_n[r] |= 1ull << p; }
    }
    if (lane == 0) s_v[r] = v;
    int cost = S.rc[r][p];
    if (p == 0) {
      uint32_t m = S.rm[r][0];
      if (!D.rdopt && skip_cost < cost) { cost = skip_cost; m = pk_mv(skx, sky); }          // the skip shortcut, src/mv-search.c:826-849 (every reference)
      if (lane == 0) S.mv16[r] = m;
    }
    XSYNC();
    return cost;
  };
  // PartitionMotionSearch (src/mv-search.c:1378) + list_prediction_cost (src/mode_decision.c:255): every reference in turn, each sub-partition's
  // result written into the field before the next one's predictor is formed; returns the cheapest reference's cost
  auto partition_search = [&](int bt, int block8, int *best_ref) __attribute__((always_inline)) -> int {
    const int base = bt == 1 ? 0 : bt == 2 ? 1 : bt == 3 ? 3 : bt == 4 ? 5 : bt == 5 ? 9 : bt == 6 ? 17 : 25;
    const int cnt = bt < 5 ? 1 : (bt < 7 ? 2 : 4);
    int best = INT_MAX;
    for (int r = 0; r < nr; r++) {
      int mc = 0;
      for (int k = 0; k < cnt; k++) {
        const int p = base + block8 * cnt + k;
        mc += block_search(r, p);
        const XPart q = c_xpart[p];
        field_set(q.x4(), q.y4(), q.w4(), q.h4(), r, bt);
      }
      const int c = (r ? D.ref_cost1 : 0) + mc;
      if (c < best) { best = c; *best_ref = r; }
    }
    return best;
  };

  // ---- encode_one_macroblock_low, inter part (src/md_low.c:112-330)
  int best_mode = 1, min_cost = INT_MAX;
  for (int mode = 1; mode < 4; mode++) {
    if (!D.valid[mode]) continue;
    int cost = 0;
    for (int block = 0; block < (mode == 1 ? 1 : 2); block++) {
      int best_ref = 0;
      cost += partition_search(mode, block, &best_ref);
      if (mode == 1) {
        field_set(0, 0, 4, 4, best_ref, 1);
        if (lane < 4) s_l0ref[1][lane] = best_ref;
      } else if (mode == 2) { if (lane < 2) s_l0ref[2][2 * block + lane] = best_ref; if (block == 0) field_set(0, 0, 4, 2, best_ref, 2); }
      else { if (lane < 2) s_l0ref[3][block + 2 * lane] = best_ref; if (block == 0) field_set(0, 0, 2, 4, best_ref, 3); }
    }
    if (cost < min_cost) { best_mode = mode; min_cost = cost; }
  }
  if (D.valid[4] || D.valid[5] || D.valid[6] || D.valid[7]) {
    int cost8x8 = 0;
    for (int block = 0; block < 4; block++) {
      int mc8 = INT_MAX, bm = 0, bref = 0;
      const int y0 = block & 2, x0 = (block & 1) * 2;
      for (int mode = 4; mode < 8; mode++) {
        if (!D.valid[mode]) continue;
        int best_ref = 0;
        int cost = partition_search(mode, block, &best_ref);
        ref_set(x0, y0, best_ref);
        if (cost != INT_MAX) cost += ((D.lambda_mf[2] * (nr <= 1 ? 0 : x_refbits(mode - 4))) >> 16) - 1;
        if (cost < mc8) { mc8 = cost; bm = mode; bref = best_ref; }
      }
      cost8x8 += mc8;
      if (lane == 0) { s_b8m[block] = bm; s_l0ref[4][block] = bref; }
      field_set(x0, y0, 2, 2, bref, bm);
    }
    if (cost8x8 < min_cost) { best_mode = 8; min_cost = cost8x8; }
  }
  XSYNC();
  // the final field of the macroblock
  for (int k8 = 0; k8 < 4; k8++)
    field_set((k8 & 1) * 2, k8 & 2, 2, 2, s_l0ref[best_mode == 8 ? 4 : best_mode][k8], best_mode == 8 ? s_b8m[k8] : best_mode);

  // ---- hand on: the sixteen field entries (and whether they changed), the decision, the new predictors and the work they ask for
  bool diff = false;
  if (lane < 16) {
    const int bx = lane & 3, by = lane >> 2;
    const size_t at = (size_t)(4 * mby + by) * D.w4 + 4 * mbx + bx;
    const uint32_t c = S.cell[(by + 1) * 6 + bx + 1];
    const int8_t r = (int8_t)((int)(c >> 28) - 2);
    const short vx = (short)cell_mvx(c), vy = (short)cell_mvy(c);
    diff = D.ref_idx[at] != r || D.mv[at * 2] != vx || D.mv[at * 2 + 1] != vy;
    D.ref_idx[at] = r; D.mv[at * 2] = vx; D.mv[at * 2 + 1] = vy;
    jmhip_mb_inter &o = D.out[addr];
    o.final_mv[lane][0] = vx; o.final_mv[lane][1] = vy;
  }
  const int changed = __ballot(diff) != 0ull;
  int any_need = skip_need;
  for (int r = 0; r < nr; r++) {
    const size_t j = (size_t)r * D.nmb + addr;
    const unsigned long long need = s_n[r];
    if (lane == 0) { D.valid_rec[j] = s_v[r]; D.need_rec[j] = need; }
    if (!need) continue;
    any_need = 1;
    const bool mine = lane < JMHIP_NPART && ((need >> lane) & 1ull);
    if (mine) *reinterpret_cast<uint32_t *>(D.jobs[j].pred_mv[lane]) = S.pm[r][lane];
    const uint32_t my_pm = lane < JMHIP_NPART ? S.pm[r][lane] : S.pm[r][0];
    const int uni = __ballot(my_pm != S.pm[r][0]) == 0ull;              // one predictor for all 41 partitions: the kernel's cheap mv-cost path
    if (FFS) {
      if (lane == 0) { D.items[atomicAdd(&D.cnt[0], 1)] = (int)j | (uni << 30); D.sub_list[atomicAdd(&D.cnt[1], 1)] = (int)j; }
    } else {
      // one item per distinct window centre among the needed partitions (the kernel writes every partition whose centre it is)
      const uint32_t ctr = mine ? x_centre(D, mv_x(my_pm), mv_y(my_pm)) : 0u;
      bool lead = mine;
      for (int k = 0; k < JMHIP_NPART; k++) {
        const uint32_t ck = (uint32_t)__shfl((int)ctr, k);
        if (k < lane && ((need >> k) & 1ull) && ck == ctr) lead = false;
      }
      const unsigned long long lb = __ballot(lead);
      int base = 0;
      if (lane == 0) { base = atomicAdd(&D.cnt[0], __popcll(lb)); D.sub_list[atomicAdd(&D.cnt[1], 1)] = (int)j; }
      base = __shfl(base, 0);
      if (lead) D.items[base + __popcll(lb & ((1ull << lane) - 1ull))] = (int)j | (lane << 24) | (uni << 30);
    }
  }
  if (lane == 0) {
    if (skip_need) { XSkip n; n.mvx = (short)skx; n.mvy = (short)sky; n.cost = sk.cost; n.state = 0; n.pad = 0; D.skip[addr] = n; D.skip_list[atomicAdd(&D.cnt[2], 1)] = addr; }
    D.pending[addr] = (uint8_t)any_need;
    D.chg_next[addr] = (uint8_t)changed;
    if (changed) atomicAdd(&D.cnt[3], 1);
    if (any_need) atomicAdd(&D.cnt[4], 1);
    atomicAdd(&D.cnt[5], 1);
    jmhip_mb_inter &o = D.out[addr];
    o.best_mode = best_mode; o.min_cost = min_cost;
#pragma unroll
    for (int k = 0; k < 4; k++) { o.b8mode[k] = best_mode == 8 ? s_b8m[k] : best_mode; o.b8ref[k] = s_l0ref[best_mode == 8 ? 4 : best_mode][k]; }
    o.skip_mv[0] = (int16_t)skx; o.skip_mv[1] = (int16_t)sky;
    o.transform8x8_flag = 0; o.cbp8ts = -1;
  }
}

// GetSkipCostMB (src/mv-search.c:1136) for the macroblocks of the list: LumaPrediction of each 4x4 block at the skip vector from reference 0
// (block-origin clamp, explicit weights), distortion4x4 of the mode-decision metric (SAD, or HadamardSAD4x4 src/me_distortion.c:182)
__global__ __launch_bounds__(64) void x_skip_kernel(XDev D)
{
  const int n = D.cnt[2];
  const int lane = threadIdx.x, sub = lane & 15, slot = lane >> 4;           // four macroblocks per wave, sixteen 4x4 blocks each
  const uint8_t *planes = D.ref_sub[D.ref_slot[0]];
  const size_t psz = (size_t)D.Wp * D.Hp;
  for (int i0 = blockIdx.x * 4; i0 < n; i0 += gridDim.x * 4) {
    const int i = i0 + slot;
    int v = 0, addr = 0;
    if (i < n) {
      addr = D.skip_list[i];
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
    if (i < n && sub == 0) { D.skip[addr].cost = v; D.skip[addr].state = 1; }
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

struct XState {
  jmhip_me_mb *jobs = nullptr; jmhip_me_result *res = nullptr;
  unsigned long long *valid_rec = nullptr, *need_rec = nullptr;
  XSkip *skip = nullptr;
  uint8_t *pending = nullptr, *chg[2] = {nullptr, nullptr};
  int *items = nullptr, *sub_list = nullptr, *skip_list = nullptr, *cnt = nullptr;
  int nmb = 0, refs = 0;
};

void x_release(XState *x)
{
  void *bufs[] = {x->jobs, x->res, x->valid_rec, x->need_rec, x->skip, x->pending, x->chg[0], x->chg[1], x->items, x->sub_list, x->skip_list, x->cnt};
  for (void *b : bufs) if (b) (void)hipFree(b);
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
              hipMalloc((void **)&x->items, sizeof(int) * nj * JMHIP_NPART) == hipSuccess && hipMalloc((void **)&x->sub_list, sizeof(int) * nj) == hipSuccess &&
              hipMalloc((void **)&x->skip_list, sizeof(int) * nmb) == hipSuccess && hipMalloc((void **)&x->cnt, sizeof(int) * 8) == hipSuccess;
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

  XDev D{};
  D.search_mode = prm->search_mode; D.R = prm->search_range; D.num_refs = nr; D.rdopt = prm->rdopt;
  for (int m = 0; m < 8; m++) D.valid[m] = prm->valid[m];
  for (int k = 0; k < 3; k++) D.lambda_mf[k] = prm->lambda_mf[k];
  D.ref_cost1 = prm->ref_cost1; D.md_metric = prm->md_metric; D.lvl_min = prm->level_mv_min; D.lvl_max = prm->level_mv_max;
  D.mb_first = prm->mb_first; D.mb_count = prm->mb_count; D.slice_mbs = prm->slice_mbs;
  D.wp_pred = prm->wp_pred; D.wp_round = prm->wp_round; D.wp_denom = prm->wp_denom; D.wp_weight0 = prm->wp_weight[0]; D.wp_offset0 = prm->wp_offset[0];
  for (int r = 0; r < nr; r++) D.ref_slot[r] = prm->ref_slot[r];
  D.W = c->W; D.H = c->H; D.Wp = c->Wp; D.Hp = c->Hp; D.mbw = c->mbw; D.mbh = c->mbh; D.w4 = c->W / 4; D.nmb = nmb;
  D.cur = c->cur_y; D.ref_sub = P.ref_sub;
  D.ref_idx = ref_idx; D.mv = mv; D.out = out;
  D.jobs = x->jobs; D.res = x->res; D.valid_rec = x->valid_rec; D.need_rec = x->need_rec; D.skip = x->skip; D.pending = x->pending;
  D.items = x->items; D.sub_list = x->sub_list; D.skip_list = x->skip_list; D.cnt = x->cnt;
  static uint32_t *dbg_dev = nullptr;
  if (getenv("JMHIP_X_DUMP") && !dbg_dev) { (void)hipMalloc((void **)&dbg_dev, 400 * 40 * 4); (void)hipMemset(dbg_dev, 0, 400 * 40 * 4); }
  D.dbg = getenv("JMHIP_X_DUMP") ? dbg_dev : nullptr;

  x_jobs_init_kernel<<<(nr * nmb + 255) / 256, 256, 0, c->stream>>>(D);       // (the reference slots of a list index may differ from call to call)
  const int cap = getenv("JMHIP_SLICE_SWEEPS") ? atoi(getenv("JMHIP_SLICE_SWEEPS")) : 400;
  const int check_every = getenv("JMHIP_SLICE_CHECK") ? std::max(1, atoi(getenv("JMHIP_SLICE_CHECK"))) : 4;
  const bool trace = getenv("JMHIP_SLICE_TRACE") != nullptr;
  const bool ffs = prm->search_mode == JMHIP_SEARCH_FASTFULL;
  int sweep = 0, quiet = 0;
  for (; sweep < cap; sweep++) {
    JM_HIP_CHECK(c, hipMemsetAsync(x->cnt, 0, sizeof(int) * 8, c->stream));
    D.first_sweep = sweep == 0; D.chg_prev = x->chg[sweep & 1]; D.chg_next = x->chg[(sweep + 1) & 1];
    if (ffs) x_sim_kernel<true><<<prm->mb_count, 64, 0, c->stream>>>(D); else x_sim_kernel<false><<<prm->mb_count, 64, 0, c->stream>>>(D);
    // list lengths are on the device: the first sweeps are sized for the whole slice, the tail for a handful of macroblocks
    int big = jm_xcd_grid(std::min(nr * prm->mb_count * 2, 65536)), small = 1024;
    if (const char *e = getenv("JMHIP_X_GRID")) big = small = std::max(8, atoi(e) & ~7);        // experiments: few workgroups, many trips each
    const int grid = sweep < 2 ? big : small;
    const bool dbg_sync = getenv("JMHIP_X_SYNC") != nullptr;
    if (dbg_sync) JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    jm_launch_me_pair_list(c, P, plds, x->jobs, x->items, x->res, x->cnt + 0, grid);
    if (dbg_sync) JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    jm_launch_me_sub_list(c, P, x->jobs, x->res, x->sub_list, x->need_rec, x->cnt + 1, sweep < 2 ? std::min(big, jm_xcd_grid(nr * prm->mb_count)) : small);
    if (dbg_sync) JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (!prm->rdopt) x_skip_kernel<<<sweep < 2 ? std::max(1, prm->mb_count / 4) : 256, 64, 0, c->stream>>>(D);
    JM_HIP_CHECK(c, hipGetLastError());
    if (const char *dp = getenv("JMHIP_X_DUMP")) {          // development aid: the job records (predictors) after sweep JMHIP_X_DUMP_SWEEP, raw
      const int ds = getenv("JMHIP_X_DUMP_SWEEP") ? atoi(getenv("JMHIP_X_DUMP_SWEEP")) : 1;
      if (sweep == ds) {
        std::vector<jmhip_me_mb> hj((size_t)nr * nmb);
        JM_HIP_CHECK(c, hipMemcpyAsync(hj.data(), x->jobs, sizeof(jmhip_me_mb) * hj.size(), hipMemcpyDeviceToHost, c->stream));
        JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
        if (FILE *f = fopen(dp, "wb")) { fwrite(hj.data(), sizeof(jmhip_me_mb), hj.size(), f); fclose(f); }
      }
    }
    if (trace || sweep < 2 || ((sweep - 1) % check_every) == 0) {
      int h[8];
      JM_HIP_CHECK(c, hipMemcpyAsync(h, x->cnt, sizeof(h), hipMemcpyDeviceToHost, c->stream));
      JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
      if (trace) {
        static double t_last = 0.0;
        struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
        const double t_now = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
        fprintf(stderr, "x sweep %d: %d simulated, %d changed what they hand on, %d with needed records: %d search items, %d refinements, %d skip costs (%.3f ms since the previous line)\n",
                sweep, h[5], h[3], h[4], h[0], h[1], h[2], t_now - t_last);
        t_last = t_now;
      }
      if (h[3] == 0 && h[4] == 0) { quiet = 1; sweep++; break; }
    }
  }
  *passes = sweep;
  if (D.dbg) {
    std::vector<uint32_t> h(400 * 40);
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemcpy(h.data(), D.dbg, h.size() * 4, hipMemcpyDeviceToHost);
    std::string path = std::string(getenv("JMHIP_X_DUMP")) + ".steps";
    if (FILE *f = fopen(path.c_str(), "wb")) { fwrite(h.data(), 4, h.size(), f); fclose(f); }
  }
  if (!quiet) return JMHIP_OK;
  x_out_kernel<<<prm->mb_count, 64, 0, c->stream>>>(D);
  JM_HIP_CHECK(c, hipGetLastError());
  *settled = 1;
  return JMHIP_OK;
}
