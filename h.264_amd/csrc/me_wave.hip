// me_wave.hip -- a whole P slice on the device: motion-vector prediction, the integer search of the configured SearchMode (FullSearch,
// FastFullSearch, UMHexagonS, EPZS), sub-pel refinement, the skip shortcut and the low-complexity (rdopt = 0) inter decision, macroblocks
// running as a 2:1 wavefront. Entry: jmhip_p_slice_search (include/jmhip.h lists the JM functions this replaces, file:line).
//
// One workgroup = one wave = one macroblock ROW. The wave walks its row left to right; before macroblock x it waits until the row above has
// finished macroblock x + 1 (progress counters in global memory, agent-scope release / acquire). That order keeps every dependency JM's raster
// order has on NEIGHBOURS: A/B/C/D vectors and references, the EPZS row memories, the UMHexagonS cost maps.
// Inside the wave the control flow is the scalar algorithm, executed redundantly by all 64 lanes on LDS scratch (every lane writes the same
// values); what is parallel is the evaluation of a BATCH of candidates (eval_dist): JM's early exits only skip work -- a distortion that is cut
// short is never accepted (callers test strict <) -- so every group of candidates whose positions do not depend on each other's outcome (the
// predictor set, one pass of a refinement pattern, a hexagon ring, the cross, the 5 / 9 sub-pel points) is evaluated in full across the lanes
// and the sequential accept / reject logic is replayed on the results.
#include "me_common.h"
#include <time.h>

namespace {

constexpr int WR = JMHIP_SLICE_REFS;
constexpr int MAXC = 128;                      // candidates per batch
constexpr int CARRY = 5;
constexpr uint8_t EP_STAMPED = 255;             // WaveDev.ep_first: the macroblock's first touch of the cell was a search stamping its centre (:1598), not a test
constexpr int SURF_PLANES = 20;                // the sixteen 4x4 blocks + the four 8x8 blocks of a macroblock
constexpr int SURF_MARGIN = 6;                 // FullSearch: the surface is built round the 16x16 centre, this much wider than the range
constexpr int WIN_MAX = 96;                    // LDS reference window side: 2 * (33 + SURF_MARGIN) + 1 + 15 = 94                       // vectors of the previous macroblock EPZS may read per reference: 16x16 + four 8x8

struct WaveDev {
  jmhip_slice_params p;
  int W, H, Wp, Hp, mbw, mbh, w4, h4;
  const uint8_t *cur;
  const uint8_t *const *ref_sub;               // [slot] -> 16 quarter-pel planes
  int8_t *ref_idx; short *mv;                  // enc_picture->ref_idx[LIST_0] [h4][w4], ->mv[LIST_0] [h4][w4][2]
  int *prog;                                   // [mbh]: macroblocks finished in the row (x + 1)
  int *flags;                                  // [0] abort (a wait timed out)
  jmhip_mb_inter *out;                         // by macroblock address
  // EPZS row memories, one stored row per macroblock row of the slice (+ row 0: the state the slice started from): a macroblock reads, per
  // 4x4 column, the row of the macroblock that wrote it last in coding order (state_row) -- which makes any schedule that respects, or
  // iterates towards, the coding order see what JM's single-row arrays would hold
  int *ep_dist;                                // EPZSDistortion[LIST_0] [rows + 1][7][w4]
  short *ep_motion;                            // EPZSMotion[LIST_0] [rows + 1][WR][7][4][w4][2]
  int row0;                                    // first macroblock row of the slice
  const short *ep_col;                         // EPZSCo_located->mv[LIST_0] [h4][w4][2]
  // EPZSMap / EPZSBlkCount (me_epzs.c:49,92,1550,1598,1757,1840): JM never clears the map, a cell counts as visited when it holds the 16-bit
  // stamp of the running search -- also when a search 65536 k calls earlier left that stamp, or the counter passes the zero the map was
  // allocated with. A macroblock keeps a fresh bitmap per search and records, per map cell, the first and the last of ITS searches that
  // tested the cell; a scan over the macroblocks in coding order (ep_alias_* kernels below) finds the tests JM would have answered from an old
  // stamp, and the macroblocks they fall in are searched again with those cells pre-marked (ep_alias)
  uint8_t *ep_first, *ep_last;                 // [nmb][ep_cells]: search of the macroblock (1-based, 0 = none) that tested the cell first (EP_STAMPED: it was stamped before any test) / touched it last
  uint16_t *ep_nsearch;                        // [nmb]: integer searches of the macroblock
  int ep_cells;                                // (2 search_range + 1)^2, padded to a multiple of 4
  const unsigned long long *ep_alias; int ep_alias_n;      // macroblock << 32 | search << 16 | cell: tests answered "visited" from an old stamp
  uint8_t *ep_alias_flag;                      // [nmb]: bit 0 the macroblock has entries in ep_alias, bit 1 search it again in the next sweep
  const short *carry_in;                       // [mbh][WR][CARRY][2]: img->all_mv the first macroblock of a row finds (speculated)
  short *carry_out;                            // [mbh][WR][CARRY][2]: what the last macroblock of a row leaves
  int *um_cost;                                // fastme_l0_cost [8][h4][w4]
  const int *um_in;                            // ... as the slice found it (what a macroblock's own entries hold until it writes them)
  short *carry_mb;                             // [nmb][WR][CARRY][2]: img->all_mv entries every macroblock leaves for the next one in coding order
  // relaxation schedule (p_slice_relax_kernel): which macroblocks changed what they hand on, last sweep / this sweep
  const uint8_t *chg_prev; uint8_t *chg_next; int *n_changed; int first_sweep;
  int reduced;                                 // a first sweep that only searches the 16x16 mode: a cheap first guess of the field (every macroblock is evaluated in full by the sweep after it)
  // ... and, exhaustive searches: which (reference, partition) call records a macroblock's last evaluation in THIS call wrote. A call whose
  // predictor equals the recorded one is a pure function of it (FastFull: and of the reference's 16x16 predictor, the window centre), so a
  // re-evaluation takes the record instead of searching again -- what is recomputed in later sweeps is only what actually changed
  unsigned long long *memo; int memo_on;
  const uint16_t *tie_tab; int tie_R;           // exhaustive searches: spiral index + 1 of every offset of a (2 tie_R + 1)^2 window, row-major (built per call)
  uint16_t *surf;                              // exhaustive searches: per row, per reference, SAD surfaces [SURF_PLANES][surf_n]
  int surf_n;                                  // candidates per plane (capacity)
  int debug;                                   // JMHIP_WAVE_DEBUG (timing experiments only; results are wrong): 1 no sub-pel, 2 no integer search, 4 no skip cost, 16 no EPZS map records
};

// The slice's parameter block lives in constant memory and the block being searched in LDS, both named directly by every device function:
// passed by reference to functions that are not inlined they become generic pointers into scratch and every field read a FLAT instruction
// (853 of them before), which is where a macroblock's 2.6 M cycles went. One slice search per device at a time (the entry point synchronises).
__constant__ WaveDev c_wave;
#define D c_wave

struct Lds {
  uint8_t cur[16][16];
  int cx[MAXC], cy[MAXC], dist[MAXC];          // evaluation batch: padded quarter-pel block origins -> distortions
  int qx[MAXC], qy[MAXC];                      // the vectors the batch entries stand for
  int px[MAXC], py[MAXC];                      // EPZS predictor list
  uint32_t map[160];                           // visited map of one search, (2R+1)^2 bits
  short all_mv[16][WR][8][2];                  // img->all_mv[by][bx][LIST_0][ref][blocktype]
  int motion_cost[8][WR][4];
  int surf_c[2 * WR][5];                       // per surface slot (0: the macroblock's, 1: a partition's own) and reference: centre (pels), half side, valid, block rows
  // the macroblock's view of the picture-level state: staged once by mb_stage, used and updated in LDS, handed on by mb_commit.
  // Grid [y + 1][x + 1] of the 4x4 blocks x = -1..4, y = -1..3: the macroblock's own sixteen and the ring its predictors read (A, B, C, D)
  // the inter decision's running state (macroblock_low): in LDS, not in registers or private arrays -- the decision loop is wrapped round the whole search
  // code, where register-resident state was both expensive (live across 6 k instructions) and, with dynamic indices, scratch
  struct Dec { int best_mode, min_cost, t8_flag, best_tflag, cbp8ts, tr8_cost, tr4_cost, cost, cost8x8, mc8; int l0ref[5][4], b8m[4], p8m[4], p8r[4], ref8ts[4]; short mv8ts[4][2]; } dec;
  int mbx, mby;
  int pass8ts;                                 // inside the 8x8-transform P8x8 pass (Transform8x8Mode): the call records go to the *8ts arrays
  int memo_live, ff_same[WR];                  // this evaluation may reuse call records (WaveDev.memo); FastFull: the reference's window centre is the recorded one
  unsigned long long memo_old[WR], memo_new[WR];
  int8_t f_ref[5][6];                          // enc_picture->ref_idx[LIST_0]
  short f_mv[5][6][2];                         // enc_picture->mv[LIST_0]
  // what only one family of search modes uses shares its space: with all of it side by side the block was 22 KB and the eighth workgroup of
  // a CU (2 waves per SIMD) no longer fitted -- 2048 resident workgroups became 1792 and a full sweep two rounds of workgroups instead of one
  union {
    __attribute__((aligned(16))) uint8_t win[WIN_MAX * WIN_MAX];   // exhaustive modes: reference window of the surface pass
    struct {                                                        // EPZS
      short ep_mot[WR][7][4][12][2];             // EPZSMotion, same columns as ep_sad
      int ep_sad[7][12];                         // EPZSDistortion, 4x4 columns 4 * mbx - 4 .. 4 * mbx + 7
      short ep_colb[6][6][2];                    // EPZSCo_located->mv of the 4x4 blocks [4 mby - 1 .. 4 mby + 4][4 mbx - 1 .. 4 mbx + 4] (temporal predictors)
      uint32_t ep_seen[160];                     // cells of JM's map (side 2 search_range + 1) some search of this macroblock has tested
      int ep_i, ep_alias_on;                     // integer searches of this macroblock so far; the macroblock has pre-marked cells
    };
    struct {                                                        // UMHexagonS and its simplified form
      int um_ref_cost[WR][8][16];                // fastme_ref_cost[ref][blocktype][by][bx]
      int um_best_cost[8][4];                    // fastme_best_cost[blocktype - 1][pic_pix_x >> 2] (only ever read after being written by the same block)
      int um_loc[8][5][6];                       // fastme_l0_cost per block type
      uint8_t um_sstate[52];                     // SearchState 7x7
    };
  };
};

// ONE instance at file scope: every device function names it directly, so its accesses are LDS instructions (ds_read / ds_write) also where a
// function is not inlined -- passed by reference it decays to a generic pointer and every access becomes a FLAT instruction
__shared__ Lds g_lds;
#define L g_lds

// which stored row of the EPZS memories holds, for macroblock (mbx, mby), the entry of 4x4 column col: the macroblock of that column in this
// row if it comes earlier in the slice, else the one above it; 0 = the state the slice started from
__device__ __forceinline__ int state_row(int col, int mbx, int mby)
{
  const int cx = col >> 2, yy = cx < mbx ? mby : mby - 1;
  if (yy < 0) return 0;
  const int a = yy * D.mbw + cx;
  if (a < D.p.mb_first || a >= D.p.mb_first + D.p.mb_count) return 0;
  return yy - D.row0 + 1;
}
#define FREF(py, px) L.f_ref[(py) - 4 * L.mby + 1][(px) - 4 * L.mbx + 1]
#define FMV(py, px, c) L.f_mv[(py) - 4 * L.mby + 1][(px) - 4 * L.mbx + 1][c]
#define UMLOC(bt, py, px) L.um_loc[bt][(py) - 4 * L.mby + 1][(px) - 4 * L.mbx + 1]

#ifdef JMHIP_WAVE_PROF
__device__ unsigned long long g_wave_prof[16];
#define WPROF_T0 unsigned long long wp_t = __builtin_amdgcn_s_memtime()
#define WPROF(k) do { const unsigned long long wp_n = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) atomicAdd(&g_wave_prof[k], wp_n - wp_t); wp_t = wp_n; } while (0)
#else
#define WPROF_T0 do { } while (0)
#define WPROF(k) do { } while (0)
#endif

__constant__ int8_t c_bsx[8] = {16, 16, 16, 8, 8, 8, 4, 4};
__constant__ int8_t c_bsy[8] = {16, 16, 8, 16, 8, 4, 8, 4};

__device__ __forceinline__ int rshift_rnd_sf(int x, int a) { return (x + (1 << (a - 1))) >> a; }
__device__ __forceinline__ int imin3(int a, int b, int c) { return min(a, min(b, c)); }

// ---------------------------------------------------------------------------------------------- candidate batches

// L.dist[k] = distortion of the bsx x bsy block at (bx, by) of the macroblock against the reference block whose origin is the padded
// quarter-pel position (L.cx[k], L.cy[k]), k < n. metric 0: computeSAD(WP) (me_distortion.c:351/413), one origin clamp per block; metric 2:
// computeSATD(WP) (:657/:734), origin clamp per 4x4 (t8: 8x8) sub-block. umv: ref_access_method.
// Three functions rather than one: a function that touches no callee-saved VGPR (v40 and up) has no save / restore frame, and the SAD path --
// half of a macroblock's ~200 evaluation calls -- needs a dozen registers, the 8x8 Hadamard path over a hundred.
#define EVAL_ARGS const uint8_t *planes, int umv, int wp, int wpw, int wpo, int bx, int by, int bsx, int bsy, int n
// SSE: computeSSE(WP) (me_distortion.c:1042/1107) -- the same samples, squared differences
template <bool SSE> __device__ __forceinline__ void eval_abs(EVAL_ARGS)
{
  const int lane = threadIdx.x;
  const size_t psz = (size_t)D.Wp * D.Hp;
  const int wpad = D.Wp - 17, hpad = D.Hp - 17;           // size_x_pad / size_y_pad, mbuffer.c:421-422
#ifdef JMHIP_WAVE_PROF
  const unsigned long long ev_t0 = __builtin_amdgcn_s_memtime();
#endif
  __syncthreads();

    const int rowdw = bsx >> 2, seg = rowdw * bsy, cpb = 64 / seg;
    const int d = lane % seg, row = d / rowdw, c4 = d - row * rowdw;
    const uint32_t curv = *reinterpret_cast<const uint32_t *>(&L.cur[by + row][bx + 4 * c4]);
    for (int base = 0; base < n; base += 4 * cpb) {          // four passes' loads in flight before the first is consumed
      uint32_t rr[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int k = base + u * cpb + lane / seg;
        rr[u] = 0;
        if (k < n) {
          const int cx = L.cx[k], cy = L.cy[k];
          int ix = cx >> 2, iy = cy >> 2;
          if (umv) { ix = clampi(ix, 0, wpad); iy = clampi(iy, 0, hpad); }
          uint32_t hi;
          fetch_row(planes + psz * ((cy & 3) * 4 + (cx & 3)) + (size_t)(iy + row) * D.Wp + ix + 4 * c4, 4, &rr[u], &hi);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int k = base + u * cpb + lane / seg;
        if (base + u * cpb >= n) break;
        uint32_t r = rr[u];
        if (wp) r = wp_apply4(r, wpw, wpo, D.p.wp_round, D.p.wp_denom);
        int v = 0;
        if (k < n) {
          if (SSE) {
#pragma unroll
            for (int q = 0; q < 4; q++) { const int df = (int)((r >> (8 * q)) & 255u) - (int)((curv >> (8 * q)) & 255u); v += df * df; }
          } else v = (int)__builtin_amdgcn_sad_u8(r, curv, 0u);
        }
        for (int o = 1; o < seg; o <<= 1) v += __shfl_xor(v, o);
        if (k < n && d == 0) L.dist[k] = v;
      }
    }
  __syncthreads();
#ifdef JMHIP_WAVE_PROF
  if (threadIdx.x == 0) { atomicAdd(&g_wave_prof[10], __builtin_amdgcn_s_memtime() - ev_t0); atomicAdd(&g_wave_prof[11], 1ull); atomicAdd(&g_wave_prof[12], (unsigned long long)n); }
#endif
}


__device__ __attribute__((noinline)) void eval_sad(EVAL_ARGS) { eval_abs<false>(planes, umv, wp, wpw, wpo, bx, by, bsx, bsy, n); }
__device__ __attribute__((noinline)) void eval_sse(EVAL_ARGS) { eval_abs<true>(planes, umv, wp, wpw, wpo, bx, by, bsx, bsy, n); }

__device__ __attribute__((noinline)) void eval_satd4(EVAL_ARGS)
{
  const int lane = threadIdx.x;
  const size_t psz = (size_t)D.Wp * D.Hp;
  const int wpad = D.Wp - 17, hpad = D.Hp - 17;           // size_x_pad / size_y_pad, mbuffer.c:421-422
#ifdef JMHIP_WAVE_PROF
  const unsigned long long ev_t0 = __builtin_amdgcn_s_memtime();
#endif
  __syncthreads();
  {
    constexpr int t8 = 0;

    const int bs = t8 ? 8 : 4, nsx = bsx / bs, nsub = nsx * (bsy / bs), cpb = 64 / nsub;
    const int s = lane % nsub, sy = s / nsx, sx = s - sy * nsx;
    {
      for (int base = 0; base < n; base += 4 * cpb) {        // four passes' loads in flight before the first is consumed
        uint32_t rv[4][4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int k = base + u * cpb + lane / nsub;
#pragma unroll
          for (int r = 0; r < 4; r++) rv[u][r] = 0;
          if (k < n) {
            const int ox = L.cx[k] + sx * 16, oy = L.cy[k] + sy * 16;
            int ix = ox >> 2, iy = oy >> 2;
            if (umv) { ix = clampi(ix, 0, wpad); iy = clampi(iy, 0, hpad); }
            const uint8_t *p = planes + psz * ((oy & 3) * 4 + (ox & 3)) + (size_t)iy * D.Wp + ix;
#pragma unroll
            for (int r = 0; r < 4; r++) {
              uint32_t hi;
              fetch_row(p + (size_t)r * D.Wp, 4, &rv[u][r], &hi);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int k = base + u * cpb + lane / nsub;
          if (base + u * cpb >= n) break;
          int v = 0;
          if (k < n) {
            int df[4][4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
              uint32_t q = rv[u][r];
              if (wp) q = wp_apply4(q, wpw, wpo, D.p.wp_round, D.p.wp_denom);
              const uint32_t cv = *reinterpret_cast<const uint32_t *>(&L.cur[by + sy * 4 + r][bx + sx * 4]);
#pragma unroll
              for (int c = 0; c < 4; c++) df[r][c] = (int)((cv >> (8 * c)) & 255u) - (int)((q >> (8 * c)) & 255u);
            }
            v = satd4x4(df);
          }
          for (int o = 1; o < nsub; o <<= 1) v += __shfl_xor(v, o);
          if (k < n && s == 0) L.dist[k] = v;
        }
      }
    }
  }
  __syncthreads();
#ifdef JMHIP_WAVE_PROF
  if (threadIdx.x == 0) { atomicAdd(&g_wave_prof[10], __builtin_amdgcn_s_memtime() - ev_t0); atomicAdd(&g_wave_prof[11], 1ull); atomicAdd(&g_wave_prof[12], (unsigned long long)n); }
#endif
}


__device__ __attribute__((noinline)) void eval_satd8(EVAL_ARGS)
{
  const int lane = threadIdx.x;
  const size_t psz = (size_t)D.Wp * D.Hp;
  const int wpad = D.Wp - 17, hpad = D.Hp - 17;           // size_x_pad / size_y_pad, mbuffer.c:421-422
#ifdef JMHIP_WAVE_PROF
  const unsigned long long ev_t0 = __builtin_amdgcn_s_memtime();
#endif
  __syncthreads();
  {
    constexpr int t8 = 1;

    const int bs = t8 ? 8 : 4, nsx = bsx / bs, nsub = nsx * (bsy / bs), cpb = 64 / nsub;
    const int s = lane % nsub, sy = s / nsx, sx = s - sy * nsx;
    for (int base = 0; base < n; base += cpb) {
      const int k = base + lane / nsub;
      int v = 0;
      if (k < n) {
        const int ox = L.cx[k] + sx * bs * 4, oy = L.cy[k] + sy * bs * 4;
        int ix = ox >> 2, iy = oy >> 2;
        if (umv) { ix = clampi(ix, 0, wpad); iy = clampi(iy, 0, hpad); }
        const uint8_t *p = planes + psz * ((oy & 3) * 4 + (ox & 3)) + (size_t)iy * D.Wp + ix;
        {
          int m[8][8];
#pragma unroll
          for (int r = 0; r < 8; r++) {
            uint32_t lo, hi;
            fetch_row(p + (size_t)r * D.Wp, 8, &lo, &hi);
            if (wp) { lo = wp_apply4(lo, wpw, wpo, D.p.wp_round, D.p.wp_denom); hi = wp_apply4(hi, wpw, wpo, D.p.wp_round, D.p.wp_denom); }
            const uint32_t c0 = *reinterpret_cast<const uint32_t *>(&L.cur[by + sy * 8 + r][bx + sx * 8]);
            const uint32_t c1 = *reinterpret_cast<const uint32_t *>(&L.cur[by + sy * 8 + r][bx + sx * 8 + 4]);
#pragma unroll
            for (int c = 0; c < 4; c++) {
              m[r][c] = (int)((c0 >> (8 * c)) & 255u) - (int)((lo >> (8 * c)) & 255u);
              m[r][c + 4] = (int)((c1 >> (8 * c)) & 255u) - (int)((hi >> (8 * c)) & 255u);
            }
            had8(m[r]);
          }
          int sad = 0;
#pragma unroll
          for (int c = 0; c < 8; c++) {
            int col[8];
#pragma unroll
            for (int r = 0; r < 8; r++) col[r] = m[r][c];
            had8(col);
#pragma unroll
            for (int r = 0; r < 8; r++) sad += iabs(col[r]);
          }
          v = (sad + 2) >> 2;                               // HadamardSAD8x8, me_distortion.c:340-346
        }
      }
      for (int o = 1; o < nsub; o <<= 1) v += __shfl_xor(v, o);
      if (k < n && s == 0) L.dist[k] = v;
    }
  }
  __syncthreads();
#ifdef JMHIP_WAVE_PROF
  if (threadIdx.x == 0) { atomicAdd(&g_wave_prof[10], __builtin_amdgcn_s_memtime() - ev_t0); atomicAdd(&g_wave_prof[11], 1ull); atomicAdd(&g_wave_prof[12], (unsigned long long)n); }
#endif
}


__device__ __forceinline__ void eval_dist(const uint8_t *planes, int metric, int t8, int umv, int wp, int wpw, int wpo, int bx, int by, int bsx, int bsy, int n)
{
  if (metric == 0) eval_sad(planes, umv, wp, wpw, wpo, bx, by, bsx, bsy, n);
  else if (metric == 1) eval_sse(planes, umv, wp, wpw, wpo, bx, by, bsx, bsy, n);
  else if (!t8) eval_satd4(planes, umv, wp, wpw, wpo, bx, by, bsx, bsy, n);
  else eval_satd8(planes, umv, wp, wpw, wpo, bx, by, bsx, bsy, n);
}

// ---------------------------------------------------------------------------------------------- neighbours and the predictor

struct Nbr { int avail[4], ref[4], mvx[4], mvy[4], posx[4], posy[4]; };

// getLuma4x4Neighbour for frame pictures (mb_access.c): luma offset (xN, yN) relative to macroblock (mbx, mby); available = inside the picture
// and inside the slice (addresses [mb_first, mb_first + mb_count))
__device__ __forceinline__ int nbr_pos(int mbx, int mby, int xN, int yN, int *px, int *py)
{
  int nx = mbx, ny = mby;
  if (xN < 0 && yN < 0) { nx--; ny--; }
  else if (xN < 0 && yN < 16) nx--;
  else if (xN >= 0 && xN < 16 && yN < 0) ny--;
  else if (xN >= 0 && xN < 16 && yN >= 0 && yN < 16) { }
  else if (xN >= 16 && yN < 0) { nx++; ny--; }
  else return 0;
  if (nx < 0 || ny < 0 || nx >= D.mbw || ny >= D.mbh) return 0;
  const int a = ny * D.mbw + nx;
  if (a < D.p.mb_first || a >= D.p.mb_first + D.p.mb_count) return 0;
  if (D.p.slice_mbs > 0) {                               // several slices in one call: the neighbour must lie in the current macroblock's slice
    const int cur = mby * D.mbw + mbx, lo = D.p.mb_first + ((cur - D.p.mb_first) / D.p.slice_mbs) * D.p.slice_mbs;
    if (a < lo) return 0;
  }
  *px = (nx * 16 + ((xN + 16) & 15)) >> 2; *py = (ny * 16 + ((yN + 16) & 15)) >> 2;
  return 1;
}

__device__ void neighbours(int mbx, int mby, int mb_x, int mb_y, int bsx, Nbr &nb)
{
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int xN = mb_x + (k == 1 ? 0 : (k == 2 ? bsx : -1)), yN = mb_y + (k == 0 ? 0 : -1);
    nb.avail[k] = nbr_pos(mbx, mby, xN, yN, &nb.posx[k], &nb.posy[k]);
    nb.ref[k] = -1; nb.mvx[k] = nb.mvy[k] = 0;
    if (nb.avail[k]) { nb.ref[k] = FREF(nb.posy[k], nb.posx[k]); nb.mvx[k] = FMV(nb.posy[k], nb.posx[k], 0); nb.mvy[k] = FMV(nb.posy[k], nb.posx[k], 1); }
  }
}

// the up-right neighbour JM cannot use because it lies in a part of the macroblock coded later (mv-search.c:108-127, me_epzs.c:1660-1688)
__device__ __forceinline__ void fix_block_c(int mb_x, int mb_y, int bsx, int *c_avail)
{
  if (mb_y > 0) {
    if (mb_x < 8) {
      if (mb_y == 8) { if (bsx == 16) *c_avail = 0; }
      else if (mb_x + bsx == 8) *c_avail = 0;
    } else if (mb_x + bsx == 16) *c_avail = 0;
  }
}

// SetMotionVectorPredictor mv-search.c:87 / UMHEXSetMotionVectorPredictor me_umhex.c:1298 (dsr: also the dynamic search range)
__device__ void mv_predictor(int mbx, int mby, int ref_frame, int mb_x, int mb_y, int bsx, int bsy, int *pmx, int *pmy,
                             int dsr, int umhex_bt, int *search_range, int *sad_abc)
{
  Nbr nb;
  neighbours(mbx, mby, mb_x, mb_y, bsx, nb);
  fix_block_c(mb_x, mb_y, bsx, &nb.avail[2]);
  if (!nb.avail[2]) { nb.avail[2] = nb.avail[3]; nb.ref[2] = nb.ref[3]; nb.mvx[2] = nb.mvx[3]; nb.mvy[2] = nb.mvy[3]; nb.posx[2] = nb.posx[3]; nb.posy[2] = nb.posy[3]; }
  const int rL = nb.avail[0] ? nb.ref[0] : -1, rU = nb.avail[1] ? nb.ref[1] : -1, rUR = nb.avail[2] ? nb.ref[2] : -1;
  int type = 0;
  if (rL == ref_frame && rU != ref_frame && rUR != ref_frame) type = 1;
  else if (rL != ref_frame && rU == ref_frame && rUR != ref_frame) type = 2;
  else if (rL != ref_frame && rU != ref_frame && rUR == ref_frame) type = 3;
  if (bsx == 8 && bsy == 16) { if (mb_x == 0) { if (rL == ref_frame) type = 1; } else if (rUR == ref_frame) type = 3; }
  else if (bsx == 16 && bsy == 8) { if (mb_y == 0) { if (rU == ref_frame) type = 2; } else if (rL == ref_frame) type = 1; }
  int sa = 0, sb = 0, sc = 0, sd = 0;
  if (dsr) {                                                           // neighbourhood SAD prediction, me_umhex.c:1441-1448
    sa = nb.avail[0] ? UMLOC(umhex_bt, nb.posy[0], nb.posx[0]) : 0;
    sb = nb.avail[1] ? UMLOC(umhex_bt, nb.posy[1], nb.posx[1]) : 0;
    sd = nb.avail[3] ? UMLOC(umhex_bt, nb.posy[3], nb.posx[3]) : 0;
    sc = nb.avail[2] ? UMLOC(umhex_bt, nb.posy[2], nb.posx[2]) : sd;
  }
  int tmp_range[2] = {0, 0};
  const int R = D.p.search_range;
#pragma unroll
  for (int hv = 0; hv < 2; hv++) {
    const int a = nb.avail[0] ? (hv ? nb.mvy[0] : nb.mvx[0]) : 0, b = nb.avail[1] ? (hv ? nb.mvy[1] : nb.mvx[1]) : 0, c = nb.avail[2] ? (hv ? nb.mvy[2] : nb.mvx[2]) : 0;
    int pv;
    if (type == 0) pv = !(nb.avail[1] || nb.avail[2]) ? a : a + b + c - imin3(a, b, c) - max(a, max(b, c));
    else pv = type == 1 ? a : (type == 2 ? b : c);
    if (hv) *pmy = pv; else *pmx = pv;
    if (dsr) {                                                         // me_umhex.c:1518-1536
      if (nb.avail[0] + nb.avail[1] + nb.avail[2] < 2) tmp_range[hv] = R;
      else {
        const int mx = max(iabs(a), max(iabs(b), iabs(c))), sum = iabs(a) + iabs(b) + iabs(c);
        const int small_range = sum == 0 ? (R + 4) >> 3 : (sum > 3 ? (R + 2) >> 2 : (3 * R + 8) >> 4);
        tmp_range[hv] = min(R, max(small_range, mx << 1));
        if (max(sa, max(sb, sc)) > D.p.umhex_thres[3][umhex_bt]) tmp_range[hv] = R;
      }
    }
  }
  if (dsr) {
    const int nr = max(tmp_range[0], tmp_range[1]);
    if (D.p.full_search == 2) *search_range = nr;
    else if (D.p.full_search == 1) *search_range = nr / (min(ref_frame, 1) + 1);
    else *search_range = nr / ((min(ref_frame, 1) + 1) * min(2, umhex_bt));    // blocktype_lut[..] is the block type itself (configfile.c:830-836)
  }
  if (sad_abc) { sad_abc[0] = sa; sad_abc[1] = sb; sad_abc[2] = sc; }
}

// ---------------------------------------------------------------------------------------------- one block search: shared context

struct Blk {
  int mbx, mby;                 // macroblock
  int mb_x, mb_y;               // block origin inside the macroblock (pels)
  int bt, bsx, bsy, ref;
  int pic_x, pic_y;             // block origin in the picture
  int pmx, pmy;                 // predictor, quarter-pel
  const uint8_t *planes;
  int wp, wpw, wpo;             // weighted reference ME
  int t8;                       // test8x8transform
};

__shared__ Blk g_blk;
#define B g_blk

__device__ __forceinline__ int mvc(int lam, int mvx_q, int mvy_q) { return mv_cost(lam, mvx_q - B.pmx, mvy_q - B.pmy); }
__device__ __forceinline__ int padq(int pic, int v_q) { return ((pic + JMHIP_PAD) << 2) + v_q; }

// visited map of one search: (2R+1)^2 bits, index relative to the search centre
__device__ __forceinline__ void map_clear(int R) { const int n = ((2 * R + 1) * (2 * R + 1) + 31) >> 5; __syncthreads(); for (int i = threadIdx.x; i < n; i += 64) L.map[i] = 0u; __syncthreads(); }
__device__ __forceinline__ int map_test_set(int R, int dx, int dy)
{
  const int i = (dy + R) * (2 * R + 1) + (dx + R);
  const uint32_t w = L.map[i >> 5], b = 1u << (i & 31);
  if (w & b) return 1;
  L.map[i >> 5] = w | b;
  return 0;
}
__device__ __forceinline__ int map_test(int R, int dx, int dy) { const int i = (dy + R) * (2 * R + 1) + (dx + R); return (L.map[i >> 5] >> (i & 31)) & 1u; }
__device__ __forceinline__ void map_set(int R, int dx, int dy) { const int i = (dy + R) * (2 * R + 1) + (dx + R); L.map[i >> 5] |= 1u << (i & 31); }
// EPZS: search `si` of this macroblock tests (or, the centre, stamps) the cell JM addresses as EPZSMap[R + dy][R + dx] (mapCenter, me_epzs.c:1537).
// Recorded per test only in a macroblock that has pre-marked cells (a set bit of its bitmaps need not be a tested cell); every other macroblock
// records a whole search at once from the search's bitmap (ep_record) -- a store per test would be waited for at every evaluation call.
__device__ __forceinline__ void ep_touch(int R, int dx, int dy, int si, bool stamped = false)
{
  const int cell = (R + dy) * (2 * D.p.search_range + 1) + (R + dx);
  const uint32_t w = L.ep_seen[cell >> 5], b = 1u << (cell & 31);
  if (threadIdx.x == 0) {
    const size_t at = (size_t)(L.mby * D.mbw + L.mbx) * D.ep_cells + cell;
    if (!(w & b)) D.ep_first[at] = stamped ? EP_STAMPED : (uint8_t)si;
    D.ep_last[at] = (uint8_t)si;
  }
  L.ep_seen[cell >> 5] = w | b;
}

// EPZS, after search `si` of a macroblock without pre-marked cells: every set bit of the search's bitmap is a cell the search tested or stamped
__device__ void ep_record(int R, int si)
{
  const int side = 2 * R + 1, nw = (side * side + 31) >> 5, jside = 2 * D.p.search_range + 1;
  const size_t at = (size_t)(L.mby * D.mbw + L.mbx) * D.ep_cells;
  const float inv = 1.0f / (float)side;
  const int centre = R * side + R;
  __syncthreads();
  for (int w = threadIdx.x; w < nw; w += 64) {
    uint32_t bits = L.map[w];
    while (bits) {
      const int i = (w << 5) + __builtin_ctz(bits);
      bits &= bits - 1;
      int row = (int)((float)i * inv);
      if (row * side > i) row--;
      if ((row + 1) * side <= i) row++;
      const int cell = row * jside + (i - row * side);
      const uint32_t b = 1u << (cell & 31);
      if (!(atomicOr(&L.ep_seen[cell >> 5], b) & b)) D.ep_first[at + cell] = i == centre ? EP_STAMPED : (uint8_t)si;
      D.ep_last[at + cell] = (uint8_t)si;
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------- SubPelBlockMotionSearch (me_fullsearch.c:341)

__device__ int subpel_full(int *mvx, int *mvy, int min_mcost)
{
  const int start_hp = D.p.metric[0] != D.p.metric[1] ? 0 : 1, start_qp = D.p.metric[1] != D.p.metric[2] ? 0 : 1;
  const int max_x4 = (D.W - B.bsx + 2 * JMHIP_PAD) << 2, max_y4 = (D.H - B.bsy + 2 * JMHIP_PAD) << 2;
  const int check0 = (!D.p.rdopt && B.ref == 0 && B.bt == 1 && *mvx == 0 && *mvy == 0);       // check_position0 (!rdopt, P slice)
  for (int phase = 0; phase < 2; phase++) {
    const int step = phase ? 1 : 2, first = phase ? start_qp : start_hp, lam = D.p.lambda_mf[phase ? 2 : 1], m = phase ? 0 : 1;
    const int p4x = padq(B.pic_x, *mvx), p4y = padq(B.pic_y, *mvy);
    const int umv = !((p4x > m) && (p4x < max_x4 - m) && (p4y > m) && (p4y < max_y4 - m));
    if (phase && !start_qp) min_mcost = INT_MAX;
    int n = 0;
    for (int pos = first; pos < 9; pos++) { L.cx[n] = p4x + c_s9x[pos] * step; L.cy[n] = p4y + c_s9y[pos] * step; n++; }
    eval_dist(B.planes, D.p.metric[phase ? 2 : 1], B.t8, umv, B.wp, B.wpw, B.wpo, B.mb_x, B.mb_y, B.bsx, B.bsy, n);
    int best = 0;
    for (int pos = first, k = 0; pos < 9; pos++, k++) {
      int mcost = mvc(lam, *mvx + c_s9x[pos] * step, *mvy + c_s9y[pos] * step);
      if (mcost >= min_mcost) continue;
      mcost += L.dist[k];
      if (phase == 0 && pos == 0 && check0) mcost -= (lam * 16) >> 16;
      if (mcost < min_mcost) { min_mcost = mcost; best = pos; }
    }
    if (best) { *mvx += c_s9x[best] * step; *mvy += c_s9y[best] * step; }
  }
  return min_mcost;
}

// ---------------------------------------------------------------------------------------------- FullSearch / FastFullSearch (exhaustive window, one wave)

// argmin over the (2R+1)^2 window round (cx, cy) [pels] of SAD + mv cost with JM's order of preference (lowest spiral index on ties).
// ffs: FastFullPelBlockMotionSearch (me_fullfast.c:833: the zero vector is tried first when !rdopt and then keeps ties);
// else FullPelBlockMotionSearch (me_fullsearch.c:47: check_for_00 bonus :129-132 incl. the wrapped first-row bound, see me_int.hip).
__device__ int full_window(int cx, int cy, int R, int ffs, int *mvx, int *mvy)
{
  const int lane = threadIdx.x, side = 2 * R + 1, npos = side * side, lam = D.p.lambda_mf[0];
  const size_t psz = (size_t)D.Wp * D.Hp;
  (void)psz;
  const int wpad = D.Wp - 17, hpad = D.Hp - 17;
  // FAST access only when the whole window is inside (me_fullsearch.c:110-118); UMV otherwise: clamping the origin is the identity inside
  const int rowdw = B.bsx >> 2, seg = rowdw * B.bsy, cpb = 64 / seg;
  const int d = lane % seg, row = d / rowdw, c4 = d - row * rowdw;
  const uint32_t curv = *reinterpret_cast<const uint32_t *>(&L.cur[B.mb_y + row][B.mb_x + 4 * c4]);
  const int check00 = (!ffs && !D.p.rdopt && B.bt == 1 && B.ref == 0);
  const int w16 = (lam * 16) >> 16;
  unsigned best = 0xffffffffu;
  __syncthreads();
  constexpr int NB = 8;                                                 // passes whose loads are in flight together (the loop is latency-bound)
  for (int base = 0; base < npos; base += NB * cpb) {
    uint32_t rr[NB];
#pragma unroll
    for (int u = 0; u < NB; u++) {
      const int k = base + u * cpb + lane / seg;
      rr[u] = 0;
      if (k < npos) {
        const int dy = k / side - R, dx = k - (dy + R) * side - R;
        int ix = B.pic_x + cx + dx + JMHIP_PAD, iy = B.pic_y + cy + dy + JMHIP_PAD;
        ix = clampi(ix, 0, wpad); iy = clampi(iy, 0, hpad);
        uint32_t hi;
        fetch_row(B.planes + (size_t)(iy + row) * D.Wp + ix + 4 * c4, 4, &rr[u], &hi);
      }
    }
#pragma unroll
    for (int u = 0; u < NB; u++) {
      const int k = base + u * cpb + lane / seg;
      if (base + u * cpb >= npos) break;
      int v = 0, dx = 0, dy = 0;
      if (k < npos) {
        dy = k / side - R; dx = k - (dy + R) * side - R;
        uint32_t r = rr[u];
        if (B.wp) r = wp_apply4(r, B.wpw, B.wpo, D.p.wp_round, D.p.wp_denom);
        v = (int)__builtin_amdgcn_sad_u8(r, curv, 0u);
      }
      int rowsum = v;                                                    // SAD of the first row (for the wrapped bound)
      for (int o = 1; o < rowdw; o <<= 1) rowsum += __shfl_xor(rowsum, o);
      for (int o = 1; o < seg; o <<= 1) v += __shfl_xor(v, o);
      if (k < npos && d == 0) {
        int mc = mv_cost(lam, ((cx + dx) << 2) - B.pmx, ((cy + dy) << 2) - B.pmy);
        int tie = spiral_pos(dx, dy) + 1;
        if (check00 && ((B.pic_x + cx + dx) << 2) == B.pic_x && ((B.pic_y + cy + dy) << 2) == B.pic_y) {
          mc -= w16;
          if (dx == 0 && dy == 0 && mc < 0) v = rowsum;                  // INT_MAX - (negative) wraps: computeSAD leaves after row 0
        }
        if (ffs && !D.p.rdopt && cx + dx == 0 && cy + dy == 0) tie = 0;                // pos_00 pre-check (!rdopt, me_fullfast.c:867)
        const int cost = mc + v;                                          // may be slightly negative through the bonus
        const unsigned key = ((unsigned)(cost + 4096) << TIE_BITS) | (unsigned)tie;
        best = min(best, key);
      }
    }
  }
  for (int o = 1; o < 64; o <<= 1) best = min(best, (unsigned)__shfl_xor((int)best, o));
  const int cost = (int)(best >> TIE_BITS) - 4096, tie = (int)(best & ((1u << TIE_BITS) - 1));
  if (tie == 0) { *mvx = 0; *mvy = 0; }
  else { int dx, dy; spiral_offset(tie - 1, &dx, &dy); *mvx = cx + dx; *mvy = cy + dy; }
  __syncthreads();
  return cost;
}


// ---------------------------------------------------------------------------------------------- exhaustive searches through SAD surfaces

// SAD is independent of the predictor: once per (macroblock, reference) the SADs of the sixteen 4x4 blocks (and their 8x8 sums) are computed
// for every displacement in a window round (scx, scy) and kept; every partition then finds its own argmin of (SAD + its own mv cost) over its
// own window -- SetupFastFullPelSearch / SetupLargerBlocks (me_fullfast.c:491, :210) made macroblock-wide. lane <-> displacement.
// slot 1: the surface of ONE partition whose centre the macroblock's surface (slot 0) does not cover, block rows [y4lo, y4hi) only
__device__ void surface_build(int scx, int scy, int Rs, int slot = 0, int y4lo = 0, int y4hi = 4)
{
  const int lane = threadIdx.x, side = 2 * Rs + 1, wside = side + 15, n = side * side;
  const int ox = B.mbx * 16 + scx - Rs + JMHIP_PAD, oy = B.mby * 16 + scy - Rs + JMHIP_PAD;     // window origin in the padded integer plane
  __syncthreads();
  const float rwside = 1.0f / (float)wside, rsd = 1.0f / (float)side;             // quotients below: (i + 0.5) * reciprocal, exact for these ranges (surface_search)
  for (int i = lane + 4 * y4lo * wside; i < (4 * y4hi + side - 1) * wside; i += 64) {     // the window rows these block rows read
    const int wy = (int)(((float)i + 0.5f) * rwside), wx = i - wy * wside;
    const int py = clampi(oy + wy, 0, D.Hp - 1), px = clampi(ox + wx, 0, D.Wp - 1);                // the ring replicates the edge: per-sample clamp == origin clamp
    int v = B.planes[(size_t)py * D.Wp + px];
    if (B.wp) v = min(max(((B.wpw * v + D.p.wp_round) >> D.p.wp_denom) + B.wpo, 0), 255);
    L.win[i] = (uint8_t)v;
  }
  __syncthreads();
  uint16_t *sf = D.surf + (((size_t)blockIdx.x * 2 + slot) * D.p.num_refs + B.ref) * SURF_PLANES * D.surf_n;
  for (int k = lane; k < n; k += 64) {
    const int dy = (int)(((float)k + 0.5f) * rsd), dx = k - dy * side;
    unsigned s4[16];
#pragma unroll
    for (int by = 0; by < 4; by++) {
      unsigned a0 = 0, a1 = 0, a2 = 0, a3 = 0;
      if (by < y4lo || by >= y4hi) { s4[by * 4] = s4[by * 4 + 1] = s4[by * 4 + 2] = s4[by * 4 + 3] = 0; continue; }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint8_t *w = &L.win[(dy + by * 4 + r) * wside + dx];
        const uint32_t *q = reinterpret_cast<const uint32_t *>(reinterpret_cast<uintptr_t>(w) & ~uintptr_t(3));
        const unsigned sh = (unsigned)(reinterpret_cast<uintptr_t>(w) & 3);
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
        const uint32_t *c = reinterpret_cast<const uint32_t *>(&L.cur[by * 4 + r][0]);
        a0 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(d1, d0, sh), c[0], a0);
        a1 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(d2, d1, sh), c[1], a1);
        a2 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(d3, d2, sh), c[2], a2);
        a3 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(d4, d3, sh), c[3], a3);
      }
      s4[by * 4] = a0; s4[by * 4 + 1] = a1; s4[by * 4 + 2] = a2; s4[by * 4 + 3] = a3;
    }
#pragma unroll
    for (int b = 0; b < 16; b++) sf[(size_t)b * D.surf_n + k] = (uint16_t)s4[b];
#pragma unroll
    for (int b8 = 0; b8 < 4; b8++) {
      const int b = (b8 >> 1) * 8 + (b8 & 1) * 2;
      sf[(size_t)(16 + b8) * D.surf_n + k] = (uint16_t)(s4[b] + s4[b + 1] + s4[b + 4] + s4[b + 5]);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  int *sc = L.surf_c[slot * WR + B.ref];
  sc[0] = scx; sc[1] = scy; sc[2] = Rs; sc[3] = 1; sc[4] = ((1 << y4hi) - 1) & ~((1 << y4lo) - 1);
}

// argmin of SAD + mv cost over the (2R+1)^2 window round (cx, cy) from the surface of B.ref. Returns INT_MIN when the window is not covered.
__device__ int surface_search(int cx, int cy, int R, int ffs, int *mvx, int *mvy, int slot = 0)
{
  const int *sc = L.surf_c[slot * WR + B.ref];
  const int scx = sc[0], scy = sc[1], Rs = sc[2];
  const int rows = ((1 << ((B.mb_y + B.bsy) >> 2)) - 1) & ~((1 << (B.mb_y >> 2)) - 1);
  if (!sc[3] || (sc[4] & rows) != rows || cx - R < scx - Rs || cx + R > scx + Rs || cy - R < scy - Rs || cy + R > scy + Rs) return INT_MIN;
  const int lane = threadIdx.x, side = 2 * R + 1, npos = side * side, sside = 2 * Rs + 1, lam = D.p.lambda_mf[0];
  const uint16_t *sf = D.surf + (((size_t)blockIdx.x * 2 + slot) * D.p.num_refs + B.ref) * SURF_PLANES * D.surf_n;
  const int x4 = B.mb_x >> 2, y4 = B.mb_y >> 2, w4 = B.bsx >> 2, h4 = B.bsy >> 2;
  // the planes this partition sums: whole 8x8 blocks where it covers them, 4x4 blocks otherwise
  int pl[4], npl = 0;
  if (w4 >= 2 && h4 >= 2) { for (int j = y4 >> 1; j < (y4 + h4) >> 1; j++) for (int i = x4 >> 1; i < (x4 + w4) >> 1; i++) pl[npl++] = 16 + j * 2 + i; }
  else { for (int j = y4; j < y4 + h4; j++) for (int i = x4; i < x4 + w4; i++) pl[npl++] = j * 4 + i; }
  const int check00 = (!ffs && !D.p.rdopt && B.bt == 1 && B.ref == 0), w16 = (lam * 16) >> 16;
  unsigned best = 0xffffffffu;
  // the surface reads are L2 hits with a long latency: eight candidates per lane are in flight at a time
  constexpr int UN = 8;
  const float rside = 1.0f / (float)side;
  const bool tabbed = D.tie_tab && R == D.tie_R;                       // the tie-break index from the per-call table instead of ~15 instructions per candidate
  const uint16_t *p0 = sf + (size_t)pl[0] * D.surf_n, *p1 = sf + (size_t)pl[npl > 1 ? 1 : 0] * D.surf_n;
  const uint16_t *p2 = sf + (size_t)pl[npl > 2 ? 2 : 0] * D.surf_n, *p3 = sf + (size_t)pl[npl > 3 ? 3 : 0] * D.surf_n;
  const int m1 = npl > 1 ? 0xffff : 0, m2 = npl > 2 ? 0xffff : 0, m3 = npl > 3 ? 0xffff : 0;
  for (int k0 = lane; k0 < npos; k0 += 64 * UN) {
    int v[UN], ddx[UN], ddy[UN], tt[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int k = min(k0 + 64 * u, npos - 1);                       // past the end: the last candidate again (harmless for a minimum)
      tt[u] = tabbed ? (int)D.tie_tab[k] : 0;
      // k / side without the integer-division sequence: k + 0.5 is never closer than 0.5 / side (> 0.007) to a multiple of side, the float
      // product's error stays below 1e-3 for k < 2^13
      const int row = (int)(((float)k + 0.5f) * rside);
      const int dy = row - R, dx = k - row * side - R;
      const int si = (cy + dy - scy + Rs) * sside + (cx + dx - scx + Rs);
      ddx[u] = dx; ddy[u] = dy;
      v[u] = (int)p0[si] + ((int)p1[si] & m1) + ((int)p2[si] & m2) + ((int)p3[si] & m3);
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int dx = ddx[u], dy = ddy[u];
      int mc = mv_cost(lam, ((cx + dx) << 2) - B.pmx, ((cy + dy) << 2) - B.pmy);
      int tie = tabbed ? tt[u] : spiral_pos(dx, dy) + 1;
      bool skip = false;
      if (check00 && ((B.pic_x + cx + dx) << 2) == B.pic_x && ((B.pic_y + cy + dy) << 2) == B.pic_y) {
        mc -= w16;
        skip = (dx == 0 && dy == 0 && mc < 0);                         // the wrapped first-row bound: settled below
      }
      if (ffs && !D.p.rdopt && cx + dx == 0 && cy + dy == 0) tie = 0;
      const unsigned key = ((unsigned)(mc + v[u] + 4096) << TIE_BITS) | (unsigned)tie;
      best = min(best, skip ? 0xffffffffu : key);
    }
  }
  for (int o = 1; o < 64; o <<= 1) best = min(best, (unsigned)__shfl_xor((int)best, o));
  if (check00 && ((B.pic_x + cx) << 2) == B.pic_x && ((B.pic_y + cy) << 2) == B.pic_y) {
    const int mc = mv_cost(lam, (cx << 2) - B.pmx, (cy << 2) - B.pmy) - w16;
    if (mc < 0) {                                                      // me_fullsearch.c:138 with min_mcost = INT_MAX: computeSAD leaves after row 0
      L.cx[0] = padq(B.pic_x, cx << 2); L.cy[0] = padq(B.pic_y, cy << 2);
      eval_dist(B.planes, 0, 0, 1, B.wp, B.wpw, B.wpo, 0, 0, 16, 1, 1);
      best = min(best, ((unsigned)(mc + L.dist[0] + 4096) << TIE_BITS) | 1u);
    }
  }
  const int cost = (int)(best >> TIE_BITS) - 4096, tie = (int)(best & ((1u << TIE_BITS) - 1));
  if (tie == 0) { *mvx = 0; *mvy = 0; }
  else { int dx, dy; spiral_offset(tie - 1, &dx, &dy); *mvx = cx + dx; *mvy = cy + dy; }
  return cost;
}

// ---------------------------------------------------------------------------------------------- EPZS (me_epzs.c)

// pattern tables, me_epzs.c:54-78 (>> mv_rescale = 2), with the (stop, nextLast, next pattern) triples of EPZSInit :367-378
struct EpPat { int8_t n, stop, next_last, next; int8_t mvx[12], mvy[12], start[12], npts[12]; };
__constant__ EpPat c_pat[6] = {
  {4, 1, 1, 0, {0, 1, 0, -1}, {1, 0, -1, 0}, {3, 0, 1, 2}, {3, 3, 3, 3}},
  {8, 1, 1, 1, {0, 1, 1, 1, 0, -1, -1, -1}, {1, 1, 0, -1, -1, -1, 0, 1}, {7, 7, 1, 1, 3, 3, 5, 5}, {3, 5, 3, 5, 3, 5, 3, 5}},
  {12, 1, 1, 2, {-1, 0, 0, 1, 2, 1, 1, 0, 0, -1, -2, -1}, {1, 2, 1, 1, 0, 0, -1, -2, -1, -1, 0, 0}, {10, 10, 10, 1, 1, 1, 4, 4, 4, 7, 7, 7}, {5, 8, 7, 5, 8, 7, 5, 8, 7, 5, 8, 7}},
  {8, 1, 1, 3, {0, 1, 2, 1, 0, -1, -2, -1}, {2, 1, 0, -1, -2, -1, 0, 1}, {6, 0, 0, 2, 2, 4, 4, 6}, {5, 3, 5, 3, 5, 3, 5, 3}},
  {12, 0, 1, 0, {0, 1, 2, 1, 0, -1, -2, -1, 0, 0, 0, -1}, {2, 1, 0, -1, -2, -1, 0, 1, 0, 0, -1, 0}, {6, 0, 0, 2, 2, 4, 4, 6, 6, 0, 2, 4}, {12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12}},
  {8, 0, 1, 0, {0, 1, 2, 1, 0, -1, -2, -1}, {2, 1, 0, -1, -2, -1, 0, 1}, {6, 0, 0, 2, 2, 4, 4, 6}, {5, 3, 5, 3, 5, 3, 5, 3}},
};
__constant__ int8_t c_blk_parent[8] = {1, 1, 1, 1, 2, 4, 4, 5};
__constant__ int8_t c_sp_hp[10][2] = {{0, 0}, {-2, 0}, {0, 2}, {2, 0}, {0, -2}, {-2, 2}, {2, 2}, {2, -2}, {-2, -2}, {-2, 2}};
__constant__ int8_t c_sp_qp[10][2] = {{0, 0}, {-1, 0}, {0, 1}, {1, 0}, {0, -1}, {-1, 1}, {1, 1}, {1, -1}, {-1, -1}, {-1, 1}};

struct EpState { int tmx, tmy, t2x, t2y, min_mcost, second; };

// one pass of candidates (vectors L.qx/qy[0..n), pels) through the visited map, the evaluation and the caller's replay
__device__ __forceinline__ int ep_access_umv(int vx, int vy)
{
  // CHECK_RANGE (me_epzs.h:23) compares the QUARTER-pel candidate with pel-unit picture sizes, as JM does
  const int cx = (B.pic_x + vx) << 2, cy = (B.pic_y + vy) << 2;
  return !((cx >= 0) && (cx < D.W - B.bsx) && (cy >= 0) && (cy < D.H - B.bsy));
}

// evaluates the candidates L.qx/qy[0..n) (integer vectors): EPZS picks the access method per candidate, so they go in two batches
__device__ void ep_eval(int n)
{
  // EPZS picks FAST or UMV access per candidate (CHECK_RANGE, me_epzs.h:23), and FAST only for blocks that lie inside the picture, where the
  // origin clamp of UMV access is the identity: every candidate is evaluated with UMV access.
  for (int k = 0; k < n; k++) { L.cx[k] = padq(B.pic_x, L.qx[k] << 2); L.cy[k] = padq(B.pic_y, L.qy[k] << 2); }
  eval_dist(B.planes, D.p.metric[0], B.t8, 1, B.wp, B.wpw, B.wpo, B.mb_x, B.mb_y, B.bsx, B.bsy, n);
}

// EPZSPelBlockMotionSearch, me_epzs.c:1500. mvx/mvy: search centre in, result out (pels). all_mv: this block's img->all_mv row [ref][blocktype].
__device__ __forceinline__ int epzs_pel_search(int mb_nr, int R, int *mvx, int *mvy)
{
  const jmhip_slice_params &P = D.p;
  const int lam = P.lambda_mf[0], bt = B.bt, ref = B.ref;
  const int cx0 = *mvx, cy0 = *mvy;                       // the centre (mv on entry)
  const int px2 = B.pic_x >> 2, py2 = B.pic_y >> 2, bshx = B.bsx >> 2, bshy = B.bsy >> 2, block_y = B.mb_y >> 2, block_x = B.mb_x >> 2;
  const int col0 = 4 * B.mbx - 4;                              // first column of the staged row memories
  int *prev_sad = &L.ep_sad[bt - 1][0] - col0;                 // indexed by picture column, as JM's EPZSDistortion[list][blocktype - 1]
  short *motion = P.epzs_spatial_mem ? &L.ep_mot[ref][bt - 1][block_y][px2 - col0][0] : nullptr;
  const int medthres = P.epzs_thres[1][bt];
  int stop = medthres;
  EpState S{cx0, cy0, 0, 0, 0, INT_MAX};
  const int si = L.ep_i + 1;                                    // EPZSBlkCount++ (:1550), counted per macroblock
  map_clear(R);
  L.ep_i = si;
  if (L.ep_alias_on) {                                          // cells whose old stamp equals this search's (ep_alias_detect_kernel)
    const int side = 2 * P.search_range + 1, addr = L.mby * D.mbw + L.mbx;
    for (int k = threadIdx.x; k < D.ep_alias_n; k += 64) {
      const unsigned long long e = D.ep_alias[k];
      if ((int)(e >> 32) != addr || (int)((e >> 16) & 0xffff) != si) continue;
      const int cell = (int)(e & 0xffff), row = cell / side, col = cell - row * side;
      if (row > 2 * R || col > 2 * R) continue;
      const int i = row * (2 * R + 1) + col;
      atomicOr(&L.map[i >> 5], 1u << (i & 31));
    }
    __syncthreads();
  }
  map_set(R, 0, 0);
  const int per_test = L.ep_alias_on;
  if (per_test) ep_touch(R, 0, 0, si, true);
  L.qx[0] = cx0; L.qy[0] = cy0;
  ep_eval(1);
  S.min_mcost = mvc(lam, cx0 << 2, cy0 << 2) + L.dist[0];
  WPROF_T0;
  const int psad = prev_sad[px2];
  if (ref > 0 && psad < medthres && psad < S.min_mcost) {                      // :1608-1623
    if (motion) { motion[0] = (short)S.tmx; motion[1] = (short)S.tmy; }
    return S.min_mcost;
  }
  if (S.min_mcost > stop) {
    Nbr nb;
    neighbours(B.mbx, B.mby, B.mb_x, B.mb_y, B.bsx, nb);
    const int mb_avail_right = B.mbx < D.mbw - 1, mb_avail_below = B.mby < D.mbh - 1;
    int c_avail = nb.avail[2], blk_right;
    if (B.mb_y > 0) {                                                          // :1660-1687
      if (B.mb_x < 8) {
        if (B.mb_y == 8) { blk_right = (B.bsx != 16) || mb_avail_right; if (B.bsx == 16) c_avail = 0; }
        else { blk_right = (B.mb_x + B.bsx != 8) || mb_avail_right; if (B.mb_x + B.bsx == 8) c_avail = 0; }
      } else { blk_right = (B.mb_x + B.bsx != 16) || mb_avail_right; if (B.mb_x + B.bsx == 16) c_avail = 0; }
    } else blk_right = (B.mb_x + B.bsx != 16) || mb_avail_right;
    const int blk_below = (B.mb_y + B.bsy != 16) || mb_avail_below;
    const int sadA = nb.avail[0] ? prev_sad[px2 - bshx] : INT_MAX, sadB = nb.avail[1] ? prev_sad[px2] : INT_MAX, sadC = c_avail ? prev_sad[px2 + bshx] : INT_MAX;
    stop = imin3(sadA, sadB, sadC);
    stop = max(stop, P.epzs_thres[0][bt]);
    stop = min(stop, P.epzs_thres[2][bt]);
    stop = (9 * max(medthres, stop) + 2 * medthres) >> 3;                      // :1698

    // ---- predictors: spatial :1061 (non-MBAFF), spatial memory :1253, temporal :1332, window :1480, block type :1433
    int np = 5;
    const int refA = nb.avail[0] ? nb.ref[0] : -1, refB = nb.avail[1] ? nb.ref[1] : -1, refC = c_avail ? nb.ref[2] : -1, refD = nb.avail[3] ? nb.ref[3] : -1;
    const int sh = 10;                                                          // 8 + mv_rescale
#define SC(r) ((r) < 0 ? 0 : P.epzs_mv_scale[ref][r])                          /* the vector of such a neighbour is zero: the product is 0 either way */
    L.px[0] = 0; L.py[0] = 0;
    if (nb.avail[0]) { L.px[1] = rshift_rnd_sf(SC(refA) * nb.mvx[0], sh); L.py[1] = rshift_rnd_sf(SC(refA) * nb.mvy[0], sh); } else { L.px[1] = 3; L.py[1] = 0; }
    if (nb.avail[1]) { L.px[2] = rshift_rnd_sf(SC(refB) * nb.mvx[1], sh); L.py[2] = rshift_rnd_sf(SC(refB) * nb.mvy[1], sh); } else { L.px[2] = 0; L.py[2] = 3; }
    if (c_avail)     { L.px[3] = rshift_rnd_sf(SC(refC) * nb.mvx[2], sh); L.py[3] = rshift_rnd_sf(SC(refC) * nb.mvy[2], sh); } else { L.px[3] = -3; L.py[3] = 0; }
    if (nb.avail[3]) { L.px[4] = rshift_rnd_sf(SC(refD) * nb.mvx[3], sh); L.py[4] = rshift_rnd_sf(SC(refD) * nb.mvy[3], sh); } else { L.px[4] = 0; L.py[4] = -3; }
#undef SC
    const int invalid_refs = (refA == -1) + (refB == -1) + (refC == -1 && refD == -1);
#define ADDP(X, Y) do { const int x_ = (X), y_ = (Y); L.px[np] = x_; L.py[np] = y_; np += ((x_ | y_) != 0); } while (0)
    if (P.epzs_spatial_mem) {
#define MOT(r, x, c) ((int)L.ep_mot[ref][bt - 1][r][(x) - col0][c])
      ADDP(px2 > 0 ? MOT(block_y, px2 - bshx, 0) : 0, px2 > 0 ? MOT(block_y, px2 - bshx, 1) : 0);
      ADDP(block_y > 0 ? MOT(block_y - bshy, px2, 0) : MOT(4 - bshy, px2, 0), block_y > 0 ? MOT(block_y - bshy, px2, 1) : MOT(4 - bshy, px2, 1));
      ADDP(px2 + bshx < D.w4 ? (block_y > 0 ? MOT(block_y - bshy, px2 + bshx, 0) : MOT(4 - bshy, px2 + bshx, 0)) : 0,
           px2 + bshx < D.w4 ? (block_y > 0 ? MOT(block_y - bshy, px2 + bshx, 1) : MOT(4 - bshy, px2 + bshx, 1)) : 0);
#undef MOT
    }
    if (P.epzs_temporal) {
      const int sc = P.epzs_mv_scale[ref][0];
#define COLP(y, x) ADDP(rshift_rnd_sf(sc * (int)L.ep_colb[(y) - (4 * B.mby - 1)][(x) - (4 * B.mbx - 1)][0], sh), rshift_rnd_sf(sc * (int)L.ep_colb[(y) - (4 * B.mby - 1)][(x) - (4 * B.mbx - 1)][1], sh))
      COLP(py2, px2);
      if (S.min_mcost > stop && ref < 2) {
        if (nb.avail[0]) {
          COLP(py2, px2 - 1);
          if (nb.avail[1]) COLP(py2 - 1, px2 - 1);
          if (blk_below) COLP(py2 + bshy, px2 - 1);
        }
        if (nb.avail[1]) COLP(py2 - 1, px2);
        if (blk_right) {
          COLP(py2, px2 + bshx);
          if (nb.avail[1]) COLP(py2 - 1, px2 + bshx);
          if (blk_below) COLP(py2 + bshy, px2 + bshx);
        }
        if (blk_below) COLP(py2 + bshy, px2);
      }
#undef COLP
    }
    if (S.min_mcost > stop && (ref < 2 && bt < 5) && P.epzs_fixed >= 1) {       // P slice: EPZSFixed > 1 || EPZSFixed (:1727-1733)
      const int ext = (bt < 5) && (invalid_refs > 2) && (ref < 1);
      const int nw = ext ? P.epzs_nwin_ext : P.epzs_nwin;
      for (int k = 0; k < nw; k++) { L.px[np] = cx0 + (ext ? P.epzs_win_ext[k][0] : P.epzs_win[k][0]); L.py[np] = cy0 + (ext ? P.epzs_win_ext[k][1] : P.epzs_win[k][1]); np++; }
    }
    if ((ref == 0 || S.min_mcost > stop) && mb_nr != 0) {                       // :1740-1744, EPZSBlockTypePredictors :1433
      const short (*am)[8][2] = L.all_mv[block_y * 4 + block_x];
      const int par = c_blk_parent[bt];
#define RR(v) (((int)(v) + 2) >> 2)                                             /* rshift_rnd(v, 2) */
      ADDP(RR(am[ref][par][0]), RR(am[ref][par][1]));
      if (ref > 0 && bt < 5) {
        ADDP(rshift_rnd_sf(P.epzs_mv_scale[ref][ref - 1] * (int)am[ref - 1][bt][0], sh), rshift_rnd_sf(P.epzs_mv_scale[ref][ref - 1] * (int)am[ref - 1][bt][1], sh));
        ADDP(rshift_rnd_sf(P.epzs_mv_scale[ref][0] * (int)am[0][bt][0], sh), rshift_rnd_sf(P.epzs_mv_scale[ref][0] * (int)am[0][bt][1], sh));
      }
      if (bt != 1) ADDP(RR(am[ref][1][0]), RR(am[ref][1][1]));
      if (bt != 4) ADDP(RR(am[ref][4][0]), RR(am[ref][4][1]));
#undef RR
    }
#undef ADDP
    WPROF(4);
    // ---- scan :1746-1795: the map is marked whatever the costs are, so the survivors are known before anything is evaluated
    int n = 0;
    for (int k = 0; k < np; k++) {
      const int tx = L.px[k], ty = L.py[k];
      if (iabs(tx - cx0) > R || iabs(ty - cy0) > R) continue;
      if (per_test) ep_touch(R, tx - cx0, ty - cy0, si);
      if (map_test_set(R, tx - cx0, ty - cy0)) continue;
      L.qx[n] = tx; L.qy[n] = ty; n++;
    }
    ep_eval(n);
    int check_median = 0;
    for (int k = 0; k < n; k++) {
      const int tx = L.qx[k], ty = L.qy[k];
      int mcost = mvc(lam, tx << 2, ty << 2);
      if (mcost >= S.second) continue;
      mcost += L.dist[k];
      if (mcost < S.min_mcost) { S.t2x = S.tmx; S.t2y = S.tmy; S.tmx = tx; S.tmy = ty; S.second = S.min_mcost; S.min_mcost = mcost; check_median = 1; }
      else if (mcost < S.second) { S.t2x = tx; S.t2y = ty; S.second = mcost; check_median = 1; }
    }
    WPROF(5);
    // ---- refinement :1801-1942
    if (S.min_mcost > stop) {
      const int search_pattern = (P.epzs_pattern >= 1 && P.epzs_pattern <= 5) ? P.epzs_pattern : 0;       // :410-431
      const int search_pattern_d = (P.epzs_dual >= 2 && P.epzs_dual <= 6) ? P.epzs_dual - 1 : 0;          // :433-454
      int pf = search_pattern;
      if (P.epzs_pattern != 0) {
        if (S.min_mcost < stop + ((3 * medthres) >> 1)) {
          if ((S.tmx == 0 && S.tmy == 0) || (iabs(S.tmx - cx0) < 2 && iabs(S.tmy - cy0) < 2)) pf = 0; else pf = 1;
        } else if (bt > 5 || (ref > 0 && bt != 1)) pf = 1;
        else pf = search_pattern;
      }
      int total = c_pat[pf].n, center_x = S.tmx, center_y = S.tmy, pattern_stop = 0, point = 0, next_last = 0, dir = 0;
      for (;;) {
        do {
          // one pass: `total` points from `point` on (wrapping); map first, then the batch, then the replay
          int cnt = 0;
          int pidx[12];
          for (int c = 0, pt = point; c < total; c++) {
            const int tx = center_x + c_pat[pf].mvx[pt], ty = center_y + c_pat[pf].mvy[pt];
            if (iabs(tx - cx0) <= R && iabs(ty - cy0) <= R) {
              if (per_test) ep_touch(R, tx - cx0, ty - cy0, si);
              if (!map_test_set(R, tx - cx0, ty - cy0)) { L.qx[cnt] = tx; L.qy[cnt] = ty; pidx[cnt] = pt; cnt++; }
            }
            if (++pt >= c_pat[pf].n) pt -= c_pat[pf].n;
          }
          ep_eval(cnt);
          for (int k = 0; k < cnt; k++) {
            int mcost = mvc(lam, L.qx[k] << 2, L.qy[k] << 2);
            if (mcost < S.min_mcost) {
              mcost += L.dist[k];
              if (mcost < S.min_mcost) { S.min_mcost = mcost; S.tmx = L.qx[k]; S.tmy = L.qy[k]; dir = pidx[k]; }
            }
          }
          if (next_last || (S.tmx == center_x && S.tmy == center_y)) {
            pattern_stop = c_pat[pf].stop; pf = c_pat[pf].next; total = c_pat[pf].n; next_last = c_pat[pf].next_last; dir = 0; point = 0;
          } else {
            total = c_pat[pf].npts[dir]; point = c_pat[pf].start[dir]; center_x = S.tmx; center_y = S.tmy;
          }
        } while (pattern_stop != 1);
        const int ps = prev_sad[px2];
        if (ref > 0 && ((4 * ps < S.min_mcost) || ((3 * ps < S.min_mcost) && (ps <= stop)))) {          // :1894-1911
          *mvx = S.tmx; *mvy = S.tmy;
          if (motion) { motion[0] = (short)S.tmx; motion[1] = (short)S.tmy; }
          return S.min_mcost;
        }
        if (!(check_median && S.min_mcost > stop && P.epzs_dual > 0)) break;     // P slice: (P_SLICE || blocktype < 5) is true
        point = 0; pattern_stop = 0; dir = 0; next_last = 0;
        if ((S.tmx == 0 && S.tmy == 0) || (S.tmx == cx0 && S.tmy == cy0)) { if (iabs(S.tmx - cx0) < 2 && iabs(S.tmy - cy0) < 2) pf = 0; else pf = 1; }
        else pf = search_pattern_d;
        total = c_pat[pf].n;
        center_x = S.t2x; center_y = S.t2y;
        check_median = 0;
      }
    }
  }
  WPROF(6);
  if (ref == 0 || prev_sad[px2] > S.min_mcost) prev_sad[px2] = S.min_mcost;                             // :1945
  if (motion) { motion[0] = (short)S.tmx; motion[1] = (short)S.tmy; }
  *mvx = S.tmx; *mvy = S.tmy;
  return S.min_mcost;
}

// one level of EPZSSubPelBlockMotionSearch (me_epzs.c:2472-2579 / :2605-2715). Returns 1 on the sub-threshold early return.
__device__ int epzs_pel(int mb_nr, int R, int *mvx, int *mvy)
{
  const int cost = epzs_pel_search(mb_nr, R, mvx, mvy);
  if (!L.ep_alias_on && !(D.debug & 16)) ep_record(R, L.ep_i);
  return cost;
}

__device__ int epzs_sub_level(int half, int start, int lam, int metric, int umv, int *mvx, int *mvy, int *min_mcost, int subthres)
{
  const int8_t (*pts)[2] = half ? c_sp_hp : c_sp_qp;
  const int p4x = padq(B.pic_x, 0), p4y = padq(B.pic_y, 0);
  // every point the level can ask for -- the cross 0..4 and whichever of 5..9 the two best of the cross select (:2433-2520) -- goes through ONE
  // evaluation: which of them JM looks at, and in what order, is replayed on the results (a point JM skips only saves it work: a distortion
  // that is cut short is never accepted)
  int n = 0;
  for (int pos = start; pos < 10; pos++) { L.cx[n] = p4x + *mvx + pts[pos][0]; L.cy[n] = p4y + *mvy + pts[pos][1]; n++; }
  eval_dist(B.planes, metric, B.t8, umv, B.wp, B.wpw, B.wpo, B.mb_x, B.mb_y, B.bsx, B.bsy, n);
  int best = 0, second_pos = 0, second = INT_MAX;
  for (int pos = start; pos < 5; pos++) {
    const int mcost = mvc(lam, *mvx + pts[pos][0], *mvy + pts[pos][1]) + L.dist[pos - start];
    if (mcost < *min_mcost) { second = *min_mcost; second_pos = best; *min_mcost = mcost; best = pos; }
    else if (mcost < second) { second = mcost; second_pos = pos; }
  }
  if (best == 0 && B.pmx == *mvx && (B.pmy - *mvy) == 0 && *min_mcost < subthres) return 1;
  int sp = 5, ep = 9;
  if (best != 0 && second_pos != 0) {
    switch (best ^ second_pos) { case 1: sp = 6; ep = 7; break; case 3: sp = 5; ep = 6; break; case 5: sp = 8; ep = 9; break; case 7: sp = 7; ep = 8; break; default: break; }
  } else {
    switch (best + second_pos) { case 0: if (half) { sp = 5; ep = 5; } break; case 1: sp = 8; ep = 10; break; case 2: sp = 5; ep = 7; break; case 5: sp = 6; ep = 8; break; case 7: sp = 7; ep = 9; break; default: break; }
  }
  if (best != 0 || (iabs(B.pmx - *mvx) + iabs(B.pmy - *mvy))) {
    for (int pos = sp; pos < ep; pos++) {
      int mcost = mvc(lam, *mvx + pts[pos][0], *mvy + pts[pos][1]);
      if (mcost >= *min_mcost) continue;
      mcost += L.dist[pos - start];
      if (mcost < *min_mcost) { *min_mcost = mcost; best = pos; }
    }
  }
  if (best) { *mvx += pts[best][0]; *mvy += pts[best][1]; }
  return 0;
}

// EPZSSubPelBlockMotionSearch, me_epzs.c:2390
__device__ int epzs_subpel(int *mvx, int *mvy, int min_mcost)
{
  const int start_hp = D.p.metric[0] != D.p.metric[1] ? 0 : 1, start_qp = D.p.metric[1] != D.p.metric[2] ? 0 : 1;
  const int max_x4 = (D.W - B.bsx + 2 * JMHIP_PAD) << 2, max_y4 = (D.H - B.bsy + 2 * JMHIP_PAD) << 2;
  int p4x = padq(B.pic_x, *mvx), p4y = padq(B.pic_y, *mvy);
  int umv = !((p4x > 1) && (p4x < max_x4 - 1) && (p4y > 1) && (p4y < max_y4 - 1));
  if (epzs_sub_level(1, start_hp, D.p.lambda_mf[1], D.p.metric[1], umv, mvx, mvy, &min_mcost, D.p.epzs_thres[3][B.bt])) return min_mcost;
  if (!start_qp) min_mcost = INT_MAX;
  p4x = padq(B.pic_x, *mvx); p4y = padq(B.pic_y, *mvy);
  umv = !((p4x > 0) && (p4x < max_x4) && (p4y > 0) && (p4y < max_y4));
  (void)epzs_sub_level(0, start_qp, D.p.lambda_mf[2], D.p.metric[2], umv, mvx, mvy, &min_mcost, D.p.epzs_thres[3][B.bt]);
  return min_mcost;
}

// ---------------------------------------------------------------------------------------------- UMHexagonS (me_umhex.c)

__constant__ int8_t c_ind_bt[8] = {0, 0, 1, 1, 2, 4, 4, 5};
__constant__ int8_t c_dia_x[4] = {-1, 0, 1, 0}, c_dia_y[4] = {0, 1, 0, -1};
__constant__ int8_t c_hex_x[6] = {2, 1, -1, -2, -1, 1}, c_hex_y[6] = {0, -2, -2, 0, 2, 2};
__constant__ int8_t c_bhx[16] = {0, -2, -4, -4, -4, -4, -4, -2, 0, 2, 4, 4, 4, 4, 4, 2}, c_bhy[16] = {4, 3, 2, 1, 0, -1, -2, -3, -4, -3, -2, -1, 0, 1, 2, 3};

struct UmState { int cx, cy, R, best_x, best_y, min_mcost, umv; };     // centre / best as VECTORS (pels, relative to the block position)

// a group of candidates L.qx/qy[0..n) through SEARCH_ONE_PIXEL (me_umhex.h:32-51): in range, not visited, mv cost below the minimum -> evaluated,
// marked visited, accepted on strict <. The positions of a group never depend on the outcome inside the group.
// The wave-wide form of a group of pairwise distinct positions (n <= 64): lane <-> candidate. JM's scan -- skip when the mv cost alone reaches the
// minimum, else add the distortion, accept on strict < -- ends on the FIRST candidate in order that attains the smallest total below the incoming
// minimum, because a skipped candidate's total could not have been smaller; that is a wave minimum plus a ballot. Every evaluated candidate is
// marked visited: JM leaves the skipped ones unmarked, but the running minimum only falls, so they would be skipped again wherever they recur.
__device__ void um_group_wave(UmState &U, int n)
{
  const int lane = threadIdx.x;
  int vx = 0, vy = 0;
  bool ok = false;
  if (lane < n) {
    vx = L.qx[lane]; vy = L.qy[lane];
    ok = iabs(vx - U.cx) <= U.R && iabs(vy - U.cy) <= U.R && !map_test(U.R, vx - U.cx, vy - U.cy);
  }
  const unsigned long long bal = __ballot(ok);
  const int m = __popcll(bal);
  if (!m) return;
  const int idx = __popcll(bal & ((1ull << lane) - 1ull));
  if (ok) { L.cx[idx] = padq(B.pic_x, vx << 2); L.cy[idx] = padq(B.pic_y, vy << 2); }
  __syncthreads();
  eval_dist(B.planes, D.p.metric[0], B.t8, U.umv, B.wp, B.wpw, B.wpo, B.mb_x, B.mb_y, B.bsx, B.bsy, m);
  int cost = INT_MAX;
  if (ok) {
    cost = mvc(D.p.lambda_mf[0], vx << 2, vy << 2) + L.dist[idx];
    const int i = (vy - U.cy + U.R) * (2 * U.R + 1) + (vx - U.cx + U.R);
    atomicOr(&L.map[i >> 5], 1u << (i & 31));
  }
  int mn = cost;
  for (int o = 1; o < 64; o <<= 1) mn = min(mn, __shfl_xor(mn, o));
  if (mn < U.min_mcost) {
    const int src = __ffsll((long long)__ballot(ok && cost == mn)) - 1;
    U.best_x = __shfl(vx, src); U.best_y = __shfl(vy, src); U.min_mcost = mn;
  }
  __syncthreads();
}

__device__ void um_group(UmState &U, int n, bool distinct = false)
{
  if (distinct && n <= 64) { um_group_wave(U, n); return; }
  int m = 0;
  for (int k = 0; k < n; k++) {
    const int vx = L.qx[k], vy = L.qy[k];
    if (iabs(vx - U.cx) > U.R || iabs(vy - U.cy) > U.R) continue;
    if (map_test(U.R, vx - U.cx, vy - U.cy)) continue;
    if (!distinct) {
      bool dup = false;                                    // the same position twice in one group: the second sees what the first left
      for (int j = 0; j < m; j++) dup |= (L.px[j] == vx && L.py[j] == vy);
      if (dup) continue;
    }
    L.px[m] = vx; L.py[m] = vy; m++;
  }
  for (int k = 0; k < m; k++) { L.cx[k] = padq(B.pic_x, L.px[k] << 2); L.cy[k] = padq(B.pic_y, L.py[k] << 2); }
  if (m) eval_dist(B.planes, D.p.metric[0], B.t8, U.umv, B.wp, B.wpw, B.wpo, B.mb_x, B.mb_y, B.bsx, B.bsy, m);
  // replay in the group's own order (duplicates re-tested against the map as it evolves). A group of pairwise distinct positions (the fixed
  // patterns) passes the two filters in the replay exactly where it passed them above -- nothing replayed here sets another position's bit --,
  // so its distortions are taken in order instead of being looked up
  int next = 0;
  for (int k = 0; k < n; k++) {
    const int vx = L.qx[k], vy = L.qy[k];
    if (iabs(vx - U.cx) > U.R || iabs(vy - U.cy) > U.R) continue;
    if (map_test(U.R, vx - U.cx, vy - U.cy)) continue;
    const int jd = next++;
    int mcost = mvc(D.p.lambda_mf[0], vx << 2, vy << 2);
    if (mcost < U.min_mcost) {
      int j = jd;
      if (!distinct) { j = 0; while (!(L.px[j] == vx && L.py[j] == vy)) j++; }
      mcost += L.dist[j];
      map_set(U.R, vx - U.cx, vy - U.cy);
      if (mcost < U.min_mcost) { U.best_x = vx; U.best_y = vy; U.min_mcost = mcost; }
    }
  }
}
// The multi-hexagon grid (me_umhex.c:474-494): nr rings of 16 points round (ix, iy), an early-termination test after every ring. The positions do
// not depend on any outcome and are pairwise distinct, so all rings are evaluated in ONE batch and replayed ring by ring; a ring past the
// terminating one is never replayed, i.e. never accepted -- its distortions were computed for nothing, as JM's partial sums are. Its positions
// ARE marked visited below (every evaluated candidate is), which JM would not have done: harmless only because the early-termination exit is
// `goto terminate_step` (me_umhex.c:488) -- no pattern runs after it, so the map is never read again. A pattern added behind the rings would
// need the marking moved into the per-ring replay.
// Returns true when the termination threshold stopped the search.
__device__ bool um_rings(UmState &U, int ix, int iy, int nr, int et)
{
  // lane <-> candidates lane and lane + 64 (ring-major order); see um_group_wave for the equivalence with JM's scan
  const int lane = threadIdx.x, n = 16 * nr;
  int vx[2], vy[2], idx[2], cost[2] = {INT_MAX, INT_MAX};
  bool ok[2];
  int base = 0;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int k = lane + 64 * h, i = (k >> 4) + 1;
    vx[h] = ix + c_bhx[k & 15] * i; vy[h] = iy + c_bhy[k & 15] * i;
    ok[h] = k < n && iabs(vx[h] - U.cx) <= U.R && iabs(vy[h] - U.cy) <= U.R && !map_test(U.R, vx[h] - U.cx, vy[h] - U.cy);
    const unsigned long long bal = __ballot(ok[h]);
    idx[h] = base + __popcll(bal & ((1ull << lane) - 1ull));
    base += __popcll(bal);
    if (ok[h]) { L.cx[idx[h]] = padq(B.pic_x, vx[h] << 2); L.cy[idx[h]] = padq(B.pic_y, vy[h] << 2); }
  }
  if (base) {
    __syncthreads();
    eval_dist(B.planes, D.p.metric[0], B.t8, U.umv, B.wp, B.wpw, B.wpo, B.mb_x, B.mb_y, B.bsx, B.bsy, base);
#pragma unroll
    for (int h = 0; h < 2; h++)
      if (ok[h]) {
        cost[h] = mvc(D.p.lambda_mf[0], vx[h] << 2, vy[h] << 2) + L.dist[idx[h]];
        const int i = (vy[h] - U.cy + U.R) * (2 * U.R + 1) + (vx[h] - U.cx + U.R);
        atomicOr(&L.map[i >> 5], 1u << (i & 31));
      }
  }
  bool stop = false;
  for (int r = 0; r < nr && !stop; r++) {                  // ring r + 1: candidates 16 r .. 16 r + 15 = lanes (16 r) & 63 .. of half (16 r) >> 6
    const int h = (16 * r) >> 6, l0 = (16 * r) & 63;
    const bool mine = lane >= l0 && lane < l0 + 16;
    const int c = mine ? (h ? cost[1] : cost[0]) : INT_MAX;
    int mn = c;
    for (int o = 1; o < 64; o <<= 1) mn = min(mn, __shfl_xor(mn, o));
    if (mn < U.min_mcost) {
      const int src = __ffsll((long long)__ballot(mine && c == mn)) - 1;
      U.best_x = __shfl(h ? vx[1] : vx[0], src); U.best_y = __shfl(h ? vy[1] : vy[0], src); U.min_mcost = mn;
    }
    stop = U.min_mcost < et;
  }
  __syncthreads();
  return stop;
}
__device__ __forceinline__ void um_diamond(UmState &U)
{
  for (int m = 0; m < 4; m++) { L.qx[m] = U.best_x + c_dia_x[m]; L.qy[m] = U.best_y + c_dia_y[m]; }
  um_group(U, 4, true);
}

// UMHEXIntegerPelBlockMotionSearch, me_umhex.c:229. mvx/mvy: centre in, result out (pels). R: the (dynamic) search range of this block.
__device__ int umhex_pel(int R, int *mvx, int *mvy, int min_mcost)
{
  const jmhip_slice_params &P = D.p;
  const int bt = B.bt, ref = B.ref, block_x = B.mb_x >> 2, block_y = B.mb_y >> 2, px2l = block_x;
  UmState U{*mvx, *mvy, R, 0 - B.pic_x, 0 - B.pic_y, min_mcost, 0};             // best_x = best_y = 0 as ABSOLUTE positions (:254)
  {
    const int center_x = B.pic_x + U.cx, center_y = B.pic_y + U.cy;
    U.umv = !((center_x > R) && (center_x < D.W - 1 - R - B.bsx) && (center_y > R) && (center_y < D.H - 1 - R - B.bsy));
  }
  int et = P.umhex_thres[0][bt];
  map_clear(R);
  {                                                                              // the centre :324-337 (always evaluated)
    L.cx[0] = padq(B.pic_x, U.cx << 2); L.cy[0] = padq(B.pic_y, U.cy << 2);
    eval_dist(B.planes, P.metric[0], B.t8, U.umv, B.wp, B.wpw, B.wpo, B.mb_x, B.mb_y, B.bsx, B.bsy, 1);
    const int mcost = mvc(P.lambda_mf[0], U.cx << 2, U.cy << 2) + L.dist[0];
    map_set(R, 0, 0);
    if (mcost < U.min_mcost) { U.min_mcost = mcost; U.best_x = U.cx; U.best_y = U.cy; }
  }
  um_diamond(U);
  if (U.cx != 0 || U.cy != 0) { L.qx[0] = 0; L.qy[0] = 0; um_group(U, 1); um_diamond(U); }
  int pred_sad = 0;
  bool done = false;
  if (ref > 0 && U.min_mcost > et && L.um_best_cost[bt][px2l] < P.umhex_thres[2][bt]) done = true;     // :365
  if (!done && U.min_mcost < et) done = true;
  if (!done) {
    // UMHEX_setup :797 (frame pictures, P slice, no intra macroblocks: flag_intra_SAD = 0)
    const short (*am)[8][2] = L.all_mv[block_y * 4 + block_x];
    int upx = 0, upy = 0, prx = 0, pry = 0, pred_ref_flag = 0, tbt = 0;
    if (bt > 1) { tbt = c_ind_bt[bt]; upx = am[ref][tbt][0]; upy = am[ref][tbt][1]; }
    if (ref > 0) {
      prx = (int)__fdiv_rn((float)((int)am[ref - 1][bt][0] * (ref + 1)), (float)ref);          // :842-845: int product, float division, truncation
      pry = (int)__fdiv_rn((float)((int)am[ref - 1][bt][1] * (ref + 1)), (float)ref);
      pred_ref_flag = 1;
    }
    if (ref > 0) pred_sad = L.um_ref_cost[ref - 1][bt][block_y * 4 + block_x];
    else if (bt > 1) pred_sad = UMLOC(tbt, B.pic_y >> 2, B.pic_x >> 2) / 2;
    else pred_sad = 0;
    et = P.umhex_thres[1][bt];
    float b1 = 0.f, b2 = 0.f;
    if (pred_sad != 0) {                                                         // :382-392, single precision, no contraction
      const float q = __fdiv_rn(P.umhex_bsize[bt], (float)(pred_sad * pred_sad));
      b1 = __fsub_rn(q, P.umhex_alpha1[bt]); b2 = __fsub_rn(q, P.umhex_alpha2[bt]);
    }
#define EARLY(lbl2, lbl1) { const float lhs = (float)(U.min_mcost - pred_sad); if (lhs < __fmul_rn((float)pred_sad, b2)) goto lbl2; else if (lhs < __fmul_rn((float)pred_sad, b1)) goto lbl1; }
    {
      int n = 0;
      // the two start-point candidates are separate SEARCH_ONE_PIXELs in JM; they are independent of each other's outcome only through
      // the minimum, which the replay handles
      if (bt > 1) { L.qx[n] = upx / 4; L.qy[n] = upy / 4; n++; }
      if (pred_ref_flag) { L.qx[n] = prx / 4; L.qy[n] = pry / 4; n++; }
      if (n) um_group(U, n);
    }
    um_diamond(U);
    EARLY(fourth_2, fourth_1)
    if (bt > 6) goto fourth_1;
    {                                                                            // unsymmetrical cross :430-453
      const int ix = U.best_x, iy = U.best_y;
      int n = 0;
      for (int i = 1; i < R; i += 2) { L.qx[n] = ix + i; L.qy[n] = iy; n++; L.qx[n] = ix - i; L.qy[n] = iy; n++; }
      for (int i = 1; i < (R / 2); i += 2) { L.qx[n] = ix; L.qy[n] = iy + i; n++; L.qx[n] = ix; L.qy[n] = iy - i; n++; }
      um_group(U, n, true);
    }
    EARLY(fourth_2, fourth_1)
    {
      const int ix = U.best_x, iy = U.best_y;
      for (int pos = 1; pos < 25; pos++) { int dx, dy; spiral_offset(pos, &dx, &dy); L.qx[pos - 1] = ix + dx; L.qy[pos - 1] = iy + dy; }
      um_group(U, 24, true);
      EARLY(fourth_2, fourth_1)
      if (um_rings(U, ix, iy, R / 4, et)) goto terminate;                        // multi-hexagon grid :475-494
    }
  fourth_1:
    for (int i = 0; i < R; i++) {
      const int ix = U.best_x, iy = U.best_y;
      for (int m = 0; m < 6; m++) { L.qx[m] = ix + c_hex_x[m]; L.qy[m] = iy + c_hex_y[m]; }
      um_group(U, 6, true);
      if (U.best_x == ix && U.best_y == iy) break;
    }
  fourth_2:
    for (int i = 0; i < R; i++) {
      const int ix = U.best_x, iy = U.best_y;
      um_diamond(U);
      if (U.best_x == ix && U.best_y == iy) break;
    }
#undef EARLY
  }
terminate:
  // :534-554
  for (int i = 0; i < (B.bsx >> 2); i++) for (int j = 0; j < (B.bsy >> 2); j++) {
    L.um_ref_cost[ref][bt][(block_y + j) * 4 + block_x + i] = U.min_mcost;
    if (ref == 0) UMLOC(bt, (B.pic_y >> 2) + j, (B.pic_x >> 2) + i) = U.min_mcost;
  }
  if (ref == 0 || L.um_best_cost[bt][px2l] > U.min_mcost) L.um_best_cost[bt][px2l] = U.min_mcost;
  *mvx = U.best_x; *mvy = U.best_y;
  return U.min_mcost;
}

// UMHEXSubPelBlockMotionSearch, me_umhex.c:562 (quarter-pel metric and lambda for every position)
__device__ int umhex_subpel(int *mvx, int *mvy, int min_mcost)
{
  const int start_hp = D.p.metric[0] != D.p.metric[1] ? 0 : 1, lam = D.p.lambda_mf[2], metric = D.p.metric[2];
  const int p4x = padq(B.pic_x, 0), p4y = padq(B.pic_y, 0);
  const short max_x4 = (short)((D.W - B.bsx + 2 * JMHIP_PAD) << 2), max_y4 = (short)((D.H - B.bsy + 2 * JMHIP_PAD) << 2);      // :590-591: short
  const int umv = !((p4x + *mvx > 1) && (p4x + *mvx < max_x4 - 1) && (p4y + *mvy > 1) && (p4y + *mvy < max_y4 - 1));
  const int cx0 = *mvx, cy0 = *mvy, pfx = (B.pmx - cx0) % 4, pfy = (B.pmy - cy0) % 4;
  int cur_x = 0, cur_y = 0;
  for (int i = 0; i < 49; i++) L.um_sstate[i] = 0;
#define SS(x, y) L.um_sstate[((y) - cy0 + 3) * 7 + ((x) - cx0 + 3)]
  {
    int n = 0;
    if (!start_hp) { L.cx[n] = p4x + cx0; L.cy[n] = p4y + cy0; n++; }
    if (pfx != 0 || pfy != 0) { L.cx[n] = p4x + cx0 + pfx; L.cy[n] = p4y + cy0 + pfy; n++; }
    if (n) eval_dist(B.planes, metric, B.t8, umv, B.wp, B.wpw, B.wpo, B.mb_x, B.mb_y, B.bsx, B.bsy, n);
    int k = 0;
    if (!start_hp) {
      const int mcost = mvc(lam, cx0, cy0) + L.dist[k++];
      SS(cx0, cy0) = 1;
      if (mcost < min_mcost) { min_mcost = mcost; cur_x = cx0; cur_y = cy0; }
    } else { SS(cx0, cy0) = 1; cur_x = cx0; cur_y = cy0; }
    if (pfx != 0 || pfy != 0) {
      const int mcost = mvc(lam, cx0 + pfx, cy0 + pfy) + L.dist[k++];
      SS(cx0 + pfx, cy0 + pfy) = 1;
      if (mcost < min_mcost) { min_mcost = mcost; cur_x = cx0 + pfx; cur_y = cy0 + pfy; }
    }
  }
  int ix = cur_x, iy = cur_y;
  for (int i = 0; i < 3; i++) {
    int n = 0, idx[4];
    for (int m = 0; m < 4; m++) {
      const int vx = ix + c_dia_x[m], vy = iy + c_dia_y[m];
      idx[m] = -1;
      if (iabs(vx - cx0) <= 3 && iabs(vy - cy0) <= 3 && !SS(vx, vy)) { L.cx[n] = p4x + vx; L.cy[n] = p4y + vy; idx[m] = n++; }
    }
    if (n) eval_dist(B.planes, metric, B.t8, umv, B.wp, B.wpw, B.wpo, B.mb_x, B.mb_y, B.bsx, B.bsy, n);
    int abort_search = 1;
    for (int m = 0; m < 4; m++) if (idx[m] >= 0) {
      const int vx = ix + c_dia_x[m], vy = iy + c_dia_y[m];
      const int mcost = mvc(lam, vx, vy) + L.dist[idx[m]];
      SS(vx, vy) = 1;
      if (mcost < min_mcost) { min_mcost = mcost; cur_x = vx; cur_y = vy; abort_search = 0; }
    }
    ix = cur_x; iy = cur_y;
    if (abort_search) break;
  }
#undef SS
  *mvx = cur_x; *mvy = cur_y;
  return min_mcost;
}


// ---------------------------------------------------------------------------------------------- simplified UMHexagonS (me_umhexsmp.c, SearchMode 2)

__constant__ int8_t c_smp_dia_x[4] = {-1, 1, 0, 0}, c_smp_dia_y[4] = {0, 0, -1, 1};
__constant__ int8_t c_smp_hex_x[6] = {-2, 2, -1, 1, -1, 1}, c_smp_hex_y[6] = {0, 0, -2, 2, 2, -2};
__constant__ int8_t c_smp_bhx[16] = {-4, 4, 0, 0, -4, 4, -4, 4, -4, 4, -4, 4, -2, 2, -2, 2}, c_smp_bhy[16] = {0, 0, -4, 4, -1, 1, 1, -1, -2, 2, 2, -2, -3, 3, 3, -3};
__constant__ int8_t c_smp_shift[8] = {0, 0, 1, 1, 2, 3, 3, 1};                  // block_type_shift_factor :44

// a group of candidates L.qx/qy[0..n) through SEARCH_ONE_PIXEL_HELPER (:56-69): in range -> evaluated, accepted on strict <. No visited map,
// no mv-cost shortcut; the positions of a group never depend on the outcome inside the group.
__device__ void smp_group(UmState &U, int n)
{
  int m = 0;
  for (int k = 0; k < n; k++) {
    const int vx = L.qx[k], vy = L.qy[k];
    if (iabs(vx - U.cx) > U.R || iabs(vy - U.cy) > U.R) continue;
    L.px[m] = vx; L.py[m] = vy;
    L.cx[m] = padq(B.pic_x, vx << 2); L.cy[m] = padq(B.pic_y, vy << 2); m++;
  }
  if (m) eval_dist(B.planes, D.p.metric[0], B.t8, U.umv, B.wp, B.wpw, B.wpo, B.mb_x, B.mb_y, B.bsx, B.bsy, m);
  for (int k = 0; k < m; k++) {
    const int mcost = mvc(D.p.lambda_mf[0], L.px[k] << 2, L.py[k] << 2) + L.dist[k];
    if (mcost < U.min_mcost) { U.best_x = L.px[k]; U.best_y = L.py[k]; U.min_mcost = mcost; }
  }
}
__device__ __forceinline__ void smp_diamond(UmState &U, int ix, int iy)
{
  for (int m = 0; m < 4; m++) { L.qx[m] = ix + c_smp_dia_x[m]; L.qy[m] = iy + c_smp_dia_y[m]; }
  smp_group(U, 4);
}

// smpUMHEXIntegerPelBlockMotionSearch :152. mvx/mvy: centre in, result out (pels); (upx, upy): smpUMHEX_pred_MV_uplayer (quarter-pel)
__device__ int smp_pel(int R, int *mvx, int *mvy, int min_mcost, int upx, int upy)
{
  const jmhip_slice_params &P = D.p;
  const int bt = B.bt, sf = c_smp_shift[bt];
  UmState U{*mvx, *mvy, R, 0 - B.pic_x, 0 - B.pic_y, min_mcost, 0};
  {
    const int center_x = B.pic_x + U.cx, center_y = B.pic_y + U.cy;
    U.umv = !((center_x > R) && (center_x < D.W - 1 - R - B.bsx) && (center_y > R) && (center_y < D.H - 1 - R - B.bsy));
  }
  {                                                                              // the centre :241-252 (no range test)
    L.cx[0] = padq(B.pic_x, U.cx << 2); L.cy[0] = padq(B.pic_y, U.cy << 2);
    eval_dist(B.planes, P.metric[0], B.t8, U.umv, B.wp, B.wpw, B.wpo, B.mb_x, B.mb_y, B.bsx, B.bsy, 1);
    const int mcost = mvc(P.lambda_mf[0], U.cx << 2, U.cy << 2) + L.dist[0];
    if (mcost < U.min_mcost) { U.min_mcost = mcost; U.best_x = U.cx; U.best_y = U.cy; }
  }
  int ix = U.best_x, iy = U.best_y;
  if (B.pmx != 0 || B.pmy != 0) { L.qx[0] = 0; L.qy[0] = 0; smp_group(U, 1); }
  if (U.min_mcost < (1000 >> sf)) {                                              // ConvergeThreshold :264-274
    smp_diamond(U, ix, iy);
    *mvx = U.best_x; *mvy = U.best_y;
    return U.min_mcost;
  }
  smp_diamond(U, ix, iy);
  if ((bt == 1 && U.min_mcost > (800 >> sf)) || U.min_mcost > (7000 >> sf)) {    // SymmetricalCrossSearchThreshold1 / 2 :287-330
    ix = U.best_x; iy = U.best_y;
    int n = 0;
    for (int i = 1; i <= R / 2; i++) {
      const int st = (i << 1) - 1;
      L.qx[n] = ix + st; L.qy[n] = iy; n++; L.qx[n] = ix - st; L.qy[n] = iy; n++;
      L.qx[n] = ix; L.qy[n] = iy + st; n++; L.qx[n] = ix; L.qy[n] = iy - st; n++;
    }
    smp_group(U, n);
    ix = U.best_x; iy = U.best_y;
    for (int m = 0; m < 6; m++) { L.qx[m] = ix + c_smp_hex_x[m]; L.qy[m] = iy + c_smp_hex_y[m]; }
    smp_group(U, 6);
    ix = U.best_x; iy = U.best_y;
    n = 0;
    for (int i = 1; i <= R / 4; i++) for (int m = 0; m < 16; m++) { L.qx[n] = ix + c_smp_bhx[m] * i; L.qy[n] = iy + c_smp_bhy[m] * i; n++; }
    smp_group(U, n);
  }
  if (bt > 1) { L.qx[0] = upx / 4; L.qy[0] = upy / 4; smp_group(U, 1); }         // :333-338
  if (U.cx != 0 || U.cy != 0) {
    L.qx[0] = 0; L.qy[0] = 0; smp_group(U, 1);
    smp_diamond(U, U.best_x, U.best_y);
  }
  if (U.min_mcost < (1000 >> sf)) {                                              // :364-376
    smp_diamond(U, U.best_x, U.best_y);
    *mvx = U.best_x; *mvy = U.best_y;
    return U.min_mcost;
  }
  for (int i = 0; i < R; i++) {
    ix = U.best_x; iy = U.best_y;
    for (int m = 0; m < 6; m++) { L.qx[m] = ix + c_smp_hex_x[m]; L.qy[m] = iy + c_smp_hex_y[m]; }
    smp_group(U, 6);
    if (U.best_x == ix && U.best_y == iy) break;
  }
  for (int i = 0; i < R; i++) {
    ix = U.best_x; iy = U.best_y;
    smp_diamond(U, ix, iy);
    if (U.best_x == ix && U.best_y == iy) break;
  }
  *mvx = U.best_x; *mvy = U.best_y;
  return U.min_mcost;
}

// smpUMHEXFullSubPelBlockMotionSearch :422 (the 16x16 block): both refinements with the QUARTER-pel metric and lambda (:461), each leaves
// its scan as soon as the minimum is below SubPelThreshold3 (:531, :586)
__device__ int smp_full_subpel(int *mvx, int *mvy, int min_mcost)
{
  const jmhip_slice_params &P = D.p;
  const int start_hp = P.metric[0] != P.metric[1] ? 0 : 1, start_qp = P.metric[1] != P.metric[2] ? 0 : 1, lam = P.lambda_mf[2], metric = P.metric[2];
  const int sf = c_smp_shift[B.bt];
  const int check0 = (!D.p.rdopt && B.ref == 0 && B.bt == 1 && *mvx == 0 && *mvy == 0);        // !rdopt, P slice
  const int max_x4 = (D.W - B.bsx + 2 * JMHIP_PAD) << 2, max_y4 = (D.H - B.bsy + 2 * JMHIP_PAD) << 2;
  for (int phase = 0; phase < 2; phase++) {
    const int start = phase ? start_qp : start_hp, step = phase ? 1 : 2, lo = phase ? 0 : 1;
    const int p4x = padq(B.pic_x, *mvx), p4y = padq(B.pic_y, *mvy);
    const int umv = !((p4x > lo) && (p4x < max_x4 - lo) && (p4y > lo) && (p4y < max_y4 - lo));
    int n = 0;
    for (int pos = start; pos < 9; pos++) { L.cx[n] = p4x + step * c_s9x[pos]; L.cy[n] = p4y + step * c_s9y[pos]; n++; }
    eval_dist(B.planes, metric, B.t8, umv, B.wp, B.wpw, B.wpo, B.mb_x, B.mb_y, B.bsx, B.bsy, n);
    int best = 0;
    for (int pos = start, k = 0; pos < 9; pos++, k++) {
      int mcost = mvc(lam, *mvx + step * c_s9x[pos], *mvy + step * c_s9y[pos]);
      if (mcost >= min_mcost) continue;
      mcost += L.dist[k];
      if (phase == 0 && pos == 0 && check0) mcost -= (lam * 16) >> 16;
      if (mcost < min_mcost) { min_mcost = mcost; best = pos; }
      if (min_mcost < (400 >> sf)) break;
    }
    if (best) { *mvx += step * c_s9x[best]; *mvy += step * c_s9y[best]; }
    if (phase == 0) {
      if (*mvx == 0 && *mvy == 0 && B.pmx == 0 && B.pmy == 0 && min_mcost < (1000 >> sf)) return min_mcost;      // SubPelThreshold1 :547-552
      if (!start_qp) min_mcost = INT_MAX;
    }
  }
  return min_mcost;
}

// smpUMHEXSubPelBlockMotionSearch :616 (block types > 1): a diamond walk inside +-3 quarter-pels of the integer vector
__device__ int smp_subpel(int *mvx, int *mvy, int min_mcost, int upx, int upy)
{
  const int start_hp = D.p.metric[0] != D.p.metric[1] ? 0 : 1, lam = D.p.lambda_mf[2], metric = D.p.metric[2], sf = c_smp_shift[B.bt];
  const int p4x = padq(B.pic_x, 0), p4y = padq(B.pic_y, 0);
  const short max_x4 = (short)((D.W - B.bsx + 2 * JMHIP_PAD) << 2), max_y4 = (short)((D.H - B.bsy + 2 * JMHIP_PAD) << 2);      // :645-646: short
  const int umv = !((p4x + *mvx > 1) && (p4x + *mvx < max_x4 - 1) && (p4y + *mvy > 1) && (p4y + *mvy < max_y4 - 1));
  const int cx0 = *mvx, cy0 = *mvy, pfx = (B.pmx - cx0) % 4, pfy = (B.pmy - cy0) % 4, pux = (upx - cx0) % 4, puy = (upy - cy0) % 4;
  int cur_x = 0, cur_y = 0;
  for (int i = 0; i < 49; i++) L.um_sstate[i] = 0;
#define SS(x, y) L.um_sstate[((y) - cy0 + 3) * 7 + ((x) - cx0 + 3)]
  SS(cx0, cy0) = 1;
  if (!start_hp) {
    L.cx[0] = p4x + cx0; L.cy[0] = p4y + cy0;
    eval_dist(B.planes, metric, B.t8, umv, B.wp, B.wpw, B.wpo, B.mb_x, B.mb_y, B.bsx, B.bsy, 1);
    const int mcost = mvc(lam, cx0, cy0) + L.dist[0];
    if (mcost < min_mcost) { min_mcost = mcost; cur_x = cx0; cur_y = cy0; }
  } else { cur_x = cx0; cur_y = cy0; }
  if (cx0 == 0 && cy0 == 0 && pfx == 0 && pux == 0 && pfy == 0 && puy == 0 && min_mcost < (1000 >> sf)) { *mvx = cur_x; *mvy = cur_y; return min_mcost; }
  if (pfx != 0 || pfy != 0) {
    L.cx[0] = p4x + cx0 + pfx; L.cy[0] = p4y + cy0 + pfy;
    eval_dist(B.planes, metric, B.t8, umv, B.wp, B.wpw, B.wpo, B.mb_x, B.mb_y, B.bsx, B.bsy, 1);
    const int mcost = mvc(lam, cx0 + pfx, cy0 + pfy) + L.dist[0];
    SS(cx0 + pfx, cy0 + pfy) = 1;
    if (mcost < min_mcost) { min_mcost = mcost; cur_x = cx0 + pfx; cur_y = cy0 + pfy; }
  }
  for (int i = 0; i < 3; i++) {
    const int ix = cur_x, iy = cur_y;
    int n = 0, idx[4];
    for (int m = 0; m < 4; m++) {
      const int vx = ix + c_smp_dia_x[m], vy = iy + c_smp_dia_y[m];
      idx[m] = -1;
      if (iabs(vx - cx0) <= 3 && iabs(vy - cy0) <= 3 && !SS(vx, vy)) { L.cx[n] = p4x + vx; L.cy[n] = p4y + vy; idx[m] = n++; }
    }
    if (n) eval_dist(B.planes, metric, B.t8, umv, B.wp, B.wpw, B.wpo, B.mb_x, B.mb_y, B.bsx, B.bsy, n);
    int abort_search = 1;
    for (int m = 0; m < 4; m++) if (idx[m] >= 0) {
      const int vx = ix + c_smp_dia_x[m], vy = iy + c_smp_dia_y[m];
      const int mcost = mvc(lam, vx, vy) + L.dist[idx[m]];
      SS(vx, vy) = 1;
      if (mcost < min_mcost) { min_mcost = mcost; cur_x = vx; cur_y = vy; abort_search = 0; }
      if (min_mcost < (400 >> sf)) { *mvx = cur_x; *mvy = cur_y; return min_mcost; }       // SubPelThreshold3 :794-799
    }
    if (abort_search) break;
  }
#undef SS
  *mvx = cur_x; *mvy = cur_y;
  return min_mcost;
}

// ---------------------------------------------------------------------------------------------- BlockMotionSearch and above

__device__ __forceinline__ int part_index(int bt, int block_x, int block_y)
{
  const int b8 = (block_y >> 1) * 2 + (block_x >> 1);
  switch (bt) {
  case 1: return 0;
  case 2: return 1 + (block_y >> 1);
  case 3: return 3 + (block_x >> 1);
  case 4: return 5 + b8;
  case 5: return 9 + b8 * 2 + (block_y & 1);
  case 6: return 17 + b8 * 2 + (block_x & 1);
  default: return 25 + b8 * 4 + (block_y & 1) * 2 + (block_x & 1);
  }
}

// FindSkipModeMotionVector, mv-search.c:1189
__device__ void find_skip_mv(int mbx, int mby)
{
  int ax, ay, bx, by, pmx = 0, pmy = 0;
  const int availA = nbr_pos(mbx, mby, -1, 0, &ax, &ay), availB = nbr_pos(mbx, mby, 0, -1, &bx, &by);
  int zl = 1, za = 1;
  if (availA) zl = (FREF(ay, ax) == 0 && FMV(ay, ax, 0) == 0 && FMV(ay, ax, 1) == 0);
  if (availB) za = (FREF(by, bx) == 0 && FMV(by, bx, 0) == 0 && FMV(by, bx, 1) == 0);
  if (!(za || zl)) mv_predictor(mbx, mby, 0, 0, 0, 16, 16, &pmx, &pmy, 0, 0, nullptr, nullptr);
  for (int b = 0; b < 16; b++) { L.all_mv[b][0][0][0] = (short)pmx; L.all_mv[b][0][0][1] = (short)pmy; }
}

// BlockMotionSearch, mv-search.c:560 (P slice, rdopt 0)
// SM: the slice's search mode as a compile-time constant -- one instantiation per mode keeps the other modes' walkers (and their registers)
// out of this function, which is called 41 times per macroblock and reference
template <int SM> __device__ int block_motion_search(int mbx, int mby, int ref, int mb_x, int mb_y, int bt, int search_range, jmhip_mb_inter *out)
{
  const jmhip_slice_params &P = D.p;
  B.mbx = mbx; B.mby = mby; B.mb_x = mb_x; B.mb_y = mb_y; B.bt = bt; B.bsx = c_bsx[bt]; B.bsy = c_bsy[bt]; B.ref = ref;
  B.pic_x = mbx * 16 + mb_x; B.pic_y = mby * 16 + mb_y;
  B.planes = D.ref_sub[P.ref_slot[ref]];
  B.wp = P.wp_me; B.wpw = P.wp_weight[ref]; B.wpo = P.wp_offset[ref];
  B.t8 = P.transform8x8_mode && bt <= 4;                                       // test8x8transform, mv-search.c:640
  const int block_x = mb_x >> 2, block_y = mb_y >> 2, pi = part_index(bt, block_x, block_y);
  const int start_hp = P.metric[0] != P.metric[1] ? 0 : 1;
  int mvx, mvy, min_mcost = INT_MAX, smp_upx = 0, smp_upy = 0;
  WPROF_T0;
  mv_predictor(mbx, mby, ref, mb_x, mb_y, B.bsx, B.bsy, &B.pmx, &B.pmy, SM == JMHIP_SEARCH_UMHEX && P.umhex_dsr, bt, &search_range, nullptr);
  WPROF(0);
  const int R = search_range;
  if ((SM == JMHIP_SEARCH_FULL || SM == JMHIP_SEARCH_FASTFULL) && L.memo_live) {
    const bool same = ((L.memo_old[ref] >> pi) & 1) && out->pred[ref][pi][0] == B.pmx && out->pred[ref][pi][1] == B.pmy;
    if (bt == 1) L.ff_same[ref] = same;                                       // (one wave: every lane stores the same value)
    else if (same && (SM == JMHIP_SEARCH_FULL || L.ff_same[ref])) {
      mvx = out->mv[ref][pi][0]; mvy = out->mv[ref][pi][1]; min_mcost = out->cost[ref][pi];
      for (int j = block_y; j < block_y + (B.bsy >> 2); j++) for (int i = block_x; i < block_x + (B.bsx >> 2); i++) { L.all_mv[j * 4 + i][ref][bt][0] = (short)mvx; L.all_mv[j * 4 + i][ref][bt][1] = (short)mvy; }
      if (threadIdx.x == 0) L.memo_new[ref] |= 1ull << pi;
      return min_mcost;
    }
  }
  if (D.debug & 2) { mvx = clampi((B.pmx + 2) >> 2, -R, R); mvy = clampi((B.pmy + 2) >> 2, -R, R); min_mcost = 1000; }
  else if (SM == JMHIP_SEARCH_UMHEX) {
    mvx = B.pmx / 4; mvy = B.pmy / 4;
    mvx = clampi(mvx, -R, R); mvy = clampi(mvy, -R, R);
    mvx = clampi(mvx, -2047 + R, 2047 - R); mvy = clampi(mvy, P.level_mv_min + R, P.level_mv_max - R);
    min_mcost = umhex_pel(R, &mvx, &mvy, min_mcost);
    __syncthreads();                                                           // lane 0 wrote the cost map
  } else if (SM == JMHIP_SEARCH_UMHEX_SIMPLE) {                                  // mv-search.c:674-706; smpUMHEX_setup :634: the upper layer's vector
    const int ub = bt > 6 ? 5 : bt > 4 ? 4 : bt == 4 ? 2 : 1;
    smp_upx = L.all_mv[block_y * 4 + block_x][ref][ub][0]; smp_upy = L.all_mv[block_y * 4 + block_x][ref][ub][1];
    mvx = B.pmx / 4; mvy = B.pmy / 4;
    if (!P.rdopt) { mvx = clampi(mvx, -R, R); mvy = clampi(mvy, -R, R); }
    mvx = clampi(mvx, -2047 + R, 2047 - R); mvy = clampi(mvy, P.level_mv_min + R, P.level_mv_max - R);
    min_mcost = smp_pel(R, &mvx, &mvy, min_mcost, smp_upx, smp_upy);
  } else if (SM == JMHIP_SEARCH_EPZS) {
    mvx = (B.pmx + 2) >> 2; mvy = (B.pmy + 2) >> 2;
    mvx = clampi(mvx, -R, R); mvy = clampi(mvy, -R, R);
    mvx = clampi(mvx, -2047 + R, 2047 - R); mvy = clampi(mvy, P.level_mv_min + R, P.level_mv_max - R);
    min_mcost = epzs_pel(mby * D.mbw + mbx, R, &mvx, &mvy);
    __syncthreads();                                                           // lane 0 wrote the row memories
  } else if (SM == JMHIP_SEARCH_FASTFULL) {
    // the window of SetupFastFullPelSearch (me_fullfast.c:550-566): centred on the 16x16 predictor of this reference, found when the reference's
    // first block (the 16x16) was searched and kept in motion_cost[0][ref][0..2]
    if (bt == 1) {
      int cx = B.pmx / 4, cy = B.pmy / 4;
      const int Rf = (P.full_search == 2 || ref == 0) ? P.search_range : P.search_range / 2;
      if (!P.rdopt) { cx = clampi(cx, -Rf, Rf); cy = clampi(cy, -Rf, Rf); }
      cx = clampi(cx, -2047 + Rf, 2047 - Rf); cy = clampi(cy, P.level_mv_min + Rf, P.level_mv_max - Rf);
      L.motion_cost[0][ref][0] = cx; L.motion_cost[0][ref][1] = cy; L.motion_cost[0][ref][2] = Rf;
      surface_build(cx, cy, Rf);
    }
    min_mcost = surface_search(L.motion_cost[0][ref][0], L.motion_cost[0][ref][1], L.motion_cost[0][ref][2], 1, &mvx, &mvy);
    if (min_mcost == INT_MIN) min_mcost = full_window(L.motion_cost[0][ref][0], L.motion_cost[0][ref][1], L.motion_cost[0][ref][2], 1, &mvx, &mvy);
  } else {
    int cx = B.pmx / 4, cy = B.pmy / 4;
    if (!P.rdopt) { cx = clampi(cx, -R, R); cy = clampi(cy, -R, R); }
    cx = clampi(cx, -2047 + R, 2047 - R); cy = clampi(cy, P.level_mv_min + R, P.level_mv_max - R);
    if (bt == 1) surface_build(cx, cy, min(R + SURF_MARGIN, 33 + SURF_MARGIN));     // the other partitions' centres are usually within the margin
    min_mcost = surface_search(cx, cy, R, 0, &mvx, &mvy);
    if (min_mcost == INT_MIN) {                                          // a centre the macroblock's surface does not cover: the partition's own surface
      min_mcost = surface_search(cx, cy, R, 0, &mvx, &mvy, 1);
      if (min_mcost == INT_MIN) { surface_build(cx, cy, R, 1, mb_y >> 2, (mb_y + B.bsy) >> 2); min_mcost = surface_search(cx, cy, R, 0, &mvx, &mvy, 1); }
      if (min_mcost == INT_MIN) min_mcost = full_window(cx, cy, R, 0, &mvx, &mvy);
    }
  }
  WPROF(1);
  const int b8i = (block_y >> 1) * 2 + (block_x >> 1), to8ts = L.pass8ts;
  if (threadIdx.x == 0) {
    if (to8ts) { out->mv_int8ts[ref][b8i][0] = (int16_t)mvx; out->mv_int8ts[ref][b8i][1] = (int16_t)mvy; out->cost_int8ts[ref][b8i] = min_mcost; }
    else { out->mv_int[ref][pi][0] = (int16_t)mvx; out->mv_int[ref][pi][1] = (int16_t)mvy; out->cost_int[ref][pi] = min_mcost; }
  }
  mvx <<= 2; mvy <<= 2;
  // sub-pel :781-827
  bool do_sub = true;
  if (SM == JMHIP_SEARCH_EPZS && ref > 0) do_sub = (2 * (long long)min_mcost < 7 * (long long)L.ep_sad[bt - 1][(B.pic_x >> 2) - (4 * mbx - 4)]);   // min_mcost < 3.5 * prevSad
  if (do_sub && !(D.debug & 1)) {
    if (!start_hp) min_mcost = INT_MAX;
    if (SM == JMHIP_SEARCH_UMHEX_SIMPLE) min_mcost = bt > 1 ? smp_subpel(&mvx, &mvy, min_mcost, smp_upx, smp_upy) : smp_full_subpel(&mvx, &mvy, min_mcost);
    else if (SM == JMHIP_SEARCH_UMHEX && bt > 3) min_mcost = umhex_subpel(&mvx, &mvy, min_mcost);
    else if (SM == JMHIP_SEARCH_EPZS && P.epzs_subpel_me) min_mcost = epzs_subpel(&mvx, &mvy, min_mcost);
    else min_mcost = subpel_full(&mvx, &mvy, min_mcost);
  }
  WPROF(2);
  if (bt == 1 && !P.rdopt && !(D.debug & 4)) {                                 // skip shortcut :826-849 (!rdopt), every reference
    find_skip_mv(mbx, mby);
    const int smx = L.all_mv[0][0][0][0], smy = L.all_mv[0][0][0][1];
    int cost;
    const uint8_t *pl0 = D.ref_sub[P.ref_slot[0]];
    if (P.md_metric == 2) {                                                    // with the 8x8 transform: distortion8x8 per 8x8 block, :1171-1176
      L.cx[0] = padq(mbx * 16, smx); L.cy[0] = padq(mby * 16, smy);
      eval_dist(pl0, 2, P.transform8x8_mode ? 1 : 0, 1, P.wp_pred, P.wp_weight[0], P.wp_offset[0], 0, 0, 16, 16, 1);
      cost = L.dist[0];
    } else {                                                                   // SAD / SSE: LumaPrediction clamps per 4x4 block
      cost = 0;
      for (int b = 0; b < 16; b++) {
        L.cx[0] = padq(mbx * 16 + (b & 3) * 4, smx); L.cy[0] = padq(mby * 16 + (b >> 2) * 4, smy);
        eval_dist(pl0, P.md_metric, 0, 1, P.wp_pred, P.wp_weight[0], P.wp_offset[0], (b & 3) * 4, (b >> 2) * 4, 4, 4, 1);
        cost += L.dist[0];
      }
    }
    cost -= (P.lambda_mf[2] + 4096) >> 13;
    if (cost < min_mcost) { min_mcost = cost; mvx = smx; mvy = smy; }
  }
  WPROF(3);
  for (int j = block_y; j < block_y + (B.bsy >> 2); j++) for (int i = block_x; i < block_x + (B.bsx >> 2); i++) { L.all_mv[j * 4 + i][ref][bt][0] = (short)mvx; L.all_mv[j * 4 + i][ref][bt][1] = (short)mvy; }
  if (threadIdx.x == 0) {
    if (to8ts) {
      out->pred8ts[ref][b8i][0] = (int16_t)B.pmx; out->pred8ts[ref][b8i][1] = (int16_t)B.pmy;
      out->mv8ts[ref][b8i][0] = (int16_t)mvx; out->mv8ts[ref][b8i][1] = (int16_t)mvy; out->cost8ts[ref][b8i] = min_mcost;
    } else {
      out->pred[ref][pi][0] = (int16_t)B.pmx; out->pred[ref][pi][1] = (int16_t)B.pmy;
      out->mv[ref][pi][0] = (int16_t)mvx; out->mv[ref][pi][1] = (int16_t)mvy; out->cost[ref][pi] = min_mcost;
      L.memo_new[ref] |= 1ull << pi;
    }
  }
  return min_mcost;
}

// field writes: into the macroblock's LDS copy (every lane writes the same value); mb_commit hands the final sixteen on
__device__ __forceinline__ void field_set(int by, int bx, int ref, int mvx, int mvy)
{
  FREF(by, bx) = (int8_t)ref; FMV(by, bx, 0) = (short)mvx; FMV(by, bx, 1) = (short)mvy;
}

// PartitionMotionSearch, mv-search.c:1378
template <int SM> __device__ __forceinline__ void partition_motion_search(int mbx, int mby, int bt, int block8, jmhip_mb_inter *out)
{
  const int8_t bx0[5][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 2, 0, 0}, {0, 2, 0, 2}};
  const int8_t by0[5][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 2, 0, 0}, {0, 0, 0, 0}, {0, 0, 2, 2}};
  const int8_t psz[8][2] = {{4, 4}, {4, 4}, {4, 2}, {2, 4}, {2, 2}, {2, 1}, {1, 2}, {1, 1}};
  const jmhip_slice_params &P = D.p;
  const int pt = bt < 4 ? bt : 4, sh0 = psz[pt][0], sv0 = psz[pt][1], sh = psz[bt][0], sv = psz[bt][1], by = by0[pt][block8], bx = bx0[pt][block8];
  for (int ref = 0; ref < P.num_refs; ref++) {
    int range;
    if (P.full_search == 2) range = P.search_range;
    else if (P.full_search == 1) range = P.search_range / (min(ref, 1) + 1);
    else range = P.search_range / ((min(ref, 1) + 1) * min(2, bt));
    int mc = 0;
    for (int v = by; v < by + sv0; v += sv) for (int h = bx; h < bx + sh0; h += sh) {
      mc += block_motion_search<SM>(mbx, mby, ref, h << 2, v << 2, bt, range, out);
      const int mvx = L.all_mv[v * 4 + h][ref][bt][0], mvy = L.all_mv[v * 4 + h][ref][bt][1];
      for (int j = 0; j < sv; j++) for (int i = 0; i < sh; i++) field_set(mby * 4 + v + j, mbx * 4 + h + i, ref, mvx, mvy);
      __syncthreads();
    }
    L.motion_cost[bt][ref][block8] = mc;
  }
}


// ---------------------------------------------------------------------------------------------- Transform8x8Mode (md_low.c:183-188, :203-326, :547-548)

// prediction-residual cost of a bs x bs block at (ox, oy) of the macroblock predicted from (ref, mv): the sum of distortion4x4 (t8 = 0) or the
// distortion8x8 (t8 = 1, bs = 8) of the mode-decision metric, as TransformDecision (macroblock.c:1458) / GetBestTransformP8x8 (rdopt.c:3262)
// form them. (LumaPrediction clamps per 4x4 block, the evaluation per 4x4 / 8x8 sub-block: the same samples, the padding ring is flat; the
// sequential layout of JM's diff64 permutes the index bits of the 8x8 block, which the Hadamard magnitudes' sum is invariant under.)
__device__ int pred_block_cost(int mbx, int mby, int ox, int oy, int bs, int ref, int mvx, int mvy, int t8)
{
  const jmhip_slice_params &P = D.p;
  L.cx[0] = padq(mbx * 16 + ox, mvx); L.cy[0] = padq(mby * 16 + oy, mvy);
  eval_dist(D.ref_sub[P.ref_slot[ref]], P.md_metric, t8, 1, P.wp_pred, P.wp_weight[ref], P.wp_offset[ref], ox, oy, bs, bs, 1);
  return L.dist[0];
}

__constant__ uint8_t c_t8_scan[64][2] = {                                       // SNGL_SCAN8x8 (i, j), transform8x8.c:171
  {0,0},{1,0},{0,1},{0,2},{1,1},{2,0},{3,0},{2,1},{1,2},{0,3},{0,4},{1,3},{2,2},{3,1},{4,0},{5,0},
  {4,1},{3,2},{2,3},{1,4},{0,5},{0,6},{1,5},{2,4},{3,3},{4,2},{5,1},{6,0},{7,0},{6,1},{5,2},{4,3},
  {3,4},{2,5},{1,6},{0,7},{1,7},{2,6},{3,5},{4,4},{5,3},{6,2},{7,1},{7,2},{6,3},{5,4},{4,5},{3,6},
  {2,7},{3,7},{4,6},{5,5},{6,4},{7,3},{7,4},{6,5},{5,6},{4,7},{5,7},{6,6},{7,5},{7,6},{6,7},{7,7}};
__constant__ uint8_t c_t8_cost[2][64] = {                                       // COEFF_COST8x8, transform8x8.c:197
  {3,3,3,3,2,2,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0},
  {9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9}};

__device__ __forceinline__ void t8_fwd_1d(const int p[8], int o[8])            // forward8x8, transform.c:248-275
{
  int a0 = p[0] + p[7], a1 = p[1] + p[6], a2 = p[2] + p[5], a3 = p[3] + p[4];
  const int b0 = a0 + a3, b1 = a1 + a2, b2 = a0 - a3, b3 = a1 - a2;
  a0 = p[0] - p[7]; a1 = p[1] - p[6]; a2 = p[2] - p[5]; a3 = p[3] - p[4];
  const int b4 = a1 + a2 + ((a0 >> 1) + a0), b5 = a0 - a3 - ((a2 >> 1) + a2);
  const int b6 = a0 + a3 - ((a1 >> 1) + a1), b7 = a1 - a2 + ((a3 >> 1) + a3);
  o[0] = b0 + b1; o[1] = b4 + (b7 >> 2); o[2] = b2 + (b3 >> 1); o[3] = b5 + (b6 >> 2);
  o[4] = b0 - b1; o[5] = b6 - (b5 >> 2); o[6] = (b2 >> 1) - b3; o[7] = (b4 >> 2) - b7;
}

// cbp8_8x8ts: which 8x8 blocks of the 8x8-transform P8x8 pass keep coefficients -- LumaResidualCoding8x8 (macroblock.c:1009) of each block: one
// 8x8 prediction (origin clamp, explicit weights), dct_8x8 (transform8x8.c:1452: forward transform, quantisation, coefficient cost with the
// CAVLC four-way interleave), a block whose cost is <= _LUMA_COEFF_COST_ (4) is dropped (:1223). Lanes 0..3, one block each; rare (only when
// that pass wins the macroblock), so nothing is shared.
__device__ int t8_pass_cbp(int mbx, int mby, const int *ref8, const short (*mv8)[2])
{
  const jmhip_slice_params &P = D.p;
  const int lane = threadIdx.x;
  int keep = 0;
  __syncthreads();
  if (lane < 4) {
    const int ox = 8 * (lane & 1), oy = 8 * (lane >> 1), ref = ref8[lane];
    const int cx = padq(mbx * 16 + ox, mv8[lane][0]), cy = padq(mby * 16 + oy, mv8[lane][1]);
    const int ix = clampi(cx >> 2, 0, D.Wp - 17), iy = clampi(cy >> 2, 0, D.Hp - 17);
    const uint8_t *src = D.ref_sub[P.ref_slot[ref]] + (size_t)D.Wp * D.Hp * ((cy & 3) * 4 + (cx & 3)) + (size_t)iy * D.Wp + ix;
    int t[8][8];
    for (int j = 0; j < 8; j++) {
      int p[8], o[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        int v = src[(size_t)j * D.Wp + i];
        if (P.wp_pred) v = min(max((((int)P.wp_weight[ref] * v + P.wp_round) >> P.wp_denom) + (int)P.wp_offset[ref], 0), 255);
        p[i] = (int)L.cur[oy + j][ox + i] - v;
      }
      t8_fwd_1d(p, o);
#pragma unroll
      for (int i = 0; i < 8; i++) t[j][i] = o[i];
    }
    for (int i = 0; i < 8; i++) {
      int p[8], o[8];
#pragma unroll
      for (int j = 0; j < 8; j++) p[j] = t[j][i];
      t8_fwd_1d(p, o);
#pragma unroll
      for (int j = 0; j < 8; j++) t[j][i] = o[j];
    }
    const int q_bits = 16 + P.t8_qp / 6;
    int run = -1, runs4[4] = {-1, -1, -1, -1}, cost = 0, nonzero = 0;
    for (int k = 0; k < 64; k++) {
      const int i = c_t8_scan[k][0], j = c_t8_scan[k][1];
      run++; runs4[k & 3]++;
      const int level = (iabs(t[j][i]) * P.t8_levelscale[j * 8 + i] + P.t8_leveloffset[j * 8 + i]) >> q_bits;
      if (level != 0) {
        nonzero = 1;
        if (P.t8_cavlc) { cost += level > 1 ? 999999 : c_t8_cost[P.t8_disthres][runs4[k & 3]]; runs4[k & 3] = -1; }
        else { cost += level > 1 ? 999999 : c_t8_cost[P.t8_disthres][run]; run = -1; }
      }
    }
    keep = nonzero && cost > 4;
  }
  const int cbp = (int)(__ballot(keep) & 15ull);
  __syncthreads();
  return cbp;
}

__device__ __forceinline__ int list0_cost(int mode, int block, int *best_ref)
{
  int best = INT_MAX;
  for (int ref = 0; ref < D.p.num_refs; ref++) {
    const int mcost = (ref ? D.p.ref_cost1 : 0) + L.motion_cost[mode][ref][block];
    if (mcost < best) { best = mcost; *best_ref = ref; }
  }
  return best;
}
__device__ __forceinline__ int refbits(int r) { return r == 0 ? 1 : (r < 3 ? 3 : 5); }       // mv-search.c:344-352, r <= 6

// encode_one_macroblock_low (md_low.c:46), inter part, for one macroblock
template <int SM> __device__ void macroblock_low(int mbx, int mby, jmhip_mb_inter *out)
{
  const jmhip_slice_params &P = D.p;
  const int8_t psz[8][2] = {{4, 4}, {4, 4}, {4, 2}, {2, 4}, {2, 2}, {2, 1}, {1, 2}, {1, 1}};
  const int bx0 = mbx * 4, by0 = mby * 4;
  // (every lane executes the same control flow and writes the same values: one wave, LDS operations in order)
  int &best_mode = L.dec.best_mode, &min_cost = L.dec.min_cost, &t8_flag = L.dec.t8_flag, &best_tflag = L.dec.best_tflag, &cbp8ts = L.dec.cbp8ts, &tr8_cost = L.dec.tr8_cost, &tr4_cost = L.dec.tr4_cost;
  int (&l0ref)[5][4] = L.dec.l0ref; int (&b8m)[4] = L.dec.b8m; int (&p8m)[4] = L.dec.p8m; int (&p8r)[4] = L.dec.p8r; int (&ref8ts)[4] = L.dec.ref8ts; short (&mv8ts)[4][2] = L.dec.mv8ts;
  const int T8 = P.transform8x8_mode;
  __syncthreads();
  best_mode = 1; min_cost = INT_MAX; t8_flag = 0; best_tflag = 0; cbp8ts = -1; tr8_cost = INT_MAX; tr4_cost = INT_MAX;
  for (int k = 0; k < 20; k++) (&l0ref[0][0])[k] = 0;
  for (int k = 0; k < 4; k++) { b8m[k] = 0; p8m[k] = 0; p8r[k] = 0; ref8ts[k] = 0; mv8ts[k][0] = 0; mv8ts[k][1] = 0; }
  if (threadIdx.x == 0) L.pass8ts = 0;
  for (int e = threadIdx.x; e < WR * 4; e += 64) {                             // the 8x8-transform pass's call records: zero unless that pass runs
    const int r = e >> 2, k = e & 3;
    out->pred8ts[r][k][0] = out->pred8ts[r][k][1] = out->mv_int8ts[r][k][0] = out->mv_int8ts[r][k][1] = out->mv8ts[r][k][0] = out->mv8ts[r][k][1] = 0;
    out->cost_int8ts[r][k] = out->cost8ts[r][k] = 0;
  }
  if (T8 == 2)                                                                 // no 4x4-transform P8x8 pass: its call records stay empty
    for (int e = threadIdx.x; e < WR * 36; e += 64) {
      const int r = e / 36, k = 5 + e % 36;
      out->pred[r][k][0] = out->pred[r][k][1] = out->mv_int[r][k][0] = out->mv_int[r][k][1] = out->mv[r][k][0] = out->mv[r][k][1] = 0;
      out->cost_int[r][k] = out->cost[r][k] = 0;
    }
  for (int r = 0; r < 2 * WR; r++) L.surf_c[r][3] = 0;
  // The 25 partition searches of the inter decision -- modes 1..3 (their blocks), the four 8x8 blocks of the 8x8-transform P8x8 pass, then per 8x8
  // block its sub-modes 4..7 -- as ONE loop with ONE call of partition_motion_search, so that the search code is inlined exactly once: as a function
  // called from three loops it saved and restored ~127 callee-saved VGPRs through scratch on every call (21-25 calls per macroblock: 1.5 GB fetched and
  // 2.1 GB written per sweep at 1080p, profiles/r03_p_slice_sq_counters.txt), which is where a third of a macroblock's time went.
  const bool any8 = P.valid[4] || P.valid[5] || P.valid[6] || P.valid[7];
  int &cost = L.dec.cost, &cost8x8 = L.dec.cost8x8, &mc8 = L.dec.mc8;
  cost = 0; cost8x8 = 0; mc8 = INT_MAX;
  for (int it = 0; it < 25; it++) {
    const int phase = it < 5 ? 0 : it < 9 ? 1 : 2;                  // 0: modes 1..3, 1: the 8x8-transform P8x8 pass, 2: the 4x4-transform P8x8 pass
    const int mode = phase == 0 ? (it == 0 ? 1 : it < 3 ? 2 : 3) : phase == 1 ? 4 : 4 + ((it - 9) & 3);
    const int block = phase == 0 ? (it == 0 ? 0 : (it - 1) & 1) : phase == 1 ? it - 5 : (it - 9) >> 2;
    const bool run = (phase == 0 ? P.valid[mode] != 0 : phase == 1 ? (any8 && T8 != 0) : (any8 && T8 != 2 && P.valid[mode] != 0)) && !(D.reduced && it > 0);
    if (phase == 0 && block == 0) cost = 0;
    if (phase == 1 && block == 0 && run) {                          // the 8x8 partition with the 8x8 transform: sub-mode 4 only (mode_decision.c:556)
      tr8_cost = 0;
      __syncthreads();
      if (threadIdx.x == 0) L.pass8ts = 1;
      __syncthreads();
    }
    if (phase == 2 && mode == 4) mc8 = INT_MAX;
    int best_ref = 0, c = 0;
    if (run) {
      partition_motion_search<SM>(mbx, mby, mode, block, out);
      c = list0_cost(mode, block, &best_ref);
    }
    if (phase == 0 && run) {
      cost += c;
      if (mode == 1) {
        for (int b = 0; b < 16; b++) field_set(by0 + (b >> 2), bx0 + (b & 3), best_ref, L.all_mv[b][best_ref][1][0], L.all_mv[b][best_ref][1][1]);
        for (int k = 0; k < 4; k++) l0ref[1][k] = best_ref;
      } else if (mode == 2) l0ref[2][2 * block] = l0ref[2][2 * block + 1] = best_ref;
      else l0ref[3][block] = l0ref[3][block + 2] = best_ref;
      if (mode > 1 && block == 0)
        for (int j = 0; j < psz[mode][1]; j++) for (int i = 0; i < psz[mode][0]; i++)
          field_set(by0 + j, bx0 + i, best_ref, L.all_mv[j * 4 + i][best_ref][mode][0], L.all_mv[j * 4 + i][best_ref][mode][1]);
      __syncthreads();
      if (mode == 1 || block == 1) {                                // the mode's last block
        if (T8) {
          // SetModesAndRefframeForBlocks (md_low.c:186, rdopt.c:1470-1500) writes the mode's best references into the picture array -- a side effect only
          // Transform8x8Mode has -- and TransformDecision predicts with them
          for (int b = 0; b < 16; b++) FREF(by0 + (b >> 2), bx0 + (b & 3)) = (int8_t)l0ref[mode][2 * (b >> 3) + ((b & 3) >> 1)];
          __syncthreads();
          if (T8 == 2) t8_flag = 1;
          else if (P.md_metric != 2) t8_flag = 0;            // SAD / SSE: cost8x8 == cost4x4, the cost stays
          else {
            int c4 = 0, c8 = 0;
            for (int k8 = 0; k8 < 4; k8++) {
              const int b = (k8 >> 1) * 8 + (k8 & 1) * 2, r = l0ref[mode][k8];
              c4 += pred_block_cost(mbx, mby, 8 * (k8 & 1), 8 * (k8 >> 1), 8, r, L.all_mv[b][r][mode][0], L.all_mv[b][r][mode][1], 0);
              c8 += pred_block_cost(mbx, mby, 8 * (k8 & 1), 8 * (k8 >> 1), 8, r, L.all_mv[b][r][mode][0], L.all_mv[b][r][mode][1], 1);
            }
            if (c8 < c4) t8_flag = 1; else { cost = cost - c8 + c4; t8_flag = 0; }
          }
        }
        if (cost < min_cost) { best_mode = mode; min_cost = cost; best_tflag = t8_flag; }
      }
    }
    if (phase == 1 && run) {
      const int j0 = block & 2, i0 = (block & 1) * 2;
      if (c != INT_MAX) c += ((P.lambda_mf[2] * (P.num_refs <= 1 ? 0 : refbits(0))) >> 16) - 1;
      tr8_cost += c;
      ref8ts[block] = best_ref; mv8ts[block][0] = L.all_mv[j0 * 4 + i0][best_ref][4][0]; mv8ts[block][1] = L.all_mv[j0 * 4 + i0][best_ref][4][1];
      for (int j = j0; j < j0 + 2; j++) for (int i = i0; i < i0 + 2; i++) field_set(by0 + j, bx0 + i, best_ref, mv8ts[block][0], mv8ts[block][1]);
      __syncthreads();
      if (block == 3) {
        if (threadIdx.x == 0) L.pass8ts = 0;
        __syncthreads();
      }
    }
    if (phase == 2 && any8 && T8 != 2 && !D.reduced) {
      const int j0 = block & 2, i0 = (block & 1) * 2;
      if (run) {
        for (int j = 0; j < 2; j++) for (int i = 0; i < 2; i++) FREF(by0 + j0 + j, bx0 + i0 + i) = (int8_t)best_ref;
        __syncthreads();
        if (c != INT_MAX) c += ((P.lambda_mf[2] * (P.num_refs <= 1 ? 0 : refbits(mode - 4))) >> 16) - 1;
        if (c < mc8) { mc8 = c; b8m[block] = mode; l0ref[4][block] = best_ref; }
      }
      if (mode == 7) {                                              // the 8x8 block is decided (mode_decision.c:531, :965)
        cost8x8 += mc8;
        for (int j = j0; j < j0 + 2; j++) for (int i = i0; i < i0 + 2; i++)
          field_set(by0 + j, bx0 + i, l0ref[4][block], L.all_mv[j * 4 + i][l0ref[4][block]][b8m[block]][0], L.all_mv[j * 4 + i][l0ref[4][block]][b8m[block]][1]);
        __syncthreads();
        if (block == 3) {
          tr4_cost = cost8x8;
          for (int k = 0; k < 4; k++) { p8m[k] = b8m[k]; p8r[k] = l0ref[4][k]; }      // the P8x8 candidate, before a winning 8x8-transform pass rewrites it below
        }
      }
    }
  }
  if (any8) {
    if (tr4_cost < min_cost || tr8_cost < min_cost) {    // md_low.c:281-326
      best_mode = 8;
      if (T8 == 2) { min_cost = tr8_cost; t8_flag = 1; }
      else if (T8) {
        if (tr8_cost < tr4_cost) { min_cost = tr8_cost; t8_flag = 1; }
        else if (tr4_cost < tr8_cost) { min_cost = tr4_cost; t8_flag = 0; }
        else {                                           // GetBestTransformP8x8: the two passes' predictions against the source
          int c4 = 0, c8 = 0;
          if (P.md_metric == 2) {
            for (int b = 0; b < 16; b++) {
              const int k8 = 2 * (b >> 3) + ((b & 3) >> 1), r = l0ref[4][k8], m8 = b8m[k8];
              c4 += pred_block_cost(mbx, mby, 4 * (b & 3), 4 * (b >> 2), 4, r, L.all_mv[b][r][m8][0], L.all_mv[b][r][m8][1], 0);
            }
            for (int k8 = 0; k8 < 4; k8++) c8 += pred_block_cost(mbx, mby, 8 * (k8 & 1), 8 * (k8 >> 1), 8, ref8ts[k8], mv8ts[k8][0], mv8ts[k8][1], 1);
          } else {
            for (int b = 0; b < 16; b++) {
              const int k8 = 2 * (b >> 3) + ((b & 3) >> 1), r = l0ref[4][k8], m8 = b8m[k8];
              c4 += pred_block_cost(mbx, mby, 4 * (b & 3), 4 * (b >> 2), 4, r, L.all_mv[b][r][m8][0], L.all_mv[b][r][m8][1], 0);
              c8 += pred_block_cost(mbx, mby, 4 * (b & 3), 4 * (b >> 2), 4, ref8ts[k8], mv8ts[k8][0], mv8ts[k8][1], 0);
            }
          }
          if (c8 < c4) { min_cost = tr8_cost; t8_flag = 1; } else { min_cost = tr4_cost; t8_flag = 0; }
        }
      } else { min_cost = tr4_cost; t8_flag = 0; }
    }
  }
  find_skip_mv(mbx, mby);
  if (best_mode != 8) t8_flag = best_tflag;
  else if (t8_flag && T8 != 2) {                         // md_low.c:547-548: an 8x8-transform winner without a coded block gives way to the 4x4 pass
    cbp8ts = t8_pass_cbp(mbx, mby, ref8ts, mv8ts);
    if (cbp8ts == 0) t8_flag = 0;
  }
  if (best_mode == 8 && t8_flag) {                       // SetCoeffAndReconstruction8x8 (rdopt.c:1595): the 8x8-transform pass's partitioning
    for (int k = 0; k < 4; k++) { b8m[k] = 4; l0ref[4][k] = ref8ts[k]; }
    __syncthreads();
    for (int b = 0; b < 16; b++) { const int k8 = 2 * (b >> 3) + ((b & 3) >> 1); L.all_mv[b][ref8ts[k8]][4][0] = mv8ts[k8][0]; L.all_mv[b][ref8ts[k8]][4][1] = mv8ts[k8][1]; }
    __syncthreads();
  }
  for (int b = 0; b < 16; b++) {
    const int k8 = 2 * (b >> 3) + ((b & 3) >> 1), m8 = best_mode == 8 ? b8m[k8] : best_mode, r = l0ref[best_mode == 8 ? 4 : best_mode][k8];
    field_set(by0 + (b >> 2), bx0 + (b & 3), r, L.all_mv[b][r][m8][0], L.all_mv[b][r][m8][1]);
    if (threadIdx.x == 0) { out->final_mv[b][0] = L.all_mv[b][r][m8][0]; out->final_mv[b][1] = L.all_mv[b][r][m8][1]; }
  }
  if (threadIdx.x == 0) {
    out->best_mode = best_mode; out->min_cost = min_cost; out->transform8x8_flag = t8_flag; out->cbp8ts = cbp8ts;
    for (int k = 0; k < 4; k++) { out->b8mode[k] = best_mode == 8 ? b8m[k] : best_mode; out->b8ref[k] = l0ref[best_mode == 8 ? 4 : best_mode][k]; out->p8mode[k] = p8m[k]; out->p8ref[k] = p8r[k]; }
    out->skip_mv[0] = L.all_mv[0][0][0][0]; out->skip_mv[1] = L.all_mv[0][0][0][1];
  }
  __syncthreads();
}


// ---------------------------------------------------------------------------------------------- a macroblock's view of the picture-level state

__device__ void all_mv_from_carry(const short *cin)
{
  for (int i = threadIdx.x; i < (int)(sizeof(L.all_mv) / 4); i += 64) reinterpret_cast<uint32_t *>(L.all_mv)[i] = 0u;
  __syncthreads();
  for (int e = threadIdx.x; e < WR * 16; e += 64) {
    const int r = e >> 4, b = e & 15, k8 = 2 * (b >> 3) + ((b & 3) >> 1);
    L.all_mv[b][r][1][0] = cin[(r * CARRY) * 2]; L.all_mv[b][r][1][1] = cin[(r * CARRY) * 2 + 1];
    L.all_mv[b][r][4][0] = cin[(r * CARRY + 1 + k8) * 2]; L.all_mv[b][r][4][1] = cin[(r * CARRY + 1 + k8) * 2 + 1];
  }
  __syncthreads();
}

// what the macroblock reads of its surroundings, once, into LDS: the source samples, the ring of the motion field (and of UMHexagonS's cost
// maps) its predictors look at, the EPZS row memories of its own columns and of one macroblock either side
__device__ void mb_stage(int mbx, int mby)
{
  const int lane = threadIdx.x;
  const jmhip_slice_params &P = D.p;
  if (lane == 0) { L.mbx = mbx; L.mby = mby; }
  {
    const int r = lane >> 2, k = lane & 3;
    *reinterpret_cast<uint32_t *>(&L.cur[r][k * 4]) = *reinterpret_cast<const uint32_t *>(D.cur + (size_t)(mby * 16 + r) * D.W + mbx * 16 + k * 4);
  }
  if (lane < 30) {
    const int gy = lane / 6, gx = lane - gy * 6, px = 4 * mbx - 1 + gx, py = 4 * mby - 1 + gy;
    const bool own = gy >= 1 && gx >= 1 && gx <= 4;
    int r = -1, vx = 0, vy = 0;
    if (!own && px >= 0 && py >= 0 && px < D.w4) { const size_t at = (size_t)py * D.w4 + px; r = D.ref_idx[at]; vx = D.mv[at * 2]; vy = D.mv[at * 2 + 1]; }
    L.f_ref[gy][gx] = (int8_t)r; L.f_mv[gy][gx][0] = (short)vx; L.f_mv[gy][gx][1] = (short)vy;
  }
  if (P.search_mode == JMHIP_SEARCH_UMHEX)
    for (int e = lane; e < 8 * 30; e += 64) {
      const int bt = e / 30, g = e - bt * 30, gy = g / 6, gx = g - gy * 6, px = 4 * mbx - 1 + gx, py = 4 * mby - 1 + gy;
      const bool own = gy >= 1 && gx >= 1 && gx <= 4;
      int v = 0;
      // own entries: what the slice found there (JM's array still holds the previous picture's value until the macroblock writes it)
      if (px >= 0 && py >= 0 && px < D.w4 && py < D.h4) v = (own ? D.um_in : D.um_cost)[((size_t)bt * D.h4 + py) * D.w4 + px];
      L.um_loc[bt][gy][gx] = v;
    }
  if (P.search_mode == JMHIP_SEARCH_EPZS) {
    // nothing of this macroblock has tested a map cell yet
    for (int e = lane; e < 160; e += 64) L.ep_seen[e] = 0u;
    {
      const size_t at = (size_t)(mby * D.mbw + mbx) * D.ep_cells;
      uint32_t *f = reinterpret_cast<uint32_t *>(D.ep_first + at), *z = reinterpret_cast<uint32_t *>(D.ep_last + at);
      if (!(D.debug & 16)) for (int e = lane; e < D.ep_cells / 4; e += 64) { f[e] = 0u; z[e] = 0u; }
    }
    if (lane == 0) { L.ep_i = 0; L.ep_alias_on = D.ep_alias_flag[mby * D.mbw + mbx] & 1; }
    const int col0 = 4 * mbx - 4;
    for (int e = lane; e < 7 * 12; e += 64) {
      const int t = e / 12, j = e - t * 12, col = clampi(col0 + j, 0, D.w4 - 1);
      L.ep_sad[t][j] = D.ep_dist[((size_t)state_row(col, mbx, mby) * 7 + t) * D.w4 + col];
    }
    if (P.epzs_temporal && lane < 36) {            // the co-located vectors the temporal predictors of this macroblock's blocks can ask for
      const int gy = lane / 6, gx = lane - gy * 6, px = clampi(4 * mbx - 1 + gx, 0, D.w4 - 1), py = clampi(4 * mby - 1 + gy, 0, D.h4 - 1);
      *reinterpret_cast<uint32_t *>(L.ep_colb[gy][gx]) = *reinterpret_cast<const uint32_t *>(D.ep_col + ((size_t)py * D.w4 + px) * 2);
    }
    if (P.epzs_spatial_mem)
      for (int e = lane; e < P.num_refs * 7 * 4 * 12; e += 64) {
        const int j = e % 12, br = (e / 12) & 3, t = (e / 48) % 7, r = e / (48 * 7);
        const int col = clampi(col0 + j, 0, D.w4 - 1);
        const size_t at = ((((size_t)state_row(col, mbx, mby) * WR + r) * 7 + t) * 4 + br) * D.w4 + col;
        *reinterpret_cast<uint32_t *>(L.ep_mot[r][t][br][j]) = *reinterpret_cast<const uint32_t *>(D.ep_motion + at * 2);
      }
  }
  __syncthreads();
}

// hands the macroblock's results on: its sixteen field entries, its cost-map / row-memory entries, the img->all_mv entries the next macroblock
// in coding order reads. Returns (wave-uniform) whether any of it differs from what was stored -- the relaxation schedule's signal.
__device__ int mb_commit(int mbx, int mby)
{
  const int lane = threadIdx.x;
  const jmhip_slice_params &P = D.p;
  __syncthreads();
  bool diff = false;
  if (lane < 16) {
    const int bx = lane & 3, by = lane >> 2;
    const size_t at = (size_t)(4 * mby + by) * D.w4 + 4 * mbx + bx;
    const int8_t r = L.f_ref[by + 1][bx + 1];
    const short vx = L.f_mv[by + 1][bx + 1][0], vy = L.f_mv[by + 1][bx + 1][1];
    diff = D.ref_idx[at] != r || D.mv[at * 2] != vx || D.mv[at * 2 + 1] != vy;
    D.ref_idx[at] = r; D.mv[at * 2] = vx; D.mv[at * 2 + 1] = vy;
  }
  if (P.search_mode == JMHIP_SEARCH_UMHEX)
    for (int e = lane; e < 128; e += 64) {
      const int bt = e >> 4, by = (e >> 2) & 3, bx = e & 3;
      const size_t at = ((size_t)bt * D.h4 + 4 * mby + by) * D.w4 + 4 * mbx + bx;
      const int v = L.um_loc[bt][by + 1][bx + 1];
      diff = diff || D.um_cost[at] != v;
      D.um_cost[at] = v;
    }
  if (P.search_mode == JMHIP_SEARCH_EPZS) {
    const size_t row = (size_t)(mby - D.row0 + 1);
    if (lane == 0) D.ep_nsearch[mby * D.mbw + mbx] = (uint16_t)L.ep_i;
    if (lane < 28) {
      const int t = lane >> 2, j = lane & 3;
      const size_t at = (row * 7 + t) * D.w4 + 4 * mbx + j;
      const int v = L.ep_sad[t][4 + j];
      diff = diff || D.ep_dist[at] != v;
      D.ep_dist[at] = v;
    }
    if (P.epzs_spatial_mem)
      for (int e = lane; e < P.num_refs * 7 * 16; e += 64) {
        const int j = e & 3, br = (e >> 2) & 3, t = (e >> 4) % 7, r = e / (16 * 7);
        const size_t at = ((((row * WR + r) * 7 + t) * 4 + br) * D.w4 + 4 * mbx + j) * 2;
        const uint32_t v = *reinterpret_cast<const uint32_t *>(L.ep_mot[r][t][br][4 + j]);
        uint32_t *g = reinterpret_cast<uint32_t *>(D.ep_motion + at);
        diff = diff || *g != v;
        *g = v;
      }
  }
  if (lane < WR * CARRY) {
    const int r = lane / CARRY, k = lane - r * CARRY, b = k ? ((k - 1) >> 1) * 8 + ((k - 1) & 1) * 2 : 0;
    const uint32_t v = *reinterpret_cast<const uint32_t *>(L.all_mv[b][r][k ? 4 : 1]);
    uint32_t *g = reinterpret_cast<uint32_t *>(D.carry_mb + ((size_t)(mby * D.mbw + mbx) * WR * CARRY + lane) * 2);
    diff = diff || *g != v;
    *g = v;
  }
  return __ballot(diff) != 0ull;
}

template <int SM> __global__ __launch_bounds__(64, 2) void p_slice_kernel(const short *carry_slice_in, short *carry_slice_out)
{
  const int lane = threadIdx.x;
  const int row0 = D.p.mb_first / D.mbw, mby = row0 + blockIdx.x;
  const int last = D.p.mb_first + D.p.mb_count - 1;
  const int x0 = max(0, D.p.mb_first - mby * D.mbw), x1 = min(D.mbw - 1, last - mby * D.mbw);
  if (x0 > x1) return;
  // img->all_mv as the previous macroblock in coding order left it: the slice's first macroblock takes the persistent carry, other row starts
  // the speculated one; inside a row the LDS array simply lives on
  all_mv_from_carry((mby * D.mbw + x0 == D.p.mb_first) ? carry_slice_in : D.carry_in + (size_t)mby * WR * CARRY * 2);
  for (int mbx = x0; mbx <= x1; mbx++) {
    // wait for the row above: its macroblock mbx + 1 (or its last one) if that one belongs to the slice
    if (mby > 0) {
      const int need = min(mbx + 1, D.mbw - 1);
      if ((mby - 1) * D.mbw + need >= D.p.mb_first) {
        long spins = 0;
        while (__hip_atomic_load(&D.prog[mby - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= need) {
          if (++spins > (1L << 28) || __hip_atomic_load(&D.flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(&D.flags[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
          }
          __builtin_amdgcn_s_sleep(8);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    }
    mb_stage(mbx, mby);
    if (threadIdx.x == 0) L.memo_live = 0;                                    // the coding-order walk evaluates every macroblock once: nothing to reuse
    __syncthreads();
#ifdef JMHIP_WAVE_PROF
    const unsigned long long mb_t0 = __builtin_amdgcn_s_memtime();
#endif
    macroblock_low<SM>(mbx, mby, D.out + (mby * D.mbw + mbx));
#ifdef JMHIP_WAVE_PROF
    if (lane == 0) { atomicAdd(&g_wave_prof[8], __builtin_amdgcn_s_memtime() - mb_t0); atomicAdd(&g_wave_prof[9], 1ull); }
#endif
    (void)mb_commit(mbx, mby);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) __hip_atomic_store(&D.prog[mby], mbx + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  // what this row leaves for the macroblock that follows in coding order
  if (lane == 0) {
    short *co = (mby * D.mbw + x1 == last) ? carry_slice_out : D.carry_out + (size_t)mby * WR * CARRY * 2;
    for (int r = 0; r < WR; r++) {
      co[(r * CARRY) * 2] = L.all_mv[0][r][1][0]; co[(r * CARRY) * 2 + 1] = L.all_mv[0][r][1][1];
      for (int k8 = 0; k8 < 4; k8++) { const int b = (k8 >> 1) * 8 + (k8 & 1) * 2; co[(r * CARRY + 1 + k8) * 2] = L.all_mv[b][r][4][0]; co[(r * CARRY + 1 + k8) * 2 + 1] = L.all_mv[b][r][4][1]; }
    }
  }
}


// The same macroblocks under a RELAXATION schedule: every macroblock of the slice at once, each reading whatever its predecessors in coding
// order last handed on (mb_commit), sweep after sweep until a sweep changes nothing. The dependencies are acyclic (raster order), so the
// fixpoint is unique and equals what the coding-order walk produces; what the walk does in mbw + 2 mbh serial steps of one macroblock each,
// the sweeps do with the whole GPU busy. A macroblock is re-evaluated only when a predecessor -- left, up-left, up, up-right, or the one before
// it in coding order (img->all_mv) -- changed what it hands on in the previous sweep; the state of the previous picture is the first guess.
template <int SM> __global__ __launch_bounds__(64, 2) void p_slice_relax_kernel(const short *carry_slice_in)
{
  const int lane = threadIdx.x, first = D.p.mb_first, last = first + D.p.mb_count - 1;
  for (int addr = first + blockIdx.x; addr <= last; addr += gridDim.x) {
    const int mbx = addr % D.mbw, mby = addr / D.mbw;
    bool active = D.first_sweep != 0;
    if (!active) {
      auto chg = [&](int nx, int ny) {
        if (nx < 0 || ny < 0 || nx >= D.mbw) return false;
        const int a = ny * D.mbw + nx;
        return a >= first && a <= last && D.chg_prev[a - first] != 0;
      };
      active = chg(mbx - 1, mby) || chg(mbx, mby - 1) || chg(mbx + 1, mby - 1) || chg(mbx - 1, mby - 1) || (addr > first && D.chg_prev[addr - 1 - first] != 0);
      if (SM == JMHIP_SEARCH_EPZS && (D.ep_alias_flag[addr] & 2)) active = true;      // its pre-marked cells changed
    }
    if (SM == JMHIP_SEARCH_EPZS && active && lane == 0) D.ep_alias_flag[addr] &= 1;
    if (!active) { if (lane == 0) D.chg_next[addr - first] = 0; continue; }
#ifdef JMHIP_WAVE_PROF
    const unsigned long long mb_t0 = __builtin_amdgcn_s_memtime();
#endif
    all_mv_from_carry(addr == first ? carry_slice_in : D.carry_mb + (size_t)(addr - 1) * WR * CARRY * 2);
    mb_stage(mbx, mby);
    if (lane < WR) {
      const int live = D.memo_on && !D.first_sweep;
      L.memo_old[lane] = live ? D.memo[(size_t)addr * WR + lane] : 0ull; L.memo_new[lane] = 0ull; L.ff_same[lane] = 0;
      if (lane == 0) L.memo_live = live;
    }
    __syncthreads();
#ifdef JMHIP_WAVE_PROF
    const unsigned long long mb_t1 = __builtin_amdgcn_s_memtime();
#endif
    macroblock_low<SM>(mbx, mby, D.out + addr);
#ifdef JMHIP_WAVE_PROF
    const unsigned long long mb_t2 = __builtin_amdgcn_s_memtime();
#endif
    const int changed = mb_commit(mbx, mby);
#ifdef JMHIP_WAVE_PROF
    if (lane == 0) {
      const unsigned long long mb_t3 = __builtin_amdgcn_s_memtime();
      atomicAdd(&g_wave_prof[8], mb_t3 - mb_t0); atomicAdd(&g_wave_prof[9], 1ull);
      atomicAdd(&g_wave_prof[13], mb_t1 - mb_t0); atomicAdd(&g_wave_prof[14], mb_t2 - mb_t1); atomicAdd(&g_wave_prof[15], mb_t3 - mb_t2);
    }
#endif
    if (D.memo_on && lane < WR) D.memo[(size_t)addr * WR + lane] = L.memo_new[lane];
    if (lane == 0) { D.chg_next[addr - first] = (uint8_t)changed; if (changed) atomicAdd(D.n_changed, 1); }
    __syncthreads();
  }
}

// after the slice: the EPZS row memories as JM's single-row arrays would now hold them (stored row 0, what the next slice starts from)
__global__ void epzs_rows_fold_kernel(int *ep_dist, short *ep_motion, int w4, int mbw, int mb_first, int mb_count, int row0, int spatial_mem)
{
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= w4) return;
  const int last = mb_first + mb_count - 1, ylast = last / mbw, cx = col >> 2;
  int src = 0;
  for (int yy = ylast; yy >= ylast - 1 && yy >= 0 && !src; yy--) { const int a = yy * mbw + cx; if (a >= mb_first && a <= last) src = yy - row0 + 1; }
  if (!src) return;
  for (int t = 0; t < 7; t++) ep_dist[(size_t)t * w4 + col] = ep_dist[((size_t)src * 7 + t) * w4 + col];
  if (spatial_mem)
    for (int e = 0; e < WR * 7 * 4; e++) {
      const size_t to = ((size_t)e * w4 + col) * 2, from = (((size_t)src * WR * 7 * 4 + e) * w4 + col) * 2;
      ep_motion[to] = ep_motion[from]; ep_motion[to + 1] = ep_motion[from + 1];
    }
}

#undef D
#undef B
#undef L
#undef FREF
#undef FMV
#undef UMLOC


// ---------------------------------------------------------------------------------------------- EPZSMap across searches (me_epzs.c:49,92,1550,1598,1757,1840)
// JM's visited map holds, per cell, the 16-bit EPZSBlkCount of the search that stamped it last, and is never cleared: a test "stamp == the
// running search's" also succeeds for a stamp left 65536 k searches ago, or for the initial zero of a cell no search has touched when the counter
// passes zero. The slice's macroblocks each searched with fresh per-search bitmaps and recorded, per cell, their first and last search that
// tested it (WaveDev.ep_first / ep_last) and how many searches they ran (ep_nsearch). These kernels walk the macroblocks in coding order per cell
// with JM's stamps and list the tests JM would have answered from an old stamp: cell c of macroblock m's search i, where nothing of m touched c
// before i (first[m][c] == i) and the stamp c carries into m equals the stamp of i. Only such tests can differ: inside a macroblock two searches
// are fewer than 65536 apart.
constexpr int EP_SEG = 32;                     // macroblocks per segment of the walk
constexpr int EP_ALIAS_CAP = 8192;
constexpr uint32_t EP_NONE = 0x10000u;

// base[k] = searches before macroblock mb_first + k (k <= mb_count), starting from the counter the slice found; one workgroup
__global__ void ep_alias_base_kernel(const uint16_t *__restrict__ nsearch, int mb_first, int mb_count, const uint32_t *count_in, uint32_t *base, uint32_t *count_out)
{
  __shared__ uint32_t part[1024];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = *count_in;
  __syncthreads();
  for (int k0 = 0; k0 < mb_count; k0 += 1024) {
    const int k = k0 + threadIdx.x;
    const uint32_t v = k < mb_count ? nsearch[mb_first + k] : 0u;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const uint32_t a = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0u;
      __syncthreads();
      part[threadIdx.x] += a;
      __syncthreads();
    }
    if (k < mb_count) base[k] = carry + part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) { base[mb_count] = carry; *count_out = carry; }
}

// seg_last[s][c]: the stamp cell c carries out of segment s if a macroblock of the segment touched it, else EP_NONE
__global__ void ep_alias_seglast_kernel(const uint8_t *__restrict__ last, const uint32_t *__restrict__ base, int mb_first, int mb_count, int cells, uint32_t *seg_last)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
  if (c >= cells) return;
  const int k0 = s * EP_SEG, k1 = min(mb_count, k0 + EP_SEG);
  uint32_t r = EP_NONE;
  for (int k = k1 - 1; k >= k0; k--) {
    const uint32_t z = last[(size_t)(mb_first + k) * cells + c];
    if (z) { r = (base[k] + z) & 0xffffu; break; }
  }
  seg_last[(size_t)s * cells + c] = r;
}

// seg_in[s][c]: the stamp cell c carries into segment s; map_out: JM's map after the slice
__global__ void ep_alias_segin_kernel(const uint32_t *__restrict__ seg_last, int nseg, int cells, const uint16_t *__restrict__ map_in, uint32_t *seg_in, uint16_t *map_out)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cells) return;
  uint32_t l = map_in[c];
  for (int s = 0; s < nseg; s++) {
    seg_in[(size_t)s * cells + c] = l;
    const uint32_t v = seg_last[(size_t)s * cells + c];
    if (v != EP_NONE) l = v;
  }
  map_out[c] = (uint16_t)l;
}

__global__ void ep_alias_detect_kernel(const uint8_t *__restrict__ first_, const uint8_t *__restrict__ last, const uint16_t *__restrict__ nsearch, const uint32_t *__restrict__ base,
                                       const uint32_t *__restrict__ seg_in, int mb_first, int mb_count, int cells, unsigned long long *list, int *n_list)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
  if (c >= cells) return;
  const int k0 = s * EP_SEG, k1 = min(mb_count, k0 + EP_SEG);
  uint32_t l = seg_in[(size_t)s * cells + c];
  for (int k = k0; k < k1; k++) {
    const size_t at = (size_t)(mb_first + k) * cells + c;
    const uint32_t b = base[k], i = (l - b) & 0xffffu, f = first_[at], z = last[at];
    if (i >= 1u && i <= nsearch[mb_first + k] && f == i) {
      const int slot = atomicAdd(n_list, 1);
      if (slot < EP_ALIAS_CAP) list[slot] = ((unsigned long long)(mb_first + k) << 32) | ((unsigned long long)i << 16) | (unsigned)c;
    }
    if (z) l = (b + z) & 0xffffu;
  }
}

// rows whose speculated start differs from what the row above left: flag[1] = first such row (min), else untouched
__global__ void carry_check_kernel(const short *carry_in, const short *carry_out, int row_first, int rows, int *flags)
{
  const int r = row_first + 1 + blockIdx.x * blockDim.x + threadIdx.x;        // row r starts from what row r - 1 left
  if (r >= row_first + rows) return;
  bool same = true;
  for (int i = 0; i < WR * CARRY * 2; i++) same &= carry_in[(size_t)r * WR * CARRY * 2 + i] == carry_out[(size_t)(r - 1) * WR * CARRY * 2 + i];
  if (!same) atomicMin(&flags[1], r);
}

struct SliceState {
  int8_t *ref_idx = nullptr; short *mv = nullptr;
  int *prog = nullptr, *flags = nullptr;
  jmhip_mb_inter *out = nullptr;
  int *ep_dist = nullptr; short *ep_motion = nullptr; short *ep_col = nullptr;        // row memories: [mbh + 1] stored rows (WaveDev)
  // EPZSMap / EPZSBlkCount: the map (16-bit stamps) and the search counter as the last slice left them (and where the running call writes the
  // next ones), the per-macroblock touch records and the scan's scratch
  uint16_t *ep_map = nullptr, *ep_map_next = nullptr; uint32_t *ep_count = nullptr, *ep_count_next = nullptr; int ep_cells = 0;
  uint8_t *ep_first = nullptr, *ep_last = nullptr, *ep_alias_flag = nullptr; uint16_t *ep_nsearch = nullptr;
  uint32_t *ep_base = nullptr, *ep_seg_last = nullptr, *ep_seg_in = nullptr; unsigned long long *ep_alias = nullptr, *ep_alias_new = nullptr; int *ep_alias_n = nullptr;
  int ep_aliases = 0;                          // map tests of the last call answered from an old stamp
  short *carry_mb = nullptr; uint8_t *chg[2] = {nullptr, nullptr};
  unsigned long long *memo = nullptr;
  uint16_t *tie_tab = nullptr; int tie_R = 0;
  short *carry_in = nullptr, *carry_out = nullptr, *carry_slice = nullptr, *carry_slice_next = nullptr;
  int *um_cost = nullptr, *um_cost_snap = nullptr;
  uint16_t *surf = nullptr; size_t surf_rows = 0; int surf_n = 0, surf_refs = 0;
  int passes = 0;
  bool has_col = false;
  int searched_from = 0, searched_to = 0;      // macroblocks [searched_from, searched_to) of the current picture have been searched (by slice calls in order)
  bool t8_any = false;                         // a slice of the current picture was searched with Transform8x8Mode: macroblocks may carry the 8x8-transform flag
};

}  // namespace

static void slice_state_release(SliceState *s)
{
  void *bufs[] = {s->ref_idx, s->mv, s->prog, s->flags, s->out, s->ep_dist, s->ep_motion, s->carry_mb, s->chg[0], s->chg[1], s->memo, s->tie_tab, s->ep_col, s->carry_in, s->carry_out,
                  s->carry_slice, s->carry_slice_next, s->um_cost, s->um_cost_snap, s->surf, s->ep_map, s->ep_map_next, s->ep_count, s->ep_count_next, s->ep_first, s->ep_last,
                  s->ep_alias_flag, s->ep_nsearch, s->ep_base, s->ep_seg_last, s->ep_seg_in, s->ep_alias, s->ep_alias_new, s->ep_alias_n};
  for (void *b : bufs) if (b) (void)hipFree(b);
  delete s;
}

// the state lives in the context as an opaque pointer (jmhip_internal.h: void *slice_state)
static SliceState *slice_state(jmhip_ctx *c)
{
  if (c->slice_state) return static_cast<SliceState *>(c->slice_state);
  SliceState *s = new SliceState();
  const size_t w4 = c->W / 4, h4 = c->H / 4, nmb = (size_t)c->mbw * c->mbh;
  bool ok = hipMalloc((void **)&s->ref_idx, w4 * h4) == hipSuccess && hipMalloc((void **)&s->mv, w4 * h4 * 4) == hipSuccess &&
            hipMalloc((void **)&s->prog, sizeof(int) * c->mbh) == hipSuccess && hipMalloc((void **)&s->flags, sizeof(int) * 4) == hipSuccess &&
            hipMalloc((void **)&s->out, sizeof(jmhip_mb_inter) * nmb) == hipSuccess &&
            hipMalloc((void **)&s->ep_dist, sizeof(int) * (c->mbh + 1) * 7 * w4) == hipSuccess &&
            hipMalloc((void **)&s->ep_motion, sizeof(short) * (c->mbh + 1) * WR * 7 * 4 * w4 * 2) == hipSuccess &&
            hipMalloc((void **)&s->carry_mb, sizeof(short) * nmb * WR * CARRY * 2) == hipSuccess &&
            hipMalloc((void **)&s->chg[0], nmb) == hipSuccess && hipMalloc((void **)&s->chg[1], nmb) == hipSuccess &&
            hipMalloc((void **)&s->memo, sizeof(unsigned long long) * nmb * WR) == hipSuccess &&
            hipMalloc((void **)&s->ep_col, sizeof(short) * h4 * w4 * 2) == hipSuccess &&
            hipMalloc((void **)&s->carry_in, sizeof(short) * c->mbh * WR * CARRY * 2) == hipSuccess && hipMalloc((void **)&s->carry_out, sizeof(short) * c->mbh * WR * CARRY * 2) == hipSuccess &&
            hipMalloc((void **)&s->carry_slice, sizeof(short) * WR * CARRY * 2) == hipSuccess && hipMalloc((void **)&s->carry_slice_next, sizeof(short) * WR * CARRY * 2) == hipSuccess &&
            hipMalloc((void **)&s->um_cost, sizeof(int) * 8 * h4 * w4) == hipSuccess && hipMalloc((void **)&s->um_cost_snap, sizeof(int) * 8 * h4 * w4) == hipSuccess;
  if (!ok) { slice_state_release(s); return nullptr; }      // nothing half-allocated stays behind: the call fails with NOMEM and may be retried
  c->slice_state = s;
  return s;
}

// the EPZS map state and scan scratch, sized for the context's picture and this search range (a different range starts from a fresh map, as a
// JM encoder configured with it would)
static int ep_state_ensure(jmhip_ctx *c, SliceState *s, int search_range)
{
  const int side = 2 * search_range + 1, cells = (side * side + 3) & ~3;
  if (s->ep_cells == cells) return JMHIP_OK;
  void *old[] = {s->ep_map, s->ep_map_next, s->ep_count, s->ep_count_next, s->ep_first, s->ep_last, s->ep_alias_flag, s->ep_nsearch, s->ep_base, s->ep_seg_last, s->ep_seg_in, s->ep_alias,
                 s->ep_alias_new, s->ep_alias_n};
  for (void *b : old) if (b) (void)hipFree(b);
  s->ep_map = s->ep_map_next = nullptr; s->ep_count = s->ep_count_next = nullptr; s->ep_first = s->ep_last = s->ep_alias_flag = nullptr; s->ep_nsearch = nullptr;
  s->ep_base = s->ep_seg_last = s->ep_seg_in = nullptr; s->ep_alias = s->ep_alias_new = nullptr; s->ep_alias_n = nullptr; s->ep_cells = 0;
  const size_t nmb = (size_t)c->mbw * c->mbh, nseg = (nmb + EP_SEG - 1) / EP_SEG;
  const bool ok = hipMalloc((void **)&s->ep_map, sizeof(uint16_t) * cells) == hipSuccess && hipMalloc((void **)&s->ep_map_next, sizeof(uint16_t) * cells) == hipSuccess &&
                  hipMalloc((void **)&s->ep_count, sizeof(uint32_t)) == hipSuccess && hipMalloc((void **)&s->ep_count_next, sizeof(uint32_t)) == hipSuccess &&
                  hipMalloc((void **)&s->ep_first, nmb * cells) == hipSuccess && hipMalloc((void **)&s->ep_last, nmb * cells) == hipSuccess &&
                  hipMalloc((void **)&s->ep_alias_flag, nmb) == hipSuccess && hipMalloc((void **)&s->ep_nsearch, sizeof(uint16_t) * nmb) == hipSuccess &&
                  hipMalloc((void **)&s->ep_base, sizeof(uint32_t) * (nmb + 1)) == hipSuccess && hipMalloc((void **)&s->ep_seg_last, sizeof(uint32_t) * nseg * cells) == hipSuccess &&
                  hipMalloc((void **)&s->ep_seg_in, sizeof(uint32_t) * nseg * cells) == hipSuccess && hipMalloc((void **)&s->ep_alias, sizeof(unsigned long long) * EP_ALIAS_CAP) == hipSuccess &&
                  hipMalloc((void **)&s->ep_alias_new, sizeof(unsigned long long) * EP_ALIAS_CAP) == hipSuccess && hipMalloc((void **)&s->ep_alias_n, sizeof(int)) == hipSuccess;
  if (!ok) return jm_fail(c, JMHIP_ERR_NOMEM, "EPZS map state of the slice search");
  s->ep_cells = cells;
  JM_HIP_CHECK(c, hipMemsetAsync(s->ep_map, 0, sizeof(uint16_t) * cells, c->stream));
  JM_HIP_CHECK(c, hipMemsetAsync(s->ep_count, 0, sizeof(uint32_t), c->stream));
  JM_HIP_CHECK(c, hipMemsetAsync(s->ep_first, 0, nmb * cells, c->stream));
  JM_HIP_CHECK(c, hipMemsetAsync(s->ep_last, 0, nmb * cells, c->stream));
  JM_HIP_CHECK(c, hipMemsetAsync(s->ep_nsearch, 0, sizeof(uint16_t) * nmb, c->stream));
  return JMHIP_OK;
}

extern "C" int jmhip_slice_state_reset(jmhip_ctx *c)
{
  if (!c) return JMHIP_ERR_ARG;
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  SliceState *s = slice_state(c);
  if (!s) return jm_fail(c, JMHIP_ERR_NOMEM, "slice search state");
  const size_t w4 = c->W / 4, h4 = c->H / 4;
  JM_HIP_CHECK(c, hipMemsetAsync(s->ep_dist, 0, sizeof(int) * (c->mbh + 1) * 7 * w4, c->stream));
  JM_HIP_CHECK(c, hipMemsetAsync(s->ep_motion, 0, sizeof(short) * (c->mbh + 1) * WR * 7 * 4 * w4 * 2, c->stream));
  JM_HIP_CHECK(c, hipMemsetAsync(s->carry_mb, 0, sizeof(short) * (size_t)c->mbw * c->mbh * WR * CARRY * 2, c->stream));
  JM_HIP_CHECK(c, hipMemsetAsync(s->ep_col, 0, sizeof(short) * h4 * w4 * 2, c->stream));
  JM_HIP_CHECK(c, hipMemsetAsync(s->carry_slice, 0, sizeof(short) * WR * CARRY * 2, c->stream));
  JM_HIP_CHECK(c, hipMemsetAsync(s->carry_in, 0, sizeof(short) * c->mbh * WR * CARRY * 2, c->stream));
  JM_HIP_CHECK(c, hipMemsetAsync(s->um_cost, 0, sizeof(int) * 8 * h4 * w4, c->stream));
  JM_HIP_CHECK(c, hipMemsetAsync(s->ref_idx, 0xff, w4 * h4, c->stream));
  JM_HIP_CHECK(c, hipMemsetAsync(s->mv, 0, w4 * h4 * 4, c->stream));
  // EPZSMap is calloc'ed and EPZSBlkCount starts at zero (me_epzs.c:49,396)
  if (s->ep_map) JM_HIP_CHECK(c, hipMemsetAsync(s->ep_map, 0, sizeof(uint16_t) * s->ep_cells, c->stream));
  if (s->ep_count) JM_HIP_CHECK(c, hipMemsetAsync(s->ep_count, 0, sizeof(uint32_t), c->stream));
  s->ep_aliases = 0;
  s->has_col = false;
  return JMHIP_OK;
}

extern "C" int jmhip_epzs_colocated_upload(jmhip_ctx *c, const int16_t *col_mv)
{
  if (!c || !col_mv) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_epzs_colocated_upload: NULL") : JMHIP_ERR_ARG;
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  SliceState *s = slice_state(c);
  if (!s) return jm_fail(c, JMHIP_ERR_NOMEM, "slice search state");
  JM_HIP_CHECK(c, hipMemcpyAsync(s->ep_col, col_mv, sizeof(short) * (size_t)(c->H / 4) * (c->W / 4) * 2, hipMemcpyHostToDevice, c->stream));
  JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  s->has_col = true;
  return JMHIP_OK;
}

extern "C" int jmhip_p_slice_search(jmhip_ctx *c, const jmhip_slice_params *prm, jmhip_mb_inter *results)
{
  if (!c || !prm) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_p_slice_search: NULL arguments") : JMHIP_ERR_ARG;
  const int nmb = c->mbw * c->mbh;
  if (prm->search_mode != JMHIP_SEARCH_FULL && prm->search_mode != JMHIP_SEARCH_FASTFULL && prm->search_mode != JMHIP_SEARCH_UMHEX &&
      prm->search_mode != JMHIP_SEARCH_UMHEX_SIMPLE && prm->search_mode != JMHIP_SEARCH_EPZS)
    return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_p_slice_search: search_mode must be -1, 0, 1, 2 or 3");
  if (prm->num_refs < 1 || prm->num_refs > JMHIP_SLICE_REFS) return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_p_slice_search: 1..JMHIP_SLICE_REFS references");
  // ranges up to 33 fit this file's LDS windows; the exhaustive searches' sweeps over the frame kernels (me_xslice.hip) take up to 40
  if (prm->search_range < 1 || prm->search_range > c->cfg.search_range || (prm->search_range > 33 && !jm_xslice_covers(prm)))
    return jm_fail(c, JMHIP_ERR_ARG, "jmhip_p_slice_search: search_range (at most the context's; at most 33, or 40 for search modes -1 / 0 with JM's default metrics and the 4x4 transform)");
  if (prm->mb_first < 0 || prm->mb_count < 1 || prm->mb_first + prm->mb_count > nmb) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_p_slice_search: slice outside the picture");
  if (!prm->valid[1]) return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_p_slice_search: the 16x16 mode must be enabled");
  if (!c->has_cur) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_p_slice_search: current picture not uploaded");
  for (int k = 0; k < 3; k++) {
    if (prm->metric[k] < 0 || prm->metric[k] > 2) return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_p_slice_search: ME metrics are SAD (0), SSE (1) or SATD (2)");
    if (prm->lambda_mf[k] < 0 || prm->lambda_mf[k] >= (1 << 24)) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_p_slice_search: lambda factor out of range");
  }
  if ((prm->search_mode == JMHIP_SEARCH_FULL || prm->search_mode == JMHIP_SEARCH_FASTFULL) && prm->metric[0] != 0)
    return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_p_slice_search: the exhaustive searches take the SAD metric at full-pel positions");
  if (prm->md_metric < 0 || prm->md_metric > 2) return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_p_slice_search: mode-decision metric SAD (0), SSE (1) or SATD (2)");
  if (prm->slice_mbs < 0) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_p_slice_search: slice_mbs");
  if (prm->rdopt && (prm->search_mode == JMHIP_SEARCH_EPZS || prm->search_mode == JMHIP_SEARCH_UMHEX))
    return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_p_slice_search: rdopt != 0 (call records for the high-complexity modes) with search modes -1, 0, 2");
  if (prm->transform8x8_mode < 0 || prm->transform8x8_mode > 2) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_p_slice_search: transform8x8_mode must be 0, 1 or 2");
  if (prm->transform8x8_mode && !prm->valid[4]) return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_p_slice_search: Transform8x8Mode needs the 8x8 sub-mode (valid[4])");
  if (prm->transform8x8_mode == 1 && (prm->t8_qp < 0 || prm->t8_qp > 51 + 48 || prm->t8_disthres < 0 || prm->t8_disthres > 1))
    return jm_fail(c, JMHIP_ERR_ARG, "jmhip_p_slice_search: the 8x8 quantiser of Transform8x8Mode 1 (t8_qp, t8_disthres)");
  if ((prm->wp_me || prm->wp_pred) && (prm->wp_denom < 0 || prm->wp_denom > 7)) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_p_slice_search: weighted-prediction denominator");
  for (int r = 0; r < prm->num_refs; r++) {
    const int sl = prm->ref_slot[r];
    if (sl < 0 || sl >= (int)c->refs.size() || !c->refs[sl].has_pic || !c->refs[sl].has_luma_sub) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_p_slice_search: reference slot without sub-pel planes (jmhip_interp_luma)");
  }
  if (prm->search_mode == JMHIP_SEARCH_EPZS && (prm->epzs_nwin > 40 || prm->epzs_nwin_ext > 100 || prm->epzs_nwin < 0 || prm->epzs_nwin_ext < 0)) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_p_slice_search: EPZS window predictor tables (jmhip_epzs_setup)");
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  SliceState *s = slice_state(c);
  if (!s) return jm_fail(c, JMHIP_ERR_NOMEM, "slice search state");
  if (prm->search_mode == JMHIP_SEARCH_EPZS && prm->epzs_temporal && !s->has_col) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_p_slice_search: EPZS temporal predictors need jmhip_epzs_colocated_upload");
  int rc = jm_ensure_ref_table(c);
  if (rc) return rc;
  const size_t w4 = c->W / 4, h4 = c->H / 4;
  const int row_first = prm->mb_first / c->mbw, row_last = (prm->mb_first + prm->mb_count - 1) / c->mbw, rows = row_last - row_first + 1;
  // (a new picture's enc_picture->ref_idx / mv need no reset: a macroblock only ever reads entries of macroblocks coded before it, and the
  // previous picture's field is the relaxation schedule's first guess)
  const bool exhaustive = prm->search_mode == JMHIP_SEARCH_FULL || prm->search_mode == JMHIP_SEARCH_FASTFULL;
  // schedule: relaxation sweeps over `relax_grid` resident workgroups (default), or JMHIP_SLICE_SCHED=wave: the coding-order wavefront only
  int relax_grid = 0;
  {
    const char *e = getenv("JMHIP_SLICE_SCHED");
    if (!(e && !strcmp(e, "wave"))) relax_grid = 2048;                                      // two waves per SIMD (256 VGPRs): all that can be resident
    if (const char *g = getenv("JMHIP_SLICE_GRID")) if (relax_grid) relax_grid = std::max(1, atoi(g));
  }
  // the exhaustive searches go through the frame kernels' sweeps (me_xslice.hip) where those cover the configuration; what they do not cover,
  // or do not settle within their cap, runs on this file's one-wave-per-macroblock kernels
  const bool x_path = exhaustive && relax_grid && jm_xslice_covers(prm);
  bool x_settled = false;
  if (x_path) {
    int xs = 0, xp = 0;
    s->passes = 0;
    jm_stage_begin(c, JMHIP_STAGE_ME_INT);
    rc = jm_xslice_run(c, prm, s->ref_idx, s->mv, s->out, &xp, &xs);
    jm_stage_end(c, JMHIP_STAGE_ME_INT);
    if (rc) return rc;
    s->passes = xp;
    x_settled = xs != 0;
    if (!x_settled && prm->search_range > 33) return jm_fail(c, JMHIP_ERR_DEVICE, "jmhip_p_slice_search: the sweeps did not settle within their cap and the range exceeds what the macroblock kernels take over (33)");
  }
  if (exhaustive && !x_settled) {
    const int rs = prm->search_range + SURF_MARGIN, n = ((2 * rs + 1) * (2 * rs + 1) + 63) & ~63;
    const size_t slots = (size_t)std::max(rows, relax_grid);      // one surface set per resident workgroup (blockIdx.x)
    if (s->surf_rows < slots || s->surf_n < n || s->surf_refs < prm->num_refs) {
      if (s->surf) JM_HIP_CHECK(c, hipFree(s->surf));
      s->surf = nullptr; s->surf_rows = 0;
      if (hipMalloc((void **)&s->surf, sizeof(uint16_t) * slots * 2 * prm->num_refs * SURF_PLANES * n) != hipSuccess) return jm_fail(c, JMHIP_ERR_NOMEM, "SAD surfaces of the slice search");
      s->surf_rows = slots; s->surf_n = n; s->surf_refs = prm->num_refs;
    }
  }
  WaveDev D{};
  D.p = *prm;
  D.surf = s->surf; D.surf_n = s->surf_n;
  D.debug = getenv("JMHIP_WAVE_DEBUG") ? atoi(getenv("JMHIP_WAVE_DEBUG")) : 0;
  if (D.debug) fprintf(stderr, "jmhip: JMHIP_WAVE_DEBUG=%d is set: stages of the slice search are SKIPPED for timing experiments, its results are WRONG\n", D.debug);
  D.W = c->W; D.H = c->H; D.Wp = c->Wp; D.Hp = c->Hp; D.mbw = c->mbw; D.mbh = c->mbh; D.w4 = (int)w4; D.h4 = (int)h4;
  D.cur = c->cur_y;
  D.ref_sub = reinterpret_cast<const uint8_t *const *>(c->ref_ptrs_dev) + 32;
  D.ref_idx = s->ref_idx; D.mv = s->mv; D.prog = s->prog; D.flags = s->flags; D.out = s->out;
  D.ep_dist = s->ep_dist; D.ep_motion = s->ep_motion; D.ep_col = s->ep_col; D.row0 = row_first;
  D.carry_in = s->carry_in; D.carry_out = s->carry_out; D.um_cost = s->um_cost; D.um_in = s->um_cost_snap; D.carry_mb = s->carry_mb;
  D.n_changed = s->flags + 2;
  D.memo = s->memo;
  if (exhaustive) {
    if (s->tie_R != prm->search_range) {
      const int R = prm->search_range, side = 2 * R + 1;
      std::vector<uint16_t> t((size_t)side * side);
      for (int dy = -R; dy <= R; dy++) for (int dx = -R; dx <= R; dx++) {        // spiral_search_x / _y order, mv-search.c:366-393 (me_common.h spiral_pos)
        const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy, l = std::max(adx, ady);
        int pos = 0;
        if (l) {
          const int k0 = (2 * l - 1) * (2 * l - 1);
          pos = (ady == l && adx < l) ? k0 + 2 * (dx + l - 1) + (dy > 0 ? 1 : 0) : k0 + 2 * (2 * l - 1) + 2 * (dy + l) + (dx > 0 ? 1 : 0);
        }
        t[(size_t)(dy + R) * side + (dx + R)] = (uint16_t)(pos + 1);
      }
      if (s->tie_tab) JM_HIP_CHECK(c, hipFree(s->tie_tab));
      s->tie_tab = nullptr; s->tie_R = 0;
      if (hipMalloc((void **)&s->tie_tab, t.size() * sizeof(uint16_t)) != hipSuccess) return jm_fail(c, JMHIP_ERR_NOMEM, "spiral table of the slice search");
      JM_HIP_CHECK(c, hipMemcpyAsync(s->tie_tab, t.data(), t.size() * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
      JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
      s->tie_R = R;
    }
    D.tie_tab = s->tie_tab; D.tie_R = s->tie_R;
  }
  D.memo_on = exhaustive && relax_grid && !prm->transform8x8_mode && !(getenv("JMHIP_SLICE_MEMO") && !atoi(getenv("JMHIP_SLICE_MEMO")));
  const bool epzs = prm->search_mode == JMHIP_SEARCH_EPZS;
  std::vector<unsigned long long> ep_used;       // EPZS: the map tests this call's searches took as answered from an old stamp (sorted)
  // a macroblock's own cost-map entries start from what the slice found (mb_stage)
  if (prm->search_mode == JMHIP_SEARCH_UMHEX) JM_HIP_CHECK(c, hipMemcpyAsync(s->um_cost_snap, s->um_cost, sizeof(int) * 8 * h4 * w4, hipMemcpyDeviceToDevice, c->stream));
  if (!x_path) s->passes = 0;
  if (epzs) {
    rc = ep_state_ensure(c, s, prm->search_range);
    if (rc) return rc;
    if (45 * prm->num_refs >= EP_STAMPED) return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_p_slice_search: EPZS search counts per macroblock are kept in 8 bits");
    JM_HIP_CHECK(c, hipMemsetAsync(s->ep_alias_flag, 0, (size_t)nmb, c->stream));
    D.ep_first = s->ep_first; D.ep_last = s->ep_last; D.ep_nsearch = s->ep_nsearch; D.ep_cells = s->ep_cells;
    D.ep_alias = s->ep_alias; D.ep_alias_n = 0; D.ep_alias_flag = s->ep_alias_flag;
    s->ep_aliases = 0;
  }
  jm_stage_begin(c, JMHIP_STAGE_ME_INT);
  bool settled = x_settled;
  int sweep_no = 0;                              // sweeps of this call so far (the changed-flag arrays alternate with it)
  // the walkers' sweep 0 searches the 16x16 mode only (JMHIP_SLICE_REDUCED0=0: a full sweep 0). Measured per new 1080p picture, bench clip: EPZS 32 / 23 / 55 -> 24 / 23 / 47 ms,
  // UMHexagonS 19.5 / 19.5 / 38 -> 16 / 16.5 / 36 ms; the smooth clip within +-5 %
  const bool reduced0 = !exhaustive && !(getenv("JMHIP_SLICE_REDUCED0") && !atoi(getenv("JMHIP_SLICE_REDUCED0")));
  // the slice under the current set of pre-marked map cells: relaxation sweeps to the fixpoint, else the coding-order walk
  auto settle = [&]() -> int {
  if (relax_grid && !settled) {
    // relaxation sweeps (p_slice_relax_kernel); the coding-order walk below remains the fallback if they do not settle within the cap
    const int cap = getenv("JMHIP_SLICE_SWEEPS") ? atoi(getenv("JMHIP_SLICE_SWEEPS")) : 160;      // a sweep costs >= one macroblock (1.2 ms), the walk 250+
    const int grid = std::min(relax_grid, prm->mb_count);
    for (int sweep = 0; sweep < cap && !settled; sweep++, sweep_no++) {
      JM_HIP_CHECK(c, hipMemsetAsync(s->flags, 0, sizeof(int) * 4, c->stream));
      // sweep 0 searches the 16x16 mode only -- a first guess of the field at a fraction of a full sweep's cost (any start reaches the same fixpoint) --
      // and sweep 1 evaluates every macroblock in full whatever changed
      D.reduced = reduced0 && sweep_no == 0;
      D.first_sweep = sweep_no == 0 || (reduced0 && sweep_no == 1); D.chg_prev = s->chg[sweep_no & 1]; D.chg_next = s->chg[(sweep_no + 1) & 1];
      JM_HIP_CHECK(c, hipMemcpyToSymbolAsync(HIP_SYMBOL(c_wave), &D, sizeof(D), 0, hipMemcpyHostToDevice, c->stream));
      switch (prm->search_mode) {
      case JMHIP_SEARCH_EPZS: p_slice_relax_kernel<JMHIP_SEARCH_EPZS><<<grid, 64, 0, c->stream>>>(s->carry_slice); break;
      case JMHIP_SEARCH_UMHEX: p_slice_relax_kernel<JMHIP_SEARCH_UMHEX><<<grid, 64, 0, c->stream>>>(s->carry_slice); break;
      case JMHIP_SEARCH_UMHEX_SIMPLE: p_slice_relax_kernel<JMHIP_SEARCH_UMHEX_SIMPLE><<<grid, 64, 0, c->stream>>>(s->carry_slice); break;
      case JMHIP_SEARCH_FASTFULL: p_slice_relax_kernel<JMHIP_SEARCH_FASTFULL><<<grid, 64, 0, c->stream>>>(s->carry_slice); break;
      default: p_slice_relax_kernel<JMHIP_SEARCH_FULL><<<grid, 64, 0, c->stream>>>(s->carry_slice); break;
      }
      JM_HIP_CHECK(c, hipGetLastError());
      s->passes++;
      int flags[4] = {0, 0, 1, 0};
      JM_HIP_CHECK(c, hipMemcpyAsync(flags, s->flags, sizeof(flags), hipMemcpyDeviceToHost, c->stream));
      JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
      settled = flags[2] == 0;
      if (getenv("JMHIP_SLICE_TRACE")) {
        static double t_last = 0.0;
        struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
        const double t_now = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
        fprintf(stderr, "sweep %d: %d macroblocks changed what they hand on (%.2f ms since the previous line)\n", sweep_no, flags[2], t_now - t_last);
        t_last = t_now;
      }
    }
    if (settled && !x_settled)
      JM_HIP_CHECK(c, hipMemcpyAsync(s->carry_slice_next, s->carry_mb + (size_t)(prm->mb_first + prm->mb_count - 1) * WR * CARRY * 2, sizeof(short) * WR * CARRY * 2, hipMemcpyDeviceToDevice, c->stream));
  }
  for (; !settled;) {
    D.reduced = 0;                                 // the coding-order walk evaluates every macroblock once, in full
    JM_HIP_CHECK(c, hipMemsetAsync(s->prog, 0, sizeof(int) * c->mbh, c->stream));
    const int init_flags[4] = {0, 1 << 30, 0, 0};
    JM_HIP_CHECK(c, hipMemcpyAsync(s->flags, init_flags, sizeof(init_flags), hipMemcpyHostToDevice, c->stream));
    JM_HIP_CHECK(c, hipMemcpyToSymbolAsync(HIP_SYMBOL(c_wave), &D, sizeof(D), 0, hipMemcpyHostToDevice, c->stream));
    switch (prm->search_mode) {
    case JMHIP_SEARCH_EPZS: p_slice_kernel<JMHIP_SEARCH_EPZS><<<rows, 64, 0, c->stream>>>(s->carry_slice, s->carry_slice_next); break;
    case JMHIP_SEARCH_UMHEX: p_slice_kernel<JMHIP_SEARCH_UMHEX><<<rows, 64, 0, c->stream>>>(s->carry_slice, s->carry_slice_next); break;
    case JMHIP_SEARCH_UMHEX_SIMPLE: p_slice_kernel<JMHIP_SEARCH_UMHEX_SIMPLE><<<rows, 64, 0, c->stream>>>(s->carry_slice, s->carry_slice_next); break;
    case JMHIP_SEARCH_FASTFULL: p_slice_kernel<JMHIP_SEARCH_FASTFULL><<<rows, 64, 0, c->stream>>>(s->carry_slice, s->carry_slice_next); break;
    default: p_slice_kernel<JMHIP_SEARCH_FULL><<<rows, 64, 0, c->stream>>>(s->carry_slice, s->carry_slice_next); break;
    }
    JM_HIP_CHECK(c, hipGetLastError());
    s->passes++;
    int flags[4] = {0, 1 << 30, 0, 0};
    if (epzs && rows > 1) carry_check_kernel<<<(rows + 63) / 64, 64, 0, c->stream>>>(s->carry_in, s->carry_out, row_first, rows, s->flags);
    JM_HIP_CHECK(c, hipMemcpyAsync(flags, s->flags, sizeof(flags), hipMemcpyDeviceToHost, c->stream));
    JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (flags[0]) { return jm_fail(c, JMHIP_ERR_DEVICE, "p_slice_kernel: a row waited too long for the row above (internal error)"); }
    if (!epzs || rows == 1 || flags[1] == (1 << 30)) break;
    if (s->passes > rows + 80) { return jm_fail(c, JMHIP_ERR_DEVICE, "p_slice_kernel: row-start speculation did not settle (internal error)"); }
    // next pass: every row starts from what the row above left in this pass; rows up to the first wrong one were exact already. (The stored
    // rows of the row memories are rewritten in coding order by every pass; row 0, what the slice found, is never written.)
    JM_HIP_CHECK(c, hipMemcpyAsync(s->carry_in + (size_t)(row_first + 1) * WR * CARRY * 2, s->carry_out + (size_t)row_first * WR * CARRY * 2,
                                   sizeof(short) * (size_t)(rows - 1) * WR * CARRY * 2, hipMemcpyDeviceToDevice, c->stream));
  }
  return JMHIP_OK;
  };
  for (int round = 0;; round++) {
    rc = settle();
    if (rc) { jm_stage_end(c, JMHIP_STAGE_ME_INT); return rc; }
    if (!epzs) break;
    // JM's map across the slice's searches: which tests would an old stamp have answered? (ep_alias_* kernels)
    const int cells = s->ep_cells, nseg = (prm->mb_count + EP_SEG - 1) / EP_SEG;
    JM_HIP_CHECK(c, hipMemsetAsync(s->ep_alias_n, 0, sizeof(int), c->stream));
    ep_alias_base_kernel<<<1, 1024, 0, c->stream>>>(s->ep_nsearch, prm->mb_first, prm->mb_count, s->ep_count, s->ep_base, s->ep_count_next);
    ep_alias_seglast_kernel<<<dim3((cells + 255) / 256, nseg), 256, 0, c->stream>>>(s->ep_last, s->ep_base, prm->mb_first, prm->mb_count, cells, s->ep_seg_last);
    ep_alias_segin_kernel<<<(cells + 255) / 256, 256, 0, c->stream>>>(s->ep_seg_last, nseg, cells, s->ep_map, s->ep_seg_in, s->ep_map_next);
    ep_alias_detect_kernel<<<dim3((cells + 255) / 256, nseg), 256, 0, c->stream>>>(s->ep_first, s->ep_last, s->ep_nsearch, s->ep_base, s->ep_seg_in, prm->mb_first, prm->mb_count, cells,
                                                                                 s->ep_alias_new, s->ep_alias_n);
    JM_HIP_CHECK(c, hipGetLastError());
    int n_new = 0;
    JM_HIP_CHECK(c, hipMemcpyAsync(&n_new, s->ep_alias_n, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (n_new > EP_ALIAS_CAP) { jm_stage_end(c, JMHIP_STAGE_ME_INT); return jm_fail(c, JMHIP_ERR_DEVICE, "jmhip_p_slice_search: more aliased EPZS map tests in one slice than the list holds"); }
    std::vector<unsigned long long> fresh((size_t)n_new);
    if (n_new) {
      JM_HIP_CHECK(c, hipMemcpyAsync(fresh.data(), s->ep_alias_new, sizeof(unsigned long long) * n_new, hipMemcpyDeviceToHost, c->stream));
      JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
      std::sort(fresh.begin(), fresh.end());
    }
    if (getenv("JMHIP_SLICE_TRACE")) fprintf(stderr, "EPZS map round %d: %d tests answered from an old stamp (%zu pre-marked in this round's searches)\n", round, n_new, ep_used.size());
    if (fresh == ep_used) break;                  // the searches ran with exactly the cells JM's map would have shown them
    if (round >= 64) { jm_stage_end(c, JMHIP_STAGE_ME_INT); return jm_fail(c, JMHIP_ERR_DEVICE, "jmhip_p_slice_search: the aliased EPZS map tests did not settle (internal error)"); }
    // search the macroblocks whose pre-marked cells changed again (they, and whatever their results change downstream, through the sweeps)
    std::vector<uint8_t> fl((size_t)nmb, 0);
    for (unsigned long long e : ep_used) fl[(size_t)(e >> 32)] |= 2;
    for (unsigned long long e : fresh) fl[(size_t)(e >> 32)] |= 3;
    ep_used.swap(fresh);
    JM_HIP_CHECK(c, hipMemcpyAsync(s->ep_alias_flag, fl.data(), (size_t)nmb, hipMemcpyHostToDevice, c->stream));
    if (!ep_used.empty()) JM_HIP_CHECK(c, hipMemcpyAsync(s->ep_alias, ep_used.data(), sizeof(unsigned long long) * ep_used.size(), hipMemcpyHostToDevice, c->stream));
    JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    D.ep_alias_n = (int)ep_used.size();
    settled = false;
  }
  if (epzs) { std::swap(s->ep_map, s->ep_map_next); std::swap(s->ep_count, s->ep_count_next); s->ep_aliases = (int)ep_used.size(); }
  if (epzs) epzs_rows_fold_kernel<<<((int)w4 + 63) / 64, 64, 0, c->stream>>>(s->ep_dist, s->ep_motion, (int)w4, c->mbw, prm->mb_first, prm->mb_count, row_first, prm->epzs_spatial_mem);
  jm_stage_end(c, JMHIP_STAGE_ME_INT);
#ifdef JMHIP_WAVE_PROF
  {
    unsigned long long h[16];
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wave_prof), sizeof(h));
    const double nmb = h[9] ? (double)h[9] : 1.0;
    fprintf(stderr, "WAVE PROF stage %.0f decide %.0f commit %.0f | ", h[13] / nmb, h[14] / nmb, h[15] / nmb);
    fprintf(stderr, "WAVE PROF cycles per macroblock (%.0f MBs): total %.0f | predictor %.0f integer %.0f subpel %.0f skip %.0f | epzs: lists %.0f scan+eval %.0f refine %.0f | eval_dist: %.0f cycles in %.1f calls, %.1f candidates\n", nmb, h[8] / nmb,
            h[0] / nmb, h[1] / nmb, h[2] / nmb, h[3] / nmb, h[4] / nmb, h[5] / nmb, h[6] / nmb, h[10] / nmb, h[11] / nmb, h[12] / nmb);
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wave_prof), z, sizeof(z));
  }
#endif
  if (!x_settled) std::swap(s->carry_slice, s->carry_slice_next);      // (the exhaustive searches neither read nor write img->all_mv's carry)
  if (prm->mb_first == 0) s->t8_any = false;
  s->t8_any = s->t8_any || prm->transform8x8_mode != 0;
  // slices in coding order extend the searched range; anything else starts a new one (a rank of the slice-parallel layout searches only its band)
  if (prm->mb_first != 0 && prm->mb_first == s->searched_to) s->searched_to = prm->mb_first + prm->mb_count;
  else { s->searched_from = prm->mb_first; s->searched_to = prm->mb_first + prm->mb_count; }
  if (results) return jmhip_slice_results_download(c, results, prm->mb_first, prm->mb_count);
  return JMHIP_OK;
}

extern "C" int jmhip_slice_results_download(jmhip_ctx *c, jmhip_mb_inter *results, int mb_first, int mb_count)
{
  if (!c || !results || mb_first < 0 || mb_count < 1 || mb_first + mb_count > c->mbw * c->mbh || !c->slice_state) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_slice_results_download: arguments") : JMHIP_ERR_ARG;
  SliceState *s = static_cast<SliceState *>(c->slice_state);
  JM_HIP_CHECK(c, hipMemcpyAsync(results, s->out + mb_first, sizeof(jmhip_mb_inter) * (size_t)mb_count, hipMemcpyDeviceToHost, c->stream));
  JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  return JMHIP_OK;
}

extern "C" int jmhip_slice_field_download(jmhip_ctx *c, int8_t *ref_idx, int16_t *mv)
{
  if (!c || !c->slice_state) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_slice_field_download: no slice has been searched") : JMHIP_ERR_ARG;
  SliceState *s = static_cast<SliceState *>(c->slice_state);
  const size_t n = (size_t)(c->W / 4) * (c->H / 4);
  if (ref_idx) JM_HIP_CHECK(c, hipMemcpyAsync(ref_idx, s->ref_idx, n, hipMemcpyDeviceToHost, c->stream));
  if (mv) JM_HIP_CHECK(c, hipMemcpyAsync(mv, s->mv, n * 4, hipMemcpyDeviceToHost, c->stream));
  JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  return JMHIP_OK;
}

extern "C" int jmhip_slice_result_info(jmhip_ctx *c, int *passes)
{
  if (!c || !c->slice_state) return JMHIP_ERR_ARG;
  if (passes) *passes = static_cast<SliceState *>(c->slice_state)->passes;
  return JMHIP_OK;
}

extern "C" int jmhip_epzs_map_upload(jmhip_ctx *c, const int16_t *map, int search_range, int blk_count)
{
  if (!c || search_range < 1 || search_range > 33) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_epzs_map_upload: arguments") : JMHIP_ERR_ARG;
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  SliceState *s = slice_state(c);
  if (!s) return jm_fail(c, JMHIP_ERR_NOMEM, "slice search state");
  const int rc = ep_state_ensure(c, s, search_range);
  if (rc) return rc;
  const int side = 2 * search_range + 1;
  const uint32_t count = (uint16_t)blk_count;
  JM_HIP_CHECK(c, hipMemsetAsync(s->ep_map, 0, sizeof(uint16_t) * s->ep_cells, c->stream));
  if (map) JM_HIP_CHECK(c, hipMemcpyAsync(s->ep_map, map, sizeof(int16_t) * side * side, hipMemcpyHostToDevice, c->stream));
  JM_HIP_CHECK(c, hipMemcpyAsync(s->ep_count, &count, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  return JMHIP_OK;
}

extern "C" int jmhip_epzs_map_info(jmhip_ctx *c, int *aliased_tests, uint32_t *searches)
{
  if (!c || !c->slice_state) return JMHIP_ERR_ARG;
  SliceState *s = static_cast<SliceState *>(c->slice_state);
  if (aliased_tests) *aliased_tests = s->ep_aliases;
  if (searches) {
    *searches = 0;
    if (s->ep_count) {
      JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
      JM_HIP_CHECK(c, hipMemcpyAsync(searches, s->ep_count, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
      JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    }
  }
  return JMHIP_OK;
}

// ---------------------------------------------------------------------------------------------- hand-over to the frame stage

namespace {
// one thread per macroblock: the slice search's record -> the search-stage result layout the frame stage reads (vector per partition: the
// one of the reference its 8x8 block settled on), the decided mode, the reference slot per 8x8 block
struct SlotMap { int s[WR]; };
__global__ void slice_to_frame_kernel(const jmhip_mb_inter *__restrict__ rec, int first, int n, int mbw, SlotMap sm,
                                      jmhip_me_mb *__restrict__ jobs, jmhip_me_result *__restrict__ res, jmhip_mb_mode *__restrict__ modes, int8_t *__restrict__ blk_ref, int candidates)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const jmhip_mb_inter &r = rec[first + i];              // job i of the frame stage = macroblock first + i of the picture
  const int *slots = sm.s;
  // the decided mode, or (candidates) the P8x8 candidate of submacroblock_mode_decision: its sub-modes and references per 8x8 block
  const bool p8 = candidates && r.p8mode[0] >= 4;
  const int32_t *bref = p8 ? r.p8ref : r.b8ref;
  jmhip_me_mb &j = jobs[i];
  j.mb_x = (int16_t)((first + i) % mbw); j.mb_y = (int16_t)((first + i) / mbw); j.ref = (int16_t)slots[bref[0]]; j.ref_is_0 = (int16_t)(bref[0] == 0);
  jmhip_me_result &o = res[i];
  for (int p = 0; p < JMHIP_NPART; p++) {
    const PartInfo q = c_part[p];
    const int ref = bref[2 * (q.y4 >> 1) + (q.x4 >> 1)];
    if (!p8 && r.best_mode == 8 && r.transform8x8_flag && p >= 5 && p < 9) {        // a P8x8 macroblock with the 8x8 transform keeps the vectors of its 8x8-transform pass
      const int k = p - 5;
      j.pred_mv[p][0] = r.pred8ts[ref][k][0]; j.pred_mv[p][1] = r.pred8ts[ref][k][1];
      o.mv[p][0] = r.mv8ts[ref][k][0]; o.mv[p][1] = r.mv8ts[ref][k][1]; o.cost[p] = r.cost8ts[ref][k];
      o.mv_int[p][0] = r.mv_int8ts[ref][k][0]; o.mv_int[p][1] = r.mv_int8ts[ref][k][1]; o.cost_int[p] = r.cost_int8ts[ref][k];
      continue;
    }
    j.pred_mv[p][0] = r.pred[ref][p][0]; j.pred_mv[p][1] = r.pred[ref][p][1];
    o.mv[p][0] = r.mv[ref][p][0]; o.mv[p][1] = r.mv[ref][p][1]; o.cost[p] = r.cost[ref][p];
    o.mv_int[p][0] = r.mv_int[ref][p][0]; o.mv_int[p][1] = r.mv_int[ref][p][1]; o.cost_int[p] = r.cost_int[ref][p];
  }
  jmhip_mb_mode m;
  m.mode = (int8_t)(p8 ? 8 : r.best_mode);
  // (P skip is not a mode of the rdopt = 0 decision: md_low.c:655 turns a 16x16 macroblock into a skip AFTER residual coding, when cbp == 0,
  // ref_idx == 0 and the vector equals skip_mv -- the caller has all three)
  for (int k = 0; k < 4; k++) { m.b8mode[k] = (int8_t)(p8 ? r.p8mode[k] : (r.best_mode == 8 ? r.b8mode[k] : 4)); blk_ref[(size_t)i * 4 + k] = (int8_t)slots[bref[k]]; }
  m.pad[0] = (int8_t)(!p8 && r.transform8x8_flag ? 1 : 0);      // luma_transform_size_8x8_flag as the decision left it (the residual coder's transform)
  m.pad[1] = m.pad[2] = 0;
  modes[i] = m;
}
}  // namespace

extern "C" int jmhip_slice_to_frame(jmhip_ctx *c, const int32_t *ref_slot, int num_refs)
{
  if (!c) return JMHIP_ERR_ARG;
  return jmhip_slice_to_frame_band(c, ref_slot, num_refs, 0, c->mbw * c->mbh);
}

static int slice_to_frame(jmhip_ctx *c, const int32_t *ref_slot, int num_refs, int mb_first, int mb_count, int candidates);
extern "C" int jmhip_slice_to_frame_band(jmhip_ctx *c, const int32_t *ref_slot, int num_refs, int mb_first, int mb_count) { return slice_to_frame(c, ref_slot, num_refs, mb_first, mb_count, 0); }
extern "C" int jmhip_slice_to_frame_candidates(jmhip_ctx *c, const int32_t *ref_slot, int num_refs, int mb_first, int mb_count) { return slice_to_frame(c, ref_slot, num_refs, mb_first, mb_count, 1); }
static int slice_to_frame(jmhip_ctx *c, const int32_t *ref_slot, int num_refs, int mb_first, int mb_count, int candidates)
{
  if (!c || !ref_slot || num_refs < 1 || num_refs > JMHIP_SLICE_REFS) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_slice_to_frame: arguments") : JMHIP_ERR_ARG;
  if (!c->slice_state) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_slice_to_frame: no slice has been searched");
  SliceState *s = static_cast<SliceState *>(c->slice_state);
  const int n = mb_count;
  if (mb_first < 0 || mb_count < 1 || mb_first + mb_count > c->mbw * c->mbh) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_slice_to_frame: macroblock range outside the picture");
  if (mb_first < s->searched_from || mb_first + mb_count > s->searched_to) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_slice_to_frame: not every macroblock of the range has been searched in the current picture (slices must cover it in order)");
  unsigned mask = 0;
  SlotMap sm{};
  for (int r = 0; r < num_refs; r++) {
    if (ref_slot[r] < 0 || ref_slot[r] >= (int)c->refs.size() || ref_slot[r] >= 8) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_slice_to_frame: reference slots 0..7");
    sm.s[r] = ref_slot[r]; mask |= 1u << ref_slot[r];
  }
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  int rc = jm_me_arrays_ensure(c, n);
  if (rc) return rc;
  if ((rc = jm_frame_buffers_ensure(c, n))) return rc;
  build_part_table();
  {                                                      // c_part is a per-device symbol of this translation unit: one upload per device, as ensure_tables() (me_int.hip)
    static bool part_uploaded[64] = {false};
    const int dev = c->cfg.device;
    if (dev < 0 || dev >= 64 || !part_uploaded[dev]) {
      JM_HIP_CHECK(c, hipMemcpyToSymbol(HIP_SYMBOL(c_part), h_part, sizeof(h_part)));
      if (dev >= 0 && dev < 64) part_uploaded[dev] = true;
    }
  }
  slice_to_frame_kernel<<<(n + 127) / 128, 128, 0, c->stream>>>(s->out, mb_first, n, c->mbw, sm, (jmhip_me_mb *)c->me_jobs_dev,
                                                                (jmhip_me_result *)c->me_res_dev, (jmhip_mb_mode *)c->fr_modes + n, (int8_t *)c->fr_blk_ref, candidates);
  JM_HIP_CHECK(c, hipGetLastError());
  // the search-stage arrays now hold this picture; a resident re-run of jmhip_me_frame on them is meaningless and is refused (geometry check)
  c->me_n = n; c->me_ref_mask = mask; c->me_last_mode = 0x7fffffff; c->me_fast_idx.clear(); c->me_gen_idx.clear();
  c->fr_from_slices = true;
  c->fr_slices_t8 = s->t8_any && !candidates;      // (the candidates are coded with the 4x4 transform: the P8x8 pass of the 4x4 transform)
  return JMHIP_OK;
}

void jm_slice_state_free(jmhip_ctx *c)
{
  SliceState *s = static_cast<SliceState *>(c->slice_state);
  if (!s) return;
  slice_state_release(s);
  c->slice_state = nullptr;
}

// ---------------------------------------------------------------------------------------------- host helpers: JM's initialisation arithmetic

static int round_log2(int v) { int r = 0; const int sq = v * v; while ((1 << (r + 1)) <= sq) r++; return (r + 1) >> 1; }      // me_epzs.c:241

// EPZSInit, me_epzs.c:333 (8-bit video, ChromaMEEnable 0: chroma_weight = 0); EPZSWindowPredictorInit :261 (EPZSSubPelGrid 0)
extern "C" void jmhip_epzs_setup(jmhip_slice_params *p, int search_range, int pattern, int dual, int fixed, int temporal, int spatial_mem, int subpel_me,
                                 int min_scale, int med_scale, int max_scale, int subpel_scale)
{
  static const int minb[8] = {0, 64, 32, 32, 16, 8, 8, 4}, medb[8] = {0, 256, 128, 128, 64, 32, 32, 16}, maxb[8] = {0, 768, 384, 384, 192, 96, 96, 48};
  p->epzs_pattern = pattern; p->epzs_dual = dual; p->epzs_fixed = fixed; p->epzs_temporal = temporal; p->epzs_spatial_mem = spatial_mem; p->epzs_subpel_me = subpel_me;
  for (int i = 0; i < 8; i++) {
    p->epzs_thres[0][i] = min_scale * minb[i]; p->epzs_thres[1][i] = med_scale * medb[i];
    p->epzs_thres[2][i] = max_scale * maxb[i]; p->epzs_thres[3][i] = subpel_scale * medb[i];
  }
  for (int mode = 0; mode < 2; mode++) {
    int n = 0;
    int16_t (*pt)[2] = mode ? p->epzs_win_ext : p->epzs_win;
    for (int pos = round_log2(search_range) - 2; pos > -1; pos--) {
      const int sp = search_range >> pos, fsp = (3 * sp + 1) >> 1;
      for (int i = 1; i >= -1; i -= 2) {
        pt[n][0] = (int16_t)(i * sp); pt[n++][1] = 0;
        pt[n][0] = (int16_t)(i * sp); pt[n++][1] = (int16_t)(i * sp);
        pt[n][0] = 0; pt[n++][1] = (int16_t)(i * sp);
        pt[n][0] = (int16_t)(-i * sp); pt[n++][1] = (int16_t)(i * sp);
      }
      if (mode)
        for (int i = 1; i >= -1; i -= 2) {
          pt[n][0] = (int16_t)(i * fsp); pt[n++][1] = (int16_t)(-i * sp);
          pt[n][0] = (int16_t)(i * fsp); pt[n++][1] = 0;
          pt[n][0] = (int16_t)(i * fsp); pt[n++][1] = (int16_t)(i * sp);
          pt[n][0] = (int16_t)(i * sp); pt[n++][1] = (int16_t)(i * fsp);
          pt[n][0] = 0; pt[n++][1] = (int16_t)(i * fsp);
          pt[n][0] = (int16_t)(-i * sp); pt[n++][1] = (int16_t)(i * fsp);
        }
    }
    if (mode) p->epzs_nwin_ext = n; else p->epzs_nwin = n;
  }
}

// the reference-to-reference scales of EPZSSliceInit, me_epzs.c:516-547
extern "C" void jmhip_epzs_scales(jmhip_slice_params *p, int poc, const int *list0_poc, int num_refs)
{
  auto clip3 = [](int lo, int hi, int x) { return x < lo ? lo : (x > hi ? hi : x); };
  for (int k = 0; k < num_refs && k < JMHIP_SLICE_REFS; k++)
    for (int i = 0; i < num_refs && i < JMHIP_SLICE_REFS; i++) {
      const int iTRb = clip3(-128, 127, poc - list0_poc[i]), iTRp = clip3(-128, 127, poc - list0_poc[k]);
      if (iTRp != 0) {
        const int prescale = (16384 + std::abs(iTRp / 2)) / iTRp;
        p->epzs_mv_scale[i][k] = clip3(-2048, 2047, (iTRb * prescale + 32) >> 6);
      } else p->epzs_mv_scale[i][k] = 256;
    }
}

// UMHEX_DefineThreshold + UMHEX_DefineThresholdMB, me_umhex.c:78-146: the same expressions in single precision
extern "C" void jmhip_umhex_setup(jmhip_slice_params *p, int dsr, int scale, int qp_n, int width)
{
  static const int qc00[6] = {13107, 11916, 10082, 9362, 8192, 7282};                              // quant_coef[rem][0][0], me_umhex.c:68-75
  static const int mref[8] = {0, 300, 120, 120, 60, 30, 30, 15}, bhex[8] = {0, 3000, 1500, 1500, 800, 400, 400, 200};
  static const int medp[8] = {0, 750, 350, 350, 170, 80, 80, 40}, tdsr[8] = {0, 2200, 1000, 1000, 500, 250, 250, 120};
  static const float a1[8] = {0, 0.01f, 0.01f, 0.01f, 0.02f, 0.03f, 0.03f, 0.04f}, a2[8] = {0, 0.06f, 0.07f, 0.07f, 0.08f, 0.12f, 0.11f, 0.15f};
  p->umhex_dsr = dsr;
  const int gb_qp_per = qp_n / 6, gb_qp_rem = qp_n % 6, gb_q_bits = 15 + gb_qp_per;
  volatile float scale_factor = (float)((1 - scale * 0.1) + scale * 0.1 * (width / 176));
  volatile float QP_factor = (float)((1.0 - 0.90 * (qp_n / 51.0f)));
  const int gb_qp_const = (1 << gb_q_bits) / 6;
  const int Thresh4x4 = ((1 << gb_q_bits) - gb_qp_const) / qc00[gb_qp_rem];
  volatile float Quantize_step = Thresh4x4 / (4 * 5.61f) * 2.0f * scale_factor;
  volatile float b[8];
  b[7] = (16 * 16) * Quantize_step;
  b[6] = b[7] * 4; b[5] = b[7] * 4; b[4] = b[5] * 4; b[3] = b[4] * 4; b[2] = b[4] * 4; b[1] = b[2] * 4; b[0] = 0;
  for (int i = 0; i < 8; i++) { p->umhex_bsize[i] = b[i]; p->umhex_alpha1[i] = a1[i]; p->umhex_alpha2[i] = a2[i]; }
  for (int i = 1; i < 8; i++) {
    volatile float t;
    t = medp[i] * scale_factor; t = t * QP_factor; p->umhex_thres[0][i] = (int)t;
    t = bhex[i] * scale_factor; t = t * QP_factor; p->umhex_thres[1][i] = (int)t;
    t = mref[i] * scale_factor; t = t * QP_factor; p->umhex_thres[2][i] = (int)t;
    t = tdsr[i] * scale_factor; t = t * QP_factor; p->umhex_thres[3][i] = (int)t;
  }
  p->umhex_thres[0][0] = p->umhex_thres[1][0] = p->umhex_thres[2][0] = p->umhex_thres[3][0] = 0;
}
