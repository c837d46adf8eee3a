// tq.hip -- batched integer transform + quantisation + dequantisation + inverse transform + reconstruction.
//
// Replaces, per block of a batch:
//   dct_4x4    lencod/src/block.c:843-947          (via pointer pDCT_4x4, inc/global.h:1376)
//   dct_8x8    lencod/src/transform8x8.c:1452-1653
//   dct_16x16  lencod/src/block.c:564-826
//   dct_chroma lencod/src/block.c:1051-1495        (4:2:0 2x2 DC :1125-1174, 4:2:2 2x4 DC :1205-1319)
//   forward4x4/inverse4x4/hadamard4x4/ihadamard4x4/forward8x8/inverse8x8  lencod/src/transform.c:31-421
// Lossless (qpprime) and SP-slice variants are not part of the path and are rejected on the host.
//
// JM quirks mirrored: dct_chroma's forward4x4(curr_res,curr_res,n1,n2) row/column swap (block.c:1120) -- for
// 4:2:2 it transforms rows 0..7 x cols 0..15 of the 16x16 tile, so the job carries the whole tile; the 4:2:2 DC
// quantiser mixes the AC level scale with the qp+3 offset table (block.c:1263); CAVLC_LEVEL_LIMIT only clamps DC
// levels (block.c:647, :1147); _CHROMA_COEFF_COST_ thresholding zeroes AC levels but leaves the runs (:1384-1410).
//
// Roofline: HBM-bound once batched (no data reuse, ~40 integer ops per coefficient). One lane owns one block
// (4x4: 16 lanes per macroblock), all butterflies are 32-bit integer add/shift in registers -- no MFMA: this is
// not a dense contraction (the 4x4 "matrix" has entries +-1, +-2 applied as shifts).
#include "jmhip_internal.h"
#include "frame_common.h"
#include <cstddef>

namespace {

constexpr int Q_BITS = 15, Q_BITS_8 = 16, DQ_BITS = 6, MAXV = 999999, CAVLC_LEVEL_LIMIT = 2063;

__constant__ uint8_t c_scan4[2][16][2] = {
  {{0,0},{1,0},{0,1},{0,2},{1,1},{2,0},{3,0},{2,1},{1,2},{0,3},{1,3},{2,2},{3,1},{3,2},{2,3},{3,3}},      // SNGL_SCAN  block.h:26
  {{0,0},{0,1},{1,0},{0,2},{0,3},{1,1},{1,2},{1,3},{2,0},{2,1},{2,2},{2,3},{3,0},{3,1},{3,2},{3,3}}};     // FIELD_SCAN block.h:35
__constant__ uint8_t c_scan8[2][64][2] = {
  {{0,0},{1,0},{0,1},{0,2},{1,1},{2,0},{3,0},{2,1},{1,2},{0,3},{0,4},{1,3},{2,2},{3,1},{4,0},{5,0},
   {4,1},{3,2},{2,3},{1,4},{0,5},{0,6},{1,5},{2,4},{3,3},{4,2},{5,1},{6,0},{7,0},{6,1},{5,2},{4,3},
   {3,4},{2,5},{1,6},{0,7},{1,7},{2,6},{3,5},{4,4},{5,3},{6,2},{7,1},{7,2},{6,3},{5,4},{4,5},{3,6},
   {2,7},{3,7},{4,6},{5,5},{6,4},{7,3},{7,4},{6,5},{5,6},{4,7},{5,7},{6,6},{7,5},{7,6},{6,7},{7,7}},      // transform8x8.c:171
  {{0,0},{0,1},{0,2},{1,0},{1,1},{0,3},{0,4},{1,2},{2,0},{1,3},{0,5},{0,6},{0,7},{1,4},{2,1},{3,0},
   {2,2},{1,5},{1,6},{1,7},{2,3},{3,1},{4,0},{3,2},{2,4},{2,5},{2,6},{2,7},{3,3},{4,1},{5,0},{4,2},
   {3,4},{3,5},{3,6},{3,7},{4,3},{5,1},{6,0},{5,2},{4,4},{4,5},{4,6},{4,7},{5,3},{6,1},{6,2},{5,4},
   {5,5},{5,6},{5,7},{6,3},{7,0},{7,1},{6,4},{6,5},{6,6},{6,7},{7,2},{7,3},{7,4},{7,5},{7,6},{7,7}}};     // transform8x8.c:184
__constant__ uint8_t c_cost4[2][16] = {{3,2,2,1,1,1,0,0,0,0,0,0,0,0,0,0}, {9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9}};   // block.h:45
__constant__ uint8_t c_cost8[2][64] = {
  {3,3,3,3,2,2,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0},
  {9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9}};
__constant__ uint8_t c_scan422[8][2] = {{0,0},{0,1},{1,0},{0,2},{0,3},{1,1},{1,2},{1,3}};                 // block.h:52
__constant__ uint8_t c_hor[4][4][4] = {{{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}}, {{0,4,0,4},{0,0,0,0},{0,0,0,0},{0,0,0,0}},
                                       {{0,4,0,4},{0,4,0,4},{0,0,0,0},{0,0,0,0}}, {{0,4,0,4},{8,12,8,12},{0,4,0,4},{8,12,8,12}}};   // block.h:61
__constant__ uint8_t c_ver[4][4][4] = {{{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}}, {{0,0,4,4},{0,0,0,0},{0,0,0,0},{0,0,0,0}},
                                       {{0,0,4,4},{8,8,12,12},{0,0,0,0},{0,0,0,0}}, {{0,0,4,4},{0,0,4,4},{8,8,12,12},{8,8,12,12}}}; // block.h:85

__device__ __forceinline__ int iabs(int x) { return x < 0 ? -x : x; }
__device__ __forceinline__ int sgnab(int a, int b) { return b < 0 ? -iabs(a) : iabs(a); }      // isignab
__device__ __forceinline__ int rsr(int x, int a) { return (x + (1 << (a - 1))) >> a; }          // rshift_rnd_sf
// iClip1 -- the empty asm keeps hipcc from forming v_ashr_pk_u8_i32 (see interp_luma.hip)
__device__ __forceinline__ int clip1(int hi, int v) { asm volatile("" : "+v"(v)); return min(max(v, 0), hi); }

// forward4x4 / inverse4x4 on a register block b[row][col]   (transform.c:31, :81)
__device__ __forceinline__ void fwd4(int b[4][4])
{
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int t0 = b[i][0] + b[i][3], t1 = b[i][1] + b[i][2], t2 = b[i][1] - b[i][2], t3 = b[i][0] - b[i][3];
    b[i][0] = t0 + t1; b[i][1] = (t3 << 1) + t2; b[i][2] = t0 - t1; b[i][3] = t3 - (t2 << 1);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int t0 = b[0][i] + b[3][i], t1 = b[1][i] + b[2][i], t2 = b[1][i] - b[2][i], t3 = b[0][i] - b[3][i];
    b[0][i] = t0 + t1; b[1][i] = t2 + (t3 << 1); b[2][i] = t0 - t1; b[3][i] = t3 - (t2 << 1);
  }
}
__device__ __forceinline__ void inv4(int b[4][4])
{
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int p0 = b[i][0] + b[i][2], p1 = b[i][0] - b[i][2], p2 = (b[i][1] >> 1) - b[i][3], p3 = b[i][1] + (b[i][3] >> 1);
    b[i][0] = p0 + p3; b[i][1] = p1 + p2; b[i][2] = p1 - p2; b[i][3] = p0 - p3;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int p0 = b[0][i] + b[2][i], p1 = b[0][i] - b[2][i], p2 = (b[1][i] >> 1) - b[3][i], p3 = b[1][i] + (b[3][i] >> 1);
    b[0][i] = p0 + p3; b[1][i] = p1 + p2; b[2][i] = p1 - p2; b[3][i] = p0 - p3;
  }
}

// generic (memory tile) versions used by the rarer kinds; m is an int[16][16] tile
__device__ void fwd4_tile(int (*m)[16], int py, int px)
{
  int b[4][4];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int i = 0; i < 4; i++) b[j][i] = m[py + j][px + i];
  fwd4(b);
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int i = 0; i < 4; i++) m[py + j][px + i] = b[j][i];
}
__device__ void inv4_tile(int (*m)[16], int py, int px)
{
  int b[4][4];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int i = 0; i < 4; i++) b[j][i] = m[py + j][px + i];
  inv4(b);
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int i = 0; i < 4; i++) m[py + j][px + i] = b[j][i];
}

__device__ __forceinline__ void fwd8_1d(const int p[8], int o[8])      // transform.c:248-275
{
  int a0 = p[0] + p[7], a1 = p[1] + p[6], a2 = p[2] + p[5], a3 = p[3] + p[4];
  const int b0 = a0 + a3, b1 = a1 + a2, b2 = a0 - a3, b3 = a1 - a2;
  a0 = p[0] - p[7]; a1 = p[1] - p[6]; a2 = p[2] - p[5]; a3 = p[3] - p[4];
  const int b4 = a1 + a2 + ((a0 >> 1) + a0), b5 = a0 - a3 - ((a2 >> 1) + a2);
  const int b6 = a0 + a3 - ((a1 >> 1) + a1), b7 = a1 - a2 + ((a3 >> 1) + a3);
  o[0] = b0 + b1; o[1] = b4 + (b7 >> 2); o[2] = b2 + (b3 >> 1); o[3] = b5 + (b6 >> 2);
  o[4] = b0 - b1; o[5] = b6 - (b5 >> 2); o[6] = (b2 >> 1) - b3; o[7] = (b4 >> 2) - b7;
}
__device__ __forceinline__ void inv8_1d(const int p[8], int o[8])      // transform.c:346-373
{
  int a0 = p[0] + p[4], a1 = p[0] - p[4], a2 = p[6] - (p[2] >> 1), a3 = p[2] + (p[6] >> 1);
  const int b0 = a0 + a3, b2 = a1 - a2, b4 = a1 + a2, b6 = a0 - a3;
  a0 = -p[3] + p[5] - p[7] - (p[7] >> 1);
  a1 =  p[1] + p[7] - p[3] - (p[3] >> 1);
  a2 = -p[1] + p[7] + p[5] + (p[5] >> 1);
  a3 =  p[3] + p[5] + p[1] + (p[1] >> 1);
  const int b1 = a0 + (a3 >> 2), b3 = a1 + (a2 >> 2), b5 = a2 - (a1 >> 2), b7 = a3 - (a0 >> 2);
  o[0] = b0 + b7; o[1] = b2 - b5; o[2] = b4 + b3; o[3] = b6 + b1;
  o[4] = b6 - b1; o[5] = b4 - b3; o[6] = b2 + b5; o[7] = b0 - b7;
}
__device__ void xf8_tile(int (*m)[16], int py, int px, bool inverse)  // forward8x8 :229 / inverse8x8 :325
{
  int t[8][8];
  for (int j = 0; j < 8; j++) {
    int p[8], o[8];
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = m[py + j][px + i];
    if (inverse) inv8_1d(p, o); else fwd8_1d(p, o);
#pragma unroll
    for (int i = 0; i < 8; i++) t[j][i] = o[i];
  }
  for (int i = 0; i < 8; i++) {
    int p[8], o[8];
#pragma unroll
    for (int j = 0; j < 8; j++) p[j] = t[j][i];
    if (inverse) inv8_1d(p, o); else fwd8_1d(p, o);
#pragma unroll
    for (int j = 0; j < 8; j++) m[py + j][px + i] = o[j];
  }
}

static_assert(offsetof(jmhip_tq_result, fadjust) % 8 == 0 && sizeof(jmhip_tq_result) % 8 == 0, "fadjust rows are written as int2");

// ---------------------------------------------------------------------------------------- dct_4x4: one lane per 4x4 block

// `select`: frame stage -- a job marked as an 8x8-transform macroblock (intra16_unused != 0) belongs to tq_luma8x8_kernel
__global__ __launch_bounds__(256) void tq_luma4x4_kernel(const jmhip_tq_job *__restrict__ jobs, const jmhip_quant *__restrict__ quants,
                                                        jmhip_tq_result *__restrict__ res, int n, int select)
{
  // XCD-contiguous block order (as mc_kernel and finalize_kernel): a job tile is read from the L2 it was written into
  const int vb = jm_xcd_item((n * 16 + 255) / 256);
  if (vb < 0) return;
  const int gid = vb * 256 + threadIdx.x;
  const int jobi = gid >> 4, blk = gid & 15;                 // blk = b8*4 + b4 (JM block order)
  if (jobi >= n) return;
  const jmhip_tq_job &job = jobs[jobi];
  if (select && job.intra16_unused) return;
  const jmhip_quant &q = quants[job.quant];
  jmhip_tq_result &o = res[jobi];
  const int b8 = blk >> 2, b4 = blk & 3;
  const int bx = 8 * (b8 & 1) + 4 * (b4 & 1), by = 8 * (b8 >> 1) + 4 * (b4 >> 1);
  const int qp_per = q.qp / 6, q_bits = Q_BITS + qp_per;

  int m[4][4], pr[4][4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t s = *reinterpret_cast<const uint32_t *>(&job.src[by + j][bx]);
    const uint32_t p = *reinterpret_cast<const uint32_t *>(&job.pred[by + j][bx]);
#pragma unroll
    for (int i = 0; i < 4; i++) { pr[j][i] = (p >> (8 * i)) & 255; m[j][i] = (int)((s >> (8 * i)) & 255) - pr[j][i]; }   // img->m7, macroblock.c:1059-1068
  }
  fwd4(m);

  int scan_pos = 0, run = -1, nonzero = 0, cost = 0;
  int fa[4][4];                                      // fadjust of this block, stored as 8-byte pairs at the end
  int *levels = o.levels[blk], *runs = o.runs[blk];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    // static (i,j) per scan position for both scan orders (block.h:26-41)
    constexpr int I0[16] = {0,1,0,0,1,2,3,2,1,0,1,2,3,3,2,3}, J0[16] = {0,0,1,2,1,0,0,1,2,3,3,2,1,2,3,3};
    constexpr int I1[16] = {0,0,1,0,0,1,1,1,2,2,2,2,3,3,3,3}, J1[16] = {0,1,0,2,3,1,2,3,0,1,2,3,0,1,2,3};
    const int i0 = I0[k], j0 = J0[k], i1 = I1[k], j1 = J1[k];
    // field scan selects a different register: evaluate both statically and pick (uniform per job)
    const int c0 = m[j0][i0], c1 = m[j1][i1];
    const int c = q.field_scan ? c1 : c0;
    const int idx = q.field_scan ? (j1 * 4 + i1) : (j0 * 4 + i0);
    run++;
    const int scaled = iabs(c) * q.levelscale[idx];
    int level = (scaled + q.leveloffset[idx]) >> q_bits;
    int deq = 0, fadj = 0;
    if (level != 0) {
      if (q.adaptive_rounding) fadj = rsr(q.adapt_rnd_weight * (scaled - (level << q_bits)), q_bits + 1);   // block.c:898
      nonzero = 1;
      cost += (level > 1) ? MAXV : c_cost4[q.disthres][run];
      level = sgnab(level, c);
      levels[scan_pos] = level; runs[scan_pos] = run; scan_pos++;
      deq = rsr((level * q.invlevelscale[idx]) << qp_per, 4);                                             // block.c:907
      run = -1;
    }
    if (q.field_scan) { m[j1][i1] = deq; fa[j1][i1] = fadj; } else { m[j0][i0] = deq; fa[j0][i0] = fadj; }
  }
  levels[scan_pos] = 0;
  o.coeff_cost[blk] = cost; o.nonzero[blk] = nonzero;
  if (q.adaptive_rounding) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      int2 *d = reinterpret_cast<int2 *>(&o.fadjust[by + j][bx]);       // the struct keeps this 8-byte aligned
      d[0] = make_int2(fa[j][0], fa[j][1]); d[1] = make_int2(fa[j][2], fa[j][3]);
    }
  }

  if (scan_pos) inv4(m);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    uint32_t w = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int v = scan_pos ? clip1(q.max_val, rsr(m[j][i], DQ_BITS) + pr[j][i]) : pr[j][i];                // block.c:934 / :942
      w |= (uint32_t)v << (8 * i);
    }
    *reinterpret_cast<uint32_t *>(&o.recon[by + j][bx]) = w;
  }
}

// ---------------------------------------------------------------------------------------- generic tile kinds: one lane per job

__device__ void load_residual(const jmhip_tq_job &job, int (*m7)[16], int rows, int cols)
{
  for (int j = 0; j < rows; j++) for (int i = 0; i < cols; i++) m7[j][i] = (int)job.src[j][i] - (int)job.pred[j][i];
}

// dct_8x8, transform8x8.c:1452 (non-lossless). One lane per 8x8 block.
__global__ __launch_bounds__(64) void tq_luma8x8_kernel(const jmhip_tq_job *__restrict__ jobs, const jmhip_quant *__restrict__ quants,
                                                       jmhip_tq_result *__restrict__ res, int n, int select)
{
  const int gid = blockIdx.x * 64 + threadIdx.x;
  const int jobi = gid >> 2, b8 = gid & 3;
  if (jobi >= n) return;
  const jmhip_tq_job &job = jobs[jobi];
  if (select && !job.intra16_unused) return;
  const jmhip_quant &q = quants[job.quant];
  jmhip_tq_result &o = res[jobi];
  const int block_x = 8 * (b8 & 1), block_y = 8 * (b8 >> 1);
  const int qp_per = q.qp / 6, q_bits = Q_BITS_8 + qp_per;
  const bool interleave = q.transform8x8_flag && q.cavlc;            // transform8x8.c:1502
  int m7[16][16];
  for (int j = 0; j < 8; j++) for (int i = 0; i < 8; i++)
    m7[block_y + j][block_x + i] = (int)job.src[block_y + j][block_x + i] - (int)job.pred[block_y + j][block_x + i];
  xf8_tile(m7, block_y, block_x, false);

  int scan_poss[4] = {0, 0, 0, 0}, runs4[4] = {-1, -1, -1, -1};
  int scan_pos = 0, run = -1, nonzero = 0, cost = 0, mc = 0;
  for (int k = 0; k < 64; k++) {
    const int i = c_scan8[q.field_scan][k][0], j = c_scan8[q.field_scan][k][1];
    run++;
    if (interleave) { mc = k & 3; runs4[mc]++; }
    int *c = &m7[block_y + j][block_x + i];
    const int scaled = iabs(*c) * q.levelscale[j * 8 + i];
    int level = (scaled + q.leveloffset[j * 8 + i]) >> q_bits;
    if (level != 0) {
      if (q.adaptive_rounding) o.fadjust[block_y + j][block_x + i] = rsr(q.adapt_rnd_weight * (scaled - (level << q_bits)), q_bits + 1);
      nonzero = 1;
      if (interleave) {
        cost += (level > 1) ? MAXV : c_cost8[q.disthres][runs4[mc]];
        o.levels[4 * b8 + mc][scan_poss[mc]] = sgnab(level, *c);
        o.runs[4 * b8 + mc][scan_poss[mc]] = runs4[mc];
        scan_poss[mc]++; runs4[mc] = -1;
      } else {
        cost += (level > 1) ? MAXV : c_cost8[q.disthres][run];
        o.levels8[b8][scan_pos] = sgnab(level, *c);
        o.runs8[b8][scan_pos] = run;
        scan_pos++; run = -1;
      }
      level = sgnab(level, *c);
      *c = rsr((level * q.invlevelscale[j * 8 + i]) << qp_per, 6);
    } else {
      if (q.adaptive_rounding) o.fadjust[block_y + j][block_x + i] = 0;
      *c = 0;
    }
  }
  if (!interleave) o.levels8[b8][scan_pos] = 0;
  else for (int k = 0; k < 4; k++) o.levels[4 * b8 + k][scan_poss[k]] = 0;
  o.coeff_cost[b8] = cost; o.nonzero[b8] = nonzero;
  if (nonzero) xf8_tile(m7, block_y, block_x, true);
  for (int j = block_y; j < block_y + 8; j++) for (int i = block_x; i < block_x + 8; i++)
    o.recon[j][i] = (uint8_t)(nonzero ? clip1(q.max_val, rsr(m7[j][i], DQ_BITS) + job.pred[j][i]) : job.pred[j][i]);
}

// dct_16x16, block.c:564 (non-lossless). One lane per macroblock.
__global__ __launch_bounds__(64) void tq_luma16x16_kernel(const jmhip_tq_job *__restrict__ jobs, const jmhip_quant *__restrict__ quants,
                                                         jmhip_tq_result *__restrict__ res, int n)
{
  const int jobi = blockIdx.x * 64 + threadIdx.x;
  if (jobi >= n) return;
  const jmhip_tq_job &job = jobs[jobi];
  const jmhip_quant &q = quants[job.quant];
  jmhip_tq_result &o = res[jobi];
  const int qp_per = q.qp / 6, q_bits = Q_BITS + qp_per;
  int M1[16][16], M4[4][4];
  load_residual(job, M1, 16, 16);
  for (int j = 0; j < 16; j += 4) for (int i = 0; i < 16; i += 4) fwd4_tile(M1, j, i);
  for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) M4[j][i] = M1[j << 2][i << 2];
  {                                                         // hadamard4x4, transform.c:131
    int t[4][4];
    for (int i = 0; i < 4; i++) {
      const int t0 = M4[i][0] + M4[i][3], t1 = M4[i][1] + M4[i][2], t2 = M4[i][1] - M4[i][2], t3 = M4[i][0] - M4[i][3];
      t[i][0] = t0 + t1; t[i][1] = t3 + t2; t[i][2] = t0 - t1; t[i][3] = t3 - t2;
    }
    for (int i = 0; i < 4; i++) {
      const int t0 = t[0][i] + t[3][i], t1 = t[1][i] + t[2][i], t2 = t[1][i] - t[2][i], t3 = t[0][i] - t[3][i];
      M4[0][i] = (t0 + t1) >> 1; M4[1][i] = (t2 + t3) >> 1; M4[2][i] = (t0 - t1) >> 1; M4[3][i] = (t3 - t2) >> 1;
    }
  }
  int run = -1, scan_pos = 0, ac_coef = 0;
  for (int k = 0; k < 16; k++) {
    const int i = c_scan4[q.field_scan][k][0], j = c_scan4[q.field_scan][k][1];
    run++;
    int level = (iabs(M4[j][i]) * q.levelscale[0] + (q.leveloffset[0] << 1)) >> (q_bits + 1);        // block.c:643
    if (level != 0) {
      if (q.cavlc && q.img_qp < 10) level = min(level, CAVLC_LEVEL_LIMIT);
      level = sgnab(level, M4[j][i]);
      o.dc_levels[scan_pos] = level; o.dc_runs[scan_pos] = run; scan_pos++;
      run = -1;
      M4[j][i] = level;
    } else M4[j][i] = 0;
  }
  o.dc_levels[scan_pos] = 0;
  {                                                         // ihadamard4x4, transform.c:180
    int t[4][4];
    for (int i = 0; i < 4; i++) {
      const int p0 = M4[i][0] + M4[i][2], p1 = M4[i][0] - M4[i][2], p2 = M4[i][1] - M4[i][3], p3 = M4[i][1] + M4[i][3];
      t[i][0] = p0 + p3; t[i][1] = p1 + p2; t[i][2] = p1 - p2; t[i][3] = p0 - p3;
    }
    for (int i = 0; i < 4; i++) {
      const int p0 = t[0][i] + t[2][i], p1 = t[0][i] - t[2][i], p2 = t[1][i] - t[3][i], p3 = t[1][i] + t[3][i];
      M4[0][i] = p0 + p3; M4[1][i] = p1 + p2; M4[2][i] = p1 - p2; M4[3][i] = p0 - p3;
    }
  }
  for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) M1[j << 2][i << 2] = rsr((M4[j][i] * q.invlevelscale[0]) << qp_per, 6);   // :667
  for (int jj = 0; jj < 4; jj++) for (int ii = 0; ii < 4; ii++) {
    const int jpos = jj << 2, ipos = ii << 2;
    const int blk = (2 * (jj >> 1) + (ii >> 1)) * 4 + 2 * (jj & 1) + (ii & 1);
    run = -1; scan_pos = 0;
    for (int k = 1; k < 16; k++) {
      const int i = c_scan4[q.field_scan][k][0], j = c_scan4[q.field_scan][k][1];
      run++;
      int *c = &M1[jpos + j][ipos + i];
      const int scaled = iabs(*c) * q.levelscale[j * 4 + i];
      int level = (scaled + q.leveloffset[j * 4 + i]) >> q_bits;
      if (level != 0) {
        if (q.adaptive_rounding) o.fadjust[jpos + j][ipos + i] = rsr(q.adapt_rnd_weight * (scaled - (level << q_bits)), q_bits + 1);
        ac_coef = 15;
        level = sgnab(level, *c);
        o.levels[blk][scan_pos] = level; o.runs[blk][scan_pos] = run; scan_pos++;
        run = -1;
        *c = rsr((level * q.invlevelscale[j * 4 + i]) << qp_per, 4);
      } else {
        *c = 0;
        if (q.adaptive_rounding) o.fadjust[jpos + j][ipos + i] = 0;
      }
    }
    o.levels[blk][scan_pos] = 0;
    inv4_tile(M1, jpos, ipos);
  }
  for (int j = 0; j < 16; j++) for (int i = 0; i < 16; i++)
    o.recon[j][i] = (uint8_t)clip1(q.max_val, rsr(M1[j][i], DQ_BITS) + job.pred[j][i]);
  o.ret = ac_coef;
}

// dct_chroma, block.c:1051 (non-lossless), one component per job. One lane per job.
__global__ __launch_bounds__(64) void tq_chroma_kernel(const jmhip_tq_job *__restrict__ jobs, const jmhip_quant *__restrict__ quants,
                                                      jmhip_tq_result *__restrict__ res, int n, int yuv)
{
  const int jobi = blockIdx.x * 64 + threadIdx.x;
  if (jobi >= n) return;
  const jmhip_tq_job &job = jobs[jobi];
  const jmhip_quant &q = quants[job.quant];
  const jmhip_quant &qdc = quants[job.quant_dc];
  jmhip_tq_result &o = res[jobi];
  const int uv = job.uv;
  int cr_cbp = job.cr_cbp_in;
  const int qp_per = q.qp / 6, q_bits = Q_BITS + qp_per;
  const int mbw = (yuv == JMHIP_YUV444) ? 16 : 8, mbh = (yuv == JMHIP_YUV420) ? 8 : 16;
  const int nb8 = ((yuv == JMHIP_YUV420) ? 2 : (yuv == JMHIP_YUV422 ? 4 : 8)) >> 1;       // img->num_blk8x8_uv >> 1
  const int uv_scale = uv * nb8;
  long long cbp = 0;
  int m7[16][16];
  load_residual(job, m7, 16, 16);     // full tile: the 4:2:2 quirk below reads columns 8..15 (carried in src/pred as JM's img->m7 holds them)
  for (int n2 = 0; n2 < mbh; n2 += 4) for (int n1 = 0; n1 < mbw; n1 += 4) fwd4_tile(m7, n1, n2);   // block.c:1116-1122 (n1,n2 swapped as in JM)

  int run = -1, scan_pos = 0, DCcoded = 0;
  if (yuv == JMHIP_YUV420) {
    int m1[4], m5[4];
    m1[0] = m7[0][0] + m7[0][4] + m7[4][0] + m7[4][4];
    m1[1] = m7[0][0] - m7[0][4] + m7[4][0] - m7[4][4];
    m1[2] = m7[0][0] + m7[0][4] - m7[4][0] - m7[4][4];
    m1[3] = m7[0][0] - m7[0][4] - m7[4][0] + m7[4][4];
    for (int k = 0; k < 4; k++) {
      run++;
      int level = (iabs(m1[k]) * q.levelscale[0] + (q.leveloffset[0] << 1)) >> (q_bits + 1);
      if (level != 0) {
        if (q.cavlc && q.img_qp < 4) level = min(level, CAVLC_LEVEL_LIMIT);
        cbp |= 0xf0000LL << (uv << 2);
        cr_cbp = max(1, cr_cbp);
        DCcoded = 1;
        level = sgnab(level, m1[k]);
        o.dc_levels[scan_pos] = level; o.dc_runs[scan_pos] = run; scan_pos++;
        run = -1;
        m1[k] = level;
      } else m1[k] = 0;
    }
    o.dc_levels[scan_pos] = 0;
    m5[0] = m1[0] + m1[1] + m1[2] + m1[3]; m5[1] = m1[0] - m1[1] + m1[2] - m1[3];
    m5[2] = m1[0] + m1[1] - m1[2] - m1[3]; m5[3] = m1[0] - m1[1] - m1[2] + m1[3];
    m7[0][0] = ((m5[0] * q.invlevelscale[0]) << qp_per) >> 5; m7[0][4] = ((m5[1] * q.invlevelscale[0]) << qp_per) >> 5;
    m7[4][0] = ((m5[2] * q.invlevelscale[0]) << qp_per) >> 5; m7[4][4] = ((m5[3] * q.invlevelscale[0]) << qp_per) >> 5;
  } else if (yuv == JMHIP_YUV422) {
    const int qp_per_dc = qdc.qp / 6, q_bits_422 = Q_BITS + qp_per_dc;
    int m3[2][4], m4[2][4], m5[4], m6[4];
    for (int j = 0; j < 16; j += 4) for (int i = 0; i < 8; i += 4) m3[i >> 2][j >> 2] = m7[j][i];
    for (int j = 0; j < 4; j++) { m4[0][j] = m3[0][j] + m3[1][j]; m4[1][j] = m3[0][j] - m3[1][j]; }
    for (int i = 0; i < 2; i++) {
      m5[0] = m4[i][0] + m4[i][3]; m5[1] = m4[i][1] + m4[i][2]; m5[2] = m4[i][1] - m4[i][2]; m5[3] = m4[i][0] - m4[i][3];
      m4[i][0] = m5[0] + m5[1]; m4[i][2] = m5[0] - m5[1]; m4[i][1] = m5[3] + m5[2]; m4[i][3] = m5[3] - m5[2];
    }
    for (int k = 0; k < 8; k++) {
      const int i = c_scan422[k][0], j = c_scan422[k][1];
      run++;
      const int level = (iabs(m4[i][j]) * q.levelscale[0] + (qdc.leveloffset[0] * 2)) >> (q_bits_422 + 1);   // block.c:1263
      if (level != 0) {
        cbp |= (long long)(int)(0xff0000u << (uv << 3));     // block.c:1268 is int arithmetic: sign-extends for uv == 1 (JM quirk, kept)
        cr_cbp = max(1, cr_cbp);
        DCcoded = 1;
        o.dc_levels[scan_pos] = sgnab(level, m4[i][j]); o.dc_runs[scan_pos] = run; scan_pos++;
        run = -1;
      }
      m3[i][j] = sgnab(level, m4[i][j]);
    }
    o.dc_levels[scan_pos] = 0;
    for (int j = 0; j < 4; j++) { m4[0][j] = m3[0][j] + m3[1][j]; m4[1][j] = m3[0][j] - m3[1][j]; }
    const int inv = qdc.invlevelscale[0];
    for (int i = 0; i < 2; i++) {
      m6[0] = m4[i][0] + m4[i][2]; m6[1] = m4[i][0] - m4[i][2]; m6[2] = m4[i][1] - m4[i][3]; m6[3] = m4[i][1] + m4[i][3];
      const int v[4] = {m6[0] + m6[3], m6[1] + m6[2], m6[1] - m6[2], m6[0] - m6[3]};
      for (int r = 0; r < 4; r++)
        m7[4 * r][i * 4] = (qp_per_dc < 4) ? ((((v[r] * inv + (1 << (3 - qp_per_dc))) >> (4 - qp_per_dc)) + 2) >> 2)
                                           : ((((v[r] * inv) << (qp_per_dc - 4)) + 2) >> 2);                   // block.c:1303-1316
    }
  }

  int coeff_cost = 0, cr_cbp_tmp = 0;
  for (int b8 = 0; b8 < nb8; b8++) for (int b4 = 0; b4 < 4; b4++) {
    const long long uv_cbpblk = 1LL << (16 + 4 * (b8 + uv_scale) + b4);                              // cbp_blk_chroma, block.h:109
    const int n1 = c_hor[yuv][b8][b4], n2 = c_ver[yuv][b8][b4], blk = b8 * 4 + b4;
    run = -1; scan_pos = 0;
    for (int k = 1; k < 16; k++) {
      const int i = c_scan4[q.field_scan][k][0], j = c_scan4[q.field_scan][k][1];
      int *c = &m7[n2 + j][n1 + i];
      ++run;
      const int scaled = iabs(*c) * q.levelscale[j * 4 + i];
      int level = (scaled + q.leveloffset[j * 4 + i]) >> q_bits;
      if (level != 0) {
        if (q.adaptive_rounding) o.fadjust[n2 + j][n1 + i] = rsr(q.adapt_rnd_weight * (scaled - (level << q_bits)), q_bits + 1);
        cbp |= uv_cbpblk;
        coeff_cost += (level > 1) ? MAXV : c_cost4[q.disthres][run];
        cr_cbp_tmp = 2;
        level = sgnab(level, *c);
        o.levels[blk][scan_pos] = level; o.runs[blk][scan_pos] = run; scan_pos++;
        run = -1;
        *c = rsr((level * q.invlevelscale[j * 4 + i]) << qp_per, 4);
      } else {
        *c = 0;
        if (q.adaptive_rounding) o.fadjust[n2 + j][n1 + i] = 0;
      }
    }
    o.levels[blk][scan_pos] = 0;
  }
  long long cbp_clear = 0;
  if (coeff_cost < 4) {                                      // _CHROMA_COEFF_COST_, defines.h:103; block.c:1384-1410
    const long long pattern = (yuv == 1) ? 0xf0000LL : (yuv == 2 ? 0xff0000LL : 0xffff0000LL);
    cr_cbp_tmp = 0;
    if (DCcoded == 0) cbp_clear = pattern << (uv << (1 + yuv));
    for (int b8 = 0; b8 < nb8; b8++) for (int b4 = 0; b4 < 4; b4++) {
      const int n1 = c_hor[yuv][b8][b4], n2 = c_ver[yuv][b8][b4], blk = b8 * 4 + b4;
      o.levels[blk][0] = 0;
      for (int k = 1; k < 16; k++) {
        m7[n2 + c_scan4[q.field_scan][k][1]][n1 + c_scan4[q.field_scan][k][0]] = 0;
        o.levels[blk][k] = 0;
      }
    }
  }
  if (cr_cbp_tmp == 2) cr_cbp = 2;
  for (int n2 = 0; n2 < mbh; n2 += 4) for (int n1 = 0; n1 < mbw; n1 += 4) inv4_tile(m7, n2, n1);
  for (int j = 0; j < mbh; j++) for (int i = 0; i < mbw; i++)
    o.recon[j][i] = (uint8_t)clip1(q.max_val, rsr(m7[j][i], DQ_BITS) + job.pred[j][i]);
  o.ret = cr_cbp;
  // cbp bits set by this call, and (in the high half of the pair) the bits the thresholding clears
  o.cbp_blk = cbp & ~cbp_clear;
  o.cbp_clear = cbp_clear;
}


// dct_chroma for 4:2:0, the common case: FOUR lanes per job (one per 4x4 block, a DPP quad), everything in registers.
// The 2x2 DC transform gathers the four DC terms with quad_perm broadcasts; lane 0 of the quad quantises the DC list
// (sequential run/level semantics) and the dequantised DCs go back the same way. block.c:1051-1495.
__device__ __forceinline__ int quad_bcast(int v, int src)      // value of lane (quad base + src) in every lane of the quad
{
  switch (src) {
  case 0: return __builtin_amdgcn_update_dpp(v, v, 0x00, 0xf, 0xf, false);
  case 1: return __builtin_amdgcn_update_dpp(v, v, 0x55, 0xf, 0xf, false);
  case 2: return __builtin_amdgcn_update_dpp(v, v, 0xAA, 0xf, 0xf, false);
  default: return __builtin_amdgcn_update_dpp(v, v, 0xFF, 0xf, 0xf, false);
  }
}

__global__ __launch_bounds__(256) void tq_chroma420_kernel(const jmhip_tq_job *__restrict__ jobs, const jmhip_quant *__restrict__ quants,
                                                          jmhip_tq_result *__restrict__ res, int n)
{
  const int vb = jm_xcd_item((n * 4 + 255) / 256);             // XCD-contiguous block order, see tq_luma4x4_kernel
  if (vb < 0) return;
  const int gid = vb * 256 + threadIdx.x;
  const int jobi = min(gid >> 2, n - 1), b4 = gid & 3;          // b4: 0 (0,0)  1 (4,0)  2 (0,4)  3 (4,4)  (hor/ver_offset[1][0])
  const bool live = (gid >> 2) < n;                             // whole quads are live or dead together
  const jmhip_tq_job &job = jobs[jobi];
  const jmhip_quant &q = quants[job.quant];
  jmhip_tq_result &o = res[jobi];
  const int uv = job.uv;
  const int bx = 4 * (b4 & 1), by = 4 * (b4 >> 1);
  const int qp_per = q.qp / 6, q_bits = Q_BITS + qp_per;

  int m[4][4], pr[4][4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t s = *reinterpret_cast<const uint32_t *>(&job.src[by + j][bx]);
    const uint32_t p = *reinterpret_cast<const uint32_t *>(&job.pred[by + j][bx]);
#pragma unroll
    for (int i = 0; i < 4; i++) { pr[j][i] = (p >> (8 * i)) & 255; m[j][i] = (int)((s >> (8 * i)) & 255) - pr[j][i]; }
  }
  fwd4(m);       // block.c:1120 swaps (n1,n2); for the symmetric 8x8 tile of 4:2:0 the same four blocks are transformed

  // ---- 2x2 DC: m1[k] over the DCs d0 (0,0) d1 (0,4)->cols d2 (4,0)->rows d3 (4,4): curr_res[0][0],[0][4],[4][0],[4][4]
  const int d0 = quad_bcast(m[0][0], 0), d1 = quad_bcast(m[0][0], 1), d2 = quad_bcast(m[0][0], 2), d3 = quad_bcast(m[0][0], 3);
  int m1[4] = {d0 + d1 + d2 + d3, d0 - d1 + d2 - d3, d0 + d1 - d2 - d3, d0 - d1 - d2 + d3};
  int run = -1, scan_pos = 0, DCcoded = 0, cr_cbp = job.cr_cbp_in;
  long long cbp = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {          // every lane of the quad computes the (cheap) DC list; lane 0 writes it
    run++;
    int level = (iabs(m1[k]) * q.levelscale[0] + (q.leveloffset[0] << 1)) >> (q_bits + 1);
    if (level != 0) {
      if (q.cavlc && q.img_qp < 4) level = min(level, CAVLC_LEVEL_LIMIT);
      cbp |= 0xf0000LL << (uv << 2);
      cr_cbp = max(1, cr_cbp);
      DCcoded = 1;
      level = sgnab(level, m1[k]);
      if (live && b4 == 0) { o.dc_levels[scan_pos] = level; o.dc_runs[scan_pos] = run; }
      scan_pos++;
      run = -1;
      m1[k] = level;
    } else m1[k] = 0;
  }
  if (live && b4 == 0) o.dc_levels[scan_pos] = 0;
  {
    const int m5[4] = {m1[0] + m1[1] + m1[2] + m1[3], m1[0] - m1[1] + m1[2] - m1[3], m1[0] + m1[1] - m1[2] - m1[3], m1[0] - m1[1] - m1[2] + m1[3]};
    m[0][0] = ((m5[b4] * q.invlevelscale[0]) << qp_per) >> 5;           // block.c:1170-1173
  }

  // ---- AC of this lane's block (scan positions 1..15)
  int coeff_cost = 0, any = 0;
  scan_pos = 0; run = -1;
  int fa[4][4];
  fa[0][0] = 0;                                      // the DC position is not dct_chroma's to write (block.c:1321: AC only); the buffer's own zero stays
  int *levels = o.levels[b4], *runs = o.runs[b4];
#pragma unroll
  for (int k = 1; k < 16; k++) {
    constexpr int I0[16] = {0,1,0,0,1,2,3,2,1,0,1,2,3,3,2,3}, J0[16] = {0,0,1,2,1,0,0,1,2,3,3,2,1,2,3,3};
    constexpr int I1[16] = {0,0,1,0,0,1,1,1,2,2,2,2,3,3,3,3}, J1[16] = {0,1,0,2,3,1,2,3,0,1,2,3,0,1,2,3};
    const int i0 = I0[k], j0 = J0[k], i1 = I1[k], j1 = J1[k];
    const int c = q.field_scan ? m[j1][i1] : m[j0][i0];
    const int idx = q.field_scan ? (j1 * 4 + i1) : (j0 * 4 + i0);
    ++run;
    const int scaled = iabs(c) * q.levelscale[idx];
    int level = (scaled + q.leveloffset[idx]) >> q_bits;
    int deq = 0, fadj = 0;
    if (level != 0) {
      if (q.adaptive_rounding) fadj = rsr(q.adapt_rnd_weight * (scaled - (level << q_bits)), q_bits + 1);
      any = 1;
      coeff_cost += (level > 1) ? MAXV : c_cost4[q.disthres][run];
      level = sgnab(level, c);
      if (live) { levels[scan_pos] = level; runs[scan_pos] = run; }
      scan_pos++;
      run = -1;
      deq = rsr((level * q.invlevelscale[idx]) << qp_per, 4);
    }
    if (q.field_scan) { m[j1][i1] = deq; fa[j1][i1] = fadj; } else { m[j0][i0] = deq; fa[j0][i0] = fadj; }
  }
  if (live) levels[scan_pos] = 0;
  if (live && q.adaptive_rounding) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      int2 *d = reinterpret_cast<int2 *>(&o.fadjust[by + j][bx]);
      d[0] = make_int2(fa[j][0], fa[j][1]); d[1] = make_int2(fa[j][2], fa[j][3]);
    }
  }
  if (any) cbp |= 1LL << (16 + 4 * uv + b4);                           // cbp_blk_chroma[uv][b4], block.h:109 (4:2:0: one 8x8 per component)

  // ---- thresholding over the four blocks of the component (_CHROMA_COEFF_COST_ = 4), block.c:1384-1410
  int total = coeff_cost;
  total += __builtin_amdgcn_update_dpp(total, total, 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
  total += __builtin_amdgcn_update_dpp(total, total, 0x4E, 0xf, 0xf, false);      // quad_perm [2,3,0,1]
  int anyq = any;
  anyq |= __builtin_amdgcn_update_dpp(anyq, anyq, 0xB1, 0xf, 0xf, false);
  anyq |= __builtin_amdgcn_update_dpp(anyq, anyq, 0x4E, 0xf, 0xf, false);
  long long cbp_clear = 0;
  int cr_cbp_tmp = anyq ? 2 : 0;
  if (total < 4) {
    cr_cbp_tmp = 0;
    if (DCcoded == 0) cbp_clear = 0xf0000LL << (uv << 2);             // cbpblk_pattern[1] << (uv << 2)
    if (live) {
#pragma unroll
      for (int k = 0; k < 16; k++) levels[k] = 0;
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int i = 0; i < 4; i++) if (i | j) m[j][i] = 0;
  }
  if (cr_cbp_tmp == 2) cr_cbp = 2;
  // OR the per-block cbp bits across the quad
  unsigned cl = (unsigned)cbp, ch = (unsigned)(cbp >> 32);
  cl |= (unsigned)__builtin_amdgcn_update_dpp((int)cl, (int)cl, 0xB1, 0xf, 0xf, false);
  cl |= (unsigned)__builtin_amdgcn_update_dpp((int)cl, (int)cl, 0x4E, 0xf, 0xf, false);
  cbp = ((long long)ch << 32) | cl;

  inv4(m);
  if (live) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      uint32_t w = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) w |= (uint32_t)clip1(q.max_val, rsr(m[j][i], DQ_BITS) + pr[j][i]) << (8 * i);
      *reinterpret_cast<uint32_t *>(&o.recon[by + j][bx]) = w;
    }
    if (b4 == 0) { o.ret = cr_cbp; o.cbp_blk = cbp & ~cbp_clear; o.cbp_clear = cbp_clear; }
  }
}


// ---------------------------------------------------------------------------------------- fused 4:2:0 frame stage
//
// mc_kernel + tq_luma4x4_kernel + tq_chroma420_kernel + finalize_kernel of the frame stage in ONE launch for the common case (4:2:0, 4x4
// transform): the prediction and source tiles never leave the CU (registers for luma, 256 B of LDS for chroma) instead of travelling through
// 1.6 KB of jmhip_tq_job per macroblock, and what the consumers need of the results is one dense 2.4 KB record (JmMbRes) instead of three
// sparse 5.8 KB structs. Four macroblocks per wave: in the luma phase every lane owns one 4x4 block (16 lanes per macroblock, JM order
// b8*4+b4, so a DPP quad is an 8x8 block), in the chroma phase lanes 0..31 own the four Cb / four Cr blocks of each macroblock (a quad per
// component, as tq_chroma420_kernel); all 64 lanes fetch the chroma prediction.
// Same arithmetic, line for line, as the kernels it replaces (they stay for 4:2:2, 4:0:0, 8x8-transform macroblocks and jmhip_tq_batch).
__global__ __launch_bounds__(128) void frame_fused_kernel(FrameDev F, const jmhip_me_mb *__restrict__ mbs, const jmhip_me_result *__restrict__ me,
                                                        const jmhip_mb_mode *__restrict__ modes_in, jmhip_mb_mode *__restrict__ modes_out,
                                                        const jmhip_quant *__restrict__ quants, JmMbRes *__restrict__ out, JmMbCoded *__restrict__ coded, int n)
{
  constexpr int NMB = 4;                               // macroblocks per wave: 16 luma lanes each, then 8 chroma lanes each
  __shared__ __attribute__((aligned(16))) JmMbRes s_rec[NMB];
  __shared__ jmhip_mb_mode s_mode[NMB];
  __shared__ short s_mv[NMB][16][2];
  __shared__ int s_ref[NMB][4];
  __shared__ uint32_t s_mvall[NMB][JMHIP_NPART];
  __shared__ int s_cost[NMB][JMHIP_NPART];
  __shared__ short s_pos[NMB][2];
  __shared__ __attribute__((aligned(4))) uint8_t s_pc[NMB][2][8][8], s_sc[NMB][2][8][8];      // chroma prediction / source tiles
  const int vb = jm_xcd_item((n + NMB - 1) / NMB);
  if (vb < 0) return;
  // two waves: wave 0 stages and then owns the luma blocks, wave 1 owns chroma (prediction, then dct_chroma) -- the two transform paths run
  // side by side instead of one after the other (the kernel is bound by one wave's latency)
  const int wv = threadIdx.x >> 6, tid = threadIdx.x & 63, i0 = vb * NMB;
  const int nlive = min(NMB, n - i0);                  // the last wave repeats macroblock n-1 in its spare groups and stores nothing for them
  auto mb_of = [&](int hh) { return min(i0 + hh, n - 1); };

  // the 41 vectors and costs of each macroblock arrive in one round trip (the mode picks among them afterwards, out of LDS)
  if (wv == 0)
  for (int e = tid; e < NMB * JMHIP_NPART; e += 64) {
    const int hh = e / JMHIP_NPART, p = e - hh * JMHIP_NPART;
    const jmhip_me_result &r = me[mb_of(hh)];
    s_mvall[hh][p] = *reinterpret_cast<const uint32_t *>(r.mv[p]);
    s_cost[hh][p] = r.cost[p];
  }
  jmhip_mb_mode m_in;
  if (wv == 0 && tid < NMB && modes_in) m_in = modes_in[mb_of(tid)];
  if (wv == 0 && tid >= 8 && tid < 8 + NMB) { const jmhip_me_mb &mb = mbs[mb_of(tid - 8)]; s_pos[tid - 8][0] = mb.mb_x; s_pos[tid - 8][1] = mb.mb_y; }
  if (wv == 0 && tid >= 16 && tid < 16 + 4 * NMB) { const int hh = (tid - 16) >> 2, k = tid & 3; s_ref[hh][k] = F.blk_ref ? F.blk_ref[(size_t)mb_of(hh) * 4 + k] : mbs[mb_of(hh)].ref; }
  __syncthreads();
  if (wv == 0 && tid < NMB) {
    jmhip_mb_mode m;
    if (modes_in) m = m_in;
    else {                                             // smallest summed motion cost; ties go to the lower mode number (as mc_kernel)
      const int *c = s_cost[tid];
      int c8 = 0;
      for (int b = 0; b < 4; b++) {
        const int s4 = c[5 + b], s5 = c[9 + 2 * b] + c[10 + 2 * b], s6 = c[17 + 2 * b] + c[18 + 2 * b];
        const int s7 = c[25 + 4 * b] + c[26 + 4 * b] + c[27 + 4 * b] + c[28 + 4 * b];
        int best = s4, bm = 4;
        if (s5 < best) { best = s5; bm = 5; }
        if (s6 < best) { best = s6; bm = 6; }
        if (s7 < best) { best = s7; bm = 7; }
        m.b8mode[b] = (int8_t)bm; c8 += best;
      }
      int best = c[0]; m.mode = 1;
      if (c[1] + c[2] < best) { best = c[1] + c[2]; m.mode = 2; }
      if (c[3] + c[4] < best) { best = c[3] + c[4]; m.mode = 3; }
      if (c8 < best) { best = c8; m.mode = 8; }
      m.pad[0] = m.pad[1] = m.pad[2] = 0;
    }
    s_mode[tid] = m;
    if (tid < nlive) modes_out[i0 + tid] = m;
  }
  __syncthreads();
  if (wv == 0) {
    const int hh = tid >> 4, l = tid & 15;
    const uint32_t v = s_mvall[hh][covering_partition(s_mode[hh], l & 3, l >> 2)];
    s_mv[hh][l][0] = (short)(v & 0xffff); s_mv[hh][l][1] = (short)(v >> 16);
  }
  __syncthreads();

  // ---- chroma prediction and source into LDS: four sample pairs per lane (mc_kernel's chroma loop, macroblock.c:1626-1650)
  if (wv == 1)
#pragma unroll
  for (int t = tid; t < NMB * 64; t += 64) {
    const int h = t >> 6, tt = t & 63;
    const int mbx = s_pos[h][0], mby = s_pos[h][1];
    const int uv = tt >> 5, q = tt & 31, j = q >> 2, ic = 2 * (q & 3);
    const int by4 = j >> 1, bx4 = ic >> 1;
    const short *mv = s_mv[h][by4 * 4 + bx4];
    const int bii = ((ic + mbx * 8) << 3) + 4 * JMHIP_PAD, bjj = ((j + mby * 8) << 3) + 4 * JMHIP_PAD;
    const int b8 = 2 * (by4 >> 1) + (bx4 >> 1), slot = s_ref[h][b8];
    int pdir = 0, slot1 = 0;
    if (F.bi) { const jmhip_mb_bipred &bm = F.bi[mb_of(h)]; pdir = bm.pdir[b8]; slot1 = bm.ref1[b8]; }
    int p0 = 0, p1 = 0, q0 = 0, q1 = 0;
    if (pdir != 1) chroma_pair(F, slot, uv, bii + mv[0], bjj + mv[1], &p0, &p1);
    if (pdir != 0) { const short *m1 = F.bi[mb_of(h)].mv1[by4 * 4 + bx4]; chroma_pair(F, slot1, uv, bii + m1[0], bjj + m1[1], &q0, &q1); }
    if (F.wp_on || pdir) { p0 = mix_pred(F, pdir, slot, slot1, uv + 1, p0, q0); p1 = mix_pred(F, pdir, slot, slot1, uv + 1, p1, q1); }
    *reinterpret_cast<uint16_t *>(&s_pc[h][uv][j][ic]) = (uint16_t)(p0 | (p1 << 8));
    const uint8_t *cs = (uv ? F.cur_v : F.cur_u) + (size_t)(mby * 8 + j) * F.Wc + mbx * 8 + ic;
    *reinterpret_cast<uint16_t *>(&s_sc[h][uv][j][ic]) = *reinterpret_cast<const uint16_t *>(cs);
  }

  int cbp_luma = 0, cbp_blk_luma = 0;                  // after the coefficient-cost thresholds (every luma lane ends up holding them)
  if (wv == 0) {
    // ---- luma block: prediction (LumaPrediction per 4x4 block, macroblock.c:836), residual, dct_4x4 (block.c:843)
    const int h = tid >> 4, l = tid & 15;
    const int mbx = s_pos[h][0], mby = s_pos[h][1];
    const bool live = h < nlive;
    JmMbRes &R = s_rec[h];
    const int blk = l, b8 = blk >> 2, b4 = blk & 3;
    const int x4 = 2 * (b8 & 1) + (b4 & 1), y4 = 2 * (b8 >> 1) + (b4 >> 1), bx = 4 * x4, by = 4 * y4;
    const jmhip_quant &q = quants[0];
    const int qp_per = q.qp / 6, q_bits = Q_BITS + qp_per;
    int m[4][4], pr[4][4];
    {
      const short *mv = s_mv[h][y4 * 4 + x4];
      const int bqx = ((mbx * 16 + bx) << 2) + 4 * JMHIP_PAD, bqy = ((mby * 16 + by) << 2) + 4 * JMHIP_PAD;
      const int xq = bqx + mv[0], yq = bqy + mv[1], slot = s_ref[h][b8];
      int pdir = 0, slot1 = 0, xq1 = 0, yq1 = 0;                                             // the second list of a B macroblock
      if (F.bi) { const jmhip_mb_bipred &bm = F.bi[mb_of(h)]; pdir = bm.pdir[b8]; slot1 = bm.ref1[b8]; xq1 = bqx + bm.mv1[y4 * 4 + x4][0]; yq1 = bqy + bm.mv1[y4 * 4 + x4][1]; }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t v0 = pdir != 1 ? luma_row4(F, slot, xq, yq, 0, 0, j) : 0u;
        const uint32_t v1 = pdir != 0 ? luma_row4(F, slot1, xq1, yq1, 0, 0, j) : 0u;
        const uint32_t sv = *reinterpret_cast<const uint32_t *>(F.cur_y + (size_t)(mby * 16 + by + j) * F.W + mbx * 16 + bx);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          int p = (int)((v0 >> (8 * k)) & 255u);
          if (F.wp_on || pdir) p = mix_pred(F, pdir, slot, slot1, 0, p, (int)((v1 >> (8 * k)) & 255u));
          pr[j][k] = p; m[j][k] = (int)((sv >> (8 * k)) & 255u) - p;                                     // img->m7, macroblock.c:1059-1068
        }
      }
    }
    fwd4(m);
    int scan_pos = 0, run = -1, nonzero = 0, cost = 0;
    int fa[4][4];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      constexpr int I0[16] = {0,1,0,0,1,2,3,2,1,0,1,2,3,3,2,3}, J0[16] = {0,0,1,2,1,0,0,1,2,3,3,2,1,2,3,3};
      constexpr int I1[16] = {0,0,1,0,0,1,1,1,2,2,2,2,3,3,3,3}, J1[16] = {0,1,0,2,3,1,2,3,0,1,2,3,0,1,2,3};
      const int i0 = I0[k], j0 = J0[k], i1 = I1[k], j1 = J1[k];
      const int c = q.field_scan ? m[j1][i1] : m[j0][i0];
      const int idx = q.field_scan ? (j1 * 4 + i1) : (j0 * 4 + i0);
      run++;
      const int scaled = iabs(c) * q.levelscale[idx];
      int level = (scaled + q.leveloffset[idx]) >> q_bits;
      int deq = 0, fadj = 0;
      if (level != 0) {
        if (q.adaptive_rounding) fadj = rsr(q.adapt_rnd_weight * (scaled - (level << q_bits)), q_bits + 1);   // block.c:898
        nonzero = 1;
        cost += (level > 1) ? MAXV : c_cost4[q.disthres][run];
        level = sgnab(level, c);
        R.lev[blk][scan_pos] = (int16_t)level; R.run[blk][scan_pos] = (uint8_t)run; scan_pos++;
        deq = rsr((level * q.invlevelscale[idx]) << qp_per, 4);                                             // block.c:907
        run = -1;
      }
      if (q.field_scan) { m[j1][i1] = deq; fa[j1][i1] = fadj; } else { m[j0][i0] = deq; fa[j0][i0] = fadj; }
    }
    R.cnt[blk] = (uint8_t)scan_pos;
    R.coeff_cost[blk] = cost;
    if (q.adaptive_rounding) {
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int k = 0; k < 4; k++) R.fadj_y[by + j][bx + k] = (int16_t)fa[j][k];
    }
    if (scan_pos) inv4(m);
    uint32_t rec[4], prd[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      uint32_t w = 0, pw = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int v = scan_pos ? clip1(q.max_val, rsr(m[j][k], DQ_BITS) + pr[j][k]) : pr[j][k];                // block.c:934 / :942
        w |= (uint32_t)v << (8 * k); pw |= (uint32_t)pr[j][k] << (8 * k);
      }
      rec[j] = w; prd[j] = pw;
      *reinterpret_cast<uint32_t *>(&R.recon_y[by + j][bx]) = w;
    }
    // ---- _LUMA_COEFF_COST_ per 8x8 block (a quad) and _LUMA_MB_COEFF_COST_ per macroblock: macroblock.c:1236-1258, :1386-1392 (finalize_kernel)
    int cost8 = cost;
    cost8 += __builtin_amdgcn_update_dpp(cost8, cost8, 0xB1, 0xf, 0xf, false);
    cost8 += __builtin_amdgcn_update_dpp(cost8, cost8, 0x4E, 0xf, 0xf, false);
    const bool keep8 = cost8 > 4;
    int sum = keep8 ? cost8 : 0;
    sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8);
    const bool keep = keep8 && sum > 5;
    int bits = (nonzero && keep) ? ((1 << (x4 + 4 * y4)) | (0x10000 << b8)) : 0;      // cbp_blk bit (block_x>>2) + block_y, macroblock.c:1050; cbp bit b8 above it
    bits |= __shfl_xor(bits, 1); bits |= __shfl_xor(bits, 2); bits |= __shfl_xor(bits, 4); bits |= __shfl_xor(bits, 8);
    cbp_blk_luma = bits & 0xffff; cbp_luma = bits >> 16;
    const unsigned long long nz = __ballot(nonzero != 0);
    if (l == 0) R.nonzero = (uint16_t)(nz >> (16 * h));
    if (live) {
#pragma unroll
      for (int j = 0; j < 4; j++)
        *reinterpret_cast<uint32_t *>(F.rec_y + (size_t)(mby * 16 + by + j) * F.W + mbx * 16 + bx) = keep ? rec[j] : prd[j];
      if (F.pred_y) {                                  // img->mpr, kept for a host that answers JM's LumaPrediction from it (jmhip_frame_keep_prediction)
#pragma unroll
        for (int j = 0; j < 4; j++) *reinterpret_cast<uint32_t *>(F.pred_y + (size_t)(mby * 16 + by + j) * F.W + mbx * 16 + bx) = prd[j];
      }
    }
  }
  if (wv == 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }   // the chroma tiles are this wave's own

  if (wv == 1 && tid < 8 * NMB) {
    // ---- chroma block: dct_chroma for 4:2:0 on a quad of lanes (tq_chroma420_kernel), block.c:1051-1495
    const int h = tid >> 3, uv = (tid >> 2) & 1, b4 = tid & 3, cb = 16 + 4 * uv + b4;
    const int mbx = s_pos[h][0], mby = s_pos[h][1];
    const bool live = h < nlive;
    JmMbRes &R = s_rec[h];
    const int bx = 4 * (b4 & 1), by = 4 * (b4 >> 1);
    const jmhip_quant &q = quants[1];
    const int qp_per = q.qp / 6, q_bits = Q_BITS + qp_per;
    int m[4][4], pr[4][4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t s = *reinterpret_cast<const uint32_t *>(&s_sc[h][uv][by + j][bx]);
      const uint32_t p = *reinterpret_cast<const uint32_t *>(&s_pc[h][uv][by + j][bx]);
#pragma unroll
      for (int k = 0; k < 4; k++) { pr[j][k] = (p >> (8 * k)) & 255; m[j][k] = (int)((s >> (8 * k)) & 255) - pr[j][k]; }
    }
    fwd4(m);
    const int d0 = quad_bcast(m[0][0], 0), d1 = quad_bcast(m[0][0], 1), d2 = quad_bcast(m[0][0], 2), d3 = quad_bcast(m[0][0], 3);
    int m1[4] = {d0 + d1 + d2 + d3, d0 - d1 + d2 - d3, d0 + d1 - d2 - d3, d0 - d1 - d2 + d3};
    int run = -1, scan_pos = 0, DCcoded = 0, cr_cbp = 0;
    long long cbp = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      run++;
      int level = (iabs(m1[k]) * q.levelscale[0] + (q.leveloffset[0] << 1)) >> (q_bits + 1);
      if (level != 0) {
        if (q.cavlc && q.img_qp < 4) level = min(level, CAVLC_LEVEL_LIMIT);
        cbp |= 0xf0000LL << (uv << 2);
        cr_cbp = max(1, cr_cbp);
        DCcoded = 1;
        level = sgnab(level, m1[k]);
        if (b4 == 0) { R.dc_lev[uv][scan_pos] = (int16_t)level; R.dc_run[uv][scan_pos] = (uint8_t)run; }
        scan_pos++;
        run = -1;
        m1[k] = level;
      } else m1[k] = 0;
    }
    if (b4 == 0) R.dc_cnt[uv] = (uint8_t)scan_pos;
    {
      const int m5[4] = {m1[0] + m1[1] + m1[2] + m1[3], m1[0] - m1[1] + m1[2] - m1[3], m1[0] + m1[1] - m1[2] - m1[3], m1[0] - m1[1] - m1[2] + m1[3]};
      m[0][0] = ((m5[b4] * q.invlevelscale[0]) << qp_per) >> 5;           // block.c:1170-1173
    }
    int coeff_cost = 0, any = 0;
    scan_pos = 0; run = -1;
    int fa[4][4];
    fa[0][0] = 0;
#pragma unroll
    for (int k = 1; k < 16; k++) {
      constexpr int I0[16] = {0,1,0,0,1,2,3,2,1,0,1,2,3,3,2,3}, J0[16] = {0,0,1,2,1,0,0,1,2,3,3,2,1,2,3,3};
      constexpr int I1[16] = {0,0,1,0,0,1,1,1,2,2,2,2,3,3,3,3}, J1[16] = {0,1,0,2,3,1,2,3,0,1,2,3,0,1,2,3};
      const int i0 = I0[k], j0 = J0[k], i1 = I1[k], j1 = J1[k];
      const int c = q.field_scan ? m[j1][i1] : m[j0][i0];
      const int idx = q.field_scan ? (j1 * 4 + i1) : (j0 * 4 + i0);
      ++run;
      const int scaled = iabs(c) * q.levelscale[idx];
      int level = (scaled + q.leveloffset[idx]) >> q_bits;
      int deq = 0, fadj = 0;
      if (level != 0) {
        if (q.adaptive_rounding) fadj = rsr(q.adapt_rnd_weight * (scaled - (level << q_bits)), q_bits + 1);
        any = 1;
        coeff_cost += (level > 1) ? MAXV : c_cost4[q.disthres][run];
        level = sgnab(level, c);
        R.lev[cb][scan_pos] = (int16_t)level; R.run[cb][scan_pos] = (uint8_t)run;
        scan_pos++;
        run = -1;
        deq = rsr((level * q.invlevelscale[idx]) << qp_per, 4);
      }
      if (q.field_scan) { m[j1][i1] = deq; fa[j1][i1] = fadj; } else { m[j0][i0] = deq; fa[j0][i0] = fadj; }
    }
    R.cnt[cb] = (uint8_t)scan_pos;
    if (q.adaptive_rounding) {
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int k = 0; k < 4; k++) R.fadj_c[uv][by + j][bx + k] = (int16_t)fa[j][k];
    }
    if (any) cbp |= 1LL << (16 + 4 * uv + b4);
    int total = coeff_cost;
    total += __builtin_amdgcn_update_dpp(total, total, 0xB1, 0xf, 0xf, false);
    total += __builtin_amdgcn_update_dpp(total, total, 0x4E, 0xf, 0xf, false);
    int anyq = any;
    anyq |= __builtin_amdgcn_update_dpp(anyq, anyq, 0xB1, 0xf, 0xf, false);
    anyq |= __builtin_amdgcn_update_dpp(anyq, anyq, 0x4E, 0xf, 0xf, false);
    long long cbp_clear = 0;
    int cr_cbp_tmp = anyq ? 2 : 0;
    if (total < 4) {                                                       // _CHROMA_COEFF_COST_, block.c:1384-1410
      cr_cbp_tmp = 0;
      if (DCcoded == 0) cbp_clear = 0xf0000LL << (uv << 2);
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int k = 0; k < 4; k++) if (k | j) m[j][k] = 0;
    }
    if (b4 == 0) R.ac_zeroed[uv] = (uint8_t)(total < 4);
    if (cr_cbp_tmp == 2) cr_cbp = 2;
    unsigned cl = (unsigned)cbp, ch = (unsigned)(cbp >> 32);
    cl |= (unsigned)__builtin_amdgcn_update_dpp((int)cl, (int)cl, 0xB1, 0xf, 0xf, false);
    cl |= (unsigned)__builtin_amdgcn_update_dpp((int)cl, (int)cl, 0x4E, 0xf, 0xf, false);
    cbp = ((long long)ch << 32) | cl;
    inv4(m);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      uint32_t w = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) w |= (uint32_t)clip1(q.max_val, rsr(m[j][k], DQ_BITS) + pr[j][k]) << (8 * k);
      *reinterpret_cast<uint32_t *>(&R.recon_c[uv][by + j][bx]) = w;
      if (live) *reinterpret_cast<uint32_t *>((uv ? F.rec_v : F.rec_u) + (size_t)(mby * 8 + by + j) * F.Wc + mbx * 8 + bx) = w;
      if (live && F.pred_u) {
        uint32_t pw = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) pw |= (uint32_t)pr[j][k] << (8 * k);
        *reinterpret_cast<uint32_t *>((uv ? F.pred_v : F.pred_u) + (size_t)(mby * 8 + by + j) * F.Wc + mbx * 8 + bx) = pw;
      }
    }
    if (b4 == 0) { R.ret[uv] = cr_cbp; R.cbp_blk[uv] = cbp & ~cbp_clear; R.cbp_clear[uv] = cbp_clear; }
  }
  __syncthreads();

  if (wv == 0 && (tid & 15) == 0 && (tid >> 4) < nlive) {         // macroblock.c:2028-2040
    const JmMbRes &R = s_rec[tid >> 4];
    const int i = i0 + (tid >> 4);
    long long cb = cbp_blk_luma;
    cb = (cb & ~R.cbp_clear[0]) | R.cbp_blk[0];
    cb = (cb & ~R.cbp_clear[1]) | R.cbp_blk[1];
    coded[i].cbp = cbp_luma + (max(R.ret[0], R.ret[1]) << 4); coded[i].pad = 0; coded[i].cbp_blk = cb;
  }
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(s_rec);
    uint4 *dst = reinterpret_cast<uint4 *>(out + i0);
    for (int k = threadIdx.x; k < nlive * (int)(sizeof(JmMbRes) / 16); k += 128) dst[k] = src[k];
  }
}

}  // namespace

int jm_launch_frame_fused(jmhip_ctx *c, const void *frame_dev, const void *mbs, const void *me, const void *modes_in, void *modes_out,
                          const void *quants, void *records, void *coded, int n)
{
  const FrameDev &F = *static_cast<const FrameDev *>(frame_dev);
  frame_fused_kernel<<<jm_xcd_grid((n + 3) / 4), 128, 0, c->stream>>>(F, (const jmhip_me_mb *)mbs, (const jmhip_me_result *)me, (const jmhip_mb_mode *)modes_in, (jmhip_mb_mode *)modes_out,
                                                                     (const jmhip_quant *)quants, (JmMbRes *)records, (JmMbCoded *)coded, n);
  if (hipGetLastError() != hipSuccess) return jm_fail(c, JMHIP_ERR_DEVICE, "frame_fused_kernel launch");
  return JMHIP_OK;
}

extern "C" void jmhip_flat_quant(jmhip_quant *q, int qp, int offset11, int is8x8)
{
  // quant_coef / dequant_coef (block.c:39-55): three position classes (0,0) (odd,odd) (mixed) per qp%6;
  // quant_coef8 / dequant_coef8 (transform8x8.c:39-167): six classes keyed on (j&3, i&3).
  static const int qc4[6][3] = {{13107, 5243, 8066}, {11916, 4660, 7490}, {10082, 4194, 6554}, {9362, 3647, 5825}, {8192, 3355, 5243}, {7282, 2893, 4559}};
  static const int dq4[6][3] = {{10, 16, 13}, {11, 18, 14}, {13, 20, 16}, {14, 23, 18}, {16, 25, 20}, {18, 29, 23}};
  static const int qc8[6][6] = {{13107, 11428, 20972, 12222, 16777, 15481}, {11916, 10826, 19174, 11058, 14980, 14290}, {10082, 8943, 15978, 9675, 12710, 11985},
                                {9362, 8228, 14913, 8931, 11984, 11259}, {8192, 7346, 13159, 7740, 10486, 9777}, {7282, 6428, 11570, 6830, 9118, 8640}};
  static const int dq8[6][6] = {{20, 18, 32, 19, 25, 24}, {22, 19, 35, 21, 28, 26}, {26, 23, 42, 24, 33, 31}, {28, 25, 45, 26, 35, 33}, {32, 28, 51, 30, 40, 38}, {36, 32, 58, 34, 46, 43}};
  if (!q || qp < 0 || qp > 87) return;
  memset(q, 0, sizeof(*q));
  const int k = qp % 6, per = qp / 6;
  q->qp = qp; q->max_val = 255; q->img_qp = qp;
  if (!is8x8) {
    for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) {
      const int cls = ((j & 1) == 0 && (i & 1) == 0) ? 0 : (((j & 1) && (i & 1)) ? 1 : 2);
      q->levelscale[j * 4 + i] = qc4[k][cls];                       // q_matrix.c:482
      q->invlevelscale[j * 4 + i] = dq4[k][cls] << 4;               // q_matrix.c:483
      q->leveloffset[j * 4 + i] = offset11 << (15 + per - 11);      // q_offsets.c:508-523
    }
  } else {
    for (int j = 0; j < 8; j++) for (int i = 0; i < 8; i++) {
      int a = j & 3, b = i & 3, ca = a == 0 ? 0 : (a == 2 ? 2 : 1), cb = b == 0 ? 0 : (b == 2 ? 2 : 1);
      if (ca > cb) { int t = ca; ca = cb; cb = t; }
      const int cls = (ca == cb) ? ca : (ca == 0 ? (cb == 1 ? 3 : 4) : 5);
      q->levelscale[j * 8 + i] = qc8[k][cls];
      q->invlevelscale[j * 8 + i] = dq8[k][cls] << 4;
      q->leveloffset[j * 8 + i] = offset11 << (16 + per - 11);
    }
  }
}

// launch the TQ kernel of `kind` over device-resident job/quant/result arrays
int jm_launch_tq(jmhip_ctx *c, int kind, int yuv_format, const void *jobs, const void *quants, void *results, int n)
{
  const jmhip_tq_job *dj = (const jmhip_tq_job *)jobs;
  const jmhip_quant *dq = (const jmhip_quant *)quants;
  jmhip_tq_result *dr = (jmhip_tq_result *)results;
  const int select = (kind & JMHIP_TQ_SELECT) ? 1 : 0;
  kind &= ~JMHIP_TQ_SELECT;
  switch (kind) {
  case JMHIP_TQ_LUMA4x4:   tq_luma4x4_kernel<<<jm_xcd_grid((n * 16 + 255) / 256), 256, 0, c->stream>>>(dj, dq, dr, n, select); break;
  case JMHIP_TQ_LUMA8x8:   tq_luma8x8_kernel<<<(n * 4 + 63) / 64, 64, 0, c->stream>>>(dj, dq, dr, n, select); break;
  case JMHIP_TQ_LUMA16x16: tq_luma16x16_kernel<<<(n + 63) / 64, 64, 0, c->stream>>>(dj, dq, dr, n); break;
  default:
    if (yuv_format == JMHIP_YUV420) tq_chroma420_kernel<<<jm_xcd_grid((n * 4 + 255) / 256), 256, 0, c->stream>>>(dj, dq, dr, n);
    else tq_chroma_kernel<<<(n + 63) / 64, 64, 0, c->stream>>>(dj, dq, dr, n, yuv_format);
    break;
  }
  JM_HIP_CHECK(c, hipGetLastError());
  return JMHIP_OK;
}

extern "C" int jmhip_tq_batch(jmhip_ctx *c, int kind, int yuv_format, const jmhip_quant *quants, int nquants,
                              const jmhip_tq_job *jobs, int n, jmhip_tq_result *results)
{
  if (!c || !quants || !jobs || !results || n <= 0 || nquants <= 0) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_tq_batch: NULL/empty arguments") : JMHIP_ERR_ARG;
  if (kind < JMHIP_TQ_LUMA4x4 || kind > JMHIP_TQ_CHROMA) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_tq_batch: unknown kind");
  if (kind == JMHIP_TQ_CHROMA && (yuv_format < JMHIP_YUV420 || yuv_format > JMHIP_YUV444)) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_tq_batch: chroma needs a chroma format");
  for (int i = 0; i < nquants; i++) {
    if (quants[i].qp < 0 || quants[i].qp > 87) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_tq_batch: qp out of range");
    if (quants[i].disthres < 0 || quants[i].disthres > 1 || quants[i].max_val != 255) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_tq_batch: disthres/max_val out of range");
  }
  for (int i = 0; i < n; i++) {
    if (jobs[i].quant < 0 || jobs[i].quant >= nquants || jobs[i].quant_dc < 0 || jobs[i].quant_dc >= nquants) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_tq_batch: quantiser index out of range");
    if (kind == JMHIP_TQ_CHROMA && (jobs[i].uv < 0 || jobs[i].uv > 1)) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_tq_batch: uv must be 0 or 1");
  }
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  if (c->tq_capacity < n) {
    if (c->tq_jobs_dev) JM_HIP_CHECK(c, hipFree(c->tq_jobs_dev));
    if (c->tq_res_dev) JM_HIP_CHECK(c, hipFree(c->tq_res_dev));
    c->tq_jobs_dev = c->tq_res_dev = nullptr; c->tq_capacity = 0;
    if (hipMalloc(&c->tq_jobs_dev, sizeof(jmhip_tq_job) * (size_t)n) != hipSuccess) return jm_fail(c, JMHIP_ERR_NOMEM, "TQ job array");
    if (hipMalloc(&c->tq_res_dev, sizeof(jmhip_tq_result) * (size_t)n) != hipSuccess) return jm_fail(c, JMHIP_ERR_NOMEM, "TQ result array");
    c->tq_capacity = n;
  }
  if (c->tq_qcap < nquants) {
    if (c->tq_quant_dev) JM_HIP_CHECK(c, hipFree(c->tq_quant_dev));
    c->tq_quant_dev = nullptr; c->tq_qcap = 0;
    if (hipMalloc(&c->tq_quant_dev, sizeof(jmhip_quant) * (size_t)nquants) != hipSuccess) return jm_fail(c, JMHIP_ERR_NOMEM, "TQ quantiser array");
    c->tq_qcap = nquants;
  }
  JM_HIP_CHECK(c, hipMemcpyAsync(c->tq_jobs_dev, jobs, sizeof(jmhip_tq_job) * (size_t)n, hipMemcpyHostToDevice, c->stream));
  JM_HIP_CHECK(c, hipMemcpyAsync(c->tq_quant_dev, quants, sizeof(jmhip_quant) * (size_t)nquants, hipMemcpyHostToDevice, c->stream));
  JM_HIP_CHECK(c, hipMemsetAsync(c->tq_res_dev, 0, sizeof(jmhip_tq_result) * (size_t)n, c->stream));
  jm_stage_begin(c, JMHIP_STAGE_TQ);
  int rc = jm_launch_tq(c, kind, yuv_format, c->tq_jobs_dev, c->tq_quant_dev, c->tq_res_dev, n);
  jm_stage_end(c, JMHIP_STAGE_TQ);
  if (rc) return rc;
  JM_HIP_CHECK(c, hipMemcpyAsync(results, c->tq_res_dev, sizeof(jmhip_tq_result) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  return JMHIP_OK;
}
