// deblock.hip -- the in-loop deblocking filter of a frame picture, in place on the context's reconstructed picture.
//
// Replaces DeblockFrame (lencod/src/loopFilter.c:87): DeblockMb :128, GetStrengthNormal :263, EdgeLoopLumaNormal :529,
// EdgeLoopChromaNormal :815. Frame pictures without MBAFF; SP/SI slices are not handled (the caller keeps JM's path).
//
// Two kernels:
//  * deblock_strength_kernel -- fully parallel, one thread per (macroblock, direction, edge): the four boundary strengths of the
//    edge (GetStrengthNormal works in groups of four samples) and alpha / beta / tc0 of each colour plane. 24 bytes per edge.
//  * deblock_filter_kernel -- JM filters macroblocks in address order, in place: macroblock (x, y) reads samples its left
//    neighbour's HORIZONTAL pass and its upper-right neighbour's VERTICAL pass have already changed. That order is part of the
//    result, so the kernel walks the 2:1 wavefront d = x + 2y (all macroblocks of one d are independent) inside ONE workgroup:
//    16 lanes per macroblock (a lane owns one line of samples across all four edges of a direction: the edges of a line are
//    filtered in registers), up to 64 macroblocks per pass, a workgroup barrier between the vertical and the horizontal pass
//    and between diagonals. mbw + 2(mbh-1) serial steps per picture: latency-bound by construction (254 steps at 1080p).
//    No inter-workgroup waiting anywhere: every wave runs the same number of barriers and leaves.
#include "jmhip_internal.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

namespace {

// Tables 8-16 / 8-17 of the standard (JM: ALPHA_TABLE, BETA_TABLE, CLIP_TAB columns 1..3; column 4 repeats column 3)
__constant__ uint8_t c_alpha[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28, 32, 36, 40, 45,
                                    50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255};
__constant__ uint8_t c_beta[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10,
                                   11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18};
__constant__ uint8_t c_tc0[3][52] = {
  {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6, 6, 7, 8, 9, 10, 11, 13},
  {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5, 5, 6, 7, 8, 8, 10, 11, 12, 13, 15, 17},
  {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 23, 25}};

// 16 bytes per (macroblock, direction, edge), fetched whole and picked apart with shifts (no dependent byte loads on the serial path):
//   bs    = bS of the four sample groups, one byte each
//   pl[p] = alpha | beta << 8 | tc0(bS 1) << 13 | tc0(bS 2) << 18 | tc0(bS 3) << 23 | on << 28      (alpha <= 255, beta <= 18, tc0 <= 25)
// on: bit 0 = the edge is filtered and some strength is non-zero; bit 1 = its luma samples too (not an inner edge of an 8x8-transform macroblock)
struct EdgeInfo { uint32_t bs; uint32_t pl[3]; };
static_assert(sizeof(EdgeInfo) == 16, "EdgeInfo layout");

__device__ __forceinline__ int iabs_(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int clip3(int lo, int hi, int v) { return min(max(v, lo), hi); }

__global__ __launch_bounds__(256) void deblock_strength_kernel(const jmhip_deblock_mb *__restrict__ mbs, const jmhip_deblock_blk *__restrict__ blks,
                                                               EdgeInfo *__restrict__ out, int mbw, int mbh, int mvlimit)
{
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int mb = t >> 3;
  if (mb >= mbw * mbh) return;
  const int dir = (t >> 2) & 1, edge = t & 3, mbx = mb % mbw, mby = mb / mbw;
  const jmhip_deblock_mb q = mbs[mb];
  EdgeInfo e;
  e.bs = e.pl[0] = e.pl[1] = e.pl[2] = 0;
  uint32_t bs[4] = {0, 0, 0, 0};
  bool filtered = q.disable_idc != 1;
  if (edge == 0) {
    // loopFilter.c:139-140 (picture border) and :163-169 (idc 2: slice border, from the availability the encoder left)
    const bool nb = dir ? (q.disable_idc == 2 ? q.avail_b != 0 : mby != 0) : (q.disable_idc == 2 ? q.avail_a != 0 : mbx != 0);
    // an "available" flag can never point outside the picture; guard the neighbour index all the same
    filtered = filtered && nb && (dir ? mby != 0 : mbx != 0);
  }
  if (filtered) {
    const int pmb = edge ? mb : (dir ? mb - mbw : mb - 1);
    const jmhip_deblock_mb p = mbs[pmb];
    const int pmbx = pmb % mbw, pmby = pmb / mbw;
    int any = 0;
    if (p.intra || q.intra) {
      for (int i = 0; i < 4; i++) bs[i] = edge == 0 ? 4 : 3;                       // :399
      any = 1;
    } else {
      const int w4 = mbw * 4;
      for (int g = 0; g < 4; g++) {
        const int qbx = dir ? g : edge, qby = dir ? edge : g;
        const int pbx = dir ? g : (edge + 3) & 3, pby = dir ? (edge + 3) & 3 : g;
        int v;
        if (((q.cbp_blk >> (qby * 4 + qbx)) & 1) || ((p.cbp_blk >> (pby * 4 + pbx)) & 1)) v = 2;        // :325
        else {
          const jmhip_deblock_blk a = blks[(size_t)(mby * 4 + qby) * w4 + mbx * 4 + qbx];
          const jmhip_deblock_blk b = blks[(size_t)(pmby * 4 + pby) * w4 + pmbx * 4 + pbx];
          const long long a0 = a.ref_id[0], a1 = a.ref_id[1], b0 = b.ref_id[0], b1 = b.ref_id[1];
          auto far = [&](int l, int m) { return (int)(iabs_(a.mv[l][0] - b.mv[m][0]) >= 4) | (int)(iabs_(a.mv[l][1] - b.mv[m][1]) >= mvlimit); };
          if ((a0 == b0 && a1 == b1) || (a0 == b1 && a1 == b0)) {
            if (a0 != a1) v = (a0 == b0) ? (far(0, 0) | far(1, 1)) : (far(0, 1) | far(1, 0));           // :346-364
            else v = (far(0, 0) | far(1, 1)) && (far(0, 1) | far(1, 0));                              // :369-379
          } else v = 1;
        }
        bs[g] = (uint32_t)v;
        any |= v;
      }
    }
    if (any) {
      const uint32_t on = 1u | ((q.transform_8x8 && (edge & 1)) ? 0u : 2u);           // filterNon8x8LumaEdgesFlag :153
      for (int pl = 0; pl < 3; pl++) {
        const int qp = pl ? (p.qpc[pl - 1] + q.qpc[pl - 1] + 1) >> 1 : (p.qp + q.qp + 1) >> 1;        // :566 / :851
        const int ia = clip3(0, 51, qp + q.alpha_c0_offset), ib = clip3(0, 51, qp + q.beta_offset);
        e.pl[pl] = (uint32_t)c_alpha[ia] | ((uint32_t)c_beta[ib] << 8) | ((uint32_t)c_tc0[0][ia] << 13) | ((uint32_t)c_tc0[1][ia] << 18) |
                   ((uint32_t)c_tc0[2][ia] << 23) | (on << 28);
      }
      e.bs = bs[0] | (bs[1] << 8) | (bs[2] << 16) | (bs[3] << 24);
    }
  }
  out[t] = e;
}

// one line across a luma-type edge; s points at q0, s[-4..3] = p3 p2 p1 p0 | q0 q1 q2 q3 (EdgeLoopLumaNormal :583-661)
__device__ __forceinline__ void luma_line(int *s, int bS, int alpha, int beta, int tc0)
{
  const int L3 = s[-4], L2 = s[-3], L1 = s[-2], L0 = s[-1], R0 = s[0], R1 = s[1], R2 = s[2], R3 = s[3];
  const int delta = R0 - L0, ad = iabs_(delta);
  if (!bS || ad >= alpha || iabs_(R0 - R1) >= beta || iabs_(L0 - L1) >= beta) return;
  if (bS == 4) {
    const int small_gap = ad < ((alpha >> 2) + 2);
    const int aq = (iabs_(R0 - R2) < beta) & small_gap, ap = (iabs_(L0 - L2) < beta) & small_gap, RL0 = L0 + R0;
    if (ap) {
      s[-3] = (((L3 + L2) << 1) + L2 + L1 + RL0 + 4) >> 3;
      s[-2] = (L2 + L1 + L0 + R0 + 2) >> 2;
      s[-1] = (R1 + ((L1 + RL0) << 1) + L2 + 4) >> 3;
    } else s[-1] = ((L1 << 1) + L0 + R1 + 2) >> 2;
    if (aq) {
      s[0] = (L1 + ((R1 + RL0) << 1) + R2 + 4) >> 3;
      s[1] = (R2 + R0 + R1 + L0 + 2) >> 2;
      s[2] = (((R3 + R2) << 1) + R2 + R1 + RL0 + 4) >> 3;
    } else s[0] = ((R1 << 1) + R0 + L1 + 2) >> 2;
  } else {
    const int RL0 = (L0 + R0 + 1) >> 1, aq = iabs_(R0 - R2) < beta, ap = iabs_(L0 - L2) < beta, c0 = tc0 + ap + aq;
    const int dif = clip3(-c0, c0, ((delta << 2) + (L1 - R1) + 4) >> 3);
    if (ap) s[-2] = L1 + clip3(-tc0, tc0, (L2 + RL0 - (L1 << 1)) >> 1);
    s[-1] = clip3(0, 255, L0 + dif);
    s[0] = clip3(0, 255, R0 - dif);
    if (aq) s[1] = R1 + clip3(-tc0, tc0, (R2 + RL0 - (R1 << 1)) >> 1);
  }
}

// one line across a chroma edge; s[-2..1] = p1 p0 | q0 q1 (EdgeLoopChromaNormal :880-905)
__device__ __forceinline__ void chroma_line(int *s, int bS, int alpha, int beta, int tc0)
{
  const int L1 = s[-2], L0 = s[-1], R0 = s[0], R1 = s[1];
  const int delta = R0 - L0;
  if (!bS || iabs_(delta) >= alpha || iabs_(R0 - R1) >= beta || iabs_(L0 - L1) >= beta) return;
  if (bS == 4) {
    s[0] = ((R1 << 1) + R0 + L1 + 2) >> 2;
    s[-1] = ((L1 << 1) + L0 + R1 + 2) >> 2;
  } else {
    const int c0 = tc0 + 1, dif = clip3(-c0, c0, ((delta << 2) + (L1 - R1) + 4) >> 3);
    s[-1] = clip3(0, 255, L0 + dif);
    s[0] = clip3(0, 255, R0 - dif);
  }
}

// the four edges of one (macroblock, direction) in registers
struct Edges4 { uint32_t u[4][4]; };
__device__ __forceinline__ Edges4 load_edges(const EdgeInfo *ei)       // global or LDS, 16-byte aligned
{
  Edges4 E;
  const uint4 *p = reinterpret_cast<const uint4 *>(ei);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint4 v = p[k];
    E.u[k][0] = v.x; E.u[k][1] = v.y; E.u[k][2] = v.z; E.u[k][3] = v.w;
  }
  return E;
}
__device__ __forceinline__ uint32_t e_plane(const uint32_t *u, int pl) { return pl == 0 ? u[1] : pl == 1 ? u[2] : u[3]; }
__device__ __forceinline__ int e_on(const uint32_t *u) { return u[1] >> 28; }
__device__ __forceinline__ int e_bs(const uint32_t *u, int g) { return (u[0] >> (8 * g)) & 255; }
__device__ __forceinline__ int e_alpha(uint32_t w) { return w & 255; }
__device__ __forceinline__ int e_beta(uint32_t w) { return (w >> 8) & 31; }
__device__ __forceinline__ int e_tc(uint32_t w, int bS) { return bS ? (w >> (13 + 5 * (min(bS, 3) - 1))) & 31 : 0; }

// the four edges of a direction on one line of samples in registers: s[0..3] belong to the neighbour, s[4..] to this macroblock
__device__ __forceinline__ void luma_edges(int *s, const Edges4 &ei, int pl, int g)
{
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const uint32_t *E = ei.u[e];
    if (e_on(E) == 3) {
      const int bS = e_bs(E, g);
      const uint32_t w = e_plane(E, pl);
      luma_line(&s[4 + 4 * e], bS, e_alpha(w), e_beta(w), e_tc(w, bS));
    }
  }
}
__device__ __forceinline__ void chroma_edges(int *s, const Edges4 &ei, int pl, int g, int off0, int off1, int off2, int off3)
{
  const int off[4] = {off0, off1, off2, off3};      // luma edge e -> chroma sample offset, < 0: no chroma edge
#pragma unroll
  for (int e = 0; e < 4; e++) {
    if (off[e] < 0) continue;
    const uint32_t *E = ei.u[e];
    if (e_on(E) & 1) {
      const int bS = e_bs(E, g);
      const uint32_t w = e_plane(E, pl);
      chroma_line(&s[4 + off[e]], bS, e_alpha(w), e_beta(w), e_tc(w, bS));
    }
  }
}

// a luma-type line of 16 samples of this macroblock + the 4 before it; `step` = distance in bytes between samples of the line
// (1: a row, vertical edges; pitch: a column, horizontal edges). g = the line's strength group (line >> 2).
template <bool ROW>
__device__ __forceinline__ void luma_type_line(uint8_t *base, int pitch, bool has_before, const Edges4 &ei, int pl, int g)
{
  int s[20];
  if (ROW) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(base - 4);
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const uint32_t v = (k || has_before) ? w[k] : 0u;
      s[4 * k] = v & 255; s[4 * k + 1] = (v >> 8) & 255; s[4 * k + 2] = (v >> 16) & 255; s[4 * k + 3] = v >> 24;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 20; k++) s[k] = (k >= 4 || has_before) ? base[(ptrdiff_t)(k - 4) * pitch] : 0;
  }
  luma_edges(s, ei, pl, g);
  if (ROW) {
    uint32_t *w = reinterpret_cast<uint32_t *>(base - 4);
#pragma unroll
    for (int k = 0; k < 5; k++)
      if (k || has_before) w[k] = (uint32_t)s[4 * k] | ((uint32_t)s[4 * k + 1] << 8) | ((uint32_t)s[4 * k + 2] << 16) | ((uint32_t)s[4 * k + 3] << 24);
  } else {
#pragma unroll
    for (int k = 1; k < 19; k++)                          // samples -3 .. 14 can change
      if (k >= 4 || has_before) base[(ptrdiff_t)(k - 4) * pitch] = (uint8_t)s[k];
  }
}

// a chroma line of N (8 or 16) samples + the ones before it; edges: luma edge e maps to chroma sample offset off[e] (< 0: none)
template <bool ROW, int N>
__device__ __forceinline__ void chroma_type_line(uint8_t *base, int pitch, bool has_before, const Edges4 &ei, int pl, int g,
                                                 int off0, int off1, int off2, int off3)
{
  int s[4 + N];
  if (ROW) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(base - 4);
#pragma unroll
    for (int k = 0; k < 1 + N / 4; k++) {
      const uint32_t v = (k || has_before) ? w[k] : 0u;
      s[4 * k] = v & 255; s[4 * k + 1] = (v >> 8) & 255; s[4 * k + 2] = (v >> 16) & 255; s[4 * k + 3] = v >> 24;
    }
  } else {
#pragma unroll
    for (int k = 2; k < 4 + N; k++) s[k] = (k >= 4 || has_before) ? base[(ptrdiff_t)(k - 4) * pitch] : 0;
    s[0] = s[1] = 0;
  }
  chroma_edges(s, ei, pl, g, off0, off1, off2, off3);
  if (ROW) {
    uint32_t *w = reinterpret_cast<uint32_t *>(base - 4);
#pragma unroll
    for (int k = 0; k < 1 + N / 4; k++)
      if (k || has_before) w[k] = (uint32_t)s[4 * k] | ((uint32_t)s[4 * k + 1] << 8) | ((uint32_t)s[4 * k + 2] << 16) | ((uint32_t)s[4 * k + 3] << 24);
  } else {
#pragma unroll
    for (int k = 3; k < 4 + N; k++)
      if (k >= 4 || has_before) base[(ptrdiff_t)(k - 4) * pitch] = (uint8_t)s[k];
  }
}

constexpr int DBK_MAXB = 64;
struct DeblockDev {
  uint8_t *y, *u, *v;
  const EdgeInfo *edges;        // [mb][dir][edge]
  int W, Wc, mbw;
  int nbands, band_row[DBK_MAXB + 1];   // workgroup b filters macroblock rows [band_row[b], band_row[b + 1]): bands whose first row has no filtered top edge are independent
  int dbg;                      // JMHIP_DBK_DEBUG (timing experiments only): 1 skip the vertical pass, 2 skip the horizontal pass, 4 skip fetch / write-back
};

// FMT: JMHIP_YUV400 / 420 / 422 / 444
template <int FMT>
__global__ __launch_bounds__(1024) void deblock_filter_kernel(DeblockDev D)
{
  const int slot = threadIdx.x >> 4, l = threadIdx.x & 15;
  const int row0 = D.band_row[blockIdx.x], rows = D.band_row[blockIdx.x + 1] - row0;
  const int last_d = D.mbw - 1 + 2 * (rows - 1);
  for (int d = 0; d <= last_d; d++) {
    const int y_lo = max(0, (d - D.mbw + 2) >> 1), y_hi = min(rows - 1, d >> 1);
    const int count = y_hi - y_lo + 1;
    for (int dir = 0; dir < 2; dir++) {
      for (int k = slot; k < count; k += 64) {
        const int mby = row0 + y_lo + k, mbx = d - 2 * (y_lo + k);
        const Edges4 ei = load_edges(D.edges + ((size_t)(mby * D.mbw + mbx) * 2 + dir) * 4);
        // the first row of a later band has its top edges off by construction and must not touch the rows above: another workgroup owns them
        const bool before = dir ? (mby != 0 && !(blockIdx.x > 0 && y_lo + k == 0)) : mbx != 0;
        if (dir == 0) {
          // vertical edges: lane = row
          luma_type_line<true>(D.y + (size_t)(mby * 16 + l) * D.W + mbx * 16, D.W, before, ei, 0, l >> 2);
          if (FMT == JMHIP_YUV444) {
            luma_type_line<true>(D.u + (size_t)(mby * 16 + l) * D.W + mbx * 16, D.W, before, ei, 1, l >> 2);
            luma_type_line<true>(D.v + (size_t)(mby * 16 + l) * D.W + mbx * 16, D.W, before, ei, 2, l >> 2);
          } else if (FMT == JMHIP_YUV420) {
            // 8 rows x 2 planes; chroma_edge[0][e][420] = 0, -, 4, - ; StrengthIdx = ((row >> 1) << 2) + (row & 1) -> group row >> 1
            uint8_t *pl = (l >> 3) ? D.v : D.u;
            const int r = l & 7;
            chroma_type_line<true, 8>(pl + (size_t)(mby * 8 + r) * D.Wc + mbx * 8, D.Wc, before, ei, 1 + (l >> 3), r >> 1, 0, -1, 4, -1);
          } else if (FMT == JMHIP_YUV422) {
            // 16 rows per plane: StrengthIdx = row -> group row >> 2
            chroma_type_line<true, 8>(D.u + (size_t)(mby * 16 + l) * D.Wc + mbx * 8, D.Wc, before, ei, 1, l >> 2, 0, -1, 4, -1);
            chroma_type_line<true, 8>(D.v + (size_t)(mby * 16 + l) * D.Wc + mbx * 8, D.Wc, before, ei, 2, l >> 2, 0, -1, 4, -1);
          }
        } else {
          // horizontal edges: lane = column
          luma_type_line<false>(D.y + (size_t)(mby * 16) * D.W + mbx * 16 + l, D.W, before, ei, 0, l >> 2);
          if (FMT == JMHIP_YUV444) {
            luma_type_line<false>(D.u + (size_t)(mby * 16) * D.W + mbx * 16 + l, D.W, before, ei, 1, l >> 2);
            luma_type_line<false>(D.v + (size_t)(mby * 16) * D.W + mbx * 16 + l, D.W, before, ei, 2, l >> 2);
          } else if (FMT == JMHIP_YUV420) {
            uint8_t *pl = (l >> 3) ? D.v : D.u;
            const int cx = l & 7;
            chroma_type_line<false, 8>(pl + (size_t)(mby * 8) * D.Wc + mbx * 8 + cx, D.Wc, before, ei, 1 + (l >> 3), cx >> 1, 0, -1, 4, -1);
          } else if (FMT == JMHIP_YUV422) {
            // chroma_edge[1][e][422] = 0, 4, 8, 12: every luma edge has a chroma edge; 8 columns -> group col >> 1
            uint8_t *pl = (l >> 3) ? D.v : D.u;
            const int cx = l & 7;
            chroma_type_line<false, 16>(pl + (size_t)(mby * 16) * D.Wc + mbx * 8 + cx, D.Wc, before, ei, 1 + (l >> 3), cx >> 1, 0, 4, 8, 12);
          }
        }
      }
      __syncthreads();
    }
  }
}

// ---- the same walk with the three live diagonals held in LDS (all four chroma formats).
// A macroblock of diagonal d is still changed on d+1 (its right neighbour's vertical pass: columns 13..15) and on d+2 (the
// macroblock below: rows 13..15), then it is final. So the kernel keeps a ring of three diagonals of 384-byte tiles
// [Y 16x16 | U 8x8 | V 8x8] in LDS: a diagonal is fetched from HBM once (16-byte row loads, prefetched one diagonal ahead into
// registers), filtered twice in LDS, touched by its neighbours there, and written back once two diagonals later. The serial path
// then only sees LDS latency; one CU's vector-memory path, which the global-memory kernel above saturates with byte accesses,
// carries 24 wide loads and stores per macroblock. Tile k of a ring slot is always filled, filtered and written back by the same
// 16 lanes, and a macroblock's two passes run in one wave, so ONE workgroup barrier per diagonal is all the data flow needs.
constexpr int DBK_ETILE = 128;
// CF: 0 = no chroma (4:0:0), 1 = 4:2:0 (two 8x8 planes), 2 = 4:2:2 (two 8-wide x 16-high planes), 3 = 4:4:4 (three 16x16 planes that all
// take the luma filter); tile = [Y 16x16 | U | V]
__host__ __device__ constexpr int dbk_tile(int cf) { return cf == 3 ? 768 : cf == 2 ? 512 : 384; }

struct Diag { int y_lo, count; };
__device__ __forceinline__ Diag diag_of(int d, int mbw, int rows, int last_d)
{
  Diag g;
  g.y_lo = max(0, (d - mbw + 2) >> 1);
  g.count = (d < 0 || d > last_d) ? 0 : min(rows - 1, d >> 1) - g.y_lo + 1;
  return g;
}
__device__ __forceinline__ void unpack4(int *s, uint32_t v) { s[0] = v & 255; s[1] = (v >> 8) & 255; s[2] = (v >> 16) & 255; s[3] = v >> 24; }
__device__ __forceinline__ uint32_t pack4(const int *s) { return (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)s[3] << 24); }

// HBM -> registers: this lane's share of macroblock k of diagonal d (one luma row, one chroma row, 8 bytes of the edge records)
struct DbkFetch { uint4 y, u4, v4; uint2 c, c2, e; };
template <int CF>
__device__ __forceinline__ void dbk_prefetch(const DeblockDev &D, int row0, int d, const Diag &g, int k, int l, int cpl, int cl, DbkFetch &f)
{
  if (k < g.count) {
    const int y = g.y_lo + k, mby = row0 + y, mbx = d - 2 * y;
    f.y = *reinterpret_cast<const uint4 *>(D.y + (size_t)(mby * 16 + l) * D.W + mbx * 16);
    if (CF == 1) f.c = *reinterpret_cast<const uint2 *>((cpl ? D.v : D.u) + (size_t)(mby * 8 + cl) * D.Wc + mbx * 8);
    if (CF == 2) {                                   // row l of both planes
      f.c = *reinterpret_cast<const uint2 *>(D.u + (size_t)(mby * 16 + l) * D.Wc + mbx * 8);
      f.c2 = *reinterpret_cast<const uint2 *>(D.v + (size_t)(mby * 16 + l) * D.Wc + mbx * 8);
    }
    if (CF == 3) {
      f.u4 = *reinterpret_cast<const uint4 *>(D.u + (size_t)(mby * 16 + l) * D.W + mbx * 16);
      f.v4 = *reinterpret_cast<const uint4 *>(D.v + (size_t)(mby * 16 + l) * D.W + mbx * 16);
    }
    f.e = reinterpret_cast<const uint2 *>(D.edges + (size_t)(mby * D.mbw + mbx) * 8)[l];
  }
}
template <int CF>
__device__ __forceinline__ void dbk_fill(uint8_t *slot, uint8_t *etiles, const Diag &g, int k, int l, int cpl, int cl, const DbkFetch &f)
{
  if (k < g.count) {
    uint8_t *tile = slot + (size_t)k * dbk_tile(CF);
    *reinterpret_cast<uint4 *>(tile + l * 16) = f.y;
    if (CF == 1) *reinterpret_cast<uint2 *>(tile + 256 + cpl * 64 + cl * 8) = f.c;
    if (CF == 2) { *reinterpret_cast<uint2 *>(tile + 256 + l * 8) = f.c; *reinterpret_cast<uint2 *>(tile + 384 + l * 8) = f.c2; }
    if (CF == 3) { *reinterpret_cast<uint4 *>(tile + 256 + l * 16) = f.u4; *reinterpret_cast<uint4 *>(tile + 512 + l * 16) = f.v4; }
    reinterpret_cast<uint2 *>(etiles + (size_t)k * DBK_ETILE)[l] = f.e;
  }
}

template <int NP, int CF>
__global__ __launch_bounds__(1024) void deblock_lds_kernel(DeblockDev D, int S)
{
  constexpr int DBK_TILE = dbk_tile(CF);
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint8_t *const etiles = lds + (size_t)3 * S * DBK_TILE;
  const int grp = threadIdx.x >> 4, l = threadIdx.x & 15;
  const int cpl = l >> 3, cl = l & 7;                       // chroma: lanes 0..7 own U lines, 8..15 V lines
  const int row0 = D.band_row[blockIdx.x], rows = D.band_row[blockIdx.x + 1] - row0;
  const int last_d = D.mbw - 1 + 2 * (rows - 1);
  DbkFetch pf0, pf1;
  pf0.y = pf1.y = pf0.u4 = pf0.v4 = pf1.u4 = pf1.v4 = make_uint4(0, 0, 0, 0); pf0.c = pf0.c2 = pf0.e = pf1.c = pf1.c2 = pf1.e = make_uint2(0, 0);

  {
    const Diag g0 = diag_of(0, D.mbw, rows, last_d);
    dbk_prefetch<CF>(D, row0, 0, g0, grp, l, cpl, cl, pf0);
    if (NP > 1) dbk_prefetch<CF>(D, row0, 0, g0, grp + 64, l, cpl, cl, pf1);
  }
  for (int d = 0; d <= last_d + 2; d++) {
    const Diag g = diag_of(d, D.mbw, rows, last_d), gl = diag_of(d - 1, D.mbw, rows, last_d), gt = diag_of(d - 2, D.mbw, rows, last_d);
    uint8_t *const slot = lds + (size_t)(d % 3) * S * DBK_TILE;
    uint8_t *const slot_l = lds + (size_t)((d + 2) % 3) * S * DBK_TILE;      // diagonal d-1
    uint8_t *const slot_t = lds + (size_t)((d + 1) % 3) * S * DBK_TILE;      // diagonal d-2
    // ---- fill: the registers fetched during the previous diagonal; then start fetching the next one
    dbk_fill<CF>(slot, etiles, g, grp, l, cpl, cl, pf0);
    if (NP > 1) dbk_fill<CF>(slot, etiles, g, grp + 64, l, cpl, cl, pf1);
    {
      const Diag gn = diag_of(d + 1, D.mbw, rows, last_d);
      dbk_prefetch<CF>(D, row0, d + 1, gn, grp, l, cpl, cl, pf0);
      if (NP > 1) dbk_prefetch<CF>(D, row0, d + 1, gn, grp + 64, l, cpl, cl, pf1);
    }
    // ---- vertical edges: lane = row
#pragma unroll
    for (int p = 0; p < NP; p++) {
      const int k = grp + 64 * p;
      if (k < g.count && !(D.dbg & 1)) {
        const int y = g.y_lo + k, mbx = d - 2 * y;
        uint8_t *tile = slot + (size_t)k * DBK_TILE;
        uint8_t *left = slot_l + (size_t)(y - gl.y_lo) * DBK_TILE;          // (mbx-1, y) lies on diagonal d-1
        const bool before = mbx != 0;
        const Edges4 ei = load_edges(reinterpret_cast<const EdgeInfo *>(etiles + (size_t)k * DBK_ETILE));
#pragma unroll
        for (int lp = 0; lp < (CF == 3 ? 3 : 1); lp++) {           // 4:4:4: the luma filter on U and V too (plane parameters 1, 2)
          int s[20];
          const int o = lp * 256 + l * 16;
          const uint4 row = *reinterpret_cast<const uint4 *>(tile + o);
          unpack4(s, before ? *reinterpret_cast<const uint32_t *>(left + o + 12) : 0u);
          unpack4(s + 4, row.x); unpack4(s + 8, row.y); unpack4(s + 12, row.z); unpack4(s + 16, row.w);
          luma_edges(s, ei, lp, l >> 2);
          if (before) *reinterpret_cast<uint32_t *>(left + o + 12) = pack4(s);
          *reinterpret_cast<uint4 *>(tile + o) = make_uint4(pack4(s + 4), pack4(s + 8), pack4(s + 12), pack4(s + 16));
        }
        // 4:2:0: 8 rows x 2 planes over the 16 lanes (strength group row >> 1); 4:2:2: 16 rows, a lane takes row l of both planes (row >> 2)
#pragma unroll
        for (int pass = 0; pass < (CF == 2 ? 2 : CF == 1 ? 1 : 0); pass++) {
          int s[12];
          const int pl = CF == 2 ? pass : cpl, o = CF == 2 ? 256 + pass * 128 + l * 8 : 256 + cpl * 64 + cl * 8;
          const uint2 row = *reinterpret_cast<const uint2 *>(tile + o);
          unpack4(s, before ? *reinterpret_cast<const uint32_t *>(left + o + 4) : 0u);
          unpack4(s + 4, row.x); unpack4(s + 8, row.y);
          chroma_edges(s, ei, 1 + pl, CF == 2 ? l >> 2 : cl >> 1, 0, -1, 4, -1);
          if (before) *reinterpret_cast<uint32_t *>(left + o + 4) = pack4(s);
          *reinterpret_cast<uint2 *>(tile + o) = make_uint2(pack4(s + 4), pack4(s + 8));
        }
      }
    }
    // No workgroup barrier here: the horizontal pass of a macroblock reads its own tile -- written just above by the 16 lanes of the
    // same wave, and a wave's LDS operations complete in order -- and the tile above it, last touched one diagonal ago.
    // ---- horizontal edges: lane = column
#pragma unroll
    for (int p = 0; p < NP; p++) {
      const int k = grp + 64 * p;
      if (k < g.count && !(D.dbg & 2)) {
        const int y = g.y_lo + k, mby = row0 + y, mbx = d - 2 * y;
        uint8_t *tile = slot + (size_t)k * DBK_TILE;
        uint8_t *top = slot_t + (size_t)(y - 1 - gt.y_lo) * DBK_TILE;       // (mbx, y-1) lies on diagonal d-2
        // first row of a band: the rows above live in HBM only -- and belong to another workgroup unless this is the call's first band
        // (a later band's first row has its top edges off by construction: it must neither read nor write back those rows)
        const bool before = mby != 0 && !(blockIdx.x > 0 && y == 0), top_lds = y > 0;
        const Edges4 ei = load_edges(reinterpret_cast<const EdgeInfo *>(etiles + (size_t)k * DBK_ETILE + 64));
#pragma unroll
        for (int lp = 0; lp < (CF == 3 ? 3 : 1); lp++) {
          int s[20];
          const int o = lp * 256 + l;
          uint8_t *gtop = (lp == 0 ? D.y : lp == 1 ? D.u : D.v) + (size_t)(mby * 16 - 4) * D.W + mbx * 16 + l;
#pragma unroll
          for (int r = 0; r < 4; r++) s[r] = !before ? 0 : top_lds ? top[o + (12 + r) * 16] : gtop[(size_t)r * D.W];
#pragma unroll
          for (int r = 0; r < 16; r++) s[4 + r] = tile[o + r * 16];
          luma_edges(s, ei, lp, l >> 2);
          if (before) {
#pragma unroll
            for (int r = 1; r < 4; r++) { if (top_lds) top[o + (12 + r) * 16] = (uint8_t)s[r]; else gtop[(size_t)r * D.W] = (uint8_t)s[r]; }
          }
#pragma unroll
          for (int r = 0; r < 15; r++) tile[o + r * 16] = (uint8_t)s[4 + r];
        }
        if (CF == 1 || CF == 2) {                       // lane = (plane, column); 4:2:2 has a chroma edge under every luma edge (rows 0, 4, 8, 12)
          constexpr int CH = CF == 2 ? 16 : 8, CP = 8 * CH;
          int s[4 + CH];
          const int o = 256 + cpl * CP + cl;
          uint8_t *gtop = (cpl ? D.v : D.u) + (size_t)(mby * CH - 2) * D.Wc + mbx * 8 + cl;
          s[0] = s[1] = 0;
#pragma unroll
          for (int r = 0; r < 2; r++) s[2 + r] = !before ? 0 : top_lds ? top[o + (CH - 2 + r) * 8] : gtop[(size_t)r * D.Wc];
#pragma unroll
          for (int r = 0; r < CH; r++) s[4 + r] = tile[o + r * 8];
          if (CF == 2) chroma_edges(s, ei, 1 + cpl, cl >> 1, 0, 4, 8, 12);
          else chroma_edges(s, ei, 1 + cpl, cl >> 1, 0, -1, 4, -1);
          if (before) { if (top_lds) top[o + (CH - 1) * 8] = (uint8_t)s[3]; else gtop[D.Wc] = (uint8_t)s[3]; }
#pragma unroll
          for (int r = 0; r < CH; r++) tile[o + r * 8] = (uint8_t)s[4 + r];
        }
      }
    }
    __syncthreads();
    // ---- diagonal d-2 is final: back to HBM
#pragma unroll
    for (int p = 0; p < NP; p++) {
      const int k = grp + 64 * p;
      if (k < gt.count && !(D.dbg & 4)) {
        const int y = gt.y_lo + k, mby = row0 + y, mbx = d - 2 - 2 * y;
        const uint8_t *tile = slot_t + (size_t)k * DBK_TILE;
        *reinterpret_cast<uint4 *>(D.y + (size_t)(mby * 16 + l) * D.W + mbx * 16) = *reinterpret_cast<const uint4 *>(tile + l * 16);
        if (CF == 1) *reinterpret_cast<uint2 *>((cpl ? D.v : D.u) + (size_t)(mby * 8 + cl) * D.Wc + mbx * 8) = *reinterpret_cast<const uint2 *>(tile + 256 + cpl * 64 + cl * 8);
        if (CF == 2) {
          *reinterpret_cast<uint2 *>(D.u + (size_t)(mby * 16 + l) * D.Wc + mbx * 8) = *reinterpret_cast<const uint2 *>(tile + 256 + l * 8);
          *reinterpret_cast<uint2 *>(D.v + (size_t)(mby * 16 + l) * D.Wc + mbx * 8) = *reinterpret_cast<const uint2 *>(tile + 384 + l * 8);
        }
        if (CF == 3) {
          *reinterpret_cast<uint4 *>(D.u + (size_t)(mby * 16 + l) * D.W + mbx * 16) = *reinterpret_cast<const uint4 *>(tile + 256 + l * 16);
          *reinterpret_cast<uint4 *>(D.v + (size_t)(mby * 16 + l) * D.W + mbx * 16) = *reinterpret_cast<const uint4 *>(tile + 512 + l * 16);
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------- relaxation schedule (4:2:0)
//
// JM filters macroblock k's vertical edges, then its horizontal edges, in place and in raster order. Written as a data flow, step D_k reads
//   the unfiltered samples of macroblock k, the four right-most columns of its left neighbour AFTER D_(k-1) (call that A_(k-1)), and the four
//   bottom rows of its upper neighbour after D_(up) AND after D_(up+1) has filtered the up neighbour's right edge (A_up overlaid with R_(up+1)),
// and produces A_k (its own 16x16 + chroma after both passes), R_k (what its left edge made of the left neighbour's columns 12..15) and
// T_k (what its top edge made of the upper neighbour's rows 12..15). The final picture is A_k overlaid with R_(k+1), then with T_(k+mbw).
// The dependencies are acyclic, so -- exactly as for the slice search (me_wave.hip) -- every macroblock can be evaluated at once from whatever its
// predecessors last produced, sweep after sweep, until a sweep changes nothing: that fixpoint is the raster-order result. A change only
// travels as far as a filter carries it (a few samples), so a handful of sweeps of ~10 us replace the 254 + serial steps of the wavefront.
// Layout of one macroblock's record in `st` (4:2:0; 4:2:2 has 16 chroma rows, 4:0:0 none): A.Y[16][16] A.U[8][8] A.V[8][8] | R.Y[16][4] R.U[8][4]
// R.V[8][4] | T.Y[4][16] T.U[4][8] T.V[4][8]. 4:4:4 stays with the wavefront kernels.
// chroma rows per macroblock / record geometry by chroma format (CF: 0 = 4:0:0, 1 = 4:2:0, 2 = 4:2:2)
template <int CF> struct DbrGeo {
  static constexpr int CH = CF == 2 ? 16 : CF == 1 ? 8 : 0;                  // chroma rows of a macroblock (8 columns)
  static constexpr int A = 256 + 2 * CH * 8, R = 64 + 2 * CH * 4, T = 64 + (CF ? 64 : 0), REC = A + R + T;
};

struct DbkRelax {
  const uint8_t *y, *u, *v;     // the unfiltered picture (read only during the sweeps)
  uint8_t *st;                  // [nmb][REC]
  const EdgeInfo *edges;
  const uint8_t *chg_prev; uint8_t *chg_next; int *n_changed;
  int W, Wc, mbw, mbh, first_sweep;
  int row0, rows;               // macroblock rows [row0, row0 + rows) are filtered; the row above row0 is read (and its bottom rows written) in the picture itself
};

template <int CF>
__global__ __launch_bounds__(64) void deblock_relax_kernel(DbkRelax D)
{
  using G = DbrGeo<CF>;
  constexpr int CH = G::CH, CT = (CH + 4) * 12;               // chroma tile: rows -4 .. CH-1, columns -4 .. 7
  __shared__ __attribute__((aligned(16))) uint8_t tY[4][20 * 20], tC[4][2][CF ? CT : 16];
  const int h = threadIdx.x >> 4, l = threadIdx.x & 15;
  const int nmb = D.mbw * D.rows, k = D.row0 * D.mbw + min((int)blockIdx.x * 4 + h, nmb - 1);
  const bool live = (int)blockIdx.x * 4 + h < nmb;
  const int mbx = k % D.mbw, mby = k / D.mbw;
  const bool up_rec = mby > D.row0;                            // the upper neighbour is part of this call (has a record); else the picture holds it
  const bool has_left = mbx > 0, has_up = mby > 0, has_ur = up_rec && mbx + 1 < D.mbw;
  bool active = D.first_sweep != 0;
  if (!active) active = (has_left && D.chg_prev[k - 1]) || (up_rec && D.chg_prev[k - D.mbw]) || (has_ur && D.chg_prev[k - D.mbw + 1]);
  active = active && live;
  uint8_t *Y = tY[h];
  uint8_t *const C2[2] = {tC[h][0], tC[h][1]};
  const uint8_t *rec = D.st + (size_t)k * G::REC;
  if (active) {
    // ---- stage: own samples (unfiltered), left margin = A_left columns 12..15, top margin = A_up rows 12..15 overlaid with R_(up+1)
    {
      const uint4 v = *reinterpret_cast<const uint4 *>(D.y + (size_t)(mby * 16 + l) * D.W + mbx * 16);
      uint32_t *d = reinterpret_cast<uint32_t *>(Y + (l + 4) * 20 + 4);
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      for (int t = l; t < 2 * CH; t += 16) {
        const int pl = t / (CH ? CH : 1), r = t - pl * CH;
        const uint2 c2 = *reinterpret_cast<const uint2 *>((pl ? D.v : D.u) + (size_t)(mby * CH + r) * D.Wc + mbx * 8);
        uint32_t *dc = reinterpret_cast<uint32_t *>(C2[pl] + (r + 4) * 12 + 4);
        dc[0] = c2.x; dc[1] = c2.y;
      }
    }
    if (has_left) {
      const uint8_t *a = rec - G::REC;
      *reinterpret_cast<uint32_t *>(Y + (l + 4) * 20) = *reinterpret_cast<const uint32_t *>(a + l * 16 + 12);
      for (int t = l; t < 2 * CH; t += 16) {
        const int pl = t / (CH ? CH : 1), r = t - pl * CH;
        *reinterpret_cast<uint32_t *>(C2[pl] + (r + 4) * 12) = *reinterpret_cast<const uint32_t *>(a + 256 + pl * CH * 8 + r * 8 + 4);
      }
    }
    if (has_up && !up_rec) {                                 // first row of a band: the rows above as the picture holds them
      const int r = l >> 2, q = l & 3;
      *reinterpret_cast<uint32_t *>(Y + r * 20 + 4 + q * 4) = *reinterpret_cast<const uint32_t *>(D.y + (size_t)(mby * 16 - 4 + r) * D.W + mbx * 16 + q * 4);
      if (CF) {
        const int pl = l >> 3, cr = (l >> 1) & 3, cq = l & 1;
        *reinterpret_cast<uint32_t *>(C2[pl] + cr * 12 + 4 + cq * 4) = *reinterpret_cast<const uint32_t *>((pl ? D.v : D.u) + (size_t)(mby * CH - 4 + cr) * D.Wc + mbx * 8 + cq * 4);
      }
    }
    if (up_rec) {
      const uint8_t *a = rec - (size_t)D.mbw * G::REC;
      const int r = l >> 2, q = l & 3;                       // luma: 4 rows x 4 dwords
      uint32_t v = *reinterpret_cast<const uint32_t *>(a + (12 + r) * 16 + q * 4);
      if (q == 3 && has_ur) v = *reinterpret_cast<const uint32_t *>(a + G::REC + G::A + (12 + r) * 4);            // R_(up+1).Y rows 12..15
      *reinterpret_cast<uint32_t *>(Y + r * 20 + 4 + q * 4) = v;
      if (CF) {                                              // chroma: per plane the last 4 rows x 2 dwords
        const int pl = l >> 3, cr = (l >> 1) & 3, cq = l & 1;
        uint32_t w = *reinterpret_cast<const uint32_t *>(a + 256 + pl * CH * 8 + (CH - 4 + cr) * 8 + cq * 4);
        if (cq == 1 && has_ur) w = *reinterpret_cast<const uint32_t *>(a + G::REC + G::A + 64 + pl * CH * 4 + (CH - 4 + cr) * 4);
        *reinterpret_cast<uint32_t *>(C2[pl] + cr * 12 + 4 + cq * 4) = w;
      }
    }
  }
  __syncthreads();
  if (active) {
    const Edges4 e0 = load_edges(D.edges + ((size_t)k * 2 + 0) * 4);
    luma_type_line<true>(Y + (l + 4) * 20 + 4, 20, has_left, e0, 0, l >> 2);
    if (CF == 1) chroma_type_line<true, 8>(C2[l >> 3] + ((l & 7) + 4) * 12 + 4, 12, has_left, e0, 1 + (l >> 3), (l & 7) >> 1, 0, -1, 4, -1);
    if (CF == 2) {                                           // 16 rows per plane: StrengthIdx = row -> group row >> 2
      chroma_type_line<true, 8>(C2[0] + (l + 4) * 12 + 4, 12, has_left, e0, 1, l >> 2, 0, -1, 4, -1);
      chroma_type_line<true, 8>(C2[1] + (l + 4) * 12 + 4, 12, has_left, e0, 2, l >> 2, 0, -1, 4, -1);
    }
  }
  __syncthreads();
  if (active) {
    const Edges4 e1 = load_edges(D.edges + ((size_t)k * 2 + 1) * 4);
    luma_type_line<false>(Y + 4 * 20 + 4 + l, 20, has_up, e1, 0, l >> 2);
    if (CF == 1) chroma_type_line<false, 8>(C2[l >> 3] + 4 * 12 + 4 + (l & 7), 12, has_up, e1, 1 + (l >> 3), (l & 7) >> 1, 0, -1, 4, -1);
    if (CF == 2) chroma_type_line<false, 16>(C2[l >> 3] + 4 * 12 + 4 + (l & 7), 12, has_up, e1, 1 + (l >> 3), (l & 7) >> 1, 0, 4, 8, 12);
  }
  __syncthreads();
  bool diff = false;
  if (active) {
    // ---- hand on: A (own), R (left margin), T (top margin); compare with what was stored
    uint8_t *o = D.st + (size_t)k * G::REC;
    auto put = [&](uint32_t *g, uint32_t v) { diff = diff || *g != v; *g = v; };
    for (int q = 0; q < 4; q++) put(reinterpret_cast<uint32_t *>(o + l * 16 + q * 4), *reinterpret_cast<const uint32_t *>(Y + (l + 4) * 20 + 4 + q * 4));
    if (has_left) put(reinterpret_cast<uint32_t *>(o + G::A + l * 4), *reinterpret_cast<const uint32_t *>(Y + (l + 4) * 20));              // R.Y
    for (int t = l; t < 2 * CH; t += 16) {
      const int pl = t / (CH ? CH : 1), r = t - pl * CH;
      const uint8_t *cs = C2[pl] + (r + 4) * 12;
      put(reinterpret_cast<uint32_t *>(o + 256 + pl * CH * 8 + r * 8), *reinterpret_cast<const uint32_t *>(cs + 4));
      put(reinterpret_cast<uint32_t *>(o + 256 + pl * CH * 8 + r * 8 + 4), *reinterpret_cast<const uint32_t *>(cs + 8));
      if (has_left) put(reinterpret_cast<uint32_t *>(o + G::A + 64 + pl * CH * 4 + r * 4), *reinterpret_cast<const uint32_t *>(cs));            // R.U / R.V
    }
    if (has_up) {
      const int r = l >> 2, q = l & 3;
      put(reinterpret_cast<uint32_t *>(o + G::A + G::R + r * 16 + q * 4), *reinterpret_cast<const uint32_t *>(Y + r * 20 + 4 + q * 4));   // T.Y
      if (CF) {
        const int pl = l >> 3, cr = (l >> 1) & 3, cq = l & 1;
        put(reinterpret_cast<uint32_t *>(o + G::A + G::R + 64 + pl * 32 + cr * 8 + cq * 4), *reinterpret_cast<const uint32_t *>(C2[pl] + cr * 12 + 4 + cq * 4));
      }
    }
  }
  const unsigned long long m = __ballot(diff);
  if (l == 0 && live) {
    const int changed = ((m >> (16 * h)) & 0xffffull) != 0;
    D.chg_next[k] = (uint8_t)changed;
    if (changed) atomicAdd(D.n_changed, 1);
  }
}

// the filtered picture from the records: A_k, columns 12..15 from R_(k+1), then rows 12..15 from T_(k+mbw)
template <int CF>
__global__ __launch_bounds__(64) void deblock_compose_kernel(const uint8_t *st, uint8_t *y, uint8_t *u, uint8_t *v, int W, int Wc, int mbw, int row0, int rows)
{
  using G = DbrGeo<CF>;
  constexpr int CH = G::CH;
  const int h = threadIdx.x >> 4, l = threadIdx.x & 15, nmb = mbw * rows;
  if ((int)blockIdx.x * 4 + h >= nmb) return;
  const int k = row0 * mbw + blockIdx.x * 4 + h;
  const int mbx = k % mbw, mby = k / mbw;
  const uint8_t *a = st + (size_t)k * G::REC;
  const bool has_r = mbx + 1 < mbw, has_b = mby + 1 < row0 + rows;
  if (mby == row0 && row0 > 0) {                             // the band's top edges changed the bottom rows of the row above: T straight into the picture
    const int r = l >> 2, q = l & 3;
    *reinterpret_cast<uint32_t *>(y + (size_t)(mby * 16 - 4 + r) * W + mbx * 16 + q * 4) = *reinterpret_cast<const uint32_t *>(a + G::A + G::R + r * 16 + q * 4);
    if (CF) {
      const int pl = l >> 3, cr = (l >> 1) & 3, cq = l & 1;
      *reinterpret_cast<uint32_t *>((pl ? v : u) + (size_t)(mby * CH - 4 + cr) * Wc + mbx * 8 + cq * 4) = *reinterpret_cast<const uint32_t *>(a + G::A + G::R + 64 + pl * 32 + cr * 8 + cq * 4);
    }
  }
  {                                                          // luma row l
    uint4 o = *reinterpret_cast<const uint4 *>(a + l * 16);
    if (has_r) o.w = *reinterpret_cast<const uint32_t *>(a + G::REC + G::A + l * 4);
    if (has_b && l >= 12) o = *reinterpret_cast<const uint4 *>(a + (size_t)mbw * G::REC + G::A + G::R + (l - 12) * 16);
    *reinterpret_cast<uint4 *>(y + (size_t)(mby * 16 + l) * W + mbx * 16) = o;
  }
  for (int t = l; t < 2 * CH; t += 16) {                     // chroma rows
    const int pl = t / (CH ? CH : 1), r = t - pl * CH;
    uint2 o = *reinterpret_cast<const uint2 *>(a + 256 + pl * CH * 8 + r * 8);
    if (has_r) o.y = *reinterpret_cast<const uint32_t *>(a + G::REC + G::A + 64 + pl * CH * 4 + r * 4);
    if (has_b && r >= CH - 4) o = *reinterpret_cast<const uint2 *>(a + (size_t)mbw * G::REC + G::A + G::R + 64 + pl * 32 + (r - (CH - 4)) * 8);
    *reinterpret_cast<uint2 *>((pl ? v : u) + (size_t)(mby * CH + r) * Wc + mbx * 8) = o;
  }
}

int ensure_recon(jmhip_ctx *c) { return jm_ensure_recon(c); }

}  // namespace

extern "C" int jmhip_recon_upload(jmhip_ctx *c, const void *Y, const void *U, const void *V, int pel_bytes)
{
  if (!c) return JMHIP_ERR_ARG;
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  int rc = ensure_recon(c);
  if (rc) return rc;
  if ((rc = jm_upload_plane(c, c->rec_y, Y, c->W, c->H, pel_bytes, c->W, 0))) return rc;
  if (c->Wc) {
    if ((rc = jm_upload_plane(c, c->rec_u, U, c->Wc, c->Hc, pel_bytes, c->Wc, 0))) return rc;
    if ((rc = jm_upload_plane(c, c->rec_v, V, c->Wc, c->Hc, pel_bytes, c->Wc, 0))) return rc;
  }
  c->rec_has_pic = true; c->rec_valid = true;
  return JMHIP_OK;
}

namespace {

struct DbkArrays { uint8_t *mbs, *blks, *edges; };

int dbk_arrays(jmhip_ctx *c, DbkArrays *a)
{
  const int nmb = c->mbw * c->mbh;
  const size_t mb_bytes = sizeof(jmhip_deblock_mb) * (size_t)nmb, blk_bytes = sizeof(jmhip_deblock_blk) * (size_t)nmb * 16;
  const size_t blk_off = (mb_bytes + 255) & ~(size_t)255, edge_off = (blk_off + blk_bytes + 255) & ~(size_t)255;
  const size_t total = edge_off + sizeof(EdgeInfo) * (size_t)nmb * 8;
  if (c->dbk_cap < total) {
    if (c->dbk_dev) JM_HIP_CHECK(c, hipFree(c->dbk_dev));
    c->dbk_dev = nullptr; c->dbk_cap = 0;
    if (hipMalloc(&c->dbk_dev, total) != hipSuccess) return jm_fail(c, JMHIP_ERR_NOMEM, "deblocking arrays");
    c->dbk_cap = total;
  }
  a->mbs = (uint8_t *)c->dbk_dev; a->blks = a->mbs + blk_off; a->edges = a->mbs + edge_off;
  return JMHIP_OK;
}

// strengths + the wavefront walk over macroblock rows [mb_row0, mb_row0 + mb_rows); the side arrays are on the device
// `starts`: rows inside (mb_row0, mb_row0 + mb_rows) at which an independent band begins (no macroblock of that row filters its top edge)
int dbk_run(jmhip_ctx *c, const DbkArrays &a, int mvlimit, int mb_row0, int mb_rows, std::vector<int> starts = {})
{
  const int nmb = c->mbw * c->mbh;
  deblock_strength_kernel<<<(nmb * 8 + 255) / 256, 256, 0, c->stream>>>((const jmhip_deblock_mb *)a.mbs, (const jmhip_deblock_blk *)a.blks,
                                                                         (EdgeInfo *)a.edges, c->mbw, c->mbh, mvlimit);
  JM_HIP_CHECK(c, hipGetLastError());
  // relaxation schedule (deblock_relax_kernel): whole 4:2:0 pictures; JMHIP_DEBLOCK_SCHED=wave keeps the wavefront kernels below, which also
  // take the other chroma formats and row bands, and finish if the sweeps have not settled within the cap
  {
    const char *sched = getenv("JMHIP_DEBLOCK_SCHED");
    const int rcf = c->cfg.yuv_format == JMHIP_YUV420 ? 1 : c->cfg.yuv_format == JMHIP_YUV422 ? 2 : c->cfg.yuv_format == JMHIP_YUV400 ? 0 : -1;
    if (rcf >= 0 && mb_rows > 0 && !(sched && !strcmp(sched, "wave"))) {
      const size_t rec_bytes = rcf == 2 ? DbrGeo<2>::REC : rcf == 1 ? DbrGeo<1>::REC : DbrGeo<0>::REC;
      const int nrange = c->mbw * mb_rows;
      const size_t st_bytes = (size_t)nmb * rec_bytes, need = st_bytes + 2 * (size_t)nmb + 64 * sizeof(int);
      if (c->dbr_cap < need) {
        if (c->dbr_dev) JM_HIP_CHECK(c, hipFree(c->dbr_dev));
        c->dbr_dev = nullptr; c->dbr_cap = 0;
        if (hipMalloc(&c->dbr_dev, need) != hipSuccess) return jm_fail(c, JMHIP_ERR_NOMEM, "deblocking records");
        c->dbr_cap = need;
      }
      uint8_t *st = (uint8_t *)c->dbr_dev, *chg = st + st_bytes;
      int *counters = reinterpret_cast<int *>(st + ((st_bytes + 2 * (size_t)nmb + 3) & ~(size_t)3));
      DbkRelax R;
      R.y = c->rec_y; R.u = c->rec_u; R.v = c->rec_v; R.st = st; R.edges = (const EdgeInfo *)a.edges;
      R.W = c->W; R.Wc = c->Wc; R.mbw = c->mbw; R.mbh = c->mbh; R.row0 = mb_row0; R.rows = mb_rows;
      const int cap = getenv("JMHIP_DEBLOCK_SWEEPS") ? atoi(getenv("JMHIP_DEBLOCK_SWEEPS")) : 48, group = 4;     // sweeps per host check
      bool settled = false;
      int sweep = 0;
      while (sweep < cap && !settled) {
        JM_HIP_CHECK(c, hipMemsetAsync(counters, 0, sizeof(int) * group, c->stream));
        for (int g = 0; g < group; g++, sweep++) {
          R.first_sweep = sweep == 0; R.chg_prev = chg + (size_t)(sweep & 1) * nmb; R.chg_next = chg + (size_t)((sweep + 1) & 1) * nmb; R.n_changed = counters + g;
          if (rcf == 2) deblock_relax_kernel<2><<<(nrange + 3) / 4, 64, 0, c->stream>>>(R);
          else if (rcf == 1) deblock_relax_kernel<1><<<(nrange + 3) / 4, 64, 0, c->stream>>>(R);
          else deblock_relax_kernel<0><<<(nrange + 3) / 4, 64, 0, c->stream>>>(R);
        }
        JM_HIP_CHECK(c, hipGetLastError());
        int last[8] = {1, 1, 1, 1, 1, 1, 1, 1};
        JM_HIP_CHECK(c, hipMemcpyAsync(last, counters, sizeof(int) * group, hipMemcpyDeviceToHost, c->stream));
        JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
        settled = last[group - 1] == 0;
        if (getenv("JMHIP_DEBLOCK_TRACE")) fprintf(stderr, "deblock sweeps %d..%d: %d %d %d %d macroblocks changed\n", sweep - group, sweep - 1, last[0], last[1], last[2], last[3]);
      }
      c->dbk_sweeps = sweep;
      if (settled) {
        if (rcf == 2) deblock_compose_kernel<2><<<(nrange + 3) / 4, 64, 0, c->stream>>>(st, c->rec_y, c->rec_u, c->rec_v, c->W, c->Wc, c->mbw, mb_row0, mb_rows);
        else if (rcf == 1) deblock_compose_kernel<1><<<(nrange + 3) / 4, 64, 0, c->stream>>>(st, c->rec_y, c->rec_u, c->rec_v, c->W, c->Wc, c->mbw, mb_row0, mb_rows);
        else deblock_compose_kernel<0><<<(nrange + 3) / 4, 64, 0, c->stream>>>(st, c->rec_y, c->rec_u, c->rec_v, c->W, c->Wc, c->mbw, mb_row0, mb_rows);
        JM_HIP_CHECK(c, hipGetLastError());
        return JMHIP_OK;
      }
    }
  }
  DeblockDev D;
  D.y = c->rec_y; D.u = c->rec_u; D.v = c->rec_v;
  D.edges = (const EdgeInfo *)a.edges;
  D.W = c->W; D.Wc = c->Wc; D.mbw = c->mbw;
  {
    // one workgroup per independent band (slices with disable_idc 2): the bands' wavefronts run side by side on different CUs.
    // JMHIP_DEBLOCK_BANDS=0 keeps one workgroup. More boundaries than workgroups allowed: keep an even subset (any subset is valid).
    const char *nb = getenv("JMHIP_DEBLOCK_BANDS");
    if (nb && atoi(nb) == 0) starts.clear();
    std::vector<int> rows{mb_row0};
    const int ns = (int)starts.size(), keep = std::min(ns, DBK_MAXB - 1);
    for (int i = 0; i < keep; i++) {
      const int r = starts[(size_t)((long long)i * ns / keep)];
      if (r > rows.back() && r < mb_row0 + mb_rows) rows.push_back(r);
    }
    rows.push_back(mb_row0 + mb_rows);
    D.nbands = (int)rows.size() - 1;
    for (int i = 0; i <= DBK_MAXB; i++) D.band_row[i] = rows[std::min<size_t>((size_t)i, rows.size() - 1)];
  }
  D.dbg = getenv("JMHIP_DBK_DEBUG") ? atoi(getenv("JMHIP_DBK_DEBUG")) : 0;
  // LDS ring when three diagonals + one diagonal of edge records fit into 160 KB; JMHIP_DEBLOCK_KERNEL=global forces the global-memory kernel
  // (kept for larger pictures and as a cross-check in the tests)
  int S = 0;
  for (int b = 0; b < D.nbands; b++) {
    const int br = D.band_row[b + 1] - D.band_row[b];
    for (int d = 0; d <= c->mbw - 1 + 2 * (br - 1); d++) {
      const int y_lo = std::max(0, (d - c->mbw + 2) >> 1), y_hi = std::min(br - 1, d >> 1);
      S = std::max(S, y_hi - y_lo + 1);
    }
  }
  const int cf = c->cfg.yuv_format == JMHIP_YUV420 ? 1 : c->cfg.yuv_format == JMHIP_YUV422 ? 2 : c->cfg.yuv_format == JMHIP_YUV444 ? 3 : 0;
  const size_t lds = (size_t)S * (3 * dbk_tile(cf) + DBK_ETILE);
  const char *force = getenv("JMHIP_DEBLOCK_KERNEL");
  const bool use_lds = S <= (cf == 3 ? 64 : 128) && lds <= 160 * 1024 && !(force && !strcmp(force, "global"));
  if (use_lds) {
    auto launch = [&](auto kernel) -> hipError_t {
      hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      // 16 lanes per macroblock of the longest diagonal, in whole waves: short bands (slices) synchronise 2-3 waves per barrier, not 16
      const int threads = S <= 64 ? std::max(64, (S * 16 + 63) / 64 * 64) : 1024;
      kernel<<<D.nbands, threads, lds, c->stream>>>(D, S);
      return hipGetLastError();
    };
    hipError_t e;
    if (S <= 64) e = cf == 3 ? launch(deblock_lds_kernel<1, 3>) : cf == 2 ? launch(deblock_lds_kernel<1, 2>) : cf == 1 ? launch(deblock_lds_kernel<1, 1>) : launch(deblock_lds_kernel<1, 0>);
    else e = cf == 2 ? launch(deblock_lds_kernel<2, 2>) : cf == 1 ? launch(deblock_lds_kernel<2, 1>) : launch(deblock_lds_kernel<2, 0>);
    JM_HIP_CHECK(c, e);
  } else
  switch (c->cfg.yuv_format) {
  case JMHIP_YUV400: deblock_filter_kernel<JMHIP_YUV400><<<D.nbands, 1024, 0, c->stream>>>(D); break;
  case JMHIP_YUV420: deblock_filter_kernel<JMHIP_YUV420><<<D.nbands, 1024, 0, c->stream>>>(D); break;
  case JMHIP_YUV422: deblock_filter_kernel<JMHIP_YUV422><<<D.nbands, 1024, 0, c->stream>>>(D); break;
  case JMHIP_YUV444: deblock_filter_kernel<JMHIP_YUV444><<<D.nbands, 1024, 0, c->stream>>>(D); break;
  default: return jm_fail(c, JMHIP_ERR_ARG, "deblocking: chroma format");
  }
  JM_HIP_CHECK(c, hipGetLastError());
  return JMHIP_OK;
}

// The filter's side information straight from what the frame stage left on the device (P macroblocks: jmhip_me_frame's vectors,
// the modes and coded-block bits of jmhip_residual_frame): what DeblockFrame would read from img->mb_data[] / enc_picture after JM
// had stored the same decisions. 16 lanes per listed macroblock, one per 4x4 block.
__device__ __forceinline__ int dbk_covering_partition(const jmhip_mb_mode &m, int x4, int y4)
{
  const int b8 = 2 * (y4 >> 1) + (x4 >> 1);
  if (m.mode == 1) return 0;
  if (m.mode == 2) return 1 + (y4 >> 1);
  if (m.mode == 3) return 3 + (x4 >> 1);
  const int bm = m.b8mode[b8];
  if (bm == 4) return 5 + b8;
  if (bm == 5) return 9 + 2 * b8 + (y4 & 1);
  if (bm == 6) return 17 + 2 * b8 + (x4 & 1);
  return 25 + 4 * b8 + 2 * (y4 & 1) + (x4 & 1);
}

__global__ __launch_bounds__(256) void deblock_inputs_kernel(const jmhip_me_mb *__restrict__ jobs, const jmhip_me_result *__restrict__ res,
                                                             const jmhip_mb_mode *__restrict__ modes, const JmMbCoded *__restrict__ coded, int n,
                                                             jmhip_deblock_params prm, int mbw, jmhip_deblock_mb *__restrict__ mbs,
                                                             jmhip_deblock_blk *__restrict__ blks)
{
  const int i = blockIdx.x * 16 + (threadIdx.x >> 4), t = threadIdx.x & 15;
  if (i >= n) return;
  const jmhip_me_mb &job = jobs[i];
  const jmhip_mb_mode m = modes[i];
  const int mbx = job.mb_x, mby = job.mb_y, x4 = t & 3, y4 = t >> 2;
  const int p = dbk_covering_partition(m, x4, y4);
  jmhip_deblock_blk b;
  b.mv[0][0] = res[i].mv[p][0]; b.mv[0][1] = res[i].mv[p][1];
  b.mv[1][0] = 0; b.mv[1][1] = 0;
  b.ref_id[0] = job.ref;                                   // one picture per reference slot: slot equality is picture equality
  b.ref_id[1] = INT64_MIN;
  blks[(size_t)(mby * 4 + y4) * (mbw * 4) + mbx * 4 + x4] = b;
  if (t == 0) {
    jmhip_deblock_mb o;
    o.intra = 0;
    o.qp = (uint8_t)prm.qp; o.qpc[0] = (uint8_t)prm.qpc[0]; o.qpc[1] = (uint8_t)prm.qpc[1];
    o.disable_idc = (uint8_t)prm.disable_idc;
    o.alpha_c0_offset = (int8_t)prm.alpha_c0_offset; o.beta_offset = (int8_t)prm.beta_offset;
    o.transform_8x8 = (uint8_t)(m.pad[0] != 0);
    o.avail_a = (uint8_t)(mbx != 0);
    o.avail_b = (uint8_t)(mby != 0 && (prm.slice_rows <= 0 || (mby % prm.slice_rows) != 0));
    o.cbp_blk = (uint16_t)(coded[i].cbp_blk & 0xffff);
    mbs[mby * mbw + mbx] = o;
  }
}

}  // namespace

extern "C" int jmhip_deblock_frame(jmhip_ctx *c, const jmhip_deblock_mb *mbs, const jmhip_deblock_blk *blks, int mvlimit, int mb_row0, int mb_rows)
{
  if (!c || !mbs || !blks) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_deblock_frame: NULL argument") : JMHIP_ERR_ARG;
  if (!c->rec_y || !c->rec_valid) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_deblock_frame: no recon picture yet");
  if (mb_rows <= 0) { mb_row0 = 0; mb_rows = c->mbh; }
  if (mb_row0 < 0 || mb_row0 + mb_rows > c->mbh) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_deblock_frame: row band outside the picture");
  if (mvlimit != 4 && mvlimit != 2) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_deblock_frame: mvlimit is 4 (frame) or 2 (field)");
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  DbkArrays a;
  int rc = dbk_arrays(c, &a);
  if (rc) return rc;
  const int nmb = c->mbw * c->mbh;
  JM_HIP_CHECK(c, hipMemcpyAsync(a.mbs, mbs, sizeof(jmhip_deblock_mb) * (size_t)nmb, hipMemcpyHostToDevice, c->stream));
  JM_HIP_CHECK(c, hipMemcpyAsync(a.blks, blks, sizeof(jmhip_deblock_blk) * (size_t)nmb * 16, hipMemcpyHostToDevice, c->stream));
  JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));             // the caller's arrays are only borrowed for the call
  // rows none of whose macroblocks filter the top edge start an independent band
  std::vector<int> starts;
  for (int r = mb_row0 + 1; r < mb_row0 + mb_rows; r++) {
    bool free_top = true;
    for (int x = 0; x < c->mbw && free_top; x++) {
      const jmhip_deblock_mb &m = mbs[r * c->mbw + x];
      free_top = m.disable_idc == 1 || (m.disable_idc == 2 && !m.avail_b);
    }
    if (free_top) starts.push_back(r);
  }
  jm_stage_begin(c, JMHIP_STAGE_DEBLOCK);
  rc = dbk_run(c, a, mvlimit, mb_row0, mb_rows, starts);
  jm_stage_end(c, JMHIP_STAGE_DEBLOCK);
  return rc;
}

extern "C" int jmhip_deblock_recon(jmhip_ctx *c, const jmhip_deblock_params *prm)
{
  if (!c || !prm) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_deblock_recon: NULL argument") : JMHIP_ERR_ARG;
  if (!c->rec_y || c->fr_n <= 0 || c->fr_n != c->me_n) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_deblock_recon: needs the macroblock list of the last jmhip_me_frame + jmhip_residual_frame");
  int row0 = prm->mb_row0, rows = prm->mb_rows;
  if (rows <= 0) { row0 = 0; rows = c->mbh; }
  if (row0 < 0 || row0 + rows > c->mbh) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_deblock_recon: row band outside the picture");
  const int mvlimit = prm->mvlimit ? prm->mvlimit : 4;
  if (mvlimit != 4 && mvlimit != 2) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_deblock_recon: mvlimit is 4 (frame) or 2 (field)");
  if (prm->qp < 0 || prm->qp > 51 || prm->qpc[0] < 0 || prm->qpc[0] > 51 || prm->qpc[1] < 0 || prm->qpc[1] > 51 || prm->disable_idc < 0 || prm->disable_idc > 2)
    return jm_fail(c, JMHIP_ERR_ARG, "jmhip_deblock_recon: quantiser / idc out of range");
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  DbkArrays a;
  int rc = dbk_arrays(c, &a);
  if (rc) return rc;
  const int nmb = c->mbw * c->mbh, n = c->fr_n;
  jm_stage_begin(c, JMHIP_STAGE_DEBLOCK);
  // macroblocks outside the list (other ranks' rows) read as "P, nothing coded, zero vectors": a band is only self-contained when
  // its first row's top edge is off (idc 2 with slice_rows), exactly as for jmhip_deblock_frame
  JM_HIP_CHECK(c, hipMemsetAsync(a.mbs, 0, (size_t)(a.edges - a.mbs), c->stream));
  (void)nmb;
  const jmhip_mb_mode *modes = (const jmhip_mb_mode *)c->fr_modes;
  const JmMbCoded *coded = reinterpret_cast<const JmMbCoded *>(modes + 2 * (size_t)n);       // layout of jmhip_residual_frame
  deblock_inputs_kernel<<<(n + 15) / 16, 256, 0, c->stream>>>((const jmhip_me_mb *)c->me_jobs_dev, (const jmhip_me_result *)c->me_res_dev, modes, coded, n,
                                                              *prm, c->mbw, (jmhip_deblock_mb *)a.mbs, (jmhip_deblock_blk *)a.blks);
  JM_HIP_CHECK(c, hipGetLastError());
  std::vector<int> starts;
  if (prm->disable_idc == 2 && prm->slice_rows > 0)
    for (int r = (row0 / prm->slice_rows + 1) * prm->slice_rows; r < row0 + rows; r += prm->slice_rows) starts.push_back(r);
  rc = dbk_run(c, a, mvlimit, row0, rows, starts);
  jm_stage_end(c, JMHIP_STAGE_DEBLOCK);
  return rc;
}
