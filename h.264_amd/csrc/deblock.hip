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

struct PlaneParams { uint8_t alpha, beta, tc[3]; };          // tc[bS-1]; bS 4 never reads it for luma and uses tc[2] nowhere
struct EdgeInfo {                                            // 24 bytes per (macroblock, direction, edge)
  uint8_t bs[4];
  PlaneParams pl[3];
  uint8_t on;                                                // bit 0: some strength is non-zero and the edge is filtered; bit 1: luma samples too
  uint8_t pad[4];
};
static_assert(sizeof(EdgeInfo) == 24, "EdgeInfo layout");

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
  e.on = 0;
  for (int i = 0; i < 4; i++) { e.bs[i] = 0; e.pad[i] = 0; }
  for (int p = 0; p < 3; p++) { e.pl[p].alpha = e.pl[p].beta = 0; e.pl[p].tc[0] = e.pl[p].tc[1] = e.pl[p].tc[2] = 0; }
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
      for (int i = 0; i < 4; i++) e.bs[i] = (uint8_t)(edge == 0 ? 4 : 3);           // :399
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
        e.bs[g] = (uint8_t)v;
        any |= v;
      }
    }
    if (any) {
      e.on = (uint8_t)(1 | ((q.transform_8x8 && (edge & 1)) ? 0 : 2));              // filterNon8x8LumaEdgesFlag :153
      for (int pl = 0; pl < 3; pl++) {
        const int qp = pl ? (p.qpc[pl - 1] + q.qpc[pl - 1] + 1) >> 1 : (p.qp + q.qp + 1) >> 1;        // :566 / :851
        const int ia = clip3(0, 51, qp + q.alpha_c0_offset), ib = clip3(0, 51, qp + q.beta_offset);
        e.pl[pl].alpha = c_alpha[ia];
        e.pl[pl].beta = c_beta[ib];
        for (int k = 0; k < 3; k++) e.pl[pl].tc[k] = c_tc0[k][ia];
      }
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

__device__ __forceinline__ int tc_of(const PlaneParams &p, int bS) { return bS ? p.tc[bS > 3 ? 2 : bS - 1] : 0; }

// a luma-type line of 16 samples of this macroblock + the 4 before it; `step` = distance in bytes between samples of the line
// (1: a row, vertical edges; pitch: a column, horizontal edges). g = the line's strength group (line >> 2).
template <bool ROW>
__device__ __forceinline__ void luma_type_line(uint8_t *base, int pitch, bool has_before, const EdgeInfo *ei, int pl, int g)
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
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const EdgeInfo &E = ei[e];
    if ((E.on & 3) == 3) {
      const int bS = E.bs[g];
      luma_line(&s[4 + 4 * e], bS, E.pl[pl].alpha, E.pl[pl].beta, tc_of(E.pl[pl], bS));
    }
  }
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
__device__ __forceinline__ void chroma_type_line(uint8_t *base, int pitch, bool has_before, const EdgeInfo *ei, int pl, int g,
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
  const int off[4] = {off0, off1, off2, off3};
#pragma unroll
  for (int e = 0; e < 4; e++) {
    if (off[e] < 0) continue;
    const EdgeInfo &E = ei[e];
    if (E.on & 1) {
      const int bS = E.bs[g];
      chroma_line(&s[4 + off[e]], bS, E.pl[pl].alpha, E.pl[pl].beta, tc_of(E.pl[pl], bS));
    }
  }
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

struct DeblockDev {
  uint8_t *y, *u, *v;
  const EdgeInfo *edges;        // [mb][dir][edge]
  int W, Wc, mbw, row0, rows;
};

// FMT: JMHIP_YUV400 / 420 / 422 / 444
template <int FMT>
__global__ __launch_bounds__(1024) void deblock_filter_kernel(DeblockDev D)
{
  const int slot = threadIdx.x >> 4, l = threadIdx.x & 15;
  const int last_d = D.mbw - 1 + 2 * (D.rows - 1);
  for (int d = 0; d <= last_d; d++) {
    const int y_lo = max(0, (d - D.mbw + 2) >> 1), y_hi = min(D.rows - 1, d >> 1);
    const int count = y_hi - y_lo + 1;
    for (int dir = 0; dir < 2; dir++) {
      for (int k = slot; k < count; k += 64) {
        const int mby = D.row0 + y_lo + k, mbx = d - 2 * (y_lo + k);
        const EdgeInfo *ei = D.edges + ((size_t)(mby * D.mbw + mbx) * 2 + dir) * 4;
        const bool before = dir ? mby != 0 : mbx != 0;
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

int ensure_recon(jmhip_ctx *c)
{
  if (c->rec_y) return JMHIP_OK;
  if (hipMalloc((void **)&c->rec_y, (size_t)c->W * c->H) != hipSuccess) return jm_fail(c, JMHIP_ERR_NOMEM, "recon picture");
  if (c->Wc && (hipMalloc((void **)&c->rec_u, (size_t)c->Wc * c->Hc) != hipSuccess || hipMalloc((void **)&c->rec_v, (size_t)c->Wc * c->Hc) != hipSuccess))
    return jm_fail(c, JMHIP_ERR_NOMEM, "recon picture");
  return JMHIP_OK;
}

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
  c->rec_has_pic = true;
  return JMHIP_OK;
}

extern "C" int jmhip_deblock_frame(jmhip_ctx *c, const jmhip_deblock_mb *mbs, const jmhip_deblock_blk *blks, int mvlimit, int mb_row0, int mb_rows)
{
  if (!c || !mbs || !blks) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_deblock_frame: NULL argument") : JMHIP_ERR_ARG;
  if (!c->rec_y || !(c->fr_n > 0 || c->rec_has_pic)) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_deblock_frame: no recon picture yet");
  if (mb_rows <= 0) { mb_row0 = 0; mb_rows = c->mbh; }
  if (mb_row0 < 0 || mb_row0 + mb_rows > c->mbh) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_deblock_frame: row band outside the picture");
  if (mvlimit != 4 && mvlimit != 2) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_deblock_frame: mvlimit is 4 (frame) or 2 (field)");
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
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
  uint8_t *base = (uint8_t *)c->dbk_dev;
  JM_HIP_CHECK(c, hipMemcpyAsync(base, mbs, mb_bytes, hipMemcpyHostToDevice, c->stream));
  JM_HIP_CHECK(c, hipMemcpyAsync(base + blk_off, blks, blk_bytes, hipMemcpyHostToDevice, c->stream));
  JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));             // the caller's arrays are only borrowed for the call
  jm_stage_begin(c, JMHIP_STAGE_DEBLOCK);
  deblock_strength_kernel<<<(nmb * 8 + 255) / 256, 256, 0, c->stream>>>((const jmhip_deblock_mb *)base, (const jmhip_deblock_blk *)(base + blk_off),
                                                                         (EdgeInfo *)(base + edge_off), c->mbw, c->mbh, mvlimit);
  JM_HIP_CHECK(c, hipGetLastError());
  DeblockDev D;
  D.y = c->rec_y; D.u = c->rec_u; D.v = c->rec_v;
  D.edges = (const EdgeInfo *)(base + edge_off);
  D.W = c->W; D.Wc = c->Wc; D.mbw = c->mbw; D.row0 = mb_row0; D.rows = mb_rows;
  switch (c->cfg.yuv_format) {
  case JMHIP_YUV400: deblock_filter_kernel<JMHIP_YUV400><<<1, 1024, 0, c->stream>>>(D); break;
  case JMHIP_YUV420: deblock_filter_kernel<JMHIP_YUV420><<<1, 1024, 0, c->stream>>>(D); break;
  case JMHIP_YUV422: deblock_filter_kernel<JMHIP_YUV422><<<1, 1024, 0, c->stream>>>(D); break;
  case JMHIP_YUV444: deblock_filter_kernel<JMHIP_YUV444><<<1, 1024, 0, c->stream>>>(D); break;
  default: return jm_fail(c, JMHIP_ERR_ARG, "jmhip_deblock_frame: chroma format");
  }
  JM_HIP_CHECK(c, hipGetLastError());
  jm_stage_end(c, JMHIP_STAGE_DEBLOCK);
  return JMHIP_OK;
}
