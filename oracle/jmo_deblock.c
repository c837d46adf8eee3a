/*
 * jmo_deblock.c -- ORACLE (test infrastructure only, see jmo.h): the in-loop deblocking filter of frame pictures.
 *
 * Follows lencod/src/loopFilter.c: DeblockFrame :87 (macroblocks in address order), DeblockMb :128 (vertical edges 0..3,
 * then horizontal edges 0..3; edge 0 only when the neighbour exists / is in the slice for idc 2; chroma on the edges
 * chroma_edge[][][] maps, :53), GetStrengthNormal :263, EdgeLoopLumaNormal :529, EdgeLoopChromaNormal :815.
 * Frame pictures without MBAFF only (mixedModeEdgeFlag, field strengths and SP/SI slices are not restated).
 */
#include "jmo.h"
#include <stdlib.h>

/* ALPHA_TABLE / BETA_TABLE / CLIP_TAB of loopFilter.c:40-52 are the standard's Table 8-16 / 8-17; generated here from the
 * standard's printed rows rather than transcribed as an array of arrays */
static const unsigned char k_alpha[52] = {
  0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28, 32, 36, 40, 45,
  50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255};
static const unsigned char k_beta[52] = {
  0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10,
  11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18};
/* tc0 for bS 1, 2, 3 (bS 4 uses the bS 3 column, as JM's fifth CLIP_TAB column repeats the fourth) */
static const unsigned char k_tc0[3][52] = {
  {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6, 6, 7, 8, 9, 10, 11, 13},
  {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5, 5, 6, 7, 8, 8, 10, 11, 12, 13, 15, 17},
  {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 23, 25}};

static int iabs_(int v) { return v < 0 ? -v : v; }
static int clip3(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }

typedef struct {
  const jmo_deblock_mb *mbs;
  const jmo_deblock_blk *blks;
  int mbw, mbh, mvlimit, scale, max_val;
} dbk;

/* GetStrengthNormal :263 -- the 16 strengths of one edge; (px, py) = first sample of the q side inside the macroblock */
static void strengths(const dbk *d, int mbx, int mby, int dir, int edge, unsigned char bs[16])
{
  const jmo_deblock_mb *q = &d->mbs[mby * d->mbw + mbx];
  /* p side: same macroblock for inner edges, left / upper neighbour for edge 0 (getNeighbour with xQ = edge-1, :298) */
  int pmbx = mbx, pmby = mby;
  if (edge == 0) { if (dir) pmby--; else pmbx--; }
  const jmo_deblock_mb *p = &d->mbs[pmby * d->mbw + pmbx];
  if (p->intra || q->intra) {
    for (int i = 0; i < 16; i++) bs[i] = (unsigned char)(edge == 0 ? 4 : 3);       /* :399 (frame pictures) */
    return;
  }
  const int w4 = d->mbw * 4;
  for (int idx = 0; idx < 16; idx += 4) {
    /* 4x4 block coordinates inside the macroblocks */
    const int qbx = dir ? (idx >> 2) : (edge >> 2), qby = dir ? (edge >> 2) : (idx >> 2);
    const int pbx = dir ? qbx : ((edge >> 2) + 3) & 3, pby = dir ? ((edge >> 2) + 3) & 3 : qby;
    int v;
    if (((q->cbp_blk >> (qby * 4 + qbx)) & 1) || ((p->cbp_blk >> (pby * 4 + pbx)) & 1)) v = 2;      /* :325 */
    else {
      const jmo_deblock_blk *a = &d->blks[(mby * 4 + qby) * w4 + mbx * 4 + qbx];       /* "p" in JM's naming: the q-side block */
      const jmo_deblock_blk *b = &d->blks[(pmby * 4 + pby) * w4 + pmbx * 4 + pbx];
      const long long a0 = a->ref_id[0], a1 = a->ref_id[1], b0 = b->ref_id[0], b1 = b->ref_id[1];
      const int lim = d->mvlimit;
#define FAR(l, m) ((iabs_(a->mv[l][0] - b->mv[m][0]) >= 4) | (iabs_(a->mv[l][1] - b->mv[m][1]) >= lim))
      if ((a0 == b0 && a1 == b1) || (a0 == b1 && a1 == b0)) {
        if (a0 != a1) v = (a0 == b0) ? (FAR(0, 0) | FAR(1, 1)) : (FAR(0, 1) | FAR(1, 0));      /* :346-364 */
        else v = (FAR(0, 0) | FAR(1, 1)) && (FAR(0, 1) | FAR(1, 0));                          /* :369-379 */
      } else v = 1;
#undef FAR
    }
    bs[idx] = bs[idx + 1] = bs[idx + 2] = bs[idx + 3] = (unsigned char)v;
  }
}

/* one line of samples across a luma-type edge (EdgeLoopLumaNormal :583-661); q points at the first q sample, inc = distance between
 * neighbouring samples across the edge */
static void luma_line(jmo_pel *q, int inc, int bS, int alpha, int beta, int tc0, int max_val)
{
  jmo_pel *p = q - inc;
  const int L3 = p[-3 * inc], L2 = p[-2 * inc], L1 = p[-inc], L0 = p[0], R0 = q[0], R1 = q[inc], R2 = q[2 * inc], R3 = q[3 * inc];
  const int delta = R0 - L0, ad = iabs_(delta);
  if (ad >= alpha || iabs_(R0 - R1) >= beta || iabs_(L0 - L1) >= beta) return;
  if (bS == 4) {
    const int small_gap = ad < ((alpha >> 2) + 2);
    const int aq = (iabs_(R0 - R2) < beta) & small_gap, ap = (iabs_(L0 - L2) < beta) & small_gap, RL0 = L0 + R0;
    if (ap) {
      p[-2 * inc] = (jmo_pel)((((L3 + L2) << 1) + L2 + L1 + RL0 + 4) >> 3);
      p[-inc] = (jmo_pel)((L2 + L1 + L0 + R0 + 2) >> 2);
      p[0] = (jmo_pel)((R1 + ((L1 + RL0) << 1) + L2 + 4) >> 3);
    } else p[0] = (jmo_pel)(((L1 << 1) + L0 + R1 + 2) >> 2);
    if (aq) {
      q[0] = (jmo_pel)((L1 + ((R1 + RL0) << 1) + R2 + 4) >> 3);
      q[inc] = (jmo_pel)((R2 + R0 + R1 + L0 + 2) >> 2);
      q[2 * inc] = (jmo_pel)((((R3 + R2) << 1) + R2 + R1 + RL0 + 4) >> 3);
    } else q[0] = (jmo_pel)(((R1 << 1) + R0 + L1 + 2) >> 2);
  } else {
    const int RL0 = (L0 + R0 + 1) >> 1, aq = iabs_(R0 - R2) < beta, ap = iabs_(L0 - L2) < beta, c0 = tc0 + ap + aq;
    const int dif = clip3(-c0, c0, ((delta << 2) + (L1 - R1) + 4) >> 3);
    if (ap) p[-inc] = (jmo_pel)(L1 + clip3(-tc0, tc0, (L2 + RL0 - (L1 << 1)) >> 1));
    p[0] = (jmo_pel)clip3(0, max_val, L0 + dif);
    q[0] = (jmo_pel)clip3(0, max_val, R0 - dif);
    if (aq) q[inc] = (jmo_pel)(R1 + clip3(-tc0, tc0, (R2 + RL0 - (R1 << 1)) >> 1));
  }
}

/* EdgeLoopChromaNormal :880-905 */
static void chroma_line(jmo_pel *q, int inc, int bS, int alpha, int beta, int tc0, int max_val)
{
  jmo_pel *p = q - inc;
  const int L1 = p[-inc], L0 = p[0], R0 = q[0], R1 = q[inc];
  const int delta = R0 - L0;
  if (iabs_(delta) >= alpha || iabs_(R0 - R1) >= beta || iabs_(L0 - L1) >= beta) return;
  if (bS == 4) {
    q[0] = (jmo_pel)(((R1 << 1) + R0 + L1 + 2) >> 2);
    p[0] = (jmo_pel)(((L1 << 1) + L0 + R1 + 2) >> 2);
  } else {
    const int c0 = tc0 + 1, dif = clip3(-c0, c0, ((delta << 2) + (L1 - R1) + 4) >> 3);
    p[0] = (jmo_pel)clip3(0, max_val, L0 + dif);
    q[0] = (jmo_pel)clip3(0, max_val, R0 - dif);
  }
}

/* alpha / beta / tc0 row for an edge between macroblocks p and q of plane pl (0 luma, 1/2 chroma): :566-575, :851-860 */
static void edge_params(const dbk *d, const jmo_deblock_mb *p, const jmo_deblock_mb *q, int pl, int *alpha, int *beta, int *ia)
{
  const int qp = pl ? (p->qpc[pl - 1] + q->qpc[pl - 1] + 1) >> 1 : (p->qp + q->qp + 1) >> 1;
  *ia = clip3(0, 51, qp + q->alpha_c0_offset);
  const int ib = clip3(0, 51, qp + q->beta_offset);
  *alpha = k_alpha[*ia] * d->scale;
  *beta = k_beta[ib] * d->scale;
}

void jmo_deblock_frame(jmo_pel *Y, jmo_pel *U, jmo_pel *V, int W, int H, int yuv_format, int bit_depth,
                       const jmo_deblock_mb *mbs, const jmo_deblock_blk *blks, int mvlimit)
{
  static const signed char chroma_edge[2][4][4] = {      /* loopFilter.c:53-63: luma edge -> chroma sample offset or < 0 (none) */
    {{-4, 0, 0, 0}, {-4, -4, -4, 4}, {-4, 4, 4, 8}, {-4, -4, -4, 12}},
    {{-4, 0, 0, 0}, {-4, -4, 4, 4}, {-4, 4, 8, 8}, {-4, -4, 12, 12}}};
  static const int pelnum_cr[2][4] = {{0, 8, 16, 16}, {0, 8, 8, 16}};
  dbk d = {mbs, blks, W / 16, H / 16, mvlimit, 1 << (bit_depth - 8), (1 << bit_depth) - 1};
  const int Wc = yuv_format == JMO_YUV420 || yuv_format == JMO_YUV422 ? W / 2 : W;
  const int cw = yuv_format == JMO_YUV444 ? 16 : 8, ch = yuv_format == JMO_YUV420 ? 8 : 16;
  jmo_pel *planes[3] = {Y, U, V};
  for (int mby = 0; mby < d.mbh; mby++)
    for (int mbx = 0; mbx < d.mbw; mbx++) {
      const jmo_deblock_mb *q = &mbs[mby * d.mbw + mbx];
      if (q->disable_idc == 1) continue;
      int left = mbx != 0, top = mby != 0;
      if (q->disable_idc == 2) { left = q->avail_a; top = q->avail_b; }
      for (int dir = 0; dir < 2; dir++)
        for (int edge = 0; edge < 4; edge++) {
          if (!edge && !(dir ? top : left)) continue;
          unsigned char bs[16];
          strengths(&d, mbx, mby, dir, edge * 4, bs);
          int any = 0;
          for (int i = 0; i < 16; i++) any |= bs[i];
          if (!any) continue;
          const jmo_deblock_mb *p = q;
          if (!edge) p = dir ? q - d.mbw : q - 1;
          int alpha, beta, ia;
          if (!(q->transform_8x8 && (edge & 1))) {                 /* filterNon8x8LumaEdgesFlag :153 */
            const int npl = yuv_format == JMO_YUV444 ? 3 : 1;       /* 4:4:4: the luma filter on all three planes, :204-208 */
            for (int pl = 0; pl < npl; pl++) {
              edge_params(&d, p, q, pl, &alpha, &beta, &ia);
              for (int pel = 0; pel < 16; pel++) {
                if (!bs[pel]) continue;
                jmo_pel *s = dir ? planes[pl] + (long)(mby * 16 + edge * 4) * W + mbx * 16 + pel
                                 : planes[pl] + (long)(mby * 16 + pel) * W + mbx * 16 + edge * 4;
                luma_line(s, dir ? W : 1, bs[pel], alpha, beta, k_tc0[bs[pel] > 3 ? 2 : bs[pel] - 1][ia] * d.scale, d.max_val);
              }
            }
          }
          if ((yuv_format == JMO_YUV420 || yuv_format == JMO_YUV422) && U) {
            const int ec = chroma_edge[dir][edge][yuv_format];
            if (ec < 0) continue;
            const int n = pelnum_cr[dir][yuv_format];
            for (int uv = 0; uv < 2; uv++) {
              edge_params(&d, p, q, 1 + uv, &alpha, &beta, &ia);
              for (int pel = 0; pel < n; pel++) {
                const int b = bs[n == 8 ? ((pel >> 1) << 2) + (pel & 1) : pel];       /* StrengthIdx :874 */
                if (!b) continue;
                jmo_pel *s = dir ? planes[1 + uv] + (long)(mby * ch + ec) * Wc + mbx * cw + pel
                                 : planes[1 + uv] + (long)(mby * ch + pel) * Wc + mbx * cw + ec;
                chroma_line(s, dir ? Wc : 1, b, alpha, beta, k_tc0[b > 3 ? 2 : b - 1][ia] * d.scale, d.max_val);
              }
            }
          }
        }
    }
}
