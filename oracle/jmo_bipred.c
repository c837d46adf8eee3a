/*
 * jmo_bipred.c -- ORACLE (test infrastructure): bi-predictive distortion and search.
 * Restates lencod/src/me_distortion.c:482-1040 (computeBiPredSAD1/2, computeBiPredSATD1/2),
 * me_fullsearch.c:164-338 (FullPelBlockMotionBiPred) and :520-745 (SubPelBlockSearchBiPred). Luma only
 * (ChromaMEEnable is forced off by the callers of this file; the chroma term of the reference stays in JM).
 *
 * JM's naming, kept here: "1" is the FIXED block (position s_mv, read from the picture listX[list][ref]), "2" is the
 * SWEPT candidate (position mv, read from listX[list^1][0]).
 */
#include "jmo.h"

static inline int iabs_(int x) { return x < 0 ? -x : x; }
static inline int clip3(int lo, int hi, int x) { return x < lo ? lo : (x > hi ? hi : x); }
static inline int clip1(int hi, int x) { return x < 0 ? 0 : (x > hi ? hi : x); }

/* FastLine4X / UMVLine4X, refbuf.c:25,37 */
static const jmo_pel *line4(const jmo_ref *r, int umv, int y, int x)
{
  int xpos = x >> 2, ypos = y >> 2;
  if (umv) { xpos = clip3(0, r->width_pad, xpos); ypos = clip3(0, r->height_pad, ypos); }
  return r->luma[(y & 3) * 4 + (x & 3)] + (long)ypos * r->Wp + xpos;
}

static inline int bipel(const jmo_bipred *b, int p1, int p2)
{
  if (!b->apply_weights) return (p1 + p2 + 1) >> 1;                                   /* :504 */
  return clip1(b->max_val, ((b->weight1 * p1 + b->weight2 * p2 + 2 * b->wp_luma_round) >> (b->luma_log_weight_denom + 1)) + b->offset_bi);  /* :583 */
}

/* computeBiPredSAD1 :482 / computeBiPredSAD2 :556. Note: the clamp geometry (width_pad) of BOTH pictures is ref1's. */
int jmo_bipred_sad(const jmo_bipred *b, const jmo_pel *src, int bsy, int bsx, int min_mcost,
                   int cand_x1, int cand_y1, int cand_x2, int cand_y2)
{
  const jmo_pel *r2 = line4(b->ref2, b->umv2, cand_y2, cand_x2), *r1 = line4(b->ref1, b->umv1, cand_y1, cand_x1);
  int mcost = 0, y, x;
  for (y = 0; y < bsy; y++) {
    for (x = 0; x < bsx; x++) mcost += iabs_((int)src[x] - bipel(b, r1[x], r2[x]));
    if (mcost >= min_mcost) return mcost;
    src += bsx; r1 += b->ref1->Wp; r2 += b->ref2->Wp;
  }
  return mcost;
}

/* computeBiPredSATD1 :824 / computeBiPredSATD2 :907 */
int jmo_bipred_satd(const jmo_bipred *b, const jmo_pel *src_pic, int bsy, int bsx, int min_mcost,
                    int cand_x1, int cand_y1, int cand_x2, int cand_y2)
{
  const int bs = b->test8x8 ? 8 : 4;
  int diff[64], mcost = 0, y, x, yy, xx;
  const jmo_pel *src_tmp = src_pic;
  for (y = 0; y < (bsy << 2); y += bs << 2) {
    for (x = 0; x < bsx; x += bs) {
      int *dp = diff;
      const jmo_pel *r2 = line4(b->ref2, b->umv2, cand_y2 + y, cand_x2 + (x << 2));
      const jmo_pel *r1 = line4(b->ref1, b->umv1, cand_y1 + y, cand_x1 + (x << 2));
      const jmo_pel *src = src_tmp + x;
      for (yy = 0; yy < bs; yy++) {
        for (xx = 0; xx < bs; xx++) *dp++ = (int)src[xx] - bipel(b, r1[xx], r2[xx]);
        r1 += b->ref1->Wp; r2 += b->ref2->Wp; src += bsx;
      }
      mcost += b->test8x8 ? jmo_hadamard_sad8x8(diff) : jmo_hadamard_sad4x4(diff);
      if (mcost > min_mcost) return mcost;
    }
    src_tmp += bsx * bs;
  }
  return mcost;
}

int jmo_bipred_dist(const jmo_bipred *b, int metric, const jmo_pel *src, int bsy, int bsx, int min_mcost,
                       int x1, int y1, int x2, int y2)
{
  /* computeBiPred1/2[] selection, mv-search.c:403-423: SAD, (SSE not restated), SATD */
  return metric == JMO_ERR_SATD ? jmo_bipred_satd(b, src, bsy, bsx, min_mcost, x1, y1, x2, y2)
                                : jmo_bipred_sad(b, src, bsy, bsx, min_mcost, x1, y1, x2, y2);
}

/* FullPelBlockMotionBiPred, me_fullsearch.c:164-338 */
int jmo_fullpel_bipred(jmo_bipred *b, const jmo_pel *orig_pic, int pic_pix_x, int pic_pix_y, int blocktype,
                       int pred_mv_x1, int pred_mv_y1, int pred_mv_x2, int pred_mv_y2,
                       short *mv_x, short *mv_y, const short *s_mv_x, const short *s_mv_y,
                       int search_range, int min_mcost, int lambda_factor)
{
  const int max_pos = (2 * search_range + 1) * (2 * search_range + 1);
  const int pred_x1 = (pic_pix_x << 2) + pred_mv_x1, pred_y1 = (pic_pix_y << 2) + pred_mv_y1;
  const int pred_x2 = (pic_pix_x << 2) + pred_mv_x2, pred_y2 = (pic_pix_y << 2) + pred_mv_y2;
  const short center_x = (short)(pic_pix_x + *mv_x), center_y = (short)(pic_pix_y + *mv_y);
  const short r1cx = (short)(pic_pix_x + *s_mv_x), r1cy = (short)(pic_pix_y + *s_mv_y);
  const int W = b->ref1->W, H = b->ref1->H;
  int bsx, bsy, pos, best_pos = 0, mcost, cand_x, cand_y;
  short *sx, *sy;
  jmo_block_size(blocktype, &bsx, &bsy);
  sx = (short *)__builtin_alloca(sizeof(short) * (max_pos + 16)); sy = (short *)__builtin_alloca(sizeof(short) * (max_pos + 16));
  jmo_spiral(search_range, sx, sy, max_pos > 9 ? max_pos : 9);
  b->umv2 = !((center_x > search_range) && (center_x < W - 1 - search_range - bsx) &&
              (center_y > search_range) && (center_y < H - 1 - search_range - bsy));          /* :287-296 */
  b->umv1 = !((r1cx > search_range) && (r1cx < W - 1 - search_range - bsx) &&
              (r1cy > search_range) && (r1cy < H - 1 - search_range - bsy));                  /* :299-308 */
  for (pos = 0; pos < max_pos; pos++) {
    cand_x = (center_x + sx[pos]) << 2; cand_y = (center_y + sy[pos]) << 2;
    mcost = jmo_mv_cost(lambda_factor, r1cx << 2, r1cy << 2, pred_x1, pred_y1);
    mcost += jmo_mv_cost(lambda_factor, cand_x, cand_y, pred_x2, pred_y2);
    if (mcost >= min_mcost) continue;
    mcost += jmo_bipred_dist(b, b->metric[JMO_F_PEL], orig_pic, bsy, bsx, min_mcost - mcost,
                         (r1cx << 2) + JMO_PAD4, (r1cy << 2) + JMO_PAD4, cand_x + JMO_PAD4, cand_y + JMO_PAD4);
    if (mcost < min_mcost) { best_pos = pos; min_mcost = mcost; }
  }
  if (best_pos) { *mv_x += sx[best_pos]; *mv_y += sy[best_pos]; }
  return min_mcost;
}

/* SubPelBlockSearchBiPred, me_fullsearch.c:520-745. (mv: swept, quarter-pel in/out; s_mv: fixed, quarter-pel.) */
int jmo_subpel_bipred(jmo_bipred *b, const jmo_pel *orig_pic, int pic_pix_x, int pic_pix_y, int blocktype,
                      int pred_mv_x, int pred_mv_y, short *mv_x, short *mv_y, const short *s_mv_x, const short *s_mv_y,
                      int search_pos2, int search_pos4, int min_mcost, const int *lambda)
{
  static const short s9x[9] = { 0, 0, 0, -1, 1, -1, 1, -1, 1 };
  static const short s9y[9] = { 0, -1, 1, -1, -1, 0, 0, 1, 1 };
  const int start_hp = b->start_hp, start_qp = b->start_qp;
  const int pic4_pix_x = (pic_pix_x + JMO_PAD) << 2, pic4_pix_y = (pic_pix_y + JMO_PAD) << 2;
  const int max_pos2 = (!start_hp ? (search_pos2 > 1 ? search_pos2 : 1) : search_pos2);
  const int smv_x = *s_mv_x + pic4_pix_x, smv_y = *s_mv_y + pic4_pix_y;
  int bsx, bsy, pos, best_pos, mcost, cx, cy, max_pos_x4, max_pos_y4, lambda_factor = lambda[JMO_H_PEL];
  jmo_block_size(blocktype, &bsx, &bsy);
  max_pos_x4 = (b->ref1->W - bsx + 2 * JMO_PAD) << 2; max_pos_y4 = (b->ref1->H - bsy + 2 * JMO_PAD) << 2;

  b->umv2 = !((pic4_pix_x + *mv_x > 1) && (pic4_pix_x + *mv_x < max_pos_x4 - 1) && (pic4_pix_y + *mv_y > 1) && (pic4_pix_y + *mv_y < max_pos_y4 - 1));
  b->umv1 = !((pic4_pix_x + *s_mv_x > 1) && (pic4_pix_x + *s_mv_x < max_pos_x4 - 1) && (pic4_pix_y + *s_mv_y > 1) && (pic4_pix_y + *s_mv_y < max_pos_y4 - 1));
  for (best_pos = 0, pos = start_hp; pos < max_pos2; pos++) {
    cx = *mv_x + (s9x[pos] << 1); cy = *mv_y + (s9y[pos] << 1);
    mcost = jmo_mv_cost(lambda_factor, cx, cy, pred_mv_x, pred_mv_y);
    if (mcost >= min_mcost) continue;
    mcost += jmo_bipred_dist(b, b->metric[JMO_H_PEL], orig_pic, bsy, bsx, min_mcost - mcost, smv_x, smv_y, cx + pic4_pix_x, cy + pic4_pix_y);
    if (mcost < min_mcost) { min_mcost = mcost; best_pos = pos; }
  }
  if (best_pos) { *mv_x += s9x[best_pos] << 1; *mv_y += s9y[best_pos] << 1; }

  b->umv2 = !((pic4_pix_x + *mv_x > 0) && (pic4_pix_x + *mv_x < max_pos_x4) && (pic4_pix_y + *mv_y > 0) && (pic4_pix_y + *mv_y < max_pos_y4));
  b->umv1 = !((pic4_pix_x + *s_mv_x > 0) && (pic4_pix_x + *s_mv_x < max_pos_x4) && (pic4_pix_y + *s_mv_y > 0) && (pic4_pix_y + *s_mv_y < max_pos_y4));
  if (!start_qp) min_mcost = JMO_INT_MAX;                                                     /* :716-717 */
  lambda_factor = lambda[JMO_Q_PEL];
  for (best_pos = 0, pos = start_qp; pos < search_pos4; pos++) {
    cx = *mv_x + s9x[pos]; cy = *mv_y + s9y[pos];
    mcost = jmo_mv_cost(lambda_factor, cx, cy, pred_mv_x, pred_mv_y);
    if (mcost >= min_mcost) continue;
    mcost += jmo_bipred_dist(b, b->metric[JMO_Q_PEL], orig_pic, bsy, bsx, min_mcost - mcost, smv_x, smv_y, cx + pic4_pix_x, cy + pic4_pix_y);
    if (mcost < min_mcost) { min_mcost = mcost; best_pos = pos; }
  }
  if (best_pos) { *mv_x += s9x[best_pos]; *mv_y += s9y[best_pos]; }
  return min_mcost;
}
