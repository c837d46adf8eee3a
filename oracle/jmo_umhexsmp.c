/*
 * jmo_umhexsmp.c -- ORACLE (test infrastructure): the simplified UMHexagonS search (input->SearchMode = 2, UM_HEX_SIMPLE).
 * Restates lencod/src/me_umhexsmp.c of the reference:
 *   smpUMHEXIntegerPelBlockMotionSearch :152    smpUMHEXSubPelBlockMotionSearch :616 (block types > 1)
 *   smpUMHEXFullSubPelBlockMotionSearch :422 (the 16x16 block)    thresholds smpUMHEX_init :101    smpUMHEX_setup :1194 (upper-layer vector)
 * The walker has no state beyond the call: smpUMHEX_l0_cost / smpUMHEX_flag_intra feed smpUMHEX_pred_SAD_uplayer, which nothing reads.
 * Every candidate is a full compute* call with JM's bound (min_mcost - mcost); partial sums of the early exits are never accepted.
 */
#include <string.h>
#include "jmo.h"

static inline int iabs_(int x) { return x < 0 ? -x : x; }
static inline int imax_(int a, int b) { return a > b ? a : b; }

static const short Diamond_X[4] = {-1, 1, 0, 0}, Diamond_Y[4] = {0, 0, -1, 1};
static const short Hexagon_X[6] = {-2, 2, -1, 1, -1, 1}, Hexagon_Y[6] = {0, 0, -2, 2, 2, -2};
static const short Big_Hexagon_X[16] = {-4, 4, 0, 0, -4, 4, -4, 4, -4, 4, -4, 4, -2, 2, -2, 2};
static const short Big_Hexagon_Y[16] = {0, 0, -4, 4, -1, 1, 1, -1, -2, 2, 2, -2, -3, 3, 3, -3};
static const short shift_factor[8] = {0, 0, 1, 1, 2, 3, 3, 1};                  /* block_type_shift_factor :44 */
enum { CrossThr1 = 800, CrossThr2 = 7000, ConvergeThr = 1000, SubPelThr1 = 1000, SubPelThr3 = 400 };   /* smpUMHEX_init :101 */
static const short s9x[9] = {0, 0, 0, -1, 1, -1, 1, -1, 1}, s9y[9] = {0, -1, 1, -1, -1, 0, 0, 1, 1};       /* spiral positions 0..8 */

/* smpUMHEXIntegerPelBlockMotionSearch :152. mv: centre in, result out (pels). up_mv: smpUMHEX_pred_MV_uplayer_X/Y (quarter-pel). */
int jmo_umhexsmp_pel_search(const jmo_me_params *p, const jmo_ref *ref, const jmo_pel *orig_pic, int pic_pix_x, int pic_pix_y, int blocktype,
                            int pred_mv_x, int pred_mv_y, short *mv_x, short *mv_y, int search_range, int min_mcost, int lambda_factor,
                            int up_mv_x, int up_mv_y)
{
  const int pred_x = (pic_pix_x << 2) + pred_mv_x, pred_y = (pic_pix_y << 2) + pred_mv_y;
  const int center_x = pic_pix_x + *mv_x, center_y = pic_pix_y + *mv_y;
  int best_x = 0, best_y = 0, iXMinNow, iYMinNow, cand_x, cand_y, mcost, bsx, bsy, i, m;
  jmo_dist d;
  jmo_block_size(blocktype, &bsx, &bsy);
  jmo_dist_from_params(p, ref, &d);
  d.chroma_me = p->chroma_me ? 1 : 0;
  d.test8x8 = p->transform8x8_mode && blocktype <= 4;
  d.umv = !((center_x > search_range) && (center_x < ref->W - 1 - search_range - bsx) && (center_y > search_range) && (center_y < ref->H - 1 - search_range - bsy));
#define ONE_PIXEL                                                                                                                        \
  if (iabs_(cand_x - center_x) <= search_range && iabs_(cand_y - center_y) <= search_range) {                                            \
    mcost = jmo_mv_cost(lambda_factor, cand_x << 2, cand_y << 2, pred_x, pred_y);                                                        \
    mcost += jmo_uni_pred(p, JMO_F_PEL, &d, orig_pic, bsy, bsx, min_mcost - mcost, (cand_x + JMO_PAD) << 2, (cand_y + JMO_PAD) << 2);    \
    if (mcost < min_mcost) { best_x = cand_x; best_y = cand_y; min_mcost = mcost; }                                                      \
  }
#define DIAMOND for (m = 0; m < 4; m++) { cand_x = iXMinNow + Diamond_X[m]; cand_y = iYMinNow + Diamond_Y[m]; ONE_PIXEL }

  cand_x = center_x; cand_y = center_y;                                         /* the centre: no range test (:241-252) */
  mcost = jmo_mv_cost(lambda_factor, cand_x << 2, cand_y << 2, pred_x, pred_y);
  mcost += jmo_uni_pred(p, JMO_F_PEL, &d, orig_pic, bsy, bsx, min_mcost - mcost, (cand_x << 2) + JMO_PAD4, (cand_y << 2) + JMO_PAD4);
  if (mcost < min_mcost) { min_mcost = mcost; best_x = cand_x; best_y = cand_y; }
  iXMinNow = best_x; iYMinNow = best_y;
  if (pred_mv_x != 0 || pred_mv_y != 0) { cand_x = pic_pix_x; cand_y = pic_pix_y; ONE_PIXEL }
  if (min_mcost < (ConvergeThr >> shift_factor[blocktype])) {                   /* :264-274 */
    DIAMOND
    *mv_x = (short)(best_x - pic_pix_x); *mv_y = (short)(best_y - pic_pix_y);
    return min_mcost;
  }
  DIAMOND
  if ((blocktype == 1 && min_mcost > (CrossThr1 >> shift_factor[blocktype])) || min_mcost > (CrossThr2 >> shift_factor[blocktype])) {
    iXMinNow = best_x; iYMinNow = best_y;
    for (i = 1; i <= search_range / 2; i++) {
      const int step = (i << 1) - 1;
      cand_x = iXMinNow + step; cand_y = iYMinNow; ONE_PIXEL
      cand_x = iXMinNow - step; ONE_PIXEL
      cand_x = iXMinNow; cand_y = iYMinNow + step; ONE_PIXEL
      cand_y = iYMinNow - step; ONE_PIXEL
    }
    iXMinNow = best_x; iYMinNow = best_y;
    for (m = 0; m < 6; m++) { cand_x = iXMinNow + Hexagon_X[m]; cand_y = iYMinNow + Hexagon_Y[m]; ONE_PIXEL }
    iXMinNow = best_x; iYMinNow = best_y;
    for (i = 1; i <= search_range / 4; i++)
      for (m = 0; m < 16; m++) { cand_x = iXMinNow + Big_Hexagon_X[m] * i; cand_y = iYMinNow + Big_Hexagon_Y[m] * i; ONE_PIXEL }
  }
  if (blocktype > 1) { cand_x = pic_pix_x + up_mv_x / 4; cand_y = pic_pix_y + up_mv_y / 4; ONE_PIXEL }
  if (center_x != pic_pix_x || center_y != pic_pix_y) {
    cand_x = pic_pix_x; cand_y = pic_pix_y; ONE_PIXEL
    iXMinNow = best_x; iYMinNow = best_y;
    DIAMOND
  }
  if (min_mcost < (ConvergeThr >> shift_factor[blocktype])) {                   /* :364-376 */
    iXMinNow = best_x; iYMinNow = best_y;
    DIAMOND
    *mv_x = (short)(best_x - pic_pix_x); *mv_y = (short)(best_y - pic_pix_y);
    return min_mcost;
  }
  for (i = 0; i < search_range; i++) {                                          /* extended hexagon :379-393 */
    iXMinNow = best_x; iYMinNow = best_y;
    for (m = 0; m < 6; m++) { cand_x = iXMinNow + Hexagon_X[m]; cand_y = iYMinNow + Hexagon_Y[m]; ONE_PIXEL }
    if (best_x == iXMinNow && best_y == iYMinNow) break;
  }
  for (i = 0; i < search_range; i++) {                                          /* small diamond :396-412 */
    iXMinNow = best_x; iYMinNow = best_y;
    DIAMOND
    if (best_x == iXMinNow && best_y == iYMinNow) break;
  }
#undef DIAMOND
#undef ONE_PIXEL
  *mv_x = (short)(best_x - pic_pix_x); *mv_y = (short)(best_y - pic_pix_y);
  return min_mcost;
}

/* smpUMHEXFullSubPelBlockMotionSearch :422 (block type 1). Both refinements take the QUARTER-pel metric and lambda (dist_method = Q_PEL, :461). */
int jmo_umhexsmp_full_subpel_search(const jmo_me_params *p, const jmo_ref *ref, const jmo_pel *orig_pic, int ref_is_0, int pic_pix_x, int pic_pix_y,
                                    int blocktype, int pred_mv_x, int pred_mv_y, short *mv_x, short *mv_y, int min_mcost, int lambda_factor)
{
  const int start_hp = (p->chroma_me == 1 || p->metric[JMO_F_PEL] != p->metric[JMO_H_PEL]) ? 0 : 1;
  const int start_qp = (p->chroma_me == 1 || p->metric[JMO_H_PEL] != p->metric[JMO_Q_PEL]) ? 0 : 1;
  const int check_position0 = (!p->rdopt && !p->is_b_slice && ref_is_0 && blocktype == 1 && *mv_x == 0 && *mv_y == 0);
  const int pic4_x = (pic_pix_x + JMO_PAD) << 2, pic4_y = (pic_pix_y + JMO_PAD) << 2;
  const int max_pos2 = !start_hp ? imax_(1, 9) : 9;
  int pos, best_pos, mcost, cmx, cmy, bsx, bsy, max_x4, max_y4;
  jmo_dist d;
  jmo_block_size(blocktype, &bsx, &bsy);
  max_x4 = (ref->W - bsx + 2 * JMO_PAD) << 2; max_y4 = (ref->H - bsy + 2 * JMO_PAD) << 2;
  jmo_dist_from_params(p, ref, &d);
  d.chroma_me = (p->chroma_me == 2) ? 1 : 0;
  d.test8x8 = p->transform8x8_mode && blocktype <= 4;
  d.umv = !((pic4_x + *mv_x > 1) && (pic4_x + *mv_x < max_x4 - 1) && (pic4_y + *mv_y > 1) && (pic4_y + *mv_y < max_y4 - 1));
  for (best_pos = 0, pos = start_hp; pos < max_pos2; pos++) {
    cmx = *mv_x + 2 * s9x[pos]; cmy = *mv_y + 2 * s9y[pos];                    /* spiral_hpel_search */
    mcost = jmo_mv_cost(lambda_factor, cmx, cmy, pred_mv_x, pred_mv_y);
    if (mcost >= min_mcost) continue;
    mcost += jmo_uni_pred(p, JMO_Q_PEL, &d, orig_pic, bsy, bsx, min_mcost - mcost, cmx + pic4_x, cmy + pic4_y);
    if (pos == 0 && check_position0) mcost -= (lambda_factor * 16) >> 16;
    if (mcost < min_mcost) { min_mcost = mcost; best_pos = pos; }
    if (min_mcost < (SubPelThr3 >> shift_factor[blocktype])) break;
  }
  if (best_pos) { *mv_x += 2 * s9x[best_pos]; *mv_y += 2 * s9y[best_pos]; }
  if (*mv_x == 0 && *mv_y == 0 && pred_mv_x == 0 && pred_mv_y == 0 && min_mcost < (SubPelThr1 >> shift_factor[blocktype])) return min_mcost;
  if (!start_qp) min_mcost = JMO_INT_MAX;
  d.umv = !((pic4_x + *mv_x > 0) && (pic4_x + *mv_x < max_x4) && (pic4_y + *mv_y > 0) && (pic4_y + *mv_y < max_y4));
  for (best_pos = 0, pos = start_qp; pos < 9; pos++) {
    cmx = *mv_x + s9x[pos]; cmy = *mv_y + s9y[pos];
    mcost = jmo_mv_cost(lambda_factor, cmx, cmy, pred_mv_x, pred_mv_y);
    if (mcost >= min_mcost) continue;
    mcost += jmo_uni_pred(p, JMO_Q_PEL, &d, orig_pic, bsy, bsx, min_mcost - mcost, cmx + pic4_x, cmy + pic4_y);
    if (mcost < min_mcost) { min_mcost = mcost; best_pos = pos; }
    if (min_mcost < (SubPelThr3 >> shift_factor[blocktype])) break;
  }
  if (best_pos) { *mv_x += s9x[best_pos]; *mv_y += s9y[best_pos]; }
  return min_mcost;
}

/* smpUMHEXSubPelBlockMotionSearch :616 (block types > 1): quarter-pel diamond walk inside +-3 quarter-pels of the integer vector */
int jmo_umhexsmp_subpel_search(const jmo_me_params *p, const jmo_ref *ref, const jmo_pel *orig_pic, int pic_pix_x, int pic_pix_y, int blocktype,
                               int pred_mv_x, int pred_mv_y, short *mv_x, short *mv_y, int min_mcost, int lambda_factor, int up_mv_x, int up_mv_y)
{
  const int start_hp = (p->chroma_me == 1 || p->metric[JMO_F_PEL] != p->metric[JMO_H_PEL]) ? 0 : 1;
  const int pic4_x = (pic_pix_x + JMO_PAD) << 2, pic4_y = (pic_pix_y + JMO_PAD) << 2, srd = 3;
  int bsx, bsy, mcost, cx, cy, i, m, currmv_x = 0, currmv_y = 0, iXMinNow, iYMinNow, abort_search;
  int pfx, pfy, pux, puy;
  short max_x4, max_y4;
  unsigned char state[7][7];
  jmo_dist d;
  jmo_block_size(blocktype, &bsx, &bsy);
  max_x4 = (short)((ref->W - bsx + 2 * JMO_PAD) << 2); max_y4 = (short)((ref->H - bsy + 2 * JMO_PAD) << 2);      /* :645-646: short */
  jmo_dist_from_params(p, ref, &d);
  d.chroma_me = (p->chroma_me == 2) ? 1 : 0;
  d.test8x8 = p->transform8x8_mode && blocktype <= 4;
  d.umv = !((pic4_x + *mv_x > 1) && (pic4_x + *mv_x < max_x4 - 1) && (pic4_y + *mv_y > 1) && (pic4_y + *mv_y < max_y4 - 1));
  pfx = (pred_mv_x - *mv_x) % 4; pfy = (pred_mv_y - *mv_y) % 4;
  pux = (up_mv_x - *mv_x) % 4; puy = (up_mv_y - *mv_y) % 4;
  memset(state, 0, sizeof(state));
#define SS(y, x) state[(y) - *mv_y + srd][(x) - *mv_x + srd]
  SS(*mv_y, *mv_x) = 1;
  if (!start_hp) {
    cx = *mv_x; cy = *mv_y;
    mcost = jmo_mv_cost(lambda_factor, cx, cy, pred_mv_x, pred_mv_y);
    mcost += jmo_uni_pred(p, JMO_Q_PEL, &d, orig_pic, bsy, bsx, min_mcost - mcost, cx + pic4_x, cy + pic4_y);
    if (mcost < min_mcost) { min_mcost = mcost; currmv_x = cx; currmv_y = cy; }
  } else { currmv_x = *mv_x; currmv_y = *mv_y; }
  if (*mv_x == 0 && *mv_y == 0 && pfx == 0 && pux == 0 && pfy == 0 && puy == 0 && min_mcost < (SubPelThr1 >> shift_factor[blocktype])) {
    *mv_x = (short)currmv_x; *mv_y = (short)currmv_y;
    return min_mcost;
  }
  if (pfx || pfy) {
    cx = *mv_x + pfx; cy = *mv_y + pfy;
    mcost = jmo_mv_cost(lambda_factor, cx, cy, pred_mv_x, pred_mv_y);
    mcost += jmo_uni_pred(p, JMO_Q_PEL, &d, orig_pic, bsy, bsx, min_mcost - mcost, cx + pic4_x, cy + pic4_y);
    SS(cy, cx) = 1;
    if (mcost < min_mcost) { min_mcost = mcost; currmv_x = cx; currmv_y = cy; }
  }
  for (i = 0; i < srd; i++) {
    abort_search = 1;
    iXMinNow = currmv_x; iYMinNow = currmv_y;
    for (m = 0; m < 4; m++) {
      cx = iXMinNow + Diamond_X[m]; cy = iYMinNow + Diamond_Y[m];
      if (iabs_(cx - *mv_x) <= srd && iabs_(cy - *mv_y) <= srd && !SS(cy, cx)) {
        mcost = jmo_mv_cost(lambda_factor, cx, cy, pred_mv_x, pred_mv_y);
        mcost += jmo_uni_pred(p, JMO_Q_PEL, &d, orig_pic, bsy, bsx, min_mcost - mcost, cx + pic4_x, cy + pic4_y);
        SS(cy, cx) = 1;
        if (mcost < min_mcost) { min_mcost = mcost; currmv_x = cx; currmv_y = cy; abort_search = 0; }
        if (min_mcost < (SubPelThr3 >> shift_factor[blocktype])) { *mv_x = (short)currmv_x; *mv_y = (short)currmv_y; return min_mcost; }
      }
    }
    if (abort_search) break;
  }
#undef SS
  *mv_x = (short)currmv_x; *mv_y = (short)currmv_y;
  return min_mcost;
}
