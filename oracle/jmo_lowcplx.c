/*
 * jmo_lowcplx.c -- ORACLE (test infrastructure): the inter decision of a P slice in low-complexity mode
 * (RDOptimization = 0) with intra modes off in inter slices (DisableIntraInInter = 1) and the 4x4 transform, as a
 * whole-slice driver: for every macroblock in raster order the chain
 *   encode_one_macroblock_low        lencod/src/md_low.c:46     (modes 1..3 :112-188, P8x8 :190-330, final parameters :543-636)
 *   PartitionMotionSearch            lencod/src/mv-search.c:1378
 *   BlockMotionSearch                lencod/src/mv-search.c:560 (predictor, search-mode dispatch, sub-pel dispatch, skip shortcut :829-849)
 *   SetMotionVectorPredictor         lencod/src/mv-search.c:87   (neighbours: getLuma4x4Neighbour, lencod/src/mb_access.c)
 *   list_prediction_cost             lencod/src/mode_decision.c:255
 *   submacroblock_mode_decision      lencod/src/mode_decision.c:530 (the rdopt = 0 path)
 *   FindSkipModeMotionVector / GetSkipCostMB  lencod/src/mv-search.c:1189 / :1136
 *   SetRefAndMotionVectors :2777, assign_enc_picture_params :3505, SetModesAndRefframeForBlocks :1262, SetMotionVectorsMB :1845 (rdopt.c)
 * is restated literally, INCLUDING what the picture-level vector / reference arrays hold between the steps (they feed the
 * predictors of the next partition: during a mode's reference loop they hold the reference being searched, after block 0 of modes
 * 2 / 3 and after every 8x8 block of P8x8 the winner, after block 1 of modes 2 / 3 the last reference searched).
 * Search modes: -1 FullSearch, 0 FastFullSearch, 1 UMHexagonS, 2 simplified UMHexagonS, 3 EPZS. Luma only (ChromaMEEnable 0), frame pictures.
 */
#include <stdlib.h>
#include <string.h>
#include "jmo.h"

static inline int imin_(int a, int b) { return a < b ? a : b; }
static inline int imax_(int a, int b) { return a > b ? a : b; }
static inline int clip3(int lo, int hi, int x) { return x < lo ? lo : (x > hi ? hi : x); }

typedef struct {
  const jmo_lowcplx_params *q;
  const jmo_ref *refs;
  const jmo_pel *cur; int cur_stride;
  int mbw, mbh, w4, h4;
  signed char *ref_idx;            /* enc_picture->ref_idx[LIST_0] [h4][w4] */
  short *mv;                       /* enc_picture->mv[LIST_0]      [h4][w4][2] */
  /* per macroblock */
  int mb_x, mb_y, mb_nr;
  short all_mv[4][4][JMO_MAX_REFS][9][2];     /* img->all_mv[by][bx][LIST_0][ref][blocktype] */
  short pred_mv[4][4][JMO_MAX_REFS][9][2];    /* img->pred_mv */
  int motion_cost[8][JMO_MAX_REFS][4];        /* motion_cost[blocktype][LIST_0][ref][block8x8] */
  jmo_fastfull ff[JMO_MAX_REFS]; int ff_done[JMO_MAX_REFS];
  jmo_mb_inter *out;
  int pass8ts;                     /* inside the 8x8-transform P8x8 pass: the call records go to the *8ts arrays */
} lc_ctx;

#define REFIDX(c, y, x) (c)->ref_idx[(size_t)(y) * (c)->w4 + (x)]
#define MVAT(c, y, x)   ((c)->mv + ((size_t)(y) * (c)->w4 + (x)) * 2)

/* getLuma4x4Neighbour for frame pictures (mb_access.c: getNonAffNeighbour + CheckAvailabilityOfNeighbors): luma offsets
 * (xN, yN) relative to the current macroblock -> availability and the 4x4 position in the picture */
static int nbr(const lc_ctx *c, int xN, int yN, int *px, int *py)
{
  int mbx = c->mb_x, mby = c->mb_y;
  if (xN < 0 && yN < 0) { mbx--; mby--; }
  else if (xN < 0 && yN < 16) mbx--;
  else if (xN >= 0 && xN < 16 && yN < 0) mby--;
  else if (xN >= 0 && xN < 16 && yN >= 0 && yN < 16) { /* current */ }
  else if (xN >= 16 && yN < 0) { mbx++; mby--; }
  else return 0;
  if (mbx < 0 || mby < 0 || mbx >= c->mbw || mby >= c->mbh) return 0;
  if (c->q->slice_id && c->q->slice_id[mby * c->mbw + mbx] != c->q->slice_id[c->mb_nr]) return 0;
  *px = (mbx * 16 + ((xN + 16) & 15)) >> 2; *py = (mby * 16 + ((yN + 16) & 15)) >> 2;
  return 1;
}

static void jmo_lc_neighbours(const lc_ctx *c, int mb_x, int mb_y, int bsx, jmo_umhex_nbr *nb)
{
  static const int dx[4] = {-1, 0, 0, -1}, dy[4] = {0, -1, -1, -1};
  int k;
  memset(nb, 0, sizeof(*nb));
  for (k = 0; k < 4; k++) {
    const int xN = mb_x + dx[k] + (k == 2 ? bsx : 0), yN = mb_y + dy[k];
    nb->available[k] = nbr(c, xN, yN, &nb->pos_x[k], &nb->pos_y[k]);
    if (nb->available[k]) {
      nb->ref[k] = REFIDX(c, nb->pos_y[k], nb->pos_x[k]);
      nb->mv[k][0] = MVAT(c, nb->pos_y[k], nb->pos_x[k])[0]; nb->mv[k][1] = MVAT(c, nb->pos_y[k], nb->pos_x[k])[1];
    }
  }
}

/* SetMotionVectorPredictor, mv-search.c:87 (frame pictures) */
static void set_mv_predictor(const lc_ctx *c, short pmv[2], int ref_frame, int block_x, int block_y, int bsx, int bsy)
{
  const int mb_x = 4 * block_x, mb_y = 4 * block_y;
  jmo_umhex_nbr nb;
  int rL, rU, rUR, type = 0, hv;
  enum { A, B, C, D };
  jmo_lc_neighbours(c, mb_x, mb_y, bsx, &nb);
  if (mb_y > 0) {
    if (mb_x < 8) {
      if (mb_y == 8) { if (bsx == 16) nb.available[C] = 0; }
      else if (mb_x + bsx == 8) nb.available[C] = 0;
    } else if (mb_x + bsx == 16) nb.available[C] = 0;
  }
  if (!nb.available[C]) { nb.available[C] = nb.available[D]; nb.ref[C] = nb.ref[D]; nb.mv[C][0] = nb.mv[D][0]; nb.mv[C][1] = nb.mv[D][1]; }
  rL = nb.available[A] ? nb.ref[A] : -1; rU = nb.available[B] ? nb.ref[B] : -1; rUR = nb.available[C] ? nb.ref[C] : -1;
  if (rL == ref_frame && rU != ref_frame && rUR != ref_frame) type = 1;
  else if (rL != ref_frame && rU == ref_frame && rUR != ref_frame) type = 2;
  else if (rL != ref_frame && rU != ref_frame && rUR == ref_frame) type = 3;
  if (bsx == 8 && bsy == 16) { if (mb_x == 0) { if (rL == ref_frame) type = 1; } else if (rUR == ref_frame) type = 3; }
  else if (bsx == 16 && bsy == 8) { if (mb_y == 0) { if (rU == ref_frame) type = 2; } else if (rL == ref_frame) type = 1; }
  for (hv = 0; hv < 2; hv++) {
    const int a = nb.available[A] ? nb.mv[A][hv] : 0, b = nb.available[B] ? nb.mv[B][hv] : 0, cc = nb.available[C] ? nb.mv[C][hv] : 0;
    int pv;
    switch (type) {
    case 0: pv = !(nb.available[B] || nb.available[C]) ? a : a + b + cc - imin_(a, imin_(b, cc)) - imax_(a, imax_(b, cc)); break;
    case 1: pv = a; break;
    case 2: pv = b; break;
    default: pv = cc; break;
    }
    pmv[hv] = (short)pv;
  }
}

static void me_params_for_ref(const lc_ctx *c, int ref, jmo_me_params *p)
{
  *p = c->q->me;
  if (p->apply_weights) { p->weight_luma = c->q->wp_weight[ref]; p->offset_luma = c->q->wp_offset[ref]; }
}

/* LumaPrediction (macroblock.c:836) of the whole macroblock for a per-4x4 assignment of list-0 reference and vector: one fetch per bs x bs
 * block (bs = 4, or 8 for the 8x8-transform residual of LumaResidualCoding8x8 :1142) with its origin clamped (UMV), explicit weights when
 * the picture parameter set has them */
static void predict_mb(const lc_ctx *c, const int ref4[16], const short (*mv4)[2], jmo_pel mpr[16][16], int bs)
{
  const jmo_lowcplx_params *q = c->q;
  int bx, by, x, y;
  for (by = 0; by < 16; by += bs) for (bx = 0; bx < 16; bx += bs) {
    const int b = (by >> 2) * 4 + (bx >> 2), r = ref4[b];
    const jmo_ref *rp = &c->refs[r];
    const int xq = ((c->mb_x * 16 + bx) << 2) + JMO_PAD4 + mv4[b][0], yq = ((c->mb_y * 16 + by) << 2) + JMO_PAD4 + mv4[b][1];
    const int xpos = clip3(0, rp->width_pad, xq >> 2), ypos = clip3(0, rp->height_pad, yq >> 2);
    const jmo_pel *src = rp->luma[(yq & 3) * 4 + (xq & 3)] + (long)ypos * rp->Wp + xpos;
    for (y = 0; y < bs; y++) for (x = 0; x < bs; x++) {
      int v = src[(long)y * rp->Wp + x];
      if (q->wp_pred) v = clip3(0, q->me.max_val, ((q->wp_weight[r] * v + q->me.wp_luma_round) >> q->me.luma_log_weight_denom) + q->wp_offset[r]);
      mpr[by + y][bx + x] = (jmo_pel)v;
    }
  }
}

/* FindSkipModeMotionVector, mv-search.c:1189 -> all_mv[..][0][0] */
static void find_skip_mv(lc_ctx *c)
{
  int ax, ay, bx, by, availA, availB, zl, za, i, j;
  short pmv[2] = {0, 0};
  availA = nbr(c, -1, 0, &ax, &ay); availB = nbr(c, 0, -1, &bx, &by);
  zl = !availA ? 1 : (REFIDX(c, ay, ax) == 0 && MVAT(c, ay, ax)[0] == 0 && MVAT(c, ay, ax)[1] == 0);
  za = !availB ? 1 : (REFIDX(c, by, bx) == 0 && MVAT(c, by, bx)[0] == 0 && MVAT(c, by, bx)[1] == 0);
  if (!(za || zl)) set_mv_predictor(c, pmv, 0, 0, 0, 16, 16);
  for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) { c->all_mv[j][i][0][0][0] = pmv[0]; c->all_mv[j][i][0][0][1] = pmv[1]; }
}

/* GetSkipCostMB, mv-search.c:1136: LumaPrediction per 4x4 block of mode 0 / reference 0 (UMV fetch per block, explicit weights
 * when the picture parameter set has them), distortion4x4 of the mode-decision metric. 4x4 transform only here. */
static int skip_cost(const lc_ctx *c, const jmo_pel *mb /*packed 16x16*/)
{
  jmo_me_params p = c->q->me;
  jmo_dist d;
  p.metric[JMO_Q_PEL] = c->q->md_metric;
  p.apply_weights = c->q->wp_pred;
  if (p.apply_weights) { p.weight_luma = c->q->wp_weight[0]; p.offset_luma = c->q->wp_offset[0]; }
  p.chroma_me = 0;
  jmo_dist_from_params(&p, &c->refs[0], &d);
  d.umv = 1; d.test8x8 = 0; d.chroma_me = 0;
  if (c->q->transform8x8_mode) {       /* rdopt 0 with the 8x8 transform: distortion8x8 of each 8x8 block, predicted per 4x4 block (mv-search.c:1171-1176) */
    jmo_pel mpr[16][16];
    int ref4[16], b, c4, c8;
    short mv4[16][2];
    for (b = 0; b < 16; b++) { ref4[b] = 0; mv4[b][0] = c->all_mv[0][0][0][0][0]; mv4[b][1] = c->all_mv[0][0][0][0][1]; }
    predict_mb(c, ref4, (const short (*)[2])mv4, mpr, 4);
    jmo_pred_costs(mb, &mpr[0][0], c->q->md_metric, 1, &c4, &c8);
    return c8;
  }
  if (c->q->md_metric == JMO_ERR_SATD)
    return jmo_uni_pred(&p, JMO_Q_PEL, &d, mb, 16, 16, JMO_INT_MAX, ((c->mb_x * 16) << 2) + JMO_PAD4 + c->all_mv[0][0][0][0][0],
                        ((c->mb_y * 16) << 2) + JMO_PAD4 + c->all_mv[0][0][0][0][1]);
  {                                   /* SAD / SSE: the block origin is clamped per 4x4 block (LumaPrediction), not per 16x16: sum 4x4 calls */
    int bx, by, cost = 0, k;
    jmo_pel blk[16];
    for (by = 0; by < 16; by += 4) for (bx = 0; bx < 16; bx += 4) {
      for (k = 0; k < 4; k++) memcpy(blk + 4 * k, mb + (by + k) * 16 + bx, 4 * sizeof(jmo_pel));
      cost += jmo_uni_pred(&p, JMO_Q_PEL, &d, blk, 4, 4, JMO_INT_MAX, ((c->mb_x * 16 + bx) << 2) + JMO_PAD4 + c->all_mv[0][0][0][0][0],
                           ((c->mb_y * 16 + by) << 2) + JMO_PAD4 + c->all_mv[0][0][0][0][1]);
    }
    return cost;
  }
}

static int part_index(int blocktype, int block_x, int block_y)
{
  const int b8 = (block_y >> 1) * 2 + (block_x >> 1);
  switch (blocktype) {
  case 1: return 0;
  case 2: return 1 + (block_y >> 1);
  case 3: return 3 + (block_x >> 1);
  case 4: return 5 + b8;
  case 5: return 9 + b8 * 2 + (block_y & 1);
  case 6: return 17 + b8 * 2 + (block_x & 1);
  default: return 25 + b8 * 4 + (block_y & 1) * 2 + (block_x & 1);
  }
}

/* BlockMotionSearch, mv-search.c:560 (P slice, rdopt 0, no chroma ME) */
static int block_motion_search(lc_ctx *c, int ref, int mb_x, int mb_y, int blocktype, int search_range)
{
  const jmo_lowcplx_params *q = c->q;
  const int block_x = mb_x >> 2, block_y = mb_y >> 2, opix_x = c->mb_x * 16, opix_y = c->mb_y * 16;
  const int pic_pix_x = opix_x + mb_x, pic_pix_y = opix_y + mb_y;
  const int start_hp = (q->me.chroma_me == 1 || q->me.metric[JMO_F_PEL] != q->me.metric[JMO_H_PEL]) ? 0 : 1;
  int bsx, bsy, min_mcost = JMO_INT_MAX, i, j;
  short mv[2], up_mv[2] = {0, 0}, *pred_mv = c->pred_mv[block_y][block_x][ref][blocktype];
  jmo_pel orig[256];
  jmo_me_params p;
  const jmo_ref *rp = &c->refs[ref];
  jmo_block_size(blocktype, &bsx, &bsy);
  me_params_for_ref(c, ref, &p);
  for (j = 0; j < bsy; j++) memcpy(orig + j * bsx, c->cur + (size_t)(pic_pix_y + j) * c->cur_stride + pic_pix_x, bsx * sizeof(jmo_pel));

  if (q->search_mode == 1) {                                /* UMHEX: predictor + dynamic search range (:646-647) */
    jmo_umhex_nbr nb;
    jmo_lc_neighbours(c, mb_x, mb_y, bsx, &nb);
    jmo_umhex_set_mv_predictor(q->umhex, pred_mv, &nb, ref, 0, block_x, block_y, bsx, bsy, blocktype, 0, q->blocktype_lut, &search_range);
  } else set_mv_predictor(c, pred_mv, ref, block_x, block_y, bsx, bsy);

  if (q->search_mode == 1) {
    short allmv[2][JMO_MAX_REFS][9][2];
    memset(allmv, 0, sizeof(allmv));
    memcpy(allmv[0], c->all_mv[block_y][block_x], sizeof(allmv[0]));
    mv[0] = pred_mv[0] / 4; mv[1] = pred_mv[1] / 4;
    mv[0] = (short)clip3(-search_range, search_range, mv[0]); mv[1] = (short)clip3(-search_range, search_range, mv[1]);
    mv[0] = (short)clip3(-2047 + search_range, 2047 - search_range, mv[0]);
    mv[1] = (short)clip3(p.level_mv_min + search_range, p.level_mv_max - search_range, mv[1]);
    min_mcost = jmo_umhex_pel_search(q->umhex, &p, rp, orig, ref, 0, (const short (*)[JMO_MAX_REFS][9][2])allmv, q->frame_ctr_b, opix_x, opix_y,
                                     pic_pix_x, pic_pix_y, blocktype, pred_mv[0], pred_mv[1], &mv[0], &mv[1], search_range, min_mcost, q->lambda_mf[JMO_F_PEL]);
  } else if (q->search_mode == 3) {                         /* EPZS :709-740 */
    jmo_epzs_nbr nb; jmo_umhex_nbr n2;
    short allmv[JMO_MAX_REFS][8][2];
    int r, bt;
    jmo_lc_neighbours(c, mb_x, mb_y, bsx, &n2);
    memcpy(nb.available, n2.available, sizeof(nb.available)); memcpy(nb.ref, n2.ref, sizeof(nb.ref)); memcpy(nb.mv, n2.mv, sizeof(nb.mv));
    for (r = 0; r < JMO_MAX_REFS; r++) for (bt = 0; bt < 8; bt++) { allmv[r][bt][0] = c->all_mv[block_y][block_x][r][bt][0]; allmv[r][bt][1] = c->all_mv[block_y][block_x][r][bt][1]; }
    mv[0] = (short)((pred_mv[0] + 2) >> 2); mv[1] = (short)((pred_mv[1] + 2) >> 2);
    mv[0] = (short)clip3(-search_range, search_range, mv[0]); mv[1] = (short)clip3(-search_range, search_range, mv[1]);
    mv[0] = (short)clip3(-2047 + search_range, 2047 - search_range, mv[0]);
    mv[1] = (short)clip3(p.level_mv_min + search_range, p.level_mv_max - search_range, mv[1]);
    min_mcost = jmo_epzs_pel_search(q->epzs, &p, rp, orig, ref, 0, &nb, (const short (*)[8][2])allmv, 1, c->mb_nr, opix_x, opix_y, pic_pix_x, pic_pix_y,
                                    blocktype, pred_mv, mv, search_range, min_mcost, q->lambda_mf[JMO_F_PEL]);
  } else if (q->search_mode == 2) {                         /* simplified UMHexagonS :674-706 (smpUMHEX_setup :634-637: the upper layer's vector) */
    const int ub = blocktype > 6 ? 5 : blocktype > 4 ? 4 : blocktype == 4 ? 2 : 1;
    up_mv[0] = c->all_mv[block_y][block_x][ref][ub][0]; up_mv[1] = c->all_mv[block_y][block_x][ref][ub][1];
    mv[0] = pred_mv[0] / 4; mv[1] = pred_mv[1] / 4;
    if (!q->me.rdopt) { mv[0] = (short)clip3(-search_range, search_range, mv[0]); mv[1] = (short)clip3(-search_range, search_range, mv[1]); }
    mv[0] = (short)clip3(-2047 + search_range, 2047 - search_range, mv[0]);
    mv[1] = (short)clip3(p.level_mv_min + search_range, p.level_mv_max - search_range, mv[1]);
    min_mcost = jmo_umhexsmp_pel_search(&p, rp, orig, pic_pix_x, pic_pix_y, blocktype, pred_mv[0], pred_mv[1], &mv[0], &mv[1], search_range, min_mcost,
                                        q->lambda_mf[JMO_F_PEL], up_mv[0], up_mv[1]);
  } else if (q->search_mode == 0) {                         /* FastFull :741-748, SetupFastFullPelSearch on first use per reference */
    if (!c->ff_done[ref]) {
      const int R = (q->full_search == 2 || ref == 0) ? q->search_range : q->search_range / 2;
      short pmv16[2];
      jmo_pel mb[256];
      for (j = 0; j < 16; j++) memcpy(mb + 16 * j, c->cur + (size_t)(opix_y + j) * c->cur_stride + opix_x, 16 * sizeof(jmo_pel));
      set_mv_predictor(c, pmv16, ref, 0, 0, 16, 16);
      free(c->ff[ref].block_sad);
      c->ff[ref].block_sad = (int *)malloc(sizeof(int) * 8 * 16 * (size_t)(2 * R + 1) * (2 * R + 1));
      jmo_fastfull_setup(&p, rp, mb, opix_x, opix_y, pmv16[0], pmv16[1], R, &c->ff[ref]);
      c->ff_done[ref] = 1;
    }
    min_mcost = jmo_fastfull_search(&p, &c->ff[ref], opix_x, opix_y, pic_pix_x, pic_pix_y, blocktype, pred_mv[0], pred_mv[1], &mv[0], &mv[1],
                                    min_mcost, q->lambda_mf[JMO_F_PEL]);
  } else {                                                  /* FullSearch :749-768 */
    jmo_search_center(&p, pred_mv[0], pred_mv[1], search_range, &mv[0], &mv[1]);
    min_mcost = jmo_fullpel_search(&p, rp, orig, ref == 0, pic_pix_x, pic_pix_y, blocktype, pred_mv[0], pred_mv[1], &mv[0], &mv[1], search_range,
                                   min_mcost, q->lambda_mf[JMO_F_PEL]);
  }
  if (c->pass8ts) {
    const int b8 = (block_y >> 1) * 2 + (block_x >> 1);
    c->out->mv_int8ts[ref][b8][0] = mv[0]; c->out->mv_int8ts[ref][b8][1] = mv[1]; c->out->cost_int8ts[ref][b8] = min_mcost;
  } else {
    const int pi = part_index(blocktype, block_x, block_y);
    c->out->mv_int[ref][pi][0] = mv[0]; c->out->mv_int[ref][pi][1] = mv[1]; c->out->cost_int[ref][pi] = min_mcost;
  }
  mv[0] <<= 2; mv[1] <<= 2;                                 /* :770-774 */

  /* sub-pel :781-827 */
  if (q->search_mode != 3 || ref == 0 || (ref > 0 && min_mcost < 3.5 * jmo_epzs_distortion_row(q->epzs, 0, blocktype - 1)[pic_pix_x >> 2])) {
    if (!start_hp) min_mcost = JMO_INT_MAX;
    if (q->search_mode == 2 && blocktype > 1)               /* :803-815 */
      min_mcost = jmo_umhexsmp_subpel_search(&p, rp, orig, pic_pix_x, pic_pix_y, blocktype, pred_mv[0], pred_mv[1], &mv[0], &mv[1], min_mcost, q->lambda_mf[JMO_Q_PEL], up_mv[0], up_mv[1]);
    else if (q->search_mode == 2)
      min_mcost = jmo_umhexsmp_full_subpel_search(&p, rp, orig, ref == 0, pic_pix_x, pic_pix_y, blocktype, pred_mv[0], pred_mv[1], &mv[0], &mv[1], min_mcost, q->lambda_mf[JMO_Q_PEL]);
    else if (q->search_mode == 1 && blocktype > 3)
      min_mcost = jmo_umhex_subpel_search(q->umhex, &p, rp, orig, pic_pix_x, pic_pix_y, blocktype, pred_mv[0], pred_mv[1], &mv[0], &mv[1], min_mcost, q->lambda_mf[JMO_Q_PEL]);
    else if (q->search_mode == 3 && q->epzs_subpel_me)
      min_mcost = jmo_epzs_subpel_search(q->epzs, &p, rp, orig, pic_pix_x, pic_pix_y, blocktype, pred_mv, mv, 9, 9, min_mcost, q->lambda_mf);
    else
      min_mcost = jmo_subpel_search(&p, rp, orig, ref == 0, pic_pix_x, pic_pix_y, blocktype, pred_mv[0], pred_mv[1], &mv[0], &mv[1], 9, 9, min_mcost, q->lambda_mf);
  }
  /* skip-mode shortcut :829-849 (every reference of the 16x16 block) */
  if (blocktype == 1 && !q->me.rdopt) {                      /* :826: if (!input->rdopt) */
    int cost;
    find_skip_mv(c);
    cost = skip_cost(c, orig) - ((q->lambda_mf[JMO_Q_PEL] + 4096) >> 13);
    if (cost < min_mcost) { min_mcost = cost; mv[0] = c->all_mv[0][0][0][0][0]; mv[1] = c->all_mv[0][0][0][0][1]; }
  }
  for (j = block_y; j < block_y + (bsy >> 2); j++) for (i = block_x; i < block_x + (bsx >> 2); i++) {
    c->all_mv[j][i][ref][blocktype][0] = mv[0]; c->all_mv[j][i][ref][blocktype][1] = mv[1];
  }
  if (c->pass8ts) {
    const int b8 = (block_y >> 1) * 2 + (block_x >> 1);
    c->out->pred8ts[ref][b8][0] = pred_mv[0]; c->out->pred8ts[ref][b8][1] = pred_mv[1];
    c->out->mv8ts[ref][b8][0] = mv[0]; c->out->mv8ts[ref][b8][1] = mv[1]; c->out->cost8ts[ref][b8] = min_mcost;
  } else {
    const int pi = part_index(blocktype, block_x, block_y);
    c->out->pred[ref][pi][0] = pred_mv[0]; c->out->pred[ref][pi][1] = pred_mv[1];
    c->out->mv[ref][pi][0] = mv[0]; c->out->mv[ref][pi][1] = mv[1]; c->out->cost[ref][pi] = min_mcost;
  }
  return min_mcost;
}

/* PartitionMotionSearch, mv-search.c:1378 */
static void partition_motion_search(lc_ctx *c, int blocktype, int block8x8)
{
  static const int bx0[5][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 2, 0, 0}, {0, 2, 0, 2}};
  static const int by0[5][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 2, 0, 0}, {0, 0, 0, 0}, {0, 0, 2, 2}};
  static const int part_size[8][2] = {{4, 4}, {4, 4}, {4, 2}, {2, 4}, {2, 2}, {2, 1}, {1, 2}, {1, 1}};
  const jmo_lowcplx_params *q = c->q;
  const int parttype = blocktype < 4 ? blocktype : 4;
  const int step_h0 = part_size[parttype][0], step_v0 = part_size[parttype][1], step_h = part_size[blocktype][0], step_v = part_size[blocktype][1];
  const int by = by0[parttype][block8x8], bx = bx0[parttype][block8x8];
  int ref, v, h, i, j;
  for (ref = 0; ref < q->num_refs; ref++) {
    int search_range, *m_cost = &c->motion_cost[blocktype][ref][block8x8];
    if (q->full_search == 2) search_range = q->search_range;
    else if (q->full_search == 1) search_range = q->search_range / (imin_(ref, 1) + 1);
    else search_range = q->search_range / ((imin_(ref, 1) + 1) * imin_(2, blocktype));
    *m_cost = 0;
    for (v = by; v < by + step_v0; v += step_v)
      for (h = bx; h < bx + step_h0; h += step_h) {
        const int pby = c->mb_y * 4 + v, pbx = c->mb_x * 4 + h;
        *m_cost += block_motion_search(c, ref, h << 2, v << 2, blocktype, search_range);
        for (j = pby; j < pby + step_v; j++) for (i = pbx; i < pbx + step_h; i++) {
          REFIDX(c, j, i) = (signed char)ref;
          MVAT(c, j, i)[0] = c->all_mv[v][h][ref][blocktype][0]; MVAT(c, j, i)[1] = c->all_mv[v][h][ref][blocktype][1];
        }
      }
  }
}

/* list_prediction_cost for LIST_0, mode_decision.c:255 (rdopt 0: (int)(2 * lambda_me[Q_PEL] * min(ref, 1)) = q->ref_cost1 for ref > 0) */
static int list0_cost(const lc_ctx *c, int mode, int block, int *best_ref)
{
  int ref, best = JMO_INT_MAX;
  for (ref = 0; ref < c->q->num_refs; ref++) {
    const int mcost = (ref ? c->q->ref_cost1 : 0) + c->motion_cost[mode][ref][block];
    if (mcost < best) { best = mcost; *best_ref = ref; }
  }
  return best;
}

static int refbits_(int ref)           /* mv-search.c:344-352 */
{
  int bits;
  if (ref == 0) return 1;
  for (bits = 3; ; bits += 2) { const int i_max = (1 << ((bits >> 1) + 1)) - 1, i_min = i_max >> 1; if (ref >= i_min && ref < i_max) return bits; }
}

static void macroblock_low(lc_ctx *c)
{
  static const int part_size[8][2] = {{4, 4}, {4, 4}, {4, 2}, {2, 4}, {2, 2}, {2, 1}, {1, 2}, {1, 1}};
  const jmo_lowcplx_params *q = c->q;
  int mode, block, best_mode = 1, min_cost = JMO_INT_MAX, cost, i, j, k, r;
  int best8x8l0ref[5][4], best8x8mode[4] = {0, 0, 0, 0};
  int p8cand_mode[4] = {0, 0, 0, 0}, p8cand_ref[4] = {0, 0, 0, 0};            /* [1..3] and [4] = P8x8 */
  const int T8 = q->transform8x8_mode;
  int t8_flag = 0, best_transform_flag = 0, cbp8ts = -1;
  int ref8ts[4] = {0, 0, 0, 0}; short mv8ts[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};       /* StoreNewMotionVectorsBlock8x8 of the 8x8-transform pass */
  int tr8_cost = JMO_INT_MAX, tr4_cost = JMO_INT_MAX;
  jmo_pel curmb[256];
  for (j = 0; j < 16; j++) memcpy(curmb + 16 * j, c->cur + (size_t)(c->mb_y * 16 + j) * c->cur_stride + c->mb_x * 16, 16 * sizeof(jmo_pel));
  const int bx0 = c->mb_x * 4, by0 = c->mb_y * 4;
  /* img->all_mv is NOT reset per macroblock (nor per slice or picture): EPZSBlockTypePredictors (me_epzs.c:1433) reads the 16x16 and
   * 8x8 vectors of the PREVIOUS macroblock in coding order before this one has searched those types -- for the first macroblock
   * of a row that is the last one of the row above. Kept. */
  memset(best8x8l0ref, 0, sizeof(best8x8l0ref));
  for (r = 0; r < JMO_MAX_REFS; r++) c->ff_done[r] = 0;                  /* ResetFastFullIntegerSearch, macroblock.c:509 */
  if (q->search_mode == 1) jmo_umhex_decide_intrabk_sad(q->umhex, 0, c->mb_x * 16, c->mb_y * 16);   /* md_low.c:81-84 */

  for (mode = 1; mode < 4; mode++) {
    if (!q->valid[mode]) continue;
    for (cost = 0, block = 0; block < (mode == 1 ? 1 : 2); block++) {
      int best_ref = 0, bm;
      partition_motion_search(c, mode, block);
      bm = list0_cost(c, mode, block, &best_ref);
      cost += bm;
      if (mode == 1) {                                        /* assign_enc_picture_params, rdopt.c:3546-3561 */
        for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) {
          REFIDX(c, by0 + j, bx0 + i) = (signed char)best_ref;
          MVAT(c, by0 + j, bx0 + i)[0] = c->all_mv[j][i][best_ref][1][0]; MVAT(c, by0 + j, bx0 + i)[1] = c->all_mv[j][i][best_ref][1][1];
        }
        for (k = 0; k < 4; k++) best8x8l0ref[1][k] = best_ref;
      } else if (mode == 2) best8x8l0ref[2][2 * block] = best8x8l0ref[2][2 * block + 1] = best_ref;
      else best8x8l0ref[3][block] = best8x8l0ref[3][block + 2] = best_ref;
      if (mode > 1 && block == 0) {                           /* SetRefAndMotionVectors, rdopt.c:2806-2821 */
        const int j1 = part_size[mode][1], i1 = part_size[mode][0];
        for (j = 0; j < j1; j++) for (i = 0; i < i1; i++) {
          REFIDX(c, by0 + j, bx0 + i) = (signed char)best_ref;
          MVAT(c, by0 + j, bx0 + i)[0] = c->all_mv[j][i][best_ref][mode][0]; MVAT(c, by0 + j, bx0 + i)[1] = c->all_mv[j][i][best_ref][mode][1];
        }
      }
    }
    if (T8) {                                                 /* TransformDecision(currMB, -1, &cost), md_low.c:183-188, macroblock.c:1458 */
      jmo_pel mpr[16][16];
      int ref4[16], c4, c8;
      short mv4[16][2];
      /* SetModesAndRefframeForBlocks(currMB, mode) (md_low.c:186, rdopt.c:1470-1500) writes the mode's best references into enc_picture->ref_idx --
       * a side effect only Transform8x8Mode has: without it the second block of modes 2 / 3 leaves the last reference searched there --
       * and SetModesAndRefframe (macroblock.c:1301) reads them back for the prediction */
      for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) {
        const int rr = best8x8l0ref[mode][2 * (j >> 1) + (i >> 1)];
        REFIDX(c, by0 + j, bx0 + i) = (signed char)rr;
        ref4[j * 4 + i] = rr; mv4[j * 4 + i][0] = c->all_mv[j][i][rr][mode][0]; mv4[j * 4 + i][1] = c->all_mv[j][i][rr][mode][1];
      }
      predict_mb(c, ref4, (const short (*)[2])mv4, mpr, 4);
      jmo_pred_costs(curmb, &mpr[0][0], q->md_metric, 0, &c4, &c8);
      if (T8 == 2 || c8 < c4) t8_flag = 1;
      else { cost = cost - c8 + c4; t8_flag = 0; }
    }
    if (cost < min_cost) { best_mode = mode; min_cost = cost; best_transform_flag = t8_flag; }
  }

  if (q->valid[4] || q->valid[5] || q->valid[6] || q->valid[7]) {       /* P8x8: md_low.c:190-330 */
    int cost8x8 = 0;
    if (T8) {                                                 /* the 8x8 partition with the 8x8 transform: sub-mode 4 only (mode_decision.c:556) */
      tr8_cost = 0;
      c->pass8ts = 1;
      for (block = 0; block < 4; block++) {
        const int pbx = bx0 + (block & 1) * 2, pby = by0 + (block & 2), j0 = block & 2, i0 = (block & 1) * 2;
        int best_ref = 0;
        partition_motion_search(c, 4, block);
        cost = list0_cost(c, 4, block, &best_ref);
        for (j = pby; j < pby + 2; j++) for (i = pbx; i < pbx + 2; i++) REFIDX(c, j, i) = (signed char)best_ref;
        if (cost != JMO_INT_MAX) cost += ((q->lambda_mf[JMO_Q_PEL] * (q->num_refs <= 1 ? 0 : refbits_(0))) >> 16) - 1;
        tr8_cost += cost;
        ref8ts[block] = best_ref; mv8ts[block][0] = c->all_mv[j0][i0][best_ref][4][0]; mv8ts[block][1] = c->all_mv[j0][i0][best_ref][4][1];
        for (j = j0; j < j0 + 2; j++) for (i = i0; i < i0 + 2; i++) {          /* SetRefAndMotionVectors, mode_decision.c:965 */
          REFIDX(c, by0 + j, bx0 + i) = (signed char)best_ref;
          MVAT(c, by0 + j, bx0 + i)[0] = mv8ts[block][0]; MVAT(c, by0 + j, bx0 + i)[1] = mv8ts[block][1];
        }
      }
      c->pass8ts = 0;
    }
    if (T8 != 2) {
    for (block = 0; block < 4; block++) {
      int min_cost8x8 = JMO_INT_MAX;
      const int pbx = bx0 + (block & 1) * 2, pby = by0 + (block & 2);
      for (mode = 4; mode < 8; mode++) {
        int best_ref = 0;
        if (!q->valid[mode]) continue;
        partition_motion_search(c, mode, block);
        cost = list0_cost(c, mode, block, &best_ref);
        for (j = pby; j < pby + 2; j++) for (i = pbx; i < pbx + 2; i++) REFIDX(c, j, i) = (signed char)best_ref;     /* mode_decision.c:682-690 */
        if (cost != JMO_INT_MAX)                              /* :729-731: REF_COST(lambda_mf[Q_PEL], B8Mode2Value(mode, 0), list) - 1 */
          cost += ((q->lambda_mf[JMO_Q_PEL] * (q->num_refs <= 1 ? 0 : refbits_(mode - 4))) >> 16) - 1;
        if (cost < min_cost8x8) { min_cost8x8 = cost; best8x8mode[block] = mode; best8x8l0ref[4][block] = best_ref; }
      }
      cost8x8 += min_cost8x8;
      {                                                       /* mode_decision.c:965: SetRefAndMotionVectors with the winning sub-mode */
        const int m8 = best8x8mode[block], r8 = best8x8l0ref[4][block], j0 = block & 2, i0 = (block & 1) * 2;
        for (j = j0; j < j0 + 2; j++) for (i = i0; i < i0 + 2; i++) {
          REFIDX(c, by0 + j, bx0 + i) = (signed char)r8;
          MVAT(c, by0 + j, bx0 + i)[0] = c->all_mv[j][i][r8][m8][0]; MVAT(c, by0 + j, bx0 + i)[1] = c->all_mv[j][i][r8][m8][1];
        }
      }
    }
    tr4_cost = cost8x8;
    for (k = 0; k < 4; k++) { p8cand_mode[k] = best8x8mode[k]; p8cand_ref[k] = best8x8l0ref[4][k]; }      /* the candidate, before a winning 8x8-transform pass rewrites it below */
    }
    if (tr4_cost < min_cost || tr8_cost < min_cost) {         /* md_low.c:281-326 */
      best_mode = 8;
      if (T8 == 2) { min_cost = tr8_cost; t8_flag = 1; }
      else if (T8) {
        if (tr8_cost < tr4_cost) { min_cost = tr8_cost; t8_flag = 1; }
        else if (tr4_cost < tr8_cost) { min_cost = tr4_cost; t8_flag = 0; }
        else {                                                /* GetBestTransformP8x8, rdopt.c:3262: the two passes' predictions */
          jmo_pel mpr[16][16];
          int ref4[16], c4, c8, dummy;
          short mv4[16][2];
          for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) {
            const int k8 = 2 * (j >> 1) + (i >> 1), rr = best8x8l0ref[4][k8], m8 = best8x8mode[k8];
            ref4[j * 4 + i] = rr; mv4[j * 4 + i][0] = c->all_mv[j][i][rr][m8][0]; mv4[j * 4 + i][1] = c->all_mv[j][i][rr][m8][1];
          }
          predict_mb(c, ref4, (const short (*)[2])mv4, mpr, 4);
          jmo_pred_costs(curmb, &mpr[0][0], q->md_metric, 0, &c4, &dummy);
          for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) { const int k8 = 2 * (j >> 1) + (i >> 1); ref4[j * 4 + i] = ref8ts[k8]; mv4[j * 4 + i][0] = mv8ts[k8][0]; mv4[j * 4 + i][1] = mv8ts[k8][1]; }
          predict_mb(c, ref4, (const short (*)[2])mv4, mpr, 8);
          jmo_pred_costs(curmb, &mpr[0][0], q->md_metric, 0, &dummy, &c8);
          if (c8 < c4) { min_cost = tr8_cost; t8_flag = 1; } else { min_cost = tr4_cost; t8_flag = 0; }
        }
      } else { min_cost = tr4_cost; t8_flag = 0; }
    }
  }
  find_skip_mv(c);                                            /* md_low.c:332-333 */
  if (best_mode != 8) t8_flag = best_transform_flag;          /* :596-597 */
  else if (t8_flag && T8 != 2) {
    /* md_low.c:547-548: the 8x8-transform pass wins only if its residual has a coded 8x8 block: LumaResidualCoding8x8 (macroblock.c:1009) of
     * its four blocks -- one 8x8 prediction each, dct_8x8, a block whose coefficient cost is <= _LUMA_COEFF_COST_ (4) counts as empty */
    jmo_pel mpr[16][16], recon[16][16];
    int ref4[16], m7[16][16], levels[4][65], runs[4][65], fadj[16][16];
    short mv4[16][2];
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) { const int k8 = 2 * (j >> 1) + (i >> 1); ref4[j * 4 + i] = ref8ts[k8]; mv4[j * 4 + i][0] = mv8ts[k8][0]; mv4[j * 4 + i][1] = mv8ts[k8][1]; }
    predict_mb(c, ref4, (const short (*)[2])mv4, mpr, 8);
    for (j = 0; j < 16; j++) for (i = 0; i < 16; i++) m7[j][i] = (int)curmb[j * 16 + i] - (int)mpr[j][i];
    cbp8ts = 0;
    for (block = 0; block < 4; block++) {
      int coeff_cost = 0;
      const int nonzero = jmo_dct_8x8((const jmo_quant *)q->q8, m7, (const jmo_pel (*)[16])mpr, block, &coeff_cost, levels, runs, recon, fadj);
      if (nonzero && coeff_cost > 4) cbp8ts |= 1 << block;
    }
    if (cbp8ts == 0) t8_flag = 0;
  }
  if (best_mode == 8 && t8_flag) {                            /* SetCoeffAndReconstruction8x8, rdopt.c:1595: the 8x8-transform pass's partitioning */
    for (k = 0; k < 4; k++) { best8x8mode[k] = 4; best8x8l0ref[4][k] = ref8ts[k]; }
    for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) { const int k8 = 2 * (j >> 1) + (i >> 1); c->all_mv[j][i][ref8ts[k8]][4][0] = mv8ts[k8][0]; c->all_mv[j][i][ref8ts[k8]][4][1] = mv8ts[k8][1]; }
  }
  c->out->transform8x8_flag = t8_flag; c->out->cbp8ts = cbp8ts;

  /* SetModesAndRefframeForBlocks + SetMotionVectorsMB (rdopt.c:1262, :1845) */
  c->out->best_mode = best_mode; c->out->min_cost = min_cost;
  for (k = 0; k < 4; k++) {
    c->out->b8mode[k] = best_mode == 8 ? best8x8mode[k] : best_mode;
    c->out->b8ref[k] = best8x8l0ref[best_mode == 8 ? 4 : best_mode][k];
    c->out->p8mode[k] = p8cand_mode[k]; c->out->p8ref[k] = p8cand_ref[k];
  }
  for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) {
    const int k8 = 2 * (j >> 1) + (i >> 1), ref = c->out->b8ref[k8], m8 = c->out->b8mode[k8];
    REFIDX(c, by0 + j, bx0 + i) = (signed char)ref;
    MVAT(c, by0 + j, bx0 + i)[0] = c->all_mv[j][i][ref][m8][0]; MVAT(c, by0 + j, bx0 + i)[1] = c->all_mv[j][i][ref][m8][1];
    c->out->final_mv[j * 4 + i][0] = c->all_mv[j][i][ref][m8][0]; c->out->final_mv[j * 4 + i][1] = c->all_mv[j][i][ref][m8][1];
  }
  c->out->skip_mv[0] = c->all_mv[0][0][0][0][0]; c->out->skip_mv[1] = c->all_mv[0][0][0][0][1];
  if (q->search_mode == 1) jmo_umhex_skip_intrabk_sad(q->umhex, best_mode, q->num_refs, q->img_number, 0, c->mb_x * 16);   /* md_low.c:678-681 */
}

/* The whole slice / picture: macroblocks [mb_first, mb_first + mb_count) in raster order. ref_idx / mv are the picture-level arrays
 * ([H/4][W/4], [H/4][W/4][2]); the caller keeps them between calls of one picture (several slices) and reads the final field there. */
void jmo_lowcplx_p_slice(const jmo_lowcplx_params *q, const jmo_ref *refs, const jmo_pel *cur, int cur_stride,
                         signed char *ref_idx, short *mv, int mb_first, int mb_count, jmo_mb_inter *out)
{
  lc_ctx *c = (lc_ctx *)calloc(1, sizeof(*c));
  int n, r;
  c->q = q; c->refs = refs; c->cur = cur; c->cur_stride = cur_stride;
  c->mbw = q->W / 16; c->mbh = q->H / 16; c->w4 = q->W / 4; c->h4 = q->H / 4;
  c->ref_idx = ref_idx; c->mv = mv;
  if (q->all_mv_state) memcpy(c->all_mv, q->all_mv_state, sizeof(c->all_mv));
  for (n = mb_first; n < mb_first + mb_count; n++) {
    c->mb_nr = n; c->mb_x = n % c->mbw; c->mb_y = n / c->mbw;
    c->out = &out[n - mb_first];
    memset(c->out, 0, sizeof(*c->out));
    macroblock_low(c);
  }
  if (q->all_mv_state) memcpy(q->all_mv_state, c->all_mv, sizeof(c->all_mv));
  for (r = 0; r < JMO_MAX_REFS; r++) free(c->ff[r].block_sad);
  free(c);
}
