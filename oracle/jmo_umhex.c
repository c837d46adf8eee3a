/*
 * jmo_umhex.c -- ORACLE (test infrastructure): UMHexagonS motion search (SearchMode = 1).
 * Restates lencod/src/me_umhex.c of the reference: UMHEX_DefineThreshold(MB) :78/:108, UMHEXIntegerPelBlockMotionSearch :229,
 * UMHEXSubPelBlockMotionSearch :562, UMHEX_decide_intrabk_SAD :745, UMHEX_skip_intrabk_SAD :769, UMHEX_setup :797,
 * UMHEXBipredIntegerPelBlockMotionSearch :916, UMHEXSetMotionVectorPredictor :1298 (with the dynamic search range),
 * and the SEARCH_ONE_PIXEL / EARLY_TERMINATION macros of lencod/inc/me_umhex.h:26-75. Frame pictures without MBAFF.
 *
 * One jmo_umhex object = the global state of me_umhex.h (cost memories, visited maps, SAD / vector predictions, thresholds).
 * The thresholds are FLOAT in the reference (me_umhex.h:95-97, me_umhex.c:116-145, :258, :389-390): the same expressions are
 * kept here, evaluated in single precision like JM's build (-ffloat-store, SSE arithmetic).
 */
#include <stdlib.h>
#include <string.h>
#include "jmo.h"

static inline int iabs_(int x) { return x < 0 ? -x : x; }
static inline int imin_(int a, int b) { return a < b ? a : b; }
static inline int imax_(int a, int b) { return a > b ? a : b; }

static const int Diamond_x[4] = {-1, 0, 1, 0}, Diamond_y[4] = {0, 1, 0, -1};
static const int Hexagon_x[6] = {2, 1, -1, -2, -1, 1}, Hexagon_y[6] = {0, -2, -2, 0, 2, 2};
static const int Big_Hexagon_x[16] = {0, -2, -4, -4, -4, -4, -4, -2, 0, 2, 4, 4, 4, 4, 4, 2};
static const int Big_Hexagon_y[16] = {4, 3, 2, 1, 0, -1, -2, -3, -4, -3, -2, -1, 0, 1, 2, 3};
static const int Multi_Ref_Thd[8] = {0, 300, 120, 120, 60, 30, 30, 15};
static const int Big_Hexagon_Thd[8] = {0, 3000, 1500, 1500, 800, 400, 400, 200};
static const int Median_Pred_Thd[8] = {0, 750, 350, 350, 170, 80, 80, 40};
static const int Threshold_DSR[8] = {0, 2200, 1000, 1000, 500, 250, 250, 120};
#define UM_Q_BITS 15
#define UM_MIN_IMG_WIDTH 176

struct jmo_umhex {
  jmo_umhex_config cfg;
  int w4, h4, stride;                 /* stride of McostState rows: 2*input->search_range+1 */
  unsigned char *mcost_state;         /* McostState, contiguous like get_mem2D */
  unsigned char search_state[7][7];   /* SearchState */
  int *ref_cost;                      /* fastme_ref_cost [max_refs][9][4][4] */
  int *l_cost[2], *l_cost_bipred[2];  /* fastme_l0/l1_cost(_bipred) [9][h4][w4] */
  int *best_cost;                     /* fastme_best_cost [7][w4] */
  unsigned char *flag_intra;          /* [w/16+1] */
  int flag_intra_sad;
  int pred_sad, pred_mv_ref[2], pred_mv_uplayer[2], pred_mv_ref_flag;
  int sad_a, sad_b, sad_c, sad_d;
  int median_thd[8], big_hex_thd[8], multi_ref_thd[8], dsr_thd[8];
  float bsize[8], alpha1[8], alpha2[8];
  short *spx, *spy;                   /* spiral_search_x/y */
};

jmo_umhex *jmo_umhex_create(const jmo_umhex_config *c)        /* UMHEX_get_mem :154 + UMHEX_DefineThreshold :78 */
{
  jmo_umhex *u = (jmo_umhex *)calloc(1, sizeof(*u));
  const int R = c->search_range, n = imax_(25, (2 * imax_(1, R) + 1) * (2 * imax_(1, R) + 1));
  int i;
  u->cfg = *c;
  u->w4 = c->width / 4; u->h4 = c->height / 4; u->stride = 2 * R + 1;
  u->mcost_state = (unsigned char *)calloc((size_t)(2 * imax_(R, c->bipred_search_range) + 1) * (2 * imax_(R, c->bipred_search_range) + 1) + 64, 1);
  u->ref_cost = (int *)calloc((size_t)c->max_refs * 9 * 16, sizeof(int));
  for (i = 0; i < 2; i++) {
    u->l_cost[i] = (int *)calloc((size_t)9 * u->h4 * u->w4, sizeof(int));
    u->l_cost_bipred[i] = (int *)calloc((size_t)9 * u->h4 * u->w4, sizeof(int));
  }
  u->best_cost = (int *)calloc((size_t)7 * u->w4, sizeof(int));
  u->flag_intra = (unsigned char *)calloc((size_t)(c->width >> 4) + 1, 1);
  u->spx = (short *)malloc(sizeof(short) * (n + 2)); u->spy = (short *)malloc(sizeof(short) * (n + 2));
  jmo_spiral(imax_(2, R), u->spx, u->spy, imax_(n, 25));
  {
    static const float a1[8] = {0, 0.01f, 0.01f, 0.01f, 0.02f, 0.03f, 0.03f, 0.04f};
    static const float a2[8] = {0, 0.06f, 0.07f, 0.07f, 0.08f, 0.12f, 0.11f, 0.15f};
    for (i = 0; i < 8; i++) { u->alpha1[i] = a1[i]; u->alpha2[i] = a2[i]; }
  }
  {                                    /* UMHEX_DefineThresholdMB :108-146, expression for expression */
    int gb_qp_per = (c->qp_n - 0) / 6, gb_qp_rem = (c->qp_n - 0) % 6;
    int gb_q_bits = UM_Q_BITS + gb_qp_per, gb_qp_const, Thresh4x4;
    float Quantize_step;
    float scale_factor = (float)((1 - c->scale * 0.1) + c->scale * 0.1 * (c->width / UM_MIN_IMG_WIDTH));
    float QP_factor = (float)((1.0 - 0.90 * (c->qp_n / 51.0f)));
    gb_qp_const = (1 << gb_q_bits) / 6;
    Thresh4x4 = ((1 << gb_q_bits) - gb_qp_const) / jmo_quant_coef[gb_qp_rem][0][0];
    Quantize_step = Thresh4x4 / (4 * 5.61f) * 2.0f * scale_factor;
    u->bsize[7] = (16 * 16) * Quantize_step;
    u->bsize[6] = u->bsize[7] * 4; u->bsize[5] = u->bsize[7] * 4; u->bsize[4] = u->bsize[5] * 4;
    u->bsize[3] = u->bsize[4] * 4; u->bsize[2] = u->bsize[4] * 4; u->bsize[1] = u->bsize[2] * 4;
    for (i = 1; i < 8; i++) {
      u->median_thd[i] = (int)(Median_Pred_Thd[i] * scale_factor * QP_factor);
      u->big_hex_thd[i] = (int)(Big_Hexagon_Thd[i] * scale_factor * QP_factor);
      u->multi_ref_thd[i] = (int)(Multi_Ref_Thd[i] * scale_factor * QP_factor);
      u->dsr_thd[i] = (int)(Threshold_DSR[i] * scale_factor * QP_factor);
    }
  }
  return u;
}

void jmo_umhex_destroy(jmo_umhex *u)
{
  int i;
  if (!u) return;
  free(u->mcost_state); free(u->ref_cost); free(u->best_cost); free(u->flag_intra); free(u->spx); free(u->spy);
  for (i = 0; i < 2; i++) { free(u->l_cost[i]); free(u->l_cost_bipred[i]); }
  free(u);
}

void jmo_umhex_thresholds(const jmo_umhex *u, int *median, int *bighex, int *multiref, int *dsr, float *bsize, float *alpha1, float *alpha2)
{
  int i;
  for (i = 0; i < 8; i++) {
    median[i] = u->median_thd[i]; bighex[i] = u->big_hex_thd[i]; multiref[i] = u->multi_ref_thd[i]; dsr[i] = u->dsr_thd[i];
    bsize[i] = u->bsize[i]; alpha1[i] = u->alpha1[i]; alpha2[i] = u->alpha2[i];
  }
}

#define LC(arr, bt, y, x) (arr)[((size_t)(bt) * u->h4 + (y)) * u->w4 + (x)]
#define RC(r, bt, y, x)   u->ref_cost[(((size_t)(r) * 9 + (bt)) * 4 + (y)) * 4 + (x)]

/* UMHEX_decide_intrabk_SAD :745 */
void jmo_umhex_decide_intrabk_sad(jmo_umhex *u, int is_i_slice, int pix_x, int pix_y)
{
  if (is_i_slice) return;
  if (pix_x == 0 && pix_y == 0) u->flag_intra_sad = 0;
  else if (pix_x == 0) u->flag_intra_sad = u->flag_intra[pix_x >> 4];
  else if (pix_y == 0) u->flag_intra_sad = u->flag_intra[(pix_x >> 4) - 1];
  else u->flag_intra_sad = (u->flag_intra[pix_x >> 4] || u->flag_intra[(pix_x >> 4) - 1] || u->flag_intra[(pix_x >> 4) + 1]);
}

/* UMHEX_skip_intrabk_SAD :769. The cost memories are cleared at [k][0..3][0..3] -- the picture's top-left corner, not the
 * macroblock's position: that is what the reference does. */
void jmo_umhex_skip_intrabk_sad(jmo_umhex *u, int best_mode, int ref_max, int img_number, int is_i_slice, int pix_x)
{
  int i, j, k, r;
  if (img_number > 0) u->flag_intra[pix_x >> 4] = (best_mode == 9 || best_mode == 10) ? 1 : 0;
  if (!is_i_slice && (best_mode == 9 || best_mode == 10))
    for (i = 0; i < 4; i++) for (j = 0; j < 4; j++) for (k = 0; k < 9; k++) {
      LC(u->l_cost[0], k, j, i) = 0; LC(u->l_cost[1], k, j, i) = 0;
      for (r = 0; r < ref_max; r++) RC(r, k, j, i) = 0;
    }
}

/* UMHEX_setup :797 (frame pictures) */
static void umhex_setup(jmo_umhex *u, int ref, int list, int block_y, int block_x, int blocktype, const short (*allmv)[JMO_MAX_REFS][9][2],
                        int is_b_slice, int frame_ctr_b, int pix_x, int pix_y)
{
  static const int indication_blocktype[8] = {0, 0, 1, 1, 2, 4, 4, 5};
  const int N_Bframe = u->cfg.successive_bframe;
  const int n_Bframe = N_Bframe ? (frame_ctr_b % (N_Bframe + 1)) : 0;
  int temp_blocktype = 0;
  if (blocktype > 1) {
    temp_blocktype = indication_blocktype[blocktype];
    u->pred_mv_uplayer[0] = allmv[list][ref][temp_blocktype][0];
    u->pred_mv_uplayer[1] = allmv[list][ref][temp_blocktype][1];
  }
  u->pred_mv_ref_flag = 0;
  if (list == 0) {
    if (ref > 0) {
      u->pred_mv_ref[0] = allmv[0][ref - 1][blocktype][0];
      u->pred_mv_ref[0] = (int)(u->pred_mv_ref[0] * (ref + 1) / (float)(ref));
      u->pred_mv_ref[1] = allmv[0][ref - 1][blocktype][1];
      u->pred_mv_ref[1] = (int)(u->pred_mv_ref[1] * (ref + 1) / (float)(ref));
      u->pred_mv_ref_flag = 1;
    }
    if (is_b_slice && ref == 0) {
      u->pred_mv_ref[0] = (int)(allmv[1][0][blocktype][0] * (-n_Bframe) / (N_Bframe - n_Bframe + 1.0f));
      u->pred_mv_ref[1] = (int)(allmv[1][0][blocktype][1] * (-n_Bframe) / (N_Bframe - n_Bframe + 1.0f));
      u->pred_mv_ref_flag = 1;
    }
  }
  if (list == 0 && ref > 0) u->pred_sad = u->flag_intra_sad ? 0 : RC(ref - 1, blocktype, block_y, block_x);
  else if (blocktype > 1) {
    if (u->flag_intra_sad) u->pred_sad = 0;
    else {
      u->pred_sad = LC(u->l_cost[list == 1], temp_blocktype, (pix_y >> 2) + block_y, (pix_x >> 2) + block_x);
      u->pred_sad /= 2;
    }
  } else u->pred_sad = 0;
}

/* ------------------------------------------------------------------ the integer walk shared by the uni- and bi-predictive forms */

typedef struct {
  jmo_umhex *u;
  const jmo_me_params *p; jmo_dist d; jmo_bipred *b; const jmo_pel *cur;
  int bsx, bsy, lambda, pred_x, pred_y, fixed_cost, c1x, c1y;
  int center_x, center_y, search_range;
  int best_x, best_y, min_mcost;
} uwalk;

#define MCS(w, cy, cx) (w)->u->mcost_state[(size_t)((cy) - (w)->center_y + (w)->search_range) * (w)->u->stride + ((cx) - (w)->center_x + (w)->search_range)]

static int u_mvcost(const uwalk *w, int cand_x, int cand_y)            /* MV_COST with mvshift 2, defines.h:127 */
{
  int c = jmo_mv_cost(w->lambda, cand_x << 2, cand_y << 2, w->pred_x, w->pred_y);
  return w->b ? c + w->fixed_cost : c;
}
static int u_dist(uwalk *w, int bound, int cand_x, int cand_y)
{
  if (w->b) return jmo_bipred_sad(w->b, w->cur, w->bsy, w->bsx, bound, (w->c1x << 2) + JMO_PAD4, (w->c1y << 2) + JMO_PAD4, (cand_x << 2) + JMO_PAD4, (cand_y << 2) + JMO_PAD4);
  return jmo_uni_pred(w->p, JMO_F_PEL, &w->d, w->cur, w->bsy, w->bsx, bound, (cand_x << 2) + JMO_PAD4, (cand_y << 2) + JMO_PAD4);
}
/* SEARCH_ONE_PIXEL(_BIPRED), me_umhex.h:32-75 */
static void search_one_pixel(uwalk *w, int cand_x, int cand_y)
{
  int mcost;
  if (iabs_(cand_x - w->center_x) > w->search_range || iabs_(cand_y - w->center_y) > w->search_range) return;
  if (MCS(w, cand_y, cand_x)) return;
  mcost = u_mvcost(w, cand_x, cand_y);
  if (mcost < w->min_mcost) {
    mcost += u_dist(w, w->min_mcost - mcost, cand_x, cand_y);
    MCS(w, cand_y, cand_x) = 1;
    if (mcost < w->min_mcost) { w->best_x = cand_x; w->best_y = cand_y; w->min_mcost = mcost; }
  }
}
static void diamond(uwalk *w) { int m, x = w->best_x, y = w->best_y; for (m = 0; m < 4; m++) search_one_pixel(w, x + Diamond_x[m], y + Diamond_y[m]); }

/* Steps "first" .. "fourth" of both searches (:395-530, :1150-1267). has_uplayer / has_ref: the two start-point candidates. Returns
 * nothing; the result is in w. big_hex_rounds: search_range/4 for the uni form, input->search_range>>2 for the bi form (:1219). */
static void umhex_main(uwalk *w, int pic_pix_x, int pic_pix_y, int blocktype, int try_uplayer, int try_ref, int skip_to_fourth1_if_small,
                       int big_hex_rounds, int et_thred, float betaFourth_1, float betaFourth_2)
{
  jmo_umhex *u = w->u;
  const int search_range = w->search_range;
  int i, m, pos, iXMinNow, iYMinNow, tx[16], ty[16];
#define EARLY_TERMINATION \
  if ((w->min_mcost - u->pred_sad) < u->pred_sad * betaFourth_2) goto fourth_2_step; \
  else if ((w->min_mcost - u->pred_sad) < u->pred_sad * betaFourth_1) goto fourth_1_step;
  if (try_uplayer) search_one_pixel(w, pic_pix_x + (u->pred_mv_uplayer[0] / 4), pic_pix_y + (u->pred_mv_uplayer[1] / 4));
  if (try_ref) search_one_pixel(w, pic_pix_x + (u->pred_mv_ref[0] / 4), pic_pix_y + (u->pred_mv_ref[1] / 4));
  diamond(w);
  EARLY_TERMINATION
  if (skip_to_fourth1_if_small && blocktype > 6) goto fourth_1_step;

  iXMinNow = w->best_x; iYMinNow = w->best_y;                         /* sec_step: unsymmetrical cross */
  for (i = 1; i < search_range; i += 2) { search_one_pixel(w, iXMinNow + i, iYMinNow); search_one_pixel(w, iXMinNow - i, iYMinNow); }
  for (i = 1; i < (search_range / 2); i += 2) { search_one_pixel(w, iXMinNow, iYMinNow + i); search_one_pixel(w, iXMinNow, iYMinNow - i); }
  EARLY_TERMINATION
  iXMinNow = w->best_x; iYMinNow = w->best_y;
  for (pos = 1; pos < 25; pos++) search_one_pixel(w, iXMinNow + u->spx[pos], iYMinNow + u->spy[pos]);     /* 5x5 */
  EARLY_TERMINATION
  memcpy(tx, Big_Hexagon_x, sizeof(tx)); memcpy(ty, Big_Hexagon_y, sizeof(ty));
  for (i = 1; i <= big_hex_rounds; i++) {
    for (m = 0; m < 16; m++) {
      const int cx = iXMinNow + tx[m], cy = iYMinNow + ty[m];
      tx[m] += Big_Hexagon_x[m]; ty[m] += Big_Hexagon_y[m];
      search_one_pixel(w, cx, cy);
    }
    if (w->min_mcost < et_thred) return;
  }
fourth_1_step:
  for (i = 0; i < search_range; i++) {
    iXMinNow = w->best_x; iYMinNow = w->best_y;
    for (m = 0; m < 6; m++) search_one_pixel(w, iXMinNow + Hexagon_x[m], iYMinNow + Hexagon_y[m]);
    if (w->best_x == iXMinNow && w->best_y == iYMinNow) break;
  }
fourth_2_step:
  for (i = 0; i < search_range; i++) {
    iXMinNow = w->best_x; iYMinNow = w->best_y;
    for (m = 0; m < 4; m++) search_one_pixel(w, iXMinNow + Diamond_x[m], iYMinNow + Diamond_y[m]);
    if (w->best_x == iXMinNow && w->best_y == iYMinNow) break;
  }
#undef EARLY_TERMINATION
}

static void betas(const jmo_umhex *u, int blocktype, float *b1, float *b2)     /* :382-392 */
{
  if (u->pred_sad == 0) { *b1 = 0; *b2 = 0; }
  else {
    *b1 = u->bsize[blocktype] / (u->pred_sad * u->pred_sad) - u->alpha1[blocktype];
    *b2 = u->bsize[blocktype] / (u->pred_sad * u->pred_sad) - u->alpha2[blocktype];
  }
}

/* UMHEXIntegerPelBlockMotionSearch :229 */
int jmo_umhex_pel_search(jmo_umhex *u, const jmo_me_params *p, const jmo_ref *ref_pic, const jmo_pel *orig_pic, int ref, int list,
                         const short (*allmv)[JMO_MAX_REFS][9][2], int frame_ctr_b, int opix_x, int opix_y, int pic_pix_x, int pic_pix_y,
                         int blocktype, int pred_mv_x, int pred_mv_y, short *mv_x, short *mv_y, int search_range, int min_mcost, int lambda_factor)
{
  uwalk w;
  int bsx, bsy, mcost, i, j, et_thred;
  const int mb_x = pic_pix_x - opix_x, mb_y = pic_pix_y - opix_y, px2 = pic_pix_x >> 2, block_x = mb_x >> 2, block_y = mb_y >> 2;
  int *sad_prediction = u->best_cost + (size_t)(blocktype - 1) * u->w4;
  float b1 = 0, b2 = 0;
  jmo_block_size(blocktype, &bsx, &bsy);
  memset(&w, 0, sizeof(w));
  w.u = u; w.p = p; w.cur = orig_pic; w.bsx = bsx; w.bsy = bsy; w.lambda = lambda_factor;
  w.pred_x = (pic_pix_x << 2) + pred_mv_x; w.pred_y = (pic_pix_y << 2) + pred_mv_y;
  w.center_x = pic_pix_x + *mv_x; w.center_y = pic_pix_y + *mv_y; w.search_range = search_range;
  w.min_mcost = min_mcost;
  jmo_dist_from_params(p, ref_pic, &w.d);
  w.d.chroma_me = p->chroma_me ? 1 : 0;
  w.d.test8x8 = p->transform8x8_mode && blocktype <= 4;
  w.d.umv = !((w.center_x > search_range) && (w.center_x < ref_pic->W - 1 - search_range - bsx) &&
              (w.center_y > search_range) && (w.center_y < ref_pic->H - 1 - search_range - bsy));       /* :309-317 */
  et_thred = u->median_thd[blocktype];
  memset(u->mcost_state, 0, (size_t)u->stride * u->stride);                                             /* :320 */

  mcost = u_mvcost(&w, w.center_x, w.center_y);
  mcost += u_dist(&w, w.min_mcost - mcost, w.center_x, w.center_y);
  MCS(&w, w.center_y, w.center_x) = 1;
  if (mcost < w.min_mcost) { w.min_mcost = mcost; w.best_x = w.center_x; w.best_y = w.center_y; }
  diamond(&w);
  if (w.center_x != pic_pix_x || w.center_y != pic_pix_y) { search_one_pixel(&w, pic_pix_x, pic_pix_y); diamond(&w); }

  if (ref > 0 && w.min_mcost > et_thred && sad_prediction[px2] < u->multi_ref_thd[blocktype]) goto terminate_step;
  if (w.min_mcost < et_thred) goto terminate_step;
  umhex_setup(u, ref, list, block_y, block_x, blocktype, allmv, p->is_b_slice, frame_ctr_b, opix_x, opix_y);
  et_thred = u->big_hex_thd[blocktype];
  betas(u, blocktype, &b1, &b2);
  umhex_main(&w, pic_pix_x, pic_pix_y, blocktype, blocktype > 1, u->pred_mv_ref_flag == 1, 1, search_range / 4, et_thred, b1, b2);

terminate_step:
  for (i = 0; i < (bsx >> 2); i++) for (j = 0; j < (bsy >> 2); j++) {
    if (list == 0) {
      RC(ref, blocktype, block_y + j, block_x + i) = w.min_mcost;
      if (ref == 0) LC(u->l_cost[0], blocktype, (opix_y >> 2) + block_y + j, (opix_x >> 2) + block_x + i) = w.min_mcost;
    } else LC(u->l_cost[1], blocktype, (opix_y >> 2) + block_y + j, (opix_x >> 2) + block_x + i) = w.min_mcost;
  }
  if (ref == 0 || sad_prediction[px2] > w.min_mcost) sad_prediction[px2] = w.min_mcost;
  *mv_x = (short)(w.best_x - pic_pix_x); *mv_y = (short)(w.best_y - pic_pix_y);
  return w.min_mcost;
}

/* UMHEXBipredIntegerPelBlockMotionSearch :916. bipred_mv_l1 = (list ? img->bipred_mv1 : img->bipred_mv2)[block_y][block_x][1][0][blocktype].
 * The metric is SAD whatever MEDistortionFPel says (:979, :986). */
int jmo_umhex_bipred_search(jmo_umhex *u, jmo_bipred *b, const jmo_pel *cur_pic, int list, const short bipred_mv_l1[2], int frame_ctr_b,
                            int opix_x, int opix_y, int pic_pix_x, int pic_pix_y, int blocktype, int pred_mv_x1, int pred_mv_y1, int pred_mv_x2, int pred_mv_y2,
                            short *mv_x, short *mv_y, const short *s_mv_x, const short *s_mv_y, int search_range, int min_mcost, int lambda_factor)
{
  uwalk w;
  int bsx, bsy, mcost, i, j, et_thred;
  const int mb_x = pic_pix_x - opix_x, mb_y = pic_pix_y - opix_y, block_x = mb_x >> 2, block_y = mb_y >> 2;
  const short center2_x = (short)(pic_pix_x + *mv_x), center2_y = (short)(pic_pix_y + *mv_y);
  const short center1_x = (short)(pic_pix_x + *s_mv_x), center1_y = (short)(pic_pix_y + *s_mv_y);
  const int W = b->ref1->W, H = b->ref1->H;
  float b1 = 0, b2 = 0;
  jmo_block_size(blocktype, &bsx, &bsy);
  memset(&w, 0, sizeof(w));
  w.u = u; w.b = b; w.cur = cur_pic; w.bsx = bsx; w.bsy = bsy; w.lambda = lambda_factor;
  w.pred_x = (pic_pix_x << 2) + pred_mv_x2; w.pred_y = (pic_pix_y << 2) + pred_mv_y2;
  w.c1x = center1_x; w.c1y = center1_y;
  w.fixed_cost = jmo_mv_cost(lambda_factor, center1_x << 2, center1_y << 2, (pic_pix_x << 2) + pred_mv_x1, (pic_pix_y << 2) + pred_mv_y1);
  w.center_x = center2_x; w.center_y = center2_y; w.search_range = search_range;
  w.best_x = center2_x; w.best_y = center2_y; w.min_mcost = min_mcost;
  et_thred = u->median_thd[blocktype];
  b->umv2 = !((center2_x > search_range) && (center2_x < W - 1 - search_range - bsx) && (center2_y > search_range) && (center2_y < H - 1 - search_range - bsy));  /* :1022 */
  b->umv1 = !((center1_y > search_range) && (center1_y < H - 1 - search_range - bsy));                                                                           /* :1033: y only */
  memset(u->mcost_state, 0, (size_t)(2 * search_range + 1) * (2 * search_range + 1));       /* :1045: a PREFIX of the map (its rows are 2*input->search_range+1 long) */

  mcost = u_mvcost(&w, center2_x, center2_y);
  mcost += u_dist(&w, JMO_INT_MAX, center2_x, center2_y);
  MCS(&w, center2_y, center2_x) = 1;
  if (mcost < w.min_mcost) { w.min_mcost = mcost; w.best_x = center2_x; w.best_y = center2_y; }
  diamond(&w);
  if (center2_x != pic_pix_x || center2_y != pic_pix_y) { search_one_pixel(&w, pic_pix_x, pic_pix_y); diamond(&w); }
  if (w.min_mcost < et_thred) goto terminate_step;
  {
    const int N_Bframe = u->cfg.successive_bframe, n_Bframe = frame_ctr_b % (N_Bframe + 1);
    if (list == 0) {
      u->pred_mv_ref[0] = (int)(bipred_mv_l1[0] * (-n_Bframe) / (N_Bframe - n_Bframe + 1.0f));
      u->pred_mv_ref[1] = (int)(bipred_mv_l1[1] * (-n_Bframe) / (N_Bframe - n_Bframe + 1.0f));
    }
    u->pred_sad = imin_(imin_(u->sad_a, u->sad_b), u->sad_c);
    et_thred = u->big_hex_thd[blocktype];
    betas(u, blocktype, &b1, &b2);
  }
  umhex_main(&w, pic_pix_x, pic_pix_y, blocktype, 0, list == 0, 0, u->cfg.search_range >> 2, et_thred, b1, b2);
terminate_step:
  for (i = 0; i < (bsx >> 2); i++) for (j = 0; j < (bsy >> 2); j++)
    LC(u->l_cost_bipred[list != 0], blocktype, (opix_y >> 2) + block_y + j, (opix_x >> 2) + block_x + i) = w.min_mcost;
  *mv_x = (short)(w.best_x - pic_pix_x); *mv_y = (short)(w.best_y - pic_pix_y);
  return w.min_mcost;
}

/* UMHEXSubPelBlockMotionSearch :562 (lambda_factor = lambda[Q_PEL], the metric is MEDistortionQPel for every position) */
int jmo_umhex_subpel_search(jmo_umhex *u, const jmo_me_params *p, const jmo_ref *ref_pic, const jmo_pel *orig_pic, int pic_pix_x, int pic_pix_y,
                            int blocktype, int pred_mv_x, int pred_mv_y, short *mv_x, short *mv_y, int min_mcost, int lambda_factor)
{
  const int start_hp = (p->chroma_me == 1 || p->metric[JMO_F_PEL] != p->metric[JMO_H_PEL]) ? 0 : 1;
  const int pic4_x = (pic_pix_x + JMO_PAD) << 2, pic4_y = (pic_pix_y + JMO_PAD) << 2, srd = 3;
  int bsx, bsy, mcost, cx, cy, i, m, currmv_x = 0, currmv_y = 0, iXMinNow, iYMinNow, abort_search, pfx, pfy;
  short max_x4, max_y4;
  jmo_dist d;
  jmo_block_size(blocktype, &bsx, &bsy);
  max_x4 = (short)((ref_pic->W - bsx + 2 * JMO_PAD) << 2); max_y4 = (short)((ref_pic->H - bsy + 2 * JMO_PAD) << 2);      /* :590-591: short */
  jmo_dist_from_params(p, ref_pic, &d);
  d.chroma_me = (p->chroma_me == 2) ? 1 : 0;
  d.test8x8 = p->transform8x8_mode && blocktype <= 4;
  d.umv = !((pic4_x + *mv_x > 1) && (pic4_x + *mv_x < max_x4 - 1) && (pic4_y + *mv_y > 1) && (pic4_y + *mv_y < max_y4 - 1));
  pfx = (pred_mv_x - *mv_x) % 4; pfy = (pred_mv_y - *mv_y) % 4;
  memset(u->search_state, 0, sizeof(u->search_state));
#define SS(y, x) u->search_state[(y) - *mv_y + srd][(x) - *mv_x + srd]
#define EVAL(X, Y) (jmo_mv_cost(lambda_factor, (X), (Y), pred_mv_x, pred_mv_y))
  if (!start_hp) {
    cx = *mv_x; cy = *mv_y;
    mcost = EVAL(cx, cy);
    mcost += jmo_uni_pred(p, JMO_Q_PEL, &d, orig_pic, bsy, bsx, min_mcost - mcost, cx + pic4_x, cy + pic4_y);
    SS(cy, cx) = 1;
    if (mcost < min_mcost) { min_mcost = mcost; currmv_x = cx; currmv_y = cy; }
  } else { SS(*mv_y, *mv_x) = 1; currmv_x = *mv_x; currmv_y = *mv_y; }
  if (pfx != 0 || pfy != 0) {
    cx = *mv_x + pfx; cy = *mv_y + pfy;
    mcost = EVAL(cx, cy);
    mcost += jmo_uni_pred(p, JMO_Q_PEL, &d, orig_pic, bsy, bsx, min_mcost - mcost, cx + pic4_x, cy + pic4_y);
    SS(cy, cx) = 1;
    if (mcost < min_mcost) { min_mcost = mcost; currmv_x = cx; currmv_y = cy; }
  }
  iXMinNow = currmv_x; iYMinNow = currmv_y;
  for (i = 0; i < srd; i++) {
    abort_search = 1;
    for (m = 0; m < 4; m++) {
      cx = iXMinNow + Diamond_x[m]; cy = iYMinNow + Diamond_y[m];
      if (iabs_(cx - *mv_x) <= srd && iabs_(cy - *mv_y) <= srd && !SS(cy, cx)) {
        mcost = EVAL(cx, cy);
        mcost += jmo_uni_pred(p, JMO_Q_PEL, &d, orig_pic, bsy, bsx, min_mcost - mcost, cx + pic4_x, cy + pic4_y);
        SS(cy, cx) = 1;
        if (mcost < min_mcost) { min_mcost = mcost; currmv_x = cx; currmv_y = cy; abort_search = 0; }
      }
    }
    iXMinNow = currmv_x; iYMinNow = currmv_y;
    if (abort_search) break;
  }
#undef SS
#undef EVAL
  *mv_x = (short)currmv_x; *mv_y = (short)currmv_y;
  return min_mcost;
}

/* UMHEXSetMotionVectorPredictor :1298 (non-MBAFF). nb: A, B, C, D as getLuma4x4Neighbour returns them, with their 4x4 positions.
 * umhex_blocktype / bipred_flag: the globals mv-search.c sets before the call (:631-632, :876). search_range is written only when
 * the dynamic search range is on. */
void jmo_umhex_set_mv_predictor(jmo_umhex *u, short pmv[2], const jmo_umhex_nbr *nb_in, int ref_frame, int list, int block_x, int block_y,
                                int blockshape_x, int blockshape_y, int umhex_blocktype, int bipred_flag, const int (*blocktype_lut)[4],
                                int *search_range)
{
  const int mb_x = 4 * block_x, mb_y = 4 * block_y, R = u->cfg.search_range;
  jmo_umhex_nbr nb = *nb_in;
  int *cost = (bipred_flag ? u->l_cost_bipred : u->l_cost)[list == 1];
  int mv_a, mv_b, mv_c, pred_vec = 0, type = 0 /* MEDIAN */, rL, rU, rUR, hv, dsr_tmp[2] = {0, 0};
  enum { A, B, C, D };
  u->sad_a = u->sad_b = u->sad_c = u->sad_d = 0;
  if (mb_y > 0) {
    if (mb_x < 8) {
      if (mb_y == 8) { if (blockshape_x == 16) nb.available[C] = 0; }
      else if (mb_x + blockshape_x == 8) nb.available[C] = 0;
    } else if (mb_x + blockshape_x == 16) nb.available[C] = 0;
  }
  if (!nb.available[C]) { nb.available[C] = nb.available[D]; nb.ref[C] = nb.ref[D]; nb.mv[C][0] = nb.mv[D][0]; nb.mv[C][1] = nb.mv[D][1]; nb.pos_x[C] = nb.pos_x[D]; nb.pos_y[C] = nb.pos_y[D]; }
  rL = nb.available[A] ? nb.ref[A] : -1; rU = nb.available[B] ? nb.ref[B] : -1; rUR = nb.available[C] ? nb.ref[C] : -1;
  if (rL == ref_frame && rU != ref_frame && rUR != ref_frame) type = 1;          /* L */
  else if (rL != ref_frame && rU == ref_frame && rUR != ref_frame) type = 2;     /* U */
  else if (rL != ref_frame && rU != ref_frame && rUR == ref_frame) type = 3;     /* UR */
  if (blockshape_x == 8 && blockshape_y == 16) {
    if (mb_x == 0) { if (rL == ref_frame) type = 1; } else { if (rUR == ref_frame) type = 3; }
  } else if (blockshape_x == 16 && blockshape_y == 8) {
    if (mb_y == 0) { if (rU == ref_frame) type = 2; } else { if (rL == ref_frame) type = 1; }
  }
  if (u->cfg.dsr == 1 || u->cfg.bipred_me == 1) {
#define NC(k) LC(cost, umhex_blocktype, nb.pos_y[k], nb.pos_x[k])
    u->sad_a = nb.available[A] ? NC(A) : 0;
    u->sad_b = nb.available[B] ? NC(B) : 0;
    u->sad_d = nb.available[D] ? NC(D) : 0;
    u->sad_c = nb.available[C] ? NC(C) : u->sad_d;
#undef NC
  }
  for (hv = 0; hv < 2; hv++) {
    mv_a = nb.available[A] ? nb.mv[A][hv] : 0; mv_b = nb.available[B] ? nb.mv[B][hv] : 0; mv_c = nb.available[C] ? nb.mv[C][hv] : 0;
    switch (type) {
    case 0: pred_vec = !(nb.available[B] || nb.available[C]) ? mv_a : mv_a + mv_b + mv_c - imin_(mv_a, imin_(mv_b, mv_c)) - imax_(mv_a, imax_(mv_b, mv_c)); break;
    case 1: pred_vec = mv_a; break;
    case 2: pred_vec = mv_b; break;
    default: pred_vec = mv_c; break;
    }
    pmv[hv] = (short)pred_vec;
    if (u->cfg.dsr) {
      const int avail = nb.available[A] + nb.available[B] + nb.available[C];
      if (avail < 2) dsr_tmp[hv] = R;
      else {
        const int mx = imax_(iabs_(mv_a), imax_(iabs_(mv_b), iabs_(mv_c))), sum = iabs_(mv_a) + iabs_(mv_b) + iabs_(mv_c);
        const int small_range = sum == 0 ? (R + 4) >> 3 : sum > 3 ? (R + 2) >> 2 : (3 * R + 8) >> 4;
        dsr_tmp[hv] = imin_(R, imax_(small_range, mx << 1));
        if (imax_(u->sad_a, imax_(u->sad_b, u->sad_c)) > u->dsr_thd[umhex_blocktype]) dsr_tmp[hv] = R;
      }
    }
  }
  if (u->cfg.dsr) {
    const int nr = imax_(dsr_tmp[0], dsr_tmp[1]);
    if (u->cfg.full_search == 2) *search_range = nr;
    else if (u->cfg.full_search == 1) *search_range = nr / (imin_(ref_frame, 1) + 1);
    else *search_range = nr / ((imin_(ref_frame, 1) + 1) * imin_(2, blocktype_lut[(blockshape_y >> 2) - 1][(blockshape_x >> 2) - 1]));
  }
}
