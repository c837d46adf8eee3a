/*
 * swap_oracle.c -- TEST INFRASTRUCTURE, container only (needs /root/reference headers at build time).
 *
 * Links the real JM (oracle/_ref/libjm.so, built -fPIC so every global call goes through the PLT/GOT)
 * and DEFINES the hot-path symbols itself, so JM's own callers -- including address-taken uses such as
 * computeUniPred[i] = computeSAD and pDCT_4x4 = dct_4x4 -- land here. Each definition marshals JM's
 * globals into the explicit arguments of the oracle restatement (oracle/jmo.h) and writes the results
 * back where JM expects them. If the restatement is right, the bitstream and the reconstruction JM writes
 * stay byte-identical to the unmodified encoder's for every cfg x SearchMode (tests/test_oracle_swap.py).
 *
 * JMO_SWAP (env, hex bitmask, default all) selects the groups that are swapped; a cleared bit forwards to
 * JM's original through dlsym(RTLD_NEXT):
 *   0x01 interpolation   0x02 SAD/SATD kernels   0x04 full/sub-pel search   0x08 fast full search
 *   0x10 dct_4x4/16x16   0x20 dct_8x8            0x40 dct_chroma            0x80 transform primitives
 *   0x100 bi-predictive full-pel + sub-pel search (FullPelBlockMotionBiPred, SubPelBlockSearchBiPred)
 *   0x200 low-complexity mode-decision costs (TransformDecision, GetSkipCostMB)
 *   0x400 in-loop deblocking filter (DeblockFrame)
 *   0x1000 UMHexagonS (predictor + dynamic search range, integer / bi-predictive / sub-pel walks, intra-block SAD flags; frame pictures)
 *   0x800 EPZS walkers (EPZSInit/SliceInit state, EPZSPel/BiPred/SubPel/SubPelBiPred searches; frame pictures, EPZSSubPelGrid 0)
 * JMO_SWAP_STATS=1 prints per-symbol call counts at exit.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#include "global.h"
#include "mbuffer.h"
#include "image.h"
#include "refbuf.h"
#include "me_distortion.h"
#include "q_matrix.h"
#include "q_offsets.h"
#include "mb_access.h"
#include "me_epzs.h"
#include "me_umhex.h"

#include "jmo.h"

extern int jm_main(int argc, char **argv);
extern int ****ptLevelOffset4x4;
extern void SetMotionVectorPredictor(short pmv[2], char **refPic, short ***tmp_mv, short ref_frame, int list,
                                     int block_x, int block_y, int blockshape_x, int blockshape_y);
extern const int LEVELMVLIMIT[17][6];

static unsigned swap_mask = 0x1fff;
static long n_calls[25], n_served[25];       /* calls seen / calls answered by the oracle (walker groups) */
enum { C_LUMA, C_CHROMA, C_SAD, C_SATD, C_FULL, C_SUB, C_FAST, C_D4, C_D8, C_D16, C_DCR, C_PRIM, C_BIFULL, C_BISUB, C_TDEC, C_SKIPC, C_DEBLOCK, C_EPZS_PEL, C_EPZS_SUB, C_EPZS_BI, C_EPZS_BISUB, C_UM_PRED, C_UM_PEL, C_UM_SUB, C_UM_BI };
static const char *c_names[] = { "getSubImagesLuma", "getSubImagesChroma", "computeSAD*", "computeSATD*",
  "FullPelBlockMotionSearch", "SubPelBlockMotionSearch", "FastFullPelBlockMotionSearch", "dct_4x4",
  "dct_8x8", "dct_16x16", "dct_chroma", "transform primitives", "FullPelBlockMotionBiPred", "SubPelBlockSearchBiPred", "TransformDecision", "GetSkipCostMB", "DeblockFrame",
  "EPZSPelBlockMotionSearch", "EPZSSubPelBlockMotionSearch", "EPZSBiPredBlockMotionSearch", "EPZSSubPelBlockSearchBiPred",
  "UMHEXSetMotionVectorPredictor", "UMHEXIntegerPelBlockMotionSearch", "UMHEXSubPelBlockMotionSearch", "UMHEXBipredIntegerPelBlockMotionSearch" };

static void *next_sym(const char *name)
{
  void *p = dlsym(RTLD_NEXT, name);
  if (!p) { fprintf(stderr, "swap_oracle: cannot find original %s\n", name); exit(97); }
  return p;
}

static void print_stats(void)
{
  int i;
  if (!getenv("JMO_SWAP_STATS")) return;
  fprintf(stderr, "swap_oracle: mask=0x%x\n", swap_mask);
  for (i = 0; i <= C_UM_BI; i++) fprintf(stderr, "  %-30s %ld served %ld\n", c_names[i], n_calls[i], n_served[i]);
}

int main(int argc, char **argv)
{
  const char *m = getenv("JMO_SWAP");
  if (m) swap_mask = (unsigned)strtoul(m, NULL, 16);
  atexit(print_stats);
  return jm_main(argc, argv);
}

/* ------------------------------------------------------------------ helpers */

static void fill_ref(jmo_ref *r, StorablePicture *s)
{
  int y, x, k;
  jmo_ref_init(r, s->size_x, s->size_y, img->yuv_format, NULL, NULL, NULL);
  for (y = 0; y < 4; y++) for (x = 0; x < 4; x++) r->luma[y * 4 + x] = s->imgY_sub[y][x][0];
  if (s->imgUV_sub && img->yuv_format != YUV400)
    for (k = 0; k < 2; k++)
      for (y = 0; y < r->cg.sub_y; y++) for (x = 0; x < r->cg.sub_x; x++)
        r->cr[k][y * r->cg.sub_x + x] = s->imgUV_sub[k][y][x][0];
  if (r->width_pad != s->size_x_pad || r->height_pad != s->size_y_pad ||
      (img->yuv_format != YUV400 && (r->width_pad_cr != s->size_x_cr_pad || r->height_pad_cr != s->size_y_cr_pad))) {
    fprintf(stderr, "swap_oracle: pad geometry mismatch\n"); exit(98);
  }
}

/* JM's search functions leave these globals set (me_fullsearch.c:80-107, :378-403; me_fullfast.c:517-547)
 * and later code depends on it: OneComponentLumaPrediction reads width_pad/height_pad through UMVLine4X
 * BEFORE assigning them (macroblock.c:816-819). A drop-in must reproduce the side effect. */
static void jm_side_effects(StorablePicture *rp)
{
  ref_pic_sub.luma = rp->p_curr_img_sub;
  width_pad = rp->size_x_pad; height_pad = rp->size_y_pad;
  if (ChromaMEEnable) {
    ref_pic_sub.crcb[0] = rp->imgUV_sub[0]; ref_pic_sub.crcb[1] = rp->imgUV_sub[1];
    width_pad_cr = rp->size_x_cr_pad; height_pad_cr = rp->size_y_cr_pad;
  }
}

/* ------------------------------------------------------------------ 0x01 interpolation */

void getSubImagesLuma(StorablePicture *s)
{
  static void (*orig)(StorablePicture *);
  n_calls[C_LUMA]++;
  if (!(swap_mask & 0x01)) { if (!orig) orig = next_sym("getSubImagesLuma"); orig(s); return; }
  {
    const int W = s->size_x, H = s->size_y, Wp = s->size_x_padded, Hp = s->size_y_padded;
    jmo_pel *in = malloc(sizeof(jmo_pel) * W * H), *out = malloc(sizeof(jmo_pel) * 16 * (size_t)Wp * Hp);
    int j, y, x;
    for (j = 0; j < H; j++) memcpy(in + (size_t)j * W, s->p_curr_img[j], sizeof(jmo_pel) * W);
    jmo_interp_luma(in, W, H, W, img->max_imgpel_value, out);
    for (y = 0; y < 4; y++) for (x = 0; x < 4; x++)
      for (j = 0; j < Hp; j++)
        memcpy(s->p_curr_img_sub[y][x][j], out + ((size_t)(y * 4 + x) * Hp + j) * Wp, sizeof(jmo_pel) * Wp);
    free(in); free(out);
  }
}

void getSubImagesChroma(StorablePicture *s)
{
  static void (*orig)(StorablePicture *);
  n_calls[C_CHROMA]++;
  if (!(swap_mask & 0x01)) { if (!orig) orig = next_sym("getSubImagesChroma"); orig(s); return; }
  {
    jmo_chroma_geom g;
    int Wc = s->size_x_cr, Hc = s->size_y_cr, Wcp, Hcp, uv, j, p;
    jmo_pel *in, *out;
    jmo_chroma_geometry(img->yuv_format, &g);
    Wcp = Wc + 2 * g.pad_x; Hcp = Hc + 2 * g.pad_y;
    in = malloc(sizeof(jmo_pel) * Wc * Hc);
    out = malloc(sizeof(jmo_pel) * (size_t)g.sub_x * g.sub_y * Wcp * Hcp);
    for (uv = 0; uv < 2; uv++) {
      for (j = 0; j < Hc; j++) memcpy(in + (size_t)j * Wc, s->imgUV[uv][j], sizeof(jmo_pel) * Wc);
      /* JM leaves the last row/column untouched: start from what is there (calloc zeros) */
      for (p = 0; p < g.sub_x * g.sub_y; p++)
        for (j = 0; j < Hcp; j++)
          memcpy(out + ((size_t)p * Hcp + j) * Wcp, s->imgUV_sub[uv][p / g.sub_x][p % g.sub_x][j], sizeof(jmo_pel) * Wcp);
      jmo_interp_chroma(in, Wc, Hc, Wc, img->yuv_format, out);
      for (p = 0; p < g.sub_x * g.sub_y; p++)
        for (j = 0; j < Hcp; j++)
          memcpy(s->imgUV_sub[uv][p / g.sub_x][p % g.sub_x][j], out + ((size_t)p * Hcp + j) * Wcp, sizeof(jmo_pel) * Wcp);
    }
    free(in); free(out);
  }
}

/* ------------------------------------------------------------------ 0x02 distortion kernels */

/* the kernels see only pointers to plane sets: rebuild the geometry from JM's globals (me_distortion.c:35-56) */
static void fill_dist(jmo_dist *d, jmo_ref *r)
{
  int y, x, k;
  memset(d, 0, sizeof(*d));
  /* size from the globals the kernels themselves use */
  jmo_ref_init(r, img_padded_size_x - 2 * IMG_PAD_SIZE, img->height, img->yuv_format, NULL, NULL, NULL);
  r->width_pad = width_pad; r->height_pad = height_pad;
  for (y = 0; y < 4; y++) for (x = 0; x < 4; x++) r->luma[y * 4 + x] = ref_pic_sub.luma[y][x][0];
  if (ChromaMEEnable) {
    r->width_pad_cr = width_pad_cr; r->height_pad_cr = height_pad_cr;
    for (k = 0; k < 2; k++)
      for (y = 0; y < r->cg.sub_y; y++) for (x = 0; x < r->cg.sub_x; x++)
        r->cr[k][y * r->cg.sub_x + x] = ref_pic_sub.crcb[k][y][x][0];
  }
  d->ref = r;
  d->umv = ref_access_method;
  d->chroma_me = ChromaMEEnable;
  d->chroma_me_weight = input->ChromaMEWeight;
  d->test8x8 = test8x8transform;
  d->max_val = img->max_imgpel_value; d->max_val_uv = img->max_imgpel_value_comp[1];
  d->weight_luma = weight_luma; d->offset_luma = offset_luma;
  d->wp_luma_round = wp_luma_round; d->luma_log_weight_denom = luma_log_weight_denom;
  d->weight_cr[0] = weight_cr[0]; d->weight_cr[1] = weight_cr[1];
  d->offset_cr[0] = offset_cr[0]; d->offset_cr[1] = offset_cr[1];
  d->wp_chroma_round = wp_chroma_round; d->chroma_log_weight_denom = chroma_log_weight_denom;
}

#define DIST_SWAP(NAME, ORFN, CNT)                                                                   \
  int NAME(imgpel *src_pic, int bsy, int bsx, int min_mcost, int cand_x, int cand_y)                 \
  {                                                                                                  \
    static int (*orig)(imgpel *, int, int, int, int, int);                                           \
    jmo_dist d; jmo_ref r;                                                                           \
    n_calls[CNT]++;                                                                                  \
    if (!(swap_mask & 0x02)) { if (!orig) orig = next_sym(#NAME); return orig(src_pic, bsy, bsx, min_mcost, cand_x, cand_y); } \
    fill_dist(&d, &r);                                                                               \
    return ORFN(&d, src_pic, bsy, bsx, min_mcost, cand_x, cand_y);                                   \
  }
DIST_SWAP(computeSAD,    jmo_sad,     C_SAD)
DIST_SWAP(computeSADWP,  jmo_sad_wp,  C_SAD)
DIST_SWAP(computeSATD,   jmo_satd,    C_SATD)
DIST_SWAP(computeSATDWP, jmo_satd_wp, C_SATD)

/* ------------------------------------------------------------------ 0x04 full-pel / sub-pel search */

static void fill_me_params(jmo_me_params *p, int list, int ref, int list_offset)
{
  memset(p, 0, sizeof(*p));
  p->rdopt = input->rdopt;
  p->is_b_slice = (img->type == B_SLICE);
  p->chroma_me = input->ChromaMEEnable;
  p->chroma_me_weight = input->ChromaMEWeight;
  p->transform8x8_mode = input->Transform8x8Mode;
  p->metric[0] = input->MEErrorMetric[0]; p->metric[1] = input->MEErrorMetric[1]; p->metric[2] = input->MEErrorMetric[2];
  p->apply_weights = ((active_pps->weighted_pred_flag && (img->type == P_SLICE || img->type == SP_SLICE)) ||
                      (active_pps->weighted_bipred_idc && (img->type == B_SLICE))) && input->UseWeightedReferenceME;
  p->max_val = img->max_imgpel_value; p->max_val_uv = img->max_imgpel_value_comp[1];
  p->level_mv_min = LEVELMVLIMIT[img->LevelIndex][0]; p->level_mv_max = LEVELMVLIMIT[img->LevelIndex][1];
  if (p->apply_weights) {
    p->weight_luma = wp_weight[list + list_offset][ref][0];
    p->offset_luma = wp_offset[list + list_offset][ref][0];
    p->weight_cr[0] = wp_weight[list + list_offset][ref][1]; p->weight_cr[1] = wp_weight[list + list_offset][ref][2];
    p->offset_cr[0] = wp_offset[list + list_offset][ref][1]; p->offset_cr[1] = wp_offset[list + list_offset][ref][2];
  }
  p->wp_luma_round = wp_luma_round; p->luma_log_weight_denom = luma_log_weight_denom;
  p->wp_chroma_round = wp_chroma_round; p->chroma_log_weight_denom = chroma_log_weight_denom;
}

int FullPelBlockMotionSearch(imgpel *orig_pic, short ref, int list, int pic_pix_x, int pic_pix_y, int blocktype,
                             short pred_mv_x, short pred_mv_y, short *mv_x, short *mv_y, int search_range,
                             int min_mcost, int lambda_factor)
{
  static int (*orig)(imgpel *, short, int, int, int, int, short, short, short *, short *, int, int, int);
  n_calls[C_FULL]++;
  if (!(swap_mask & 0x04) || getenv("JMO_NO_FULL")) {
    if (!orig) orig = next_sym("FullPelBlockMotionSearch");
    return orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, mv_x, mv_y, search_range, min_mcost, lambda_factor);
  }
  {
    int list_offset = img->mb_data[img->current_mb_nr].list_offset;
    jmo_me_params p; jmo_ref r;
    fill_me_params(&p, list, ref, list_offset);
    fill_ref(&r, listX[list + list_offset][ref]);
    jm_side_effects(listX[list + list_offset][ref]);
    if ((ChromaMEEnable != 0) != (p.chroma_me != 0)) { fprintf(stderr, "swap_oracle: ChromaMEEnable state\n"); exit(98); }
    if (getenv("JMO_SWAP_VERIFY")) {
      short ox = *mv_x, oy = *mv_y, jx = *mv_x, jy = *mv_y; int c1, c2;
      if (!orig) orig = next_sym("FullPelBlockMotionSearch");
      c1 = orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, &jx, &jy, search_range, min_mcost, lambda_factor);
      c2 = jmo_fullpel_search(&p, &r, orig_pic, ref == 0, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, &ox, &oy, search_range, min_mcost, lambda_factor);
      if (c1 != c2 || jx != ox || jy != oy)
        fprintf(stderr, "FULLPEL MISMATCH ref=%d list=%d pix=(%d,%d) bt=%d pred=(%d,%d) in=(%d,%d) R=%d min=%d lam=%d: jm=(%d,%d,%d) or=(%d,%d,%d)\n",
                ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, *mv_x, *mv_y, search_range, min_mcost, lambda_factor, jx, jy, c1, ox, oy, c2);
      *mv_x = jx; *mv_y = jy; return c1;
    }
    return jmo_fullpel_search(&p, &r, orig_pic, ref == 0, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y,
                              mv_x, mv_y, search_range, min_mcost, lambda_factor);
  }
}

int SubPelBlockMotionSearch(imgpel *orig_pic, short ref, int list, int pic_pix_x, int pic_pix_y, int blocktype,
                            short pred_mv_x, short pred_mv_y, short *mv_x, short *mv_y, int search_pos2,
                            int search_pos4, int min_mcost, int *lambda)
{
  static int (*orig)(imgpel *, short, int, int, int, int, short, short, short *, short *, int, int, int, int *);
  n_calls[C_SUB]++;
  if (!(swap_mask & 0x04) || getenv("JMO_NO_SUB")) {
    if (!orig) orig = next_sym("SubPelBlockMotionSearch");
    return orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, mv_x, mv_y, search_pos2, search_pos4, min_mcost, lambda);
  }
  {
    int list_offset = img->mb_data[img->current_mb_nr].list_offset;
    jmo_me_params p; jmo_ref r;
    fill_me_params(&p, list, ref, list_offset);
    fill_ref(&r, listX[list + list_offset][ref]);
    jm_side_effects(listX[list + list_offset][ref]);
    if ((ChromaMEEnable != 0) != (p.chroma_me == 2)) { fprintf(stderr, "swap_oracle: ChromaMEEnable state (subpel)\n"); exit(98); }
    /* JM's kernels read the global test8x8transform; the oracle derives it like mv-search.c:640 */
    if (test8x8transform != (input->Transform8x8Mode && blocktype <= 4)) { fprintf(stderr, "swap_oracle: test8x8transform state\n"); exit(98); }
    if (getenv("JMO_SWAP_VERIFY")) {
      short ox = *mv_x, oy = *mv_y, jx = *mv_x, jy = *mv_y; int c1, c2;
      if (!orig) orig = next_sym("SubPelBlockMotionSearch");
      c1 = orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, &jx, &jy, search_pos2, search_pos4, min_mcost, lambda);
      c2 = jmo_subpel_search(&p, &r, orig_pic, ref == 0, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, &ox, &oy, search_pos2, search_pos4, min_mcost, lambda);
      if (c1 != c2 || jx != ox || jy != oy)
        fprintf(stderr, "SUBPEL MISMATCH ref=%d list=%d pix=(%d,%d) bt=%d pred=(%d,%d) in=(%d,%d) min=%d lam=%d,%d: jm=(%d,%d,%d) or=(%d,%d,%d) t8=%d\n",
                ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, *mv_x, *mv_y, min_mcost, lambda[1], lambda[2], jx, jy, c1, ox, oy, c2, test8x8transform);
      *mv_x = jx; *mv_y = jy; return c1;
    }
    return jmo_subpel_search(&p, &r, orig_pic, ref == 0, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y,
                             mv_x, mv_y, search_pos2, search_pos4, min_mcost, lambda);
  }
}

/* ------------------------------------------------------------------ 0x100 bi-predictive search */

/* "1" = fixed block on listX[list][ref], "2" = swept candidate on listX[list^1][0] (me_fullsearch.c:206-207) */
static void fill_bipred(jmo_bipred *b, jmo_ref *r1, jmo_ref *r2, short ref, int list, int blocktype)
{
  int list_offset = img->mb_data[img->current_mb_nr].list_offset;
  int apply_weights = (active_pps->weighted_bipred_idc > 0);
  short offset1 = (apply_weights ? (list == 0 ? wp_offset[list_offset][ref][0] : wp_offset[list_offset + 1][0][ref]) : 0);
  short offset2 = (apply_weights ? (list == 0 ? wp_offset[list_offset + 1][ref][0] : wp_offset[list_offset][0][ref]) : 0);
  StorablePicture *p1 = listX[list + list_offset][ref], *p2 = listX[(list ^ 1) + list_offset][0];
  memset(b, 0, sizeof(*b));
  fill_ref(r1, p1); fill_ref(r2, p2);
  /* JM clamps BOTH pictures with ref1's pad limits (the globals width_pad/height_pad, :213-214) */
  r2->width_pad = r1->width_pad; r2->height_pad = r1->height_pad;
  b->ref1 = r1; b->ref2 = r2;
  b->test8x8 = test8x8transform; b->max_val = img->max_imgpel_value;
  b->apply_weights = apply_weights;
  if (apply_weights) {
    b->weight1 = list == 0 ? wbp_weight[list_offset][ref][0][0] : wbp_weight[list_offset + LIST_1][0][ref][0];
    b->weight2 = list == 0 ? wbp_weight[list_offset + LIST_1][ref][0][0] : wbp_weight[list_offset][0][ref][0];
    b->offset_bi = (offset1 + offset2 + 1) >> 1;
  } else { b->weight1 = b->weight2 = 1 << luma_log_weight_denom; b->offset_bi = 0; }
  b->wp_luma_round = wp_luma_round; b->luma_log_weight_denom = luma_log_weight_denom;
  b->metric[0] = input->MEErrorMetric[0]; b->metric[1] = input->MEErrorMetric[1]; b->metric[2] = input->MEErrorMetric[2];
  b->start_hp = start_me_refinement_hp; b->start_qp = start_me_refinement_qp;
  /* side effects later code may read (as jm_side_effects) */
  ref_pic1_sub.luma = p1->p_curr_img_sub; ref_pic2_sub.luma = p2->p_curr_img_sub;
  width_pad = p1->size_x_pad; height_pad = p1->size_y_pad;
  (void)blocktype;
}

static int bipred_swappable(void)
{
  return (swap_mask & 0x100) && !ChromaMEEnable && input->MEErrorMetric[0] != ERROR_SSE && input->MEErrorMetric[1] != ERROR_SSE &&
         input->MEErrorMetric[2] != ERROR_SSE;
}

int FullPelBlockMotionBiPred(imgpel *orig_pic, short ref, int list, int pic_pix_x, int pic_pix_y, int blocktype,
                             short pred_mv_x1, short pred_mv_y1, short pred_mv_x2, short pred_mv_y2,
                             short *mv_x, short *mv_y, short *s_mv_x, short *s_mv_y, int search_range, int min_mcost, int lambda_factor)
{
  static int (*orig)(imgpel *, short, int, int, int, int, short, short, short, short, short *, short *, short *, short *, int, int, int);
  n_calls[C_BIFULL]++;
  if (!bipred_swappable()) {
    if (!orig) orig = next_sym("FullPelBlockMotionBiPred");
    return orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x1, pred_mv_y1, pred_mv_x2, pred_mv_y2, mv_x, mv_y, s_mv_x, s_mv_y, search_range, min_mcost, lambda_factor);
  }
  {
    jmo_bipred b; jmo_ref r1, r2;
    fill_bipred(&b, &r1, &r2, ref, list, blocktype);
    if (getenv("JMO_SWAP_VERIFY")) {
      short ox = *mv_x, oy = *mv_y, jx = *mv_x, jy = *mv_y; int c1, c2;
      if (!orig) orig = next_sym("FullPelBlockMotionBiPred");
      c1 = orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x1, pred_mv_y1, pred_mv_x2, pred_mv_y2, &jx, &jy, s_mv_x, s_mv_y, search_range, min_mcost, lambda_factor);
      c2 = jmo_fullpel_bipred(&b, orig_pic, pic_pix_x, pic_pix_y, blocktype, pred_mv_x1, pred_mv_y1, pred_mv_x2, pred_mv_y2, &ox, &oy, s_mv_x, s_mv_y, search_range, min_mcost, lambda_factor);
      if (c1 != c2 || jx != ox || jy != oy)
        fprintf(stderr, "BIPRED FULLPEL MISMATCH list=%d pix=(%d,%d) in=(%d,%d) s=(%d,%d) R=%d min=%d: jm=(%d,%d,%d) or=(%d,%d,%d)\n",
                list, pic_pix_x, pic_pix_y, *mv_x, *mv_y, *s_mv_x, *s_mv_y, search_range, min_mcost, jx, jy, c1, ox, oy, c2);
      *mv_x = jx; *mv_y = jy; return c1;
    }
    return jmo_fullpel_bipred(&b, orig_pic, pic_pix_x, pic_pix_y, blocktype, pred_mv_x1, pred_mv_y1, pred_mv_x2, pred_mv_y2,
                              mv_x, mv_y, s_mv_x, s_mv_y, search_range, min_mcost, lambda_factor);
  }
}

int SubPelBlockSearchBiPred(imgpel *orig_pic, short ref, int list, int pic_pix_x, int pic_pix_y, int blocktype,
                            short pred_mv_x, short pred_mv_y, short *mv_x, short *mv_y, short *s_mv_x, short *s_mv_y,
                            int search_pos2, int search_pos4, int min_mcost, int *lambda)
{
  static int (*orig)(imgpel *, short, int, int, int, int, short, short, short *, short *, short *, short *, int, int, int, int *);
  n_calls[C_BISUB]++;
  if (!bipred_swappable()) {
    if (!orig) orig = next_sym("SubPelBlockSearchBiPred");
    return orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, mv_x, mv_y, s_mv_x, s_mv_y, search_pos2, search_pos4, min_mcost, lambda);
  }
  {
    jmo_bipred b; jmo_ref r1, r2;
    fill_bipred(&b, &r1, &r2, ref, list, blocktype);
    if (getenv("JMO_SWAP_VERIFY")) {
      short ox = *mv_x, oy = *mv_y, jx = *mv_x, jy = *mv_y; int c1, c2;
      if (!orig) orig = next_sym("SubPelBlockSearchBiPred");
      c1 = orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, &jx, &jy, s_mv_x, s_mv_y, search_pos2, search_pos4, min_mcost, lambda);
      c2 = jmo_subpel_bipred(&b, orig_pic, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, &ox, &oy, s_mv_x, s_mv_y, search_pos2, search_pos4, min_mcost, lambda);
      if (c1 != c2 || jx != ox || jy != oy)
        fprintf(stderr, "BIPRED SUBPEL MISMATCH list=%d pix=(%d,%d) in=(%d,%d) s=(%d,%d) min=%d: jm=(%d,%d,%d) or=(%d,%d,%d)\n",
                list, pic_pix_x, pic_pix_y, *mv_x, *mv_y, *s_mv_x, *s_mv_y, min_mcost, jx, jy, c1, ox, oy, c2);
      *mv_x = jx; *mv_y = jy; return c1;
    }
    return jmo_subpel_bipred(&b, orig_pic, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, mv_x, mv_y, s_mv_x, s_mv_y,
                             search_pos2, search_pos4, min_mcost, lambda);
  }
}


/* ------------------------------------------------------------------ 0x800 EPZS */

static jmo_epzs *epzs_state;

static int epzs_swapped(void)
{
  return (swap_mask & 0x800) && epzs_state;
}

int EPZSInit(void)
{
  static int (*orig)(void);
  int r;
  if (!orig) orig = next_sym("EPZSInit");
  r = orig();                                   /* JM keeps its own state too: the unswapped configurations and mv-search.c:595 use it */
  if ((swap_mask & 0x800) && !input->EPZSSubPelGrid && !input->PicInterlace && !input->MbInterlace) {
    jmo_epzs_config c;
    memset(&c, 0, sizeof(c));
    c.search_range = input->search_range; c.bipred_me = input->BiPredMotionEstimation; c.bipred_search_range = input->BiPredMESearchRange;
    c.pattern = input->EPZSPattern; c.dual = input->EPZSDual; c.fixed = input->EPZSFixed; c.temporal = input->EPZSTemporal;
    c.spatial_mem = input->EPZSSpatialMem;
    c.min_scale = input->EPZSMinThresScale; c.med_scale = input->EPZSMedThresScale; c.max_scale = input->EPZSMaxThresScale;
    c.subpel_scale = input->EPZSSubPelThresScale;
    c.width = img->width; c.height = img->height; c.width_cr = img->width_cr; c.height_cr = img->height_cr;
    c.bitdepth_luma = img->bitdepth_luma; c.bitdepth_chroma = img->bitdepth_chroma;
    c.chroma_me = input->ChromaMEEnable; c.chroma_me_weight = input->ChromaMEWeight;
    c.max_refs = img->max_num_references;
    epzs_state = jmo_epzs_create(&c);
  }
  return r;
}

void EPZSSliceInit(EPZSColocParams *p, StorablePicture **lx[6])
{
  static void (*orig)(EPZSColocParams *, StorablePicture **[6]);
  if (!orig) orig = next_sym("EPZSSliceInit");
  orig(p, lx);
  if (epzs_swapped()) {
    jmo_epzs_slice s;
    const int list = img->type == B_SLICE ? LIST_1 : LIST_0, w4 = img->width / 4, h4 = img->height / 4;
    StorablePicture *fs[2];
    short *cmv[2] = {NULL, NULL}; long long *cid[2] = {NULL, NULL};
    int i, j, k;
    memset(&s, 0, sizeof(s));
    s.is_b_slice = img->type == B_SLICE; s.poc = enc_picture->poc;
    for (j = 0; j < 2; j++) { s.list_size[j] = listXsize[j]; for (i = 0; i < listXsize[j]; i++) s.list_poc[j][i] = lx[j][i]->poc; }
    s.num_ref_idx_l0_active = img->num_ref_idx_l0_active;
    for (i = 0; i < MAX_LIST_SIZE; i++) s.ref_pic_num_l0[i] = enc_picture->ref_pic_num[LIST_0][i];
    if (input->EPZSTemporal) {
      fs[0] = lx[list][0]; fs[1] = listXsize[list] > 1 ? lx[list][1] : lx[list][0];
      for (k = 0; k < 2; k++) {
        cmv[k] = malloc(sizeof(short) * 2 * w4 * h4); cid[k] = malloc(sizeof(long long) * w4 * h4);
        for (j = 0; j < h4; j++) for (i = 0; i < w4; i++) {
          cmv[k][(j * w4 + i) * 2] = fs[k]->mv[LIST_0][j][i][0]; cmv[k][(j * w4 + i) * 2 + 1] = fs[k]->mv[LIST_0][j][i][1];
          cid[k][j * w4 + i] = fs[k]->ref_id[LIST_0][j][i];
        }
        s.col_mv[k] = cmv[k]; s.col_ref_id[k] = cid[k];
      }
    }
    jmo_epzs_slice_init(epzs_state, &s);
    if (getenv("JMO_SWAP_VERIFY") && input->EPZSTemporal) {
      const short *c = jmo_epzs_colocated(epzs_state);
      long bad = 0;
      for (k = 0; k < 2; k++) for (j = 0; j < h4; j++) for (i = 0; i < w4; i++)
        bad += c[((k * h4 + j) * w4 + i) * 2] != p->mv[k][j][i][0] || c[((k * h4 + j) * w4 + i) * 2 + 1] != p->mv[k][j][i][1];
      if (bad) fprintf(stderr, "EPZS COLOCATED MISMATCH: %ld vectors\n", bad);
    }
    for (k = 0; k < 2; k++) { free(cmv[k]); free(cid[k]); }
  }
}

static void epzs_neighbours(jmo_epzs_nbr *nb, int mb_x, int mb_y, int bsx, char **refPic, short ***tmp_mv)
{
  PixelPos blk[4];
  int k;
  getLuma4x4Neighbour(img->current_mb_nr, mb_x - 1, mb_y, &blk[0]);
  getLuma4x4Neighbour(img->current_mb_nr, mb_x, mb_y - 1, &blk[1]);
  getLuma4x4Neighbour(img->current_mb_nr, mb_x + bsx, mb_y - 1, &blk[2]);
  getLuma4x4Neighbour(img->current_mb_nr, mb_x - 1, mb_y - 1, &blk[3]);
  memset(nb, 0, sizeof(*nb));
  for (k = 0; k < 4; k++) {
    nb->available[k] = blk[k].available;
    if (blk[k].available) {
      nb->ref[k] = refPic[blk[k].pos_y][blk[k].pos_x];
      nb->mv[k][0] = tmp_mv[blk[k].pos_y][blk[k].pos_x][0]; nb->mv[k][1] = tmp_mv[blk[k].pos_y][blk[k].pos_x][1];
    }
  }
}

int EPZSPelBlockMotionSearch(imgpel *cur_pic, short ref, int list, int list_offset, char ***refPic, short ****tmp_mv, int pic_pix_x, int pic_pix_y,
                             int blocktype, short pred_mv[2], short mv[2], int search_range, int min_mcost, int lambda_factor)
{
  static int (*orig)(imgpel *, short, int, int, char ***, short ****, int, int, int, short[2], short[2], int, int, int);
  n_calls[C_EPZS_PEL]++;
  if (!epzs_swapped() || img->MbaffFrameFlag || img->structure != FRAME || list_offset) {
    if (!orig) orig = next_sym("EPZSPelBlockMotionSearch");
    return orig(cur_pic, ref, list, list_offset, refPic, tmp_mv, pic_pix_x, pic_pix_y, blocktype, pred_mv, mv, search_range, min_mcost, lambda_factor);
  }
  n_served[C_EPZS_PEL]++;
  {
    jmo_me_params p; jmo_ref r; jmo_epzs_nbr nb;
    short allmv[JMO_MAX_REFS][8][2];
    const int mb_x = pic_pix_x - img->opix_x, mb_y = pic_pix_y - img->opix_y, px2 = pic_pix_x >> 2;
    int rr, bt, cost;
    fill_me_params(&p, list, ref, list_offset);
    p.apply_weights = (active_pps->weighted_pred_flag > 0 || (active_pps->weighted_bipred_idc && (img->type == B_SLICE))) && input->UseWeightedReferenceME;   /* :1546 */
    if (p.apply_weights) {
      p.weight_luma = wp_weight[list + list_offset][ref][0]; p.offset_luma = wp_offset[list + list_offset][ref][0];
      p.weight_cr[0] = wp_weight[list + list_offset][ref][1]; p.weight_cr[1] = wp_weight[list + list_offset][ref][2];
      p.offset_cr[0] = wp_offset[list + list_offset][ref][1]; p.offset_cr[1] = wp_offset[list + list_offset][ref][2];
    }
    fill_ref(&r, listX[list + list_offset][ref]);
    jm_side_effects(listX[list + list_offset][ref]);
    epzs_neighbours(&nb, mb_x, mb_y, input->blc_size[blocktype][0], refPic[list], tmp_mv[list]);
    memset(allmv, 0, sizeof(allmv));
    for (rr = 0; rr < img->max_num_references && rr < JMO_MAX_REFS; rr++) for (bt = 0; bt < 8; bt++) {
      allmv[rr][bt][0] = img->all_mv[mb_y >> 2][mb_x >> 2][list][rr][bt][0]; allmv[rr][bt][1] = img->all_mv[mb_y >> 2][mb_x >> 2][list][rr][bt][1];
    }
    if (getenv("JMO_SWAP_VERIFY")) {
      short jm[2] = {mv[0], mv[1]}, om[2] = {mv[0], mv[1]};
      int c1, before = EPZSDistortion[list][blocktype - 1][px2];
      if (!orig) orig = next_sym("EPZSPelBlockMotionSearch");
      c1 = orig(cur_pic, ref, list, list_offset, refPic, tmp_mv, pic_pix_x, pic_pix_y, blocktype, pred_mv, jm, search_range, min_mcost, lambda_factor);
      jmo_epzs_distortion_row(epzs_state, list, blocktype - 1)[px2] = before;
      cost = jmo_epzs_pel_search(epzs_state, &p, &r, cur_pic, ref, list, &nb, allmv, img->type == P_SLICE, img->current_mb_nr, img->opix_x, img->opix_y,
                                 pic_pix_x, pic_pix_y, blocktype, pred_mv, om, search_range, min_mcost, lambda_factor);
      if (c1 != cost || jm[0] != om[0] || jm[1] != om[1] || EPZSDistortion[list][blocktype - 1][px2] != jmo_epzs_distortion_row(epzs_state, list, blocktype - 1)[px2])
        fprintf(stderr, "EPZS PEL MISMATCH mb=%d ref=%d list=%d pix=(%d,%d) bt=%d pred=(%d,%d) in=(%d,%d): jm=(%d,%d,%d,sad %d) or=(%d,%d,%d,sad %d)\n", img->current_mb_nr, ref, list,
                pic_pix_x, pic_pix_y, blocktype, pred_mv[0], pred_mv[1], mv[0], mv[1], jm[0], jm[1], c1, EPZSDistortion[list][blocktype - 1][px2], om[0], om[1], cost,
                jmo_epzs_distortion_row(epzs_state, list, blocktype - 1)[px2]);
      jmo_epzs_distortion_row(epzs_state, list, blocktype - 1)[px2] = EPZSDistortion[list][blocktype - 1][px2];
      mv[0] = jm[0]; mv[1] = jm[1];
      return c1;
    }
    cost = jmo_epzs_pel_search(epzs_state, &p, &r, cur_pic, ref, list, &nb, allmv, img->type == P_SLICE, img->current_mb_nr, img->opix_x, img->opix_y,
                               pic_pix_x, pic_pix_y, blocktype, pred_mv, mv, search_range, min_mcost, lambda_factor);
    EPZSDistortion[list][blocktype - 1][px2] = jmo_epzs_distortion_row(epzs_state, list, blocktype - 1)[px2];      /* mv-search.c:595,783 read JM's array */
    img_width = r.W; img_height = r.H;
    return cost;
  }
}

int EPZSSubPelBlockMotionSearch(imgpel *orig_pic, short ref, int list, int pic_pix_x, int pic_pix_y, int blocktype, short pred_mv[2], short mv[2],
                                int search_pos2, int search_pos4, int min_mcost, int *lambda)
{
  static int (*orig)(imgpel *, short, int, int, int, int, short[2], short[2], int, int, int, int *);
  const int list_offset = img->mb_data[img->current_mb_nr].list_offset;
  n_calls[C_EPZS_SUB]++;
  if (!epzs_swapped() || img->MbaffFrameFlag || img->structure != FRAME || list_offset) {
    if (!orig) orig = next_sym("EPZSSubPelBlockMotionSearch");
    return orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv, mv, search_pos2, search_pos4, min_mcost, lambda);
  }
  n_served[C_EPZS_SUB]++;
  {
    jmo_me_params p; jmo_ref r;
    fill_me_params(&p, list, ref, list_offset);
    fill_ref(&r, listX[list + list_offset][ref]);
    jm_side_effects(listX[list + list_offset][ref]);
    if (getenv("JMO_SWAP_VERIFY")) {
      short jm[2] = {mv[0], mv[1]}, om[2] = {mv[0], mv[1]};
      int c1, c2;
      if (!orig) orig = next_sym("EPZSSubPelBlockMotionSearch");
      c1 = orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv, jm, search_pos2, search_pos4, min_mcost, lambda);
      c2 = jmo_epzs_subpel_search(epzs_state, &p, &r, orig_pic, pic_pix_x, pic_pix_y, blocktype, pred_mv, om, search_pos2, search_pos4, min_mcost, lambda);
      if (c1 != c2 || jm[0] != om[0] || jm[1] != om[1])
        fprintf(stderr, "EPZS SUBPEL MISMATCH mb=%d ref=%d pix=(%d,%d) bt=%d pred=(%d,%d) in=(%d,%d) min=%d: jm=(%d,%d,%d) or=(%d,%d,%d)\n", img->current_mb_nr, ref, pic_pix_x, pic_pix_y,
                blocktype, pred_mv[0], pred_mv[1], mv[0], mv[1], min_mcost, jm[0], jm[1], c1, om[0], om[1], c2);
      mv[0] = jm[0]; mv[1] = jm[1];
      return c1;
    }
    return jmo_epzs_subpel_search(epzs_state, &p, &r, orig_pic, pic_pix_x, pic_pix_y, blocktype, pred_mv, mv, search_pos2, search_pos4, min_mcost, lambda);
  }
}

int EPZSBiPredBlockMotionSearch(imgpel *cur_pic, short ref, int list, int list_offset, char ***refPic, short ****tmp_mv, int pic_pix_x, int pic_pix_y,
                                int blocktype, short *pred_mv1, short *pred_mv2, short mv[2], short s_mv[2], int search_range, int min_mcost, int lambda_factor)
{
  static int (*orig)(imgpel *, short, int, int, char ***, short ****, int, int, int, short *, short *, short[2], short[2], int, int, int);
  n_calls[C_EPZS_BI]++;
  if (!epzs_swapped() || !bipred_swappable() || img->MbaffFrameFlag || img->structure != FRAME || list_offset) {
    if (!orig) orig = next_sym("EPZSBiPredBlockMotionSearch");
    return orig(cur_pic, ref, list, list_offset, refPic, tmp_mv, pic_pix_x, pic_pix_y, blocktype, pred_mv1, pred_mv2, mv, s_mv, search_range, min_mcost, lambda_factor);
  }
  n_served[C_EPZS_BI]++;
  {
    jmo_bipred b; jmo_ref r1, r2; jmo_epzs_nbr nb;
    fill_bipred(&b, &r1, &r2, ref, list, blocktype);
    /* EPZS reads wp_offset[..][0][0] for list 1 (:2013, :2017) where the full search reads [0][ref] (me_fullsearch.c:190-191) */
    if (b.apply_weights) {
      short o1 = list == 0 ? wp_offset[list_offset][ref][0] : wp_offset[list_offset + LIST_1][0][0];
      short o2 = list == 0 ? wp_offset[list_offset + LIST_1][ref][0] : wp_offset[list_offset][0][0];
      b.offset_bi = (o1 + o2 + 1) >> 1;
    }
    epzs_neighbours(&nb, pic_pix_x - img->opix_x, pic_pix_y - img->opix_y, input->blc_size[blocktype][0], refPic[list], tmp_mv[list]);
    if (getenv("JMO_SWAP_VERIFY")) {
      short jm[2] = {mv[0], mv[1]}, om[2] = {mv[0], mv[1]};
      int c1, c2;
      if (!orig) orig = next_sym("EPZSBiPredBlockMotionSearch");
      c1 = orig(cur_pic, ref, list, list_offset, refPic, tmp_mv, pic_pix_x, pic_pix_y, blocktype, pred_mv1, pred_mv2, jm, s_mv, search_range, min_mcost, lambda_factor);
      c2 = jmo_epzs_bipred_search(epzs_state, &b, cur_pic, ref, list, &nb, img->opix_x, img->opix_y, pic_pix_x, pic_pix_y, blocktype, pred_mv1, pred_mv2, om, s_mv, search_range, min_mcost, lambda_factor);
      if (c1 != c2 || jm[0] != om[0] || jm[1] != om[1])
        fprintf(stderr, "EPZS BIPRED MISMATCH mb=%d list=%d pix=(%d,%d) in=(%d,%d) s=(%d,%d) R=%d: jm=(%d,%d,%d) or=(%d,%d,%d)\n", img->current_mb_nr, list, pic_pix_x, pic_pix_y,
                mv[0], mv[1], s_mv[0], s_mv[1], search_range, jm[0], jm[1], c1, om[0], om[1], c2);
      mv[0] = jm[0]; mv[1] = jm[1];
      return c1;
    }
    return jmo_epzs_bipred_search(epzs_state, &b, cur_pic, ref, list, &nb, img->opix_x, img->opix_y, pic_pix_x, pic_pix_y, blocktype, pred_mv1, pred_mv2,
                                  mv, s_mv, search_range, min_mcost, lambda_factor);
  }
}

int EPZSSubPelBlockSearchBiPred(imgpel *orig_pic, short ref, int list, int pic_pix_x, int pic_pix_y, int blocktype, short *pred_mv1, short *pred_mv2,
                                short mv[2], short s_mv[2], int search_pos2, int search_pos4, int min_mcost, int *lambda)
{
  static int (*orig)(imgpel *, short, int, int, int, int, short *, short *, short[2], short[2], int, int, int, int *);
  const int list_offset = img->mb_data[img->current_mb_nr].list_offset;
  n_calls[C_EPZS_BISUB]++;
  if (!epzs_swapped() || !bipred_swappable() || img->MbaffFrameFlag || img->structure != FRAME || list_offset) {
    if (!orig) orig = next_sym("EPZSSubPelBlockSearchBiPred");
    return orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv1, pred_mv2, mv, s_mv, search_pos2, search_pos4, min_mcost, lambda);
  }
  n_served[C_EPZS_BISUB]++;
  {
    jmo_bipred b; jmo_ref r1, r2;
    fill_bipred(&b, &r1, &r2, ref, list, blocktype);
    if (b.apply_weights) {                                  /* :2747-2748 */
      short o1 = list == 0 ? wp_offset[list_offset][ref][0] : wp_offset[list_offset + 1][0][0];
      short o2 = list == 0 ? wp_offset[list_offset + 1][ref][0] : wp_offset[list_offset][0][0];
      b.offset_bi = (o1 + o2 + 1) >> 1;
    }
    if (getenv("JMO_SWAP_VERIFY")) {
      short jm[2] = {mv[0], mv[1]}, om[2] = {mv[0], mv[1]};
      int c1, c2;
      if (!orig) orig = next_sym("EPZSSubPelBlockSearchBiPred");
      c1 = orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv1, pred_mv2, jm, s_mv, search_pos2, search_pos4, min_mcost, lambda);
      c2 = jmo_epzs_subpel_bipred(&b, orig_pic, pic_pix_x, pic_pix_y, blocktype, pred_mv1, pred_mv2, om, s_mv, search_pos2, search_pos4, min_mcost, lambda);
      if (c1 != c2 || jm[0] != om[0] || jm[1] != om[1])
        fprintf(stderr, "EPZS BIPRED SUBPEL MISMATCH mb=%d list=%d pix=(%d,%d) in=(%d,%d) s=(%d,%d): jm=(%d,%d,%d) or=(%d,%d,%d)\n", img->current_mb_nr, list, pic_pix_x, pic_pix_y,
                mv[0], mv[1], s_mv[0], s_mv[1], jm[0], jm[1], c1, om[0], om[1], c2);
      mv[0] = jm[0]; mv[1] = jm[1];
      return c1;
    }
    return jmo_epzs_subpel_bipred(&b, orig_pic, pic_pix_x, pic_pix_y, blocktype, pred_mv1, pred_mv2, mv, s_mv, search_pos2, search_pos4, min_mcost, lambda);
  }
}


/* ------------------------------------------------------------------ 0x1000 UMHexagonS */

static jmo_umhex *umhex_state;
static int umhex_decided, umhex_on;

/* decided once: the state is either all the oracle's or all JM's */
static int umhex_swapped(void)
{
  if (!umhex_decided) {
    umhex_decided = 1;
    umhex_on = (swap_mask & 0x1000) && !input->PicInterlace && !input->MbInterlace &&
               !(input->BiPredMotionEstimation && (input->ChromaMEEnable || !bipred_swappable()));
    if (umhex_on) {
      jmo_umhex_config c;
      memset(&c, 0, sizeof(c));
      c.search_range = input->search_range; c.bipred_search_range = input->BiPredMotionEstimation ? input->BiPredMESearchRange : 0;
      c.dsr = input->UMHexDSR; c.scale = input->UMHexScale; c.qp_n = input->qpN;
      c.bipred_me = input->BiPredMotionEstimation; c.full_search = input->full_search; c.successive_bframe = input->successive_Bframe;
      c.width = img->width; c.height = img->height; c.max_refs = img->max_num_references;
      umhex_state = jmo_umhex_create(&c);
    }
  }
  return umhex_on;
}

void UMHEX_decide_intrabk_SAD(void)
{
  static void (*orig)(void);
  if (!umhex_swapped()) { if (!orig) orig = next_sym("UMHEX_decide_intrabk_SAD"); orig(); return; }
  jmo_umhex_decide_intrabk_sad(umhex_state, img->type == I_SLICE, img->pix_x, img->pix_y);
}

void UMHEX_skip_intrabk_SAD(int best_mode, int ref_max)
{
  static void (*orig)(int, int);
  if (!umhex_swapped()) { if (!orig) orig = next_sym("UMHEX_skip_intrabk_SAD"); orig(best_mode, ref_max); return; }
  jmo_umhex_skip_intrabk_sad(umhex_state, best_mode, ref_max, img->number, img->type == I_SLICE, img->pix_x);
}

void UMHEXSetMotionVectorPredictor(short pmv[2], char **refPic, short ***tmp_mv, short ref_frame, int list, int block_x, int block_y,
                                   int blockshape_x, int blockshape_y, int *search_range)
{
  static void (*orig)(short[2], char **, short ***, short, int, int, int, int, int, int *);
  n_calls[C_UM_PRED]++;
  if (!umhex_swapped()) {
    if (!orig) orig = next_sym("UMHEXSetMotionVectorPredictor");
    orig(pmv, refPic, tmp_mv, ref_frame, list, block_x, block_y, blockshape_x, blockshape_y, search_range);
    return;
  }
  n_served[C_UM_PRED]++;
  {
    PixelPos blk[4];
    jmo_umhex_nbr nb;
    const int mb_x = 4 * block_x, mb_y = 4 * block_y;
    int k;
    getLuma4x4Neighbour(img->current_mb_nr, mb_x - 1, mb_y, &blk[0]);
    getLuma4x4Neighbour(img->current_mb_nr, mb_x, mb_y - 1, &blk[1]);
    getLuma4x4Neighbour(img->current_mb_nr, mb_x + blockshape_x, mb_y - 1, &blk[2]);
    getLuma4x4Neighbour(img->current_mb_nr, mb_x - 1, mb_y - 1, &blk[3]);
    memset(&nb, 0, sizeof(nb));
    for (k = 0; k < 4; k++) {
      nb.available[k] = blk[k].available;
      if (blk[k].available) {
        nb.ref[k] = refPic[blk[k].pos_y][blk[k].pos_x];
        nb.mv[k][0] = tmp_mv[blk[k].pos_y][blk[k].pos_x][0]; nb.mv[k][1] = tmp_mv[blk[k].pos_y][blk[k].pos_x][1];
        nb.pos_x[k] = blk[k].pos_x; nb.pos_y[k] = blk[k].pos_y;
      }
    }
    jmo_umhex_set_mv_predictor(umhex_state, pmv, &nb, ref_frame, list, block_x, block_y, blockshape_x, blockshape_y, UMHEX_blocktype, bipred_flag,
                               (const int (*)[4])input->blocktype_lut, search_range);
  }
}

static void umhex_allmv(short (*allmv)[JMO_MAX_REFS][9][2], int block_y, int block_x)
{
  int l, r, bt;
  memset(allmv, 0, sizeof(short) * 2 * JMO_MAX_REFS * 9 * 2);
  for (l = 0; l < 2; l++) for (r = 0; r < img->max_num_references && r < JMO_MAX_REFS; r++) for (bt = 0; bt < 9; bt++) {
    allmv[l][r][bt][0] = img->all_mv[block_y][block_x][l][r][bt][0]; allmv[l][r][bt][1] = img->all_mv[block_y][block_x][l][r][bt][1];
  }
}

int UMHEXIntegerPelBlockMotionSearch(imgpel *orig_pic, short ref, int list, int pic_pix_x, int pic_pix_y, int blocktype, short pred_mv_x, short pred_mv_y,
                                     short *mv_x, short *mv_y, int search_range, int min_mcost, int lambda_factor)
{
  static int (*orig)(imgpel *, short, int, int, int, int, short, short, short *, short *, int, int, int);
  n_calls[C_UM_PEL]++;
  if (!umhex_swapped()) {
    if (!orig) orig = next_sym("UMHEXIntegerPelBlockMotionSearch");
    return orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, mv_x, mv_y, search_range, min_mcost, lambda_factor);
  }
  n_served[C_UM_PEL]++;
  {
    jmo_me_params p; jmo_ref r;
    short allmv[2][JMO_MAX_REFS][9][2];
    fill_me_params(&p, list, ref, 0);
    fill_ref(&r, listX[list][ref]);
    jm_side_effects(listX[list][ref]);
    umhex_allmv(allmv, (pic_pix_y - img->opix_y) >> 2, (pic_pix_x - img->opix_x) >> 2);
    img_width = r.W; img_height = r.H;
    return jmo_umhex_pel_search(umhex_state, &p, &r, orig_pic, ref, list, (const short (*)[JMO_MAX_REFS][9][2])allmv, frame_ctr[B_SLICE], img->opix_x, img->opix_y,
                                pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, mv_x, mv_y, search_range, min_mcost, lambda_factor);
  }
}

int UMHEXSubPelBlockMotionSearch(imgpel *orig_pic, short ref, int list, int pic_pix_x, int pic_pix_y, int blocktype, short pred_mv_x, short pred_mv_y,
                                 short *mv_x, short *mv_y, int search_pos2, int search_pos4, int min_mcost, int lambda_factor)
{
  static int (*orig)(imgpel *, short, int, int, int, int, short, short, short *, short *, int, int, int, int);
  n_calls[C_UM_SUB]++;
  if (!umhex_swapped()) {
    if (!orig) orig = next_sym("UMHEXSubPelBlockMotionSearch");
    return orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, mv_x, mv_y, search_pos2, search_pos4, min_mcost, lambda_factor);
  }
  n_served[C_UM_SUB]++;
  {
    jmo_me_params p; jmo_ref r;
    fill_me_params(&p, list, ref, 0);
    fill_ref(&r, listX[list][ref]);
    jm_side_effects(listX[list][ref]);
    return jmo_umhex_subpel_search(umhex_state, &p, &r, orig_pic, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, mv_x, mv_y, min_mcost, lambda_factor);
  }
}

int UMHEXBipredIntegerPelBlockMotionSearch(imgpel *cur_pic, short ref, int list, int pic_pix_x, int pic_pix_y, int blocktype, short pred_mv_x1, short pred_mv_y1,
                                           short pred_mv_x2, short pred_mv_y2, short *mv_x, short *mv_y, short *s_mv_x, short *s_mv_y, int search_range,
                                           int min_mcost, int lambda_factor)
{
  static int (*orig)(imgpel *, short, int, int, int, int, short, short, short, short, short *, short *, short *, short *, int, int, int);
  n_calls[C_UM_BI]++;
  if (!umhex_swapped()) {
    if (!orig) orig = next_sym("UMHEXBipredIntegerPelBlockMotionSearch");
    return orig(cur_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x1, pred_mv_y1, pred_mv_x2, pred_mv_y2, mv_x, mv_y, s_mv_x, s_mv_y, search_range, min_mcost, lambda_factor);
  }
  n_served[C_UM_BI]++;
  {
    jmo_bipred b; jmo_ref r1, r2;
    const int block_y = (pic_pix_y - img->opix_y) >> 2, block_x = (pic_pix_x - img->opix_x) >> 2;
    short ******bmv = list ? img->bipred_mv1 : img->bipred_mv2;
    short l1[2] = { bmv[block_y][block_x][1][0][blocktype][0], bmv[block_y][block_x][1][0][blocktype][1] };
    fill_bipred(&b, &r1, &r2, ref, list, blocktype);           /* offsets as :963-964 == me_fullsearch.c:190-191 */
    return jmo_umhex_bipred_search(umhex_state, &b, cur_pic, list, l1, frame_ctr[B_SLICE], img->opix_x, img->opix_y, pic_pix_x, pic_pix_y, blocktype,
                                   pred_mv_x1, pred_mv_y1, pred_mv_x2, pred_mv_y2, mv_x, mv_y, s_mv_x, s_mv_y, search_range, min_mcost, lambda_factor);
  }
}

/* ------------------------------------------------------------------ 0x200 low-complexity mode-decision costs */

extern void SetModesAndRefframe(Macroblock *currMB, int b8, short *p_dir, int *l0_mode, int *l1_mode, short *l0_ref, short *l1_ref);
extern void LumaPrediction(Macroblock *currMB, int block_x, int block_y, int block_size_x, int block_size_y, int p_dir, int l0_mode,
                           int l1_mode, short l0_ref_idx, short l1_ref_idx);

static void mb_cur(jmo_pel cur[256]) { int j; for (j = 0; j < 16; j++) memcpy(cur + 16 * j, &pCurImg[img->opix_y + j][img->opix_x], 16 * sizeof(imgpel)); }

int TransformDecision(Macroblock *currMB, int block_check, int *cost)
{
  static int (*orig)(Macroblock *, int, int *);
  n_calls[C_TDEC]++;
  if (!(swap_mask & 0x200) || block_check != -1 || input->ModeDecisionMetric == ERROR_SSE) {
    if (!orig) orig = next_sym("TransformDecision");
    return orig(currMB, block_check, cost);
  }
  {
    int b8, bx, by, l0_mode, l1_mode, c4, c8;
    short p_dir, l0_ref, l1_ref;
    jmo_pel cur[256];
    /* the predictions are JM's own (LumaPrediction per 4x4 block, :1493); img->mpr is also the side effect later code sees */
    for (b8 = 0; b8 < 4; b8++) {
      SetModesAndRefframe(currMB, b8, &p_dir, &l0_mode, &l1_mode, &l0_ref, &l1_ref);
      for (by = (b8 >> 1) << 3; by < ((b8 >> 1) << 3) + 8; by += 4)
        for (bx = (b8 & 1) << 3; bx < ((b8 & 1) << 3) + 8; bx += 4)
          LumaPrediction(currMB, bx, by, 4, 4, p_dir, l0_mode, l1_mode, l0_ref, l1_ref);
    }
    mb_cur(cur);
    jmo_pred_costs(cur, &img->mpr[0][0][0], input->ModeDecisionMetric, 0, &c4, &c8);
    if (input->Transform8x8Mode == 2) return 1;
    if (c8 < c4) return 1;
    *cost = (*cost - c8 + c4);
    return 0;
  }
}

int GetSkipCostMB(Macroblock *currMB)
{
  static int (*orig)(Macroblock *);
  n_calls[C_SKIPC]++;
  if (!(swap_mask & 0x200) || input->ModeDecisionMetric == ERROR_SSE) {
    if (!orig) orig = next_sym("GetSkipCostMB");
    return orig(currMB);
  }
  {
    int bx, by, c4, c8;
    jmo_pel cur[256];
    for (by = 0; by < 16; by += 4) for (bx = 0; bx < 16; bx += 4) LumaPrediction(currMB, bx, by, 4, 4, 0, 0, 0, 0, 0);
    mb_cur(cur);
    jmo_pred_costs(cur, &img->mpr[0][0][0], input->ModeDecisionMetric, 1, &c4, &c8);
    return (input->rdopt == 0 && input->Transform8x8Mode) ? c8 : c4;      /* mv-search.c:1167-1177 */
  }
}

/* ------------------------------------------------------------------ 0x400 in-loop deblocking filter */

void DeblockFrame(ImageParameters *im, imgpel **imgY, imgpel ***imgUV)
{
  static void (*orig)(ImageParameters *, imgpel **, imgpel ***);
  const int W = im->width, H = im->height, mbw = W / 16, nmb = (int)im->PicSizeInMbs, w4 = W / 4, h4 = H / 4;
  int i, l, x, y;
  n_calls[C_DEBLOCK]++;
  if (!(swap_mask & 0x400) || im->MbaffFrameFlag || im->structure != FRAME || im->type == SP_SLICE || im->type == SI_SLICE ||
      (imgUV && im->yuv_format != YUV400 && im->bitdepth_chroma != im->bitdepth_luma) || (im->yuv_format == YUV444 && IS_INDEPENDENT(input)) ||
      imgY[1] != imgY[0] + W) {
    if (!orig) orig = next_sym("DeblockFrame");
    orig(im, imgY, imgUV);
    return;
  }
  {
    jmo_deblock_mb *mbs = calloc(nmb, sizeof(*mbs));
    jmo_deblock_blk *blks = calloc((size_t)w4 * h4, sizeof(*blks));
    for (i = 0; i < nmb; i++) {
      Macroblock *m = &im->mb_data[i];
      if (m->mb_type == IPCM) { m->qp = 0; m->qpc[0] = 0; m->qpc[1] = 0; }        /* loopFilter.c:105-113, a side effect that stays */
      mbs[i].intra = m->mb_type == I4MB || m->mb_type == I8MB || m->mb_type == I16MB || m->mb_type == IPCM;
      mbs[i].qp = (unsigned char)m->qp; mbs[i].qpc[0] = (unsigned char)m->qpc[0]; mbs[i].qpc[1] = (unsigned char)m->qpc[1];
      mbs[i].disable_idc = (unsigned char)m->LFDisableIdc;
      mbs[i].alpha_c0_offset = (signed char)m->LFAlphaC0Offset; mbs[i].beta_offset = (signed char)m->LFBetaOffset;
      mbs[i].transform_8x8 = (unsigned char)m->luma_transform_size_8x8_flag;
      mbs[i].avail_a = (unsigned char)m->mbAvailA; mbs[i].avail_b = (unsigned char)m->mbAvailB;
      mbs[i].cbp_blk = (unsigned short)(m->cbp_blk & 0xffff);
    }
    for (y = 0; y < h4; y++) for (x = 0; x < w4; x++) {
      jmo_deblock_blk *b = &blks[y * w4 + x];
      for (l = 0; l < 2; l++) {
        b->mv[l][0] = enc_picture->mv[l][y][x][0]; b->mv[l][1] = enc_picture->mv[l][y][x][1];
        b->ref_id[l] = enc_picture->ref_idx[l][y][x] < 0 ? INT64_MIN : enc_picture->ref_pic_id[l][y][x];
      }
    }
    jmo_deblock_frame(imgY[0], imgUV && im->yuv_format != YUV400 ? imgUV[0][0] : NULL, imgUV && im->yuv_format != YUV400 ? imgUV[1][0] : NULL,
                      W, H, im->yuv_format, im->bitdepth_luma, mbs, blks, 4);
    free(mbs); free(blks);
    im->current_mb_nr = nmb - 1;            /* where DeblockMb :178 leaves it */
    (void)mbw;
  }
}

/* ------------------------------------------------------------------ 0x08 fast full search */

#define FF_MAXREF 16
static struct { int done; jmo_fastfull ff; } ff_state[2][FF_MAXREF];

void ResetFastFullIntegerSearch(void)
{
  static void (*orig)(void);
  int l, r;
  if (!orig) orig = next_sym("ResetFastFullIntegerSearch");
  orig();
  for (l = 0; l < 2; l++) for (r = 0; r < FF_MAXREF; r++) ff_state[l][r].done = 0;
}

int FastFullPelBlockMotionSearch(imgpel *orig_pic, short ref, int list, int pic_pix_x, int pic_pix_y, int blocktype,
                                 short pred_mv_x, short pred_mv_y, short *mv_x, short *mv_y, int search_range,
                                 int min_mcost, int lambda_factor)
{
  static int (*orig)(imgpel *, short, int, int, int, int, short, short, short *, short *, int, int, int);
  n_calls[C_FAST]++;
  if (!(swap_mask & 0x08)) {
    if (!orig) orig = next_sym("FastFullPelBlockMotionSearch");
    return orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, mv_x, mv_y, search_range, min_mcost, lambda_factor);
  }
  {
    int list_offset = img->mb_data[img->current_mb_nr].list_offset;
    jmo_me_params p;
    jmo_fastfull *ff = &ff_state[list][ref].ff;
    fill_me_params(&p, list, ref, list_offset);
    if (!ff_state[list][ref].done) {
      /* SetupFastFullPelSearch, me_fullfast.c:491: range per reference as InitializeFastFullIntegerSearch :127-140 */
      int R = (input->full_search == 2 || ref == 0) ? input->search_range : input->search_range / 2;
      int max_pos = (2 * R + 1) * (2 * R + 1), y, k;
      short pmv[2];
      jmo_ref r;
      jmo_pel mb[768], *dst = mb;
      fill_ref(&r, listX[list + list_offset][ref]);
      jm_side_effects(listX[list + list_offset][ref]);
      SetMotionVectorPredictor(pmv, enc_picture->ref_idx[list], enc_picture->mv[list], ref, list, 0, 0, 16, 16);
      for (y = img->opix_y; y < img->opix_y + 16; y++) { memcpy(dst, &pCurImg[y][img->opix_x], 16 * sizeof(imgpel)); dst += 16; }
      if (ChromaMEEnable)
        for (k = 0; k < 2; k++)
          for (y = img->opix_c_y; y < img->opix_c_y + img->mb_cr_size_y; y++) {
            memcpy(dst, &imgUV_org[k][y][img->opix_c_x], img->mb_cr_size_x * sizeof(imgpel)); dst += img->mb_cr_size_x;
          }
      free(ff->block_sad);
      ff->block_sad = malloc(sizeof(int) * 8 * 16 * (size_t)max_pos);
      p.chroma_me = ChromaMEEnable;
      jmo_fastfull_setup(&p, &r, mb, img->opix_x, img->opix_y, pmv[0], pmv[1], R, ff);
      ff_state[list][ref].done = 1;
    }
    return jmo_fastfull_search(&p, ff, img->opix_x, img->opix_y, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y,
                               mv_x, mv_y, min_mcost, lambda_factor);
  }
}

/* ------------------------------------------------------------------ 0x10..0x40 transform + quant */

static void flat(int *dst, int **src, int n) { int j, i; for (j = 0; j < n; j++) for (i = 0; i < n; i++) dst[j * n + i] = src[j][i]; }

static void fill_quant(jmo_quant *q, int qp, int *ls, int *ils, int *lo, int **levelscale, int **invlevelscale,
                       int **leveloffset, int n, Macroblock *currMB, int weight, int max_val)
{
  memset(q, 0, sizeof(*q));
  flat(ls, levelscale, n); flat(ils, invlevelscale, n); flat(lo, leveloffset, n);
  q->qp = qp; q->levelscale = ls; q->invlevelscale = ils; q->leveloffset = lo;
  q->adaptive_rounding = img->AdaptiveRounding; q->adapt_rnd_weight = weight;
  q->field_scan = currMB->is_field_mode; q->disthres = input->disthres;
  q->max_val = max_val; q->cavlc = (input->symbol_mode == CAVLC); q->img_qp = img->qp;
  q->transform8x8_flag = currMB->luma_transform_size_8x8_flag;
}

static void tile_in(int (*t)[16], int **src, int rows, int cols) { int j, i; if (src) for (j = 0; j < rows; j++) for (i = 0; i < cols; i++) t[j][i] = src[j][i]; }
static void tile_out(int **dst, int (*t)[16], int rows, int cols) { int j, i; if (dst) for (j = 0; j < rows; j++) for (i = 0; i < cols; i++) dst[j][i] = t[j][i]; }

int dct_4x4(Macroblock *currMB, ColorPlane pl, int block_x, int block_y, int *coeff_cost, int intra)
{
  static int (*orig)(Macroblock *, ColorPlane, int, int, int *, int);
  n_calls[C_D4]++;
  if (!(swap_mask & 0x10) || (currMB->qp_scaled[pl] == 0 && img->lossless_qpprime_flag == 1)) {
    if (!orig) orig = next_sym("dct_4x4");
    return orig(currMB, pl, block_x, block_y, coeff_cost, intra);
  }
  {
    int ls[16], ils[16], lo[16], fadj[16][16], j, i, nz;
    jmo_pel recon[16][16];
    jmo_quant q;
    int qp = currMB->qp_scaled[pl], qp_rem = qp_rem_matrix[qp];
    int pos_x = block_x >> 2, pos_y = block_y >> 2;
    int b8 = 2 * (pos_y >> 1) + (pos_x >> 1), b4 = 2 * (pos_y & 1) + (pos_x & 1);
    int **fa = img->AdaptiveRounding ? (pl ? img->fadjust4x4Cr[pl - 1][intra] : img->fadjust4x4[intra]) : NULL;
    imgpel **img_enc = enc_picture->p_curr_img;
    fill_quant(&q, qp, ls, ils, lo, LevelScale4x4Comp[pl][intra][qp_rem], InvLevelScale4x4Comp[pl][intra][qp_rem],
               ptLevelOffset4x4[intra][qp], 4, currMB, AdaptRndWeight, img->max_imgpel_value);
    tile_in(fadj, fa, 16, 16);
    nz = jmo_dct_4x4(&q, img->m7[pl], (const jmo_pel (*)[16])img->mpr[pl], block_x, block_y, coeff_cost,
                     img->cofAC[b8 + (pl << 2)][b4][0], img->cofAC[b8 + (pl << 2)][b4][1], recon, fadj);
    tile_out(fa, fadj, 16, 16);
    for (j = block_y; j < block_y + 4; j++) for (i = block_x; i < block_x + 4; i++)
      img_enc[img->pix_y + j][img->pix_x + i] = recon[j][i];
    return nz;
  }
}

int dct_8x8(Macroblock *currMB, ColorPlane pl, int b8, int *coeff_cost, int intra)
{
  static int (*orig)(Macroblock *, ColorPlane, int, int *, int);
  n_calls[C_D8]++;
  if (!(swap_mask & 0x20) || (currMB->qp_scaled[pl] == 0 && img->lossless_qpprime_flag == 1)) {
    if (!orig) orig = next_sym("dct_8x8");
    return orig(currMB, pl, b8, coeff_cost, intra);
  }
  {
    int ls[64], ils[64], lo[64], fadj[16][16], lev[4][65], run[4][65], j, i, k, nz;
    jmo_pel recon[16][16];
    jmo_quant q;
    int qp = currMB->qp_scaled[pl], qp_rem = qp_rem_matrix[qp];
    int block_x = 8 * (b8 & 1), block_y = 8 * (b8 >> 1);
    int **fa = img->AdaptiveRounding ? (pl ? img->fadjust8x8Cr[pl - 1][intra] : img->fadjust8x8[intra]) : NULL;
    imgpel **img_enc = enc_picture->p_curr_img;
    fill_quant(&q, qp, ls, ils, lo, LevelScale8x8Comp[pl][intra][qp_rem], InvLevelScale8x8Comp[pl][intra][qp_rem],
               LevelOffset8x8Comp[pl][intra][qp], 8, currMB, AdaptRndWeight, img->max_imgpel_value);
    tile_in(fadj, fa, 16, 16);
    for (k = 0; k < 4; k++) {
      memcpy(lev[k], img->cofAC[b8 + (pl << 2)][k][0], sizeof(int) * 65);
      memcpy(run[k], img->cofAC[b8 + (pl << 2)][k][1], sizeof(int) * 65);
    }
    nz = jmo_dct_8x8(&q, img->m7[pl], (const jmo_pel (*)[16])img->mpr[pl], b8, coeff_cost, lev, run, recon, fadj);
    for (k = 0; k < 4; k++) {
      memcpy(img->cofAC[b8 + (pl << 2)][k][0], lev[k], sizeof(int) * 65);
      memcpy(img->cofAC[b8 + (pl << 2)][k][1], run[k], sizeof(int) * 65);
    }
    tile_out(fa, fadj, 16, 16);
    for (j = block_y; j < block_y + 8; j++) for (i = block_x; i < block_x + 8; i++)
      img_enc[img->pix_y + j][img->pix_x + i] = recon[j][i];
    return nz;
  }
}

int dct_16x16(Macroblock *currMB, ColorPlane pl, int new_intra_mode)
{
  static int (*orig)(Macroblock *, ColorPlane, int);
  n_calls[C_D16]++;
  if (!(swap_mask & 0x10) || (currMB->qp_scaled[pl] == 0 && img->lossless_qpprime_flag == 1) || img->type == SP_SLICE) {
    if (!orig) orig = next_sym("dct_16x16");
    return orig(currMB, pl, new_intra_mode);
  }
  {
    int ls[16], ils[16], lo[16], fadj[16][16], acl[16][16], acr[16][16], j, i, b, ac;
    jmo_pel cur[16][16], recon[16][16];
    jmo_quant q;
    int qp = currMB->qp_scaled[pl], qp_rem = qp_rem_matrix[qp];
    int **fa = img->AdaptiveRounding ? (pl ? img->fadjust4x4Cr[pl - 1][2] : img->fadjust4x4[2]) : NULL;
    imgpel **img_enc = enc_picture->p_curr_img;
    fill_quant(&q, qp, ls, ils, lo, LevelScale4x4Comp[pl][1][qp_rem], InvLevelScale4x4Comp[pl][1][qp_rem],
               ptLevelOffset4x4[1][qp], 4, currMB, AdaptRndWeight, img->max_imgpel_value);
    for (j = 0; j < 16; j++) memcpy(cur[j], &pCurImg[img->opix_y + j][img->opix_x], 16 * sizeof(imgpel));
    tile_in(fadj, fa, 16, 16);
    for (b = 0; b < 16; b++) {
      memcpy(acl[b], img->cofAC[(b >> 2) + (pl << 2)][b & 3][0], sizeof(int) * 16);
      memcpy(acr[b], img->cofAC[(b >> 2) + (pl << 2)][b & 3][1], sizeof(int) * 16);
    }
    ac = jmo_dct_16x16(&q, (const jmo_pel (*)[16])cur, (const jmo_pel (*)[16])img->mpr_16x16[pl][new_intra_mode],
                       img->cofDC[pl][0], img->cofDC[pl][1], acl, acr, recon, fadj);
    for (b = 0; b < 16; b++) {
      memcpy(img->cofAC[(b >> 2) + (pl << 2)][b & 3][0], acl[b], sizeof(int) * 16);
      memcpy(img->cofAC[(b >> 2) + (pl << 2)][b & 3][1], acr[b], sizeof(int) * 16);
    }
    tile_out(fa, fadj, 16, 16);
    for (j = 0; j < 16; j++) for (i = 0; i < 16; i++) img_enc[img->pix_y + j][img->pix_x + i] = recon[j][i];
    return ac;
  }
}

int dct_chroma(Macroblock *currMB, int uv, int cr_cbp)
{
  static int (*orig)(Macroblock *, int, int);
  n_calls[C_DCR]++;
  if (!(swap_mask & 0x40) || ((currMB->qp + img->bitdepth_luma_qp_scale) == 0 && img->lossless_qpprime_flag == 1) ||
      img->yuv_format == YUV444) {
    if (!orig) orig = next_sym("dct_chroma");
    return orig(currMB, uv, cr_cbp);
  }
  {
    int ls[16], ils[16], lo[16], lsdc[16], ilsdc[16], lodc[16], fadj[16][16], acl[8][16], acr[8][16], j, i, b, ret;
    jmo_pel recon[16][16];
    jmo_quant q, qdc;
    long long cbp_blk = currMB->cbp_blk;
    int intra = IS_INTRA(currMB);
    int cur_qp = currMB->qpc[uv] + img->bitdepth_chroma_qp_scale;
    int cur_qp_dc = currMB->qpc[uv] + 3 + img->bitdepth_chroma_qp_scale;
    int nb = (img->num_blk8x8_uv >> 1) * 4, uv_scale = uv * (img->num_blk8x8_uv >> 1);
    int **fa = img->AdaptiveRounding ? img->fadjust4x4Cr[intra][uv] : NULL;
    fill_quant(&q, cur_qp, ls, ils, lo, LevelScale4x4Comp[uv + 1][intra][qp_rem_matrix[cur_qp]],
               InvLevelScale4x4Comp[uv + 1][intra][qp_rem_matrix[cur_qp]], LevelOffset4x4Comp[uv + 1][intra][cur_qp],
               4, currMB, AdaptRndCrWeight, img->max_imgpel_value_comp[1]);
    qdc = q;
    if (img->yuv_format == YUV422) {
      fill_quant(&qdc, cur_qp_dc, lsdc, ilsdc, lodc, LevelScale4x4Comp[uv + 1][intra][qp_rem_matrix[cur_qp_dc]],
                 InvLevelScale4x4Comp[uv + 1][intra][qp_rem_matrix[cur_qp_dc]], LevelOffset4x4Comp[uv + 1][intra][cur_qp_dc],
                 4, currMB, AdaptRndCrWeight, img->max_imgpel_value_comp[1]);
    }
    tile_in(fadj, fa, img->mb_cr_size_y, img->mb_cr_size_x);
    for (b = 0; b < nb; b++) {
      memcpy(acl[b], img->cofAC[4 + (b >> 2) + uv_scale][b & 3][0], sizeof(int) * 16);
      memcpy(acr[b], img->cofAC[4 + (b >> 2) + uv_scale][b & 3][1], sizeof(int) * 16);
    }
    ret = jmo_dct_chroma(&q, &qdc, img->yuv_format, uv, cr_cbp, img->m7[uv + 1], (const jmo_pel (*)[16])img->mpr[uv + 1],
                         img->cofDC[uv + 1][0], img->cofDC[uv + 1][1], acl, acr, recon, fadj, &cbp_blk);
    for (b = 0; b < nb; b++) {
      memcpy(img->cofAC[4 + (b >> 2) + uv_scale][b & 3][0], acl[b], sizeof(int) * 16);
      memcpy(img->cofAC[4 + (b >> 2) + uv_scale][b & 3][1], acr[b], sizeof(int) * 16);
    }
    tile_out(fa, fadj, img->mb_cr_size_y, img->mb_cr_size_x);
    currMB->cbp_blk = cbp_blk;
    for (j = 0; j < img->mb_cr_size_y; j++) for (i = 0; i < img->mb_cr_size_x; i++)
      enc_picture->imgUV[uv][img->pix_c_y + j][img->pix_c_x + i] = recon[j][i];
    return ret;
  }
}

/* ------------------------------------------------------------------ 0x80 primitives */

#define PRIM16(NAME, ORFN)                                                                       \
  void NAME(int (*a)[16], int (*b)[16], int pos_y, int pos_x)                                    \
  {                                                                                              \
    static void (*orig)(int (*)[16], int (*)[16], int, int);                                     \
    n_calls[C_PRIM]++;                                                                           \
    if (!(swap_mask & 0x80)) { if (!orig) orig = next_sym(#NAME); orig(a, b, pos_y, pos_x); return; } \
    ORFN(a, b, pos_y, pos_x);                                                                    \
  }
PRIM16(forward4x4, jmo_forward4x4)
PRIM16(inverse4x4, jmo_inverse4x4)
PRIM16(forward8x8, jmo_forward8x8)
PRIM16(inverse8x8, jmo_inverse8x8)

#define PRIM4(NAME, ORFN)                                                                        \
  void NAME(int (*a)[4], int (*b)[4])                                                            \
  {                                                                                              \
    static void (*orig)(int (*)[4], int (*)[4]);                                                 \
    n_calls[C_PRIM]++;                                                                           \
    if (!(swap_mask & 0x80)) { if (!orig) orig = next_sym(#NAME); orig(a, b); return; }          \
    ORFN(a, b);                                                                                  \
  }
PRIM4(hadamard4x4, jmo_hadamard4x4)
PRIM4(ihadamard4x4, jmo_ihadamard4x4)
