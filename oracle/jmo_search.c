/*
 * jmo_search.c -- ORACLE (test infrastructure): integer and sub-pel block motion search.
 * Restates lencod/src/me_fullsearch.c, lencod/src/me_fullfast.c and the search-centre / table parts
 * of lencod/src/mv-search.c of the reference.
 */
#include <stdlib.h>
#include <string.h>
#include "jmo.h"

static inline int clip3(int lo, int hi, int x) { return x < lo ? lo : (x > hi ? hi : x); }
static inline int clip1(int hi, int x) { return x < 0 ? 0 : (x > hi ? hi : x); }
static inline int imax_(int a, int b) { return a > b ? a : b; }

/* mvbits, mv-search.c:333-341: mvbits[0]=1; |d| in [2^(k-1),2^k) -> 2k+1 */
static int mvbits_slow(int d)
{
  int a = d < 0 ? -d : d, k = 0;
  if (a == 0) return 1;
  while (a) { a >>= 1; k++; }
  return 2 * k + 1;
}
/* JM indexes a table (mvbits[], built once in Init_Motion_Search_Module); so does the port */
#define MVB_RANGE 8192
static signed char mvb_tab[2 * MVB_RANGE + 1];
static int mvb_ready = 0;
static void mvb_init(void)
{
  int d;
  for (d = -MVB_RANGE; d <= MVB_RANGE; d++) mvb_tab[d + MVB_RANGE] = (signed char)mvbits_slow(d);
  mvb_ready = 1;
}
int jmo_mvbits(int d)
{
  if (!mvb_ready) mvb_init();
  if (d < -MVB_RANGE || d > MVB_RANGE) return mvbits_slow(d);
  return mvb_tab[d + MVB_RANGE];
}

/* MV_COST_SMP + WEIGHTED_COST, defines.h:125-128 */
int jmo_mv_cost(int f, int cx, int cy, int px, int py)
{
  return (f * (jmo_mvbits(cx - px) + jmo_mvbits(cy - py))) >> 16;
}

/* spiral_search_x/y, mv-search.c:366-393 (the half-pel spiral is the same table << 1) */
void jmo_spiral(int search_range, short *sx, short *sy, int max_points)
{ /* writes (2*max(1,R)+1)^2 entries; max_points is the caller's capacity (checked) */
  int k = 1, l, i;
  if (max_points < (2 * imax_(1, search_range) + 1) * (2 * imax_(1, search_range) + 1)) abort();
  sx[0] = sy[0] = 0;
  for (l = 1; l <= imax_(1, search_range); l++) {
    for (i = -l + 1; i < l; i++) {
      sx[k] = (short)i;  sy[k++] = (short)-l;
      sx[k] = (short)i;  sy[k++] = (short)l;
    }
    for (i = -l; i <= l; i++) {
      sx[k] = (short)-l; sy[k++] = (short)i;
      sx[k] = (short)l;  sy[k++] = (short)i;
    }
  }
}

/* spiral tables are built once per search range, like JM's Init_Motion_Search_Module (not re-entrant, like JM) */
static short *sp_x = 0, *sp_y = 0;
static int sp_R = -1;
static void spiral_cached(int R, const short **sx, const short **sy)
{
  if (sp_R != R) {
    int n = (2 * imax_(1, R) + 1) * (2 * imax_(1, R) + 1);
    free(sp_x); free(sp_y);
    sp_x = (short *)malloc(sizeof(short) * n); sp_y = (short *)malloc(sizeof(short) * n);
    jmo_spiral(R, sp_x, sp_y, n);
    sp_R = R;
  }
  *sx = sp_x; *sy = sp_y;
}

/* blc_size, configfile.c:805-841 */
void jmo_block_size(int blocktype, int *bsx, int *bsy)
{
  static const int bx[8] = { 16, 16, 16, 8, 8, 8, 4, 4 };
  static const int by[8] = { 16, 16, 8, 16, 8, 4, 8, 4 };
  *bsx = bx[blocktype]; *bsy = by[blocktype];
}

void jmo_dist_from_params(const jmo_me_params *p, const jmo_ref *ref, jmo_dist *d)
{
  memset(d, 0, sizeof(*d));
  d->ref = ref;
  d->chroma_me_weight = p->chroma_me_weight;
  d->max_val = p->max_val; d->max_val_uv = p->max_val_uv;
  d->weight_luma = p->weight_luma; d->offset_luma = p->offset_luma;
  d->wp_luma_round = p->wp_luma_round; d->luma_log_weight_denom = p->luma_log_weight_denom;
  d->weight_cr[0] = p->weight_cr[0]; d->weight_cr[1] = p->weight_cr[1];
  d->offset_cr[0] = p->offset_cr[0]; d->offset_cr[1] = p->offset_cr[1];
  d->wp_chroma_round = p->wp_chroma_round; d->chroma_log_weight_denom = p->chroma_log_weight_denom;
}

/* computeUniPred[level + 3*apply_weights], mv-search.c:400-424 */
int jmo_uni_pred(const jmo_me_params *p, int level, const jmo_dist *d, const jmo_pel *src, int bsy, int bsx,
                    int min_mcost, int cx, int cy)
{
  switch (p->metric[level]) {
  case JMO_ERR_SAD: return p->apply_weights ? jmo_sad_wp(d, src, bsy, bsx, min_mcost, cx, cy)
                                            : jmo_sad(d, src, bsy, bsx, min_mcost, cx, cy);
  case JMO_ERR_SSE: return p->apply_weights ? jmo_sse_wp(d, src, bsy, bsx, min_mcost, cx, cy)
                                            : jmo_sse(d, src, bsy, bsx, min_mcost, cx, cy);
  default:          return p->apply_weights ? jmo_satd_wp(d, src, bsy, bsx, min_mcost, cx, cy)
                                            : jmo_satd(d, src, bsy, bsx, min_mcost, cx, cy);
  }
}

/* mv-search.c:752-762 */
void jmo_search_center(const jmo_me_params *p, int pred_mv_x, int pred_mv_y, int search_range,
                       short *mv_x, short *mv_y)
{
  int mx = pred_mv_x / 4, my = pred_mv_y / 4;          /* C division: truncation toward zero */
  if (!p->rdopt) {
    mx = clip3(-search_range, search_range, mx);
    my = clip3(-search_range, search_range, my);
  }
  mx = clip3(-2047 + search_range, 2047 - search_range, mx);
  my = clip3(p->level_mv_min + search_range, p->level_mv_max - search_range, my);
  *mv_x = (short)mx; *mv_y = (short)my;
}

/* FullPelBlockMotionSearch, me_fullsearch.c:47-155 */
int jmo_fullpel_search(const jmo_me_params *p, const jmo_ref *ref, const jmo_pel *orig_pic, int ref_is_0,
                       int pic_pix_x, int pic_pix_y, int blocktype, int pred_mv_x, int pred_mv_y,
                       short *mv_x, short *mv_y, int search_range, int min_mcost, int lambda_factor)
{
  int pos, cand_x, cand_y, mcost, best_pos = 0, bsx, bsy;
  const int max_pos = (2 * search_range + 1) * (2 * search_range + 1);
  const int pred_x = (pic_pix_x << 2) + pred_mv_x, pred_y = (pic_pix_y << 2) + pred_mv_y;
  const int center_x = pic_pix_x + *mv_x, center_y = pic_pix_y + *mv_y;
  const int check_for_00 = (blocktype == 1 && !p->rdopt && !p->is_b_slice && ref_is_0);   /* :75 */
  const short *sx, *sy;
  jmo_dist d;
  jmo_block_size(blocktype, &bsx, &bsy);
  spiral_cached(search_range, &sx, &sy);
  jmo_dist_from_params(p, ref, &d);
  d.chroma_me = p->chroma_me ? 1 : 0;                  /* mv-search.c:612 */
  d.test8x8 = p->transform8x8_mode && blocktype <= 4;  /* mv-search.c:640 */
  /* :110-118 */
  d.umv = !((center_x > search_range) && (center_x < ref->W - 1 - search_range - bsx) &&
            (center_y > search_range) && (center_y < ref->H - 1 - search_range - bsy));

  for (pos = 0; pos < max_pos; pos++) {
    cand_x = (center_x + sx[pos]) << 2;
    cand_y = (center_y + sy[pos]) << 2;
    mcost = jmo_mv_cost(lambda_factor, cand_x, cand_y, pred_x, pred_y);
    if (check_for_00 && cand_x == pic_pix_x && cand_y == pic_pix_y)     /* :129 (quarter-pel vs pel, as in JM) */
      mcost -= (lambda_factor * 16) >> 16;
    if (mcost >= min_mcost) continue;
    mcost += jmo_uni_pred(p, JMO_F_PEL, &d, orig_pic, bsy, bsx, min_mcost - mcost, cand_x + JMO_PAD4, cand_y + JMO_PAD4);
    if (mcost < min_mcost) { best_pos = pos; min_mcost = mcost; }
  }
  if (best_pos) { *mv_x += sx[best_pos]; *mv_y += sy[best_pos]; }
  return min_mcost;
}

/* SubPelBlockMotionSearch, me_fullsearch.c:341-511.
 * start_me_refinement_hp/qp as mv-search.c:396-397. */
int jmo_subpel_search(const jmo_me_params *p, const jmo_ref *ref, const jmo_pel *orig_pic, int ref_is_0,
                      int pic_pix_x, int pic_pix_y, int blocktype, int pred_mv_x, int pred_mv_y,
                      short *mv_x, short *mv_y, int search_pos2, int search_pos4, int min_mcost,
                      const int *lambda)
{
  static const short s9x[9] = { 0, 0, 0, -1, 1, -1, 1, -1, 1 };   /* spiral positions 0..8 */
  static const short s9y[9] = { 0, -1, 1, -1, -1, 0, 0, 1, 1 };
  const int start_hp = (p->chroma_me == 1 || p->metric[JMO_F_PEL] != p->metric[JMO_H_PEL]) ? 0 : 1;
  const int start_qp = (p->chroma_me == 1 || p->metric[JMO_H_PEL] != p->metric[JMO_Q_PEL]) ? 0 : 1;
  int pos, best_pos, mcost, cand_mv_x, cand_mv_y, cmv_x, cmv_y, bsx, bsy;
  const int check_position0 = (!p->rdopt && !p->is_b_slice && ref_is_0 && blocktype == 1 && *mv_x == 0 && *mv_y == 0);
  const int pic4_pix_x = (pic_pix_x + JMO_PAD) << 2, pic4_pix_y = (pic_pix_y + JMO_PAD) << 2;
  const int max_pos2 = (!start_hp ? imax_(1, search_pos2) : search_pos2);
  int max_pos_x4, max_pos_y4, lambda_factor = lambda[JMO_H_PEL];
  jmo_dist d;
  jmo_block_size(blocktype, &bsx, &bsy);
  max_pos_x4 = (ref->W - bsx + 2 * JMO_PAD) << 2;
  max_pos_y4 = (ref->H - bsy + 2 * JMO_PAD) << 2;
  jmo_dist_from_params(p, ref, &d);
  d.chroma_me = (p->chroma_me == 2) ? 1 : 0;           /* mv-search.c:779 (ME_YUV_FP_SP == 2, global.h:91-92) */
  d.test8x8 = p->transform8x8_mode && blocktype <= 4;

  /* half-pel */
  d.umv = !((pic4_pix_x + *mv_x > 1) && (pic4_pix_x + *mv_x < max_pos_x4 - 1) &&
            (pic4_pix_y + *mv_y > 1) && (pic4_pix_y + *mv_y < max_pos_y4 - 1));
  for (best_pos = 0, pos = start_hp; pos < max_pos2; pos++) {
    cand_mv_x = *mv_x + (s9x[pos] << 1);
    cand_mv_y = *mv_y + (s9y[pos] << 1);
    mcost = jmo_mv_cost(lambda_factor, cand_mv_x, cand_mv_y, pred_mv_x, pred_mv_y);
    if (mcost >= min_mcost) continue;
    cmv_x = cand_mv_x + pic4_pix_x; cmv_y = cand_mv_y + pic4_pix_y;
    mcost += jmo_uni_pred(p, JMO_H_PEL, &d, orig_pic, bsy, bsx, min_mcost - mcost, cmv_x, cmv_y);
    if (pos == 0 && check_position0) mcost -= (lambda_factor * 16) >> 16;   /* :439-442 */
    if (mcost < min_mcost) { min_mcost = mcost; best_pos = pos; }
  }
  if (best_pos) { *mv_x += s9x[best_pos] << 1; *mv_y += s9y[best_pos] << 1; }
  if (!start_qp) min_mcost = JMO_INT_MAX;

  /* quarter-pel */
  d.umv = !((pic4_pix_x + *mv_x > 0) && (pic4_pix_x + *mv_x < max_pos_x4) &&
            (pic4_pix_y + *mv_y > 0) && (pic4_pix_y + *mv_y < max_pos_y4));
  lambda_factor = lambda[JMO_Q_PEL];
  for (best_pos = 0, pos = start_qp; pos < search_pos4; pos++) {
    cand_mv_x = *mv_x + s9x[pos];
    cand_mv_y = *mv_y + s9y[pos];
    mcost = jmo_mv_cost(lambda_factor, cand_mv_x, cand_mv_y, pred_mv_x, pred_mv_y);
    if (mcost >= min_mcost) continue;
    cmv_x = cand_mv_x + pic4_pix_x; cmv_y = cand_mv_y + pic4_pix_y;
    mcost += jmo_uni_pred(p, JMO_Q_PEL, &d, orig_pic, bsy, bsx, min_mcost - mcost, cmv_x, cmv_y);
    if (mcost < min_mcost) { min_mcost = mcost; best_pos = pos; }
  }
  if (best_pos) { *mv_x += s9x[best_pos]; *mv_y += s9y[best_pos]; }
  return min_mcost;
}

/* BlockMotionSearch for SearchMode=-1, from the predictor on: mv-search.c:751-826 */
int jmo_block_search_full(const jmo_me_params *p, const jmo_ref *ref, const jmo_pel *orig_pic, int ref_is_0,
                          int pic_pix_x, int pic_pix_y, int blocktype, int pred_mv_x, int pred_mv_y,
                          int search_range, const int *lambda, short *mv_out, short *mv_int, int *cost_int)
{
  const int start_hp = (p->chroma_me == 1 || p->metric[JMO_F_PEL] != p->metric[JMO_H_PEL]) ? 0 : 1;
  short mv[2];
  int min_mcost = JMO_INT_MAX;
  jmo_search_center(p, pred_mv_x, pred_mv_y, search_range, &mv[0], &mv[1]);
  min_mcost = jmo_fullpel_search(p, ref, orig_pic, ref_is_0, pic_pix_x, pic_pix_y, blocktype,
                                 pred_mv_x, pred_mv_y, &mv[0], &mv[1], search_range, min_mcost, lambda[JMO_F_PEL]);
  if (mv_int) { mv_int[0] = mv[0]; mv_int[1] = mv[1]; }
  if (cost_int) *cost_int = min_mcost;
  mv[0] <<= 2; mv[1] <<= 2;                             /* :770-774 */
  if (!start_hp) min_mcost = JMO_INT_MAX;               /* :785-788 */
  min_mcost = jmo_subpel_search(p, ref, orig_pic, ref_is_0, pic_pix_x, pic_pix_y, blocktype,
                                pred_mv_x, pred_mv_y, &mv[0], &mv[1], 9, 9, min_mcost, lambda);
  mv_out[0] = mv[0]; mv_out[1] = mv[1];
  return min_mcost;
}

/* ------------------------------------------------------------------------------------ fast full search */

/* SetupFastFullPelSearch (non-GEN_ME branch) me_fullfast.c:491-823 + SetupLargerBlocks :210-273.
 * pmv is the 16x16 predictor (:550). SAD only (dist_method = byte_abs when MEErrorMetric[0]==SAD, :512;
 * quad = squared error otherwise). */
void jmo_fastfull_setup(const jmo_me_params *p, const jmo_ref *ref, const jmo_pel *orig_mb,
                        int opix_x, int opix_y, int pmv_x, int pmv_y, int search_range, jmo_fastfull *ff)
{
  const int max_pos = (2 * search_range + 1) * (2 * search_range + 1);
  const int max_width = ref->W - 17, max_height = ref->H - 17;
  const int npts = imax_(9, max_pos);
  short *sx = (short *)malloc(sizeof(short) * (npts + 2)), *sy = (short *)malloc(sizeof(short) * (npts + 2));
  int cx, cy, pos, range_partly_outside, k, x, y, blky;
  int *bs = ff->block_sad;
  const int sse = p->metric[0] != JMO_ERR_SAD;
#define BS(type, blk) (bs + ((long)(type) * 16 + (blk)) * max_pos)
  jmo_spiral(search_range, sx, sy, npts);
  ff->search_range = search_range; ff->max_pos = max_pos;

  cx = pmv_x / 4; cy = pmv_y / 4;
  if (!p->rdopt) { cx = clip3(-search_range, search_range, cx); cy = clip3(-search_range, search_range, cy); }
  cx = clip3(-2047 + search_range, 2047 - search_range, cx);
  cy = clip3(p->level_mv_min + search_range, p->level_mv_max - search_range, cy);
  cx += opix_x; cy += opix_y;
  ff->center_x = cx; ff->center_y = cy;

  range_partly_outside = !(cx >= search_range && cx <= max_width - search_range &&
                           cy >= search_range && cy <= max_height - search_range);
  ff->pos_00 = 0;
  if (!p->rdopt) {
    int rx = opix_x - cx, ry = opix_y - cy;
    for (pos = 0; pos < max_pos; pos++) if (rx == sx[pos] && ry == sy[pos]) { ff->pos_00 = pos; break; }
  }

  for (pos = 0; pos < max_pos; pos++) {
    int abs_y = cy + sy[pos], abs_x = cx + sx[pos];
    int abs_y4 = (abs_y + JMO_PAD) << 2, abs_x4 = (abs_x + JMO_PAD) << 2;
    int umv = 0, xpos, ypos;
    const jmo_pel *refp, *src = orig_mb;
    if (range_partly_outside) umv = !(abs_y >= 0 && abs_y <= max_height && abs_x >= 0 && abs_x <= max_width);
    xpos = abs_x4 >> 2; ypos = abs_y4 >> 2;
    if (umv) { xpos = clip3(0, ref->width_pad, xpos); ypos = clip3(0, ref->height_pad, ypos); }
    refp = ref->luma[0] + (long)ypos * ref->Wp + xpos;  /* plane [0][0] */
    for (blky = 0; blky < 4; blky++) {
      int l[4] = { 0, 0, 0, 0 };
      for (y = 0; y < 4; y++) {
        for (x = 0; x < 16; x++) {
          int rv = p->apply_weights ? clip1(p->max_val, ((p->weight_luma * refp[x] + p->wp_luma_round) >> p->luma_log_weight_denom) + p->offset_luma) : refp[x];
          int df = rv - *src++;
          l[x >> 2] += sse ? df * df : (df < 0 ? -df : df);
        }
        refp += ref->Wp;
      }
      for (k = 0; k < 4; k++) BS(7, blky * 4 + k)[pos] = l[k];
    }
    if (p->chroma_me) {                                 /* :776-814 */
      const jmo_chroma_geom *g = &ref->cg;
      for (k = 0; k < 2; k++) {
        int cxp = abs_x4 >> g->shift_x, cyp = abs_y4 >> g->shift_y, bindex = 0;
        const jmo_pel *cp;
        if (umv) { cxp = clip3(0, ref->width_pad_cr, cxp); cyp = clip3(0, ref->height_pad_cr, cyp); }
        cp = ref->cr[k][(abs_y4 & g->mask_y) * g->sub_x + (abs_x4 & g->mask_x)] + (long)cyp * ref->Wcp + cxp;
        for (blky = 0; blky < 4; blky++) {
          int l[4] = { 0, 0, 0, 0 };
          for (y = 0; y < g->mb_cr_size_y; y += 4) {
            int q, n = g->mb_cr_size_x / 4;
            const jmo_pel *rp = cp;
            for (q = 0; q < 4; q++)
              for (x = 0; x < n; x++) {
                int rv = p->apply_weights ? clip1(p->max_val_uv, ((p->weight_cr[k] * *rp + p->wp_chroma_round) >> p->chroma_log_weight_denom) + p->offset_cr[k]) : *rp;
                int df = rv - *src++;
                rp++;
                l[q] += sse ? df * df : (df < 0 ? -df : df);
              }
            cp += ref->Wcp;
          }
          for (x = 0; x < 4; x++) BS(7, bindex++)[pos] += l[x];
        }
      }
    }
  }

  /* SetupLargerBlocks :210-273 */
  for (pos = 0; pos < max_pos; pos++) {
    int b;
    for (b = 0; b < 4; b++)  { BS(6, b)[pos]     = BS(7, b)[pos]     + BS(7, b + 4)[pos];       /* 4x8 */
                               BS(6, b + 8)[pos] = BS(7, b + 8)[pos] + BS(7, b + 12)[pos]; }
    for (b = 0; b < 16; b += 2) BS(5, b)[pos] = BS(7, b)[pos] + BS(7, b + 1)[pos];                 /* 8x4 */
    for (b = 0; b < 4; b += 2)  { BS(4, b)[pos]     = BS(6, b)[pos]     + BS(6, b + 1)[pos];       /* 8x8 */
                                  BS(4, b + 8)[pos] = BS(6, b + 8)[pos] + BS(6, b + 9)[pos]; }
    BS(3, 0)[pos] = BS(4, 0)[pos] + BS(4, 8)[pos];  BS(3, 2)[pos] = BS(4, 2)[pos] + BS(4, 10)[pos]; /* 8x16 */
    BS(2, 0)[pos] = BS(4, 0)[pos] + BS(4, 2)[pos];  BS(2, 8)[pos] = BS(4, 8)[pos] + BS(4, 10)[pos]; /* 16x8 */
    BS(1, 0)[pos] = BS(3, 0)[pos] + BS(3, 2)[pos];                                                  /* 16x16 */
  }
#undef BS
  free(sx); free(sy);
}

/* FastFullPelBlockMotionSearch, me_fullfast.c:833-903 */
int jmo_fastfull_search(const jmo_me_params *p, const jmo_fastfull *ff, int opix_x, int opix_y,
                        int pic_pix_x, int pic_pix_y, int blocktype, int pred_mv_x, int pred_mv_y,
                        short *mv_x, short *mv_y, int min_mcost, int lambda_factor)
{
  const int max_pos = ff->max_pos;
  const int block_index = (pic_pix_y - opix_y) + ((pic_pix_x - opix_x) >> 2);
  const int *block_sad = ff->block_sad + ((long)blocktype * 16 + block_index) * max_pos;
  const int offset_x = ff->center_x - opix_x, offset_y = ff->center_y - opix_y;
  const int npts = imax_(9, max_pos);
  short *sx = (short *)malloc(sizeof(short) * (npts + 2)), *sy = (short *)malloc(sizeof(short) * (npts + 2));
  int pos, best_pos = 0, mcost;
  jmo_spiral(ff->search_range, sx, sy, npts);
  if (!p->rdopt) {                                      /* :867-876 */
    mcost = block_sad[ff->pos_00] + jmo_mv_cost(lambda_factor, 0, 0, pred_mv_x, pred_mv_y);
    if (mcost < min_mcost) { min_mcost = mcost; best_pos = ff->pos_00; }
  }
  for (pos = 0; pos < max_pos; pos++) {
    if (block_sad[pos] < min_mcost) {
      int cand_x = (offset_x + sx[pos]) << 2, cand_y = (offset_y + sy[pos]) << 2;
      mcost = block_sad[pos] + jmo_mv_cost(lambda_factor, cand_x, cand_y, pred_mv_x, pred_mv_y);
      if (mcost < min_mcost) { min_mcost = mcost; best_pos = pos; }
    }
  }
  *mv_x = (short)(offset_x + sx[best_pos]);
  *mv_y = (short)(offset_y + sy[best_pos]);
  free(sx); free(sy);
  return min_mcost;
}
