/*
 * jmo_epzs.c -- ORACLE (test infrastructure): EPZS motion search (SearchMode = 3).
 * Restates lencod/src/me_epzs.c of the reference: EPZSInit :333, EPZSSliceInit :501 (frame pictures of a
 * frame_mbs_only sequence: the :516-547 scale table and the :986-1030 co-located field), the predictor builders
 * :1061-1490, EPZSPelBlockMotionSearch :1500, EPZSBiPredBlockMotionSearch :1971, EPZSSubPelBlockMotionSearch :2390,
 * EPZSSubPelBlockSearchBiPred :2728. EPZSSubPelGrid = 0 only (mv_rescale = 2, EPZSGrid = 0); field pictures and
 * MBAFF are not restated (the swap harness leaves those configurations to JM, like everything else in this oracle).
 *
 * Unlike the other oracle files this one carries STATE, because the algorithm does: the per-row distortion memory
 * (EPZSDistortion), the spatial-memory vectors (EPZSMotion), the visited map with its 16-bit generation counter
 * (EPZSMap / EPZSBlkCount -- never cleared, so a position visited exactly 65536 calls earlier reads as visited: kept),
 * the POC-distance scales and the scaled co-located vectors of the slice. One jmo_epzs object = the file-static
 * state of me_epzs.c.
 */
#include <stdlib.h>
#include <string.h>
#include "jmo.h"

static inline int iabs_(int x) { return x < 0 ? -x : x; }
static inline int imin_(int a, int b) { return a < b ? a : b; }
static inline int imax_(int a, int b) { return a > b ? a : b; }
static inline int clip3(int lo, int hi, int x) { return x < lo ? lo : (x > hi ? hi : x); }
static inline int rshift_rnd_sf(int x, int a) { return (x + (1 << (a - 1))) >> a; }          /* ifunctions.h:132 */
static inline int rshift_rnd(int x, int a) { return a > 0 ? ((x + (1 << (a - 1))) >> a) : (x << (-a)); } /* ifunctions.h:122 */

#define MV_RESCALE 2                 /* EPZSSubPelGrid = 0, me_epzs.c:343 */
#define MAX_PRED   256

typedef struct { int mv[2]; int start_nmbr; int next_points; } spoint;
typedef struct { int n; spoint pt[12]; int stop_search, next_last, next; } pattern;          /* EPZSStructure */
enum { P_SDIAMOND, P_SQUARE, P_EDIAMOND, P_LDIAMOND, P_SBDIAMOND, P_PMVFAST, P_COUNT };

struct jmo_epzs {
  jmo_epzs_config cfg;
  int medthres[8], maxthres[8], minthres[8], subthres[8];
  pattern pat[P_COUNT];
  int search_pattern, search_pattern_d;
  int nwin, nwin_ext;
  int win[MAX_PRED][2], win_ext[MAX_PRED][2];
  int searcharray;
  short *map;                        /* EPZSMap [searcharray][searcharray] */
  short blk_count;                   /* EPZSBlkCount */
  /* NOT part of JM: a shadow of the map with the FULL ordinal of the search that last stamped each cell, kept beside the 16-bit stamps only to
   * COUNT how often the 16-bit comparison answers "visited" for a cell this search never stamped (a stamp 65536 k searches old, or the zero
   * of a cell never stamped when the counter passes zero). The results never read it. */
  unsigned *map_ord; unsigned ord; long alias_events;
  unsigned *first_ord;               /* ... and the ordinal of the FIRST search that tested or stamped each cell since jmo_epzs_map_set (0: none yet) */
  int ideal_map;                     /* what-if (tools/find_epzs_alias.py only): answer such tests as a map cleared per search would */
  int w4, h4;
  int *distortion;                   /* EPZSDistortion [6][7][w4] */
  short *motion;                     /* EPZSMotion [6][max_refs][7][4][w4][2] */
  short *col_mv;                     /* EPZSCo_located->mv [2][h4][w4][2] */
  int mv_scale[6][JMO_MAX_REFS][JMO_MAX_REFS];
};

/* pattern_data, me_epzs.c:54-78 */
static const int pattern_data[5][12][4] = {
  { {0, 4, 3, 3}, {4, 0, 0, 3}, {0, -4, 1, 3}, {-4, 0, 2, 3} },
  { {0, 4, 7, 3}, {4, 4, 7, 5}, {4, 0, 1, 3}, {4, -4, 1, 5}, {0, -4, 3, 3}, {-4, -4, 3, 5}, {-4, 0, 5, 3}, {-4, 4, 5, 5} },
  { {-4, 4, 10, 5}, {0, 8, 10, 8}, {0, 4, 10, 7}, {4, 4, 1, 5}, {8, 0, 1, 8}, {4, 0, 1, 7},
    {4, -4, 4, 5}, {0, -8, 4, 8}, {0, -4, 4, 7}, {-4, -4, 7, 5}, {-8, 0, 7, 8}, {-4, 0, 7, 7} },
  { {0, 8, 6, 5}, {4, 4, 0, 3}, {8, 0, 0, 5}, {4, -4, 2, 3}, {0, -8, 2, 5}, {-4, -4, 4, 3}, {-8, 0, 4, 5}, {-4, 4, 6, 3} },
  { {0, 8, 6, 12}, {4, 4, 0, 12}, {8, 0, 0, 12}, {4, -4, 2, 12}, {0, -8, 2, 12}, {-4, -4, 4, 12}, {-8, 0, 4, 12}, {-4, 4, 6, 12},
    {0, 2, 6, 12}, {2, 0, 0, 12}, {0, -2, 2, 12}, {-2, 0, 4, 12} }
};
static const int minthres_base[8] = {0, 64, 32, 32, 16, 8, 8, 4};
static const int medthres_base[8] = {0, 256, 128, 128, 64, 32, 32, 16};
static const int maxthres_base[8] = {0, 768, 384, 384, 192, 96, 96, 48};
static const short blk_parent[8] = {1, 1, 1, 1, 2, 4, 4, 5};
static const short search_point_hp[10][2] = {{0, 0}, {-2, 0}, {0, 2}, {2, 0}, {0, -2}, {-2, 2}, {2, 2}, {2, -2}, {-2, -2}, {-2, 2}};
static const short search_point_qp[10][2] = {{0, 0}, {-1, 0}, {0, 1}, {1, 0}, {0, -1}, {-1, 1}, {1, 1}, {1, -1}, {-1, -1}, {-1, 1}};

static void assign_pattern(pattern *p, int n, int type, int stop, int next_last, int next)   /* :219 */
{
  int i;
  p->n = n;
  for (i = 0; i < n; i++) {
    p->pt[i].mv[0] = pattern_data[type][i][0] >> MV_RESCALE;
    p->pt[i].mv[1] = pattern_data[type][i][1] >> MV_RESCALE;
    p->pt[i].start_nmbr = pattern_data[type][i][2];
    p->pt[i].next_points = pattern_data[type][i][3];
  }
  p->stop_search = stop; p->next_last = next_last; p->next = next;
}

static int round_log2(int v)          /* :241 */
{
  int r = 0, sq = v * v;
  while ((1 << (r + 1)) <= sq) r++;
  return (r + 1) >> 1;
}

static int window_init(int search_range, int (*pt)[2], int mode)     /* EPZSWindowPredictorInit :261 (search_range_qpel = 0) */
{
  int pos, i, n = 0;
  for (pos = round_log2(search_range) - 2; pos > -1; pos--) {
    const int sp = search_range >> pos, fsp = (3 * sp + 1) >> 1;
    for (i = 1; i >= -1; i -= 2) {
      pt[n][0] = i * sp;  pt[n++][1] = 0;
      pt[n][0] = i * sp;  pt[n++][1] = i * sp;
      pt[n][0] = 0;       pt[n++][1] = i * sp;
      pt[n][0] = -i * sp; pt[n++][1] = i * sp;
    }
    if (mode)
      for (i = 1; i >= -1; i -= 2) {
        pt[n][0] = i * fsp;  pt[n++][1] = -i * sp;
        pt[n][0] = i * fsp;  pt[n++][1] = 0;
        pt[n][0] = i * fsp;  pt[n++][1] = i * sp;
        pt[n][0] = i * sp;   pt[n++][1] = i * fsp;
        pt[n][0] = 0;        pt[n++][1] = i * fsp;
        pt[n][0] = -i * sp;  pt[n++][1] = i * fsp;
      }
  }
  return n;
}

jmo_epzs *jmo_epzs_create(const jmo_epzs_config *c)                   /* EPZSInit :333 */
{
  jmo_epzs *e = (jmo_epzs *)calloc(1, sizeof(*e));
  const int pel_error_me = 1 << (c->bitdepth_luma - 8), pel_error_me_cr = 1 << (c->bitdepth_chroma - 8);
  const double chroma_weight = c->chroma_me ? pel_error_me_cr * c->chroma_me_weight * (double)(c->width_cr * c->height_cr) / (double)(c->width * c->height) : 0;
  int i;
  static const int pat_of[6] = {P_SDIAMOND, P_SQUARE, P_EDIAMOND, P_LDIAMOND, P_SBDIAMOND, P_PMVFAST};
  e->cfg = *c;
  e->searcharray = c->bipred_me ? 2 * imax_(c->search_range, c->bipred_search_range) + 1 : 2 * c->search_range + 1;
  for (i = 0; i < 8; i++) {
    e->medthres[i] = c->med_scale * (medthres_base[i] * pel_error_me + (int)(medthres_base[i] * chroma_weight + 0.5));
    e->maxthres[i] = c->max_scale * (maxthres_base[i] * pel_error_me + (int)(maxthres_base[i] * chroma_weight + 0.5));
    e->minthres[i] = c->min_scale * (minthres_base[i] * pel_error_me + (int)(minthres_base[i] * chroma_weight + 0.5));
    e->subthres[i] = c->subpel_scale * (medthres_base[i] * pel_error_me + (int)(medthres_base[i] * chroma_weight + 0.5));
  }
  assign_pattern(&e->pat[P_SDIAMOND], 4, 0, 1, 1, P_SDIAMOND);         /* :367-378 */
  assign_pattern(&e->pat[P_SQUARE], 8, 1, 1, 1, P_SQUARE);
  assign_pattern(&e->pat[P_EDIAMOND], 12, 2, 1, 1, P_EDIAMOND);
  assign_pattern(&e->pat[P_LDIAMOND], 8, 3, 1, 1, P_LDIAMOND);
  assign_pattern(&e->pat[P_SBDIAMOND], 12, 4, 0, 1, P_SDIAMOND);
  assign_pattern(&e->pat[P_PMVFAST], 8, 3, 0, 1, P_SDIAMOND);
  e->nwin = window_init(c->search_range, e->win, 0);
  e->nwin_ext = window_init(c->search_range, e->win_ext, 1);
  e->w4 = c->width / 4; e->h4 = c->height / 4;
  e->distortion = (int *)calloc((size_t)6 * 7 * e->w4, sizeof(int));
  e->map = (short *)calloc((size_t)e->searcharray * e->searcharray, sizeof(short));
  e->map_ord = (unsigned *)calloc((size_t)e->searcharray * e->searcharray, sizeof(unsigned));
  e->first_ord = (unsigned *)calloc((size_t)e->searcharray * e->searcharray, sizeof(unsigned));
  if (c->spatial_mem) e->motion = (short *)calloc((size_t)6 * c->max_refs * 7 * 4 * e->w4 * 2, sizeof(short));
  if (c->temporal) e->col_mv = (short *)calloc((size_t)2 * e->h4 * e->w4 * 2, sizeof(short));
  e->search_pattern = pat_of[(c->pattern >= 1 && c->pattern <= 5) ? c->pattern : 0];        /* :410-431 */
  e->search_pattern_d = pat_of[(c->dual >= 2 && c->dual <= 6) ? c->dual - 1 : 0];          /* :433-454 */
  return e;
}

void jmo_epzs_destroy(jmo_epzs *e)
{
  if (!e) return;
  free(e->distortion); free(e->map); free(e->map_ord); free(e->first_ord); free(e->motion); free(e->col_mv); free(e);
}

int *jmo_epzs_distortion_row(jmo_epzs *e, int list, int blocktype_m1) { return e->distortion + ((size_t)list * 7 + blocktype_m1) * e->w4; }
int jmo_epzs_threshold(const jmo_epzs *e, int which, int blocktype)
{ return which == 0 ? e->minthres[blocktype] : which == 1 ? e->medthres[blocktype] : which == 2 ? e->maxthres[blocktype] : e->subthres[blocktype]; }
int jmo_epzs_mv_scale(const jmo_epzs *e, int list, int i, int k) { return e->mv_scale[list][i][k]; }
const short *jmo_epzs_colocated(const jmo_epzs *e) { return e->col_mv; }

/* EPZSSliceInit :501 -- frame picture, frame_mbs_only: scale table :516-547, epzs_scale :557-607, co-located :986-1030 */
void jmo_epzs_slice_init(jmo_epzs *e, const jmo_epzs_slice *s)
{
  const int list = s->is_b_slice ? 1 : 0;
  int i, j, k, iTRb, iTRp, prescale, iref;
  int epzs_scale[2][2][JMO_MAX_LIST];
  for (j = 0; j < 2; j++)
    for (k = 0; k < s->list_size[j]; k++)
      for (i = 0; i < s->list_size[j]; i++) {
        iTRb = clip3(-128, 127, s->poc - s->list_poc[j][i]);
        iTRp = clip3(-128, 127, s->poc - s->list_poc[j][k]);
        if (iTRp != 0) {
          prescale = (16384 + iabs_(iTRp / 2)) / iTRp;
          e->mv_scale[j][i][k] = clip3(-2048, 2047, rshift_rnd_sf(iTRb * prescale, 6));
        } else e->mv_scale[j][i][k] = 256;
      }
  if (!e->cfg.temporal) return;
  for (j = 0; j < 2; j++) for (i = 0; i < JMO_MAX_LIST; i++) { epzs_scale[0][j][i] = 256; epzs_scale[1][j][i] = 256; }   /* only [..][0..5] in JM; same values */
  for (i = 0; i < s->list_size[0]; i++) {                               /* j = 0 only without MBAFF */
    iTRb = clip3(-128, 127, s->poc - s->list_poc[0][i]);
    iTRp = clip3(-128, 127, s->list_poc[list][0] - s->list_poc[0][i]);
    if (iTRp != 0) { prescale = (16384 + iabs_(iTRp / 2)) / iTRp; prescale = clip3(-2048, 2047, rshift_rnd_sf(iTRb * prescale, 6)); }
    else prescale = 256;
    epzs_scale[0][0][i] = rshift_rnd_sf(e->mv_scale[0][0][i] * prescale, 8);
    epzs_scale[0][1][i] = prescale - 256;
    if (s->list_size[list] > 1) {
      iTRp = clip3(-128, 127, s->list_poc[list][1] - s->list_poc[0][i]);
      if (iTRp != 0) { prescale = (16384 + iabs_(iTRp / 2)) / iTRp; prescale = clip3(-2048, 2047, rshift_rnd_sf(iTRb * prescale, 6)); }
      else prescale = 256;
      epzs_scale[1][0][i] = rshift_rnd_sf(e->mv_scale[0][1][i] * prescale, 8);
      epzs_scale[1][1][i] = prescale - 256;
    } else { epzs_scale[1][0][i] = epzs_scale[0][0][i]; epzs_scale[1][1][i] = epzs_scale[0][1][i]; }
  }
  /* :986-1030 */
  for (j = 0; j < e->h4; j++)
    for (i = 0; i < e->w4; i++) {
      const size_t at = (size_t)j * e->w4 + i;
      int ts0 = 256, ts1 = 0, loffset = 0;
      short *m0 = e->col_mv + at * 2, *m1 = e->col_mv + ((size_t)e->h4 * e->w4 + at) * 2;
      if (s->col_ref_id[0][at] < 0 && s->list_size[0] > 1) loffset = 1;
      if (s->col_ref_id[loffset][at] != -1) {
        const short *fm = s->col_mv[loffset] + at * 2;
        for (iref = 0; iref < imin_(s->num_ref_idx_l0_active, s->list_size[0]); iref++)
          if (s->ref_pic_num_l0[iref] == s->col_ref_id[loffset][at]) { ts0 = epzs_scale[loffset][0][iref]; ts1 = epzs_scale[loffset][1][iref]; break; }
        m0[0] = (short)clip3(-32768, 32767, rshift_rnd_sf(ts0 * fm[0], 8));
        m0[1] = (short)clip3(-32768, 32767, rshift_rnd_sf(ts0 * fm[1], 8));
        m1[0] = (short)clip3(-32768, 32767, rshift_rnd_sf(ts1 * fm[0], 8));
        m1[1] = (short)clip3(-32768, 32767, rshift_rnd_sf(ts1 * fm[1], 8));
      } else { m0[0] = m0[1] = m1[0] = m1[1] = 0; }
    }
}

/* ------------------------------------------------------------------ the integer walk, shared by the uni- and bi-predictive forms */

typedef struct {
  jmo_epzs *e;
  const jmo_me_params *p;
  jmo_dist d;                         /* uni */
  jmo_bipred *b;                      /* bi  */
  const jmo_pel *cur;
  int bsx, bsy, lambda, img_w, img_h;
  int pred_x, pred_y;                 /* uni: predictor; bi: predictor 2 (of the swept vector) */
  int fixed_cost, c1x, c1y;           /* bi: mv cost and position of the fixed block */
} walk_ctx;

static int cost_mv(const walk_ctx *w, int cand_x, int cand_y)
{
  int c = jmo_mv_cost(w->lambda, cand_x, cand_y, w->pred_x, w->pred_y);
  return w->b ? c + w->fixed_cost : c;
}
static int cost_dist(walk_ctx *w, int bound, int cand_x, int cand_y)
{
  if (w->b) return jmo_bipred_dist(w->b, w->b->metric[JMO_F_PEL], w->cur, w->bsy, w->bsx, bound, w->c1x + JMO_PAD4, w->c1y + JMO_PAD4, cand_x + JMO_PAD4, cand_y + JMO_PAD4);
  /* CHECK_RANGE (me_epzs.h:23) compares the QUARTER-pel candidate with pel-unit picture sizes, as JM does */
  w->d.umv = !((cand_x >= 0) && (cand_x < w->img_w - w->bsx) && (cand_y >= 0) && (cand_y < w->img_h - w->bsy));
  return jmo_uni_pred(w->p, JMO_F_PEL, &w->d, w->cur, w->bsy, w->bsx, bound, cand_x + JMO_PAD4, cand_y + JMO_PAD4);
}

#define MAP(e, y, x) ((e)->map[(size_t)(y) * (e)->searcharray + (x)])
#define MAP_ORD(e, y, x) ((e)->map_ord[(size_t)(y) * (e)->searcharray + (x)])
#define FIRST_ORD(e, y, x) ((e)->first_ord[(size_t)(y) * (e)->searcharray + (x)])
/* the shadow's bookkeeping for one test of cell (y, x): an alias is a cell the 16-bit stamp calls visited that this search did not stamp */
static inline void map_shadow(jmo_epzs *e, int y, int x)
{
  if (MAP(e, y, x) == e->blk_count && MAP_ORD(e, y, x) != e->ord) {
    e->alias_events++;
    if (e->ideal_map) MAP(e, y, x) = (short)(e->blk_count - 1);
  }
  MAP_ORD(e, y, x) = e->ord;
  if (!FIRST_ORD(e, y, x)) FIRST_ORD(e, y, x) = e->ord;
}

/* the refinement loop, me_epzs.c:1801-1942 (uni) / :2215-2341 (bi). Returns 1 when the uni form's ref > 0 early return fires. */
static int refine(walk_ctx *w, int pat0, int pic_pix_x, int pic_pix_y, const short mv[2], int search_range, int blocktype, int ref,
                  int *tempmv, int *tempmv2, int *min_mcost, int stop_criterion, int check_median, int is_p_slice, const int *prev_sad_at)
{
  jmo_epzs *e = w->e;
  const int map_cx = search_range - mv[0], map_cy = search_range - mv[1];
  int pf = pat0, total = e->pat[pf].n, center_x = tempmv[0], center_y = tempmv[1];
  int pattern_stop = 0, point = 0, next_last = 0, dir = 0, check_pts, tmv[2], cand_x, cand_y, mcost;
  for (;;) {
    do {
      check_pts = total;
      do {
        tmv[0] = center_x + e->pat[pf].pt[point].mv[0];
        tmv[1] = center_y + e->pat[pf].pt[point].mv[1];
        cand_x = (pic_pix_x + tmv[0]) << MV_RESCALE;
        cand_y = (pic_pix_y + tmv[1]) << MV_RESCALE;
        if (iabs_(tmv[0] - mv[0]) <= search_range && iabs_(tmv[1] - mv[1]) <= search_range) {
          map_shadow(e, map_cy + tmv[1], map_cx + tmv[0]);
          if (MAP(e, map_cy + tmv[1], map_cx + tmv[0]) != e->blk_count) MAP(e, map_cy + tmv[1], map_cx + tmv[0]) = e->blk_count;
          else {
            if (++point >= e->pat[pf].n) point -= e->pat[pf].n;
            check_pts--;
            continue;
          }
          mcost = cost_mv(w, cand_x, cand_y);
          if (mcost < *min_mcost) {
            mcost += cost_dist(w, *min_mcost - mcost, cand_x, cand_y);
            if (mcost < *min_mcost) { *min_mcost = mcost; tempmv[0] = tmv[0]; tempmv[1] = tmv[1]; dir = point; }
          }
        }
        if (++point >= e->pat[pf].n) point -= e->pat[pf].n;
        check_pts--;
      } while (check_pts > 0);
      if (next_last || (tempmv[0] == center_x && tempmv[1] == center_y)) {
        pattern_stop = e->pat[pf].stop_search;
        pf = e->pat[pf].next;
        total = e->pat[pf].n;
        next_last = e->pat[pf].next_last;
        dir = 0; point = 0;
      } else {
        total = e->pat[pf].pt[dir].next_points;
        point = e->pat[pf].pt[dir].start_nmbr;
        center_x = tempmv[0]; center_y = tempmv[1];
      }
    } while (pattern_stop != 1);

    if (!w->b && ref > 0 &&                                                                     /* :1894-1911 (frame pictures) */
        ((4 * *prev_sad_at < *min_mcost) || ((3 * *prev_sad_at < *min_mcost) && (*prev_sad_at <= stop_criterion))))
      return 1;

    /* second best predictor :1914-1940 / :2314-2339 */
    if (!(check_median && (w->b ? blocktype < 5 : (is_p_slice || blocktype < 5)) && *min_mcost > stop_criterion && e->cfg.dual > 0)) break;
    point = 0; pattern_stop = 0; dir = 0; next_last = 0;
    if ((tempmv[0] == 0 && tempmv[1] == 0) || (tempmv[0] == mv[0] && tempmv[1] == mv[1])) {
      if (iabs_(tempmv[0] - mv[0]) < (2 << (2 - MV_RESCALE)) && iabs_(tempmv[1] - mv[1]) < (2 << (2 - MV_RESCALE))) pf = P_SDIAMOND;
      else pf = P_SQUARE;
    } else pf = e->search_pattern_d;
    total = e->pat[pf].n;
    center_x = tempmv2[0]; center_y = tempmv2[1];
    check_median = 0;
  }
  return 0;
}

/* the predictor scan, :1746-1795 (uni) / :2161-2211 (bi) */
static int scan_predictors(walk_ctx *w, int (*pred)[2], int prednum, int pic_pix_x, int pic_pix_y, const short mv[2], int search_range,
                           int *tempmv, int *tempmv2, int *min_mcost, int *second_mcost)
{
  jmo_epzs *e = w->e;
  const int map_cx = search_range - mv[0], map_cy = search_range - mv[1];
  int pos, check_median = 0;
  for (pos = 0; pos < prednum; pos++) {
    const int tx = pred[pos][0], ty = pred[pos][1];
    const int outside = iabs_(tx - mv[0]) > search_range || iabs_(ty - mv[1]) > search_range;
    int cand_x, cand_y, mcost;
    if (outside && (!w->b || tx || ty)) continue;        /* the bi form still tests an out-of-range ZERO vector (:2166) */
    if (!outside) {
      map_shadow(e, map_cy + ty, map_cx + tx);
      if (MAP(e, map_cy + ty, map_cx + tx) == e->blk_count) continue;
      MAP(e, map_cy + ty, map_cx + tx) = e->blk_count;
    }
    cand_x = (pic_pix_x + tx) << MV_RESCALE; cand_y = (pic_pix_y + ty) << MV_RESCALE;
    mcost = cost_mv(w, cand_x, cand_y);
    if (mcost >= *second_mcost) continue;
    mcost += cost_dist(w, *second_mcost - mcost, cand_x, cand_y);
    if (mcost < *min_mcost) {
      tempmv2[0] = tempmv[0]; tempmv2[1] = tempmv[1];
      tempmv[0] = tx; tempmv[1] = ty;
      *second_mcost = *min_mcost; *min_mcost = mcost; check_median = 1;
    } else if (mcost < *second_mcost) {
      tempmv2[0] = tx; tempmv2[1] = ty; *second_mcost = mcost; check_median = 1;
    }
  }
  return check_median;
}

/* block_c fix-up shared by both forms (:1660-1688, :2137-2151); returns block_available_right */
static int fix_block_c(int mb_x, int mb_y, int bsx, int mb_available_right, int *c_avail)
{
  int right;
  if (mb_y > 0) {
    if (mb_x < 8) {
      if (mb_y == 8) { right = (bsx != 16) || mb_available_right; if (bsx == 16) *c_avail = 0; }
      else { right = (mb_x + bsx != 8) || mb_available_right; if (mb_x + bsx == 8) *c_avail = 0; }
    } else { right = (mb_x + bsx != 16) || mb_available_right; if (mb_x + bsx == 16) *c_avail = 0; }
  } else right = (mb_x + bsx != 16) || mb_available_right;
  return right;
}

/* EPZSSpatialPredictors :1061 (non-MBAFF branch); pred[0..4]; returns invalid_refs */
static int spatial_predictors(const jmo_epzs *e, const jmo_epzs_nbr *nb, int c_avail, int list, int ref, int (*pred)[2])
{
  /* JM indexes mot_scale[refX] with refX = -1 for an available neighbour that holds no vector of this list: that reads the
   * element BEFORE the row (the last one of the previous row; for list 0 / ref 0 whatever precedes the table in memory). The vector
   * it multiplies is the zero vector JM keeps for such blocks, so the product is 0 either way; the flat index keeps the read in
   * the table and returns 0 only where JM would leave it */
  const int *flat = &e->mv_scale[0][0][0];
  const long row = ((long)list * JMO_MAX_REFS + ref) * JMO_MAX_REFS;
  const int sh = 8 + MV_RESCALE;
  const int refA = nb->available[0] ? nb->ref[0] : -1, refB = nb->available[1] ? nb->ref[1] : -1;
  const int refC = c_avail ? nb->ref[2] : -1, refD = nb->available[3] ? nb->ref[3] : -1;
  pred[0][0] = pred[0][1] = 0;
#define SC(r) (row + (r) < 0 ? 0 : flat[row + (r)])
  if (nb->available[0]) { pred[1][0] = rshift_rnd_sf(SC(refA) * nb->mv[0][0], sh); pred[1][1] = rshift_rnd_sf(SC(refA) * nb->mv[0][1], sh); }
  else { pred[1][0] = 12 >> MV_RESCALE; pred[1][1] = 0; }
  if (nb->available[1]) { pred[2][0] = rshift_rnd_sf(SC(refB) * nb->mv[1][0], sh); pred[2][1] = rshift_rnd_sf(SC(refB) * nb->mv[1][1], sh); }
  else { pred[2][0] = 0; pred[2][1] = 12 >> MV_RESCALE; }
  if (c_avail) { pred[3][0] = rshift_rnd_sf(SC(refC) * nb->mv[2][0], sh); pred[3][1] = rshift_rnd_sf(SC(refC) * nb->mv[2][1], sh); }
  else { pred[3][0] = -(12 >> MV_RESCALE); pred[3][1] = 0; }
  if (nb->available[3]) { pred[4][0] = rshift_rnd_sf(SC(refD) * nb->mv[3][0], sh); pred[4][1] = rshift_rnd_sf(SC(refD) * nb->mv[3][1], sh); }
  else { pred[4][0] = 0; pred[4][1] = -(12 >> MV_RESCALE); }
#undef SC
  return (refA == -1) + (refB == -1) + (refC == -1 && refD == -1);
}

#define ADD_PRED(X, Y) do { pred[prednum][0] = (X); pred[prednum][1] = (Y); prednum += ((pred[prednum][0] | pred[prednum][1]) != 0); } while (0)

/* EPZSPelBlockMotionSearch :1500 */
int jmo_epzs_pel_search(jmo_epzs *e, const jmo_me_params *p, const jmo_ref *ref_pic, const jmo_pel *cur_pic, int ref, int list,
                        const jmo_epzs_nbr *nb, const short (*allmv)[8][2], int is_p_slice, int current_mb_nr, int opix_x, int opix_y,
                        int pic_pix_x, int pic_pix_y, int blocktype, const short pred_mv[2], short mv[2], int search_range,
                        int min_mcost, int lambda_factor)
{
  int bsx, bsy, prednum = 5, pred[MAX_PRED][2];
  int tempmv[2] = {mv[0], mv[1]}, tempmv2[2] = {0, 0}, second_mcost = JMO_INT_MAX, stop, check_median = 0, invalid_refs;
  int *prev_sad = jmo_epzs_distortion_row(e, list, blocktype - 1);
  short *motion = NULL;
  walk_ctx w;
  (void)min_mcost;
  jmo_block_size(blocktype, &bsx, &bsy);
  {
    const int bshx = bsx >> 2, bshy = bsy >> 2, mb_x = pic_pix_x - opix_x, mb_y = pic_pix_y - opix_y;
    const int px2 = pic_pix_x >> 2, py2 = pic_pix_y >> 2, block_y = mb_y >> 2;
    const int center_x = pic_pix_x + mv[0], center_y = pic_pix_y + mv[1];
    int cand_x = center_x << MV_RESCALE, cand_y = center_y << MV_RESCALE;
    memset(&w, 0, sizeof(w));
    w.e = e; w.p = p; w.cur = cur_pic; w.bsx = bsx; w.bsy = bsy; w.lambda = lambda_factor;
    w.img_w = ref_pic->W; w.img_h = ref_pic->H;
    w.pred_x = (pic_pix_x << 2) + pred_mv[0]; w.pred_y = (pic_pix_y << 2) + pred_mv[1];
    jmo_dist_from_params(p, ref_pic, &w.d);
    w.d.chroma_me = p->chroma_me ? 1 : 0;                 /* mv-search.c:612 */
    w.d.test8x8 = p->transform8x8_mode && blocktype <= 4;  /* mv-search.c:640 */
    stop = e->medthres[blocktype];
    e->blk_count = (short)(e->blk_count + 1);              /* :1550 */
    e->ord++;
    if (e->cfg.spatial_mem) motion = e->motion + (((((size_t)list * e->cfg.max_refs + ref) * 7 + (blocktype - 1)) * 4 + block_y) * e->w4 + px2) * 2;
    MAP(e, search_range, search_range) = e->blk_count;     /* :1598 */
    MAP_ORD(e, search_range, search_range) = e->ord;
    if (!FIRST_ORD(e, search_range, search_range)) FIRST_ORD(e, search_range, search_range) = e->ord;
    min_mcost = cost_mv(&w, cand_x, cand_y);
    min_mcost += cost_dist(&w, JMO_INT_MAX, cand_x, cand_y);

    if (ref > 0 && prev_sad[px2] < e->medthres[blocktype] && prev_sad[px2] < min_mcost) {        /* :1608-1623 */
      if (motion) { motion[0] = (short)tempmv[0]; motion[1] = (short)tempmv[1]; }
      return min_mcost;
    }
    if (min_mcost > stop) {
      const int mb_avail_right = (opix_x >> 4) < (ref_pic->W >> 4) - 1, mb_avail_below = (opix_y >> 4) < (ref_pic->H >> 4) - 1;
      int c_avail = nb->available[2];
      const int blk_right = fix_block_c(mb_x, mb_y, bsx, mb_avail_right, &c_avail);
      const int blk_below = (mb_y + bsy != 16) || mb_avail_below;
      const int sadA = nb->available[0] ? prev_sad[px2 - bshx] : JMO_INT_MAX;
      const int sadB = nb->available[1] ? prev_sad[px2] : JMO_INT_MAX;
      const int sadC = c_avail ? prev_sad[px2 + bshx] : JMO_INT_MAX;
      int pat0 = e->search_pattern;
      stop = imin_(sadA, imin_(sadB, sadC));
      stop = imax_(stop, e->minthres[blocktype]);
      stop = imin_(stop, e->maxthres[blocktype]);
      stop = (9 * imax_(e->medthres[blocktype], stop) + 2 * e->medthres[blocktype]) >> 3;          /* :1698 */

      invalid_refs = spatial_predictors(e, nb, c_avail, list, ref, pred);
      if (e->cfg.spatial_mem) {                            /* EPZSSpatialMemPredictors :1253 */
        const short *m = e->motion + ((((size_t)list * e->cfg.max_refs + ref) * 7 + (blocktype - 1)) * 4) * e->w4 * 2;
        const int iw = ref_pic->W >> 2, by = block_y;
#define MOT(r, x, c) m[((size_t)(r) * e->w4 + (x)) * 2 + (c)]
        ADD_PRED(px2 > 0 ? MOT(by, px2 - bshx, 0) : 0, px2 > 0 ? MOT(by, px2 - bshx, 1) : 0);
        ADD_PRED(by > 0 ? MOT(by - bshy, px2, 0) : MOT(4 - bshy, px2, 0), by > 0 ? MOT(by - bshy, px2, 1) : MOT(4 - bshy, px2, 1));
        ADD_PRED(px2 + bshx < iw ? (by > 0 ? MOT(by - bshy, px2 + bshx, 0) : MOT(4 - bshy, px2 + bshx, 0)) : 0,
                 px2 + bshx < iw ? (by > 0 ? MOT(by - bshy, px2 + bshx, 1) : MOT(4 - bshy, px2 + bshx, 1)) : 0);
#undef MOT
      }
      if (e->cfg.temporal) {                               /* EPZSTemporalPredictors :1332 */
        const int sc = e->mv_scale[list][ref][0], sh = 8 + MV_RESCALE;
        const short *col = e->col_mv + (size_t)list * e->h4 * e->w4 * 2;
#define COL(y, x, c) col[((size_t)(y) * e->w4 + (x)) * 2 + (c)]
#define ADD_COL(y, x) ADD_PRED(rshift_rnd_sf(sc * COL(y, x, 0), sh), rshift_rnd_sf(sc * COL(y, x, 1), sh))
        ADD_COL(py2, px2);
        if (min_mcost > stop && ref < 2) {
          if (nb->available[0]) {
            ADD_COL(py2, px2 - 1);
            if (nb->available[1]) ADD_COL(py2 - 1, px2 - 1);
            if (blk_below) ADD_COL(py2 + bshy, px2 - 1);
          }
          if (nb->available[1]) ADD_COL(py2 - 1, px2);
          if (blk_right) {
            ADD_COL(py2, px2 + bshx);
            if (nb->available[1]) ADD_COL(py2 - 1, px2 + bshx);
            if (blk_below) ADD_COL(py2 + bshy, px2 + bshx);
          }
          if (blk_below) ADD_COL(py2 + bshy, px2);
        }
#undef ADD_COL
#undef COL
      }
      /* window predictors :1727-1733 (frame pictures) */
      if (min_mcost > stop && (ref < 2 && blocktype < 5) && (e->cfg.fixed > 1 || (e->cfg.fixed && is_p_slice))) {
        const int ext = (blocktype < 5) && (invalid_refs > 2) && (ref < 1);
        const int n = ext ? e->nwin_ext : e->nwin, (*wp)[2] = ext ? e->win_ext : e->win;
        int k;
        for (k = 0; k < n; k++) { pred[prednum][0] = mv[0] + wp[k][0]; pred[prednum][1] = mv[1] + wp[k][1]; prednum++; }
      }
      /* block-type / reference predictors :1740-1744, EPZSBlockTypePredictors :1433 */
      if ((ref == 0 || min_mcost > stop) && current_mb_nr != 0) {
        const int sh = 8 + MV_RESCALE;
        ADD_PRED(rshift_rnd(allmv[ref][blk_parent[blocktype]][0], MV_RESCALE), rshift_rnd(allmv[ref][blk_parent[blocktype]][1], MV_RESCALE));
        if (ref > 0 && blocktype < 5) {
          ADD_PRED(rshift_rnd_sf(e->mv_scale[list][ref][ref - 1] * allmv[ref - 1][blocktype][0], sh), rshift_rnd_sf(e->mv_scale[list][ref][ref - 1] * allmv[ref - 1][blocktype][1], sh));
          ADD_PRED(rshift_rnd_sf(e->mv_scale[list][ref][0] * allmv[0][blocktype][0], sh), rshift_rnd_sf(e->mv_scale[list][ref][0] * allmv[0][blocktype][1], sh));
        }
        if (blocktype != 1) ADD_PRED(rshift_rnd(allmv[ref][1][0], MV_RESCALE), rshift_rnd(allmv[ref][1][1], MV_RESCALE));
        if (blocktype != 4) ADD_PRED(rshift_rnd(allmv[ref][4][0], MV_RESCALE), rshift_rnd(allmv[ref][4][1], MV_RESCALE));
      }
      check_median = scan_predictors(&w, pred, prednum, pic_pix_x, pic_pix_y, mv, search_range, tempmv, tempmv2, &min_mcost, &second_mcost);

      if (min_mcost > stop) {                              /* :1801-1818 */
        if (e->cfg.pattern != 0) {
          if (min_mcost < stop + ((3 * e->medthres[blocktype]) >> 1)) {
            if ((tempmv[0] == 0 && tempmv[1] == 0) || (iabs_(tempmv[0] - mv[0]) < (2 << (2 - MV_RESCALE)) && iabs_(tempmv[1] - mv[1]) < (2 << (2 - MV_RESCALE)))) pat0 = P_SDIAMOND;
            else pat0 = P_SQUARE;
          } else if (blocktype > 5 || (ref > 0 && blocktype != 1)) pat0 = P_SQUARE;
          else pat0 = e->search_pattern;
        }
        if (refine(&w, pat0, pic_pix_x, pic_pix_y, mv, search_range, blocktype, ref, tempmv, tempmv2, &min_mcost, stop, check_median, is_p_slice, &prev_sad[px2])) {
          mv[0] = (short)tempmv[0]; mv[1] = (short)tempmv[1];
          if (motion) { motion[0] = (short)tempmv[0]; motion[1] = (short)tempmv[1]; }
          return min_mcost;
        }
      }
    }
    if (ref == 0 || prev_sad[px2] > min_mcost) prev_sad[px2] = min_mcost;                        /* :1945 */
    if (motion) { motion[0] = (short)tempmv[0]; motion[1] = (short)tempmv[1]; }
    mv[0] = (short)tempmv[0]; mv[1] = (short)tempmv[1];
    return min_mcost;
  }
}

/* EPZSBiPredBlockMotionSearch :1971. b: as for jmo_fullpel_bipred (ref1 = listX[list][ref] holds the FIXED block s_mv, ref2 the swept mv). */
int jmo_epzs_bipred_search(jmo_epzs *e, jmo_bipred *b, const jmo_pel *cur_pic, int ref, int list, const jmo_epzs_nbr *nb,
                           int opix_x, int opix_y, int pic_pix_x, int pic_pix_y, int blocktype, const short pred_mv1[2], const short pred_mv2[2],
                           short mv[2], const short s_mv[2], int search_range, int min_mcost, int lambda_factor)
{
  int bsx, bsy, pred[MAX_PRED][2];
  int tempmv[2] = {mv[0], mv[1]}, tempmv2[2] = {0, 0}, second_mcost = JMO_INT_MAX, stop, check_median;
  walk_ctx w;
  (void)min_mcost;
  jmo_block_size(blocktype, &bsx, &bsy);
  {
    const int mb_x = pic_pix_x - opix_x, mb_y = pic_pix_y - opix_y;
    const int center2_x = pic_pix_x + mv[0], center2_y = pic_pix_y + mv[1], center1_x = pic_pix_x + s_mv[0], center1_y = pic_pix_y + s_mv[1];
    const int W = b->ref1->W, H = b->ref1->H;
    memset(&w, 0, sizeof(w));
    w.e = e; w.b = b; w.cur = cur_pic; w.bsx = bsx; w.bsy = bsy; w.lambda = lambda_factor;
    w.pred_x = (pic_pix_x << 2) + pred_mv2[0]; w.pred_y = (pic_pix_y << 2) + pred_mv2[1];
    w.c1x = center1_x << MV_RESCALE; w.c1y = center1_y << MV_RESCALE;
    w.fixed_cost = jmo_mv_cost(lambda_factor, w.c1x, w.c1y, (pic_pix_x << 2) + pred_mv1[0], (pic_pix_y << 2) + pred_mv1[1]);
    stop = e->medthres[blocktype];
    e->blk_count = (short)(e->blk_count + 1);
    e->ord++;
    b->umv2 = !((center2_x > search_range) && (center2_x < (W - bsx) - search_range) && (center2_y > search_range) && (center2_y < (H - bsy) - search_range));   /* :2083 */
    b->umv1 = !((center1_x > search_range) && (center1_x < (W - bsx) - search_range) && (center1_y > search_range) && (center1_y < (H - bsy) - search_range));   /* :2094 */
    MAP(e, search_range, search_range) = e->blk_count;
    MAP_ORD(e, search_range, search_range) = e->ord;
    if (!FIRST_ORD(e, search_range, search_range)) FIRST_ORD(e, search_range, search_range) = e->ord;
    min_mcost = cost_mv(&w, center2_x << MV_RESCALE, center2_y << MV_RESCALE);
    min_mcost += cost_dist(&w, JMO_INT_MAX, center2_x << MV_RESCALE, center2_y << MV_RESCALE);
    if (min_mcost > stop) {
      int c_avail = nb->available[2], pat0 = e->search_pattern;
      (void)fix_block_c(mb_x, mb_y, bsx, 1, &c_avail);
      stop = (11 * e->medthres[blocktype]) >> 3;
      (void)spatial_predictors(e, nb, c_avail, list, ref, pred);
      check_median = scan_predictors(&w, pred, 5, pic_pix_x, pic_pix_y, mv, search_range, tempmv, tempmv2, &min_mcost, &second_mcost);
      if (min_mcost > stop) {
        if (e->cfg.pattern != 0) {
          if (min_mcost < stop + ((3 * e->medthres[blocktype]) >> 1)) {
            if ((tempmv[0] == 0 && tempmv[1] == 0) || (iabs_(tempmv[0] - mv[0]) < (2 << (2 - MV_RESCALE)) && iabs_(tempmv[1] - mv[1]) < (2 << (2 - MV_RESCALE)))) pat0 = P_SDIAMOND;
            else pat0 = P_SQUARE;
          } else if (blocktype > 5 || (ref > 0 && blocktype != 1)) pat0 = P_SQUARE;
          else pat0 = e->search_pattern;
        }
        (void)refine(&w, pat0, pic_pix_x, pic_pix_y, mv, search_range, blocktype, ref, tempmv, tempmv2, &min_mcost, stop, check_median, 0, NULL);
      }
    }
    mv[0] = (short)tempmv[0]; mv[1] = (short)tempmv[1];
    return min_mcost;
  }
}

/* ------------------------------------------------------------------ sub-pel */

/* the start/end positions of the directional second stage, :2499-2550 (half) / :2634-2688 (quarter: case 0 is commented out there) */
static void second_stage(int best_pos, int second_pos, int half, int *start_pos, int *end_pos)
{
  if (best_pos != 0 && second_pos != 0) {
    switch (best_pos ^ second_pos) {
    case 1: *start_pos = 6; *end_pos = 7; break;
    case 3: *start_pos = 5; *end_pos = 6; break;
    case 5: *start_pos = 8; *end_pos = 9; break;
    case 7: *start_pos = 7; *end_pos = 8; break;
    default: break;
    }
  } else {
    switch (best_pos + second_pos) {
    case 0: if (half) { *start_pos = 5; *end_pos = 5; } break;
    case 1: *start_pos = 8; *end_pos = 10; break;
    case 2: *start_pos = 5; *end_pos = 7; break;
    case 5: *start_pos = 6; *end_pos = 8; break;
    case 7: *start_pos = 7; *end_pos = 9; break;
    default: break;
    }
  }
}

typedef struct {
  const jmo_me_params *p; jmo_dist d; jmo_bipred *b; const jmo_pel *cur; int bsx, bsy, level, metric;
  int smv_x, smv_y, fixed_cost;
} sub_ctx;

static int sub_dist(sub_ctx *s, int bound, int x, int y)
{
  if (s->b) return jmo_bipred_dist(s->b, s->b->metric[s->level], s->cur, s->bsy, s->bsx, bound, s->smv_x, s->smv_y, x, y);
  return jmo_uni_pred(s->p, s->level, &s->d, s->cur, s->bsy, s->bsx, bound, x, y);
}

/* one refinement level of EPZSSubPelBlockMotionSearch / ...BiPred. Returns 1 on the sub-threshold early return. */
static int sub_level(sub_ctx *s, const short (*pts)[2], int half, int start, int max_pos, int lambda_factor, const short pred_mv[2], short mv[2],
                     int pic4_x, int pic4_y, int *min_mcost, int subthres, int early_return)
{
  int pos, best_pos = 0, second_pos = 0, second_mcost = JMO_INT_MAX, mcost, cx, cy, start_pos = 5, end_pos = max_pos;
  for (pos = start; pos < 5; pos++) {
    cx = mv[0] + pts[pos][0]; cy = mv[1] + pts[pos][1];
    mcost = jmo_mv_cost(lambda_factor, cx, cy, pred_mv[0], pred_mv[1]) + s->fixed_cost;
    mcost += sub_dist(s, JMO_INT_MAX, cx + pic4_x, cy + pic4_y);
    if (mcost < *min_mcost) { second_mcost = *min_mcost; second_pos = best_pos; *min_mcost = mcost; best_pos = pos; }
    else if (mcost < second_mcost) { second_mcost = mcost; second_pos = pos; }
  }
  if (early_return && best_pos == 0 && pred_mv[0] == mv[0] && (pred_mv[1] - mv[1]) == 0 && *min_mcost < subthres) return 1;
  second_stage(best_pos, second_pos, half, &start_pos, &end_pos);
  if (best_pos != 0 || (iabs_(pred_mv[0] - mv[0]) + iabs_(pred_mv[1] - mv[1])))
    for (pos = start_pos; pos < end_pos; pos++) {
      cx = mv[0] + pts[pos][0]; cy = mv[1] + pts[pos][1];
      mcost = jmo_mv_cost(lambda_factor, cx, cy, pred_mv[0], pred_mv[1]) + s->fixed_cost;
      if (mcost >= *min_mcost) continue;
      mcost += sub_dist(s, *min_mcost - mcost, cx + pic4_x, cy + pic4_y);
      if (mcost < *min_mcost) { *min_mcost = mcost; best_pos = pos; }
    }
  if (best_pos) { mv[0] += pts[best_pos][0]; mv[1] += pts[best_pos][1]; }
  return 0;
}

/* EPZSSubPelBlockMotionSearch :2390 */
int jmo_epzs_subpel_search(const jmo_epzs *e, const jmo_me_params *p, const jmo_ref *ref_pic, const jmo_pel *orig_pic,
                           int pic_pix_x, int pic_pix_y, int blocktype, const short pred_mv[2], short mv[2],
                           int search_pos2, int search_pos4, int min_mcost, const int *lambda)
{
  const int start_hp = (p->chroma_me == 1 || p->metric[JMO_F_PEL] != p->metric[JMO_H_PEL]) ? 0 : 1;
  const int start_qp = (p->chroma_me == 1 || p->metric[JMO_H_PEL] != p->metric[JMO_Q_PEL]) ? 0 : 1;
  const int pic4_x = (pic_pix_x + JMO_PAD) << 2, pic4_y = (pic_pix_y + JMO_PAD) << 2;
  const int max_pos2 = (!start_hp || !start_qp) ? imax_(1, search_pos2) : search_pos2;
  int bsx, bsy, max_x4, max_y4;
  sub_ctx s;
  jmo_block_size(blocktype, &bsx, &bsy);
  max_x4 = (ref_pic->W - bsx + 2 * JMO_PAD) << 2; max_y4 = (ref_pic->H - bsy + 2 * JMO_PAD) << 2;
  memset(&s, 0, sizeof(s));
  s.p = p; s.cur = orig_pic; s.bsx = bsx; s.bsy = bsy;
  jmo_dist_from_params(p, ref_pic, &s.d);
  s.d.chroma_me = (p->chroma_me == 2) ? 1 : 0;           /* mv-search.c:779 */
  s.d.test8x8 = p->transform8x8_mode && blocktype <= 4;
  s.level = JMO_H_PEL;
  s.d.umv = !((pic4_x + mv[0] > 1) && (pic4_x + mv[0] < max_x4 - 1) && (pic4_y + mv[1] > 1) && (pic4_y + mv[1] < max_y4 - 1));
  if (sub_level(&s, search_point_hp, 1, start_hp, max_pos2, lambda[JMO_H_PEL], pred_mv, mv, pic4_x, pic4_y, &min_mcost, e->subthres[blocktype], 1)) return min_mcost;
  if (!start_qp) min_mcost = JMO_INT_MAX;
  s.level = JMO_Q_PEL;
  s.d.umv = !((pic4_x + mv[0] > 0) && (pic4_x + mv[0] < max_x4) && (pic4_y + mv[1] > 0) && (pic4_y + mv[1] < max_y4));
  (void)sub_level(&s, search_point_qp, 0, start_qp, search_pos4, lambda[JMO_Q_PEL], pred_mv, mv, pic4_x, pic4_y, &min_mcost, e->subthres[blocktype], 1);
  return min_mcost;
}

/* EPZSSubPelBlockSearchBiPred :2728 (mv: swept, s_mv: fixed; pred_mv1 belongs to mv, pred_mv2 to s_mv -- JM's argument naming) */
int jmo_epzs_subpel_bipred(jmo_bipred *b, const jmo_pel *orig_pic, int pic_pix_x, int pic_pix_y, int blocktype,
                           const short pred_mv1[2], const short pred_mv2[2], short mv[2], const short s_mv[2],
                           int search_pos2, int search_pos4, int min_mcost, const int *lambda)
{
  const int pic4_x = (pic_pix_x + JMO_PAD) << 2, pic4_y = (pic_pix_y + JMO_PAD) << 2;
  const int start_hp = (min_mcost == JMO_INT_MAX) ? 0 : b->start_hp;
  const int max_pos2 = (!b->start_hp || !b->start_qp) ? imax_(1, search_pos2) : search_pos2;
  int bsx, bsy, max_x4, max_y4;
  sub_ctx s;
  jmo_block_size(blocktype, &bsx, &bsy);
  max_x4 = (b->ref1->W - bsx + 2 * JMO_PAD) << 2; max_y4 = (b->ref1->H - bsy + 2 * JMO_PAD) << 2;
  memset(&s, 0, sizeof(s));
  s.b = b; s.cur = orig_pic; s.bsx = bsx; s.bsy = bsy;
  s.smv_x = s_mv[0] + pic4_x; s.smv_y = s_mv[1] + pic4_y;
  s.level = JMO_H_PEL;
  s.fixed_cost = jmo_mv_cost(lambda[JMO_H_PEL], s_mv[0], s_mv[1], pred_mv2[0], pred_mv2[1]);
  b->umv2 = !((pic4_x + mv[0] > 1) && (pic4_x + mv[0] < max_x4 - 1) && (pic4_y + mv[1] > 1) && (pic4_y + mv[1] < max_y4 - 1));
  b->umv1 = !((pic4_x + s_mv[0] > 1) && (pic4_x + s_mv[0] < max_x4 - 1) && (pic4_y + s_mv[1] > 1) && (pic4_y + s_mv[1] < max_y4 - 1));
  (void)sub_level(&s, search_point_hp, 1, start_hp, max_pos2, lambda[JMO_H_PEL], pred_mv1, mv, pic4_x, pic4_y, &min_mcost, 0, 0);
  b->umv2 = !((pic4_x + mv[0] > 0) && (pic4_x + mv[0] < max_x4) && (pic4_y + mv[1] > 0) && (pic4_y + mv[1] < max_y4));
  b->umv1 = !((pic4_x + s_mv[0] > 0) && (pic4_x + s_mv[0] < max_x4) && (pic4_y + s_mv[1] > 0) && (pic4_y + s_mv[1] < max_y4));
  if (!b->start_qp) min_mcost = JMO_INT_MAX;
  s.level = JMO_Q_PEL;
  s.fixed_cost = jmo_mv_cost(lambda[JMO_Q_PEL], s_mv[0], s_mv[1], pred_mv2[0], pred_mv2[1]);
  (void)sub_level(&s, search_point_qp, 0, b->start_qp, search_pos4, lambda[JMO_Q_PEL], pred_mv1, mv, pic4_x, pic4_y, &min_mcost, 0, 0);
  return min_mcost;
}

/* the alias bookkeeping (not JM's): searches so far, and how many map tests read "visited" from a stamp this search did not write */
unsigned jmo_epzs_search_count(const jmo_epzs *e) { return e->ord; }
long jmo_epzs_alias_events(const jmo_epzs *e) { return e->alias_events; }
/* per cell, the ordinal (jmo_epzs_search_count) of the first search that tested or stamped it since jmo_epzs_map_set / create; 0 = none */
void jmo_epzs_first_touch(const jmo_epzs *e, unsigned *out) { memcpy(out, e->first_ord, sizeof(unsigned) * (size_t)e->searcharray * e->searcharray); }
void jmo_epzs_ideal_map(jmo_epzs *e, int on) { e->ideal_map = on; }
/* EPZSMap / EPZSBlkCount as a running encoder holds them (map: [searcharray][searcharray] stamps, NULL = all zero). The shadow takes every
 * stamp as written in the previous 65536-search epoch, so a test it answers counts as an alias event. */
void jmo_epzs_map_set(jmo_epzs *e, const short *map, int blk_count)
{
  const size_t n = (size_t)e->searcharray * e->searcharray;
  size_t i;
  for (i = 0; i < n; i++) { e->map[i] = map ? map[i] : 0; e->map_ord[i] = (unsigned short)e->map[i]; e->first_ord[i] = 0; }
  e->blk_count = (short)blk_count;
  e->ord = 65536u + (unsigned short)blk_count;
}
