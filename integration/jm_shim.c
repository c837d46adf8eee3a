/*
 * jm_shim.c -- the reference-side binding of libjmhip.so: JM (lencod) with its hot path on an MI355X.
 *
 * Built against JM's own headers and the UNMODIFIED JM compiled as a shared object. This file
 * DEFINES JM's hot-path symbols, so every caller inside JM -- direct calls and the pointer tables pDCT_4x4 etc. --
 * lands here; each definition marshals JM's globals into one C-ABI call (include/jmhip.h) and writes the results
 * back where JM expects them. Nothing here computes: a configuration the device path does not cover is FORWARDED to
 * JM's own function (dlsym RTLD_NEXT) and counted; a failing jmhip_* call aborts loudly (JM's error convention is
 * error()/exit). JMHIP_SHIM_STATS=1 prints, per symbol, how many calls ran on the device and how many were forwarded.
 * JMHIP_SHIM (hex mask, default all): 0x01 sub-pel planes, 0x04 full-pel + sub-pel search, 0x08 fast full search,
 * 0x10 dct_4x4/dct_16x16, 0x20 dct_8x8, 0x40 dct_chroma, 0x100 distortion surfaces for EPZS / UMHexagonS integer walks,
 * 0x200 bi-predictive full-pel + sub-pel search,
 * 0x400 RD-off mode-decision costs (TransformDecision, GetSkipCostMB),
 * 0x800 in-loop deblocking filter (DeblockFrame),
 * 0x1000 SLICE-LEVEL binding: in low-complexity mode with intra off in P slices (RDOptimization 0, DisableIntraInInter 1, no B pictures;
 *        4x4 transform or Transform8x8Mode 1 / 2; every search mode) the whole motion search + inter decision of a P slice -- or of all
 *        fixed-size slices of a picture -- is ONE device call (jmhip_p_slice_search) issued when JM asks for the slice's first block;
 *        every BlockMotionSearch call of the slice is then answered from its result record -- after checking that JM's own
 *        motion-vector predictor equals the one the device used (anything else is a fatal error, never a silent difference).
 * 0x2000 (with 0x1000) SPECULATIVE slice binding for the configurations the exact form does not cover (RDOptimization 1 / 2, intra
 *        candidates in P pictures; search modes -1, 0, 2, whose result is a pure function of the predictor): the device's low-complexity
 *        decision is only a guess of JM's; a call is answered from the record when JM's predictor equals the recorded one and runs JM's
 *        own search otherwise (counted as forwarded), so the bitstream is JM's whatever the guess was worth.
 * 0x4000 (with 0x1000, exact form, 4:2:0, 4x4 transform) the FRAME STAGE at slice level: right after the slice search the device also predicts,
 *        transforms, quantises and reconstructs every macroblock of the slice from its own decision (jmhip_slice_to_frame_band ->
 *        jmhip_residual_frame), and JM's LumaPrediction / ChromaPrediction4x4 / dct_4x4 / dct_chroma calls of the slice are answered from the
 *        prediction picture and the per-macroblock records -- a prediction call after checking that JM asks for the block the device decided
 *        (mode, reference, vectors), a dct call after checking that JM's img->mpr / img->m7 are the prediction and residual the device used
 *        (anything else is a fatal error).
 * 0x8000 the sub-pel planes stay on the device until JM is about to read them: getSubImagesLuma / getSubImagesChroma still upload the finished picture
 *        and build the planes, but only the integer plane (which JM's weighted-prediction estimation reads) crosses back at once; the other planes
 *        of a picture are fetched into JM's rows the first time a call is FORWARDED to JM code that reads reference planes (JM's own
 *        BlockMotionSearch, LumaPrediction / LumaPredictionBi / ChromaPrediction / ChromaPrediction4x4) -- in pictures the slice binding serves
 *        completely, never.
 *
 * The proof of the drop-in claim is tests/test_jm_shim.py: the bitstream and the reconstruction this encoder
 * writes are byte-identical to the unmodified encoder's.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <time.h>

#include "global.h"
#include "mbuffer.h"
#include "image.h"
#include "refbuf.h"
#include "me_distortion.h"
#include "q_matrix.h"
#include "q_offsets.h"
#include "me_epzs.h"

#include "jmhip.h"

extern int jm_main(int argc, char **argv);
extern int ****ptLevelOffset4x4;
extern void SetMotionVectorPredictor(short pmv[2], char **refPic, short ***tmp_mv, short ref_frame, int list,
                                     int block_x, int block_y, int blockshape_x, int blockshape_y);
extern const int LEVELMVLIMIT[17][6];
extern int *mvbits;                       /* src/mv-search.c:59 */

enum { S_LUMA, S_CHROMA, S_FULL, S_SUB, S_FAST, S_D4, S_D8, S_D16, S_DCR, S_WALK, S_SAD, S_SATD, S_BIFULL, S_BISUB, S_TDEC, S_SKIPC, S_BIDC, S_DEBLOCK, S_SLICE, S_BMS,
       S_FRAME, S_D4R, S_DCRR, S_LPRED, S_CPRED, S_LAZY, S_COUNT };
static const char *s_names[S_COUNT] = { "getSubImagesLuma", "getSubImagesChroma", "FullPelBlockMotionSearch",
  "SubPelBlockMotionSearch", "FastFullPelBlockMotionSearch", "dct_4x4", "dct_8x8", "dct_16x16", "dct_chroma",
  "EPZS_UMHex_integer_walks", "computeSAD", "computeSATD", "FullPelBlockMotionBiPred", "SubPelBlockSearchBiPred",
  "TransformDecision", "GetSkipCostMB", "BIDPartitionCost", "DeblockFrame",
  "P slices (one device call each)", "BlockMotionSearch",
  "frame stage of P slices", "dct_4x4 (slice records)", "dct_chroma (slice records)", "LumaPrediction (slice)", "ChromaPrediction4x4 (slice)", "sub-pel planes fetched on demand" };
static long n_dev[S_COUNT], n_fwd[S_COUNT];
static double t_dev[S_COUNT], t_last[S_COUNT];             /* JMHIP_SHIM_STATS: wall seconds inside the coarse device-side hooks (planes, slice search, loop filter) */
static int stats_on;
static double now_s(void) { struct timespec ts; if (!stats_on) return 0.0; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
static unsigned shim_mask = 0xdfff;
static int verify;            /* JMHIP_SHIM_VERIFY=1: run JM's own search beside the device's and report differences */
static jmhip_ctx *g;
static int g_w, g_h;

static void *next_sym(const char *name)
{
  void *p = dlsym(RTLD_NEXT, name);
  if (!p) { fprintf(stderr, "jm_shim: cannot find JM's %s\n", name); exit(97); }
  return p;
}

static void die(const char *what, int rc)
{
  fprintf(stderr, "jm_shim: %s failed: %s (%s)\n", what, jmhip_strerror(rc), g ? jmhip_last_error(g) : "no context");
  exit(96);
}
#define OK(call) do { int rc_ = (call); if (rc_) die(#call, rc_); } while (0)

static long sl_slices(void), sl_passes(void);
static void print_stats(void)
{
  int i;
  if (!getenv("JMHIP_SHIM_STATS")) return;
  fprintf(stderr, "jm_shim: mask=0x%x\n", shim_mask);
  for (i = 0; i < S_COUNT; i++) {
    fprintf(stderr, "  %-30s device %8ld  forwarded %8ld", s_names[i], n_dev[i], n_fwd[i]);
    if (t_dev[i] > 0.0) fprintf(stderr, "  %8.1f ms inside the hook (last call %.1f ms)", t_dev[i] * 1e3, t_last[i] * 1e3);
    fprintf(stderr, "\n");
  }
  if (sl_slices()) fprintf(stderr, "  slice binding: %ld slices, %ld kernel passes\n", sl_slices(), sl_passes());
}

int main(int argc, char **argv)
{
  const char *m = getenv("JMHIP_SHIM");
  if (m) shim_mask = (unsigned)strtoul(m, NULL, 16);
  verify = getenv("JMHIP_SHIM_VERIFY") != NULL;
  stats_on = getenv("JMHIP_SHIM_STATS") != NULL;
  atexit(print_stats);
  return jm_main(argc, argv);
}

/* ------------------------------------------------------------------ context, reference slots, current picture */

#define MAX_SLOTS 16
static struct { StorablePicture *pic; unsigned long stamp; int has_chroma; int host_luma, host_chroma; } slots[MAX_SLOTS];   /* host_*: JM's rows hold the sub-pel planes */
static unsigned long slot_clock;

/* The context is created on first use for the sequence's FRAME size; 8-bit only. Field pictures and other sizes
 * stay in JM. */
static int ctx_ready(void)
{
  if (img->bitdepth_luma != 8 || (img->yuv_format != YUV400 && img->bitdepth_chroma != 8)) return 0;
  if (!g) {
    jmhip_config cfg;
    int rc;
    memset(&cfg, 0, sizeof(cfg));
    cfg.device = getenv("JMHIP_DEVICE") ? atoi(getenv("JMHIP_DEVICE")) : 0;
    cfg.width = img->width; cfg.height = img->height;
    cfg.yuv_format = img->yuv_format; cfg.bit_depth = 8;
    cfg.max_refs = MAX_SLOTS; cfg.search_range = input->search_range;
    if (img->structure != FRAME) return 0;              /* img->height is the field height while a field is coded */
    rc = jmhip_ctx_create(&cfg, &g);
    if (rc) { fprintf(stderr, "jm_shim: jmhip_ctx_create: %s\n", jmhip_strerror(rc)); exit(96); }
    g_w = cfg.width; g_h = cfg.height;
  }
  return 1;
}

static int frame_ok(StorablePicture *s)
{
  return ctx_ready() && s->size_x == g_w && s->size_y == g_h;
}

static int slot_find(StorablePicture *s)
{
  int i;
  for (i = 0; i < MAX_SLOTS; i++) if (slots[i].pic == s) return i;
  return -1;
}

static int slot_assign(StorablePicture *s)
{
  int i = slot_find(s), k;
  if (i < 0) for (i = 0, k = 1; k < MAX_SLOTS; k++) if (slots[k].stamp < slots[i].stamp) i = k;   /* least recently built */
  slots[i].pic = s; slots[i].stamp = ++slot_clock; slots[i].has_chroma = 0; slots[i].host_luma = slots[i].host_chroma = 0;
  return i;
}

/* mask 0x8000, not for 4:4:4 (its chroma planes are read through the luma paths of the per-plane code as well) */
static int lazy_planes(void) { return (shim_mask & 0x8000) && img->yuv_format != YUV444; }

static unsigned long pic_serial;          /* bumped whenever a new source picture goes to the device */

/* the source picture: uploaded once per coded picture */
static int cur_ready(void)
{
  static struct { void *enc; int number, bfr, type; imgpel *y; } key;
  if (!g || img->structure != FRAME || img->MbaffFrameFlag || img->width != g_w || img->height != g_h) return 0;
  if (key.enc != (void *)enc_picture || key.number != img->number || key.bfr != img->b_frame_to_code || key.type != img->type ||
      key.y != pCurImg[0]) {
    OK(jmhip_cur_upload(g, pCurImg[0], img->yuv_format != YUV400 ? imgUV_org[0][0] : NULL,
                        img->yuv_format != YUV400 ? imgUV_org[1][0] : NULL, (int)sizeof(imgpel), img->width, img->width_cr, 0));
    key.enc = enc_picture; key.number = img->number; key.bfr = img->b_frame_to_code; key.type = img->type; key.y = pCurImg[0];
    pic_serial++;
  }
  return 1;
}

/* JM's search functions leave these globals set (me_fullsearch.c:80-107, :378-403; me_fullfast.c:517-547) and later
 * code depends on it: OneComponentLumaPrediction reads width_pad/height_pad through UMVLine4X BEFORE assigning them
 * (macroblock.c:816-819). */
static void jm_side_effects(StorablePicture *rp)
{
  ref_pic_sub.luma = rp->p_curr_img_sub;
  width_pad = rp->size_x_pad; height_pad = rp->size_y_pad;
}

/* ------------------------------------------------------------------ sub-pel planes */

void getSubImagesLuma(StorablePicture *s)
{
  static void (*orig)(StorablePicture *);
  static void **rows; static size_t rows_n;
  if (!(shim_mask & 0x01) || !frame_ok(s) || s->p_curr_img != s->imgY) {
    /* (4:4:4 independent planes interpolate U/V through this function too: left to JM) */
    int i = slot_find(s);
    if (i >= 0) slots[i].pic = NULL;
    if (!orig) orig = next_sym("getSubImagesLuma");
    n_fwd[S_LUMA]++; orig(s); return;
  }
  {
    const int Hp = s->size_y_padded;
    const size_t need = (size_t)16 * Hp;
    int slot = slot_assign(s), p, j;
    const double t0 = now_s();
    if (rows_n < need) { free(rows); rows = malloc(need * sizeof(*rows)); rows_n = need; }
    /* UnifiedOneForthPix calls this on the finished picture, chroma included (image.c:1642-1659) */
    OK(jmhip_ref_upload(g, slot, s->imgY[0], img->yuv_format != YUV400 ? s->imgUV[0][0] : NULL,
                        img->yuv_format != YUV400 ? s->imgUV[1][0] : NULL, (int)sizeof(imgpel), s->size_x, s->size_x_cr, 0));
    OK(jmhip_interp_luma(g, slot));
    /* JM's own MC reads imgY_sub on the host: every plane straight into JM's rows -- or (mask 0x8000) only the integer plane now (the weight
       estimation of src/weighted_prediction.c reads p_curr_img_sub[0][0]) and the rest when a forwarded call is about to read them (host_planes) */
    for (p = 0; p < 16; p++) for (j = 0; j < Hp; j++) rows[(size_t)p * Hp + j] = (p == 0 || !lazy_planes()) ? s->p_curr_img_sub[p >> 2][p & 3][j] : NULL;
    OK(jmhip_ref_download_luma_rows(g, slot, rows, (int)sizeof(imgpel)));
    slots[slot].host_luma = !lazy_planes();
    n_dev[S_LUMA]++; t_last[S_LUMA] = now_s() - t0; t_dev[S_LUMA] += t_last[S_LUMA];
  }
}

void getSubImagesChroma(StorablePicture *s)
{
  static void (*orig)(StorablePicture *);
  static void **rows; static size_t rows_n;
  int slot = g ? slot_find(s) : -1;
  if (!(shim_mask & 0x01) || slot < 0 || img->yuv_format == YUV400) {
    if (!orig) orig = next_sym("getSubImagesChroma");
    n_fwd[S_CHROMA]++; orig(s); return;
  }
  {
    const int sub_x = img->yuv_format == YUV444 ? 4 : 8, sub_y = img->yuv_format == YUV420 ? 8 : 4;
    const int Hcp = s->size_y_cr + 2 * img_pad_size_uv_y;
    const size_t need = (size_t)sub_x * sub_y * Hcp;
    int uv, p, j;
    const double t0 = now_s();
    if (rows_n < need) { free(rows); rows = malloc(need * sizeof(*rows)); rows_n = need; }
    OK(jmhip_interp_chroma(g, slot));
    slots[slot].has_chroma = 1; slots[slot].host_chroma = 0;
    if (!lazy_planes()) {
      for (uv = 0; uv < 2; uv++) {
        for (p = 0; p < sub_x * sub_y; p++) for (j = 0; j < Hcp; j++) rows[(size_t)p * Hcp + j] = s->imgUV_sub[uv][p / sub_x][p % sub_x][j];
        OK(jmhip_ref_download_chroma_rows(g, slot, uv, rows, (int)sizeof(imgpel)));
      }
      slots[slot].host_chroma = 1;
    }
    n_dev[S_CHROMA]++; t_last[S_CHROMA] = now_s() - t0; t_dev[S_CHROMA] += t_last[S_CHROMA];
  }
}

/* mask 0x8000: JM code that reads reference planes is about to run -- make JM's rows of every reference picture in the lists hold them */
static void host_planes(StorablePicture *s)
{
  static void **rows; static size_t rows_n;
  const int i = g ? slot_find(s) : -1;
  int p, j, uv;
  if (i < 0 || (slots[i].host_luma && (slots[i].host_chroma || !slots[i].has_chroma))) return;
  {
    const int Hp = s->size_y_padded, sub_x = img->yuv_format == YUV444 ? 4 : 8, sub_y = img->yuv_format == YUV420 ? 8 : 4;
    const int Hcp = s->size_y_cr + 2 * img_pad_size_uv_y;
    size_t need = (size_t)16 * Hp;
    const double t0 = now_s();
    if ((size_t)sub_x * sub_y * Hcp > need) need = (size_t)sub_x * sub_y * Hcp;
    if (rows_n < need) { free(rows); rows = malloc(need * sizeof(*rows)); rows_n = need; }
    if (!slots[i].host_luma) {
      for (p = 0; p < 16; p++) for (j = 0; j < Hp; j++) rows[(size_t)p * Hp + j] = p ? s->p_curr_img_sub[p >> 2][p & 3][j] : NULL;     /* plane 0 went at once */
      OK(jmhip_ref_download_luma_rows(g, i, rows, (int)sizeof(imgpel)));
      slots[i].host_luma = 1;
    }
    if (slots[i].has_chroma && !slots[i].host_chroma) {
      for (uv = 0; uv < 2; uv++) {
        for (p = 0; p < sub_x * sub_y; p++) for (j = 0; j < Hcp; j++) rows[(size_t)p * Hcp + j] = s->imgUV_sub[uv][p / sub_x][p % sub_x][j];
        OK(jmhip_ref_download_chroma_rows(g, i, uv, rows, (int)sizeof(imgpel)));
      }
      slots[i].host_chroma = 1;
    }
    n_dev[S_LAZY]++; t_last[S_LAZY] = now_s() - t0; t_dev[S_LAZY] += t_last[S_LAZY];
  }
}
static void host_planes_lists(void)
{
  int l, j;
  if (!lazy_planes() || !g) return;
  for (l = 0; l < 6; l++) for (j = 0; j < listXsize[l]; j++) if (listX[l][j]) host_planes(listX[l][j]);
}

/* LumaPredictionBi (src/macroblock.c:948) and ChromaPrediction (:1712) read reference planes and are never answered from the device: forwarded, planes first */
void LumaPredictionBi(Macroblock *currMB, int block_x, int block_y, int block_size_x, int block_size_y, int l0_mode, int l1_mode, short l0_ref_idx, short l1_ref_idx, int list)
{
  static void (*orig)(Macroblock *, int, int, int, int, int, int, short, short, int);
  if (!orig) orig = next_sym("LumaPredictionBi");
  host_planes_lists();
  orig(currMB, block_x, block_y, block_size_x, block_size_y, l0_mode, l1_mode, l0_ref_idx, l1_ref_idx, list);
}
void ChromaPrediction(Macroblock *currMB, int uv, int block_x, int block_y, int block_size_x, int block_size_y, int p_dir, int l0_mode, int l1_mode, short l0_ref_idx, short l1_ref_idx)
{
  static void (*orig)(Macroblock *, int, int, int, int, int, int, int, int, short, short);
  if (!orig) orig = next_sym("ChromaPrediction");
  host_planes_lists();
  orig(currMB, uv, block_x, block_y, block_size_x, block_size_y, p_dir, l0_mode, l1_mode, l0_ref_idx, l1_ref_idx);
}

/* ------------------------------------------------------------------ motion search */

static int partition_of(int blocktype, int x, int y)     /* (x, y): block origin inside the macroblock, pel */
{
  static int tab[8][16], built;
  if (!built) {
    int p, bt, x4, y4, w4, h4;
    memset(tab, -1, sizeof(tab));
    for (p = 0; p < JMHIP_NPART; p++) { jmhip_partition_info(p, &bt, &x4, &y4, &w4, &h4); tab[bt][y4 * 4 + x4] = p; }
    built = 1;
  }
  if (blocktype < 1 || blocktype > 7 || (x & 3) || (y & 3) || x < 0 || y < 0 || x > 12 || y > 12) return -1;
  return tab[blocktype][(y >> 2) * 4 + (x >> 2)];
}

/* what the device search covers: luma-only SAD (integer) / SATD (sub-pel), plain or weighted reference, frame pictures */
/* weighted reference ME of the search just admitted by me_ok(): wp_weight / wp_offset [list][ref][0] (src/me_fullsearch.c:76-91) */
static struct { int on, weight, offset, weight_cr[2], offset_cr[2]; } me_wp;

/* fixed_metrics: 0 = the caller serves luma distortions of whatever metric (the walkers' surfaces); 1 = every MEErrorMetric[] and
 * ChromaMEEnable setting the device search has (FullPel / SubPel / FastFull through jmhip_me_frame: JM's defaults on the fast kernels,
 * everything else through metric_set) */
static int me_ok_metric(short ref, int list, StorablePicture **rp, int *slot, int fixed_metrics)
{
  int list_offset = img->mb_data[img->current_mb_nr].list_offset, m;
  int weighted = ((active_pps->weighted_pred_flag && (img->type == P_SLICE || img->type == SP_SLICE)) ||
                  (active_pps->weighted_bipred_idc && (img->type == B_SLICE))) && input->UseWeightedReferenceME;
  if (list_offset) return 0;
  if (input->ChromaMEEnable && !fixed_metrics) return 0;
  me_wp.on = weighted;
  if (weighted) {
    me_wp.weight = wp_weight[list + list_offset][ref][0]; me_wp.offset = wp_offset[list + list_offset][ref][0];
    for (m = 0; m < 2; m++) { me_wp.weight_cr[m] = wp_weight[list + list_offset][ref][1 + m]; me_wp.offset_cr[m] = wp_offset[list + list_offset][ref][1 + m]; }
  }
  if (!cur_ready()) return 0;
  *rp = listX[list][ref];
  *slot = slot_find(*rp);
  if (*slot >= 0 && input->ChromaMEEnable && !slots[*slot].has_chroma) return 0;
  return *slot >= 0;
}

static int me_ok(short ref, int list, StorablePicture **rp, int *slot) { return me_ok_metric(ref, list, rp, slot, 1); }

static void me_params(jmhip_me_params *prm, int mode, int range, int lam_f, int lam_h, int lam_q, int p)
{
  int k;
  memset(prm, 0, sizeof(*prm));
  if (me_wp.on) {                                   /* one search = one reference: the same weights for every slot is exact */
    prm->wp_enable = 1; prm->wp_round = wp_luma_round; prm->wp_denom = luma_log_weight_denom;
    for (k = 0; k < 16; k++) { prm->wp_weight[k] = (int16_t)me_wp.weight; prm->wp_offset[k] = (int16_t)me_wp.offset; }
  }
  prm->search_mode = mode; prm->search_range = range; prm->rdopt = input->rdopt; prm->is_b_slice = (img->type == B_SLICE);
  prm->level_mv_min = LEVELMVLIMIT[img->LevelIndex][0]; prm->level_mv_max = LEVELMVLIMIT[img->LevelIndex][1];
  prm->lambda[0] = lam_f; prm->lambda[1] = lam_h; prm->lambda[2] = lam_q;
  prm->transform8x8_mode = input->Transform8x8Mode; prm->subpel = 0;
  prm->partition_mask = 1ull << p;
  /* input->MEErrorMetric[] / ChromaMEEnable: anything but JM's defaults goes to the general search (me_metric.hip) */
  prm->metric_set = 1;
  for (k = 0; k < 3; k++) prm->metric[k] = input->MEErrorMetric[k];
  prm->chroma_me = input->ChromaMEEnable; prm->chroma_me_weight = input->ChromaMEWeight;
  if (me_wp.on && input->ChromaMEEnable) {
    prm->wp_chroma_round = wp_chroma_round; prm->wp_chroma_denom = chroma_log_weight_denom;
    for (k = 0; k < 16; k++) {
      prm->wp_weight_cr[k][0] = (int16_t)me_wp.weight_cr[0]; prm->wp_weight_cr[k][1] = (int16_t)me_wp.weight_cr[1];
      prm->wp_offset_cr[k][0] = (int16_t)me_wp.offset_cr[0]; prm->wp_offset_cr[k][1] = (int16_t)me_wp.offset_cr[1];
    }
  }
}

/* the reference-picture globals JM's search functions leave behind when the chroma term is on (me_fullsearch.c:93-108) */
static void jm_side_effects_chroma(StorablePicture *rp)
{
  if (!input->ChromaMEEnable) return;
  ref_pic_sub.crcb[0] = rp->imgUV_sub[0]; ref_pic_sub.crcb[1] = rp->imgUV_sub[1];
  width_pad_cr = rp->size_x_cr_pad; height_pad_cr = rp->size_y_cr_pad;
}

int FullPelBlockMotionSearch(imgpel *orig_pic, short ref, int list, int pic_pix_x, int pic_pix_y, int blocktype,
                             short pred_mv_x, short pred_mv_y, short *mv_x, short *mv_y, int search_range,
                             int min_mcost, int lambda_factor)
{
  static int (*orig)(imgpel *, short, int, int, int, int, short, short, short *, short *, int, int, int);
  StorablePicture *rp; int slot, p = -1, ok;
  ok = (shim_mask & 0x04) && min_mcost == INT_MAX && search_range <= input->search_range && me_ok(ref, list, &rp, &slot) &&
       (p = partition_of(blocktype, pic_pix_x - img->opix_x, pic_pix_y - img->opix_y)) >= 0;
  if (ok) {
    /* the device derives the search centre from the predictor like BlockMotionSearch does (mv-search.c:752-762);
       any other caller-supplied centre is JM's business */
    int cx = pred_mv_x / 4, cy = pred_mv_y / 4;
    if (!input->rdopt) { cx = iClip3(-search_range, search_range, cx); cy = iClip3(-search_range, search_range, cy); }
    cx = iClip3(-2047 + search_range, 2047 - search_range, cx);
    cy = iClip3(LEVELMVLIMIT[img->LevelIndex][0] + search_range, LEVELMVLIMIT[img->LevelIndex][1] - search_range, cy);
    ok = (cx == *mv_x && cy == *mv_y);
  }
  if (!ok) {
    if (!orig) orig = next_sym("FullPelBlockMotionSearch");
    n_fwd[S_FULL]++;
    return orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, mv_x, mv_y, search_range, min_mcost, lambda_factor);
  }
  {
    jmhip_me_params prm; jmhip_me_mb mb; jmhip_me_result r;
    me_params(&prm, JMHIP_SEARCH_FULL, search_range, lambda_factor, 0, 0, p);
    memset(&mb, 0, sizeof(mb));
    mb.mb_x = img->opix_x >> 4; mb.mb_y = img->opix_y >> 4; mb.ref = slot; mb.ref_is_0 = (ref == 0);
    mb.pred_mv[p][0] = pred_mv_x; mb.pred_mv[p][1] = pred_mv_y;
    jm_side_effects(rp); jm_side_effects_chroma(rp);
    OK(jmhip_me_frame(g, &prm, &mb, 1, &r));
    if (verify) {
      short jx = *mv_x, jy = *mv_y; int c;
      if (!orig) orig = next_sym("FullPelBlockMotionSearch");
      c = orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, &jx, &jy, search_range, min_mcost, lambda_factor);
      if (c != r.cost_int[p] || jx != r.mv_int[p][0] || jy != r.mv_int[p][1])
        fprintf(stderr, "jm_shim VERIFY FullPel: type=%d ref=%d list=%d pix=(%d,%d) bt=%d pred=(%d,%d) ctr=(%d,%d) R=%d lam=%d: jm=(%d,%d,%d) dev=(%d,%d,%d)\n",
                img->type, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, *mv_x, *mv_y, search_range, lambda_factor,
                jx, jy, c, r.mv_int[p][0], r.mv_int[p][1], r.cost_int[p]);
    }
    *mv_x = r.mv_int[p][0]; *mv_y = r.mv_int[p][1];
    n_dev[S_FULL]++;
    return r.cost_int[p];
  }
}

int SubPelBlockMotionSearch(imgpel *orig_pic, short ref, int list, int pic_pix_x, int pic_pix_y, int blocktype,
                            short pred_mv_x, short pred_mv_y, short *mv_x, short *mv_y, int search_pos2,
                            int search_pos4, int min_mcost, int *lambda)
{
  static int (*orig)(imgpel *, short, int, int, int, int, short, short, short *, short *, int, int, int, int *);
  StorablePicture *rp; int slot, p = -1, ok;
  /* mv-search.c:396, :785-788: with equal metrics at the integer and half-pel level the refinement starts from the integer minimum */
  const int start_hp = !(input->ChromaMEEnable == 1 || input->MEErrorMetric[F_PEL] != input->MEErrorMetric[H_PEL]);
  ok = (shim_mask & 0x04) && (min_mcost == INT_MAX || start_hp) && search_pos2 == 9 && search_pos4 == 9 && !((*mv_x | *mv_y) & 3) &&
       me_ok(ref, list, &rp, &slot) && (p = partition_of(blocktype, pic_pix_x - img->opix_x, pic_pix_y - img->opix_y)) >= 0 &&
       test8x8transform == (input->Transform8x8Mode && blocktype <= 4);
  if (!ok) {
    if (!orig) orig = next_sym("SubPelBlockMotionSearch");
    n_fwd[S_SUB]++;
    return orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, mv_x, mv_y, search_pos2, search_pos4, min_mcost, lambda);
  }
  {
    jmhip_me_params prm; jmhip_me_mb mb; jmhip_me_result r;
    me_params(&prm, JMHIP_SEARCH_FULL, input->search_range, lambda[F_PEL], lambda[H_PEL], lambda[Q_PEL], p);
    prm.subpel = 1;
    memset(&mb, 0, sizeof(mb)); memset(&r, 0, sizeof(r));
    mb.mb_x = img->opix_x >> 4; mb.mb_y = img->opix_y >> 4; mb.ref = slot; mb.ref_is_0 = (ref == 0);
    mb.pred_mv[p][0] = pred_mv_x; mb.pred_mv[p][1] = pred_mv_y;
    r.mv_int[p][0] = *mv_x >> 2; r.mv_int[p][1] = *mv_y >> 2; r.cost_int[p] = min_mcost;
    jm_side_effects(rp); jm_side_effects_chroma(rp);
    OK(jmhip_me_subpel(g, &prm, &mb, 1, &r));
    if (verify) {
      short jx = *mv_x, jy = *mv_y; int c;
      if (!orig) orig = next_sym("SubPelBlockMotionSearch");
      c = orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, &jx, &jy, search_pos2, search_pos4, min_mcost, lambda);
      if (c != r.cost[p] || jx != r.mv[p][0] || jy != r.mv[p][1])
        fprintf(stderr, "jm_shim VERIFY SubPel: type=%d ref=%d list=%d pix=(%d,%d) bt=%d pred=(%d,%d) in=(%d,%d) lam=%d,%d t8=%d: jm=(%d,%d,%d) dev=(%d,%d,%d)\n",
                img->type, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, *mv_x, *mv_y, lambda[1], lambda[2], test8x8transform,
                jx, jy, c, r.mv[p][0], r.mv[p][1], r.cost[p]);
    }
    *mv_x = r.mv[p][0]; *mv_y = r.mv[p][1];
    n_dev[S_SUB]++;
    return r.cost[p];
  }
}

/* SetupFastFullPelSearch runs once per (macroblock, list, ref) around the 16x16 predictor (me_fullfast.c:491-566);
 * the per-block calls reuse that centre. ResetFastFullIntegerSearch marks a new macroblock. */
static struct { int done; short pmv[2]; } ff_state[2][MAX_SLOTS];

void ResetFastFullIntegerSearch(void)
{
  static void (*orig)(void);
  if (!orig) orig = next_sym("ResetFastFullIntegerSearch");
  orig();
  memset(ff_state, 0, sizeof(ff_state));
}

int FastFullPelBlockMotionSearch(imgpel *orig_pic, short ref, int list, int pic_pix_x, int pic_pix_y, int blocktype,
                                 short pred_mv_x, short pred_mv_y, short *mv_x, short *mv_y, int search_range,
                                 int min_mcost, int lambda_factor)
{
  static int (*orig)(imgpel *, short, int, int, int, int, short, short, short *, short *, int, int, int);
  StorablePicture *rp; int slot, p = -1, ok;
  ok = (shim_mask & 0x08) && min_mcost == INT_MAX && ref < MAX_SLOTS && list < 2 && me_ok(ref, list, &rp, &slot) &&
       (p = partition_of(blocktype, pic_pix_x - img->opix_x, pic_pix_y - img->opix_y)) >= 0;
  if (ok && !ff_state[list][ref].done) {
    SetMotionVectorPredictor(ff_state[list][ref].pmv, enc_picture->ref_idx[list], enc_picture->mv[list], ref, list, 0, 0, 16, 16);
    ff_state[list][ref].done = 1;
  }
  /* the 16x16 block is searched with the predictor the window was centred on */
  if (ok && p == 0 && (pred_mv_x != ff_state[list][ref].pmv[0] || pred_mv_y != ff_state[list][ref].pmv[1])) ok = 0;
  if (!ok) {
    if (!orig) orig = next_sym("FastFullPelBlockMotionSearch");
    n_fwd[S_FAST]++;
    return orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, mv_x, mv_y, search_range, min_mcost, lambda_factor);
  }
  {
    /* range per reference as InitializeFastFullIntegerSearch, me_fullfast.c:127-140 */
    int R = (input->full_search == 2 || ref == 0) ? input->search_range : input->search_range / 2;
    jmhip_me_params prm; jmhip_me_mb mb; jmhip_me_result r;
    me_params(&prm, JMHIP_SEARCH_FASTFULL, R, lambda_factor, 0, 0, p);
    memset(&mb, 0, sizeof(mb));
    mb.mb_x = img->opix_x >> 4; mb.mb_y = img->opix_y >> 4; mb.ref = slot; mb.ref_is_0 = (ref == 0);
    mb.pred_mv[0][0] = ff_state[list][ref].pmv[0]; mb.pred_mv[0][1] = ff_state[list][ref].pmv[1];   /* the window centre */
    mb.pred_mv[p][0] = pred_mv_x; mb.pred_mv[p][1] = pred_mv_y;
    jm_side_effects(rp); jm_side_effects_chroma(rp);
    OK(jmhip_me_frame(g, &prm, &mb, 1, &r));
    if (verify) {
      short jx = *mv_x, jy = *mv_y; int c;
      if (!orig) orig = next_sym("FastFullPelBlockMotionSearch");
      c = orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, &jx, &jy, search_range, min_mcost, lambda_factor);
      if (c != r.cost_int[p] || jx != r.mv_int[p][0] || jy != r.mv_int[p][1])
        fprintf(stderr, "jm_shim VERIFY FastFull: type=%d ref=%d list=%d pix=(%d,%d) bt=%d pred=(%d,%d) lam=%d: jm=(%d,%d,%d) dev=(%d,%d,%d)\n",
                img->type, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, lambda_factor, jx, jy, c, r.mv_int[p][0], r.mv_int[p][1], r.cost_int[p]);
    }
    *mv_x = r.mv_int[p][0]; *mv_y = r.mv_int[p][1];
    n_dev[S_FAST]++;
    return r.cost_int[p];
  }
}

/* ------------------------------------------------------------------ bi-predictive search (B slices, 16x16) */

/* "1" = fixed block on listX[list][ref], "2" = swept candidate on listX[list^1][0] (src/me_fullsearch.c:206-207) */
static int bipred_ok(short ref, int list, int blocktype, int *slot1, int *slot2, jmhip_bipred_params *prm, int *lambda3)
{
  int list_offset = img->mb_data[img->current_mb_nr].list_offset;
  int apply_weights = (active_pps->weighted_bipred_idc > 0);
  StorablePicture *p1, *p2;
  if (!(shim_mask & 0x200) || blocktype != 1 || list_offset || ChromaMEEnable || input->ChromaMEEnable) return 0;
  if (input->MEErrorMetric[F_PEL] != ERROR_SAD || input->MEErrorMetric[H_PEL] != ERROR_SATD || input->MEErrorMetric[Q_PEL] != ERROR_SATD) return 0;
  if (start_me_refinement_hp != 0 || start_me_refinement_qp != 1 || !cur_ready()) return 0;
  p1 = listX[list][ref]; p2 = listX[list ^ 1][0];
  *slot1 = slot_find(p1); *slot2 = slot_find(p2);
  if (*slot1 < 0 || *slot2 < 0) return 0;
  memset(prm, 0, sizeof(*prm));
  prm->lambda[0] = lambda3[0]; prm->lambda[1] = lambda3[1]; prm->lambda[2] = lambda3[2];
  prm->transform8x8_mode = test8x8transform;
  prm->apply_weights = apply_weights;
  if (apply_weights) {
    short offset1 = list == 0 ? wp_offset[0][ref][0] : wp_offset[1][0][ref];
    short offset2 = list == 0 ? wp_offset[1][ref][0] : wp_offset[0][0][ref];
    prm->weight1 = list == 0 ? wbp_weight[0][ref][0][0] : wbp_weight[LIST_1][0][ref][0];
    prm->weight2 = list == 0 ? wbp_weight[LIST_1][ref][0][0] : wbp_weight[0][0][ref][0];
    prm->offset_bi = (offset1 + offset2 + 1) >> 1;
  }
  prm->wp_luma_round = wp_luma_round; prm->luma_log_weight_denom = luma_log_weight_denom;
  /* the globals JM's own function would leave behind */
  ref_pic1_sub.luma = p1->p_curr_img_sub; ref_pic2_sub.luma = p2->p_curr_img_sub;
  width_pad = p1->size_x_pad; height_pad = p1->size_y_pad;
  return 1;
}

int FullPelBlockMotionBiPred(imgpel *orig_pic, short ref, int list, int pic_pix_x, int pic_pix_y, int blocktype,
                             short pred_mv_x1, short pred_mv_y1, short pred_mv_x2, short pred_mv_y2,
                             short *mv_x, short *mv_y, short *s_mv_x, short *s_mv_y, int search_range, int min_mcost, int lambda_factor)
{
  static int (*orig)(imgpel *, short, int, int, int, int, short, short, short, short, short *, short *, short *, short *, int, int, int);
  jmhip_bipred_params prm; jmhip_bipred_job job; jmhip_bipred_result r;
  int s1, s2, lam3[3];
  lam3[0] = lambda_factor; lam3[1] = lam3[2] = 0;
  if (search_range > 44 || pic_pix_x != img->opix_x || pic_pix_y != img->opix_y || !bipred_ok(ref, list, blocktype, &s1, &s2, &prm, lam3)) {
    if (!orig) orig = next_sym("FullPelBlockMotionBiPred");
    n_fwd[S_BIFULL]++;
    return orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x1, pred_mv_y1, pred_mv_x2, pred_mv_y2, mv_x, mv_y, s_mv_x, s_mv_y, search_range, min_mcost, lambda_factor);
  }
  memset(&job, 0, sizeof(job));
  job.mb_x = pic_pix_x >> 4; job.mb_y = pic_pix_y >> 4; job.ref1 = s1; job.ref2 = s2;
  job.s_mv[0] = *s_mv_x; job.s_mv[1] = *s_mv_y; job.mv[0] = *mv_x; job.mv[1] = *mv_y;
  job.pred1[0] = pred_mv_x1; job.pred1[1] = pred_mv_y1; job.pred2[0] = pred_mv_x2; job.pred2[1] = pred_mv_y2;
  job.min_mcost = min_mcost; job.search_range = search_range; job.stage = 0;
  OK(jmhip_bipred_search(g, &prm, &job, 1, &r));
  *mv_x = r.mv[0]; *mv_y = r.mv[1];
  n_dev[S_BIFULL]++;
  return r.cost;
}

int SubPelBlockSearchBiPred(imgpel *orig_pic, short ref, int list, int pic_pix_x, int pic_pix_y, int blocktype,
                            short pred_mv_x, short pred_mv_y, short *mv_x, short *mv_y, short *s_mv_x, short *s_mv_y,
                            int search_pos2, int search_pos4, int min_mcost, int *lambda)
{
  static int (*orig)(imgpel *, short, int, int, int, int, short, short, short *, short *, short *, short *, int, int, int, int *);
  jmhip_bipred_params prm; jmhip_bipred_job job; jmhip_bipred_result r;
  int s1, s2;
  if (search_pos2 != 9 || search_pos4 != 9 || pic_pix_x != img->opix_x || pic_pix_y != img->opix_y ||
      !bipred_ok(ref, list, blocktype, &s1, &s2, &prm, lambda)) {
    if (!orig) orig = next_sym("SubPelBlockSearchBiPred");
    n_fwd[S_BISUB]++;
    return orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, mv_x, mv_y, s_mv_x, s_mv_y, search_pos2, search_pos4, min_mcost, lambda);
  }
  memset(&job, 0, sizeof(job));
  job.mb_x = pic_pix_x >> 4; job.mb_y = pic_pix_y >> 4; job.ref1 = s1; job.ref2 = s2;
  job.s_mv[0] = *s_mv_x; job.s_mv[1] = *s_mv_y; job.mv[0] = *mv_x; job.mv[1] = *mv_y;
  job.pred2[0] = pred_mv_x; job.pred2[1] = pred_mv_y;
  job.min_mcost = min_mcost; job.stage = 1;
  OK(jmhip_bipred_search(g, &prm, &job, 1, &r));
  *mv_x = r.mv[0]; *mv_y = r.mv[1];
  n_dev[S_BISUB]++;
  return r.cost;
}

/* ------------------------------------------------------------------ EPZS / UMHexagonS: JM's walker, the device's distortions */

/* EPZSPelBlockMotionSearch (src/me_epzs.c:1500) and UMHEXIntegerPelBlockMotionSearch (src/me_umhex.c:229) choose their next
 * candidate from the costs of the previous ones, so the walk itself stays JM's code. What they spend their time in is
 * computeSAD / computeSATD at integer positions: the shim fetches, once per (macroblock, reference), the distortion of
 * EVERY integer displacement in the granularity of JM's early exits (jmhip_distortion_surface) and answers the kernels'
 * calls from it, partial sums included. A candidate outside the fetched window, at a sub-pel position, with weights or
 * chroma terms goes to JM's own kernel. */
static struct { int on; imgpel *orig; int bx0, by0, bsx, bsy, pic_x, pic_y, slot, px, py; StorablePicture *rp; } walk;
static struct { unsigned long serial; int mb_nr, cx, cy, R, w, o, rnd, den; uint16_t *buf; } surf[4][MAX_SLOTS];   /* kind + 2 * weighted */

static void walk_begin(imgpel *orig, short ref, int list, int pic_pix_x, int pic_pix_y, int blocktype, int pmx, int pmy)
{
  StorablePicture *rp; int slot;
  walk.on = 0;
  if (!(shim_mask & 0x100) || ref < 0 || !me_ok_metric(ref, list, &rp, &slot, 0)) return;
  walk.on = 1; walk.orig = orig; walk.rp = rp; walk.slot = slot;
  walk.pic_x = pic_pix_x; walk.pic_y = pic_pix_y;
  walk.bx0 = pic_pix_x - img->opix_x; walk.by0 = pic_pix_y - img->opix_y;
  walk.bsx = input->blc_size[blocktype][0]; walk.bsy = input->blc_size[blocktype][1];
  walk.px = pmx; walk.py = pmy;
  n_dev[S_WALK]++;
}

/* the surface of `kind` (+ weights) for the current macroblock and the walk's reference; NULL if the displacement is not covered */
static const uint16_t *surface_at(int kind, int wp, int cand_x, int cand_y, int nv)
{
  int mvx, mvy, ax, ay, UW, k = kind + 2 * wp;
  if (!walk.on || ((cand_x | cand_y) & 3) || ref_pic_sub.luma != walk.rp->p_curr_img_sub) return NULL;
  if (surf[k][walk.slot].serial != pic_serial || surf[k][walk.slot].mb_nr != img->current_mb_nr || !surf[k][walk.slot].buf ||
      (wp && (surf[k][walk.slot].w != weight_luma || surf[k][walk.slot].o != offset_luma || surf[k][walk.slot].rnd != wp_luma_round ||
              surf[k][walk.slot].den != luma_log_weight_denom))) {
    jmhip_surface_job job;
    int R = input->search_range > 32 ? 32 : input->search_range;
    UW = 2 * R + 1;
    if (!surf[k][walk.slot].buf || surf[k][walk.slot].R != R) {
      free(surf[k][walk.slot].buf);
      surf[k][walk.slot].buf = malloc((size_t)UW * UW * nv * sizeof(uint16_t));
    }
    memset(&job, 0, sizeof(job));
    job.mb_x = img->opix_x >> 4; job.mb_y = img->opix_y >> 4; job.ref = walk.slot; job.R = R;
    /* centred between the zero vector and the first block's predictor: the walks start from both */
    job.cx = iClip3(-R, R, walk.px / 8); job.cy = iClip3(-R, R, walk.py / 8);
    if (wp) { job.wp = 1; job.weight = weight_luma; job.offset = offset_luma; job.wp_round = wp_luma_round; job.wp_denom = luma_log_weight_denom; }
    OK(jmhip_distortion_surface(g, kind, &job, 1, surf[k][walk.slot].buf));
    surf[k][walk.slot].serial = pic_serial; surf[k][walk.slot].mb_nr = img->current_mb_nr;
    surf[k][walk.slot].cx = job.cx; surf[k][walk.slot].cy = job.cy; surf[k][walk.slot].R = R;
    surf[k][walk.slot].w = weight_luma; surf[k][walk.slot].o = offset_luma; surf[k][walk.slot].rnd = wp_luma_round; surf[k][walk.slot].den = luma_log_weight_denom;
  }
  UW = 2 * surf[k][walk.slot].R + 1;
  mvx = (cand_x >> 2) - IMG_PAD_SIZE - walk.pic_x; mvy = (cand_y >> 2) - IMG_PAD_SIZE - walk.pic_y;
  ax = mvx - surf[k][walk.slot].cx + surf[k][walk.slot].R; ay = mvy - surf[k][walk.slot].cy + surf[k][walk.slot].R;
  if (ax < 0 || ay < 0 || ax >= UW || ay >= UW) return NULL;
  return surf[k][walk.slot].buf + ((size_t)ay * UW + ax) * nv;
}

/* computeSAD / computeSADWP (src/me_distortion.c:351, :413) and computeSATD / computeSATDWP (:657, :734) share their loop
 * structure and exits; the weighted forms only read weighted reference samples, which the weighted surface already holds. */
static int sad_from_surface(const uint16_t *s, int bsy, int bsx, int min_mcost)
{
  int mcost = 0, y, gx, g0 = walk.bx0 >> 2, g1 = (walk.bx0 + bsx) >> 2;
  for (y = walk.by0; y < walk.by0 + bsy; y++) {       /* :364-375, the row-wise exit at :373 */
    for (gx = g0; gx < g1; gx++) mcost += s[y * 4 + gx];
    if (mcost >= min_mcost) return mcost;
  }
  return mcost;
}

static int satd_from_surface(const uint16_t *s, int bsy, int bsx, int min_mcost)
{
  int mcost = 0, y, x;                                /* :669-731: sub-blocks y outer, x inner; exit on '>' */
  if (!test8x8transform) {
    for (y = walk.by0; y < walk.by0 + bsy; y += 4)
      for (x = walk.bx0; x < walk.bx0 + bsx; x += 4) { mcost += s[(y >> 2) * 4 + (x >> 2)]; if (mcost > min_mcost) return mcost; }
  } else {
    for (y = walk.by0; y < walk.by0 + bsy; y += 8)
      for (x = walk.bx0; x < walk.bx0 + bsx; x += 8) { mcost += s[16 + (y >> 3) * 2 + (x >> 3)]; if (mcost > min_mcost) return mcost; }
  }
  return mcost;
}

#define DIST_HOOK(NAME, KIND, WP, NV, COUNTER, FROM, EXTRA)                                                          \
  int NAME(imgpel *src_pic, int bsy, int bsx, int min_mcost, int cand_x, int cand_y)                                  \
  {                                                                                                                   \
    static int (*orig)(imgpel *, int, int, int, int, int);                                                            \
    const uint16_t *s = NULL;                                                                                         \
    if (walk.on && src_pic == walk.orig && bsx == walk.bsx && bsy == walk.bsy && !ChromaMEEnable && (EXTRA))          \
      s = surface_at(KIND, WP, cand_x, cand_y, NV);                                                                   \
    if (!s) { if (!orig) orig = next_sym(#NAME); n_fwd[COUNTER]++; return orig(src_pic, bsy, bsx, min_mcost, cand_x, cand_y); } \
    n_dev[COUNTER]++;                                                                                                 \
    return FROM(s, bsy, bsx, min_mcost);                                                                              \
  }
DIST_HOOK(computeSAD,    JMHIP_SURFACE_SAD_ROWS,    0, 64, S_SAD,  sad_from_surface,  1)
DIST_HOOK(computeSADWP,  JMHIP_SURFACE_SAD_ROWS,    1, 64, S_SAD,  sad_from_surface,  1)
DIST_HOOK(computeSATD,   JMHIP_SURFACE_SATD_BLOCKS, 0, 20, S_SATD, satd_from_surface, (!test8x8transform || !((bsx | bsy) & 7)))
DIST_HOOK(computeSATDWP, JMHIP_SURFACE_SATD_BLOCKS, 1, 20, S_SATD, satd_from_surface, (!test8x8transform || !((bsx | bsy) & 7)))

int EPZSPelBlockMotionSearch(imgpel *cur_pic, short ref, int list, int list_offset, char ***refPic, short ****tmp_mv,
                             int pic_pix_x, int pic_pix_y, int blocktype, short pred_mv[2], short mv[2], int search_range,
                             int min_mcost, int lambda_factor)
{
  static int (*orig)(imgpel *, short, int, int, char ***, short ****, int, int, int, short *, short *, int, int, int);
  int r;
  if (!orig) orig = next_sym("EPZSPelBlockMotionSearch");
  if (!list_offset) walk_begin(cur_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv[0], pred_mv[1]);
  r = orig(cur_pic, ref, list, list_offset, refPic, tmp_mv, pic_pix_x, pic_pix_y, blocktype, pred_mv, mv, search_range, min_mcost, lambda_factor);
  walk.on = 0;
  return r;
}

#define UMHEX_WALK(NAME)                                                                                              \
  int NAME(imgpel *orig_pic, short ref, int list, int pic_pix_x, int pic_pix_y, int blocktype, short pred_mv_x,        \
           short pred_mv_y, short *mv_x, short *mv_y, int search_range, int min_mcost, int lambda_factor)             \
  {                                                                                                                   \
    static int (*orig)(imgpel *, short, int, int, int, int, short, short, short *, short *, int, int, int);           \
    int r;                                                                                                            \
    if (!orig) orig = next_sym(#NAME);                                                                                \
    walk_begin(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y);                           \
    r = orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, mv_x, mv_y, search_range, min_mcost, lambda_factor); \
    walk.on = 0;                                                                                                      \
    return r;                                                                                                         \
  }
UMHEX_WALK(UMHEXIntegerPelBlockMotionSearch)
UMHEX_WALK(smpUMHEXIntegerPelBlockMotionSearch)

/* ------------------------------------------------------------------ RD-off mode-decision costs */

extern void SetModesAndRefframe(Macroblock *currMB, int b8, short *p_dir, int *l0_mode, int *l1_mode, short *l0_ref, short *l1_ref);
extern void LumaPrediction(Macroblock *currMB, int block_x, int block_y, int block_size_x, int block_size_y, int p_dir, int l0_mode,
                           int l1_mode, short l0_ref_idx, short l1_ref_idx);

/* one job from JM's state: per 4x4 block the list-0 vector of its mode and the slot of its reference; 0 if the device cannot take it
 * (bi-pred / list 1 / direct blocks, weighted prediction, field pictures, SSE) */
static int predcost_job(Macroblock *currMB, jmhip_predcost_job *job, int skip)
{
  int b8, bx, by;
  int weighted = (active_pps->weighted_pred_flag && (img->type == P_SLICE || img->type == SP_SLICE)) || (active_pps->weighted_bipred_idc && img->type == B_SLICE);
  if (weighted || input->ModeDecisionMetric == ERROR_SSE || !cur_ready() || img->mb_data[img->current_mb_nr].list_offset) return 0;
  memset(job, 0, sizeof(*job));
  job->mb_x = img->opix_x >> 4; job->mb_y = img->opix_y >> 4; job->blocks = 0xffff;
  for (b8 = 0; b8 < 4; b8++) {
    short p_dir = 0, l0_ref = 0, l1_ref = 0; int l0_mode = 0, l1_mode = 0, slot;
    if (!skip) {
      SetModesAndRefframe(currMB, b8, &p_dir, &l0_mode, &l1_mode, &l0_ref, &l1_ref);
      if (p_dir != 0 || l0_mode < 1 || l0_mode > 7 || l0_ref < 0) return 0;
    }
    slot = slot_find(listX[LIST_0][l0_ref]);
    if (slot < 0) return 0;
    for (by = (b8 >> 1) * 2; by < (b8 >> 1) * 2 + 2; by++)
      for (bx = (b8 & 1) * 2; bx < (b8 & 1) * 2 + 2; bx++) {
        short *mv = img->all_mv[by][bx][LIST_0][l0_ref][l0_mode];      /* LumaPrediction, macroblock.c:851-870 */
        job->mv[by * 4 + bx][0] = mv[0]; job->mv[by * 4 + bx][1] = mv[1];
        job->ref[by * 4 + bx] = (int8_t)slot;
      }
  }
  return 1;
}

int TransformDecision(Macroblock *currMB, int block_check, int *cost)
{
  static int (*orig)(Macroblock *, int, int *);
  jmhip_predcost_job job; int32_t out[1][2];
  if (!(shim_mask & 0x400) || block_check != -1 || !predcost_job(currMB, &job, 0)) {
    if (!orig) orig = next_sym("TransformDecision");
    n_fwd[S_TDEC]++;
    return orig(currMB, block_check, cost);
  }
  OK(jmhip_pred_cost_batch(g, &job, 1, input->ModeDecisionMetric, JMHIP_DIFF64_SEQUENTIAL, out));
  {
    /* JM leaves the macroblock's prediction in img->mpr (later code may read it): keep that side effect with JM's own routine */
    int b8, bx, by, l0_mode, l1_mode; short p_dir, l0_ref, l1_ref;
    for (b8 = 0; b8 < 4; b8++) {
      SetModesAndRefframe(currMB, b8, &p_dir, &l0_mode, &l1_mode, &l0_ref, &l1_ref);
      for (by = (b8 >> 1) << 3; by < ((b8 >> 1) << 3) + 8; by += 4)
        for (bx = (b8 & 1) << 3; bx < ((b8 & 1) << 3) + 8; bx += 4) LumaPrediction(currMB, bx, by, 4, 4, p_dir, l0_mode, l1_mode, l0_ref, l1_ref);
    }
  }
  n_dev[S_TDEC]++;
  if (input->Transform8x8Mode == 2) return 1;          /* macroblock.c:1508-1517 */
  if (out[0][1] < out[0][0]) return 1;
  *cost = (*cost - out[0][1] + out[0][0]);
  return 0;
}

int GetSkipCostMB(Macroblock *currMB)
{
  static int (*orig)(Macroblock *);
  jmhip_predcost_job job; int32_t out[1][2];
  if (!(shim_mask & 0x400) || !predcost_job(currMB, &job, 1)) {
    if (!orig) orig = next_sym("GetSkipCostMB");
    n_fwd[S_SKIPC]++;
    return orig(currMB);
  }
  OK(jmhip_pred_cost_batch(g, &job, 1, input->ModeDecisionMetric, JMHIP_DIFF64_RASTER, out));
  { int bx, by; for (by = 0; by < 16; by += 4) for (bx = 0; bx < 16; bx += 4) LumaPrediction(currMB, bx, by, 4, 4, 0, 0, 0, 0, 0); }   /* img->mpr side effect */
  n_dev[S_SKIPC]++;
  return (input->rdopt == 0 && input->Transform8x8Mode) ? out[0][1] : out[0][0];      /* mv-search.c:1167-1177 */
}

/* BIDPartitionCost (src/mv-search.c:1050): mvd bits on the host, the bi-predicted residual distortion of the partition on the device */
int BIDPartitionCost(Macroblock *currMB, int blocktype, int block8x8, short ref_l0, short ref_l1, int lambda_factor)
{
  static int (*orig)(Macroblock *, int, int, short, short, int);
  static const int bx0[5][4] = {{0,0,0,0}, {0,0,0,0}, {0,0,0,0}, {0,2,0,0}, {0,2,0,2}};
  static const int by0[5][4] = {{0,0,0,0}, {0,0,0,0}, {0,2,0,0}, {0,0,0,0}, {0,0,2,2}};
  jmhip_predcost_job job; int32_t out[1][2];
  int parttype = blocktype < 4 ? blocktype : 4;
  int step_h0 = input->part_size[parttype][0], step_v0 = input->part_size[parttype][1];
  int step_h = input->part_size[blocktype][0], step_v = input->part_size[blocktype][1];
  int bx = bx0[parttype][block8x8], by = by0[parttype][block8x8], v, h, mvd_bits = 0, s0, s1;
  int weighted = (active_pps->weighted_bipred_idc && img->type == B_SLICE) || (active_pps->weighted_pred_flag && (img->type == P_SLICE || img->type == SP_SLICE));
  int ok = (shim_mask & 0x400) && input->ModeDecisionMetric != ERROR_SSE && cur_ready() && !img->mb_data[img->current_mb_nr].list_offset &&
           ref_l0 >= 0 && ref_l1 >= 0;
  if (ok) { s0 = slot_find(listX[LIST_0][ref_l0]); s1 = slot_find(listX[LIST_1][ref_l1]); ok = s0 >= 0 && s1 >= 0; }
  if (!ok) {
    if (!orig) orig = next_sym("BIDPartitionCost");
    n_fwd[S_BIDC]++;
    return orig(currMB, blocktype, block8x8, ref_l0, ref_l1, lambda_factor);
  }
  memset(&job, 0, sizeof(job));
  job.mb_x = img->opix_x >> 4; job.mb_y = img->opix_y >> 4;
  job.weighted = weighted; job.wp_round = wp_luma_round; job.wp_denom = luma_log_weight_denom;
  for (v = by; v < by + step_v0; v += step_v)
    for (h = bx; h < bx + step_h0; h += step_h) {
      mvd_bits += mvbits[img->all_mv[v][h][LIST_0][ref_l0][blocktype][0] - img->pred_mv[v][h][LIST_0][ref_l0][blocktype][0]];
      mvd_bits += mvbits[img->all_mv[v][h][LIST_0][ref_l0][blocktype][1] - img->pred_mv[v][h][LIST_0][ref_l0][blocktype][1]];
      mvd_bits += mvbits[img->all_mv[v][h][LIST_1][ref_l1][blocktype][0] - img->pred_mv[v][h][LIST_1][ref_l1][blocktype][0]];
      mvd_bits += mvbits[img->all_mv[v][h][LIST_1][ref_l1][blocktype][1] - img->pred_mv[v][h][LIST_1][ref_l1][blocktype][1]];
    }
  for (v = by; v < by + step_v0; v++)
    for (h = bx; h < bx + step_h0; h++) {
      int b = v * 4 + h;
      short ****mv_array = img->all_mv[v][h];
      /* the bi-pred ME vectors replace the ordinary ones for the 16x16 / ref 0 pair (LumaPrediction, macroblock.c:862-863) */
      if (currMB->bi_pred_me && ref_l0 == 0 && ref_l1 == 0 && blocktype == 1) mv_array = currMB->bi_pred_me == 1 ? img->bipred_mv1[v][h] : img->bipred_mv2[v][h];
      job.blocks |= (uint16_t)(1u << b);
      job.bi[b] = 1; job.ref[b] = (int8_t)s0; job.ref1[b] = (int8_t)s1;
      job.mv[b][0] = mv_array[LIST_0][ref_l0][blocktype][0]; job.mv[b][1] = mv_array[LIST_0][ref_l0][blocktype][1];
      job.mv1[b][0] = mv_array[LIST_1][ref_l1][blocktype][0]; job.mv1[b][1] = mv_array[LIST_1][ref_l1][blocktype][1];
      if (weighted) {
        job.w0[b] = wbp_weight[0][ref_l0][ref_l1][0]; job.w1[b] = wbp_weight[1][ref_l0][ref_l1][0];
        job.off[b] = (wp_offset[0][ref_l0][0] + wp_offset[1][ref_l1][0] + 1) >> 1;
      }
    }
  OK(jmhip_pred_cost_batch(g, &job, 1, input->ModeDecisionMetric, JMHIP_DIFF64_RASTER, out));
  for (v = by; v < by + step_v0; v++)                   /* img->mpr side effect, with JM's own routine */
    for (h = bx; h < bx + step_h0; h++) LumaPrediction(currMB, h << 2, v << 2, 4, 4, 2, blocktype, blocktype, ref_l0, ref_l1);
  n_dev[S_BIDC]++;
  return WEIGHTED_COST(lambda_factor, mvd_bits) + ((input->Transform8x8Mode && blocktype <= 4) ? out[0][1] : out[0][0]);
}

/* ------------------------------------------------------------------ transform + quantisation + reconstruction */

static void fill_quant(jmhip_quant *q, int qp, int **levelscale, int **invlevelscale, int **leveloffset, int n,
                       Macroblock *currMB, int weight, int max_val)
{
  int j, i;
  memset(q, 0, sizeof(*q));
  for (j = 0; j < n; j++) for (i = 0; i < n; i++) {
    q->levelscale[j * n + i] = levelscale[j][i]; q->invlevelscale[j * n + i] = invlevelscale[j][i]; q->leveloffset[j * n + i] = leveloffset[j][i];
  }
  q->qp = qp; q->adaptive_rounding = img->AdaptiveRounding; q->adapt_rnd_weight = weight;
  q->field_scan = currMB->is_field_mode; q->disthres = input->disthres;
  q->max_val = max_val; q->cavlc = (input->symbol_mode == CAVLC); q->img_qp = img->qp;
  q->transform8x8_flag = currMB->luma_transform_size_8x8_flag;
}

/* job tiles: src = m7 + mpr (the original samples where the block is coded; elsewhere JM's m7 may be stale and the
 * value is irrelevant: every block of a job is computed on its own). Returns 0 if a coded sample leaves 8 bits. */
static int fill_tiles(jmhip_tq_job *job, int (*m7)[16], imgpel (*mpr)[16], int x0, int y0, int w, int h)
{
  int j, i, ok = 1;
  for (j = 0; j < 16; j++) for (i = 0; i < 16; i++) {
    int s = m7[j][i] + mpr[j][i], in = (j >= y0 && j < y0 + h && i >= x0 && i < x0 + w);
    if (in && (s < 0 || s > 255 || mpr[j][i] > 255)) ok = 0;
    job->src[j][i] = (uint8_t)(s < 0 ? 0 : s > 255 ? 255 : s);
    job->pred[j][i] = (uint8_t)mpr[j][i];
  }
  return ok;
}

static void put_list(int *lev_dst, int *run_dst, const int32_t *lev, const int32_t *run, int max)
{
  int k;
  for (k = 0; k < max; k++) { lev_dst[k] = lev[k]; run_dst[k] = run[k]; if (!lev[k]) break; }
}

static int fr_dct4(Macroblock *currMB, int block_x, int block_y, int *coeff_cost, int *ret);
static int fr_dctc(Macroblock *currMB, int uv, int cr_cbp, int *ret);

int dct_4x4(Macroblock *currMB, ColorPlane pl, int block_x, int block_y, int *coeff_cost, int intra)
{
  static int (*orig)(Macroblock *, ColorPlane, int, int, int *, int);
  static jmhip_tq_job job; static jmhip_tq_result res; static jmhip_quant q;
  { int r; if (pl == 0 && !intra && fr_dct4(currMB, block_x, block_y, coeff_cost, &r)) return r; }
  int ok = (shim_mask & 0x10) && ctx_ready() && !(currMB->qp_scaled[pl] == 0 && img->lossless_qpprime_flag == 1) &&
           img->type != SP_SLICE;
  if (ok) ok = fill_tiles(&job, img->m7[pl], img->mpr[pl], block_x, block_y, 4, 4);
  if (!ok) {
    if (!orig) orig = next_sym("dct_4x4");
    n_fwd[S_D4]++;
    return orig(currMB, pl, block_x, block_y, coeff_cost, intra);
  }
  {
    int qp = currMB->qp_scaled[pl], qp_rem = qp_rem_matrix[qp], j, i;
    int pos_x = block_x >> 2, pos_y = block_y >> 2;
    int b8 = 2 * (pos_y >> 1) + (pos_x >> 1), b4 = 2 * (pos_y & 1) + (pos_x & 1), blk = b8 * 4 + b4;
    int **fa = img->AdaptiveRounding ? (pl ? img->fadjust4x4Cr[pl - 1][intra] : img->fadjust4x4[intra]) : NULL;
    imgpel **img_enc = enc_picture->p_curr_img;
    fill_quant(&q, qp, LevelScale4x4Comp[pl][intra][qp_rem], InvLevelScale4x4Comp[pl][intra][qp_rem],
               ptLevelOffset4x4[intra][qp], 4, currMB, AdaptRndWeight, img->max_imgpel_value);
    job.quant = 0;
    OK(jmhip_tq_batch(g, JMHIP_TQ_LUMA4x4, img->yuv_format, &q, 1, &job, 1, &res));
    put_list(img->cofAC[b8 + (pl << 2)][b4][0], img->cofAC[b8 + (pl << 2)][b4][1], res.levels[blk], res.runs[blk], 17);
    *coeff_cost += res.coeff_cost[blk];
    for (j = block_y; j < block_y + 4; j++) for (i = block_x; i < block_x + 4; i++) {
      img_enc[img->pix_y + j][img->pix_x + i] = res.recon[j][i];
      if (fa) fa[j][i] = res.fadjust[j][i];
    }
    n_dev[S_D4]++;
    return res.nonzero[blk];
  }
}

int dct_8x8(Macroblock *currMB, ColorPlane pl, int b8, int *coeff_cost, int intra)
{
  static int (*orig)(Macroblock *, ColorPlane, int, int *, int);
  static jmhip_tq_job job; static jmhip_tq_result res; static jmhip_quant q;
  int block_x = 8 * (b8 & 1), block_y = 8 * (b8 >> 1);
  int ok = (shim_mask & 0x20) && ctx_ready() && !(currMB->qp_scaled[pl] == 0 && img->lossless_qpprime_flag == 1) &&
           img->type != SP_SLICE;
  if (ok) ok = fill_tiles(&job, img->m7[pl], img->mpr[pl], block_x, block_y, 8, 8);
  if (!ok) {
    if (!orig) orig = next_sym("dct_8x8");
    n_fwd[S_D8]++;
    return orig(currMB, pl, b8, coeff_cost, intra);
  }
  {
    int qp = currMB->qp_scaled[pl], qp_rem = qp_rem_matrix[qp], j, i, k;
    int **fa = img->AdaptiveRounding ? (pl ? img->fadjust8x8Cr[pl - 1][intra] : img->fadjust8x8[intra]) : NULL;
    imgpel **img_enc = enc_picture->p_curr_img;
    fill_quant(&q, qp, LevelScale8x8Comp[pl][intra][qp_rem], InvLevelScale8x8Comp[pl][intra][qp_rem],
               LevelOffset8x8Comp[pl][intra][qp], 8, currMB, AdaptRndWeight, img->max_imgpel_value);
    job.quant = 0;
    OK(jmhip_tq_batch(g, JMHIP_TQ_LUMA8x8, img->yuv_format, &q, 1, &job, 1, &res));
    if (q.transform8x8_flag && q.cavlc)          /* four interleaved 4x4 lists, transform8x8.c:1560-1580 */
      for (k = 0; k < 4; k++)
        put_list(img->cofAC[b8 + (pl << 2)][k][0], img->cofAC[b8 + (pl << 2)][k][1], res.levels[4 * b8 + k], res.runs[4 * b8 + k], 17);
    else
      put_list(img->cofAC[b8 + (pl << 2)][0][0], img->cofAC[b8 + (pl << 2)][0][1], res.levels8[b8], res.runs8[b8], 65);
    *coeff_cost += res.coeff_cost[b8];
    for (j = block_y; j < block_y + 8; j++) for (i = block_x; i < block_x + 8; i++) {
      img_enc[img->pix_y + j][img->pix_x + i] = res.recon[j][i];
      if (fa) fa[j][i] = res.fadjust[j][i];
    }
    n_dev[S_D8]++;
    return res.nonzero[b8];
  }
}

int dct_16x16(Macroblock *currMB, ColorPlane pl, int new_intra_mode)
{
  static int (*orig)(Macroblock *, ColorPlane, int);
  static jmhip_tq_job job; static jmhip_tq_result res; static jmhip_quant q;
  int ok = (shim_mask & 0x10) && ctx_ready() && pl == 0 &&
           !(currMB->qp_scaled[pl] == 0 && img->lossless_qpprime_flag == 1) && img->type != SP_SLICE;
  if (!ok) {
    if (!orig) orig = next_sym("dct_16x16");
    n_fwd[S_D16]++;
    return orig(currMB, pl, new_intra_mode);
  }
  {
    int qp = currMB->qp_scaled[pl], qp_rem = qp_rem_matrix[qp], j, i, b;
    int **fa = img->AdaptiveRounding ? img->fadjust4x4[2] : NULL;
    imgpel **img_enc = enc_picture->p_curr_img;
    fill_quant(&q, qp, LevelScale4x4Comp[pl][1][qp_rem], InvLevelScale4x4Comp[pl][1][qp_rem],
               ptLevelOffset4x4[1][qp], 4, currMB, AdaptRndWeight, img->max_imgpel_value);
    for (j = 0; j < 16; j++) for (i = 0; i < 16; i++) {
      job.src[j][i] = (uint8_t)pCurImg[img->opix_y + j][img->opix_x + i];
      job.pred[j][i] = (uint8_t)img->mpr_16x16[pl][new_intra_mode][j][i];
    }
    job.quant = 0;
    OK(jmhip_tq_batch(g, JMHIP_TQ_LUMA16x16, img->yuv_format, &q, 1, &job, 1, &res));
    put_list(img->cofDC[pl][0], img->cofDC[pl][1], res.dc_levels, res.dc_runs, 17);
    for (b = 0; b < 16; b++) put_list(img->cofAC[(b >> 2) + (pl << 2)][b & 3][0], img->cofAC[(b >> 2) + (pl << 2)][b & 3][1], res.levels[b], res.runs[b], 16);
    for (j = 0; j < 16; j++) for (i = 0; i < 16; i++) {
      img_enc[img->pix_y + j][img->pix_x + i] = res.recon[j][i];
      if (fa && ((j | i) & 3)) fa[j][i] = res.fadjust[j][i];      /* JM leaves the DC positions alone, block.c:690-720 */
    }
    n_dev[S_D16]++;
    return res.ret;
  }
}

int dct_chroma(Macroblock *currMB, int uv, int cr_cbp)
{
  static int (*orig)(Macroblock *, int, int);
  static jmhip_tq_job job; static jmhip_tq_result res; static jmhip_quant q[2];
  { int r; if (fr_dctc(currMB, uv, cr_cbp, &r)) return r; }
  int ok = (shim_mask & 0x40) && ctx_ready() && img->yuv_format != YUV444 && img->yuv_format != YUV400 &&
           !((currMB->qp + img->bitdepth_luma_qp_scale) == 0 && img->lossless_qpprime_flag == 1) && img->type != SP_SLICE;
  if (ok) ok = fill_tiles(&job, img->m7[uv + 1], img->mpr[uv + 1], 0, 0, img->mb_cr_size_x, img->mb_cr_size_y);
  if (!ok) {
    if (!orig) orig = next_sym("dct_chroma");
    n_fwd[S_DCR]++;
    return orig(currMB, uv, cr_cbp);
  }
  {
    int intra = IS_INTRA(currMB), j, i, b;
    int cur_qp = currMB->qpc[uv] + img->bitdepth_chroma_qp_scale;
    int cur_qp_dc = currMB->qpc[uv] + 3 + img->bitdepth_chroma_qp_scale;
    int nb = (img->num_blk8x8_uv >> 1) * 4, uv_scale = uv * (img->num_blk8x8_uv >> 1);
    int **fa = img->AdaptiveRounding ? img->fadjust4x4Cr[intra][uv] : NULL;
    fill_quant(&q[0], cur_qp, LevelScale4x4Comp[uv + 1][intra][qp_rem_matrix[cur_qp]],
               InvLevelScale4x4Comp[uv + 1][intra][qp_rem_matrix[cur_qp]], LevelOffset4x4Comp[uv + 1][intra][cur_qp],
               4, currMB, AdaptRndCrWeight, img->max_imgpel_value_comp[1]);
    q[1] = q[0];
    if (img->yuv_format == YUV422)
      fill_quant(&q[1], cur_qp_dc, LevelScale4x4Comp[uv + 1][intra][qp_rem_matrix[cur_qp_dc]],
                 InvLevelScale4x4Comp[uv + 1][intra][qp_rem_matrix[cur_qp_dc]], LevelOffset4x4Comp[uv + 1][intra][cur_qp_dc],
                 4, currMB, AdaptRndCrWeight, img->max_imgpel_value_comp[1]);
    job.quant = 0; job.quant_dc = 1; job.uv = uv; job.cr_cbp_in = cr_cbp;
    OK(jmhip_tq_batch(g, JMHIP_TQ_CHROMA, img->yuv_format, q, 2, &job, 1, &res));
    put_list(img->cofDC[uv + 1][0], img->cofDC[uv + 1][1], res.dc_levels, res.dc_runs, 17);
    for (b = 0; b < nb; b++)
      put_list(img->cofAC[4 + (b >> 2) + uv_scale][b & 3][0], img->cofAC[4 + (b >> 2) + uv_scale][b & 3][1], res.levels[b], res.runs[b], 16);
    currMB->cbp_blk = (currMB->cbp_blk & ~res.cbp_clear) | res.cbp_blk;
    for (j = 0; j < img->mb_cr_size_y; j++) for (i = 0; i < img->mb_cr_size_x; i++) {
      enc_picture->imgUV[uv][img->pix_c_y + j][img->pix_c_x + i] = res.recon[j][i];
      if (fa && ((j | i) & 3)) fa[j][i] = res.fadjust[j][i];      /* AC positions only, block.c:1321-1380 */
    }
    n_dev[S_DCR]++;
    return res.ret;
  }
}

/* ------------------------------------------------------------------ 0x800 in-loop deblocking filter */


/* ------------------------------------------------------------------ 0x1000 slice-level binding */

extern void FindSkipModeMotionVector(Macroblock *currMB);
extern int frame_ctr[5];

static struct {
  int active;                 /* results below belong to the slice being coded */
  unsigned long serial;       /* pic_serial of the picture they belong to */
  int slice_nr, mb_first, mb_count;
  jmhip_mb_inter *rec; int cap;
  int decided, on;            /* once per run: is the configuration covered? */
  int multi;                  /* the records cover every remaining slice of the picture (slice_mbs) */
  int spec;                   /* speculative mode (mask 0x2000): JM decides with rate-distortion costs / intra candidates; a call is answered only when
                                 its predictor equals the record's, otherwise JM's own BlockMotionSearch runs */
  int started;                /* device state reset done */
  long passes, slices;
} sl;


/* ------------------------------------------------------------------ 0x4000 the frame stage at slice level (exact form of the slice binding) */

static struct {
  int active;                 /* records and prediction picture below belong to macroblocks [mb_first, mb_first + mb_count) of picture `serial` */
  unsigned long serial;
  int mb_first, mb_count;
  jmhip_mb_residual *rec; int cap;
  imgpel *pred[3]; int pred_ready;
  /* ... and the same for the P8x8 CANDIDATE of every macroblock, which JM predicts and transforms per 8x8 block inside submacroblock_mode_decision
     (src/mode_decision.c:874) before it decides: luma only (jmhip_slice_to_frame_candidates) */
  jmhip_mb_residual *rec2; int cap2; imgpel *pred2; int have2;
  /* JM also predicts and transforms blocks the decision then discards (the P8x8 candidate of every macroblock, src/mode_decision.c:874): a dct call
     is answered from the record only when the prediction it works on was -- per macroblock, the luma 4x4 blocks / chroma 4x4 blocks whose LAST
     prediction call asked for the decided block and was answered from the prediction picture */
  int ok_mb; unsigned long ok_serial; unsigned ok_luma, ok_luma2, ok_chroma[2];      /* ok_luma2: ... was answered from the candidate's prediction picture */
} fr;

static void fr_ok_mb(void)
{
  if (fr.ok_mb != img->current_mb_nr || fr.ok_serial != pic_serial) { fr.ok_mb = img->current_mb_nr; fr.ok_serial = pic_serial; fr.ok_luma = fr.ok_luma2 = 0; fr.ok_chroma[0] = fr.ok_chroma[1] = 0; }
}
static unsigned fr_luma_bits(int block_x, int block_y, int bsx, int bsy)
{
  unsigned m = 0; int bx, by;
  for (by = block_y >> 2; by < (block_y + bsy) >> 2; by++) for (bx = block_x >> 2; bx < (block_x + bsx) >> 2; bx++) m |= 1u << (by * 4 + bx);
  return m;
}

static const jmhip_mb_residual *fr_rec_cur(void)
{
  if (!fr.active || fr.serial != pic_serial || img->type != P_SLICE) return NULL;
  if (img->current_mb_nr < fr.mb_first || img->current_mb_nr >= fr.mb_first + fr.mb_count) return NULL;
  return &fr.rec[img->current_mb_nr - fr.mb_first];
}

static void fr_diverged(const char *what, int x, int y)
{
  fprintf(stderr, "jm_shim: frame binding diverged at mb %d, %s block (%d,%d): JM's prediction / residual is not what the device coded\n", img->current_mb_nr, what, x, y);
  exit(95);
}

/* right after the slice search: prediction, residual, transform, quantisation and reconstruction of the slice's macroblocks on the device */
static void fr_run(const jmhip_slice_params *p, int first, int count)
{
  static jmhip_quant q[3];
  jmhip_quant qv;
  jmhip_frame_wp wp;
  Macroblock *mb = &img->mb_data[first];
  const int qp = mb->qp_scaled[0], qc0 = mb->qpc[0] + img->bitdepth_chroma_qp_scale, qc1 = mb->qpc[1] + img->bitdepth_chroma_qp_scale;
  const double t0 = now_s();
  int r, c;
  fr.active = 0;
  if (!(shim_mask & 0x4000) || sl.spec || img->yuv_format != YUV420 || input->Transform8x8Mode || img->type != P_SLICE || img->NoResidueDirect ||
      (qp == 0 && img->lossless_qpprime_flag == 1) || mb->is_field_mode) return;
  for (r = 0; r < p->num_refs; r++) if (p->ref_slot[r] >= 8 && !slots[p->ref_slot[r]].has_chroma) return;      /* chroma samples are computed from reference slots 0..7 only */
  /* the inter quantisers of the slice (no rate control, no macroblock-level dquant: one qp); Cb and Cr must share theirs */
  fill_quant(&q[0], qp, LevelScale4x4Comp[0][0][qp_rem_matrix[qp]], InvLevelScale4x4Comp[0][0][qp_rem_matrix[qp]], ptLevelOffset4x4[0][qp], 4, mb, AdaptRndWeight, img->max_imgpel_value);
  fill_quant(&q[1], qc0, LevelScale4x4Comp[1][0][qp_rem_matrix[qc0]], InvLevelScale4x4Comp[1][0][qp_rem_matrix[qc0]], LevelOffset4x4Comp[1][0][qc0], 4, mb, AdaptRndCrWeight, img->max_imgpel_value_comp[1]);
  fill_quant(&qv, qc1, LevelScale4x4Comp[2][0][qp_rem_matrix[qc1]], InvLevelScale4x4Comp[2][0][qp_rem_matrix[qc1]], LevelOffset4x4Comp[2][0][qc1], 4, mb, AdaptRndCrWeight, img->max_imgpel_value_comp[2]);
  if (memcmp(&q[1], &qv, sizeof(qv))) return;
  q[2] = q[1];
  memset(&wp, 0, sizeof(wp));
  if (active_pps->weighted_pred_flag) {
    wp.enable = 1; wp.luma_round = wp_luma_round; wp.luma_denom = luma_log_weight_denom; wp.chroma_round = wp_chroma_round; wp.chroma_denom = chroma_log_weight_denom;
    for (r = 0; r < p->num_refs; r++) for (c = 0; c < 3; c++) { wp.weight[p->ref_slot[r]][c] = (int16_t)wp_weight[0][r][c]; wp.offset[p->ref_slot[r]][c] = (int16_t)wp_offset[0][r][c]; }
  }
  OK(jmhip_frame_wp_set(g, &wp));
  OK(jmhip_frame_keep_prediction(g, 1));
  fr.have2 = 0;
  if (input->InterSearch[0][4] || input->InterSearch[0][5] || input->InterSearch[0][6] || input->InterSearch[0][7]) {
    /* first the P8x8 candidates (their reconstruction is not a picture: the decision's pass below overwrites it) */
    OK(jmhip_slice_to_frame_candidates(g, p->ref_slot, p->num_refs, first, count));
    OK(jmhip_residual_frame(g, NULL, q));
    if (fr.cap2 < count) { free(fr.rec2); fr.rec2 = malloc(sizeof(jmhip_mb_residual) * (size_t)count); fr.cap2 = count; }
    if (!fr.pred2) fr.pred2 = malloc(sizeof(imgpel) * (size_t)g_w * g_h);
    if (!fr.rec2 || !fr.pred2) { fprintf(stderr, "jm_shim: out of memory\n"); exit(96); }
    OK(jmhip_residual_records_download(g, fr.rec2, count));
    OK(jmhip_pred_download(g, fr.pred2, NULL, NULL, (int)sizeof(imgpel)));
    fr.have2 = 1;
  }
  OK(jmhip_slice_to_frame_band(g, p->ref_slot, p->num_refs, first, count));
  OK(jmhip_residual_frame(g, NULL, q));
  if (fr.cap < count) { free(fr.rec); fr.rec = malloc(sizeof(jmhip_mb_residual) * (size_t)count); fr.cap = count; }
  if (!fr.pred_ready) {
    fr.pred[0] = malloc(sizeof(imgpel) * (size_t)g_w * g_h); fr.pred[1] = malloc(sizeof(imgpel) * (size_t)(g_w / 2) * (g_h / 2)); fr.pred[2] = malloc(sizeof(imgpel) * (size_t)(g_w / 2) * (g_h / 2));
    fr.pred_ready = 1;
  }
  if (!fr.rec || !fr.pred[0] || !fr.pred[1] || !fr.pred[2]) { fprintf(stderr, "jm_shim: out of memory\n"); exit(96); }
  OK(jmhip_residual_records_download(g, fr.rec, count));
  OK(jmhip_pred_download(g, fr.pred[0], fr.pred[1], fr.pred[2], (int)sizeof(imgpel)));
  fr.active = 1; fr.serial = pic_serial; fr.mb_first = first; fr.mb_count = count;
  n_dev[S_FRAME]++; t_last[S_FRAME] = now_s() - t0; t_dev[S_FRAME] += t_last[S_FRAME];
}

/* does JM ask for the prediction the device formed for the luma 4x4 blocks [bx0, bx1) x [by0, by1) of the current macroblock? */
static int fr_asks_decided(int bx0, int by0, int bx1, int by1, int p_dir, int l0_mode, short l0_ref)
{
  const jmhip_mb_inter *d = &sl.rec[img->current_mb_nr - sl.mb_first];
  int bx, by;
  if (p_dir != 0 || l0_ref < 0 || l0_ref >= JMHIP_SLICE_REFS || l0_mode < 1 || l0_mode > 7) return 0;
  for (by = by0; by < by1; by++) for (bx = bx0; bx < bx1; bx++) {
    const int b8 = 2 * (by >> 1) + (bx >> 1), mode = d->best_mode == 8 ? d->b8mode[b8] : d->best_mode;
    const short *v = img->all_mv[by][bx][LIST_0][l0_ref][l0_mode];
    if (l0_mode != mode || l0_ref != d->b8ref[b8] || v[0] != d->final_mv[by * 4 + bx][0] || v[1] != d->final_mv[by * 4 + bx][1]) {
      static int told;
      if (getenv("JMHIP_SHIM_DEBUG") && told++ < 12)
        fprintf(stderr, "jm_shim debug: mb %d block (%d,%d): JM asks mode %d ref %d mv (%d,%d), device decided best_mode %d b8mode %d ref %d mv (%d,%d)\n", img->current_mb_nr, bx, by, l0_mode, l0_ref,
                v[0], v[1], d->best_mode, d->b8mode[b8], d->b8ref[b8], d->final_mv[by * 4 + bx][0], d->final_mv[by * 4 + bx][1]);
      return 0;
    }
  }
  return 1;
}

/* ... or for the P8x8 candidate's prediction of them (sub-mode and reference of the 8x8 block, the vectors the slice search found for that sub-mode)? */
static int fr_asks_candidate(int bx0, int by0, int bx1, int by1, int p_dir, int l0_mode, short l0_ref)
{
  const jmhip_mb_inter *d = &sl.rec[img->current_mb_nr - sl.mb_first];
  int bx, by;
  if (!fr.have2 || p_dir != 0 || l0_ref < 0 || l0_ref >= JMHIP_SLICE_REFS || l0_mode < 4 || l0_mode > 7) return 0;
  for (by = by0; by < by1; by++) for (bx = bx0; bx < bx1; bx++) {
    const int b8 = 2 * (by >> 1) + (bx >> 1);
    const int ox = (l0_mode == 4 || l0_mode == 5) ? (bx & ~1) : bx, oy = (l0_mode == 4 || l0_mode == 6) ? (by & ~1) : by;      /* origin of the sub-partition that covers the block */
    const int pi = partition_of(l0_mode, ox << 2, oy << 2);
    const short *v = img->all_mv[by][bx][LIST_0][l0_ref][l0_mode];
    if (l0_mode != d->p8mode[b8] || l0_ref != d->p8ref[b8] || pi < 0 || v[0] != d->mv[l0_ref][pi][0] || v[1] != d->mv[l0_ref][pi][1]) return 0;
  }
  return 1;
}

void LumaPrediction(Macroblock *currMB, int block_x, int block_y, int block_size_x, int block_size_y, int p_dir, int l0_mode, int l1_mode, short l0_ref_idx, short l1_ref_idx)
{
  static void (*orig)(Macroblock *, int, int, int, int, int, int, int, short, short);
  if (fr_rec_cur()) { fr_ok_mb(); fr.ok_luma &= ~fr_luma_bits(block_x, block_y, block_size_x, block_size_y); fr.ok_luma2 &= ~fr_luma_bits(block_x, block_y, block_size_x, block_size_y); }
  if (fr_rec_cur() && sl.active && fr_asks_decided(block_x >> 2, block_y >> 2, (block_x + block_size_x) >> 2, (block_y + block_size_y) >> 2, p_dir, l0_mode, l0_ref_idx)) {
    int j;
    fr.ok_luma |= fr_luma_bits(block_x, block_y, block_size_x, block_size_y);
    for (j = block_y; j < block_y + block_size_y; j++)
      memcpy(&img->mpr[0][j][block_x], fr.pred[0] + (size_t)(img->pix_y + j) * g_w + img->pix_x + block_x, sizeof(imgpel) * (size_t)block_size_x);
    width_pad = listX[LIST_0][l0_ref_idx]->size_x_pad; height_pad = listX[LIST_0][l0_ref_idx]->size_y_pad;      /* OneComponentLumaPrediction, macroblock.c:817-818 */
    n_dev[S_LPRED]++;
    return;
  }
  if (fr_rec_cur() && sl.active && fr_asks_candidate(block_x >> 2, block_y >> 2, (block_x + block_size_x) >> 2, (block_y + block_size_y) >> 2, p_dir, l0_mode, l0_ref_idx)) {
    int j;
    fr.ok_luma2 |= fr_luma_bits(block_x, block_y, block_size_x, block_size_y);
    for (j = block_y; j < block_y + block_size_y; j++)
      memcpy(&img->mpr[0][j][block_x], fr.pred2 + (size_t)(img->pix_y + j) * g_w + img->pix_x + block_x, sizeof(imgpel) * (size_t)block_size_x);
    width_pad = listX[LIST_0][l0_ref_idx]->size_x_pad; height_pad = listX[LIST_0][l0_ref_idx]->size_y_pad;
    n_dev[S_LPRED]++;
    return;
  }
  if (!orig) orig = next_sym("LumaPrediction");
  if (fr_rec_cur()) n_fwd[S_LPRED]++;
  host_planes_lists();
  orig(currMB, block_x, block_y, block_size_x, block_size_y, p_dir, l0_mode, l1_mode, l0_ref_idx, l1_ref_idx);
}

void ChromaPrediction4x4(Macroblock *currMB, int uv, int block_x, int block_y, int p_dir, int l0_mode, int l1_mode, short l0_ref_idx, short l1_ref_idx)
{
  static void (*orig)(Macroblock *, int, int, int, int, int, int, short, short);
  if (fr_rec_cur()) { fr_ok_mb(); fr.ok_chroma[uv] &= ~(1u << ((block_y >> 2) * 2 + (block_x >> 2))); }
  /* 4:2:0: the chroma 4x4 block at (block_x, block_y) is predicted with the vectors of the luma 8x8 block at twice that position */
  if (fr_rec_cur() && sl.active && fr_asks_decided(block_x >> 1, block_y >> 1, (block_x >> 1) + 2, (block_y >> 1) + 2, p_dir, l0_mode, l0_ref_idx)) {
    int j;
    fr.ok_chroma[uv] |= 1u << ((block_y >> 2) * 2 + (block_x >> 2));
    for (j = block_y; j < block_y + 4; j++)
      memcpy(&img->mpr[uv + 1][j][block_x], fr.pred[uv + 1] + (size_t)(img->pix_c_y + j) * (g_w / 2) + img->pix_c_x + block_x, sizeof(imgpel) * 4);
    n_dev[S_CPRED]++;
    return;
  }
  if (!orig) orig = next_sym("ChromaPrediction4x4");
  if (fr_rec_cur()) n_fwd[S_CPRED]++;
  host_planes_lists();
  orig(currMB, uv, block_x, block_y, p_dir, l0_mode, l1_mode, l0_ref_idx, l1_ref_idx);
}

/* dct_4x4 (src/block.c:843) of an inter luma block, from the macroblock's record */
static int fr_dct4(Macroblock *currMB, int block_x, int block_y, int *coeff_cost, int *ret)
{
  const jmhip_mb_residual *r = fr_rec_cur();
  const imgpel *pred0 = fr.pred[0];
  if (!r || IS_INTRA(currMB) || currMB->luma_transform_size_8x8_flag) return 0;
  fr_ok_mb();
  {
    const unsigned bit = 1u << ((block_y >> 2) * 4 + (block_x >> 2));
    if (fr.ok_luma & bit) ;                                                      /* the decided block */
    else if (fr.ok_luma2 & bit) { r = &fr.rec2[img->current_mb_nr - fr.mb_first]; pred0 = fr.pred2; }      /* the P8x8 candidate's block */
    else { n_fwd[S_D4R]++; return 0; }                                           /* JM's own prediction, so JM's transform */
  }
  {
    const int pos_x = block_x >> 2, pos_y = block_y >> 2, b8 = 2 * (pos_y >> 1) + (pos_x >> 1), b4 = 2 * (pos_y & 1) + (pos_x & 1), blk = b8 * 4 + b4;
    const int n = r->cnt[blk];
    int *lev = img->cofAC[b8][b4][0], *run = img->cofAC[b8][b4][1];
    int **fa = img->AdaptiveRounding ? img->fadjust4x4[0] : NULL;
    imgpel **img_enc = enc_picture->p_curr_img;
    int j, i, k;
    for (j = block_y; j < block_y + 4; j++) for (i = block_x; i < block_x + 4; i++) {
      const int pr = pred0[(size_t)(img->pix_y + j) * g_w + img->pix_x + i];
      if (img->mpr[0][j][i] != pr || img->m7[0][j][i] != pCurImg[img->opix_y + j][img->opix_x + i] - pr) fr_diverged("luma", block_x, block_y);
    }
    for (k = 0; k < n; k++) { lev[k] = r->lev[blk][k]; run[k] = r->run[blk][k]; }
    lev[n] = 0;
    *coeff_cost += r->coeff_cost[blk];
    for (j = block_y; j < block_y + 4; j++) for (i = block_x; i < block_x + 4; i++) {
      img_enc[img->pix_y + j][img->pix_x + i] = r->recon_y[j][i];
      if (fa) fa[j][i] = r->fadj_y[j][i];
    }
    n_dev[S_D4R]++;
    *ret = (r->nonzero >> blk) & 1;
    return 1;
  }
}

/* dct_chroma (src/block.c:1051) of an inter macroblock's component, 4:2:0, from the record */
static int fr_dctc(Macroblock *currMB, int uv, int cr_cbp, int *ret)
{
  const jmhip_mb_residual *r = fr_rec_cur();
  if (!r || IS_INTRA(currMB)) return 0;
  fr_ok_mb();
  if (fr.ok_chroma[uv] != 15u) { n_fwd[S_DCRR]++; return 0; }
  {
    int **fa = img->AdaptiveRounding ? img->fadjust4x4Cr[0][uv] : NULL;
    const int wc = g_w / 2;
    int j, i, k, b, n = r->dc_cnt[uv];
    for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) {
      const int pr = fr.pred[uv + 1][(size_t)(img->pix_c_y + j) * wc + img->pix_c_x + i];
      if (img->mpr[uv + 1][j][i] != pr || img->m7[uv + 1][j][i] != imgUV_org[uv][img->opix_c_y + j][img->opix_c_x + i] - pr) fr_diverged(uv ? "Cr" : "Cb", i, j);
    }
    for (k = 0; k < n; k++) { img->cofDC[uv + 1][0][k] = r->dc_lev[uv][k]; img->cofDC[uv + 1][1][k] = r->dc_run[uv][k]; }
    img->cofDC[uv + 1][0][n] = 0;
    for (b = 0; b < 4; b++) {
      int *lev = img->cofAC[4 + uv][b][0], *run = img->cofAC[4 + uv][b][1];
      n = r->ac_zeroed[uv] ? 0 : r->cnt[16 + 4 * uv + b];              /* _CHROMA_COEFF_COST_ (:1384-1410): every AC level of the component reads 0 */
      for (k = 0; k < n; k++) { lev[k] = r->lev[16 + 4 * uv + b][k]; run[k] = r->run[16 + 4 * uv + b][k]; }
      lev[n] = 0;
    }
    currMB->cbp_blk = (currMB->cbp_blk & ~r->cbp_clear[uv]) | r->cbp_blk[uv];
    for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) {
      enc_picture->imgUV[uv][img->pix_c_y + j][img->pix_c_x + i] = r->recon_c[uv][j][i];
      if (fa && ((j | i) & 3)) fa[j][i] = r->fadj_c[uv][j][i];          /* AC positions only, block.c:1321-1380 */
    }
    n_dev[S_DCRR]++;
    *ret = cr_cbp > r->ret[uv] ? cr_cbp : r->ret[uv];
    return 1;
  }
}

static int slice_mode_covered(void)
{
  int m;
  if (!sl.decided) {
    sl.decided = 1;
    /* exact mode: the low-complexity decision without intra candidates is reproduced on the device, every call is answered.
       speculative mode (mask 0x2000, exhaustive searches and the simplified UMHexagonS -- pure functions of the predictor): whatever JM's decision
       is (RDOptimization 1 / 2, intra candidates, adaptive rounding), the device's low-complexity decision is the guess that keeps the predictors of
       the following macroblocks right most of the time */
    const int exact = input->rdopt == 0 && input->DisableIntraInInter && input->successive_Bframe == 0 && (input->Transform8x8Mode != 1 || !input->AdaptiveRounding);
    sl.spec = !exact && (shim_mask & 0x2000) && (input->SearchMode == -1 || input->SearchMode == 0 || input->SearchMode == 2);
    sl.on = (shim_mask & 0x1000) && (exact || sl.spec) && !input->PicInterlace &&
            !input->MbInterlace && !input->ChromaMEEnable && !input->DisableSubpelME &&
            /* Transform8x8Mode 1: the device quantises the 8x8-transform P8x8 pass to decide the partitioning (md_low.c:547): the slice's inter 8x8 tables
               as they stand when the slice begins (exact mode: no adaptive rounding) */
            (input->Transform8x8Mode == 0 || input->InterSearch[0][4]) &&
            (input->SearchMode == -1 || input->SearchMode == 0 || input->SearchMode == 1 || input->SearchMode == 2 || input->SearchMode == 3) &&
            !(input->SearchMode <= 0 && input->MEErrorMetric[F_PEL] != ERROR_SAD) && !input->EPZSSubPelGrid &&
            /* ranges up to 33; up to 40 where the exhaustive searches' sweeps over the frame kernels cover the configuration (me_xslice.hip) */
            (input->search_range <= 33 || (input->search_range <= 40 && input->SearchMode <= 0 && input->full_search == 2 && !input->Transform8x8Mode &&
                                           input->MEErrorMetric[H_PEL] == ERROR_SATD && input->MEErrorMetric[Q_PEL] == ERROR_SATD)) &&
            input->num_ref_frames <= JMHIP_SLICE_REFS && (input->slice_mode == 0 || input->slice_mode == 1) &&
            input->num_slice_groups_minus1 == 0 && !input->sp_periodicity && !input->BiPredMotionEstimation && input->InterSearch[0][1] &&
            !input->RestrictRef && !input->CtxAdptLagrangeMult && !input->RCEnable;
    for (m = 0; m < 3; m++) if (input->MEErrorMetric[m] != ERROR_SAD && input->MEErrorMetric[m] != ERROR_SSE && input->MEErrorMetric[m] != ERROR_SATD) sl.on = 0;
    if (input->ModeDecisionMetric != ERROR_SAD && input->ModeDecisionMetric != ERROR_SSE && input->ModeDecisionMetric != ERROR_SATD) sl.on = 0;
  }
  return sl.on;
}

/* one device call for the slice that begins at the current macroblock */
static void slice_run(int *lambda_factor)
{
  jmhip_slice_params p;
  const int nmb = (int)img->PicSizeInMbs, first = img->current_mb_nr;
  int r, m, count = nmb - first, pocs[JMHIP_SLICE_REFS];
  const double t0 = now_s();
  memset(&p, 0, sizeof(p));
  /* fixed-size slices: all remaining slices of the picture in this one call (slice_mbs) */
  sl.multi = input->slice_mode == 1 && input->slice_argument < count;
  if (sl.multi) p.slice_mbs = input->slice_argument;
  else if (input->slice_mode == 1 && input->slice_argument < count) count = input->slice_argument;
  p.search_mode = input->SearchMode; p.search_range = input->search_range; p.full_search = input->full_search; p.num_refs = listXsize[LIST_0];
  for (r = 0; r < listXsize[LIST_0]; r++) {
    p.ref_slot[r] = slot_find(listX[LIST_0][r]);
    if (p.ref_slot[r] < 0) { fprintf(stderr, "jm_shim: slice binding: reference %d has no device slot\n", r); exit(95); }
    pocs[r] = listX[LIST_0][r]->poc;
    if (active_pps->weighted_pred_flag) { p.wp_weight[r] = (int16_t)wp_weight[0][r][0]; p.wp_offset[r] = (int16_t)wp_offset[0][r][0]; }   /* allocated only with WP (lencod.c:2063) */
  }
  for (m = 1; m < 8; m++) p.valid[m] = input->InterSearch[0][m] != 0;
  if (input->Transform8x8Mode == 2) p.valid[5] = p.valid[6] = p.valid[7] = 0;
  for (m = 0; m < 3; m++) { p.lambda_mf[m] = lambda_factor[m]; p.metric[m] = input->MEErrorMetric[m]; }
  p.ref_cost1 = (int)(2 * img->lambda_me[img->type][img->qp][Q_PEL]);
  p.md_metric = input->ModeDecisionMetric;
  p.level_mv_min = LEVELMVLIMIT[img->LevelIndex][0]; p.level_mv_max = LEVELMVLIMIT[img->LevelIndex][1];
  p.wp_pred = active_pps->weighted_pred_flag != 0; p.wp_me = p.wp_pred && input->UseWeightedReferenceME;
  p.wp_round = wp_luma_round; p.wp_denom = luma_log_weight_denom;
  p.mb_first = first; p.mb_count = count;
  p.rdopt = input->rdopt;
  p.transform8x8_mode = input->Transform8x8Mode;
  if (input->Transform8x8Mode == 1) {                  /* the inter 8x8 luma quantiser (transform8x8.c:1487-1489) of the slice's macroblocks (no rate control: one qp) */
    const Macroblock *mb = &img->mb_data[img->current_mb_nr];
    const int qp = mb->qp_scaled[0], qp_rem = qp_rem_matrix[qp];
    int i, j;
    p.t8_qp = qp; p.t8_cavlc = (input->symbol_mode == CAVLC); p.t8_disthres = input->disthres;
    for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) {
      p.t8_levelscale[j * 8 + i] = LevelScale8x8Comp[0][0][qp_rem][j][i]; p.t8_leveloffset[j * 8 + i] = LevelOffset8x8Comp[0][0][qp][j][i];
    }
  }
  if (!sl.started) { OK(jmhip_slice_state_reset(g)); sl.started = 1; }
  if (input->SearchMode == 3) {
    jmhip_epzs_setup(&p, input->search_range, input->EPZSPattern, input->EPZSDual, input->EPZSFixed, input->EPZSTemporal, input->EPZSSpatialMem,
                     input->EPZSSubPelME, input->EPZSMinThresScale, input->EPZSMedThresScale, input->EPZSMaxThresScale, input->EPZSSubPelThresScale);
    jmhip_epzs_scales(&p, enc_picture->poc, pocs, listXsize[LIST_0]);
    if (input->EPZSTemporal) {                          /* the scaled co-located field JM's EPZSSliceInit just built (src/me_epzs.c:986-1030) */
      const int w4 = img->width / 4, h4 = img->height / 4;
      int16_t *col = malloc(sizeof(int16_t) * 2 * w4 * h4);
      int i, j;
      for (j = 0; j < h4; j++) for (i = 0; i < w4; i++) { col[(j * w4 + i) * 2] = EPZSCo_located->mv[LIST_0][j][i][0]; col[(j * w4 + i) * 2 + 1] = EPZSCo_located->mv[LIST_0][j][i][1]; }
      OK(jmhip_epzs_colocated_upload(g, col));
      free(col);
    }
  }
  if (input->SearchMode == 1) jmhip_umhex_setup(&p, input->UMHexDSR, input->UMHexScale, input->qpN, img->width);
  if (sl.cap < count) { free(sl.rec); sl.rec = malloc(sizeof(jmhip_mb_inter) * (size_t)count); sl.cap = count; }
  OK(jmhip_p_slice_search(g, &p, sl.rec));
  { int n = 0; jmhip_slice_result_info(g, &n); sl.passes += n; }
  sl.slices++;
  sl.active = 1; sl.serial = pic_serial; sl.slice_nr = img->current_slice_nr; sl.mb_first = first; sl.mb_count = count;
  n_dev[S_SLICE]++; t_last[S_SLICE] = now_s() - t0; t_dev[S_SLICE] += t_last[S_SLICE];
  fr_run(&p, first, count);
}

static long sl_slices(void) { return sl.slices; }
static long sl_passes(void) { return sl.passes; }

int BlockMotionSearch(short ref, int list, int mb_x, int mb_y, int blocktype, int search_range, int *lambda_factor)
{
  static int (*orig)(short, int, int, int, int, int, int *);
  int covered = slice_mode_covered() && img->type == P_SLICE && list == 0 && img->structure == FRAME && !img->MbaffFrameFlag && ctx_ready() && cur_ready() &&
                listXsize[LIST_0] <= JMHIP_SLICE_REFS;
  if (!covered) {
    if (!orig) orig = next_sym("BlockMotionSearch");
    host_planes_lists();
    n_fwd[S_BMS]++;
    return orig(ref, list, mb_x, mb_y, blocktype, search_range, lambda_factor);
  }
  if (!(sl.active && sl.serial == pic_serial && (sl.slice_nr == img->current_slice_nr || sl.multi) && img->current_mb_nr >= sl.mb_first && img->current_mb_nr < sl.mb_first + sl.mb_count))
    slice_run(lambda_factor);
  {
    const jmhip_mb_inter *r = &sl.rec[img->current_mb_nr - sl.mb_first];
    const int p = partition_of(blocktype, mb_x, mb_y), bx = mb_x >> 2, by = mb_y >> 2, bsx = input->blc_size[blocktype][0], bsy = input->blc_size[blocktype][1];
    short pmv[2], *pred_mv = img->pred_mv[by][bx][list][ref][blocktype];
    int i, j;
    /* Transform8x8Mode: JM searches the 8x8 blocks in the 8x8-transform P8x8 pass first (md_low.c:203-228) and, mode 1, again in the 4x4-transform
       pass: the first call per (macroblock, reference, block) is answered from the record of that pass */
    static int t8_mb = -1; static unsigned long t8_serial; static unsigned char t8_seen[JMHIP_SLICE_REFS][4];
    int first8 = 0;
    const int16_t *rpred, *rmv; int rcost;
    if (t8_mb != img->current_mb_nr || t8_serial != pic_serial) { t8_mb = img->current_mb_nr; t8_serial = pic_serial; memset(t8_seen, 0, sizeof(t8_seen)); }
    if (input->Transform8x8Mode && blocktype == 4 && p >= 5 && !t8_seen[ref][p - 5]) { first8 = 1; t8_seen[ref][p - 5] = 1; }
    rpred = (p < 0) ? NULL : first8 ? r->pred8ts[ref][p - 5] : r->pred[ref][p];
    rmv = (p < 0) ? NULL : first8 ? r->mv8ts[ref][p - 5] : r->mv[ref][p];
    rcost = (p < 0) ? 0 : first8 ? r->cost8ts[ref][p - 5] : r->cost[ref][p];
    /* the device predicted from ITS picture arrays; JM predicts from enc_picture: they must agree, or the slice has diverged */
    SetMotionVectorPredictor(pmv, enc_picture->ref_idx[list], enc_picture->mv[list], ref, list, bx, by, bsx, bsy);
    if (sl.spec) {
      /* speculative mode: a record is JM's answer exactly when the call's predictor is the record's and nothing the search reads beside it differs:
         FastFullSearch -- the window centre, i.e. the 16x16 call of this (macroblock, reference) matched too; simplified UMHexagonS -- the upper
         layer's vector, i.e. no earlier call of this (macroblock, reference) went to JM */
      static int sp_mb = -1; static unsigned long sp_serial; static unsigned char sp_ok16[JMHIP_SLICE_REFS], sp_dirty[JMHIP_SLICE_REFS];
      int hit;
      if (sp_mb != img->current_mb_nr || sp_serial != pic_serial) { sp_mb = img->current_mb_nr; sp_serial = pic_serial; memset(sp_ok16, 0, sizeof(sp_ok16)); memset(sp_dirty, 0, sizeof(sp_dirty)); }
      hit = p >= 0 && pmv[0] == rpred[0] && pmv[1] == rpred[1];
      if (hit && input->SearchMode == 0 && blocktype != 1 && !sp_ok16[ref]) hit = 0;
      if (hit && input->SearchMode == 2 && sp_dirty[ref]) hit = 0;
      if (blocktype == 1) sp_ok16[ref] = (unsigned char)hit;          /* the window centre of FastFullSearch follows from the 16x16 predictor alone */
      if (hit && blocktype == 1 && !input->rdopt) {
        /* RDOptimization 0: the 16x16 record went through the skip shortcut (src/mv-search.c:826-849), whose vector depends on whether the
           neighbours A and B are zero-vector references to picture 0 (FindSkipModeMotionVector :1189) -- not on this call's predictor alone.
           The record is JM's answer only if the device's skip vector is the one JM finds now (JM's own call computes it at :829 anyway). */
        FindSkipModeMotionVector(&img->mb_data[img->current_mb_nr]);
        if (img->all_mv[0][0][0][0][0][0] != r->skip_mv[0] || img->all_mv[0][0][0][0][0][1] != r->skip_mv[1]) hit = 0;
      }
      if (!hit) {
        sp_dirty[ref] = 1;
        if (!orig) orig = next_sym("BlockMotionSearch");
    host_planes_lists();
        n_fwd[S_BMS]++;
        return orig(ref, list, mb_x, mb_y, blocktype, search_range, lambda_factor);
      }
    } else
    if (p < 0 || pmv[0] != rpred[0] || pmv[1] != rpred[1]) {
      fprintf(stderr, "jm_shim: slice binding diverged at mb %d ref %d blocktype %d block (%d,%d)%s: JM predictor (%d,%d), device (%d,%d)\n",
              img->current_mb_nr, ref, blocktype, bx, by, first8 ? " [8x8-transform pass]" : "", pmv[0], pmv[1], p < 0 ? 0 : rpred[0], p < 0 ? 0 : rpred[1]);
      exit(95);
    }
    pred_mv[0] = pmv[0]; pred_mv[1] = pmv[1];
    for (j = by; j < by + (bsy >> 2); j++) for (i = bx; i < bx + (bsx >> 2); i++) {
      img->all_mv[j][i][list][ref][blocktype][0] = rmv[0]; img->all_mv[j][i][list][ref][blocktype][1] = rmv[1];
    }
    /* what BlockMotionSearch leaves behind besides its results (src/mv-search.c:612, :640, :779, :837) */
    ChromaMEEnable = 0;
    test8x8transform = input->Transform8x8Mode && blocktype <= 4;
    jm_side_effects(listX[LIST_0][ref]);
    if (blocktype == 1) FindSkipModeMotionVector(&img->mb_data[img->current_mb_nr]);
    if (verify) {
      short ax = img->all_mv[by][bx][list][ref][blocktype][0], ay = img->all_mv[by][bx][list][ref][blocktype][1];
      int c;
      if (!orig) orig = next_sym("BlockMotionSearch");
    host_planes_lists();
      c = orig(ref, list, mb_x, mb_y, blocktype, search_range, lambda_factor);
      if (c != rcost || img->all_mv[by][bx][list][ref][blocktype][0] != ax || img->all_mv[by][bx][list][ref][blocktype][1] != ay)
        fprintf(stderr, "jm_shim VERIFY BlockMotionSearch mb %d ref %d bt %d (%d,%d): jm=(%d,%d,%d) dev=(%d,%d,%d)\n", img->current_mb_nr, ref, blocktype, bx, by,
                img->all_mv[by][bx][list][ref][blocktype][0], img->all_mv[by][bx][list][ref][blocktype][1], c, ax, ay, rcost);
      return c;
    }
    n_dev[S_BMS]++;
    return rcost;
  }
}

/* DeblockFrame (src/loopFilter.c:87): the picture goes up, jmhip_deblock_frame filters it in JM's macroblock order, it comes back.
 * MBAFF, field pictures and SP/SI slices stay in JM. */
void DeblockFrame(ImageParameters *im, imgpel **imgY, imgpel ***imgUV)
{
  static void (*orig)(ImageParameters *, imgpel **, imgpel ***);
  static jmhip_deblock_mb *mbs;
  static jmhip_deblock_blk *blks;
  const int W = im->width, H = im->height, nmb = (int)im->PicSizeInMbs, w4 = W / 4, h4 = H / 4;
  const int chroma = imgUV && im->yuv_format != YUV400;
  int i, l, x, y;
  double t0;
  int ok = (shim_mask & 0x800) && !im->MbaffFrameFlag && im->structure == FRAME && im->type != SP_SLICE && im->type != SI_SLICE &&
           imgUV && !(im->yuv_format == YUV444 && IS_INDEPENDENT(input)) && ctx_ready() && W == g_w && H == g_h &&
           imgY[1] == imgY[0] + W && (!chroma || imgUV[0][1] == imgUV[0][0] + im->width_cr);
  if (!ok) {
    n_fwd[S_DEBLOCK]++;
    if (!orig) orig = next_sym("DeblockFrame");
    orig(im, imgY, imgUV);
    return;
  }
  n_dev[S_DEBLOCK]++;
  t0 = now_s();
  if (!mbs) {
    mbs = malloc(sizeof(*mbs) * (size_t)nmb);
    blks = malloc(sizeof(*blks) * (size_t)w4 * h4);
    if (!mbs || !blks) { fprintf(stderr, "jm_shim: out of memory\n"); exit(96); }
  }
  for (i = 0; i < nmb; i++) {
    Macroblock *m = &im->mb_data[i];
    if (m->mb_type == IPCM) { m->qp = 0; m->qpc[0] = 0; m->qpc[1] = 0; }          /* loopFilter.c:105-113: a side effect JM keeps */
    mbs[i].intra = m->mb_type == I4MB || m->mb_type == I8MB || m->mb_type == I16MB || m->mb_type == IPCM;
    mbs[i].qp = (uint8_t)m->qp; mbs[i].qpc[0] = (uint8_t)m->qpc[0]; mbs[i].qpc[1] = (uint8_t)m->qpc[1];
    mbs[i].disable_idc = (uint8_t)m->LFDisableIdc;
    mbs[i].alpha_c0_offset = (int8_t)m->LFAlphaC0Offset; mbs[i].beta_offset = (int8_t)m->LFBetaOffset;
    mbs[i].transform_8x8 = (uint8_t)m->luma_transform_size_8x8_flag;
    mbs[i].avail_a = (uint8_t)m->mbAvailA; mbs[i].avail_b = (uint8_t)m->mbAvailB;
    mbs[i].cbp_blk = (uint16_t)(m->cbp_blk & 0xffff);
  }
  for (y = 0; y < h4; y++) for (x = 0; x < w4; x++) {
    jmhip_deblock_blk *b = &blks[y * w4 + x];
    for (l = 0; l < 2; l++) {
      b->mv[l][0] = enc_picture->mv[l][y][x][0]; b->mv[l][1] = enc_picture->mv[l][y][x][1];
      b->ref_id[l] = enc_picture->ref_idx[l][y][x] < 0 ? INT64_MIN : enc_picture->ref_pic_id[l][y][x];
    }
  }
  OK(jmhip_recon_upload(g, imgY[0], chroma ? imgUV[0][0] : NULL, chroma ? imgUV[1][0] : NULL, 2));
  OK(jmhip_deblock_frame(g, mbs, blks, 4, 0, 0));
  OK(jmhip_recon_download(g, imgY[0], chroma ? imgUV[0][0] : NULL, chroma ? imgUV[1][0] : NULL, 2));
  im->current_mb_nr = nmb - 1;                                                     /* where DeblockMb :178 leaves it */
  t_last[S_DEBLOCK] = now_s() - t0; t_dev[S_DEBLOCK] += t_last[S_DEBLOCK];
}
