/*
 * tap_capture.c -- TEST INFRASTRUCTURE, container only (needs /root/reference headers at build time).
 *
 * Golden-vector capture: links the real JM (oracle/_ref/libjm.so), DEFINES the hot-path symbols, forwards every call to
 * JM's original (dlsym RTLD_NEXT) and records sampled calls -- explicit arguments, the implicit globals the function
 * reads, and its outputs -- as a stream of int32 records in $JM_TAP_OUT. tests/golden/make_golden.py runs it on the
 * reference's own clips and packs the stream into the committed .npz fixtures. Only DATA leaves: no reference text.
 *
 * Record = { magic 0x4a4d5450, kind, n_ints, payload[n_ints] }.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "global.h"
#include "mbuffer.h"
#include "image.h"
#include "refbuf.h"
#include "me_distortion.h"
#include "q_matrix.h"
#include "q_offsets.h"

extern int jm_main(int argc, char **argv);
extern int ****ptLevelOffset4x4;

enum { K_LUMA = 1, K_CHROMA = 2, K_FULLPEL = 3, K_SUBPEL = 4, K_FASTFULL = 5, K_DCT4 = 6, K_DCT8 = 7, K_DCT16 = 8, K_DCTC = 9 };
static FILE *fo;
static long seen[16], kept[16];
static int every[16] = { 0, 1, 1, 37, 37, 53, 211, 101, 7, 29 };   /* keep one call in N */
static int cap[16]   = { 0, 4, 4, 400, 400, 400, 300, 200, 60, 200 };

static void *next_sym(const char *n) { void *p = dlsym(RTLD_NEXT, n); if (!p) { fprintf(stderr, "tap: no %s\n", n); exit(97); } return p; }
static int want(int k) { seen[k]++; if (kept[k] >= cap[k] || (seen[k] % every[k])) return 0; kept[k]++; return 1; }

static int *buf; static int nbuf, cbuf;
static void b_reset(void) { nbuf = 0; }
static void b_put(int v) { if (nbuf == cbuf) { cbuf = cbuf ? 2 * cbuf : 1 << 16; buf = realloc(buf, sizeof(int) * cbuf); } buf[nbuf++] = v; }
static void b_flush(int kind) { int h[3] = { 0x4a4d5450, kind, nbuf }; fwrite(h, 4, 3, fo); fwrite(buf, 4, nbuf, fo); }

int main(int argc, char **argv)
{
  const char *o = getenv("JM_TAP_OUT");
  fo = fopen(o ? o : "jm_tap.bin", "wb");
  if (!fo) return 96;
  int rc = jm_main(argc, argv);
  fclose(fo);
  return rc;
}

static int pic_id(StorablePicture *s)          /* stable small id per stored picture (by first-seen order) */
{
  static StorablePicture *tab[64]; static int n;
  for (int i = 0; i < n; i++) if (tab[i] == s) return i;
  if (n < 64) tab[n++] = s;
  return n - 1;
}
/* FNV-1a over 16-bit samples: cheap, order-sensitive plane digest */
static unsigned digest(imgpel **rows, int h, int w)
{
  unsigned d = 2166136261u;
  for (int j = 0; j < h; j++) for (int i = 0; i < w; i++) { d ^= rows[j][i]; d *= 16777619u; }
  return d;
}

void getSubImagesLuma(StorablePicture *s)
{
  static void (*orig)(StorablePicture *);
  if (!orig) orig = next_sym("getSubImagesLuma");
  orig(s);
  /* every reference picture is recorded in full (the search records replay against it); digests of the 16 planes */
  b_reset();
  b_put(pic_id(s)); b_put(s->size_x); b_put(s->size_y); b_put(img->max_imgpel_value);
  for (int j = 0; j < s->size_y; j++) for (int i = 0; i < s->size_x; i++) b_put(s->p_curr_img[j][i]);
  for (int p = 0; p < 16; p++) b_put((int)digest(s->p_curr_img_sub[p >> 2][p & 3], s->size_y_padded, s->size_x_padded));
  /* a few explicit rows of the diagonal-most planes incl. the ring */
  for (int p = 0; p < 16; p += 5) for (int j = 0; j < 3; j++) for (int i = 0; i < s->size_x_padded; i++) b_put(s->p_curr_img_sub[p >> 2][p & 3][j * 17 % s->size_y_padded][i]);
  b_flush(K_LUMA);
}

void getSubImagesChroma(StorablePicture *s)
{
  static void (*orig)(StorablePicture *);
  if (!orig) orig = next_sym("getSubImagesChroma");
  orig(s);
  int subx = img->yuv_format == YUV444 ? 4 : 8, suby = img->yuv_format == YUV420 ? 8 : 4;
  int Wcp = s->size_x_cr + 2 * img_pad_size_uv_x, Hcp = s->size_y_cr + 2 * img_pad_size_uv_y;
  b_reset();
  b_put(pic_id(s)); b_put(s->size_x_cr); b_put(s->size_y_cr); b_put(img->yuv_format);
  for (int uv = 0; uv < 2; uv++) {
    for (int j = 0; j < s->size_y_cr; j++) for (int i = 0; i < s->size_x_cr; i++) b_put(s->imgUV[uv][j][i]);
    for (int y = 0; y < suby; y++) for (int x = 0; x < subx; x++) b_put((int)digest(s->imgUV_sub[uv][y][x], Hcp, Wcp));
  }
  b_flush(K_CHROMA);
}

static void put_me_globals(int list, int ref, StorablePicture *rp)
{
  int lo = img->mb_data[img->current_mb_nr].list_offset;
  int aw = ((active_pps->weighted_pred_flag && (img->type == P_SLICE || img->type == SP_SLICE)) ||
            (active_pps->weighted_bipred_idc && (img->type == B_SLICE))) && input->UseWeightedReferenceME;
  b_put(pic_id(rp)); b_put(input->rdopt); b_put(img->type == B_SLICE); b_put(input->ChromaMEEnable);
  b_put(input->Transform8x8Mode); b_put(input->MEErrorMetric[0]); b_put(input->MEErrorMetric[1]); b_put(input->MEErrorMetric[2]);
  b_put(aw); b_put(aw ? wp_weight[list + lo][ref][0] : 0); b_put(aw ? wp_offset[list + lo][ref][0] : 0);
  b_put(wp_luma_round); b_put(luma_log_weight_denom); b_put(img->yuv_format);
}

int FullPelBlockMotionSearch(imgpel *orig_pic, short ref, int list, int pic_pix_x, int pic_pix_y, int blocktype,
                             short pred_mv_x, short pred_mv_y, short *mv_x, short *mv_y, int search_range, int min_mcost, int lambda_factor)
{
  static int (*orig)(imgpel *, short, int, int, int, int, short, short, short *, short *, int, int, int);
  if (!orig) orig = next_sym("FullPelBlockMotionSearch");
  int keep = !input->ChromaMEEnable && want(K_FULLPEL);
  short ix = *mv_x, iy = *mv_y;
  int r = orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, mv_x, mv_y, search_range, min_mcost, lambda_factor);
  if (keep) {
    StorablePicture *rp = listX[list + img->mb_data[img->current_mb_nr].list_offset][ref];
    int bsx = input->blc_size[blocktype][0], bsy = input->blc_size[blocktype][1];
    b_reset();
    put_me_globals(list, ref, rp);
    b_put(ref == 0); b_put(pic_pix_x); b_put(pic_pix_y); b_put(blocktype); b_put(pred_mv_x); b_put(pred_mv_y);
    b_put(ix); b_put(iy); b_put(search_range); b_put(min_mcost); b_put(lambda_factor);
    b_put(*mv_x); b_put(*mv_y); b_put(r);
    for (int k = 0; k < bsx * bsy; k++) b_put(orig_pic[k]);
    b_flush(K_FULLPEL);
  }
  return r;
}

int SubPelBlockMotionSearch(imgpel *orig_pic, short ref, int list, int pic_pix_x, int pic_pix_y, int blocktype,
                            short pred_mv_x, short pred_mv_y, short *mv_x, short *mv_y, int search_pos2, int search_pos4, int min_mcost, int *lambda)
{
  static int (*orig)(imgpel *, short, int, int, int, int, short, short, short *, short *, int, int, int, int *);
  if (!orig) orig = next_sym("SubPelBlockMotionSearch");
  int keep = !input->ChromaMEEnable && want(K_SUBPEL);
  short ix = *mv_x, iy = *mv_y;
  int r = orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, mv_x, mv_y, search_pos2, search_pos4, min_mcost, lambda);
  if (keep) {
    StorablePicture *rp = listX[list + img->mb_data[img->current_mb_nr].list_offset][ref];
    int bsx = input->blc_size[blocktype][0], bsy = input->blc_size[blocktype][1];
    b_reset();
    put_me_globals(list, ref, rp);
    b_put(ref == 0); b_put(pic_pix_x); b_put(pic_pix_y); b_put(blocktype); b_put(pred_mv_x); b_put(pred_mv_y);
    b_put(ix); b_put(iy); b_put(search_pos2); b_put(search_pos4); b_put(min_mcost); b_put(lambda[0]); b_put(lambda[1]); b_put(lambda[2]);
    b_put(*mv_x); b_put(*mv_y); b_put(r);
    for (int k = 0; k < bsx * bsy; k++) b_put(orig_pic[k]);
    b_flush(K_SUBPEL);
  }
  return r;
}

extern void SetMotionVectorPredictor(short pmv[2], char **refPic, short ***tmp_mv, short ref_frame, int list,
                                     int block_x, int block_y, int blockshape_x, int blockshape_y);
int FastFullPelBlockMotionSearch(imgpel *orig_pic, short ref, int list, int pic_pix_x, int pic_pix_y, int blocktype,
                                 short pred_mv_x, short pred_mv_y, short *mv_x, short *mv_y, int search_range, int min_mcost, int lambda_factor)
{
  static int (*orig)(imgpel *, short, int, int, int, int, short, short, short *, short *, int, int, int);
  if (!orig) orig = next_sym("FastFullPelBlockMotionSearch");
  int keep = !input->ChromaMEEnable && want(K_FASTFULL);
  short pmv[2] = { 0, 0 };
  /* the 16x16 predictor JM uses for the window centre (me_fullfast.c:550); valid only at the first call of the MB/ref,
   * so it is re-derived the same way JM does it -- the neighbours have not changed within the macroblock's searches of
   * block type 1, which is what the record keeps */
  if (keep && blocktype == 1) SetMotionVectorPredictor(pmv, enc_picture->ref_idx[list], enc_picture->mv[list], ref, list, 0, 0, 16, 16);
  int r = orig(orig_pic, ref, list, pic_pix_x, pic_pix_y, blocktype, pred_mv_x, pred_mv_y, mv_x, mv_y, search_range, min_mcost, lambda_factor);
  if (keep && blocktype == 1) {
    StorablePicture *rp = listX[list + img->mb_data[img->current_mb_nr].list_offset][ref];
    int R = (input->full_search == 2 || ref == 0) ? input->search_range : input->search_range / 2;
    b_reset();
    put_me_globals(list, ref, rp);
    b_put(ref == 0); b_put(img->opix_x); b_put(img->opix_y); b_put(pic_pix_x); b_put(pic_pix_y); b_put(blocktype);
    b_put(pmv[0]); b_put(pmv[1]); b_put(pred_mv_x); b_put(pred_mv_y); b_put(R); b_put(min_mcost); b_put(lambda_factor);
    b_put(*mv_x); b_put(*mv_y); b_put(r);
    for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) b_put(pCurImg[img->opix_y + y][img->opix_x + x]);
    b_flush(K_FASTFULL);
  } else if (keep) kept[K_FASTFULL]--;
  return r;
}

static void put_tab(int **t, int n) { for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) b_put(t[j][i]); }
static void put_tile_int(int (*t)[16]) { for (int j = 0; j < 16; j++) for (int i = 0; i < 16; i++) b_put(t[j][i]); }
static void put_tile_pel(imgpel (*t)[16]) { for (int j = 0; j < 16; j++) for (int i = 0; i < 16; i++) b_put(t[j][i]); }
static void put_quant(Macroblock *mb, int qp, int **ls, int **ils, int **lo, int n, int weight, int maxv)
{
  b_put(qp); b_put(img->AdaptiveRounding); b_put(weight); b_put(mb->is_field_mode); b_put(input->disthres); b_put(maxv);
  b_put(input->symbol_mode == CAVLC); b_put(img->qp); b_put(mb->luma_transform_size_8x8_flag);
  put_tab(ls, n); put_tab(ils, n); put_tab(lo, n);
}
static void put_recon(imgpel **rows, int y0, int x0, int h, int w) { for (int j = 0; j < h; j++) for (int i = 0; i < w; i++) b_put(rows[y0 + j][x0 + i]); }
static void put_fadj(int **fa, int h, int w) { for (int j = 0; j < h; j++) for (int i = 0; i < w; i++) b_put(fa ? fa[j][i] : 0); }

int dct_4x4(Macroblock *currMB, ColorPlane pl, int block_x, int block_y, int *coeff_cost, int intra)
{
  static int (*orig)(Macroblock *, ColorPlane, int, int, int *, int);
  if (!orig) orig = next_sym("dct_4x4");
  int lossless = (currMB->qp_scaled[pl] == 0 && img->lossless_qpprime_flag == 1);
  int keep = !lossless && pl == 0 && want(K_DCT4);
  int qp = currMB->qp_scaled[pl], cc_in = *coeff_cost;
  if (keep) {
    b_reset();
    b_put(block_x); b_put(block_y); b_put(intra); b_put(cc_in);
    put_quant(currMB, qp, LevelScale4x4Comp[pl][intra][qp_rem_matrix[qp]], InvLevelScale4x4Comp[pl][intra][qp_rem_matrix[qp]],
              ptLevelOffset4x4[intra][qp], 4, AdaptRndWeight, img->max_imgpel_value);
    put_tile_int(img->m7[pl]); put_tile_pel(img->mpr[pl]);
  }
  int r = orig(currMB, pl, block_x, block_y, coeff_cost, intra);
  if (keep) {
    int b8 = 2 * (block_y >> 3) + (block_x >> 3), b4 = 2 * ((block_y >> 2) & 1) + ((block_x >> 2) & 1);
    b_put(r); b_put(*coeff_cost);
    for (int k = 0; k < 17; k++) b_put(img->cofAC[b8][b4][0][k]);
    for (int k = 0; k < 17; k++) b_put(img->cofAC[b8][b4][1][k]);
    put_recon(enc_picture->p_curr_img, img->pix_y + block_y, img->pix_x + block_x, 4, 4);
    put_fadj(img->AdaptiveRounding ? img->fadjust4x4[intra] : NULL, 16, 16);
    b_flush(K_DCT4);
  }
  return r;
}

int dct_8x8(Macroblock *currMB, ColorPlane pl, int b8, int *coeff_cost, int intra)
{
  static int (*orig)(Macroblock *, ColorPlane, int, int *, int);
  if (!orig) orig = next_sym("dct_8x8");
  int lossless = (currMB->qp_scaled[pl] == 0 && img->lossless_qpprime_flag == 1);
  int keep = !lossless && pl == 0 && want(K_DCT8);
  int qp = currMB->qp_scaled[pl], cc_in = *coeff_cost;
  if (keep) {
    b_reset();
    b_put(b8); b_put(intra); b_put(cc_in);
    put_quant(currMB, qp, LevelScale8x8Comp[pl][intra][qp_rem_matrix[qp]], InvLevelScale8x8Comp[pl][intra][qp_rem_matrix[qp]],
              LevelOffset8x8Comp[pl][intra][qp], 8, AdaptRndWeight, img->max_imgpel_value);
    put_tile_int(img->m7[pl]); put_tile_pel(img->mpr[pl]);
  }
  int r = orig(currMB, pl, b8, coeff_cost, intra);
  if (keep) {
    b_put(r); b_put(*coeff_cost);
    for (int q = 0; q < 4; q++) { for (int k = 0; k < 65; k++) b_put(img->cofAC[b8][q][0][k]); for (int k = 0; k < 65; k++) b_put(img->cofAC[b8][q][1][k]); }
    put_recon(enc_picture->p_curr_img, img->pix_y + 8 * (b8 >> 1), img->pix_x + 8 * (b8 & 1), 8, 8);
    put_fadj(img->AdaptiveRounding ? img->fadjust8x8[intra] : NULL, 16, 16);
    b_flush(K_DCT8);
  }
  return r;
}

int dct_16x16(Macroblock *currMB, ColorPlane pl, int new_intra_mode)
{
  static int (*orig)(Macroblock *, ColorPlane, int);
  if (!orig) orig = next_sym("dct_16x16");
  int lossless = (currMB->qp_scaled[pl] == 0 && img->lossless_qpprime_flag == 1);
  int keep = !lossless && pl == 0 && img->type != SP_SLICE && want(K_DCT16);
  int qp = currMB->qp_scaled[pl];
  if (keep) {
    b_reset();
    b_put(new_intra_mode);
    put_quant(currMB, qp, LevelScale4x4Comp[pl][1][qp_rem_matrix[qp]], InvLevelScale4x4Comp[pl][1][qp_rem_matrix[qp]],
              ptLevelOffset4x4[1][qp], 4, AdaptRndWeight, img->max_imgpel_value);
    for (int j = 0; j < 16; j++) for (int i = 0; i < 16; i++) b_put(pCurImg[img->opix_y + j][img->opix_x + i]);
    put_tile_pel(img->mpr_16x16[pl][new_intra_mode]);
  }
  int r = orig(currMB, pl, new_intra_mode);
  if (keep) {
    b_put(r);
    for (int k = 0; k < 17; k++) b_put(img->cofDC[pl][0][k]);
    for (int k = 0; k < 17; k++) b_put(img->cofDC[pl][1][k]);
    for (int b = 0; b < 16; b++) { for (int k = 0; k < 16; k++) b_put(img->cofAC[b >> 2][b & 3][0][k]); for (int k = 0; k < 16; k++) b_put(img->cofAC[b >> 2][b & 3][1][k]); }
    put_recon(enc_picture->p_curr_img, img->pix_y, img->pix_x, 16, 16);
    put_fadj(img->AdaptiveRounding ? img->fadjust4x4[2] : NULL, 16, 16);
    b_flush(K_DCT16);
  }
  return r;
}

int dct_chroma(Macroblock *currMB, int uv, int cr_cbp)
{
  static int (*orig)(Macroblock *, int, int);
  if (!orig) orig = next_sym("dct_chroma");
  int lossless = ((currMB->qp + img->bitdepth_luma_qp_scale) == 0 && img->lossless_qpprime_flag == 1);
  int keep = !lossless && img->yuv_format != YUV444 && want(K_DCTC);
  int intra = IS_INTRA(currMB);
  int qp = currMB->qpc[uv] + img->bitdepth_chroma_qp_scale, qpdc = qp + 3;
  long long cbp_in = currMB->cbp_blk;
  if (keep) {
    b_reset();
    b_put(uv); b_put(cr_cbp); b_put(img->yuv_format); b_put((int)(cbp_in & 0xffffffff)); b_put((int)(cbp_in >> 32));
    put_quant(currMB, qp, LevelScale4x4Comp[uv + 1][intra][qp_rem_matrix[qp]], InvLevelScale4x4Comp[uv + 1][intra][qp_rem_matrix[qp]],
              LevelOffset4x4Comp[uv + 1][intra][qp], 4, AdaptRndCrWeight, img->max_imgpel_value_comp[1]);
    put_quant(currMB, qpdc, LevelScale4x4Comp[uv + 1][intra][qp_rem_matrix[qpdc]], InvLevelScale4x4Comp[uv + 1][intra][qp_rem_matrix[qpdc]],
              LevelOffset4x4Comp[uv + 1][intra][qpdc], 4, AdaptRndCrWeight, img->max_imgpel_value_comp[1]);
    put_tile_int(img->m7[uv + 1]); put_tile_pel(img->mpr[uv + 1]);
  }
  int r = orig(currMB, uv, cr_cbp);
  if (keep) {
    int nb8 = img->num_blk8x8_uv >> 1, uvs = uv * nb8;
    b_put(r); b_put((int)(currMB->cbp_blk & 0xffffffff)); b_put((int)(currMB->cbp_blk >> 32));
    for (int k = 0; k < 17; k++) b_put(img->cofDC[uv + 1][0][k]);
    for (int k = 0; k < 17; k++) b_put(img->cofDC[uv + 1][1][k]);
    for (int b = 0; b < 8; b++) {
      for (int k = 0; k < 16; k++) b_put(b < nb8 * 4 ? img->cofAC[4 + (b >> 2) + uvs][b & 3][0][k] : 0);
      for (int k = 0; k < 16; k++) b_put(b < nb8 * 4 ? img->cofAC[4 + (b >> 2) + uvs][b & 3][1][k] : 0);
    }
    put_recon(enc_picture->imgUV[uv], img->pix_c_y, img->pix_c_x, img->mb_cr_size_y, img->mb_cr_size_x);
    put_fadj(img->AdaptiveRounding ? img->fadjust4x4Cr[intra][uv] : NULL, img->mb_cr_size_y, img->mb_cr_size_x);
    b_flush(K_DCTC);
  }
  return r;
}
