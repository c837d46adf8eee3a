/*
 * tap_field.c -- TEST INFRASTRUCTURE, container only (needs /root/reference headers at build time).
 *
 * Frame-level golden capture for the low-complexity P-slice inter decision (oracle/jmo_lowcplx.c): links the real JM
 * (oracle/_ref/libjm.so), forwards everything, and records per coded P picture
 *   - the source luma, the integer luma of every list-0 reference, POCs and the co-located vector field EPZS reads,
 *   - the Lagrangian factors of the slice,
 *   - every BlockMotionSearch call: (macroblock, reference, block type, block) -> predictor, vector, cost,
 *   - the final field: mb_type, b8mode, enc_picture->ref_idx / mv of LIST_0.
 * Stream of int32 records { magic, kind, n, payload[n] } in $JM_TAP_OUT; tests/golden/make_golden_field.py packs it.
 * Only DATA leaves the reference.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "global.h"
#include "mbuffer.h"
#include "image.h"

extern int jm_main(int argc, char **argv);
extern int frame_ctr[5];

enum { K_FRAME = 20, K_CALLS = 21, K_FIELD = 22 };
static FILE *fo;
static int *buf; static int nbuf, cbuf;
static int *calls; static int ncalls, ccalls;
static void b_reset(void) { nbuf = 0; }
static void b_put(int v) { if (nbuf == cbuf) { cbuf = cbuf ? 2 * cbuf : 1 << 16; buf = realloc(buf, sizeof(int) * cbuf); } buf[nbuf++] = v; }
static void b_flush(int kind) { int h[3] = { 0x4a4d5450, kind, nbuf }; fwrite(h, 4, 3, fo); fwrite(buf, 4, nbuf, fo); }
static void c_put(int v) { if (ncalls == ccalls) { ccalls = ccalls ? 2 * ccalls : 1 << 16; calls = realloc(calls, sizeof(int) * ccalls); } calls[ncalls++] = v; }
static void *next_sym(const char *n) { void *p = dlsym(RTLD_NEXT, n); if (!p) { fprintf(stderr, "tap: no %s\n", n); exit(97); } return p; }

int main(int argc, char **argv)
{
  const char *o = getenv("JM_TAP_OUT");
  int rc;
  fo = fopen(o ? o : "jm_field.bin", "wb");
  if (!fo) return 96;
  rc = jm_main(argc, argv);
  fclose(fo);
  return rc;
}

int BlockMotionSearch(short ref, int list, int mb_x, int mb_y, int blocktype, int search_range, int *lambda_factor)
{
  static int (*orig)(short, int, int, int, int, int, int *);
  int r;
  if (!orig) orig = next_sym("BlockMotionSearch");
  r = orig(ref, list, mb_x, mb_y, blocktype, search_range, lambda_factor);
  if (img->type == P_SLICE && list == 0) {
    const int by = mb_y >> 2, bx = mb_x >> 2;
    c_put(img->current_mb_nr); c_put(ref); c_put(blocktype); c_put(bx); c_put(by);
    c_put(img->pred_mv[by][bx][list][ref][blocktype][0]); c_put(img->pred_mv[by][bx][list][ref][blocktype][1]);
    c_put(img->all_mv[by][bx][list][ref][blocktype][0]); c_put(img->all_mv[by][bx][list][ref][blocktype][1]);
    c_put(r); c_put(search_range); c_put(lambda_factor[0]);
  }
  return r;
}

void DeblockFrame(ImageParameters *im, imgpel **imgY, imgpel ***imgUV)
{
  static void (*orig)(ImageParameters *, imgpel **, imgpel ***);
  if (!orig) orig = next_sym("DeblockFrame");
  if (im->type == P_SLICE && imgY == enc_picture->imgY) {
    const int W = im->width, H = im->height, w4 = W / 4, h4 = H / 4, nmb = (int)im->PicSizeInMbs;
    int i, j, r, k;
    b_reset();
    b_put(W); b_put(H); b_put(im->qp); b_put(listXsize[0]); b_put(im->number); b_put(frame_ctr[B_SLICE]);
    for (k = 0; k < 3; k++) b_put(im->lambda_mf[P_SLICE][im->qp][k]);
    b_put((int)(2 * im->lambda_me[P_SLICE][im->qp][Q_PEL]));
    b_put(enc_picture->poc); b_put(im->num_ref_idx_l0_active);
    for (r = 0; r < listXsize[0]; r++) { b_put(listX[0][r]->poc); b_put((int)(enc_picture->ref_pic_num[LIST_0][r] & 0xffffffff)); b_put((int)(enc_picture->ref_pic_num[LIST_0][r] >> 32)); }
    for (j = 0; j < H; j++) for (i = 0; i < W; i++) b_put(pCurImg[j][i]);
    for (r = 0; r < listXsize[0]; r++) for (j = 0; j < H; j++) for (i = 0; i < W; i++) b_put(listX[0][r]->imgY[j][i]);
    /* what EPZSSliceInit reads of the co-located pictures listX[0][0] and [1]: mv[LIST_0], ref_id[LIST_0] */
    for (k = 0; k < 2; k++) {
      StorablePicture *fs = listX[0][(k && listXsize[0] > 1) ? 1 : 0];
      for (j = 0; j < h4; j++) for (i = 0; i < w4; i++) {
        b_put(fs->mv[LIST_0][j][i][0]); b_put(fs->mv[LIST_0][j][i][1]);
        b_put((int)(fs->ref_id[LIST_0][j][i] & 0xffffffff)); b_put((int)(fs->ref_id[LIST_0][j][i] >> 32));
      }
    }
    b_flush(K_FRAME);
    b_reset();
    for (i = 0; i < ncalls; i++) b_put(calls[i]);
    b_flush(K_CALLS);
    b_reset();
    for (i = 0; i < nmb; i++) {
      Macroblock *m = &im->mb_data[i];
      b_put(m->mb_type); b_put(m->slice_nr);
      for (k = 0; k < 4; k++) b_put(m->b8mode[k]);
    }
    for (j = 0; j < h4; j++) for (i = 0; i < w4; i++) { b_put(enc_picture->ref_idx[LIST_0][j][i]); b_put(enc_picture->mv[LIST_0][j][i][0]); b_put(enc_picture->mv[LIST_0][j][i][1]); }
    b_flush(K_FIELD);
  }
  ncalls = 0;
  orig(im, imgY, imgUV);
}
