/*
 * jmo_dist.c -- ORACLE (test infrastructure): block distortion kernels.
 * Restates lencod/src/me_distortion.c and lencod/src/refbuf.c of the reference.
 * Return values reproduce JM's early exits exactly (partial sums included).
 */
#include "jmo.h"

static inline int iabs_(int x) { return x < 0 ? -x : x; }
static inline int clip3(int lo, int hi, int x) { return x < lo ? lo : (x > hi ? hi : x); }
static inline int clip1(int hi, int x) { return x < 0 ? 0 : (x > hi ? hi : x); }

/* FastLine4X / UMVLine4X, refbuf.c:25,37: quarter-pel (y,x) incl. the +80 pad offset -> row pointer */
static const jmo_pel *luma_line(const jmo_dist *d, int y, int x)
{
  const jmo_ref *r = d->ref;
  int xpos = x >> 2, ypos = y >> 2;
  if (d->umv) {
    xpos = clip3(0, r->width_pad, xpos);
    ypos = clip3(0, r->height_pad, ypos);
  }
  return r->luma[(y & 3) * 4 + (x & 3)] + (long)ypos * r->Wp + xpos;
}

/* FastLine8X_chroma / UMVLine8X_chroma, refbuf.c:53,69 */
static const jmo_pel *chroma_line(const jmo_dist *d, int k, int y, int x)
{
  const jmo_ref *r = d->ref;
  const jmo_chroma_geom *g = &r->cg;
  int xpos = x >> g->shift_x, ypos = y >> g->shift_y;
  if (d->umv) {
    xpos = clip3(0, r->width_pad_cr, xpos);
    ypos = clip3(0, r->height_pad_cr, ypos);
  }
  return r->cr[k][(y & g->mask_y) * g->sub_x + (x & g->mask_x)] + (long)ypos * r->Wcp + xpos;
}

#define WP_LUMA(d, v)  clip1((d)->max_val, ((((d)->weight_luma * (v)) + (d)->wp_luma_round) >> (d)->luma_log_weight_denom) + (d)->offset_luma)
#define WP_CR(d, k, v) clip1((d)->max_val_uv, ((((d)->weight_cr[k] * (v)) + (d)->wp_chroma_round) >> (d)->chroma_log_weight_denom) + (d)->offset_cr[k])

/* computeSAD me_distortion.c:351 / computeSADWP :413 (wp != 0) / computeSSE :1042 (sse != 0) / computeSSEWP :1107 (both) */
static int sad_core(const jmo_dist *d, const jmo_pel *src_pic, int bsy, int bsx, int min_mcost,
                    int cand_x, int cand_y, int wp, int sse)
{
  const jmo_ref *r = d->ref;
  int mcost = 0, y, x;
  const jmo_pel *src = src_pic;
  const jmo_pel *ref = luma_line(d, cand_y, cand_x);
  if (!wp && !sse && !d->chroma_me) {                   /* the plain computeSAD loop, :364-375 */
    for (y = 0; y < bsy; y++) {
      for (x = 0; x < bsx; x += 4) {
        mcost += iabs_(src[0] - ref[x]) + iabs_(src[1] - ref[x + 1]) + iabs_(src[2] - ref[x + 2]) + iabs_(src[3] - ref[x + 3]);
        src += 4;
      }
      if (mcost >= min_mcost) return mcost;
      ref += r->Wp;
    }
    return mcost;
  }
  for (y = 0; y < bsy; y++) {
    for (x = 0; x < bsx; x++) {
      int rv = wp ? WP_LUMA(d, ref[x]) : ref[x];
      int df = *src++ - rv;
      mcost += sse ? df * df : iabs_(df);
    }
    if (mcost >= min_mcost) return mcost;               /* :373 row-wise early exit */
    ref += r->Wp;
  }
  if (d->chroma_me) {                                   /* :376-402 */
    int bsx_c = bsx >> (r->cg.shift_x - 2), bsy_c = bsy >> (r->cg.shift_y - 2), k;
    for (k = 0; k < 2; k++) {
      int mcr = 0;
      src = src_pic + (256 << k);
      ref = chroma_line(d, k, cand_y, cand_x);
      for (y = 0; y < bsy_c; y++) {
        for (x = 0; x < bsx_c; x++) {
          int rv = wp ? WP_CR(d, k, ref[x]) : ref[x];
          int df = *src++ - rv;
          mcr += sse ? df * df : iabs_(df);
        }
        ref += r->Wcp;
      }
      mcost += d->chroma_me_weight * mcr;
      if (mcost >= min_mcost) return mcost;
    }
  }
  return mcost;
}

int jmo_sad(const jmo_dist *d, const jmo_pel *src, int bsy, int bsx, int min_mcost, int cand_x, int cand_y)
{ return sad_core(d, src, bsy, bsx, min_mcost, cand_x, cand_y, 0, 0); }
int jmo_sad_wp(const jmo_dist *d, const jmo_pel *src, int bsy, int bsx, int min_mcost, int cand_x, int cand_y)
{ return sad_core(d, src, bsy, bsx, min_mcost, cand_x, cand_y, 1, 0); }
int jmo_sse(const jmo_dist *d, const jmo_pel *src, int bsy, int bsx, int min_mcost, int cand_x, int cand_y)
{ return sad_core(d, src, bsy, bsx, min_mcost, cand_x, cand_y, 0, 1); }
/* computeSSEWP me_distortion.c:1107: the weighted samples of computeSADWP, squared differences */
int jmo_sse_wp(const jmo_dist *d, const jmo_pel *src, int bsy, int bsx, int min_mcost, int cand_x, int cand_y)
{ return sad_core(d, src, bsy, bsx, min_mcost, cand_x, cand_y, 1, 1); }

/* HadamardSAD4x4, me_distortion.c:182-264: 2-D 4x4 Hadamard of the difference block, sum |.|, (s+1)>>1.
 * The butterfly below is the separable form of the reference's four-stage network; |.| sums are
 * invariant to the output ordering/sign conventions of that network. */
int jmo_hadamard_sad4x4(const int *diff)
{
  int m[16], d[16], k, satd = 0;
  for (k = 0; k < 4; k++) {           /* columns: rows 0..3 of each column k */
    int a = diff[k] + diff[12 + k], b = diff[4 + k] + diff[8 + k];
    int c = diff[4 + k] - diff[8 + k], e = diff[k] - diff[12 + k];
    m[k] = a + b; m[4 + k] = e + c; m[8 + k] = a - b; m[12 + k] = e - c;
  }
  for (k = 0; k < 4; k++) {           /* rows */
    const int *r = m + 4 * k;
    int a = r[0] + r[3], b = r[1] + r[2], c = r[1] - r[2], e = r[0] - r[3];
    d[4 * k] = a + b; d[4 * k + 1] = a - b; d[4 * k + 2] = c + e; d[4 * k + 3] = e - c;
  }
  for (k = 0; k < 16; k++) satd += iabs_(d[k]);
  return (satd + 1) >> 1;
}

/* HadamardSAD8x8, me_distortion.c:272-343: (sum |H8 D H8| + 2) >> 2 */
int jmo_hadamard_sad8x8(const int *diff)
{
  int m1[8][8], m2[8][8], m3[8][8], i, j, sad = 0;
  for (j = 0; j < 8; j++) {
    const int *p = diff + 8 * j;
    for (i = 0; i < 4; i++) { m2[j][i] = p[i] + p[i + 4]; m2[j][i + 4] = p[i] - p[i + 4]; }
    m1[j][0] = m2[j][0] + m2[j][2]; m1[j][1] = m2[j][1] + m2[j][3];
    m1[j][2] = m2[j][0] - m2[j][2]; m1[j][3] = m2[j][1] - m2[j][3];
    m1[j][4] = m2[j][4] + m2[j][6]; m1[j][5] = m2[j][5] + m2[j][7];
    m1[j][6] = m2[j][4] - m2[j][6]; m1[j][7] = m2[j][5] - m2[j][7];
    for (i = 0; i < 8; i += 2) { m2[j][i] = m1[j][i] + m1[j][i + 1]; m2[j][i + 1] = m1[j][i] - m1[j][i + 1]; }
  }
  for (i = 0; i < 8; i++) {
    for (j = 0; j < 4; j++) { m3[j][i] = m2[j][i] + m2[j + 4][i]; m3[j + 4][i] = m2[j][i] - m2[j + 4][i]; }
    m1[0][i] = m3[0][i] + m3[2][i]; m1[1][i] = m3[1][i] + m3[3][i];
    m1[2][i] = m3[0][i] - m3[2][i]; m1[3][i] = m3[1][i] - m3[3][i];
    m1[4][i] = m3[4][i] + m3[6][i]; m1[5][i] = m3[5][i] + m3[7][i];
    m1[6][i] = m3[4][i] - m3[6][i]; m1[7][i] = m3[5][i] - m3[7][i];
    for (j = 0; j < 8; j += 2) { m2[j][i] = m1[j][i] + m1[j + 1][i]; m2[j + 1][i] = m1[j][i] - m1[j + 1][i]; }
  }
  for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) sad += iabs_(m2[j][i]);
  return (sad + 2) >> 2;
}

/* computeSATD me_distortion.c:657 / computeSATDWP :734. Note the line pointer is fetched PER sub-block
 * (:678, :704), so under UMV access each 4x4/8x8 sub-block origin is clamped on its own; the early exit
 * is block-wise with a strict '>' (:690, :720). No chroma term. */
static int satd_core(const jmo_dist *d, const jmo_pel *src_pic, int bsy, int bsx, int min_mcost,
                     int cand_x, int cand_y, int wp)
{
  const jmo_ref *r = d->ref;
  const int bs = d->test8x8 ? 8 : 4;
  int diff[64], mcost = 0, y, x, yy, xx;
  const jmo_pel *src_tmp = src_pic;
  for (y = cand_y; y < cand_y + (bsy << 2); y += bs << 2) {
    for (x = 0; x < bsx; x += bs) {
      int *dp = diff;
      const jmo_pel *ref = luma_line(d, y, cand_x + (x << 2));
      const jmo_pel *src = src_tmp + x;
      for (yy = 0; yy < bs; yy++) {
        for (xx = 0; xx < bs; xx++) {
          int rv = wp ? WP_LUMA(d, ref[xx]) : ref[xx];
          *dp++ = src[xx] - rv;
        }
        ref += r->Wp;
        src += bsx;
      }
      mcost += d->test8x8 ? jmo_hadamard_sad8x8(diff) : jmo_hadamard_sad4x4(diff);
      if (mcost > min_mcost) return mcost;
    }
    src_tmp += bsx * bs;
  }
  return mcost;
}

int jmo_satd(const jmo_dist *d, const jmo_pel *src, int bsy, int bsx, int min_mcost, int cand_x, int cand_y)
{ return satd_core(d, src, bsy, bsx, min_mcost, cand_x, cand_y, 0); }
int jmo_satd_wp(const jmo_dist *d, const jmo_pel *src, int bsy, int bsx, int min_mcost, int cand_x, int cand_y)
{ return satd_core(d, src, bsy, bsx, min_mcost, cand_x, cand_y, 1); }
