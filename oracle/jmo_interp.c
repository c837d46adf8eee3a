/*
 * jmo_interp.c -- ORACLE (test infrastructure): sub-pel reference plane generation.
 * Restates lencod/src/img_luma.c and lencod/src/img_chroma.c of the reference as pure functions.
 */
#include <string.h>
#include "jmo.h"

static inline int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline int clip1(int hi, int x) { return x < 0 ? 0 : (x > hi ? hi : x); }   /* iClip1, ifunctions.h */
static inline int rsr(int x, int a) { return (x + (1 << (a - 1))) >> a; }          /* rshift_rnd_sf */

/* chroma_mc_setup, lencod.c:2851-2884; plane counts img_chroma.c:390-405 */
void jmo_chroma_geometry(int yuv_format, jmo_chroma_geom *g)
{
  memset(g, 0, sizeof(*g));
  if (yuv_format == JMO_YUV420) {
    g->sub_x = 8; g->sub_y = 8; g->mul_x = 1; g->mul_y = 1;
    g->pad_x = JMO_PAD >> 1; g->pad_y = JMO_PAD >> 1;
    g->mask_x = 7; g->mask_y = 7; g->shift_x = 3; g->shift_y = 3;
    g->mb_cr_size_x = 8; g->mb_cr_size_y = 8;
  } else if (yuv_format == JMO_YUV422) {
    g->sub_x = 8; g->sub_y = 4; g->mul_x = 1; g->mul_y = 2;
    g->pad_x = JMO_PAD >> 1; g->pad_y = JMO_PAD;
    g->mask_x = 7; g->mask_y = 3; g->shift_x = 3; g->shift_y = 2;
    g->mb_cr_size_x = 8; g->mb_cr_size_y = 16;
  } else if (yuv_format == JMO_YUV444) {
    g->sub_x = 4; g->sub_y = 4; g->mul_x = 2; g->mul_y = 2;
    g->pad_x = JMO_PAD; g->pad_y = JMO_PAD;
    g->mask_x = 3; g->mask_y = 3; g->shift_x = 2; g->shift_y = 2;
    g->mb_cr_size_x = 16; g->mb_cr_size_y = 16;
  }
}

/*
 * getSubImagesLuma, img_luma.c:45-104. The sixteen planes:
 *   [0][0] getSubImageInteger   :120  picture replicated into the 20-pel ring
 *   [0][2] getHorSubImageSixTap :181  (1,-5,20,20,-5,1) along x, x clamped to the PADDED row
 *                                     (:207-267), raw sums kept (imgY_sub_tmp), (is+16)>>5 clipped
 *   [2][0] getVerSubImageSixTap :293  same along y on [0][0], y clamped to the padded plane (:308-367)
 *   [2][2] getVerSubImageSixTapTmp :390  along y on the raw sums, (is+512)>>10 clipped
 *   others: (a+b+1)>>1 of two of those, with a +1 column / +1 row offset on one operand clamped to
 *           the last column / row (:483-639), paired exactly as :78-103.
 */
void jmo_interp_luma(const jmo_pel *img, int W, int H, int stride, int max_val, jmo_pel *out)
{
  const int Wp = W + 2 * JMO_PAD, Hp = H + 2 * JMO_PAD;
  const long plane = (long)Wp * Hp;
#define P(py, px) (out + ((py) * 4 + (px)) * plane)
  jmo_pel *p00 = P(0, 0), *p02 = P(0, 2), *p20 = P(2, 0), *p22 = P(2, 2);
  int *tmp;     /* imgY_sub_tmp, lencod.c:2135 */
  int j, i;

  /* [0][0] */
  for (j = 0; j < Hp; j++) {
    const jmo_pel *src = img + (long)clampi(j - JMO_PAD, 0, H - 1) * stride;
    jmo_pel *dst = p00 + (long)j * Wp;
    for (i = 0; i < Wp; i++) dst[i] = src[clampi(i - JMO_PAD, 0, W - 1)];
  }

  tmp = (int *)__builtin_malloc(sizeof(int) * plane);

  /* [0][2] + raw sums */
  for (j = 0; j < Hp; j++) {
    const jmo_pel *s = p00 + (long)j * Wp;
    for (i = 0; i < Wp; i++) {
      int a = s[i], d = s[clampi(i + 1, 0, Wp - 1)];
      int b = s[clampi(i - 1, 0, Wp - 1)], e = s[clampi(i + 2, 0, Wp - 1)];
      int c = s[clampi(i - 2, 0, Wp - 1)], f = s[clampi(i + 3, 0, Wp - 1)];
      int is = 20 * (a + d) - 5 * (b + e) + (c + f);
      tmp[(long)j * Wp + i] = is;
      p02[(long)j * Wp + i] = (jmo_pel)clip1(max_val, rsr(is, 5));
    }
  }

  /* [2][0] and [2][2] */
  for (j = 0; j < Hp; j++) {
    long ra = (long)j * Wp, rd = (long)clampi(j + 1, 0, Hp - 1) * Wp;
    long rb = (long)clampi(j - 1, 0, Hp - 1) * Wp, re = (long)clampi(j + 2, 0, Hp - 1) * Wp;
    long rc = (long)clampi(j - 2, 0, Hp - 1) * Wp, rf = (long)clampi(j + 3, 0, Hp - 1) * Wp;
    for (i = 0; i < Wp; i++) {
      int is = 20 * (p00[ra + i] + p00[rd + i]) - 5 * (p00[rb + i] + p00[re + i]) + (p00[rc + i] + p00[rf + i]);
      p20[ra + i] = (jmo_pel)clip1(max_val, rsr(is, 5));
      is = 20 * (tmp[ra + i] + tmp[rd + i]) - 5 * (tmp[rb + i] + tmp[re + i]) + (tmp[rc + i] + tmp[rf + i]);
      p22[ra + i] = (jmo_pel)clip1(max_val, rsr(is, 10));
    }
  }
  __builtin_free(tmp);

  /* quarter-pel planes */
  for (j = 0; j < Hp; j++) {
    long r = (long)j * Wp, rn = (long)clampi(j + 1, 0, Hp - 1) * Wp;
    for (i = 0; i < Wp; i++) {
      int in = clampi(i + 1, 0, Wp - 1);
      P(0, 1)[r + i] = (jmo_pel)rsr(p00[r + i] + p02[r + i], 1);     /* :78  */
      P(1, 0)[r + i] = (jmo_pel)rsr(p00[r + i] + p20[r + i], 1);     /* :80  */
      P(1, 1)[r + i] = (jmo_pel)rsr(p02[r + i] + p20[r + i], 1);     /* :82  */
      P(1, 2)[r + i] = (jmo_pel)rsr(p02[r + i] + p22[r + i], 1);     /* :84  */
      P(2, 1)[r + i] = (jmo_pel)rsr(p20[r + i] + p22[r + i], 1);     /* :86  */
      P(0, 3)[r + i] = (jmo_pel)rsr(p02[r + i] + p00[r + in], 1);    /* :89  */
      P(1, 3)[r + i] = (jmo_pel)rsr(p02[r + i] + p20[r + in], 1);    /* :91  */
      P(2, 3)[r + i] = (jmo_pel)rsr(p22[r + i] + p20[r + in], 1);    /* :93  */
      P(3, 0)[r + i] = (jmo_pel)rsr(p20[r + i] + p00[rn + i], 1);    /* :96  */
      P(3, 1)[r + i] = (jmo_pel)rsr(p20[r + i] + p02[rn + i], 1);    /* :98  */
      P(3, 2)[r + i] = (jmo_pel)rsr(p22[r + i] + p02[rn + i], 1);    /* :100 */
      P(3, 3)[r + i] = (jmo_pel)rsr(p02[rn + i] + p20[r + in], 1);   /* :103, getDiagSubImageBiLinear :607 */
    }
  }
#undef P
}

/*
 * getSubImagesChroma, img_chroma.c:374-443, one component. For plane (suby, subx) with k = suby*mul_y,
 * l = subx*mul_x the weights are w00=(8-k)(8-l), w01=(8-k)l, w10=k(8-l), w11=kl (:412-420) and every
 * generateChroma* case (:34-360) equals (w00*a+w01*b+w10*c+w11*d+32)>>6 with the source coordinates
 * clamped to the picture -- including the ring, which is filled from the edge-interpolated values
 * (:230-246, :318-335). JM's loops run to size-1 (:63, :129), so the LAST padded row and the LAST padded
 * column are never written and keep calloc's zero (memalloc.c:142): mirrored by not touching them.
 */
void jmo_interp_chroma(const jmo_pel *img, int Wc, int Hc, int stride, int yuv_format, jmo_pel *out)
{
  jmo_chroma_geom g;
  int Wcp, Hcp, suby, subx, j, i;
  long plane;
  jmo_chroma_geometry(yuv_format, &g);
  Wcp = Wc + 2 * g.pad_x; Hcp = Hc + 2 * g.pad_y;
  plane = (long)Wcp * Hcp;
  for (suby = 0; suby < g.sub_y; suby++) {
    int k = suby * g.mul_y;
    for (subx = 0; subx < g.sub_x; subx++) {
      int l = subx * g.mul_x;
      int w00 = (8 - k) * (8 - l), w01 = (8 - k) * l, w10 = k * (8 - l), w11 = k * l;
      jmo_pel *dst = out + (long)(suby * g.sub_x + subx) * plane;
      for (j = 0; j < Hcp - 1; j++) {
        int y0 = clampi(j - g.pad_y, 0, Hc - 1), y1 = clampi(j - g.pad_y + 1, 0, Hc - 1);
        const jmo_pel *r0 = img + (long)y0 * stride, *r1 = img + (long)y1 * stride;
        for (i = 0; i < Wcp - 1; i++) {
          int x0 = clampi(i - g.pad_x, 0, Wc - 1), x1 = clampi(i - g.pad_x + 1, 0, Wc - 1);
          int v = w00 * r0[x0] + w01 * r0[x1] + w10 * r1[x0] + w11 * r1[x1];
          dst[(long)j * Wcp + i] = (jmo_pel)rsr(v, 6);
        }
      }
    }
  }
}

/* StorablePicture geometry: mbuffer.c:413-426 */
void jmo_ref_init(jmo_ref *r, int W, int H, int yuv_format, const jmo_pel *luma,
                  const jmo_pel *cb, const jmo_pel *cr)
{
  memset(r, 0, sizeof(*r));
  r->W = W; r->H = H;
  r->Wp = W + 2 * JMO_PAD; r->Hp = H + 2 * JMO_PAD;
  r->width_pad = W + 2 * JMO_PAD - 1 - 16;
  r->height_pad = H + 2 * JMO_PAD - 1 - 16;
  if (luma) { int k; for (k = 0; k < 16; k++) r->luma[k] = luma + (long)k * r->Wp * r->Hp; }
  r->yuv_format = yuv_format;
  jmo_chroma_geometry(yuv_format, &r->cg);
  if (yuv_format != JMO_YUV400) {
    r->Wc = (yuv_format == JMO_YUV444) ? W : W / 2;
    r->Hc = (yuv_format == JMO_YUV420) ? H / 2 : H;
    r->Wcp = r->Wc + 2 * r->cg.pad_x;
    r->Hcp = r->Hc + 2 * r->cg.pad_y;
    r->width_pad_cr  = (r->Wc - 1) + (r->cg.pad_x << 1) - r->cg.mb_cr_size_x;
    r->height_pad_cr = (r->Hc - 1) + (r->cg.pad_y << 1) - r->cg.mb_cr_size_y;
  }
  if (yuv_format != JMO_YUV400) {
    int k, n = r->cg.sub_x * r->cg.sub_y;
    for (k = 0; k < n; k++) {
      r->cr[0][k] = cb ? cb + (long)k * r->Wcp * r->Hcp : 0;
      r->cr[1][k] = cr ? cr + (long)k * r->Wcp * r->Hcp : 0;
    }
  }
}

/* FNV-1a over 16-bit samples, the plane digest of oracle/tap/tap_capture.c (golden fixtures) */
unsigned jmo_fnv1a16(const jmo_pel *p, long n)
{
  unsigned d = 2166136261u;
  long i;
  for (i = 0; i < n; i++) { d ^= p[i]; d *= 16777619u; }
  return d;
}
