/*
 * jmo_tq.c -- ORACLE (test infrastructure): integer transforms, quantisation, reconstruction.
 * Restates lencod/src/transform.c, the hot part of lencod/src/block.c (dct_4x4, dct_16x16,
 * dct_chroma) and lencod/src/transform8x8.c (dct_8x8) of the reference. Lossless (qpprime) paths
 * and SP-slice variants are out of scope.
 */
#include <string.h>
#include "jmo.h"

static inline int iabs_(int x) { return x < 0 ? -x : x; }
static inline int isignab(int a, int b) { return b < 0 ? -iabs_(a) : iabs_(a); }   /* ifunctions.h */
static inline int rsr(int x, int a) { return (x + (1 << (a - 1))) >> a; }          /* rshift_rnd_sf */
static inline int clip1(int hi, int x) { return x < 0 ? 0 : (x > hi ? hi : x); }
static inline int imin_(int a, int b) { return a < b ? a : b; }
static inline int imax_(int a, int b) { return a > b ? a : b; }

#define Q_BITS    15
#define Q_BITS_8  16
#define DQ_BITS   6
#define DQ_BITS_8 6
#define CAVLC_LEVEL_LIMIT 2063

const int jmo_quant_coef[6][4][4] = {                                               /* block.c:39 */
  {{13107, 8066,13107, 8066},{ 8066, 5243, 8066, 5243},{13107, 8066,13107, 8066},{ 8066, 5243, 8066, 5243}},
  {{11916, 7490,11916, 7490},{ 7490, 4660, 7490, 4660},{11916, 7490,11916, 7490},{ 7490, 4660, 7490, 4660}},
  {{10082, 6554,10082, 6554},{ 6554, 4194, 6554, 4194},{10082, 6554,10082, 6554},{ 6554, 4194, 6554, 4194}},
  {{ 9362, 5825, 9362, 5825},{ 5825, 3647, 5825, 3647},{ 9362, 5825, 9362, 5825},{ 5825, 3647, 5825, 3647}},
  {{ 8192, 5243, 8192, 5243},{ 5243, 3355, 5243, 3355},{ 8192, 5243, 8192, 5243},{ 5243, 3355, 5243, 3355}},
  {{ 7282, 4559, 7282, 4559},{ 4559, 2893, 4559, 2893},{ 7282, 4559, 7282, 4559},{ 4559, 2893, 4559, 2893}}
};
const int jmo_dequant_coef[6][4][4] = {                                             /* block.c:48 */
  {{10, 13, 10, 13},{ 13, 16, 13, 16},{10, 13, 10, 13},{ 13, 16, 13, 16}},
  {{11, 14, 11, 14},{ 14, 18, 14, 18},{11, 14, 11, 14},{ 14, 18, 14, 18}},
  {{13, 16, 13, 16},{ 16, 20, 16, 20},{13, 16, 13, 16},{ 16, 20, 16, 20}},
  {{14, 18, 14, 18},{ 18, 23, 18, 23},{14, 18, 14, 18},{ 18, 23, 18, 23}},
  {{16, 20, 16, 20},{ 20, 25, 20, 25},{16, 20, 16, 20},{ 20, 25, 20, 25}},
  {{18, 23, 18, 23},{ 23, 29, 23, 29},{18, 23, 18, 23},{ 23, 29, 23, 29}}
};
const unsigned char jmo_qp_scale_cr[52] = {                                         /* block.c:64 */
  0, 1, 2, 3, 4, 5, 6, 7, 8, 9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,
  28,29,29,30,31,32,32,33,34,34,35,35,36,36,37,37,37,38,38,38,39,39,39,39
};
const unsigned char jmo_sngl_scan[16][2] = {                                        /* block.h:26 */
  {0,0},{1,0},{0,1},{0,2},{1,1},{2,0},{3,0},{2,1},{1,2},{0,3},{1,3},{2,2},{3,1},{3,2},{2,3},{3,3}
};
const unsigned char jmo_field_scan[16][2] = {                                       /* block.h:35 */
  {0,0},{0,1},{1,0},{0,2},{0,3},{1,1},{1,2},{1,3},{2,0},{2,1},{2,2},{2,3},{3,0},{3,1},{3,2},{3,3}
};
const unsigned char jmo_coeff_cost4x4[2][16] = {                                    /* block.h:45 */
  {3,2,2,1,1,1,0,0,0,0,0,0,0,0,0,0},
  {9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9}
};
static const unsigned char scan_yuv422[8][2] = {                                    /* block.h:52 */
  {0,0},{0,1},{1,0},{0,2},{0,3},{1,1},{1,2},{1,3}
};
/* hor_offset / ver_offset, block.h:61-107, rows for 4:2:0, 4:2:2, 4:4:4 (index yuv_format) */
static const unsigned char hor_offset[4][4][4] = {
  {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}},
  {{0,4,0,4},{0,0,0,0},{0,0,0,0},{0,0,0,0}},
  {{0,4,0,4},{0,4,0,4},{0,0,0,0},{0,0,0,0}},
  {{0,4,0,4},{8,12,8,12},{0,4,0,4},{8,12,8,12}}
};
static const unsigned char ver_offset[4][4][4] = {
  {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}},
  {{0,0,4,4},{0,0,0,0},{0,0,0,0},{0,0,0,0}},
  {{0,0,4,4},{8,8,12,12},{0,0,0,0},{0,0,0,0}},
  {{0,0,4,4},{0,0,4,4},{8,8,12,12},{8,8,12,12}}
};
static const unsigned char cbp_blk_chroma[8][4] = {                                 /* block.h:109 */
  {16,17,18,19},{20,21,22,23},{24,25,26,27},{28,29,30,31},
  {32,33,34,35},{36,37,38,39},{40,41,42,43},{44,45,46,47}
};

/* ------------------------------------------------------------------ transform primitives */

/* forward4x4, transform.c:31 */
void jmo_forward4x4(int (*block)[16], int (*tblock)[16], int pos_y, int pos_x)
{
  int tmp[16], i, *pt = tmp;
  for (i = pos_y; i < pos_y + 4; i++) {
    const int *pb = &block[i][pos_x];
    int t0 = pb[0] + pb[3], t1 = pb[1] + pb[2], t2 = pb[1] - pb[2], t3 = pb[0] - pb[3];
    *pt++ = t0 + t1; *pt++ = (t3 << 1) + t2; *pt++ = t0 - t1; *pt++ = t3 - (t2 << 1);
  }
  for (i = 0; i < 4; i++) {
    int p0 = tmp[i], p1 = tmp[4 + i], p2 = tmp[8 + i], p3 = tmp[12 + i];
    int t0 = p0 + p3, t1 = p1 + p2, t2 = p1 - p2, t3 = p0 - p3;
    tblock[pos_y][pos_x + i] = t0 + t1;
    tblock[pos_y + 1][pos_x + i] = t2 + (t3 << 1);
    tblock[pos_y + 2][pos_x + i] = t0 - t1;
    tblock[pos_y + 3][pos_x + i] = t3 - (t2 << 1);
  }
}

/* inverse4x4, transform.c:81 */
void jmo_inverse4x4(int (*tblock)[16], int (*block)[16], int pos_y, int pos_x)
{
  int tmp[16], i, *pt = tmp;
  for (i = pos_y; i < pos_y + 4; i++) {
    const int *pb = &tblock[i][pos_x];
    int p0 = pb[0] + pb[2], p1 = pb[0] - pb[2], p2 = (pb[1] >> 1) - pb[3], p3 = pb[1] + (pb[3] >> 1);
    *pt++ = p0 + p3; *pt++ = p1 + p2; *pt++ = p1 - p2; *pt++ = p0 - p3;
  }
  for (i = 0; i < 4; i++) {
    int t0 = tmp[i], t1 = tmp[4 + i], t2 = tmp[8 + i], t3 = tmp[12 + i];
    int p0 = t0 + t2, p1 = t0 - t2, p2 = (t1 >> 1) - t3, p3 = t1 + (t3 >> 1);
    block[pos_y][pos_x + i] = p0 + p3;
    block[pos_y + 1][pos_x + i] = p1 + p2;
    block[pos_y + 2][pos_x + i] = p1 - p2;
    block[pos_y + 3][pos_x + i] = p0 - p3;
  }
}

/* hadamard4x4, transform.c:131 (output >> 1) */
void jmo_hadamard4x4(int (*block)[4], int (*tblock)[4])
{
  int tmp[16], i, *pt = tmp;
  for (i = 0; i < 4; i++) {
    const int *pb = block[i];
    int t0 = pb[0] + pb[3], t1 = pb[1] + pb[2], t2 = pb[1] - pb[2], t3 = pb[0] - pb[3];
    *pt++ = t0 + t1; *pt++ = t3 + t2; *pt++ = t0 - t1; *pt++ = t3 - t2;
  }
  for (i = 0; i < 4; i++) {
    int p0 = tmp[i], p1 = tmp[4 + i], p2 = tmp[8 + i], p3 = tmp[12 + i];
    int t0 = p0 + p3, t1 = p1 + p2, t2 = p1 - p2, t3 = p0 - p3;
    tblock[0][i] = (t0 + t1) >> 1; tblock[1][i] = (t2 + t3) >> 1;
    tblock[2][i] = (t0 - t1) >> 1; tblock[3][i] = (t3 - t2) >> 1;
  }
}

/* ihadamard4x4, transform.c:180 */
void jmo_ihadamard4x4(int (*tblock)[4], int (*block)[4])
{
  int tmp[16], i, *pt = tmp;
  for (i = 0; i < 4; i++) {
    const int *pb = tblock[i];
    int p0 = pb[0] + pb[2], p1 = pb[0] - pb[2], p2 = pb[1] - pb[3], p3 = pb[1] + pb[3];
    *pt++ = p0 + p3; *pt++ = p1 + p2; *pt++ = p1 - p2; *pt++ = p0 - p3;
  }
  for (i = 0; i < 4; i++) {
    int t0 = tmp[i], t1 = tmp[4 + i], t2 = tmp[8 + i], t3 = tmp[12 + i];
    int p0 = t0 + t2, p1 = t0 - t2, p2 = t1 - t3, p3 = t1 + t3;
    block[0][i] = p0 + p3; block[1][i] = p1 + p2; block[2][i] = p1 - p2; block[3][i] = p0 - p3;
  }
}

static void fwd8(const int *p, int *o, int so)        /* one 8-point forward pass, transform.c:248-275 */
{
  int a0 = p[0] + p[7], a1 = p[1] + p[6], a2 = p[2] + p[5], a3 = p[3] + p[4];
  int b0 = a0 + a3, b1 = a1 + a2, b2 = a0 - a3, b3 = a1 - a2, b4, b5, b6, b7;
  a0 = p[0] - p[7]; a1 = p[1] - p[6]; a2 = p[2] - p[5]; a3 = p[3] - p[4];
  b4 = a1 + a2 + ((a0 >> 1) + a0);
  b5 = a0 - a3 - ((a2 >> 1) + a2);
  b6 = a0 + a3 - ((a1 >> 1) + a1);
  b7 = a1 - a2 + ((a3 >> 1) + a3);
  o[0 * so] = b0 + b1;        o[1 * so] = b4 + (b7 >> 2);
  o[2 * so] = b2 + (b3 >> 1); o[3 * so] = b5 + (b6 >> 2);
  o[4 * so] = b0 - b1;        o[5 * so] = b6 - (b5 >> 2);
  o[6 * so] = (b2 >> 1) - b3; o[7 * so] = (b4 >> 2) - b7;
}

/* forward8x8, transform.c:229 */
void jmo_forward8x8(int (*block)[16], int (*tblock)[16], int pos_y, int pos_x)
{
  int tmp[64], col[8], i, k;
  for (i = 0; i < 8; i++) fwd8(&block[pos_y + i][pos_x], tmp + 8 * i, 1);
  for (i = 0; i < 8; i++) {
    for (k = 0; k < 8; k++) col[k] = tmp[8 * k + i];
    fwd8(col, &tblock[pos_y][pos_x + i], 16);
  }
}

static void inv8(const int *p, int *o, int so)        /* one 8-point inverse pass, transform.c:346-373 */
{
  int a0 = p[0] + p[4], a1 = p[0] - p[4], a2 = p[6] - (p[2] >> 1), a3 = p[2] + (p[6] >> 1);
  int b0 = a0 + a3, b2 = a1 - a2, b4 = a1 + a2, b6 = a0 - a3, b1, b3, b5, b7;
  a0 = -p[3] + p[5] - p[7] - (p[7] >> 1);
  a1 =  p[1] + p[7] - p[3] - (p[3] >> 1);
  a2 = -p[1] + p[7] + p[5] + (p[5] >> 1);
  a3 =  p[3] + p[5] + p[1] + (p[1] >> 1);
  b1 = a0 + (a3 >> 2); b3 = a1 + (a2 >> 2); b5 = a2 - (a1 >> 2); b7 = a3 - (a0 >> 2);
  o[0 * so] = b0 + b7; o[1 * so] = b2 - b5; o[2 * so] = b4 + b3; o[3 * so] = b6 + b1;
  o[4 * so] = b6 - b1; o[5 * so] = b4 - b3; o[6 * so] = b2 + b5; o[7 * so] = b0 - b7;
}

/* inverse8x8, transform.c:325 */
void jmo_inverse8x8(int (*tblock)[16], int (*block)[16], int pos_y, int pos_x)
{
  int tmp[64], col[8], i, k;
  for (i = 0; i < 8; i++) inv8(&tblock[pos_y + i][pos_x], tmp + 8 * i, 1);
  for (i = 0; i < 8; i++) {
    for (k = 0; k < 8; k++) col[k] = tmp[8 * k + i];
    inv8(col, &block[pos_y][pos_x + i], 16);
  }
}

/* ------------------------------------------------------------------ flat tables */

void jmo_flat_tables4x4(int qp, int offset11, int *levelscale, int *invlevelscale, int *leveloffset)
{
  int k = qp % 6, per = qp / 6, j, i;
  for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) {
    levelscale[j * 4 + i] = jmo_quant_coef[k][j][i];                /* q_matrix.c:482 */
    invlevelscale[j * 4 + i] = jmo_dequant_coef[k][j][i] << 4;      /* q_matrix.c:483 */
    leveloffset[j * 4 + i] = offset11 << (Q_BITS + per - 11);       /* q_offsets.c:508-523 */
  }
}

/* quant_coef8 / dequant_coef8 (transform8x8.c:39-167) are the standard's normAdjust8x8 / LevelScale8x8
 * tables: six position classes per qp%6, keyed on (j%4, i%4) with 1 and 3 equivalent. Class order
 * below: (0,0) (odd,odd) (2,2) (0,odd) (0,2) (odd,2). Checked entry-by-entry against the reference
 * tables by oracle/tap/check_tables.py (container only). */
static const int q8_class[6][6] = {
  {13107, 11428, 20972, 12222, 16777, 15481}, {11916, 10826, 19174, 11058, 14980, 14290},
  {10082,  8943, 15978,  9675, 12710, 11985}, { 9362,  8228, 14913,  8931, 11984, 11259},
  { 8192,  7346, 13159,  7740, 10486,  9777}, { 7282,  6428, 11570,  6830,  9118,  8640}
};
static const int dq8_class[6][6] = {
  {20, 18, 32, 19, 25, 24}, {22, 19, 35, 21, 28, 26}, {26, 23, 42, 24, 33, 31},
  {28, 25, 45, 26, 35, 33}, {32, 28, 51, 30, 40, 38}, {36, 32, 58, 34, 46, 43}
};
static int class8(int j, int i)
{
  int a = j & 3, b = i & 3, ca = a == 0 ? 0 : (a == 2 ? 2 : 1), cb = b == 0 ? 0 : (b == 2 ? 2 : 1);
  if (ca > cb) { int t = ca; ca = cb; cb = t; }
  if (ca == cb) return ca;                 /* (0,0)->0 (1,1)->1 (2,2)->2 */
  if (ca == 0) return cb == 1 ? 3 : 4;     /* (0,1)->3 (0,2)->4 */
  return 5;                                /* (1,2)->5 */
}
int jmo_quant_coef8(int k, int j, int i)   { return q8_class[k][class8(j, i)]; }
int jmo_dequant_coef8(int k, int j, int i) { return dq8_class[k][class8(j, i)]; }

void jmo_flat_tables8x8(int qp, int offset11, int *levelscale, int *invlevelscale, int *leveloffset)
{
  int k = qp % 6, per = qp / 6, j, i;
  for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) {
    levelscale[j * 8 + i] = jmo_quant_coef8(k, j, i);                /* q_matrix.c:626 */
    invlevelscale[j * 8 + i] = jmo_dequant_coef8(k, j, i) << 4;      /* q_matrix.c:627 */
    leveloffset[j * 8 + i] = offset11 << (Q_BITS_8 + per - 11);      /* q_offsets.c CalculateOffset8Param */
  }
}

/* ------------------------------------------------------------------ dct_4x4 */

/* block.c:843-947 */
int jmo_dct_4x4(const jmo_quant *q, int (*m7)[16], const jmo_pel (*mpr)[16], int block_x, int block_y,
                int *coeff_cost, int *levels, int *runs, jmo_pel (*recon)[16], int (*fadjust)[16])
{
  int m4[16][16];
  const unsigned char (*pos_scan)[2] = q->field_scan ? jmo_field_scan : jmo_sngl_scan;
  const unsigned char *c_cost = jmo_coeff_cost4x4[q->disthres];
  const int qp_per = q->qp / 6, q_bits = Q_BITS + qp_per;
  int coeff_ctr, i, j, level, scan_pos = 0, run = -1, nonzero = 0;

  jmo_forward4x4(m7, m4, block_y, block_x);
  for (coeff_ctr = 0; coeff_ctr < 16; coeff_ctr++) {
    int *c, scaled;
    i = pos_scan[coeff_ctr][0]; j = pos_scan[coeff_ctr][1];
    run++;
    c = &m4[block_y + j][block_x + i];
    scaled = iabs_(*c) * q->levelscale[j * 4 + i];
    level = (scaled + q->leveloffset[j * 4 + i]) >> q_bits;
    if (level != 0) {
      if (q->adaptive_rounding)
        fadjust[block_y + j][block_x + i] = rsr(q->adapt_rnd_weight * (scaled - (level << q_bits)), q_bits + 1);
      nonzero = 1;
      *coeff_cost += (level > 1) ? JMO_MAX_VALUE : c_cost[run];
      level = isignab(level, *c);
      levels[scan_pos] = level; runs[scan_pos++] = run;
      *c = rsr((level * q->invlevelscale[j * 4 + i]) << qp_per, 4);
      run = -1;
    } else {
      if (q->adaptive_rounding) fadjust[block_y + j][block_x + i] = 0;
      *c = 0;
    }
  }
  levels[scan_pos] = 0;
  if (scan_pos) {
    jmo_inverse4x4(m4, m7, block_y, block_x);
    for (j = block_y; j < block_y + 4; j++)
      for (i = block_x; i < block_x + 4; i++)
        recon[j][i] = (jmo_pel)clip1(q->max_val, rsr(m7[j][i], DQ_BITS) + mpr[j][i]);
  } else {
    for (j = block_y; j < block_y + 4; j++)
      for (i = block_x; i < block_x + 4; i++) recon[j][i] = mpr[j][i];
  }
  return nonzero;
}

/* ------------------------------------------------------------------ dct_8x8 */

const unsigned char jmo_sngl_scan8x8[64][2] = {                                     /* transform8x8.c:171 */
  {0,0},{1,0},{0,1},{0,2},{1,1},{2,0},{3,0},{2,1},{1,2},{0,3},{0,4},{1,3},{2,2},{3,1},{4,0},{5,0},
  {4,1},{3,2},{2,3},{1,4},{0,5},{0,6},{1,5},{2,4},{3,3},{4,2},{5,1},{6,0},{7,0},{6,1},{5,2},{4,3},
  {3,4},{2,5},{1,6},{0,7},{1,7},{2,6},{3,5},{4,4},{5,3},{6,2},{7,1},{7,2},{6,3},{5,4},{4,5},{3,6},
  {2,7},{3,7},{4,6},{5,5},{6,4},{7,3},{7,4},{6,5},{5,6},{4,7},{5,7},{6,6},{7,5},{7,6},{6,7},{7,7}
};
const unsigned char jmo_field_scan8x8[64][2] = {                                    /* transform8x8.c:184 */
  {0,0},{0,1},{0,2},{1,0},{1,1},{0,3},{0,4},{1,2},{2,0},{1,3},{0,5},{0,6},{0,7},{1,4},{2,1},{3,0},
  {2,2},{1,5},{1,6},{1,7},{2,3},{3,1},{4,0},{3,2},{2,4},{2,5},{2,6},{2,7},{3,3},{4,1},{5,0},{4,2},
  {3,4},{3,5},{3,6},{3,7},{4,3},{5,1},{6,0},{5,2},{4,4},{4,5},{4,6},{4,7},{5,3},{6,1},{6,2},{5,4},
  {5,5},{5,6},{5,7},{6,3},{7,0},{7,1},{6,4},{6,5},{6,6},{6,7},{7,2},{7,3},{7,4},{7,5},{7,6},{7,7}
};
const unsigned char jmo_coeff_cost8x8[2][64] = {                                    /* transform8x8.c:197 */
  {3,3,3,3,2,2,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,0,0,0,0,0,0,0,0,
   0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0},
  {9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,
   9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9}
};

/* transform8x8.c:1452-1653 (non-lossless branch) */
int jmo_dct_8x8(const jmo_quant *q, int (*m7)[16], const jmo_pel (*mpr)[16], int b8, int *coeff_cost,
                int (*levels)[65], int (*runs)[65], jmo_pel (*recon)[16], int (*fadjust)[16])
{
  const int block_x = 8 * (b8 & 1), block_y = 8 * (b8 >> 1);
  const unsigned char (*pos_scan)[2] = q->field_scan ? jmo_field_scan8x8 : jmo_sngl_scan8x8;
  const unsigned char *c_cost = jmo_coeff_cost8x8[q->disthres];
  const int qp_per = q->qp / 6, q_bits = Q_BITS_8 + qp_per;
  const int interleave = q->transform8x8_flag && q->cavlc;           /* :1502 */
  int scan_poss[4] = { 0, 0, 0, 0 }, runs4[4] = { -1, -1, -1, -1 };
  int coeff_ctr, i, j, level, scan_pos = 0, run = -1, nonzero = 0, mc = 0;

  jmo_forward8x8(m7, m7, block_y, block_x);
  for (coeff_ctr = 0; coeff_ctr < 64; coeff_ctr++) {
    int *c, scaled;
    i = pos_scan[coeff_ctr][0]; j = pos_scan[coeff_ctr][1];
    run++;
    if (interleave) { mc = coeff_ctr & 3; runs4[mc]++; }
    c = &m7[block_y + j][block_x + i];
    scaled = iabs_(*c) * q->levelscale[j * 8 + i];
    level = (scaled + q->leveloffset[j * 8 + i]) >> q_bits;
    if (level != 0) {
      if (q->adaptive_rounding)
        fadjust[block_y + j][block_x + i] = rsr(q->adapt_rnd_weight * (scaled - (level << q_bits)), q_bits + 1);
      nonzero = 1;
      if (interleave) {
        *coeff_cost += (level > 1) ? JMO_MAX_VALUE : c_cost[runs4[mc]];
        levels[mc][scan_poss[mc]] = isignab(level, *c);
        runs[mc][scan_poss[mc]++] = runs4[mc];
        runs4[mc] = -1;
      } else {
        *coeff_cost += (level > 1) ? JMO_MAX_VALUE : c_cost[run];
        levels[0][scan_pos] = isignab(level, *c);
        runs[0][scan_pos++] = run;
        run = -1;
      }
      level = isignab(level, *c);
      *c = rsr((level * q->invlevelscale[j * 8 + i]) << qp_per, 6);
    } else {
      if (q->adaptive_rounding) fadjust[block_y + j][block_x + i] = 0;
      *c = 0;
    }
  }
  if (!interleave) levels[0][scan_pos] = 0;
  else for (i = 0; i < 4; i++) levels[i][scan_poss[i]] = 0;

  if (nonzero) {
    jmo_inverse8x8(m7, m7, block_y, block_x);
    for (j = block_y; j < block_y + 8; j++)
      for (i = block_x; i < block_x + 8; i++)
        recon[j][i] = (jmo_pel)clip1(q->max_val, rsr(m7[j][i], DQ_BITS_8) + mpr[j][i]);
  } else {
    for (j = block_y; j < block_y + 8; j++)
      for (i = block_x; i < block_x + 8; i++) recon[j][i] = mpr[j][i];
  }
  return nonzero;
}

/* ------------------------------------------------------------------ dct_16x16 */

/* block.c:564-826 (non-lossless branch). leveloffset is ptLevelOffset4x4[1][qp] (intra). */
int jmo_dct_16x16(const jmo_quant *q, const jmo_pel (*cur)[16], const jmo_pel (*pred)[16],
                  int *dc_levels, int *dc_runs, int (*ac_levels)[16], int (*ac_runs)[16],
                  jmo_pel (*recon)[16], int (*fadjust)[16])
{
  int M1[16][16], M4[4][4];
  const unsigned char (*pos_scan)[2] = q->field_scan ? jmo_field_scan : jmo_sngl_scan;
  const int qp_per = q->qp / 6, q_bits = Q_BITS + qp_per;
  int i, j, ii, jj, run, scan_pos, coeff_ctr, level, ac_coef = 0;

  for (j = 0; j < 16; j++) for (i = 0; i < 16; i++) M1[j][i] = cur[j][i] - pred[j][i];
  for (j = 0; j < 16; j += 4) for (i = 0; i < 16; i += 4) jmo_forward4x4(M1, M1, j, i);
  for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) M4[j][i] = M1[j << 2][i << 2];
  jmo_hadamard4x4(M4, M4);

  run = -1; scan_pos = 0;
  for (coeff_ctr = 0; coeff_ctr < 16; coeff_ctr++) {
    i = pos_scan[coeff_ctr][0]; j = pos_scan[coeff_ctr][1];
    run++;
    level = (iabs_(M4[j][i]) * q->levelscale[0] + (q->leveloffset[0] << 1)) >> (q_bits + 1);   /* :643 */
    if (level != 0) {
      if (q->cavlc && q->img_qp < 10) level = imin_(level, CAVLC_LEVEL_LIMIT);
      level = isignab(level, M4[j][i]);
      dc_levels[scan_pos] = level; dc_runs[scan_pos++] = run;
      run = -1;
      M4[j][i] = level;
    } else M4[j][i] = 0;
  }
  dc_levels[scan_pos] = 0;
  jmo_ihadamard4x4(M4, M4);
  for (j = 0; j < 4; j++) for (i = 0; i < 4; i++)
    M1[j << 2][i << 2] = rsr((M4[j][i] * q->invlevelscale[0]) << qp_per, 6);                    /* :667 */

  for (jj = 0; jj < 4; jj++) for (ii = 0; ii < 4; ii++) {
    const int jpos = jj << 2, ipos = ii << 2;
    const int b8 = 2 * (jj >> 1) + (ii >> 1), b4 = 2 * (jj & 1) + (ii & 1);
    int *ACLevel = ac_levels[b8 * 4 + b4], *ACRun = ac_runs[b8 * 4 + b4];
    run = -1; scan_pos = 0;
    for (coeff_ctr = 1; coeff_ctr < 16; coeff_ctr++) {
      int *c, scaled;
      i = pos_scan[coeff_ctr][0]; j = pos_scan[coeff_ctr][1];
      run++;
      c = &M1[jpos + j][ipos + i];
      scaled = iabs_(*c) * q->levelscale[j * 4 + i];
      level = (scaled + q->leveloffset[j * 4 + i]) >> q_bits;
      if (level != 0) {
        if (q->adaptive_rounding)
          fadjust[jpos + j][ipos + i] = rsr(q->adapt_rnd_weight * (scaled - (level << q_bits)), q_bits + 1);
        ac_coef = 15;
        level = isignab(level, *c);
        ACLevel[scan_pos] = level; ACRun[scan_pos++] = run;
        run = -1;
        *c = rsr((level * q->invlevelscale[j * 4 + i]) << qp_per, 4);
      } else {
        *c = 0;
        if (q->adaptive_rounding) fadjust[jpos + j][ipos + i] = 0;
      }
    }
    ACLevel[scan_pos] = 0;
    jmo_inverse4x4(M1, M1, jpos, ipos);
  }
  for (j = 0; j < 16; j++) for (i = 0; i < 16; i++)
    recon[j][i] = (jmo_pel)clip1(q->max_val, rsr(M1[j][i], DQ_BITS) + pred[j][i]);
  return ac_coef;
}

/* ------------------------------------------------------------------ dct_chroma */

/* block.c:1051-1495 (non-lossless branch) */
int jmo_dct_chroma(const jmo_quant *q, const jmo_quant *qdc, int yuv, int uv, int cr_cbp,
                   int (*m7)[16], const jmo_pel (*mpr)[16], int *dc_levels, int *dc_runs,
                   int (*ac_levels)[16], int (*ac_runs)[16],
                   jmo_pel (*recon)[16], int (*fadjust)[16], long long *cbp_blk)
{
  static const long long cbpblk_pattern[4] = { 0, 0xf0000, 0xff0000, 0xffff0000LL };
  const unsigned char (*pos_scan)[2] = q->field_scan ? jmo_field_scan : jmo_sngl_scan;
  const unsigned char *c_cost = jmo_coeff_cost4x4[q->disthres];
  const int qp_per = q->qp / 6, q_bits = Q_BITS + qp_per;
  const int mb_cr_size_x = (yuv == JMO_YUV444) ? 16 : 8, mb_cr_size_y = (yuv == JMO_YUV420) ? 8 : 16;
  const int num_blk8x8_uv = (yuv == JMO_YUV420) ? 2 : (yuv == JMO_YUV422 ? 4 : 8);   /* lencod.c init_img */
  const int uv_scale = uv * (num_blk8x8_uv >> 1);
  int m1[4], m5[4], m6[4], m3[4][4], m4[4][4];
  int i, j, n1, n2, coeff_ctr, level, scan_pos, run, b8, b4;
  int coeff_cost = 0, cr_cbp_tmp = 0, DCcoded = 0;

  /* :1116-1122 -- note the (n1, n2) argument order: rows 0..mb_cr_size_x-1, cols 0..mb_cr_size_y-1 */
  for (n2 = 0; n2 < mb_cr_size_y; n2 += 4)
    for (n1 = 0; n1 < mb_cr_size_x; n1 += 4)
      jmo_forward4x4(m7, m7, n1, n2);

  if (yuv == JMO_YUV420) {
    run = -1; scan_pos = 0;
    m1[0] = m7[0][0] + m7[0][4] + m7[4][0] + m7[4][4];
    m1[1] = m7[0][0] - m7[0][4] + m7[4][0] - m7[4][4];
    m1[2] = m7[0][0] + m7[0][4] - m7[4][0] - m7[4][4];
    m1[3] = m7[0][0] - m7[0][4] - m7[4][0] + m7[4][4];
    for (coeff_ctr = 0; coeff_ctr < 4; coeff_ctr++) {
      run++;
      level = (iabs_(m1[coeff_ctr]) * q->levelscale[0] + (q->leveloffset[0] << 1)) >> (q_bits + 1);
      if (level != 0) {
        if (q->cavlc && q->img_qp < 4) level = imin_(level, CAVLC_LEVEL_LIMIT);
        *cbp_blk |= 0xf0000LL << (uv << 2);
        cr_cbp = imax_(1, cr_cbp);
        DCcoded = 1;
        level = isignab(level, m1[coeff_ctr]);
        dc_levels[scan_pos] = level; dc_runs[scan_pos++] = run;
        run = -1;
        m1[coeff_ctr] = level;
      } else m1[coeff_ctr] = 0;
    }
    dc_levels[scan_pos] = 0;
    m5[0] = m1[0] + m1[1] + m1[2] + m1[3];
    m5[1] = m1[0] - m1[1] + m1[2] - m1[3];
    m5[2] = m1[0] + m1[1] - m1[2] - m1[3];
    m5[3] = m1[0] - m1[1] - m1[2] + m1[3];
    m7[0][0] = ((m5[0] * q->invlevelscale[0]) << qp_per) >> 5;
    m7[0][4] = ((m5[1] * q->invlevelscale[0]) << qp_per) >> 5;
    m7[4][0] = ((m5[2] * q->invlevelscale[0]) << qp_per) >> 5;
    m7[4][4] = ((m5[3] * q->invlevelscale[0]) << qp_per) >> 5;
  } else if (yuv == JMO_YUV422) {
    const int qp_per_dc = qdc->qp / 6, q_bits_422 = Q_BITS + qp_per_dc;
    for (j = 0; j < mb_cr_size_y; j += 4)
      for (i = 0; i < mb_cr_size_x; i += 4) m3[i >> 2][j >> 2] = m7[j][i];
    for (j = 0; j < 4; j++) { m4[0][j] = m3[0][j] + m3[1][j]; m4[1][j] = m3[0][j] - m3[1][j]; }
    for (i = 0; i < 2; i++) {
      m5[0] = m4[i][0] + m4[i][3]; m5[1] = m4[i][1] + m4[i][2];
      m5[2] = m4[i][1] - m4[i][2]; m5[3] = m4[i][0] - m4[i][3];
      m4[i][0] = m5[0] + m5[1]; m4[i][2] = m5[0] - m5[1];
      m4[i][1] = m5[3] + m5[2]; m4[i][3] = m5[3] - m5[2];
    }
    run = -1; scan_pos = 0;
    for (coeff_ctr = 0; coeff_ctr < 8; coeff_ctr++) {
      i = scan_yuv422[coeff_ctr][0]; j = scan_yuv422[coeff_ctr][1];
      run++;
      /* :1263 -- AC levelscale with the DC (qp+3) offset table, as in JM */
      level = (iabs_(m4[i][j]) * q->levelscale[0] + (qdc->leveloffset[0] * 2)) >> (q_bits_422 + 1);
      if (level != 0) {
        *cbp_blk |= (long long)(int)(0xff0000u << (uv << 3));   /* block.c:1268: int arithmetic in JM -- for uv=1 the value is negative and sign-extends into bits 32..63 */
        cr_cbp = imax_(1, cr_cbp);
        DCcoded = 1;
        dc_levels[scan_pos] = isignab(level, m4[i][j]); dc_runs[scan_pos++] = run;
        run = -1;
      }
      m3[i][j] = isignab(level, m4[i][j]);
    }
    dc_levels[scan_pos] = 0;
    for (j = 0; j < 4; j++) { m4[0][j] = m3[0][j] + m3[1][j]; m4[1][j] = m3[0][j] - m3[1][j]; }
    for (i = 0; i < 2; i++) {
      const int inv = qdc->invlevelscale[0];
      m6[0] = m4[i][0] + m4[i][2]; m6[1] = m4[i][0] - m4[i][2];
      m6[2] = m4[i][1] - m4[i][3]; m6[3] = m4[i][1] + m4[i][3];
      if (qp_per_dc < 4) {
        m7[0][i * 4]  = ((((m6[0] + m6[3]) * inv + (1 << (3 - qp_per_dc))) >> (4 - qp_per_dc)) + 2) >> 2;
        m7[4][i * 4]  = ((((m6[1] + m6[2]) * inv + (1 << (3 - qp_per_dc))) >> (4 - qp_per_dc)) + 2) >> 2;
        m7[8][i * 4]  = ((((m6[1] - m6[2]) * inv + (1 << (3 - qp_per_dc))) >> (4 - qp_per_dc)) + 2) >> 2;
        m7[12][i * 4] = ((((m6[0] - m6[3]) * inv + (1 << (3 - qp_per_dc))) >> (4 - qp_per_dc)) + 2) >> 2;
      } else {
        m7[0][i * 4]  = ((((m6[0] + m6[3]) * inv) << (qp_per_dc - 4)) + 2) >> 2;
        m7[4][i * 4]  = ((((m6[1] + m6[2]) * inv) << (qp_per_dc - 4)) + 2) >> 2;
        m7[8][i * 4]  = ((((m6[1] - m6[2]) * inv) << (qp_per_dc - 4)) + 2) >> 2;
        m7[12][i * 4] = ((((m6[0] - m6[3]) * inv) << (qp_per_dc - 4)) + 2) >> 2;
      }
    }
  }

  /* AC :1321-1380 */
  for (b8 = 0; b8 < (num_blk8x8_uv >> 1); b8++) for (b4 = 0; b4 < 4; b4++) {
    const long long uv_cbpblk = 1LL << cbp_blk_chroma[b8 + uv_scale][b4];
    int *ACLevel = ac_levels[b8 * 4 + b4], *ACRun = ac_runs[b8 * 4 + b4];
    n1 = hor_offset[yuv][b8][b4]; n2 = ver_offset[yuv][b8][b4];
    run = -1; scan_pos = 0;
    for (coeff_ctr = 1; coeff_ctr < 16; coeff_ctr++) {
      int *c, scaled;
      i = pos_scan[coeff_ctr][0]; j = pos_scan[coeff_ctr][1];
      c = &m7[n2 + j][n1 + i];
      ++run;
      scaled = iabs_(*c) * q->levelscale[j * 4 + i];
      level = (scaled + q->leveloffset[j * 4 + i]) >> q_bits;
      if (level != 0) {
        if (q->adaptive_rounding)
          fadjust[n2 + j][n1 + i] = rsr(q->adapt_rnd_weight * (scaled - (level << q_bits)), q_bits + 1);
        *cbp_blk |= uv_cbpblk;
        coeff_cost += (level > 1) ? JMO_MAX_VALUE : c_cost[run];
        cr_cbp_tmp = 2;
        level = isignab(level, *c);
        ACLevel[scan_pos] = level; ACRun[scan_pos++] = run;
        run = -1;
        *c = rsr((level * q->invlevelscale[j * 4 + i]) << qp_per, 4);
      } else {
        *c = 0;
        if (q->adaptive_rounding) fadjust[n2 + j][n1 + i] = 0;
      }
    }
    ACLevel[scan_pos] = 0;
  }

  /* thresholding :1384-1410 (_CHROMA_COEFF_COST_ = 4, defines.h:103) */
  if (coeff_cost < 4) {
    const long long uv_cbpblk = cbpblk_pattern[yuv] << (uv << (1 + yuv));
    cr_cbp_tmp = 0;
    for (b8 = 0; b8 < (num_blk8x8_uv >> 1); b8++) for (b4 = 0; b4 < 4; b4++) {
      int *ACLevel = ac_levels[b8 * 4 + b4];
      n1 = hor_offset[yuv][b8][b4]; n2 = ver_offset[yuv][b8][b4];
      if (DCcoded == 0) *cbp_blk &= ~uv_cbpblk;
      ACLevel[0] = 0;
      for (coeff_ctr = 1; coeff_ctr < 16; coeff_ctr++) {
        m7[n2 + pos_scan[coeff_ctr][1]][n1 + pos_scan[coeff_ctr][0]] = 0;
        ACLevel[coeff_ctr] = 0;
      }
    }
  }
  if (cr_cbp_tmp == 2) cr_cbp = 2;

  for (n2 = 0; n2 < mb_cr_size_y; n2 += 4)
    for (n1 = 0; n1 < mb_cr_size_x; n1 += 4)
      jmo_inverse4x4(m7, m7, n2, n1);
  for (j = 0; j < mb_cr_size_y; j++)
    for (i = 0; i < mb_cr_size_x; i++)
      recon[j][i] = (jmo_pel)clip1(q->max_val, rsr(m7[j][i], DQ_BITS) + mpr[j][i]);
  return cr_cbp;
}
