/*
 * jmo_frame.c -- ORACLE (test infrastructure): the hot path over a list of macroblocks, in C so that the
 * cpu_baseline leg of bench.py times the restated JM algorithm and not Python. Per macroblock and reference:
 * 41 x BlockMotionSearch chain (mv-search.c:751-826 via jmo_block_search_full: FullPelBlockMotionSearch with
 * computeSAD early exits, SubPelBlockMotionSearch with computeSATD), then for the partitioning with the smallest
 * summed motion cost: LumaPrediction per 4x4 block (macroblock.c:836), chroma prediction from the eighth-pel planes
 * (macroblock.c:1593), dct_4x4 x16 and dct_chroma x2 (block.c:843, :1051). Single-threaded like JM.
 */
#include <string.h>
#include "jmo.h"

static const signed char PART[41][5] = {
  {1,0,0,4,4}, {2,0,0,4,2},{2,0,2,4,2}, {3,0,0,2,4},{3,2,0,2,4},
  {4,0,0,2,2},{4,2,0,2,2},{4,0,2,2,2},{4,2,2,2,2},
  {5,0,0,2,1},{5,0,1,2,1},{5,2,0,2,1},{5,2,1,2,1},{5,0,2,2,1},{5,0,3,2,1},{5,2,2,2,1},{5,2,3,2,1},
  {6,0,0,1,2},{6,1,0,1,2},{6,2,0,1,2},{6,3,0,1,2},{6,0,2,1,2},{6,1,2,1,2},{6,2,2,1,2},{6,3,2,1,2},
  {7,0,0,1,1},{7,1,0,1,1},{7,0,1,1,1},{7,1,1,1,1},{7,2,0,1,1},{7,3,0,1,1},{7,2,1,1,1},{7,3,1,1,1},
  {7,0,2,1,1},{7,1,2,1,1},{7,0,3,1,1},{7,1,3,1,1},{7,2,2,1,1},{7,3,2,1,1},{7,2,3,1,1},{7,3,3,1,1}
};

static int covering(int mode, const int *b8mode, int x4, int y4)
{
  int b8 = 2 * (y4 >> 1) + (x4 >> 1);
  if (mode == 1) return 0;
  if (mode == 2) return 1 + (y4 >> 1);
  if (mode == 3) return 3 + (x4 >> 1);
  switch (b8mode[b8]) {
  case 4: return 5 + b8;
  case 5: return 9 + 2 * b8 + (y4 & 1);
  case 6: return 17 + 2 * b8 + (x4 & 1);
  default: return 25 + 4 * b8 + 2 * (y4 & 1) + (x4 & 1);
  }
}

static inline int clip3(int lo, int hi, int x) { return x < lo ? lo : (x > hi ? hi : x); }

/* returns a checksum over all outputs (keeps the optimiser honest; also a parity handle for the tests) */
long long jmo_hotpath_mbs(const jmo_me_params *p, const jmo_ref *ref, const jmo_pel *curY, const jmo_pel *curU, const jmo_pel *curV,
                          int W, const short *mb_xy /*[n][2]*/, const short *preds /*[n][41][2]*/, int n, int R, const int *lambda,
                          const jmo_quant *q_luma, const jmo_quant *q_chroma, short *mv_out /*[n][41][2] or NULL*/, int *cost_out /*[n][41] or NULL*/)
{
  long long sum = 0;
  const int Wc = ref->Wc, mcw = ref->cg.mb_cr_size_x, mch = ref->cg.mb_cr_size_y;
  for (int i = 0; i < n; i++) {
    const int mbx = mb_xy[2 * i], mby = mb_xy[2 * i + 1];
    short mv[41][2]; int cost[41];
    for (int q = 0; q < 41; q++) {
      const int bt = PART[q][0], px = mbx * 16 + 4 * PART[q][1], py = mby * 16 + 4 * PART[q][2];
      const int bsx = 4 * PART[q][3], bsy = 4 * PART[q][4];
      jmo_pel orig[768];
      for (int y = 0; y < bsy; y++) memcpy(orig + y * bsx, curY + (long)(py + y) * W + px, bsx * sizeof(jmo_pel));   /* mv-search.c:607-611 */
      cost[q] = jmo_block_search_full(p, ref, orig, 1, px, py, bt, preds[(i * 41 + q) * 2], preds[(i * 41 + q) * 2 + 1], R, lambda, mv[q], 0, 0);
      sum += cost[q] + mv[q][0] * 3 + mv[q][1] * 5;
      if (mv_out) { mv_out[(i * 41 + q) * 2] = mv[q][0]; mv_out[(i * 41 + q) * 2 + 1] = mv[q][1]; }
      if (cost_out) cost_out[i * 41 + q] = cost[q];
    }
    if (!q_luma) continue;
    /* partitioning with the smallest summed motion cost (bench stand-in for the mode decision) */
    int b8mode[4], c8 = 0, mode = 1, best = cost[0];
    for (int b = 0; b < 4; b++) {
      int s[4] = { cost[5 + b], cost[9 + 2 * b] + cost[10 + 2 * b], cost[17 + 2 * b] + cost[18 + 2 * b],
                   cost[25 + 4 * b] + cost[26 + 4 * b] + cost[27 + 4 * b] + cost[28 + 4 * b] };
      int bm = 0;
      for (int k = 1; k < 4; k++) if (s[k] < s[bm]) bm = k;
      b8mode[b] = 4 + bm; c8 += s[bm];
    }
    if (cost[1] + cost[2] < best) { best = cost[1] + cost[2]; mode = 2; }
    if (cost[3] + cost[4] < best) { best = cost[3] + cost[4]; mode = 3; }
    if (c8 < best) { best = c8; mode = 8; }
    /* luma: prediction per 4x4 block + dct_4x4 */
    int m7[16][16], fadj[16][16], levels[17], runs[17];
    jmo_pel mpr[16][16], recon[16][16];
    short mv4[16][2];
    for (int y4 = 0; y4 < 4; y4++) for (int x4 = 0; x4 < 4; x4++) {
      const int pq = covering(mode, b8mode, x4, y4);
      mv4[y4 * 4 + x4][0] = mv[pq][0]; mv4[y4 * 4 + x4][1] = mv[pq][1];
      const int xq = ((mbx * 16 + 4 * x4) << 2) + JMO_PAD4 + mv[pq][0], yq = ((mby * 16 + 4 * y4) << 2) + JMO_PAD4 + mv[pq][1];
      const int xpos = clip3(0, ref->width_pad, xq >> 2), ypos = clip3(0, ref->height_pad, yq >> 2);
      const jmo_pel *src = ref->luma[(yq & 3) * 4 + (xq & 3)] + (long)ypos * ref->Wp + xpos;
      for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
        mpr[4 * y4 + y][4 * x4 + x] = src[(long)y * ref->Wp + x];
        m7[4 * y4 + y][4 * x4 + x] = curY[(long)(mby * 16 + 4 * y4 + y) * W + mbx * 16 + 4 * x4 + x] - mpr[4 * y4 + y][4 * x4 + x];
      }
    }
    for (int b = 0; b < 16; b++) {
      int cc = 0;
      const int bx = 8 * ((b >> 2) & 1) + 4 * (b & 1), by = 8 * (b >> 3) + 4 * ((b >> 1) & 1);
      sum += jmo_dct_4x4(q_luma, m7, (const jmo_pel (*)[16])mpr, bx, by, &cc, levels, runs, recon, fadj);
      sum += levels[0] + cc % 7;
    }
    for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) sum += recon[y][x];
    /* chroma */
    if (q_chroma && ref->yuv_format != JMO_YUV400) {
      const jmo_chroma_geom *g = &ref->cg;
      for (int uv = 0; uv < 2; uv++) {
        const jmo_pel *cur = uv ? curV : curU;
        int dcl[17], dcr[17], acl[8][16], acr[8][16];
        long long cbp = 0;
        memset(m7, 0, sizeof(m7)); memset(mpr, 0, sizeof(mpr));
        for (int j = 0; j < mch; j++) for (int ic = 0; ic < mcw; ic += 2) {
          const short *m = mv4[(j >> (4 - g->shift_y)) * 4 + (ic >> (4 - g->shift_x))];
          const int ii = ((ic + mbx * mcw) << g->shift_x) + JMO_PAD4 + m[0], jj = ((j + mby * mch) << g->shift_y) + JMO_PAD4 + m[1];
          const int xpos = clip3(0, ref->width_pad_cr, ii >> g->shift_x), ypos = clip3(0, ref->height_pad_cr, jj >> g->shift_y);
          const jmo_pel *src = ref->cr[uv][(jj & g->mask_y) * g->sub_x + (ii & g->mask_x)] + (long)ypos * ref->Wcp + xpos;
          for (int x = 0; x < 2; x++) {
            mpr[j][ic + x] = src[x];
            m7[j][ic + x] = cur[(long)(mby * mch + j) * Wc + mbx * mcw + ic + x] - src[x];
          }
        }
        sum += jmo_dct_chroma(q_chroma, q_chroma, ref->yuv_format, uv, 0, m7, (const jmo_pel (*)[16])mpr, dcl, dcr, acl, acr, recon, fadj, &cbp);
        sum += dcl[0] + (int)cbp;
      }
    }
  }
  return sum;
}

/* ------------------------------------------------------------------ low-complexity mode-decision costs */

/* distortion4x4 / distortion8x8, me_distortion.c:76 / :110 (ModeDecisionMetric: 0 SAD, 1 SSE through img->quad[] = the squares, 2 SATD) */
static int dist4(const int *d, int metric)
{
  int k, s = 0;
  if (metric == JMO_ERR_SATD) return jmo_hadamard_sad4x4(d);
  for (k = 0; k < 16; k++) s += metric == JMO_ERR_SSE ? d[k] * d[k] : (d[k] < 0 ? -d[k] : d[k]);
  return s;
}
static int dist8(const int *d, int metric)
{
  int k, s = 0;
  if (metric == JMO_ERR_SATD) return jmo_hadamard_sad8x8(d);
  for (k = 0; k < 64; k++) s += metric == JMO_ERR_SSE ? d[k] * d[k] : (d[k] < 0 ? -d[k] : d[k]);
  return s;
}

/* The cost pair of TransformDecision (macroblock.c:1458-1520) for one macroblock whose prediction (per 4x4 block) is `mpr`:
 * cost4x4 = sum of distortion4x4 over the sixteen 4x4 residual blocks; cost8x8 = sum over the four 8x8 blocks of
 * distortion8x8(diff64), where diff64 holds the block's four 4x4 residual blocks ONE AFTER THE OTHER (16 values each, :1496-1500)
 * and is then read as eight rows of eight (:1505) -- not the 8x8 block's raster. `proper8x8` != 0 selects the true raster instead,
 * which is what GetSkipCostMB builds (curr_diff[8][8] copied row by row, mv-search.c:1161-1176). */
void jmo_pred_costs(const jmo_pel *cur /*[16][16]*/, const jmo_pel *mpr /*[16][16]*/, int metric, int proper8x8, int *cost4x4, int *cost8x8)
{
  int b8, c4 = 0, c8 = 0;
  for (b8 = 0; b8 < 4; b8++) {
    const int mb_y = (b8 >> 1) << 3, mb_x = (b8 & 1) << 3;
    int diff64[64], raster[64], k = 0, by, bx, j, i;
    for (by = mb_y; by < mb_y + 8; by += 4)
      for (bx = mb_x; bx < mb_x + 8; bx += 4) {
        const int *dp = &diff64[k];
        for (j = by; j < by + 4; j++)
          for (i = bx; i < bx + 4; i++, k++) {
            diff64[k] = (int)cur[j * 16 + i] - (int)mpr[j * 16 + i];
            raster[(j - mb_y) * 8 + (i - mb_x)] = diff64[k];
          }
        c4 += dist4(dp, metric);
      }
    c8 += dist8(proper8x8 ? raster : diff64, metric);
  }
  *cost4x4 = c4; *cost8x8 = c8;
}
