// frame_common.h -- what the frame-stage translation units (frame.hip: MC / finalize / entry points, tq.hip: the fused 4:2:0 kernel) share.
#pragma once
#include "jmhip_internal.h"

namespace {

struct FrameDev {
  int W, H, Wp, Hp, Wc, Hc, Wcp, Hcp, mbw;
  int yuv, shift_x, shift_y, mask_x, mask_y, sub_x, mb_cw, mb_ch;
  const uint8_t *cur_y, *cur_u, *cur_v;
  const uint8_t *const *ref_sub, *const *ref_cb, *const *ref_cr;
  uint8_t *rec_y, *rec_u, *rec_v;
  uint8_t *pred_y, *pred_u, *pred_v;     // the prediction picture (img->mpr of every macroblock), or NULL: jmhip_frame_keep_prediction (fused 4:2:0 stage)
  // chroma prediction without the eighth-pel planes: the sample a plane WOULD hold, computed from the integer chroma picture
  // (reference slots 0..7; see mc_kernel)
  int fly, mul_x, mul_y, pad_cx, pad_cy;
  const uint8_t *ref_u[8], *ref_v[8];
  // explicit weighted prediction of P slices (LumaPrediction macroblock.c:880-914, ChromaPrediction4x4 :1895-1903), per reference SLOT
  int wp_on, wp_lround, wp_ldenom, wp_cround, wp_cdenom;
  short wp_w[16][3], wp_o[16][3];
  const int8_t *blk_ref;          // [n][4]: reference slot of each 8x8 block (NULL: the macroblock's one reference, jmhip_me_mb.ref)
  // B macroblocks (jmhip_frame_bipred_set): second list per macroblock, weights of the second list by reference slot
  const jmhip_mb_bipred *bi;
  short bw0[4][4][3], bw1[4][4][3], bu1[4][3], bo1[4][3];
};

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return min(max(x, lo), hi); }

// partition index (me_common.h table order) that covers luma 4x4 block (x4,y4) for a macroblock mode
__device__ __forceinline__ int covering_partition(const jmhip_mb_mode &m, int x4, int y4)
{
  const int b8 = 2 * (y4 >> 1) + (x4 >> 1);
  switch (m.mode) {
  case 1: return 0;
  case 2: return 1 + (y4 >> 1);
  case 3: return 3 + (x4 >> 1);
  default:
    switch (m.b8mode[b8]) {
    case 4: return 5 + b8;
    case 5: return 9 + 2 * b8 + (y4 & 1);
    case 6: return 17 + 2 * b8 + (x4 & 1);
    default: return 25 + 4 * b8 + 2 * (y4 & 1) + (x4 & 1);
    }
  }
}

// one predicted sample from its list-0 / list-1 fetches p0 / p1 (LumaPrediction macroblock.c:880-940, ChromaPrediction4x4 :1768-1830);
// comp 0 luma, 1 Cb, 2 Cr; s0 / s1 reference slots. The bi-predictive chroma mix uses the luma denominator, as JM does (:1781).
__device__ __forceinline__ int mix_pred(const FrameDev &F, int pdir, int s0, int s1, int comp, int p0, int p1)
{
  if (!F.wp_on) return pdir == 2 ? (p0 + p1 + 1) >> 1 : (pdir ? p1 : p0);
  const int rnd = comp ? F.wp_cround : F.wp_lround, den = comp ? F.wp_cdenom : F.wp_ldenom;
  if (pdir == 2)
    return clampi((((int)F.bw0[s0][s1][comp] * p0 + (int)F.bw1[s0][s1][comp] * p1 + 2 * rnd) >> (F.wp_ldenom + 1)) + (((int)F.wp_o[s0][comp] + (int)F.bo1[s1][comp] + 1) >> 1), 0, 255);
  if (pdir == 0) return clampi((((int)F.wp_w[s0][comp] * p0 + rnd) >> den) + F.wp_o[s0][comp], 0, 255);
  return clampi((((int)F.bu1[s1][comp] * p1 + rnd) >> den) + F.bo1[s1][comp], 0, 255);
}

// chroma sample pair of one component at eighth-pel position (ii, jj) [padded units] of reference `slot`: from the eighth-pel planes or,
// when they were not built, computed from the integer chroma picture (identical values; see mc_kernel). 4:2:0 / 4:2:2 geometry from F.
__device__ __forceinline__ void chroma_pair(const FrameDev &F, int slot, int uv, int ii, int jj, int *p0, int *p1)
{
  const int xpos = clampi(ii >> F.shift_x, 0, F.Wcp - 1 - F.mb_cw), ypos = clampi(jj >> F.shift_y, 0, F.Hcp - 1 - F.mb_ch);      // mbuffer.c:425-426
  if (F.fly) {
    const uint8_t *pic = uv ? F.ref_v[slot] : F.ref_u[slot];
    const int k = (jj & F.mask_y) * F.mul_y, l = (ii & F.mask_x) * F.mul_x;
    const uint8_t *r0 = pic + (size_t)clampi(ypos - F.pad_cy, 0, F.Hc - 1) * F.Wc;
    const uint8_t *r1 = pic + (size_t)clampi(ypos - F.pad_cy + 1, 0, F.Hc - 1) * F.Wc;
    const int xa = clampi(xpos - F.pad_cx, 0, F.Wc - 1), xb = clampi(xpos - F.pad_cx + 1, 0, F.Wc - 1), xc = clampi(xpos - F.pad_cx + 2, 0, F.Wc - 1);
    const int a0 = r0[xa], a1 = r0[xb], a2 = r0[xc], b0 = r1[xa], b1 = r1[xb], b2 = r1[xc];
    const int h00 = a0 * (8 - l) + a1 * l, h01 = a1 * (8 - l) + a2 * l, h10 = b0 * (8 - l) + b1 * l, h11 = b1 * (8 - l) + b2 * l;
    *p0 = (h00 * (8 - k) + h10 * k + 32) >> 6; *p1 = (h01 * (8 - k) + h11 * k + 32) >> 6;
  } else {
    const uint8_t *planes = (uv ? F.ref_cr : F.ref_cb)[slot];
    const uint8_t *src = planes + (size_t)((jj & F.mask_y) * F.sub_x + (ii & F.mask_x)) * F.Wcp * F.Hcp + (size_t)ypos * F.Wcp + xpos;
    *p0 = src[0]; *p1 = src[1];
  }
}

__device__ __forceinline__ uint32_t fetch4(const uint8_t *p)
{
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  const uint32_t *q = reinterpret_cast<const uint32_t *>(a & ~uintptr_t(3));
  return __builtin_amdgcn_alignbyte(q[1], q[0], (unsigned)(a & 3));
}

// four luma samples of one row of a 4x4 block predicted from reference `slot` at quarter-pel (xq, yq) [padded units], row rr; ox4 / oy4:
// the block's offset inside its 8x8 block when the prediction is formed per 8x8 block (8x8 transform), else 0
__device__ __forceinline__ uint32_t luma_row4(const FrameDev &F, int slot, int xq, int yq, int ox4, int oy4, int rr)
{
  const int xpos = clampi(xq >> 2, 0, F.Wp - 1 - 16) + ox4, ypos = clampi(yq >> 2, 0, F.Hp - 1 - 16) + oy4;   // UMVLine4X, refbuf.c:37
  return fetch4(F.ref_sub[slot] + (size_t)((yq & 3) * 4 + (xq & 3)) * F.Wp * F.Hp + (size_t)(ypos + rr) * F.Wp + xpos);
}

}  // namespace

// Device-resident result record of one macroblock from the fused 4:2:0 frame stage (tq.hip frame_fused_kernel): the ABI's jmhip_mb_residual
// (include/jmhip.h), which jmhip_residual_records_download hands out as it is and jmhip_residual_download expands into jmhip_tq_result structs --
// the same fields the separate TQ kernels write, 2.4 KB instead of 17 KB.
typedef jmhip_mb_residual JmMbRes;
static_assert(sizeof(JmMbRes) % 16 == 0, "records are copied out of LDS as 16-byte pieces");

int jm_launch_frame_fused(jmhip_ctx *c, const void *frame_dev, const void *mbs, const void *me, const void *modes_in, void *modes_out,
                          const void *quants, void *records, void *coded, int n);
