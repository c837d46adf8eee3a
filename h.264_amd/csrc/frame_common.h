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
  // chroma prediction without the eighth-pel planes: the sample a plane WOULD hold, computed from the integer chroma picture
  // (reference slots 0..3; see mc_kernel)
  int fly, mul_x, mul_y, pad_cx, pad_cy;
  const uint8_t *ref_u[4], *ref_v[4];
  // explicit weighted prediction of P slices (LumaPrediction macroblock.c:880-914, ChromaPrediction4x4 :1895-1903), per reference SLOT
  int wp_on, wp_lround, wp_ldenom, wp_cround, wp_cdenom;
  short wp_w[16][3], wp_o[16][3];
  const int8_t *blk_ref;          // [n][4]: reference slot of each 8x8 block (NULL: the macroblock's one reference, jmhip_me_mb.ref)
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

__device__ __forceinline__ uint32_t fetch4(const uint8_t *p)
{
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  const uint32_t *q = reinterpret_cast<const uint32_t *>(a & ~uintptr_t(3));
  return __builtin_amdgcn_alignbyte(q[1], q[0], (unsigned)(a & 3));
}

}  // namespace

// Device-resident result record of one macroblock from the fused 4:2:0 frame stage (tq.hip frame_fused_kernel). jmhip_residual_download
// expands it into the ABI's jmhip_tq_result structs (include/jmhip.h): the same fields the separate TQ kernels write, 2.4 KB instead of 17 KB.
struct JmMbRes {
  int16_t lev[24][16];           // (level) lists in scan order: luma blocks 0..15 (JM order b8*4+b4), Cb 16..19, Cr 20..23 (AC)
  uint8_t run[24][16];
  uint8_t cnt[24];               // entries of each list; the ABI's 0 terminator follows them
  int16_t dc_lev[2][4];          // chroma DC lists
  uint8_t dc_run[2][4];
  uint8_t dc_cnt[2];
  uint8_t ac_zeroed[2];          // _CHROMA_COEFF_COST_ thresholding hit: the AC levels of the component read 0, the runs stay (block.c:1384-1410)
  uint8_t pad0[4];
  int32_t coeff_cost[16];        // luma, per 4x4 block
  int32_t ret[2];                // dct_chroma's cr_cbp per component
  uint16_t nonzero;              // luma: bit blk = dct_4x4's return value
  uint16_t pad1[3];
  int64_t cbp_blk[2], cbp_clear[2];
  int16_t fadj_y[16][16];        // adaptive rounding only
  int16_t fadj_c[2][8][8];
  uint8_t recon_y[16][16];       // the transform path's reconstruction (before the coefficient-cost decision picks it or the prediction)
  uint8_t recon_c[2][8][8];
  uint8_t pad2[8];
};
static_assert(sizeof(JmMbRes) % 16 == 0, "records are copied out of LDS as 16-byte pieces");

int jm_launch_frame_fused(jmhip_ctx *c, const void *frame_dev, const void *mbs, const void *me, const void *modes_in, void *modes_out,
                          const void *quants, void *records, void *coded, int n);
