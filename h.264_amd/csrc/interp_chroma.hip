// interp_chroma.hip -- the eighth-pel (4:2:0), 8x4 (4:2:2) or quarter-pel (4:4:4) chroma planes.
//
// Replaces getSubImagesChroma (lencod/src/img_chroma.c:374-443) and generateChroma00/01/10/XX (:34-360).
// Plane (suby, subx), k = suby*mul_y, l = subx*mul_x: w00=(8-k)(8-l), w01=(8-k)l, w10=k(8-l), w11=kl (:412-420),
// value = (w00*a + w01*b + w10*c + w11*d + 32) >> 6 with source coordinates clamped to the picture. All four
// generateChroma* cases, including the ring filled from edge-interpolated values (:230-246, :318-335), are
// that one expression (weights sum to 64). JM's loops stop one short (:63, :129): the last padded row and the
// last padded column are never written and keep calloc's zero (memalloc.c:142) -- reproduced here.
//
// Roofline: HBM write-bound, 1 byte read per sub_x*sub_y bytes written. Horizontal blends are shared across
// the sub_y vertical phases (2 multiply-adds per output sample instead of 4).
#include "jmhip_internal.h"

namespace {

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return min(max(x, lo), hi); }

template <int SUBX, int SUBY, int MULX, int MULY>
__global__ __launch_bounds__(256) void interp_chroma_kernel(const uint8_t *__restrict__ src_u, uint8_t *__restrict__ out_u,
                                                           const uint8_t *__restrict__ src_v, uint8_t *__restrict__ out_v,
                                                           int Wc, int Hc, int Wcp, int Hcp, int pad_x, int pad_y, int row4_0)
{
  const uint8_t *__restrict__ src = blockIdx.z ? src_v : src_u;      // both components in one launch
  uint8_t *__restrict__ out = blockIdx.z ? out_v : out_u;
  const int gi = (blockIdx.x * 64 + threadIdx.x) * 4;
  const int gj = (blockIdx.y + row4_0) * 4 + threadIdx.y;
  if (gi >= Wcp || gj >= Hcp - 1) return;          // last row keeps its zeros
  const uint8_t *r0 = src + (size_t)clampi(gj - pad_y, 0, Hc - 1) * Wc;
  const uint8_t *r1 = src + (size_t)clampi(gj - pad_y + 1, 0, Hc - 1) * Wc;
  // five source samples of the two rows as adjacent pairs of 16-bit lanes: all sums stay below 2^15 (8*8*255 + 32), so the
  // blends run on packed 16-bit lanes (v_pk_mul_lo_u16 / v_pk_mad_u16 / v_pk_lshrrev_b16), two samples per instruction
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  unsigned short a[5], b[5];
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const int x = clampi(gi + k - pad_x, 0, Wc - 1);
    a[k] = r0[x]; b[k] = r1[x];
  }
  const us2 a01 = {a[0], a[1]}, a12 = {a[1], a[2]}, a23 = {a[2], a[3]}, a34 = {a[3], a[4]};
  const us2 b01 = {b[0], b[1]}, b12 = {b[1], b[2]}, b23 = {b[2], b[3]}, b34 = {b[3], b[4]};
  const bool last_group = (gi + 3 == Wcp - 1);     // last padded column keeps its zero
  const size_t plane = (size_t)Wcp * Hcp;
  uint32_t *o = reinterpret_cast<uint32_t *>(out + (size_t)gj * Wcp + gi);
#pragma unroll
  for (int sx = 0; sx < SUBX; sx++) {
    const unsigned short l = (unsigned short)(sx * MULX), l8 = (unsigned short)(8 - sx * MULX);
    const us2 h0lo = a01 * l8 + a12 * l, h0hi = a23 * l8 + a34 * l;      // samples (0,1) and (2,3) of the upper row
    const us2 h1lo = b01 * l8 + b12 * l, h1hi = b23 * l8 + b34 * l;
    uint32_t *p = o + (size_t)sx * (plane / 4);                       // plane (sy, sx): stride SUBX planes per sy
#pragma unroll
    for (int sy = 0; sy < SUBY; sy++, p += (size_t)SUBX * (plane / 4)) {
      const unsigned short k = (unsigned short)(sy * MULY), k8 = (unsigned short)(8 - sy * MULY);
      const us2 lo = (h0lo * k8 + h1lo * k + (unsigned short)32) >> (unsigned short)6;
      const us2 hi = (h0hi * k8 + h1hi * k + (unsigned short)32) >> (unsigned short)6;
      uint32_t v = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi), __builtin_bit_cast(uint32_t, lo), 0x06040200u);   // the four low bytes
      if (last_group) v &= 0x00ffffffu;
      *p = v;
    }
  }
}

}  // namespace

// rows [prow0, prow1) of the padded chroma planes, widened to groups of 4; everything when prow1 <= prow0
int jm_launch_interp_chroma(jmhip_ctx *c, int ref, int prow0, int prow1)
{
  if ((c->Wcp & 3) || ((size_t)c->Wcp * c->Hcp) % 4) return jm_fail(c, JMHIP_ERR_ARG, "interp_chroma: padded width must be a multiple of 4");
  RefSlot &r = c->refs[ref];
  int g0 = 0, g1 = (c->Hcp + 3) / 4;
  if (prow1 > prow0) { g0 = (prow0 < 0 ? 0 : prow0) / 4; const int e = ((prow1 > c->Hcp ? c->Hcp : prow1) + 3) / 4; g1 = e < g1 ? e : g1; }
  if (g1 <= g0) return JMHIP_OK;
  dim3 grid((c->Wcp / 4 + 63) / 64, g1 - g0, 2), block(64, 4);
  switch (c->cfg.yuv_format) {
  case JMHIP_YUV420: interp_chroma_kernel<8, 8, 1, 1><<<grid, block, 0, c->stream>>>(r.u, r.cr_sub[0], r.v, r.cr_sub[1], c->Wc, c->Hc, c->Wcp, c->Hcp, c->cg.pad_x, c->cg.pad_y, g0); break;
  case JMHIP_YUV422: interp_chroma_kernel<8, 4, 1, 2><<<grid, block, 0, c->stream>>>(r.u, r.cr_sub[0], r.v, r.cr_sub[1], c->Wc, c->Hc, c->Wcp, c->Hcp, c->cg.pad_x, c->cg.pad_y, g0); break;
  case JMHIP_YUV444: interp_chroma_kernel<4, 4, 2, 2><<<grid, block, 0, c->stream>>>(r.u, r.cr_sub[0], r.v, r.cr_sub[1], c->Wc, c->Hc, c->Wcp, c->Hcp, c->cg.pad_x, c->cg.pad_y, g0); break;
  default: return jm_fail(c, JMHIP_ERR_ARG, "interp_chroma: no chroma");
  }
  JM_HIP_CHECK(c, hipGetLastError());
  return JMHIP_OK;
}
