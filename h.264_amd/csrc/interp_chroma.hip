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
__global__ __launch_bounds__(256) void interp_chroma_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ out,
                                                           int Wc, int Hc, int Wcp, int Hcp, int pad_x, int pad_y, int row4_0)
{
  const int gi = (blockIdx.x * 64 + threadIdx.x) * 4;
  const int gj = (blockIdx.y + row4_0) * 4 + threadIdx.y;
  if (gi >= Wcp || gj >= Hcp - 1) return;          // last row keeps its zeros
  const uint8_t *r0 = src + (size_t)clampi(gj - pad_y, 0, Hc - 1) * Wc;
  const uint8_t *r1 = src + (size_t)clampi(gj - pad_y + 1, 0, Hc - 1) * Wc;
  int a[5], b[5];
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const int x = clampi(gi + k - pad_x, 0, Wc - 1);
    a[k] = r0[x]; b[k] = r1[x];
  }
  const bool last_group = (gi + 3 == Wcp - 1);     // last padded column keeps its zero
  const size_t plane = (size_t)Wcp * Hcp;
  uint32_t *o = reinterpret_cast<uint32_t *>(out + (size_t)gj * Wcp + gi);
#pragma unroll
  for (int sx = 0; sx < SUBX; sx++) {
    const int l = sx * MULX;
    int h0[4], h1[4];
#pragma unroll
    for (int x = 0; x < 4; x++) { h0[x] = (8 - l) * a[x] + l * a[x + 1]; h1[x] = (8 - l) * b[x] + l * b[x + 1]; }
#pragma unroll
    for (int sy = 0; sy < SUBY; sy++) {
      const int k = sy * MULY;
      uint32_t v = 0;
#pragma unroll
      for (int x = 0; x < 4; x++) v |= (uint32_t)(((8 - k) * h0[x] + k * h1[x] + 32) >> 6) << (8 * x);
      if (last_group) v &= 0x00ffffffu;
      o[(size_t)(sy * SUBX + sx) * (plane / 4)] = v;
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
  dim3 grid((c->Wcp / 4 + 63) / 64, g1 - g0), block(64, 4);
  for (int uv = 0; uv < 2; uv++) {
    const uint8_t *src = uv ? r.v : r.u;
    uint8_t *dst = r.cr_sub[uv];
    switch (c->cfg.yuv_format) {
    case JMHIP_YUV420: interp_chroma_kernel<8, 8, 1, 1><<<grid, block, 0, c->stream>>>(src, dst, c->Wc, c->Hc, c->Wcp, c->Hcp, c->cg.pad_x, c->cg.pad_y, g0); break;
    case JMHIP_YUV422: interp_chroma_kernel<8, 4, 1, 2><<<grid, block, 0, c->stream>>>(src, dst, c->Wc, c->Hc, c->Wcp, c->Hcp, c->cg.pad_x, c->cg.pad_y, g0); break;
    case JMHIP_YUV444: interp_chroma_kernel<4, 4, 2, 2><<<grid, block, 0, c->stream>>>(src, dst, c->Wc, c->Hc, c->Wcp, c->Hcp, c->cg.pad_x, c->cg.pad_y, g0); break;
    default: return jm_fail(c, JMHIP_ERR_ARG, "interp_chroma: no chroma");
    }
    JM_HIP_CHECK(c, hipGetLastError());
  }
  return JMHIP_OK;
}
