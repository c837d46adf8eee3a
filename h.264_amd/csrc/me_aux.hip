// me_aux.hip -- the distortion services round the searches:
//   distortion_kernel    computeSAD / SADWP / SATD / SATDWP for arbitrary candidate lists           (jmhip_distortion_batch)
//   surface_kernel<K>    every integer displacement in early-exit granularity, for EPZS / UMHexagonS (jmhip_distortion_surface)
//   bipred_kernel        FullPelBlockMotionBiPred / SubPelBlockSearchBiPred                          (jmhip_bipred_search)
//   predcost_kernel      TransformDecision / GetSkipCostMB / BIDPartitionCost residual costs         (jmhip_pred_cost_batch)
#include "me_common.h"

// ------------------------------------------------------------------------------------------------ distortion batch

namespace {

__device__ __forceinline__ int wp_pel(const jmhip_dist_job &j, int v)
{
  if (!j.wp) return v;
  const int w = ((j.weight * v + j.wp_round) >> j.wp_denom) + j.offset;     // me_distortion.c:431
  return min(max(w, 0), 255);
}

// computeSAD / computeSADWP / computeSATD / computeSATDWP (me_distortion.c:351, :413, :657, :734) for one candidate
// per lane, full evaluation. One lane per job: this is the primitive for host-driven searches (EPZS, UMHexagonS),
// whose candidate lists are short and data dependent.
__global__ __launch_bounds__(64) void distortion_kernel(MeDev P, const jmhip_dist_job *__restrict__ jobs, int n, int32_t *__restrict__ out)
{
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const jmhip_dist_job j = jobs[i];
  const uint8_t *sub = P.ref_sub[j.ref];
  const size_t plane = (size_t)P.Wp * P.Hp;
  const int width_pad = P.Wp - 1 - 16, height_pad = P.Hp - 1 - 16;
  int total = 0;
  if (j.use_satd == 0) {
    int xpos = j.cand_x >> 2, ypos = j.cand_y >> 2;
    if (j.umv) { xpos = clampi(xpos, 0, width_pad); ypos = clampi(ypos, 0, height_pad); }
    const uint8_t *rp = sub + (size_t)((j.cand_y & 3) * 4 + (j.cand_x & 3)) * plane + (size_t)ypos * P.Wp + xpos;
    for (int y = 0; y < j.bsy; y++) {
      const uint8_t *cp = P.cur + (size_t)(j.pic_y + y) * P.W + j.pic_x;
      for (int x = 0; x < j.bsx; x++) total += iabs((int)cp[x] - wp_pel(j, rp[(size_t)y * P.Wp + x]));
    }
  } else {
    const int bs = j.use_satd == 2 ? 8 : 4;
    for (int by = 0; by < j.bsy; by += bs)
      for (int bx = 0; bx < j.bsx; bx += bs) {
        const int xq = j.cand_x + (bx << 2), yq = j.cand_y + (by << 2);
        int xpos = xq >> 2, ypos = yq >> 2;
        if (j.umv) { xpos = clampi(xpos, 0, width_pad); ypos = clampi(ypos, 0, height_pad); }   // per sub-block, :678
        const uint8_t *rp = sub + (size_t)((yq & 3) * 4 + (xq & 3)) * plane + (size_t)ypos * P.Wp + xpos;
        if (bs == 4) {
          int d[4][4];
#pragma unroll
          for (int r = 0; r < 4; r++)
#pragma unroll
            for (int x = 0; x < 4; x++)
              d[r][x] = (int)P.cur[(size_t)(j.pic_y + by + r) * P.W + j.pic_x + bx + x] - wp_pel(j, rp[(size_t)r * P.Wp + x]);
          total += satd4x4(d);
        } else {
          int m2[8][8], s = 0;
          for (int r = 0; r < 8; r++) {
            int row[8];
#pragma unroll
            for (int x = 0; x < 8; x++) row[x] = (int)P.cur[(size_t)(j.pic_y + by + r) * P.W + j.pic_x + bx + x] - wp_pel(j, rp[(size_t)r * P.Wp + x]);
            had8(row);
#pragma unroll
            for (int x = 0; x < 8; x++) m2[r][x] = row[x];
          }
          for (int x = 0; x < 8; x++) {
            int col[8];
#pragma unroll
            for (int r = 0; r < 8; r++) col[r] = m2[r][x];
            had8(col);
#pragma unroll
            for (int r = 0; r < 8; r++) s += iabs(col[r]);
          }
          total += (s + 2) >> 2;
        }
      }
  }
  out[i] = total;
}

}  // namespace

extern "C" int jmhip_distortion_batch(jmhip_ctx *c, const jmhip_dist_job *jobs, int n, int32_t *out)
{
  if (!c || !jobs || !out || n <= 0) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_distortion_batch: NULL/empty arguments") : JMHIP_ERR_ARG;
  if (!c->has_cur) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_distortion_batch: current picture not uploaded");
  for (int i = 0; i < n; i++) {
    const jmhip_dist_job &j = jobs[i];
    if (j.ref < 0 || j.ref >= (int)c->refs.size() || !c->refs[j.ref].has_luma_sub) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_distortion_batch: reference planes not built");
    if ((j.bsx != 4 && j.bsx != 8 && j.bsx != 16) || (j.bsy != 4 && j.bsy != 8 && j.bsy != 16)) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_distortion_batch: block size");
    if (j.pic_x < 0 || j.pic_y < 0 || j.pic_x + j.bsx > c->W || j.pic_y + j.bsy > c->H) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_distortion_batch: block outside the picture");
    if (j.use_satd < 0 || j.use_satd > 2 || (j.use_satd == 2 && ((j.bsx | j.bsy) & 7))) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_distortion_batch: metric");
    // FAST_ACCESS is only legal where JM would choose it: the whole block inside the padded plane
    const int xp = j.cand_x >> 2, yp = j.cand_y >> 2;
    if (!j.umv && (xp < 0 || yp < 0 || xp + j.bsx > c->Wp || yp + j.bsy > c->Hp)) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_distortion_batch: FAST access outside the padded plane");
    if (j.wp && (j.wp_denom < 0 || j.wp_denom > 15)) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_distortion_batch: weight denominator");
  }
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  int rc = jm_ensure_ref_table(c);
  if (rc) return rc;
  void *dj = nullptr, *dout = nullptr;
  if (hipMalloc(&dj, sizeof(jmhip_dist_job) * (size_t)n) != hipSuccess || hipMalloc(&dout, sizeof(int32_t) * (size_t)n) != hipSuccess) {
    (void)hipFree(dj); return jm_fail(c, JMHIP_ERR_NOMEM, "distortion batch arrays");
  }
  MeDev P{};
  P.W = c->W; P.H = c->H; P.Wp = c->Wp; P.Hp = c->Hp; P.cur = c->cur_y;
  P.ref_y = reinterpret_cast<const uint8_t *const *>(c->ref_ptrs_dev);
  P.ref_sub = reinterpret_cast<const uint8_t *const *>(c->ref_ptrs_dev) + 32;
  hipError_t e = hipMemcpyAsync(dj, jobs, sizeof(jmhip_dist_job) * (size_t)n, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) { distortion_kernel<<<(n + 63) / 64, 64, 0, c->stream>>>(P, (const jmhip_dist_job *)dj, n, (int32_t *)dout); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipMemcpyAsync(out, dout, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(dj); (void)hipFree(dout);
  if (e != hipSuccess) { c->err = std::string("jmhip_distortion_batch: ") + hipGetErrorString(e); return JMHIP_ERR_DEVICE; }
  return JMHIP_OK;
}

// ------------------------------------------------------------------------------------------------ distortion surfaces

namespace {

// One workgroup per (macroblock, reference): the (2R+1)^2 integer displacements around (cx, cy). The window is staged in
// LDS with per-sample clamping; lane <-> displacement.
template <int KIND>
__global__ __launch_bounds__(256) void surface_kernel(MeDev P, const jmhip_surface_job *__restrict__ jobs, uint16_t *__restrict__ out)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ __attribute__((aligned(16))) uint8_t s_cur[16][16];
  __shared__ __attribute__((aligned(16))) uint32_t s_c16[16][8];
  const jmhip_surface_job job = jobs[blockIdx.x];
  const int tid = threadIdx.x, R = job.R, UW = 2 * R + 1;
  const int pitch = (UW + 15 + 3 + 4) & ~3;
  const int bx = job.mb_x * 16 + job.cx - R, by = job.mb_y * 16 + job.cy - R;
  const uint8_t *ref = P.ref_y[job.ref];
  for (int d = tid; d < (pitch >> 2) * (UW + 15); d += 256) {
    const int y = d / (pitch >> 2), xw = d - y * (pitch >> 2);
    const uint8_t *row = ref + (size_t)clampi(by + y, 0, P.H - 1) * P.W;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) v |= (uint32_t)row[clampi(bx + xw * 4 + k, 0, P.W - 1)] << (8 * k);
    *reinterpret_cast<uint32_t *>(smem + (size_t)y * pitch + xw * 4) = v;
  }
  if (tid < 64) {
    const int r = tid >> 2, k = tid & 3;
    const uint32_t v = *reinterpret_cast<const uint32_t *>(P.cur + (size_t)(job.mb_y * 16 + r) * P.W + job.mb_x * 16 + k * 4);
    *reinterpret_cast<uint32_t *>(&s_cur[r][k * 4]) = v;
    const uint32_t bias = (r & 3) ? 0u : 0x80000000u;
    s_c16[r][2 * k] = __builtin_amdgcn_perm(0u, v, 0x0c010c00u) + bias;
    s_c16[r][2 * k + 1] = __builtin_amdgcn_perm(0u, v, 0x0c030c02u);
  }
  __syncthreads();
  constexpr int NV = KIND == JMHIP_SURFACE_SAD_ROWS ? 64 : 20;
  uint16_t *o = out + (size_t)blockIdx.x * UW * UW * NV;
  for (int c = tid; c < UW * UW; c += 256) {
    const int ay = c / UW, ax = c - ay * UW;
    const uint8_t *wrow = smem + (size_t)ay * pitch + (ax & ~3);
    const unsigned sh = ax & 3;
    uint32_t rr[16][4];                              // the 16x16 reference block at this displacement, packed bytes
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const uint32_t *wp = reinterpret_cast<const uint32_t *>(wrow + (size_t)r * pitch);
      const uint32_t d0 = wp[0], d1 = wp[1], d2 = wp[2], d3 = wp[3], d4 = wp[4];
      rr[r][0] = __builtin_amdgcn_alignbyte(d1, d0, sh); rr[r][1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
      rr[r][2] = __builtin_amdgcn_alignbyte(d3, d2, sh); rr[r][3] = __builtin_amdgcn_alignbyte(d4, d3, sh);
    }
    if (job.wp) {                                    // weighted reference samples, me_distortion.c:431
#pragma unroll
      for (int r = 0; r < 16; r++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
          uint32_t v = 0;
#pragma unroll
          for (int b = 0; b < 4; b++) {
            const int s = (int)((rr[r][k] >> (8 * b)) & 255);
            v |= (uint32_t)min(max(((job.weight * s + job.wp_round) >> job.wp_denom) + job.offset, 0), 255) << (8 * b);
          }
          rr[r][k] = v;
        }
    }
    uint32_t *o32 = reinterpret_cast<uint32_t *>(o + (size_t)c * NV);
    if (KIND == JMHIP_SURFACE_SAD_ROWS) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const uint32_t *cw = reinterpret_cast<const uint32_t *>(&s_cur[r][0]);
        const uint32_t s0 = __builtin_amdgcn_sad_u8(rr[r][0], cw[0], 0u), s1 = __builtin_amdgcn_sad_u8(rr[r][1], cw[1], 0u);
        const uint32_t s2 = __builtin_amdgcn_sad_u8(rr[r][2], cw[2], 0u), s3 = __builtin_amdgcn_sad_u8(rr[r][3], cw[3], 0u);
        o32[2 * r] = s0 | (s1 << 16); o32[2 * r + 1] = s2 | (s3 << 16);
      }
    } else {
      uint32_t v4[16];
#pragma unroll
      for (int b = 0; b < 16; b++) {
        const int y0 = (b >> 2) * 4, k = b & 3;
        uint32_t c01[4], c23[4], rf[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { c01[r] = s_c16[y0 + r][2 * k]; c23[r] = s_c16[y0 + r][2 * k + 1]; rf[r] = rr[y0 + r][k]; }
        v4[b] = (uint32_t)satd4x4_packed(c01, c23, rf);
      }
#pragma unroll
      for (int b = 0; b < 8; b++) o32[b] = v4[2 * b] | (v4[2 * b + 1] << 16);
      uint32_t v8[4];
#pragma unroll 1
      for (int b = 0; b < 4; b++) {
        const int y0 = (b >> 1) * 8, k = (b & 1) * 2;
        int m2[8][8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
          const uint32_t c0 = *reinterpret_cast<const uint32_t *>(&s_cur[y0 + r][4 * k]), c1 = *reinterpret_cast<const uint32_t *>(&s_cur[y0 + r][4 * k + 4]);
          const uint32_t lo = b & 1 ? rr[y0 + r][2] : rr[y0 + r][0], hi = b & 1 ? rr[y0 + r][3] : rr[y0 + r][1];
          int row[8];
#pragma unroll
          for (int x = 0; x < 4; x++) { row[x] = (int)((c0 >> (8 * x)) & 255) - (int)((lo >> (8 * x)) & 255); row[4 + x] = (int)((c1 >> (8 * x)) & 255) - (int)((hi >> (8 * x)) & 255); }
          had8(row);
#pragma unroll
          for (int x = 0; x < 8; x++) m2[r][x] = row[x];
        }
        int s = 0;
#pragma unroll
        for (int x = 0; x < 8; x++) {
          int col[8];
#pragma unroll
          for (int r = 0; r < 8; r++) col[r] = m2[r][x];
          had8(col);
#pragma unroll
          for (int r = 0; r < 8; r++) s += iabs(col[r]);
        }
        v8[b] = (uint32_t)((s + 2) >> 2);
      }
      o32[8] = v8[0] | (v8[1] << 16); o32[9] = v8[2] | (v8[3] << 16);
    }
  }
}

}  // namespace

extern "C" int jmhip_distortion_surface(jmhip_ctx *c, int kind, const jmhip_surface_job *jobs, int n, uint16_t *out)
{
  if (!c || !jobs || !out || n <= 0) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_distortion_surface: NULL/empty arguments") : JMHIP_ERR_ARG;
  if (kind != JMHIP_SURFACE_SAD_ROWS && kind != JMHIP_SURFACE_SATD_BLOCKS) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_distortion_surface: kind");
  if (!c->has_cur) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_distortion_surface: current picture not uploaded");
  const int R = jobs[0].R;
  for (int i = 0; i < n; i++) {
    const jmhip_surface_job &j = jobs[i];
    if (j.R != R || R < 0 || R > 64) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_distortion_surface: one range (0..64) per call");
    if (j.mb_x < 0 || j.mb_x >= c->mbw || j.mb_y < 0 || j.mb_y >= c->mbh) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_distortion_surface: macroblock outside the picture");
    if (j.ref < 0 || j.ref >= (int)c->refs.size() || !c->refs[j.ref].has_pic) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_distortion_surface: reference slot not uploaded");
    if (j.cx < -4096 || j.cx > 4096 || j.cy < -4096 || j.cy > 4096) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_distortion_surface: centre out of range");
    if (j.wp && (j.wp_denom < 0 || j.wp_denom > 15)) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_distortion_surface: weight denominator");
  }
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  int rc = jm_ensure_ref_table(c);
  if (rc) return rc;
  const int UW = 2 * R + 1, nv = kind == JMHIP_SURFACE_SAD_ROWS ? 64 : 20;
  const size_t bytes = (size_t)n * UW * UW * nv * sizeof(uint16_t);
  const int pitch = (UW + 15 + 3 + 4) & ~3;
  const size_t lds = (size_t)pitch * (UW + 15) + 16;
  if (lds > 48 * 1024) return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_distortion_surface: range too large for one LDS window");
  if (c->surf_cap < bytes || c->surf_jobs_cap < (size_t)n) {
    if (c->surf_dev) JM_HIP_CHECK(c, hipFree(c->surf_dev));
    if (c->surf_jobs_dev) JM_HIP_CHECK(c, hipFree(c->surf_jobs_dev));
    c->surf_dev = c->surf_jobs_dev = nullptr; c->surf_cap = 0; c->surf_jobs_cap = 0;
    if (hipMalloc(&c->surf_dev, bytes) != hipSuccess || hipMalloc(&c->surf_jobs_dev, sizeof(jmhip_surface_job) * (size_t)n) != hipSuccess)
      return jm_fail(c, JMHIP_ERR_NOMEM, "distortion surface arrays");
    c->surf_cap = bytes; c->surf_jobs_cap = (size_t)n;
  }
  MeDev P{};
  P.W = c->W; P.H = c->H; P.Wp = c->Wp; P.Hp = c->Hp; P.cur = c->cur_y;
  P.ref_y = reinterpret_cast<const uint8_t *const *>(c->ref_ptrs_dev);
  JM_HIP_CHECK(c, hipMemcpyAsync(c->surf_jobs_dev, jobs, sizeof(jmhip_surface_job) * (size_t)n, hipMemcpyHostToDevice, c->stream));
  if (kind == JMHIP_SURFACE_SAD_ROWS) surface_kernel<JMHIP_SURFACE_SAD_ROWS><<<n, 256, lds, c->stream>>>(P, (const jmhip_surface_job *)c->surf_jobs_dev, (uint16_t *)c->surf_dev);
  else surface_kernel<JMHIP_SURFACE_SATD_BLOCKS><<<n, 256, lds, c->stream>>>(P, (const jmhip_surface_job *)c->surf_jobs_dev, (uint16_t *)c->surf_dev);
  JM_HIP_CHECK(c, hipGetLastError());
  JM_HIP_CHECK(c, hipMemcpyAsync(out, c->surf_dev, bytes, hipMemcpyDeviceToHost, c->stream));
  JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  return JMHIP_OK;
}

// ------------------------------------------------------------------------------------------------ bi-predictive search

namespace {

struct BiDev {
  int W, H, Wp, Hp, lam_f, lam_h, lam_q, t8x8, wp, w1, w2, off, rnd, den;
  const uint8_t *cur;
  const uint8_t *const *ref_y;
  const uint8_t *const *ref_sub;
};

// the bi-predicted sample quartet: (a + b + 1) >> 1 is v_lerp_u8 with an all-ones selector; weights go sample by sample
__device__ __forceinline__ uint32_t bipel4(const BiDev &B, uint32_t a, uint32_t b)
{
  if (!B.wp) return __builtin_amdgcn_lerp(a, b, 0x01010101u);                     // me_distortion.c:504
  uint32_t r = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int p1 = (a >> (8 * k)) & 255, p2 = (b >> (8 * k)) & 255;
    const int v = ((B.w1 * p1 + B.w2 * p2 + 2 * B.rnd) >> (B.den + 1)) + B.off;    // :583
    r |= (uint32_t)min(max(v, 0), 255) << (8 * k);
  }
  return r;
}

__global__ __launch_bounds__(256) void bipred_kernel(BiDev B, const jmhip_bipred_job *__restrict__ jobs, jmhip_bipred_result *__restrict__ res)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ __attribute__((aligned(16))) uint8_t s_cur[16][16];
  __shared__ __attribute__((aligned(16))) uint32_t s_c16[16][8];
  __shared__ uint32_t s_fix[16][4];                 // stage 0: the fixed 16x16 block of picture 1
  __shared__ unsigned s_best;
  __shared__ int s_satd[9], s_mv[2], s_min;
  const jmhip_bipred_job job = jobs[blockIdx.x];
  const int tid = threadIdx.x;
  const int ox = job.mb_x * 16, oy = job.mb_y * 16;
  if (tid < 64) {
    const int r = tid >> 2, k = tid & 3;
    const uint32_t v = *reinterpret_cast<const uint32_t *>(B.cur + (size_t)(oy + r) * B.W + ox + k * 4);
    *reinterpret_cast<uint32_t *>(&s_cur[r][k * 4]) = v;
    const uint32_t bias = (r & 3) ? 0u : 0x80000000u;
    s_c16[r][2 * k] = __builtin_amdgcn_perm(0u, v, 0x0c010c00u) + bias;
    s_c16[r][2 * k + 1] = __builtin_amdgcn_perm(0u, v, 0x0c030c02u);
  }
  if (tid == 0) s_best = 0xffffffffu;

  if (job.stage == 0) {
    // ---------------- FullPelBlockMotionBiPred: integer planes, per-sample clamp == UMV origin clamp on the 20-pel ring
    const int R = job.search_range, UW = 2 * R + 1;
    const int pitch = (UW + 15 + 3 + 4) & ~3;
    const uint8_t *ref1 = B.ref_y[job.ref1], *ref2 = B.ref_y[job.ref2];
    const int bx = ox + job.mv[0] - R, by = oy + job.mv[1] - R;
    for (int d = tid; d < (pitch >> 2) * (UW + 15); d += 256) {
      const int y = d / (pitch >> 2), xw = d - y * (pitch >> 2);
      const uint8_t *row = ref2 + (size_t)clampi(by + y, 0, B.H - 1) * B.W;
      uint32_t v = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) v |= (uint32_t)row[clampi(bx + xw * 4 + k, 0, B.W - 1)] << (8 * k);
      *reinterpret_cast<uint32_t *>(smem + (size_t)y * pitch + xw * 4) = v;
    }
    if (tid < 64) {
      const int r = tid >> 2, k = tid & 3;
      const uint8_t *row = ref1 + (size_t)clampi(oy + job.s_mv[1] + r, 0, B.H - 1) * B.W;
      uint32_t v = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) v |= (uint32_t)row[clampi(ox + job.s_mv[0] + k * 4 + j, 0, B.W - 1)] << (8 * j);
      s_fix[r][k] = v;
    }
    __syncthreads();
    const int c1 = mv_cost(B.lam_f, 4 * job.s_mv[0] - job.pred1[0], 4 * job.s_mv[1] - job.pred1[1]);
    unsigned best = 0xffffffffu;
    for (int c = tid; c < UW * UW; c += 256) {
      const int ay = c / UW, ax = c - ay * UW;
      const uint8_t *wrow = smem + (size_t)ay * pitch + (ax & ~3);
      const unsigned sh = ax & 3;
      unsigned sad = 0;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const uint32_t *wp = reinterpret_cast<const uint32_t *>(wrow + (size_t)r * pitch);
        const uint32_t d0 = wp[0], d1 = wp[1], d2 = wp[2], d3 = wp[3], d4 = wp[4];
        const uint32_t *cw = reinterpret_cast<const uint32_t *>(&s_cur[r][0]);
        sad = __builtin_amdgcn_sad_u8(bipel4(B, s_fix[r][0], __builtin_amdgcn_alignbyte(d1, d0, sh)), cw[0], sad);
        sad = __builtin_amdgcn_sad_u8(bipel4(B, s_fix[r][1], __builtin_amdgcn_alignbyte(d2, d1, sh)), cw[1], sad);
        sad = __builtin_amdgcn_sad_u8(bipel4(B, s_fix[r][2], __builtin_amdgcn_alignbyte(d3, d2, sh)), cw[2], sad);
        sad = __builtin_amdgcn_sad_u8(bipel4(B, s_fix[r][3], __builtin_amdgcn_alignbyte(d4, d3, sh)), cw[3], sad);
      }
      const int mvx = job.mv[0] - R + ax, mvy = job.mv[1] - R + ay;
      const unsigned cost = (unsigned)(c1 + mv_cost(B.lam_f, 4 * mvx - job.pred2[0], 4 * mvy - job.pred2[1])) + sad;
      best = min(best, (cost << TIE_BITS) | (unsigned)spiral_pos(ax - R, ay - R));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, off, 64));
    if ((tid & 63) == 0) atomicMin(&s_best, best);
    __syncthreads();
    if (tid == 0) {
      const unsigned k = s_best;
      const int cost = (int)(k >> TIE_BITS);
      int dx = 0, dy = 0;
      jmhip_bipred_result o;
      if (cost < job.min_mcost) { spiral_offset((int)(k & ((1u << TIE_BITS) - 1)), &dx, &dy); o.cost = cost; }
      else o.cost = job.min_mcost;
      o.mv[0] = (int16_t)(job.mv[0] + dx); o.mv[1] = (int16_t)(job.mv[1] + dy);
      res[blockIdx.x] = o;
    }
    return;
  }

  // ---------------- SubPelBlockSearchBiPred: quarter-pel planes, origin clamp per SATD sub-block of BOTH pictures
  const uint8_t *sub1 = B.ref_sub[job.ref1], *sub2 = B.ref_sub[job.ref2];
  const size_t plane = (size_t)B.Wp * B.Hp;
  const int width_pad = B.Wp - 1 - 16, height_pad = B.Hp - 1 - 16;
  const int p4x = (ox + JMHIP_PAD) << 2, p4y = (oy + JMHIP_PAD) << 2;
  const int max_x4 = (B.W - 16 + 2 * JMHIP_PAD) << 2, max_y4 = (B.H - 16 + 2 * JMHIP_PAD) << 2;
  const int nblk = B.t8x8 ? 4 : 16, bs = B.t8x8 ? 8 : 4;
  if (tid == 0) { s_mv[0] = job.mv[0]; s_mv[1] = job.mv[1]; s_min = job.min_mcost; }
  for (int phase = 0; phase < 2; phase++) {        // start_me_refinement_hp == 0, _qp == 1 (SAD full-pel, SATD sub-pel)
    const int step = phase ? 1 : 2, first = phase ? 1 : 0, ncand = 9 - first;
    if (tid < 9) s_satd[tid] = 0;
    __syncthreads();
    const int mvx = s_mv[0], mvy = s_mv[1];
    const int m = phase ? 0 : 1;                     // me_fullsearch.c:642-661 vs :694-713
    const int umv2 = !((p4x + mvx > m) && (p4x + mvx < max_x4 - m) && (p4y + mvy > m) && (p4y + mvy < max_y4 - m));
    const int umv1 = !((p4x + job.s_mv[0] > m) && (p4x + job.s_mv[0] < max_x4 - m) && (p4y + job.s_mv[1] > m) && (p4y + job.s_mv[1] < max_y4 - m));
    for (int idx = tid; idx < nblk * ncand; idx += 256) {
      const int ci = idx / nblk, b = idx - ci * nblk, cand = first + ci;
      const int bxo = B.t8x8 ? 8 * (b & 1) : 4 * (b & 3), byo = B.t8x8 ? 8 * (b >> 1) : 4 * (b >> 2);
      const int x2 = p4x + mvx + step * c_s9x[cand] + (bxo << 2), y2 = p4y + mvy + step * c_s9y[cand] + (byo << 2);
      const int x1 = p4x + job.s_mv[0] + (bxo << 2), y1 = p4y + job.s_mv[1] + (byo << 2);
      int xp2 = x2 >> 2, yp2 = y2 >> 2, xp1 = x1 >> 2, yp1 = y1 >> 2;
      if (umv2) { xp2 = clampi(xp2, 0, width_pad); yp2 = clampi(yp2, 0, height_pad); }
      if (umv1) { xp1 = clampi(xp1, 0, width_pad); yp1 = clampi(yp1, 0, height_pad); }
      const uint8_t *r2 = sub2 + (size_t)((y2 & 3) * 4 + (x2 & 3)) * plane + (size_t)yp2 * B.Wp + xp2;
      const uint8_t *r1 = sub1 + (size_t)((y1 & 3) * 4 + (x1 & 3)) * plane + (size_t)yp1 * B.Wp + xp1;
      int v;
      if (bs == 4) {
        uint32_t rf[4], c01[4], c23[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          uint32_t a, bq, hi;
          fetch_row(r1 + (size_t)r * B.Wp, 4, &a, &hi);
          fetch_row(r2 + (size_t)r * B.Wp, 4, &bq, &hi);
          rf[r] = bipel4(B, a, bq);
          c01[r] = s_c16[byo + r][bxo >> 1]; c23[r] = s_c16[byo + r][(bxo >> 1) + 1];
        }
        v = satd4x4_packed(c01, c23, rf);
      } else {
        int m2[8][8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
          uint32_t a0, a1, b0, b1;
          fetch_row(r1 + (size_t)r * B.Wp, 8, &a0, &a1);
          fetch_row(r2 + (size_t)r * B.Wp, 8, &b0, &b1);
          const uint32_t lo = bipel4(B, a0, b0), hi = bipel4(B, a1, b1);
          const uint32_t c0 = *reinterpret_cast<const uint32_t *>(&s_cur[byo + r][bxo]), c1 = *reinterpret_cast<const uint32_t *>(&s_cur[byo + r][bxo + 4]);
          int row[8];
#pragma unroll
          for (int x = 0; x < 4; x++) { row[x] = (int)((c0 >> (8 * x)) & 255) - (int)((lo >> (8 * x)) & 255); row[4 + x] = (int)((c1 >> (8 * x)) & 255) - (int)((hi >> (8 * x)) & 255); }
          had8(row);
#pragma unroll
          for (int x = 0; x < 8; x++) m2[r][x] = row[x];
        }
        int s = 0;
#pragma unroll
        for (int x = 0; x < 8; x++) {
          int col[8];
#pragma unroll
          for (int r = 0; r < 8; r++) col[r] = m2[r][x];
          had8(col);
#pragma unroll
          for (int r = 0; r < 8; r++) s += iabs(col[r]);
        }
        v = (s + 2) >> 2;
      }
      atomicAdd(&s_satd[cand], v);
    }
    __syncthreads();
    if (tid == 0) {
      const int lam = phase ? B.lam_q : B.lam_h;
      int min_mcost = s_min, best = 0;
      for (int pos = first; pos < 9; pos++) {
        int mcost = mv_cost(lam, mvx + step * c_s9x[pos] - job.pred2[0], mvy + step * c_s9y[pos] - job.pred2[1]);
        if (mcost >= min_mcost) continue;
        mcost += s_satd[pos];
        if (mcost < min_mcost) { min_mcost = mcost; best = pos; }
      }
      s_mv[0] = mvx + step * c_s9x[best]; s_mv[1] = mvy + step * c_s9y[best]; s_min = min_mcost;
    }
    __syncthreads();
  }
  if (tid == 0) { jmhip_bipred_result o; o.mv[0] = (int16_t)s_mv[0]; o.mv[1] = (int16_t)s_mv[1]; o.cost = s_min; res[blockIdx.x] = o; }
}

}  // namespace

extern "C" int jmhip_bipred_search(jmhip_ctx *c, const jmhip_bipred_params *prm, const jmhip_bipred_job *jobs, int n, jmhip_bipred_result *results)
{
  if (!c || !prm || !jobs || !results || n <= 0) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_bipred_search: NULL/empty arguments") : JMHIP_ERR_ARG;
  if (!c->has_cur) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_bipred_search: current picture not uploaded");
  int maxR = 0;
  for (int i = 0; i < n; i++) {
    const jmhip_bipred_job &j = jobs[i];
    if (j.mb_x < 0 || j.mb_x >= c->mbw || j.mb_y < 0 || j.mb_y >= c->mbh) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_bipred_search: macroblock outside the picture");
    for (int k = 0; k < 2; k++) {
      const int s = k ? j.ref2 : j.ref1;
      if (s < 0 || s >= (int)c->refs.size() || !c->refs[s].has_pic || (j.stage == 1 && !c->refs[s].has_luma_sub))
        return jm_fail(c, JMHIP_ERR_ARG, "jmhip_bipred_search: reference slot not ready");
    }
    if (j.stage != 0 && j.stage != 1) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_bipred_search: stage");
    if (j.stage == 0 && (j.search_range < 0 || j.search_range > 44)) return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_bipred_search: search range > 44");
    if (j.stage == 0 && j.search_range > maxR) maxR = j.search_range;
    for (int k = 0; k < 2; k++) if (j.mv[k] < -8192 || j.mv[k] > 8192 || j.s_mv[k] < -8192 || j.s_mv[k] > 8192) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_bipred_search: vector out of range");
  }
  for (int k = 0; k < 3; k++) if (prm->lambda[k] < 0 || prm->lambda[k] > 4000000) return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_bipred_search: lambda factor out of the packed key range");
  if (prm->apply_weights && (prm->luma_log_weight_denom < 0 || prm->luma_log_weight_denom > 14)) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_bipred_search: weight denominator");
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  int rc = jm_ensure_ref_table(c);
  if (rc) return rc;
  void *dj = nullptr, *dr = nullptr;
  if (hipMalloc(&dj, sizeof(jmhip_bipred_job) * (size_t)n) != hipSuccess || hipMalloc(&dr, sizeof(jmhip_bipred_result) * (size_t)n) != hipSuccess) {
    (void)hipFree(dj); return jm_fail(c, JMHIP_ERR_NOMEM, "bi-pred arrays");
  }
  BiDev B{};
  B.W = c->W; B.H = c->H; B.Wp = c->Wp; B.Hp = c->Hp; B.cur = c->cur_y;
  B.lam_f = prm->lambda[0]; B.lam_h = prm->lambda[1]; B.lam_q = prm->lambda[2]; B.t8x8 = prm->transform8x8_mode ? 1 : 0;
  B.wp = prm->apply_weights ? 1 : 0; B.w1 = prm->weight1; B.w2 = prm->weight2; B.off = prm->offset_bi; B.rnd = prm->wp_luma_round; B.den = prm->luma_log_weight_denom;
  B.ref_y = reinterpret_cast<const uint8_t *const *>(c->ref_ptrs_dev);
  B.ref_sub = reinterpret_cast<const uint8_t *const *>(c->ref_ptrs_dev) + 32;
  const int UW = 2 * maxR + 1, pitch = (UW + 15 + 3 + 4) & ~3;
  const size_t lds = (size_t)pitch * (UW + 15) + 16;
  hipError_t e = hipMemcpyAsync(dj, jobs, sizeof(jmhip_bipred_job) * (size_t)n, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) { bipred_kernel<<<n, 256, lds, c->stream>>>(B, (const jmhip_bipred_job *)dj, (jmhip_bipred_result *)dr); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipMemcpyAsync(results, dr, sizeof(jmhip_bipred_result) * (size_t)n, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(dj); (void)hipFree(dr);
  if (e != hipSuccess) { c->err = std::string("jmhip_bipred_search: ") + hipGetErrorString(e); return JMHIP_ERR_DEVICE; }
  return JMHIP_OK;
}

// ------------------------------------------------------------------------------------------------ RD-off mode-decision costs

namespace {

// One wavefront per macroblock: lanes 0..15 fetch their 4x4 prediction block (quarter-pel plane of the block's vector, UMV clamp
// of the block origin) and form its residual; lanes 0..15 then produce distortion4x4, lanes 0..3 distortion8x8.
__global__ __launch_bounds__(64) void predcost_kernel(MeDev P, const jmhip_predcost_job *__restrict__ jobs, int n, int metric, int layout,
                                                     int32_t *__restrict__ out)
{
  __shared__ int s_diff[16][16];                     // residual of 4x4 block b (raster y*4+x) as 16 values, row-major inside the block
  __shared__ int s_c4[16], s_c8[4];
  const int i = blockIdx.x, tid = threadIdx.x;
  if (i >= n) return;
  const jmhip_predcost_job &job = jobs[i];
  if (tid < 16) {
    const int x4 = tid & 3, y4 = tid >> 2;
    // LumaPrediction (macroblock.c:836-945): one or two quarter-pel fetches, each with its own UMV origin clamp, then the mix
    uint32_t pv[4];
    {
      const int xq = ((job.mb_x * 16 + 4 * x4 + JMHIP_PAD) << 2) + job.mv[tid][0], yq = ((job.mb_y * 16 + 4 * y4 + JMHIP_PAD) << 2) + job.mv[tid][1];
      const int xpos = clampi(xq >> 2, 0, P.Wp - 1 - 16), ypos = clampi(yq >> 2, 0, P.Hp - 1 - 16);           // UMVLine4X, refbuf.c:37
      const uint8_t *src = P.ref_sub[job.ref[tid]] + (size_t)((yq & 3) * 4 + (xq & 3)) * P.Wp * P.Hp + (size_t)ypos * P.Wp + xpos;
      uint32_t hi;
#pragma unroll
      for (int r = 0; r < 4; r++) fetch_row(src + (size_t)r * P.Wp, 4, &pv[r], &hi);
    }
    if (job.bi[tid]) {
      const int xq = ((job.mb_x * 16 + 4 * x4 + JMHIP_PAD) << 2) + job.mv1[tid][0], yq = ((job.mb_y * 16 + 4 * y4 + JMHIP_PAD) << 2) + job.mv1[tid][1];
      const int xpos = clampi(xq >> 2, 0, P.Wp - 1 - 16), ypos = clampi(yq >> 2, 0, P.Hp - 1 - 16);
      const uint8_t *src = P.ref_sub[job.ref1[tid]] + (size_t)((yq & 3) * 4 + (xq & 3)) * P.Wp * P.Hp + (size_t)ypos * P.Wp + xpos;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        uint32_t q, hi;
        fetch_row(src + (size_t)r * P.Wp, 4, &q, &hi);
        if (!job.weighted) pv[r] = __builtin_amdgcn_lerp(pv[r], q, 0x01010101u);                              // (a + b + 1) >> 1, :917
        else {
          uint32_t v = 0;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int a = (pv[r] >> (8 * k)) & 255, b = (q >> (8 * k)) & 255;
            v |= (uint32_t)min(max(((job.w0[tid] * a + job.w1[tid] * b + 2 * job.wp_round) >> (job.wp_denom + 1)) + job.off[tid], 0), 255) << (8 * k);   // :884-890
          }
          pv[r] = v;
        }
      }
    } else if (job.weighted) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int a = (pv[r] >> (8 * k)) & 255;
          v |= (uint32_t)min(max(((job.w0[tid] * a + job.wp_round) >> job.wp_denom) + job.off[tid], 0), 255) << (8 * k);                                     // :892-899
        }
        pv[r] = v;
      }
    }
    int d[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint32_t cv = *reinterpret_cast<const uint32_t *>(P.cur + (size_t)(job.mb_y * 16 + 4 * y4 + r) * P.W + job.mb_x * 16 + 4 * x4);
#pragma unroll
      for (int x = 0; x < 4; x++) { d[r][x] = (int)((cv >> (8 * x)) & 255) - (int)((pv[r] >> (8 * x)) & 255); s_diff[tid][r * 4 + x] = d[r][x]; }
    }
    int c;
    if (metric == 2) c = satd4x4(d);
    else { c = 0; for (int r = 0; r < 4; r++) for (int x = 0; x < 4; x++) c += iabs(d[r][x]); }
    s_c4[tid] = c;
  }
  __syncthreads();
  if (tid < 4) {
    const int b8 = tid, bx4 = 2 * (b8 & 1), by4 = 2 * (b8 >> 1);
    // diff64 as JM builds it: the four 4x4 blocks (raster inside the 8x8) one after the other ...
    int seq[64];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int k = 0; k < 16; k++) seq[q * 16 + k] = s_diff[(by4 + (q >> 1)) * 4 + bx4 + (q & 1)][k];
    int m2[8][8];
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
      for (int x = 0; x < 8; x++)
        m2[r][x] = layout == JMHIP_DIFF64_RASTER ? s_diff[(by4 + (r >> 2)) * 4 + bx4 + (x >> 2)][(r & 3) * 4 + (x & 3)]   // ... or the true raster
                                                 : seq[r * 8 + x];
    int c = 0;
    if (metric == 2) {
#pragma unroll
      for (int r = 0; r < 8; r++) had8(m2[r]);
#pragma unroll
      for (int x = 0; x < 8; x++) {
        int col[8];
#pragma unroll
        for (int r = 0; r < 8; r++) col[r] = m2[r][x];
        had8(col);
#pragma unroll
        for (int r = 0; r < 8; r++) c += iabs(col[r]);
      }
      c = (c + 2) >> 2;
    } else {
      for (int r = 0; r < 8; r++) for (int x = 0; x < 8; x++) c += iabs(m2[r][x]);
    }
    s_c8[tid] = c;
  }
  __syncthreads();
  if (tid == 0) {
    int c4 = 0, c8 = 0;
    for (int b = 0; b < 16; b++) if ((job.blocks >> b) & 1) c4 += s_c4[b];
    for (int b8 = 0; b8 < 4; b8++) {
      const int o = 8 * (b8 >> 1) + 2 * (b8 & 1);      // top-left 4x4 of the 8x8
      const unsigned need = (1u << o) | (1u << (o + 1)) | (1u << (o + 4)) | (1u << (o + 5));
      if ((job.blocks & need) == need) c8 += s_c8[b8];
    }
    out[2 * i] = c4; out[2 * i + 1] = c8;
  }
}

}  // namespace

extern "C" int jmhip_pred_cost_batch(jmhip_ctx *c, const jmhip_predcost_job *jobs, int n, int metric, int layout, int32_t (*out)[2])
{
  if (!c || !jobs || !out || n <= 0) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_pred_cost_batch: NULL/empty arguments") : JMHIP_ERR_ARG;
  if (metric != 0 && metric != 2) return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_pred_cost_batch: metric must be 0 (SAD) or 2 (SATD)");
  if (layout != JMHIP_DIFF64_SEQUENTIAL && layout != JMHIP_DIFF64_RASTER) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_pred_cost_batch: layout");
  if (!c->has_cur) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_pred_cost_batch: current picture not uploaded");
  for (int i = 0; i < n; i++) {
    const jmhip_predcost_job &j = jobs[i];
    if (j.mb_x < 0 || j.mb_x >= c->mbw || j.mb_y < 0 || j.mb_y >= c->mbh) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_pred_cost_batch: macroblock outside the picture");
    if (j.weighted && (j.wp_denom < 0 || j.wp_denom > 14)) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_pred_cost_batch: weight denominator");
    for (int b = 0; b < 16; b++) {
      if (j.ref[b] < 0 || j.ref[b] >= (int)c->refs.size() || !c->refs[j.ref[b]].has_luma_sub) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_pred_cost_batch: sub-pel planes of a reference not built");
      if (j.mv[b][0] < -8192 || j.mv[b][0] > 8192 || j.mv[b][1] < -8192 || j.mv[b][1] > 8192) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_pred_cost_batch: vector out of range");
      if (j.bi[b]) {
        if (j.ref1[b] < 0 || j.ref1[b] >= (int)c->refs.size() || !c->refs[j.ref1[b]].has_luma_sub) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_pred_cost_batch: sub-pel planes of the second reference not built");
        if (j.mv1[b][0] < -8192 || j.mv1[b][0] > 8192 || j.mv1[b][1] < -8192 || j.mv1[b][1] > 8192) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_pred_cost_batch: second vector out of range");
      }
    }
  }
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  int rc = jm_ensure_ref_table(c);
  if (rc) return rc;
  void *dj = nullptr, *dout = nullptr;
  if (hipMalloc(&dj, sizeof(jmhip_predcost_job) * (size_t)n) != hipSuccess || hipMalloc(&dout, sizeof(int32_t) * 2 * (size_t)n) != hipSuccess) {
    (void)hipFree(dj); return jm_fail(c, JMHIP_ERR_NOMEM, "prediction-cost arrays");
  }
  MeDev P{};
  P.W = c->W; P.H = c->H; P.Wp = c->Wp; P.Hp = c->Hp; P.cur = c->cur_y;
  P.ref_sub = reinterpret_cast<const uint8_t *const *>(c->ref_ptrs_dev) + 32;
  hipError_t e = hipMemcpyAsync(dj, jobs, sizeof(jmhip_predcost_job) * (size_t)n, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) { predcost_kernel<<<n, 64, 0, c->stream>>>(P, (const jmhip_predcost_job *)dj, n, metric, layout, (int32_t *)dout); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipMemcpyAsync(out, dout, sizeof(int32_t) * 2 * (size_t)n, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(dj); (void)hipFree(dout);
  if (e != hipSuccess) { c->err = std::string("jmhip_pred_cost_batch: ") + hipGetErrorString(e); return JMHIP_ERR_DEVICE; }
  return JMHIP_OK;
}
