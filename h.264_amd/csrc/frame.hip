// frame.hip -- frame stage after the search: MC prediction -> residual -> transform/quant -> thresholding -> recon.
//
// Device-side restatement of the inter path of LumaResidualCoding / ChromaResidualCoding for P macroblocks without
// the 8x8 transform (the step between ME and transform):
//   SetModesAndRefframe + LumaResidualCoding8x8   lencod/src/macroblock.c:1009-1300  (4x4 loop :1039-1110,
//                                                  _LUMA_COEFF_COST_ thresholding :1236-1258)
//   LumaResidualCoding (MB-level _LUMA_MB_COEFF_COST_)               macroblock.c:1357-1449
//   LumaPrediction / OneComponentLumaPrediction                       macroblock.c:836, :807   (UMVLine4X per 4x4 block)
//   OneComponentChromaPrediction4x4_retrieve                          macroblock.c:1593-1652   (UMVLine8X_chroma per 2 samples)
//   ChromaResidualCoding: cr_cbp chained over U,V, cbp += cr_cbp<<4   macroblock.c:1944-2044
// The transform/quant itself is tq.hip's dct_4x4 / dct_chroma kernels run over device-resident job arrays.
// Which mode a macroblock takes is the host's decision (JM's mode decision is out of scope); without one the
// device picks the partitioning with the smallest summed motion cost.
#include "jmhip_internal.h"
#include <utility>
#include <vector>
#include <cstring>
#include <cstdlib>

#include "frame_common.h"

namespace {

__global__ __launch_bounds__(64) void mc_kernel(FrameDev F, const jmhip_me_mb *__restrict__ mbs, const jmhip_me_result *__restrict__ me,
                                               const jmhip_mb_mode *__restrict__ modes_in, jmhip_mb_mode *__restrict__ modes_out,
                                               jmhip_tq_job *__restrict__ jobs_y, jmhip_tq_job *__restrict__ jobs_c, int n_items)
{
  __shared__ jmhip_mb_mode s_mode;
  __shared__ short s_mv[16][2];
  __shared__ int s_ref[4];
  const int i = jm_xcd_item(n_items), tid = threadIdx.x;
  if (i < 0) return;
  const jmhip_me_mb &mb = mbs[i];
  const jmhip_me_result &r = me[i];
  const int mbx = mb.mb_x, mby = mb.mb_y;

  if (tid == 0) {
    jmhip_mb_mode m;
    if (modes_in) m = modes_in[i];
    else {
      // smallest summed motion cost; ties go to the lower mode number
      const int *c = r.cost;
      int c8 = 0;
      for (int b = 0; b < 4; b++) {
        const int s4 = c[5 + b], s5 = c[9 + 2 * b] + c[10 + 2 * b], s6 = c[17 + 2 * b] + c[18 + 2 * b];
        const int s7 = c[25 + 4 * b] + c[26 + 4 * b] + c[27 + 4 * b] + c[28 + 4 * b];
        int best = s4, bm = 4;
        if (s5 < best) { best = s5; bm = 5; }
        if (s6 < best) { best = s6; bm = 6; }
        if (s7 < best) { best = s7; bm = 7; }
        m.b8mode[b] = (int8_t)bm; c8 += best;
      }
      int best = c[0]; m.mode = 1;
      if (c[1] + c[2] < best) { best = c[1] + c[2]; m.mode = 2; }
      if (c[3] + c[4] < best) { best = c[3] + c[4]; m.mode = 3; }
      if (c8 < best) { best = c8; m.mode = 8; }
      m.pad[0] = m.pad[1] = m.pad[2] = 0;
    }
    s_mode = m;
    modes_out[i] = m;
  }
  __syncthreads();
  if (tid < 16) {
    const int p = covering_partition(s_mode, tid & 3, tid >> 2);
    s_mv[tid][0] = r.mv[p][0]; s_mv[tid][1] = r.mv[p][1];
  }
  if (tid < 4) s_ref[tid] = F.blk_ref ? F.blk_ref[(size_t)i * 4 + tid] : mb.ref;
  __syncthreads();

  jmhip_tq_job &jy = jobs_y[i];
  if (tid < 16) {                                    // luma: one 4x4 block per lane, LumaPrediction(..., 4, 4, ...)
    const int x4 = tid & 3, y4 = tid >> 2;
    // with the 8x8 transform LumaPrediction is called per 8x8 block (macroblock.c:1143): the UMV clamp then applies to the
    // 8x8 block's origin and this 4x4 block sits at its offset inside it
    const int t8 = s_mode.pad[0] ? 1 : 0, ox4 = t8 ? (x4 & 1) * 4 : 0, oy4 = t8 ? (y4 & 1) * 4 : 0;
    const int bqx = ((mbx * 16 + 4 * x4 - ox4) << 2) + 4 * JMHIP_PAD, bqy = ((mby * 16 + 4 * y4 - oy4) << 2) + 4 * JMHIP_PAD;
    const int xq = bqx + s_mv[tid][0], yq = bqy + s_mv[tid][1];                            // pic_opix_x + mv, macroblock.c:851
    const int b8 = 2 * (y4 >> 1) + (x4 >> 1), slot = s_ref[b8];
    int pdir = 0, slot1 = 0, xq1 = 0, yq1 = 0;                                               // the second list of a B macroblock
    if (F.bi) { const jmhip_mb_bipred &bm = F.bi[i]; pdir = bm.pdir[b8]; slot1 = bm.ref1[b8]; xq1 = bqx + bm.mv1[tid][0]; yq1 = bqy + bm.mv1[tid][1]; }
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
      const uint32_t v0 = pdir != 1 ? luma_row4(F, slot, xq, yq, ox4, oy4, rr) : 0u;
      const uint32_t v1 = pdir != 0 ? luma_row4(F, slot1, xq1, yq1, ox4, oy4, rr) : 0u;
      uint32_t pv = v0;
      if (F.wp_on || pdir) {
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) w |= (uint32_t)mix_pred(F, pdir, slot, slot1, 0, (int)((v0 >> (8 * k)) & 255u), (int)((v1 >> (8 * k)) & 255u)) << (8 * k);
        pv = w;
      }
      *reinterpret_cast<uint32_t *>(&jy.pred[4 * y4 + rr][4 * x4]) = pv;
      *reinterpret_cast<uint32_t *>(&jy.src[4 * y4 + rr][4 * x4]) =
          *reinterpret_cast<const uint32_t *>(F.cur_y + (size_t)(mby * 16 + 4 * y4 + rr) * F.W + mbx * 16 + 4 * x4);
    }
    if (tid == 0) { jy.quant = s_mode.pad[0] ? 3 : 0; jy.quant_dc = 0; jy.uv = 0; jy.cr_cbp_in = 0; jy.intra16_unused = s_mode.pad[0] ? 1 : 0; }
  }

  if (F.yuv != JMHIP_YUV400) {
    // chroma: every pair of samples is fetched with the motion vector of the luma 4x4 block above it (macroblock.c:1626-1650)
    const int rsx = 4 - F.shift_x, rsy = 4 - F.shift_y;
    const int npairs = F.mb_ch * (F.mb_cw / 2);
    for (int t = tid; t < 2 * npairs; t += 64) {
      const int uv = t / npairs, q = t - uv * npairs;
      const int j = q / (F.mb_cw / 2), ic = 2 * (q - j * (F.mb_cw / 2));
      const int by4 = j >> rsy, bx4 = ic >> rsx;                     // luma 4x4 block indices
      const short *mv = s_mv[by4 * 4 + bx4];
      const int bii = ((ic + mbx * F.mb_cw) << F.shift_x) + 4 * JMHIP_PAD, bjj = ((j + mby * F.mb_ch) << F.shift_y) + 4 * JMHIP_PAD;
      jmhip_tq_job &jc = jobs_c[2 * i + uv];
      const int b8 = 2 * (by4 >> 1) + (bx4 >> 1), slot = s_ref[b8];
      int pdir = 0, slot1 = 0;
      if (F.bi) { pdir = F.bi[i].pdir[b8]; slot1 = F.bi[i].ref1[b8]; }
      // a sample of plane (jj & mask_y, ii & mask_x) of getSubImagesChroma at the clamped position: from the planes, or -- when they were not
      // built -- the same value computed from the integer chroma picture (chroma_pair, frame_common.h; img_chroma.c:412-420)
      int p0 = 0, p1 = 0, q0 = 0, q1 = 0;
      if (pdir != 1) chroma_pair(F, slot, uv, bii + mv[0], bjj + mv[1], &p0, &p1);
      if (pdir != 0) { const short *m1 = F.bi[i].mv1[by4 * 4 + bx4]; chroma_pair(F, slot1, uv, bii + m1[0], bjj + m1[1], &q0, &q1); }
      if (F.wp_on || pdir) { p0 = mix_pred(F, pdir, slot, slot1, uv + 1, p0, q0); p1 = mix_pred(F, pdir, slot, slot1, uv + 1, p1, q1); }
      // ic is even: the two samples go out as one 16-bit store
      *reinterpret_cast<uint16_t *>(&jc.pred[j][ic]) = (uint16_t)(p0 | (p1 << 8));
      const uint8_t *cs = (uv ? F.cur_v : F.cur_u) + (size_t)(mby * F.mb_ch + j) * F.Wc + mbx * F.mb_cw + ic;
      *reinterpret_cast<uint16_t *>(&jc.src[j][ic]) = *reinterpret_cast<const uint16_t *>(cs);      // even column of an even-width plane: aligned
      if (q == 0) { jc.quant = 1; jc.quant_dc = 2; jc.uv = uv; jc.cr_cbp_in = 0; jc.intra16_unused = 0; }
    }
  }
}

using MbCoded = JmMbCoded;

__global__ __launch_bounds__(64) void finalize_kernel(FrameDev F, const jmhip_me_mb *__restrict__ mbs, const jmhip_tq_job *__restrict__ jobs_y,
                                                     const jmhip_tq_result *__restrict__ res_y, const jmhip_tq_job *__restrict__ jobs_c,
                                                     const jmhip_tq_result *__restrict__ res_c, MbCoded *__restrict__ coded, int n)
{
  __shared__ int s_keep[4], s_mbkeep;
  const int i = jm_xcd_item(n), tid = threadIdx.x;
  if (i < 0) return;
  const jmhip_me_mb &mb = mbs[i];
  const jmhip_tq_result &ry = res_y[i];
  if (tid == 0) {
    int cbp = 0, sum = 0;
    long long cbp_blk = 0;
    const bool t8 = jobs_y[i].intra16_unused != 0;   // luma_transform_size_8x8_flag of this macroblock (set by mc_kernel)
    for (int b8 = 0; b8 < 4; b8++) {
      int cost = 0, any = 0;
      if (t8) {                                        // macroblock.c:1181-1188
        cost = ry.coeff_cost[b8];
        if (ry.nonzero[b8]) { any = 1; cbp_blk |= 51LL << (4 * b8 - 2 * (b8 & 1)); }
      } else
      for (int b4 = 0; b4 < 4; b4++) {
        cost += ry.coeff_cost[b8 * 4 + b4];
        if (ry.nonzero[b8 * 4 + b4]) {
          any = 1;
          const int x4 = 2 * (b8 & 1) + (b4 & 1), y4 = 2 * (b8 >> 1) + (b4 >> 1);
          cbp_blk |= 1LL << (x4 + 4 * y4);           // cbp_blk_mask = (block_x>>2) + block_y, macroblock.c:1050
        }
      }
      if (any) cbp |= 1 << b8;
      int keep = 1;
      if (cost <= 4) {                                // _LUMA_COEFF_COST_, macroblock.c:1236-1245
        cost = 0; keep = 0;
        cbp &= 63 - (1 << b8);
        cbp_blk &= ~(51LL << (4 * b8 - 2 * (b8 & 1)));
      }
      s_keep[b8] = keep; sum += cost;
    }
    int mbkeep = 1;
    if (sum <= 5) { cbp &= 0xfffff0; cbp_blk &= 0xff0000; mbkeep = 0; }     // _LUMA_MB_COEFF_COST_, macroblock.c:1386-1392
    s_mbkeep = mbkeep;
    if (F.yuv != JMHIP_YUV400) {
      const jmhip_tq_result &ru = res_c[2 * i], &rv = res_c[2 * i + 1];
      cbp_blk = (cbp_blk & ~ru.cbp_clear) | ru.cbp_blk;
      cbp_blk = (cbp_blk & ~rv.cbp_clear) | rv.cbp_blk;
      cbp += max(ru.ret, rv.ret) << 4;                // macroblock.c:2028-2040
    }
    coded[i].cbp = cbp; coded[i].pad = 0; coded[i].cbp_blk = cbp_blk;
  }
  __syncthreads();
  // recon picture: 64 lanes x one dword (4 samples) per row quarter
  {
    const int row = tid >> 2, cq = tid & 3;          // 16 rows x 4 dwords
    const int b8 = 2 * (row >> 3) + (cq >> 1);
    const bool keep = s_mbkeep && s_keep[b8];
    const uint32_t v = keep ? *reinterpret_cast<const uint32_t *>(&ry.recon[row][cq * 4])
                            : *reinterpret_cast<const uint32_t *>(&jobs_y[i].pred[row][cq * 4]);
    *reinterpret_cast<uint32_t *>(F.rec_y + (size_t)(mb.mb_y * 16 + row) * F.W + mb.mb_x * 16 + cq * 4) = v;
  }
  if (F.yuv != JMHIP_YUV400) {
    const int ndw = F.mb_ch * (F.mb_cw / 4);
    for (int t = tid; t < 2 * ndw; t += 64) {
      const int uv = t / ndw, q = t - uv * ndw, row = q / (F.mb_cw / 4), cq = q - row * (F.mb_cw / 4);
      const uint32_t v = *reinterpret_cast<const uint32_t *>(&res_c[2 * i + uv].recon[row][cq * 4]);
      *reinterpret_cast<uint32_t *>((uv ? F.rec_v : F.rec_u) + (size_t)(mb.mb_y * F.mb_ch + row) * F.Wc + mb.mb_x * F.mb_cw + cq * 4) = v;
    }
  }
  (void)jobs_c;
}

}  // namespace
int jm_frame_buffers_ensure(jmhip_ctx *c, int n);
namespace {
int ensure_frame_buffers(jmhip_ctx *c, int n) { return jm_frame_buffers_ensure(c, n); }
}  // namespace

int jm_frame_buffers_ensure(jmhip_ctx *c, int n)
{
  { int rc = jm_ensure_recon(c); if (rc) return rc; }
  if (c->fr_capacity >= n) return JMHIP_OK;
  void **bufs[] = {&c->fr_jobs_y, &c->fr_jobs_c, &c->fr_res_y, &c->fr_res_c, &c->fr_modes, &c->fr_blk_ref, &c->fr_rec};
  for (auto b : bufs) { if (*b) JM_HIP_CHECK(c, hipFree(*b)); *b = nullptr; }
  c->fr_capacity = 0;
  // the chroma job tiles are only partly written per frame (mb_cr_size columns/rows): zero the rest once
  bool ok = hipMalloc(&c->fr_jobs_y, sizeof(jmhip_tq_job) * (size_t)n) == hipSuccess &&
            hipMalloc(&c->fr_jobs_c, sizeof(jmhip_tq_job) * (size_t)n * 2) == hipSuccess &&
            hipMalloc(&c->fr_res_y, sizeof(jmhip_tq_result) * (size_t)n) == hipSuccess &&
            hipMalloc(&c->fr_res_c, sizeof(jmhip_tq_result) * (size_t)n * 2) == hipSuccess &&
            hipMalloc(&c->fr_modes, (sizeof(jmhip_mb_mode) * 2 + sizeof(JmMbCoded)) * (size_t)n) == hipSuccess &&
            hipMalloc(&c->fr_blk_ref, 4 * (size_t)n) == hipSuccess &&
            hipMalloc(&c->fr_rec, sizeof(JmMbRes) * (size_t)n) == hipSuccess;
  if (!ok) {                                            // leave nothing half-allocated behind (fr_capacity stays 0)
    for (auto b : bufs) { if (*b) (void)hipFree(*b); *b = nullptr; }
    return jm_fail(c, JMHIP_ERR_NOMEM, "frame-stage arrays");
  }
  JM_HIP_CHECK(c, hipMemsetAsync(c->fr_jobs_c, 0, sizeof(jmhip_tq_job) * (size_t)n * 2, c->stream));
  JM_HIP_CHECK(c, hipMemsetAsync(c->fr_res_y, 0, sizeof(jmhip_tq_result) * (size_t)n, c->stream));
  JM_HIP_CHECK(c, hipMemsetAsync(c->fr_res_c, 0, sizeof(jmhip_tq_result) * (size_t)n * 2, c->stream));
  if (!c->fr_quant && hipMalloc(&c->fr_quant, sizeof(jmhip_quant) * 4) != hipSuccess) return jm_fail(c, JMHIP_ERR_NOMEM, "frame-stage quantisers");
  c->fr_capacity = n;
  return JMHIP_OK;
}

extern "C" int jmhip_frame_wp_set(jmhip_ctx *c, const jmhip_frame_wp *wp)
{
  if (!c) return JMHIP_ERR_ARG;
  if (!wp || !wp->enable) { memset(&c->fr_wp, 0, sizeof(c->fr_wp)); return JMHIP_OK; }
  if (wp->luma_denom < 0 || wp->luma_denom > 7 || wp->chroma_denom < 0 || wp->chroma_denom > 7) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_frame_wp_set: log weight denominators are 0..7");
  c->fr_wp = *wp;
  return JMHIP_OK;
}

extern "C" int jmhip_frame_bipred_set(jmhip_ctx *c, const jmhip_mb_bipred *bi, int n, const jmhip_frame_bw *bw)
{
  if (!c) return JMHIP_ERR_ARG;
  if (!bi) { c->fr_bi_n = 0; return JMHIP_OK; }
  if (n <= 0) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_frame_bipred_set: n");
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 4; k++) {
      if (bi[i].pdir[k] < 0 || bi[i].pdir[k] > 2) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_frame_bipred_set: prediction direction 0..2");
      if (bi[i].pdir[k] && (bi[i].ref1[k] < 0 || bi[i].ref1[k] >= 4 || bi[i].ref1[k] >= (int)c->refs.size() || !c->refs[bi[i].ref1[k]].has_luma_sub))
        return jm_fail(c, JMHIP_ERR_ARG, "jmhip_frame_bipred_set: list-1 reference slot 0..3 with quarter-pel planes (jmhip_interp_luma)");
    }
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  if (c->fr_bi_capacity < n) {
    if (c->fr_bi) JM_HIP_CHECK(c, hipFree(c->fr_bi));
    c->fr_bi = nullptr; c->fr_bi_capacity = 0;
    if (hipMalloc(&c->fr_bi, sizeof(jmhip_mb_bipred) * (size_t)n) != hipSuccess) return jm_fail(c, JMHIP_ERR_NOMEM, "second-list array of the frame stage");
    c->fr_bi_capacity = n;
  }
  JM_HIP_CHECK(c, hipMemcpyAsync(c->fr_bi, bi, sizeof(jmhip_mb_bipred) * (size_t)n, hipMemcpyHostToDevice, c->stream));
  JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));             // caller-owned array
  c->fr_bi_n = n; c->fr_bi_mask = 0;
  for (int i = 0; i < n; i++) for (int k = 0; k < 4; k++) if (bi[i].pdir[k]) c->fr_bi_mask |= 1u << bi[i].ref1[k];
  if (bw) c->fr_bw = *bw; else memset(&c->fr_bw, 0, sizeof(c->fr_bw));
  return JMHIP_OK;
}

extern "C" int jmhip_residual_frame(jmhip_ctx *c, const jmhip_mb_mode *modes, const jmhip_quant quants[3])
{
  return jmhip_residual_frame_q(c, modes, quants, 3);
}

extern "C" int jmhip_residual_frame_q(jmhip_ctx *c, const jmhip_mb_mode *modes, const jmhip_quant *quants, int nquants)
{
  if (!c || !quants || (nquants != 3 && nquants != 4)) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_residual_frame: NULL arguments / 3 or 4 quantisers") : JMHIP_ERR_ARG;
  const int n = c->me_n;
  if (n <= 0 || !c->me_res_dev) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_residual_frame: no motion search results on the device (call jmhip_me_frame first)");
  if (c->cfg.yuv_format == JMHIP_YUV444) return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_residual_frame: 4:4:4 chroma goes through the luma path in JM (not built)");
  for (int k = 0; k < nquants; k++) {
    if (quants[k].qp < 0 || quants[k].qp > 87 || quants[k].max_val != 255 || quants[k].disthres < 0 || quants[k].disthres > 1)
      return jm_fail(c, JMHIP_ERR_ARG, "jmhip_residual_frame: quantiser out of range");
    if ((quants[k].transform8x8_flag != 0) != (k == 3)) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_residual_frame: transform8x8_flag belongs to quants[3] (the 8x8 luma quantiser) only");
  }
  bool any_t8 = false;
  if (modes)
    for (int i = 0; i < n; i++) {
      const jmhip_mb_mode &m = modes[i];
      bool ok = m.mode == 1 || m.mode == 2 || m.mode == 3 || m.mode == 8;
      if (m.mode == 8) for (int b = 0; b < 4; b++) ok = ok && m.b8mode[b] >= 4 && m.b8mode[b] <= 7;
      if (m.pad[0]) {                                // luma_transform_size_8x8_flag: no partition below 8x8 (macroblock.c:1458-1480)
        any_t8 = true;
        if (m.mode == 8) for (int b = 0; b < 4; b++) ok = ok && m.b8mode[b] == 4;
        ok = ok && nquants == 4 && m.pad[0] == 1;
      }
      if (!ok) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_residual_frame: bad macroblock mode");
    }
  if (!modes && c->fr_from_slices && c->fr_slices_t8) {       // modes left on the device by a slice search with Transform8x8Mode
    any_t8 = true;
    if (nquants != 4) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_residual_frame: the searched slices use the 8x8 transform: quants[3] (the 8x8 luma quantiser) is needed");
  }
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  int rc = ensure_frame_buffers(c, n);
  if (rc) return rc;
  if ((rc = jm_ensure_ref_table(c))) return rc;
  // chroma prediction reads the eighth-pel planes when every used reference has them (JM's ChromaMCBuffer = 1 layout); otherwise the
  // same sample values are computed in mc_kernel from the integer chroma pictures (reference slots 0..7)
  bool chroma_fly = false;
  const unsigned used_refs = c->me_ref_mask | (c->fr_bi_n ? c->fr_bi_mask : 0u);      // list 0 of the search stage, list 1 of jmhip_frame_bipred_set
  for (size_t k = 0; k < c->refs.size(); k++)
    if ((used_refs >> k) & 1) {
      if (!c->refs[k].has_luma_sub) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_residual_frame: quarter-pel planes of a used reference not built (jmhip_interp_luma)");
      if (c->Wc && !c->refs[k].has_cr_sub) {
        if (k >= 8 || !c->refs[k].has_pic) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_residual_frame: chroma planes of a used reference not built (jmhip_interp_chroma)");
        chroma_fly = true;
      }
    }

  jmhip_mb_mode *modes_in_dev = nullptr, *modes_out_dev = (jmhip_mb_mode *)c->fr_modes;
  if (!modes && c->fr_from_slices) modes_in_dev = modes_out_dev + n;        // left there by jmhip_slice_to_frame
  if (modes) {
    modes_in_dev = modes_out_dev + n;
    JM_HIP_CHECK(c, hipMemcpyAsync(modes_in_dev, modes, sizeof(jmhip_mb_mode) * (size_t)n, hipMemcpyHostToDevice, c->stream));
  }
  MbCoded *coded_dev = reinterpret_cast<MbCoded *>(modes_out_dev + 2 * (size_t)n);
  // the three quantisers are copied into a context-owned host block first: the caller's array is only borrowed for the call
  memcpy(c->fr_quant_host, quants, sizeof(jmhip_quant) * nquants);
  JM_HIP_CHECK(c, hipMemcpyAsync(c->fr_quant, c->fr_quant_host, sizeof(jmhip_quant) * nquants, hipMemcpyHostToDevice, c->stream));
  if (modes) JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));      // caller-owned mode array

  FrameDev F{};
  F.W = c->W; F.H = c->H; F.Wp = c->Wp; F.Hp = c->Hp; F.Wc = c->Wc; F.Hc = c->Hc; F.Wcp = c->Wcp; F.Hcp = c->Hcp; F.mbw = c->mbw;
  F.yuv = c->cfg.yuv_format; F.shift_x = c->cg.shift_x; F.shift_y = c->cg.shift_y; F.mask_x = c->cg.mask_x; F.mask_y = c->cg.mask_y;
  F.sub_x = c->cg.sub_x; F.mb_cw = c->cg.mb_w; F.mb_ch = c->cg.mb_h;
  F.cur_y = c->cur_y; F.cur_u = c->cur_u; F.cur_v = c->cur_v;
  const uint8_t *const *tab = reinterpret_cast<const uint8_t *const *>(c->ref_ptrs_dev);
  F.ref_sub = tab + 32; F.ref_cb = tab + 64; F.ref_cr = tab + 96;
  F.rec_y = c->rec_y; F.rec_u = c->rec_u; F.rec_v = c->rec_v;
  F.pred_y = c->keep_pred ? c->pred_y : nullptr; F.pred_u = c->keep_pred ? c->pred_u : nullptr; F.pred_v = c->keep_pred ? c->pred_v : nullptr;
  F.fly = chroma_fly ? 1 : 0; F.mul_x = c->cg.mul_x; F.mul_y = c->cg.mul_y; F.pad_cx = c->cg.pad_x; F.pad_cy = c->cg.pad_y;
  for (int k = 0; k < 8; k++) { F.ref_u[k] = k < (int)c->refs.size() ? c->refs[k].u : nullptr; F.ref_v[k] = k < (int)c->refs.size() ? c->refs[k].v : nullptr; }
  F.blk_ref = c->fr_from_slices ? (const int8_t *)c->fr_blk_ref : nullptr;
  F.bi = nullptr;
  if (c->fr_bi_n) {
    if (c->fr_bi_n != n) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_residual_frame: jmhip_frame_bipred_set was given another number of macroblocks");
    F.bi = (const jmhip_mb_bipred *)c->fr_bi;
    for (int a = 0; a < 4; a++) for (int q = 0; q < 3; q++) {
      F.bu1[a][q] = c->fr_bw.weight1[a][q]; F.bo1[a][q] = c->fr_bw.offset1[a][q];
      for (int b = 0; b < 4; b++) { F.bw0[a][b][q] = c->fr_bw.w0[a][b][q]; F.bw1[a][b][q] = c->fr_bw.w1[a][b][q]; }
    }
  }
  F.wp_on = c->fr_wp.enable ? 1 : 0; F.wp_lround = c->fr_wp.luma_round; F.wp_ldenom = c->fr_wp.luma_denom; F.wp_cround = c->fr_wp.chroma_round; F.wp_cdenom = c->fr_wp.chroma_denom;
  for (int k = 0; k < 16; k++) for (int q = 0; q < 3; q++) { F.wp_w[k][q] = c->fr_wp.weight[k][q]; F.wp_o[k][q] = c->fr_wp.offset[k][q]; }

  // the common case -- 4:2:0, 4x4 transform everywhere -- is ONE kernel that keeps the tiles on the CU and leaves a dense record per macroblock
  // (tq.hip frame_fused_kernel); JMHIP_FRAME_FUSED=0 keeps the separate kernels (also taken by 4:2:2, 4:0:0 and 8x8-transform macroblocks)
  bool fused = F.yuv == JMHIP_YUV420 && !any_t8 && quants[0].adapt_rnd_weight >= 0 && quants[0].adapt_rnd_weight < 32768 &&
               quants[1].adapt_rnd_weight >= 0 && quants[1].adapt_rnd_weight < 32768;
  if (const char *e = getenv("JMHIP_FRAME_FUSED")) if (!strcmp(e, "0")) fused = false;
  if (fused) {
    jm_stage_begin(c, JMHIP_STAGE_MC);
    rc = jm_launch_frame_fused(c, &F, c->me_jobs_dev, c->me_res_dev, modes_in_dev, modes_out_dev, c->fr_quant, c->fr_rec, coded_dev, n);
    jm_stage_end(c, JMHIP_STAGE_MC);                  // (the TQ stage has no launch of its own here: its time reads 0)
    if (rc) return rc;
    c->fr_n = n; c->rec_valid = true; c->fr_fused = true; c->pred_valid = c->keep_pred;
    return JMHIP_OK;
  }
  c->fr_fused = false; c->pred_valid = false;
  jm_stage_begin(c, JMHIP_STAGE_MC);
  mc_kernel<<<jm_xcd_grid(n), 64, 0, c->stream>>>(F, (const jmhip_me_mb *)c->me_jobs_dev, (const jmhip_me_result *)c->me_res_dev, modes_in_dev, modes_out_dev,
                                                  (jmhip_tq_job *)c->fr_jobs_y, (jmhip_tq_job *)c->fr_jobs_c, n);
  jm_stage_end(c, JMHIP_STAGE_MC);
  JM_HIP_CHECK(c, hipGetLastError());
  jm_stage_begin(c, JMHIP_STAGE_TQ);
  rc = jm_launch_tq(c, JMHIP_TQ_LUMA4x4 | JMHIP_TQ_SELECT, F.yuv, c->fr_jobs_y, c->fr_quant, c->fr_res_y, n);
  if (!rc && any_t8) rc = jm_launch_tq(c, JMHIP_TQ_LUMA8x8 | JMHIP_TQ_SELECT, F.yuv, c->fr_jobs_y, c->fr_quant, c->fr_res_y, n);
  if (!rc && F.yuv != JMHIP_YUV400) rc = jm_launch_tq(c, JMHIP_TQ_CHROMA, F.yuv, c->fr_jobs_c, c->fr_quant, c->fr_res_c, 2 * n);
  if (!rc) {
    finalize_kernel<<<jm_xcd_grid(n), 64, 0, c->stream>>>(F, (const jmhip_me_mb *)c->me_jobs_dev, (const jmhip_tq_job *)c->fr_jobs_y, (const jmhip_tq_result *)c->fr_res_y,
                                                          (const jmhip_tq_job *)c->fr_jobs_c, (const jmhip_tq_result *)c->fr_res_c, coded_dev, n);
    if (hipGetLastError() != hipSuccess) rc = jm_fail(c, JMHIP_ERR_DEVICE, "finalize_kernel launch");
  }
  jm_stage_end(c, JMHIP_STAGE_TQ);
  if (rc) return rc;
  c->fr_n = n; c->rec_valid = true;
  return JMHIP_OK;
}

extern "C" int jmhip_residual_download(jmhip_ctx *c, jmhip_tq_result *luma, jmhip_tq_result *chroma, jmhip_mb_mode *modes_out,
                                       int32_t *cbp, int64_t *cbp_blk, int n)
{
  if (!c) return JMHIP_ERR_ARG;
  if (n <= 0 || n > c->fr_n) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_residual_download: more macroblocks requested than processed");
  std::vector<JmMbRes> recs;
  if (c->fr_fused && (luma || (chroma && c->Wc))) {
    recs.resize(n);
    JM_HIP_CHECK(c, hipMemcpyAsync(recs.data(), c->fr_rec, sizeof(JmMbRes) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  } else {
    if (luma) JM_HIP_CHECK(c, hipMemcpyAsync(luma, c->fr_res_y, sizeof(jmhip_tq_result) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (chroma && c->Wc) JM_HIP_CHECK(c, hipMemcpyAsync(chroma, c->fr_res_c, sizeof(jmhip_tq_result) * (size_t)n * 2, hipMemcpyDeviceToHost, c->stream));
  }
  if (modes_out) JM_HIP_CHECK(c, hipMemcpyAsync(modes_out, c->fr_modes, sizeof(jmhip_mb_mode) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  std::vector<MbCoded> coded;
  if (cbp || cbp_blk) {
    coded.resize(n);
    // the coded array sits after the 2*n modes of the last call (n == fr_n)
    const MbCoded *coded_dev = reinterpret_cast<const MbCoded *>((jmhip_mb_mode *)c->fr_modes + 2 * (size_t)c->fr_n);
    JM_HIP_CHECK(c, hipMemcpyAsync(coded.data(), coded_dev, sizeof(MbCoded) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  }
  JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < n && (cbp || cbp_blk); i++) { if (cbp) cbp[i] = coded[i].cbp; if (cbp_blk) cbp_blk[i] = coded[i].cbp_blk; }
  // fused frame stage: expand the dense records into the ABI's result structs -- the fields the separate kernels write, zero elsewhere
  for (int i = 0; i < n && !recs.empty(); i++) {
    const JmMbRes &R = recs[i];
    if (luma) {
      jmhip_tq_result &o = luma[i];
      memset(&o, 0, sizeof(o));
      for (int b = 0; b < 16; b++) {
        for (int k = 0; k < R.cnt[b]; k++) { o.levels[b][k] = R.lev[b][k]; o.runs[b][k] = R.run[b][k]; }
        o.coeff_cost[b] = R.coeff_cost[b]; o.nonzero[b] = (R.nonzero >> b) & 1;
      }
      for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) { o.recon[y][x] = R.recon_y[y][x]; if (c->fr_quant_host[0].adaptive_rounding) o.fadjust[y][x] = R.fadj_y[y][x]; }
    }
    for (int uv = 0; uv < 2 && chroma && c->Wc; uv++) {
      jmhip_tq_result &o = chroma[2 * i + uv];
      memset(&o, 0, sizeof(o));
      for (int b = 0; b < 4; b++)
        for (int k = 0; k < R.cnt[16 + 4 * uv + b]; k++) { o.levels[b][k] = R.ac_zeroed[uv] ? 0 : R.lev[16 + 4 * uv + b][k]; o.runs[b][k] = R.run[16 + 4 * uv + b][k]; }
      for (int k = 0; k < R.dc_cnt[uv]; k++) { o.dc_levels[k] = R.dc_lev[uv][k]; o.dc_runs[k] = R.dc_run[uv][k]; }
      for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) { o.recon[y][x] = R.recon_c[uv][y][x]; if (c->fr_quant_host[1].adaptive_rounding) o.fadjust[y][x] = R.fadj_c[uv][y][x]; }
      o.ret = R.ret[uv]; o.cbp_blk = R.cbp_blk[uv]; o.cbp_clear = R.cbp_clear[uv];
    }
  }
  return JMHIP_OK;
}

extern "C" int jmhip_residual_records_download(jmhip_ctx *c, jmhip_mb_residual *records, int n)
{
  if (!c || !records) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_residual_records_download: NULL") : JMHIP_ERR_ARG;
  if (n <= 0 || n > c->fr_n) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_residual_records_download: more macroblocks requested than processed");
  if (!c->fr_fused) return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_residual_records_download: the last frame stage did not take the fused 4:2:0 kernel (use jmhip_residual_download)");
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  JM_HIP_CHECK(c, hipMemcpyAsync(records, c->fr_rec, sizeof(jmhip_mb_residual) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  return JMHIP_OK;
}

extern "C" int jmhip_frame_keep_prediction(jmhip_ctx *c, int on)
{
  if (!c) return JMHIP_ERR_ARG;
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  if (on && !c->pred_y) {
    bool ok = hipMalloc((void **)&c->pred_y, (size_t)c->W * c->H) == hipSuccess;
    if (ok && c->Wc) ok = hipMalloc((void **)&c->pred_u, (size_t)c->Wc * c->Hc) == hipSuccess && hipMalloc((void **)&c->pred_v, (size_t)c->Wc * c->Hc) == hipSuccess;
    if (!ok) {
      if (c->pred_y) (void)hipFree(c->pred_y);
      if (c->pred_u) (void)hipFree(c->pred_u);
      if (c->pred_v) (void)hipFree(c->pred_v);
      c->pred_y = c->pred_u = c->pred_v = nullptr;
      return jm_fail(c, JMHIP_ERR_NOMEM, "prediction picture");
    }
  }
  c->keep_pred = on != 0;
  if (!on) c->pred_valid = false;
  return JMHIP_OK;
}

extern "C" int jmhip_pred_download(jmhip_ctx *c, void *Y, void *U, void *V, int pel_bytes)
{
  if (!c) return JMHIP_ERR_ARG;
  if (!c->pred_y || !c->pred_valid) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_pred_download: no prediction picture (jmhip_frame_keep_prediction before a fused 4:2:0 jmhip_residual_frame)");
  int rc = Y ? jm_download_planes(c, c->pred_y, (size_t)c->W * c->H, Y, pel_bytes) : JMHIP_OK;
  if (rc) return rc;
  if (c->Wc && U && V) {
    if ((rc = jm_download_planes(c, c->pred_u, (size_t)c->Wc * c->Hc, U, pel_bytes))) return rc;
    if ((rc = jm_download_planes(c, c->pred_v, (size_t)c->Wc * c->Hc, V, pel_bytes))) return rc;
  }
  return JMHIP_OK;
}

namespace {
__global__ void set_plane_pointer(const uint8_t **table, int slot, const uint8_t *p) { table[slot] = p; }
}

// The reconstruction BECOMES reference slot `ref`: the slot's picture planes and the recon planes trade places (both are the
// context's, same geometry), so nothing is copied; the one device-side pointer the search kernels read is patched in stream order.
extern "C" int jmhip_recon_to_ref(jmhip_ctx *c, int ref)
{
  if (!c) return JMHIP_ERR_ARG;
  if (ref < 0 || ref >= (int)c->refs.size()) return jm_fail(c, JMHIP_ERR_ARG, "ref slot out of range");
  if (!c->rec_y || !c->rec_valid) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_recon_to_ref: no recon picture yet");
  RefSlot &r = c->refs[ref];
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  std::swap(r.y, c->rec_y);
  if (c->Wc) { std::swap(r.u, c->rec_u); std::swap(r.v, c->rec_v); }
  if (c->ref_ptrs_dev) {
    if (c->table_fix.idx >= 0 && c->table_fix.idx != ref) { int rc = jm_flush_table_fix(c); if (rc) return rc; }
    c->table_fix.idx = ref; c->table_fix.ptr = r.y;
  }
  r.has_pic = true; r.has_luma_sub = false; r.has_cr_sub = false;
  // the recon planes now hold the slot's OLD picture: nothing may read them as "the reconstruction" until the next
  // jmhip_residual_frame / jmhip_recon_upload has written a new one
  c->rec_valid = false;
  return JMHIP_OK;
}

int jm_flush_table_fix(jmhip_ctx *c)
{
  if (c->table_fix.idx < 0 || !c->ref_ptrs_dev) { c->table_fix.idx = -1; return JMHIP_OK; }
  set_plane_pointer<<<1, 1, 0, c->stream>>>(reinterpret_cast<const uint8_t **>(c->ref_ptrs_dev), c->table_fix.idx, c->table_fix.ptr);
  c->table_fix.idx = -1;
  JM_HIP_CHECK(c, hipGetLastError());
  return JMHIP_OK;
}

extern "C" int jmhip_recon_copy_band(jmhip_ctx *c, void *Y, void *U, void *V, int mb_row0, int mb_rows)
{
  if (!c || !Y) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_recon_copy_band: NULL destination") : JMHIP_ERR_ARG;
  if (!c->rec_y || !c->rec_valid) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_recon_copy_band: no recon picture yet");
  if (mb_row0 < 0 || mb_rows <= 0 || mb_row0 + mb_rows > c->mbh) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_recon_copy_band: band outside the picture");
  const size_t y0 = (size_t)mb_row0 * 16 * c->W, yn = (size_t)mb_rows * 16 * c->W;
  JM_HIP_CHECK(c, hipMemcpyAsync(Y, c->rec_y + y0, yn, hipMemcpyDeviceToDevice, c->stream));
  if (c->Wc && U && V) {
    const size_t c0 = (size_t)mb_row0 * c->cg.mb_h * c->Wc, cn = (size_t)mb_rows * c->cg.mb_h * c->Wc;
    JM_HIP_CHECK(c, hipMemcpyAsync(U, c->rec_u + c0, cn, hipMemcpyDeviceToDevice, c->stream));
    JM_HIP_CHECK(c, hipMemcpyAsync(V, c->rec_v + c0, cn, hipMemcpyDeviceToDevice, c->stream));
  }
  return JMHIP_OK;
}

// ---- one-buffer band exchange for slice-parallel ranks: a rank's reconstructed band travels as ONE chunk [Y rows | U rows | V rows]
//      (band_rows macroblock rows, the same for every rank; the last rank's missing rows are padding), so the per-frame exchange
//      is a single all-gather and a single scatter kernel instead of three of each.
namespace {

// chunk geometry in bytes for `band` macroblock rows
struct BandGeom { int W, Wc, H, Hc, mb_h, band; size_t ybytes, cbytes, chunk; };

__host__ __device__ inline BandGeom band_geom(int W, int Wc, int H, int Hc, int mb_h, int band)
{
  BandGeom g;
  g.W = W; g.Wc = Wc; g.H = H; g.Hc = Hc; g.mb_h = mb_h; g.band = band;
  g.ybytes = (size_t)band * 16 * W; g.cbytes = (size_t)band * mb_h * Wc; g.chunk = g.ybytes + 2 * g.cbytes;
  return g;
}

// dir 0: picture planes (rank's rows) -> chunk; dir 1: `world` chunks -> picture planes. One dword per thread iteration.
__global__ __launch_bounds__(256) void band_copy_kernel(BandGeom g, uint8_t *py, uint8_t *pu, uint8_t *pv, uint8_t *chunks, int first_rank, int nranks, int dir)
{
  const size_t chunk_dw = g.chunk / 4, total = chunk_dw * (size_t)nranks;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / chunk_dw) + first_rank;
    size_t off = (i - (size_t)(r - first_rank) * chunk_dw) * 4;           // byte offset inside the chunk
    uint8_t *plane; int pw, ph, rows_per_mb;
    if (off < g.ybytes) { plane = py; pw = g.W; ph = g.H; rows_per_mb = 16; }
    else if (off < g.ybytes + g.cbytes) { off -= g.ybytes; plane = pu; pw = g.Wc; ph = g.Hc; rows_per_mb = g.mb_h; }
    else { off -= g.ybytes + g.cbytes; plane = pv; pw = g.Wc; ph = g.Hc; rows_per_mb = g.mb_h; }
    const int row = (int)(off / pw), col = (int)(off - (size_t)row * pw);
    const int prow = r * g.band * rows_per_mb + row;                       // picture row of this chunk row
    if (prow >= ph) continue;                                              // padding rows of the last band
    uint32_t *c = reinterpret_cast<uint32_t *>(chunks + (size_t)(r - first_rank) * g.chunk) + (i - (size_t)(r - first_rank) * chunk_dw);
    uint32_t *p = reinterpret_cast<uint32_t *>(plane + (size_t)prow * pw + col);
    if (dir == 0) *c = *p; else *p = *c;
  }
}

}  // namespace

extern "C" size_t jmhip_band_chunk_bytes(jmhip_ctx *c, int band_rows)
{
  if (!c || band_rows <= 0) return 0;
  return band_geom(c->W, c->Wc, c->H, c->Hc, c->Wc ? c->cg.mb_h : 0, band_rows).chunk;
}

extern "C" int jmhip_recon_pack_band(jmhip_ctx *c, void *chunk, int rank, int band_rows)
{
  if (!c || !chunk || rank < 0 || band_rows <= 0) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_recon_pack_band: arguments") : JMHIP_ERR_ARG;
  if (!c->rec_y || !c->rec_valid) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_recon_pack_band: no recon picture yet");
  if ((c->W & 3) || (c->Wc & 3)) return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_recon_pack_band: plane widths must be multiples of 4");
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  const BandGeom g = band_geom(c->W, c->Wc, c->H, c->Hc, c->Wc ? c->cg.mb_h : 0, band_rows);
  band_copy_kernel<<<256, 256, 0, c->stream>>>(g, c->rec_y, c->rec_u, c->rec_v, (uint8_t *)chunk, rank, 1, 0);
  JM_HIP_CHECK(c, hipGetLastError());
  return JMHIP_OK;
}

extern "C" int jmhip_ref_unpack_bands(jmhip_ctx *c, int ref, const void *chunks, int world, int band_rows)
{
  if (!c || !chunks || world <= 0 || band_rows <= 0) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_ref_unpack_bands: arguments") : JMHIP_ERR_ARG;
  if (ref < 0 || ref >= (int)c->refs.size()) return jm_fail(c, JMHIP_ERR_ARG, "ref slot out of range");
  if (world * band_rows < c->mbh) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_ref_unpack_bands: the bands do not cover the picture");
  if ((c->W & 3) || (c->Wc & 3)) return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_ref_unpack_bands: plane widths must be multiples of 4");
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  RefSlot &r = c->refs[ref];
  const BandGeom g = band_geom(c->W, c->Wc, c->H, c->Hc, c->Wc ? c->cg.mb_h : 0, band_rows);
  band_copy_kernel<<<512, 256, 0, c->stream>>>(g, r.y, r.u, r.v, (uint8_t *)chunks, 0, world, 1);
  JM_HIP_CHECK(c, hipGetLastError());
  r.has_pic = true; r.has_luma_sub = false; r.has_cr_sub = false;
  return JMHIP_OK;
}

extern "C" int jmhip_recon_download(jmhip_ctx *c, void *Y, void *U, void *V, int pel_bytes)
{
  if (!c || !Y) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_recon_download: NULL output") : JMHIP_ERR_ARG;
  if (!c->rec_y || !c->rec_valid) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_recon_download: no recon picture yet");
  int rc = jm_download_planes(c, c->rec_y, (size_t)c->W * c->H, Y, pel_bytes);
  if (rc) return rc;
  if (c->Wc && U && V) {
    if ((rc = jm_download_planes(c, c->rec_u, (size_t)c->Wc * c->Hc, U, pel_bytes))) return rc;
    if ((rc = jm_download_planes(c, c->rec_v, (size_t)c->Wc * c->Hc, V, pel_bytes))) return rc;
  }
  return JMHIP_OK;
}
