// me_metric.hip -- the BlockMotionSearch chain of jmhip_me_frame / jmhip_me_subpel for every error metric and the chroma term.
//
// The fast kernels (me_int.hip, me_sub.hip) cover JM's default metrics: SAD at integer positions, Hadamard SAD at sub-pel positions,
// luma only. This file is the general form behind the same entry points (jmhip_me_params.metric_set = 1):
//   computeUniPred[level + 3 * apply_weights]   lencod/src/mv-search.c:400-424  (level = F_PEL / H_PEL / Q_PEL, input->MEErrorMetric[])
//   computeSAD :351 / computeSADWP :413 / computeSSE :1042 / computeSSEWP :1107 / computeSATD :657 / computeSATDWP :734   lencod/src/me_distortion.c
//   the chroma term of SAD / SADWP / SSE (input->ChromaMEEnable, :376-402, :443-470, :1072-1098): the Cb and Cr blocks that belong to
//     the luma block, read from the eighth-pel chroma planes (UMVLine8X_chroma, refbuf.c:69), times input->ChromaMEWeight
//   start_me_refinement_hp / _qp   mv-search.c:396-397: when two levels share a metric (and ChromaMEEnable != 1) the refinement skips
//     its centre position and competes with the carried minimum instead
//   FullPelBlockMotionSearch me_fullsearch.c:47, SubPelBlockMotionSearch :341, SetupFastFullPelSearch me_fullfast.c:491 (whose block
//     distortions are SAD when MEErrorMetric[F_PEL] is SAD and SQUARED error otherwise -- also for Hadamard -- and whose chroma term
//     is added unweighted, :776-814), FastFullPelBlockMotionSearch :833
// One workgroup per macroblock. Integer stage: lane <-> candidate of the union window (luma staged in LDS with per-sample clamping,
// chroma read from the planes), the sixteen 4x4 leaf distortions (and the four 8x8 Hadamard values when the 8x8 transform is tested)
// are shared by all 41 partitions through the partition tree; argmin over (cost, spiral index) as 64-bit keys. Sub-pel stage: one
// lane per (partition, position) evaluates its whole block the way JM's function does (block or sub-block origin clamp under UMV
// access). Not tuned: JM's defaults never come here; what matters is that every metric combination gives JM's vectors and costs.
#include "me_common.h"

namespace {

struct MetricDev {
  int metric[3];                       // F_PEL, H_PEL, Q_PEL: 0 SAD, 1 SSE, 2 Hadamard SAD
  int chroma_int, chroma_sub;          // chroma term at integer / at sub-pel positions (ChromaMEEnable >= 1 / == 2)
  int chroma_w;                        // input->ChromaMEWeight
  int start_hp, start_qp, skip_int;
  int Wc, Hc, Wcp, Hcp, shift_x, shift_y, mask_x, mask_y, sub_x, wpad_c, hpad_c, csx, csy, mbc_w, mbc_h;
  const uint8_t *cur_c[2];
  const uint8_t *const *ref_c[2];      // [slot] eighth-pel plane stacks of Cb / Cr
  int wpc_round, wpc_denom;
  short wpc_w[16][2], wpc_o[16][2];
  int c_off, c_cw, c_ch, nfx, nfy;     // chroma windows in LDS behind the luma window: [uv][fy][fx][c_ch][c_cw]
};

// partition table (JM's PartitionMotionSearch order, include/jmhip.h): blocktype, x4, y4, w4, h4
struct PI { int8_t bt, x4, y4, w4, h4; };
__device__ PI part_info(int p)
{
  if (p == 0) return {1, 0, 0, 4, 4};
  if (p < 3) return {2, 0, (int8_t)(2 * (p - 1)), 4, 2};
  if (p < 5) return {3, (int8_t)(2 * (p - 3)), 0, 2, 4};
  if (p < 9) { const int b = p - 5; return {4, (int8_t)(2 * (b & 1)), (int8_t)(2 * (b >> 1)), 2, 2}; }
  if (p < 17) { const int b = (p - 9) >> 1, k = (p - 9) & 1; return {5, (int8_t)(2 * (b & 1)), (int8_t)(2 * (b >> 1) + k), 2, 1}; }
  if (p < 25) { const int b = (p - 17) >> 1, k = (p - 17) & 1; return {6, (int8_t)(2 * (b & 1) + k), (int8_t)(2 * (b >> 1)), 1, 2}; }
  const int b = (p - 25) >> 2, k = (p - 25) & 3;
  return {7, (int8_t)(2 * (b & 1) + (k & 1)), (int8_t)(2 * (b >> 1) + (k >> 1)), 1, 1};
}

__device__ __forceinline__ int wp1(int v, int w, int o, int rnd, int den) { return min(max((((w * v) + rnd) >> den) + o, 0), 255); }
__device__ __forceinline__ int pel_err(int sse, int d) { return sse ? d * d : iabs(d); }

__device__ int satd8x8_rows(int m2[8][8])      // rows already transformed; columns + sum here. HadamardSAD8x8, me_distortion.c:272
{
  int s = 0;
  for (int x = 0; x < 8; x++) {
    int col[8];
#pragma unroll
    for (int r = 0; r < 8; r++) col[r] = m2[r][x];
    had8(col);
#pragma unroll
    for (int r = 0; r < 8; r++) s += iabs(col[r]);
  }
  return (s + 2) >> 2;
}

// One block at one quarter-pel candidate, the way JM's compute* functions read it (cand incl. the +80 pad offset)
__device__ int block_dist(const MeDev &P, const MetricDev &M, int metric, int chroma, int slot, int pic_x, int pic_y, int bsx, int bsy,
                          int cand_x, int cand_y, int umv, int t8)
{
  const uint8_t *sub = P.ref_sub[slot];
  const size_t plane = (size_t)P.Wp * P.Hp;
  const int width_pad = P.Wp - 1 - 16, height_pad = P.Hp - 1 - 16;
  const int w = P.wp_w[slot], o = P.wp_o[slot];
  int total = 0;
  if (metric != 2) {
    const int sse = metric == 1;
    int xpos = cand_x >> 2, ypos = cand_y >> 2;
    if (umv) { xpos = clampi(xpos, 0, width_pad); ypos = clampi(ypos, 0, height_pad); }
    const uint8_t *rp = sub + (size_t)((cand_y & 3) * 4 + (cand_x & 3)) * plane + (size_t)ypos * P.Wp + xpos;
    for (int y = 0; y < bsy; y++) {
      const uint8_t *cp = P.cur + (size_t)(pic_y + y) * P.W + pic_x;
      for (int x = 0; x < bsx; x++) {
        int rv = rp[(size_t)y * P.Wp + x];
        if (P.wp_on) rv = wp1(rv, w, o, P.wp_round, P.wp_denom);
        total += pel_err(sse, (int)cp[x] - rv);
      }
    }
    if (chroma) {
      const int bcx = bsx >> M.csx, bcy = bsy >> M.csy, pcx = pic_x >> M.csx, pcy = pic_y >> M.csy;
      for (int k = 0; k < 2; k++) {
        int cxp = cand_x >> M.shift_x, cyp = cand_y >> M.shift_y;
        if (umv) { cxp = clampi(cxp, 0, M.wpad_c); cyp = clampi(cyp, 0, M.hpad_c); }
        const uint8_t *cr = M.ref_c[k][slot] + (size_t)((cand_y & M.mask_y) * M.sub_x + (cand_x & M.mask_x)) * ((size_t)M.Wcp * M.Hcp) + (size_t)cyp * M.Wcp + cxp;
        int mcr = 0;
        for (int y = 0; y < bcy; y++)
          for (int x = 0; x < bcx; x++) {
            int rv = cr[(size_t)y * M.Wcp + x];
            if (P.wp_on) rv = wp1(rv, M.wpc_w[slot][k], M.wpc_o[slot][k], M.wpc_round, M.wpc_denom);
            mcr += pel_err(sse, (int)M.cur_c[k][(size_t)(pcy + y) * M.Wc + pcx + x] - rv);
          }
        total += M.chroma_w * mcr;
      }
    }
    return total;
  }
  const int bs = t8 ? 8 : 4;
  for (int by = 0; by < bsy; by += bs)
    for (int bx = 0; bx < bsx; bx += bs) {
      const int xq = cand_x + (bx << 2), yq = cand_y + (by << 2);
      int xpos = xq >> 2, ypos = yq >> 2;
      if (umv) { xpos = clampi(xpos, 0, width_pad); ypos = clampi(ypos, 0, height_pad); }      // per sub-block, me_distortion.c:678
      const uint8_t *rp = sub + (size_t)((yq & 3) * 4 + (xq & 3)) * plane + (size_t)ypos * P.Wp + xpos;
      if (bs == 4) {
        int d[4][4];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int x = 0; x < 4; x++) {
            int rv = rp[(size_t)r * P.Wp + x];
            if (P.wp_on) rv = wp1(rv, w, o, P.wp_round, P.wp_denom);
            d[r][x] = (int)P.cur[(size_t)(pic_y + by + r) * P.W + pic_x + bx + x] - rv;
          }
        total += satd4x4(d);
      } else {
        int m2[8][8];
        for (int r = 0; r < 8; r++) {
          int row[8];
#pragma unroll
          for (int x = 0; x < 8; x++) {
            int rv = rp[(size_t)r * P.Wp + x];
            if (P.wp_on) rv = wp1(rv, w, o, P.wp_round, P.wp_denom);
            row[x] = (int)P.cur[(size_t)(pic_y + by + r) * P.W + pic_x + bx + x] - rv;
          }
          had8(row);
#pragma unroll
          for (int x = 0; x < 8; x++) m2[r][x] = row[x];
        }
        total += satd8x8_rows(m2);
      }
    }
  return total;
}

constexpr unsigned long long KEY64_INVALID = ~0ull;
constexpr long long COST_BIAS = 1ll << 31;

__global__ __launch_bounds__(256) void me_metric_kernel(MeDev P, MetricDev M, const jmhip_me_mb *__restrict__ jobs, const int *__restrict__ job_index,
                                                        jmhip_me_result *__restrict__ res_all, int n_items)
{
  const int item = jm_xcd_item(n_items);
  if (item < 0) return;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ int s_cx[JMHIP_NPART], s_cy[JMHIP_NPART], s_px[JMHIP_NPART], s_py[JMHIP_NPART];
  __shared__ int s_u[8];
  __shared__ __attribute__((aligned(16))) uint8_t s_cur[256];
  __shared__ uint8_t s_cuc[2][256];
  __shared__ unsigned long long s_key[JMHIP_NPART];
  __shared__ int s_mv[JMHIP_NPART][2], s_cost[JMHIP_NPART];

  const int tid = threadIdx.x;
  const int mbi = job_index ? job_index[item] : item;
  const jmhip_me_mb &job = jobs[mbi];
  jmhip_me_result &res = res_all[mbi];
  const int mbx = job.mb_x, mby = job.mb_y, slot = job.ref;
  const unsigned long long mask = P.mask;
  const int t8 = P.t8x8;
  const int wl = P.wp_w[slot], ol = P.wp_o[slot];

  if (tid < JMHIP_NPART) {
    const int src = (P.mode == JMHIP_SEARCH_FASTFULL) ? 0 : tid;
    int cx, cy;
    search_center(P, job.pred_mv[src][0], job.pred_mv[src][1], &cx, &cy);
    s_cx[tid] = cx; s_cy[tid] = cy;
    s_px[tid] = job.pred_mv[tid][0]; s_py[tid] = job.pred_mv[tid][1];
    s_key[tid] = KEY64_INVALID;
  }
  s_cur[tid] = P.cur[(size_t)(mby * 16 + (tid >> 4)) * P.W + mbx * 16 + (tid & 15)];
  __syncthreads();

  if (!M.skip_int) {
    if (tid == 0) {
      int x0 = 1 << 30, x1 = -(1 << 30), y0 = 1 << 30, y1 = -(1 << 30);
      for (int p = 0; p < JMHIP_NPART; p++) if ((mask >> p) & 1) {
        x0 = min(x0, s_cx[p]); x1 = max(x1, s_cx[p]); y0 = min(y0, s_cy[p]); y1 = max(y1, s_cy[p]);
      }
      s_u[0] = x0 - P.R; s_u[1] = y0 - P.R; s_u[2] = x1 - x0 + 2 * P.R + 1; s_u[3] = y1 - y0 + 2 * P.R + 1;
    }
    __syncthreads();
    const int umin_x = s_u[0], umin_y = s_u[1], UW = s_u[2], UH = s_u[3];
    const int pitch = P.win_pitch;
    if (UW + 15 > pitch || UH + 15 > P.win_rows) { if (tid < JMHIP_NPART) res.cost_int[tid] = -2; return; }
    {                                                   // luma window, per-sample clamp (== the padded plane under UMV access), weighted
      const uint8_t *ref = P.ref_y[slot];
      const int bx = mbx * 16 + umin_x, by = mby * 16 + umin_y;
      for (int d = tid; d < (UW + 15) * (UH + 15); d += 256) {
        const int y = d / (UW + 15), x = d - y * (UW + 15);
        int v = ref[(size_t)clampi(by + y, 0, P.H - 1) * P.W + clampi(bx + x, 0, P.W - 1)];
        if (P.wp_on) v = wp1(v, wl, ol, P.wp_round, P.wp_denom);
        smem[(size_t)y * pitch + x] = (uint8_t)v;
      }
    }
    if (M.chroma_int) {
      // chroma windows: the planes an integer luma displacement can address (fraction 0 or 4 eighths per axis in 4:2:0), per-sample
      // clamp one short of the planes' last row / column -- JM never writes those (img_chroma.c:63, :129) and its origin clamp never reads them
      const int cxo = ((mbx * 16 + umin_x + 20) << 2) >> M.shift_x, cyo = ((mby * 16 + umin_y + 20) << 2) >> M.shift_y;
      const int per = M.c_cw * M.c_ch, total = 2 * M.nfy * M.nfx * per;
      for (int d = tid; d < total; d += 256) {
        const int pl = d / per, e = d - pl * per, y = e / M.c_cw, x = e - y * M.c_cw;
        const int k = pl / (M.nfy * M.nfx), f = pl - k * (M.nfy * M.nfx), fy = (f / M.nfx) * 4, fx = (f % M.nfx) * 4;
        int v = M.ref_c[k][slot][(size_t)(fy * M.sub_x + fx) * ((size_t)M.Wcp * M.Hcp) + (size_t)clampi(cyo + y, 0, M.Hcp - 2) * M.Wcp + clampi(cxo + x, 0, M.Wcp - 2)];
        if (P.wp_on) v = wp1(v, M.wpc_w[slot][k], M.wpc_o[slot][k], M.wpc_round, M.wpc_denom);
        smem[M.c_off + d] = (uint8_t)v;
      }
      for (int d = tid; d < 2 * M.mbc_w * M.mbc_h; d += 256) {
        const int k = d / (M.mbc_w * M.mbc_h), e = d - k * (M.mbc_w * M.mbc_h), y = e / M.mbc_w, x = e - y * M.mbc_w;
        s_cuc[k][y * 16 + x] = M.cur_c[k][(size_t)(mby * M.mbc_h + y) * M.Wc + mbx * M.mbc_w + x];
      }
    }
    __syncthreads();

    // FastFullSearch builds its distortions as SAD or squared error (dist_method, me_fullfast.c:512), with the chroma term unweighted
    const int ff = P.mode == JMHIP_SEARCH_FASTFULL;
    const int mf = ff ? (M.metric[0] == 0 ? 0 : 1) : M.metric[0];
    const int cw = ff ? 1 : M.chroma_w;
    const int w16 = (P.lam_f * 16) >> 16;
    const int quirk00 = (P.mode == JMHIP_SEARCH_FULL) && !P.rdopt && !P.is_b && job.ref_is_0;
    const int ff00 = ff && !P.rdopt;
    unsigned long long best[JMHIP_NPART];
#pragma unroll
    for (int p = 0; p < JMHIP_NPART; p++) best[p] = KEY64_INVALID;

    for (int c = tid; c < UW * UH; c += 256) {
      const int ay = c / UW, ax = c - ay * UW;
      const int mvx = umin_x + ax, mvy = umin_y + ay;
      int leaf[16], leaf8[4] = {0, 0, 0, 0};
      if (mf != 2) {
        // rows as dwords: five aligned window dwords per row, shifted into place (the row pitch leaves room for the fifth)
#pragma unroll
        for (int b = 0; b < 16; b++) leaf[b] = 0;
        const uint32_t *cur32 = reinterpret_cast<const uint32_t *>(s_cur);
        const uint8_t *wrow = smem + (size_t)ay * pitch + (ax & ~3);
        const unsigned sh = ax & 3;
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const uint32_t *wp = reinterpret_cast<const uint32_t *>(wrow + (size_t)r * pitch);
          const uint32_t d0 = wp[0], d1 = wp[1], d2 = wp[2], d3 = wp[3], d4 = wp[4];
          const uint32_t rr[4] = {__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh),
                                  __builtin_amdgcn_alignbyte(d3, d2, sh), __builtin_amdgcn_alignbyte(d4, d3, sh)};
#pragma unroll
          for (int g = 0; g < 4; g++) {
            const uint32_t cv = cur32[r * 4 + g];
            if (mf == 0) leaf[(r >> 2) * 4 + g] = (int)__builtin_amdgcn_sad_u8(rr[g], cv, (unsigned)leaf[(r >> 2) * 4 + g]);
            else {
#pragma unroll
              for (int k = 0; k < 4; k++) { const int d = (int)((rr[g] >> (8 * k)) & 255u) - (int)((cv >> (8 * k)) & 255u); leaf[(r >> 2) * 4 + g] += d * d; }
            }
          }
        }
        if (M.chroma_int) {
          // chroma block of the leaf: UMVLine8X_chroma at the candidate of the leaf's own origin (the partition's chroma block is the
          // union of its leaves' blocks; the origin clamp equals a per-sample clamp on the padded plane, whose ring is flat)
          const int X4 = (mbx * 16 + mvx + 20) << 2, Y4 = (mby * 16 + mvy + 20) << 2;
          const int fxi = (X4 & M.mask_x) >> 2, fyi = (Y4 & M.mask_y) >> 2;
          const int cxo = ((mbx * 16 + umin_x + 20) << 2) >> M.shift_x, cyo = ((mby * 16 + umin_y + 20) << 2) >> M.shift_y;
          const int lw = 4 >> M.csx, lh = 4 >> M.csy;
          for (int k = 0; k < 2; k++) {
            const uint8_t *pl = smem + M.c_off + ((k * M.nfy + fyi) * M.nfx + fxi) * (M.c_cw * M.c_ch);
#pragma unroll
            for (int b = 0; b < 16; b++) {                  // static leaf index: the array stays in registers
              const int ox = (b & 3) * 4, oy = (b >> 2) * 4;
              const int cx0 = ((X4 + (ox << 2)) >> M.shift_x) - cxo, cy0 = ((Y4 + (oy << 2)) >> M.shift_y) - cyo;
              int s = 0;
              for (int r = 0; r < lh; r++)
                for (int x = 0; x < lw; x++)
                  s += pel_err(mf == 1, (int)s_cuc[k][((oy >> M.csy) + r) * 16 + (ox >> M.csx) + x] - (int)pl[(cy0 + r) * M.c_cw + cx0 + x]);
              leaf[b] += cw * s;
            }
          }
        }
      } else {
#pragma unroll
        for (int b = 0; b < 16; b++) {
          const int ox = (b & 3) * 4, oy = (b >> 2) * 4;
          int d[4][4];
#pragma unroll
          for (int r = 0; r < 4; r++)
#pragma unroll
            for (int x = 0; x < 4; x++) d[r][x] = (int)s_cur[(oy + r) * 16 + ox + x] - (int)smem[(size_t)(ay + oy + r) * pitch + ax + ox + x];
          leaf[b] = satd4x4(d);
        }
        if (t8)
          for (int b = 0; b < 4; b++) {
            const int ox = (b & 1) * 8, oy = (b >> 1) * 8;
            int m2[8][8];
            for (int r = 0; r < 8; r++) {
              int row[8];
#pragma unroll
              for (int x = 0; x < 8; x++) row[x] = (int)s_cur[(oy + r) * 16 + ox + x] - (int)smem[(size_t)(ay + oy + r) * pitch + ax + ox + x];
              had8(row);
#pragma unroll
              for (int x = 0; x < 8; x++) m2[r][x] = row[x];
            }
            leaf8[b] = satd8x8_rows(m2);
          }
      }
      const int use8 = (mf == 2) && t8;               // test8x8transform = Transform8x8Mode && blocktype <= 4, mv-search.c:640
#pragma unroll
      for (int p = 0; p < JMHIP_NPART; p++) {
        if (!((mask >> p) & 1)) continue;
        const PI q = part_info(p);
        int dist = 0;
        if (use8 && q.bt <= 4) { for (int y = q.y4 >> 1; y < (q.y4 + q.h4) >> 1; y++) for (int x = q.x4 >> 1; x < (q.x4 + q.w4) >> 1; x++) dist += leaf8[y * 2 + x]; }
        else { for (int y = q.y4; y < q.y4 + q.h4; y++) for (int x = q.x4; x < q.x4 + q.w4; x++) dist += leaf[y * 4 + x]; }
        const int dx = mvx - s_cx[p], dy = mvy - s_cy[p];
        if (iabs(dx) > P.R || iabs(dy) > P.R) continue;
        int tie = spiral_pos(dx, dy) + 1;
        if (ff00 && mvx == 0 && mvy == 0) tie = 0;
        long long cost = (long long)mv_cost(P.lam_f, 4 * mvx - s_px[p], 4 * mvy - s_py[p]) + dist;
        if (p == 0 && quirk00 && 4 * (mbx * 16 + mvx) == mbx * 16 && 4 * (mby * 16 + mvy) == mby * 16) cost -= w16;
        const unsigned long long key = ((unsigned long long)(cost + COST_BIAS) << 16) | (unsigned)tie;
        best[p] = key < best[p] ? key : best[p];
      }
    }
#pragma unroll
    for (int p = 0; p < JMHIP_NPART; p++) if (best[p] != KEY64_INVALID) atomicMin(&s_key[p], best[p]);
    __syncthreads();
    if (tid < JMHIP_NPART && ((mask >> tid) & 1)) {
      const int p = tid;
      const unsigned long long k = s_key[p];
      int cost = (int)((long long)(k >> 16) - COST_BIAS), tie = (int)(k & 0xffffu);
      int mvx, mvy;
      if (tie == 0) { mvx = 0; mvy = 0; }
      else { int dx, dy; spiral_offset(tie - 1, &dx, &dy); mvx = s_cx[p] + dx; mvy = s_cy[p] + dy; }
      // the wrapped early-exit bound of spiral position 0 (see wrapped_bound_00v, me_int.hip): with a negative motion cost JM's
      // compute function leaves after its first row (SAD, SSE: me_distortion.c:373, :1066) or first sub-block (Hadamard: :690, :720)
      if (p == 0 && P.mode == JMHIP_SEARCH_FULL && !P.rdopt && !P.is_b && job.ref_is_0 && !mbx && !mby && !s_cx[0] && !s_cy[0]) {
        const int c0 = mv_cost(P.lam_f, -s_px[0], -s_py[0]) - w16;
        if (c0 < 0) {
          int part = 0;
          const uint8_t *ry = P.ref_y[slot];
          if (mf != 2) {
            for (int x = 0; x < 16; x++) { int rv = ry[x]; if (P.wp_on) rv = wp1(rv, wl, ol, P.wp_round, P.wp_denom); part += pel_err(mf == 1, (int)s_cur[x] - rv); }
          } else if (!t8) {
            int d[4][4];
            for (int r = 0; r < 4; r++) for (int x = 0; x < 4; x++) { int rv = ry[(size_t)r * P.W + x]; if (P.wp_on) rv = wp1(rv, wl, ol, P.wp_round, P.wp_denom); d[r][x] = (int)s_cur[r * 16 + x] - rv; }
            part = satd4x4(d);
          } else {
            int m2[8][8];
            for (int r = 0; r < 8; r++) {
              int row[8];
              for (int x = 0; x < 8; x++) { int rv = ry[(size_t)r * P.W + x]; if (P.wp_on) rv = wp1(rv, wl, ol, P.wp_round, P.wp_denom); row[x] = (int)s_cur[r * 16 + x] - rv; }
              had8(row);
              for (int x = 0; x < 8; x++) m2[r][x] = row[x];
            }
            part = satd8x8_rows(m2);
          }
          if (c0 + part <= cost) { mvx = 0; mvy = 0; cost = c0 + part; }
        }
      }
      s_mv[p][0] = mvx; s_mv[p][1] = mvy; s_cost[p] = cost;
      res.mv_int[p][0] = (int16_t)mvx; res.mv_int[p][1] = (int16_t)mvy; res.cost_int[p] = cost;
      if (!P.subpel) { res.mv[p][0] = (int16_t)(mvx << 2); res.mv[p][1] = (int16_t)(mvy << 2); res.cost[p] = cost; }
    }
  } else if (tid < JMHIP_NPART) {
    s_mv[tid][0] = res.mv_int[tid][0]; s_mv[tid][1] = res.mv_int[tid][1]; s_cost[tid] = res.cost_int[tid];
  }
  if (tid < JMHIP_NPART && !((mask >> tid) & 1)) {
    res.mv_int[tid][0] = res.mv_int[tid][1] = 0; res.cost_int[tid] = -1;
    res.mv[tid][0] = res.mv[tid][1] = 0; res.cost[tid] = -1;
  }
  if (!P.subpel) return;

  // ---- SubPelBlockMotionSearch, me_fullsearch.c:341-511: half-pel round the integer vector, quarter-pel round the half-pel winner
  for (int phase = 0; phase < 2; phase++) {
    __syncthreads();
    const int start = phase ? M.start_qp : M.start_hp;
    if (tid < JMHIP_NPART) {
      if (phase == 0) { s_mv[tid][0] <<= 2; s_mv[tid][1] <<= 2; }                    // mv-search.c:770-774
      s_key[tid] = start ? (((unsigned long long)((long long)s_cost[tid] + COST_BIAS) << 4) | 0u) : KEY64_INVALID;
    }
    __syncthreads();
    const int lam = phase ? P.lam_q : P.lam_h, metric = M.metric[1 + phase];
    for (int e = tid; e < JMHIP_NPART * 9; e += 256) {
      const int p = e / 9, k = e - p * 9;
      if (!((mask >> p) & 1) || k < start) continue;
      const PI q = part_info(p);
      const int bsx = q.w4 * 4, bsy = q.h4 * 4, pic_x = mbx * 16 + q.x4 * 4, pic_y = mby * 16 + q.y4 * 4;
      const int pic4x = (pic_x + 20) << 2, pic4y = (pic_y + 20) << 2;
      const int max_x4 = (P.W - bsx + 40) << 2, max_y4 = (P.H - bsy + 40) << 2;
      const int mx = s_mv[p][0], my = s_mv[p][1];
      const int lo = phase ? 0 : 1;                     // :412-420 (half-pel), :468-476 (quarter-pel)
      const int umv = !((pic4x + mx > lo) && (pic4x + mx < max_x4 - lo) && (pic4y + my > lo) && (pic4y + my < max_y4 - lo));
      const int cmx = mx + (c_s9x[k] << (1 - phase)), cmy = my + (c_s9y[k] << (1 - phase));
      long long cost = mv_cost(lam, cmx - s_px[p], cmy - s_py[p]);
      cost += block_dist(P, M, metric, M.chroma_sub, slot, pic_x, pic_y, bsx, bsy, cmx + pic4x, cmy + pic4y, umv, t8 && q.bt <= 4);
      if (phase == 0 && k == 0 && !P.rdopt && !P.is_b && job.ref_is_0 && q.bt == 1 && mx == 0 && my == 0) cost -= (lam * 16) >> 16;      // check_position0, :439-442
      atomicMin(&s_key[p], ((unsigned long long)(cost + COST_BIAS) << 4) | (unsigned)k);
    }
    __syncthreads();
    if (tid < JMHIP_NPART && ((mask >> tid) & 1)) {
      const unsigned long long kk = s_key[tid];
      const int k = (int)(kk & 15u);
      s_cost[tid] = (int)((long long)(kk >> 4) - COST_BIAS);
      s_mv[tid][0] += c_s9x[k] << (1 - phase); s_mv[tid][1] += c_s9y[k] << (1 - phase);
    }
  }
  __syncthreads();
  if (tid < JMHIP_NPART && ((mask >> tid) & 1)) { res.mv[tid][0] = (int16_t)s_mv[tid][0]; res.mv[tid][1] = (int16_t)s_mv[tid][1]; res.cost[tid] = s_cost[tid]; }
}

}  // namespace

// Host side: called by jmhip_me_frame_async / jmhip_me_subpel when jmhip_me_params.metric_set asks for it.
int jm_me_metric_check(jmhip_ctx *c, const jmhip_me_params *prm, unsigned ref_mask, const char *who)
{
  char msg[200];
  for (int k = 0; k < 3; k++)
    if (prm->metric[k] < 0 || prm->metric[k] > 2) { snprintf(msg, sizeof msg, "%s: metric[] must be 0 (SAD), 1 (SSE) or 2 (Hadamard SAD)", who); return jm_fail(c, JMHIP_ERR_ARG, msg); }
  if (prm->chroma_me < 0 || prm->chroma_me > 2) { snprintf(msg, sizeof msg, "%s: chroma_me must be 0, 1 or 2 (input->ChromaMEEnable)", who); return jm_fail(c, JMHIP_ERR_ARG, msg); }
  if (prm->wp_enable) {
    if (prm->chroma_me && (prm->wp_chroma_denom < 0 || prm->wp_chroma_denom > 7 || prm->wp_chroma_round < 0 || prm->wp_chroma_round > 64)) {
      snprintf(msg, sizeof msg, "%s: chroma weighted-prediction denominator / rounding out of range", who); return jm_fail(c, JMHIP_ERR_ARG, msg);
    }
  }
  if (prm->chroma_me) {
    if (!c->Wc || !c->cur_u || !c->cur_v) { snprintf(msg, sizeof msg, "%s: chroma_me needs chroma planes of the current picture", who); return jm_fail(c, JMHIP_ERR_ARG, msg); }
    if (prm->chroma_me_weight < 0 || prm->chroma_me_weight > 64) { snprintf(msg, sizeof msg, "%s: chroma_me_weight out of range", who); return jm_fail(c, JMHIP_ERR_ARG, msg); }
    for (size_t k = 0; k < c->refs.size(); k++)
      if (((ref_mask >> k) & 1) && !c->refs[k].has_cr_sub) { snprintf(msg, sizeof msg, "%s: chroma_me needs the eighth-pel chroma planes of the reference (jmhip_interp_chroma)", who); return jm_fail(c, JMHIP_ERR_ARG, msg); }
  }
  const bool sub_needed = prm->subpel || prm->chroma_me;
  (void)sub_needed;
  return JMHIP_OK;
}

int jm_launch_me_metric(jmhip_ctx *c, const jmhip_me_params *prm, const MeDev &P, const jmhip_me_mb *jobs_dev, const int *idx_dev,
                        jmhip_me_result *res_dev, int n, size_t lds, int skip_int)
{
  MetricDev M{};
  for (int k = 0; k < 3; k++) M.metric[k] = prm->metric[k];
  M.chroma_int = prm->chroma_me ? 1 : 0;                // mv-search.c:612
  M.chroma_sub = prm->chroma_me == 2 ? 1 : 0;           // mv-search.c:779 (ME_YUV_FP_SP)
  M.chroma_w = prm->chroma_me_weight;
  M.start_hp = (prm->chroma_me == 1 || prm->metric[0] != prm->metric[1]) ? 0 : 1;      // mv-search.c:396-397
  M.start_qp = (prm->chroma_me == 1 || prm->metric[1] != prm->metric[2]) ? 0 : 1;
  M.skip_int = skip_int;
  M.Wc = c->Wc; M.Hc = c->Hc; M.Wcp = c->Wcp; M.Hcp = c->Hcp;
  M.shift_x = c->cg.shift_x; M.shift_y = c->cg.shift_y; M.mask_x = c->cg.mask_x; M.mask_y = c->cg.mask_y; M.sub_x = c->cg.sub_x;
  M.mbc_w = c->cg.mb_w; M.mbc_h = c->cg.mb_h;
  M.wpad_c = (c->Wc - 1) + 2 * c->cg.pad_x - c->cg.mb_w; M.hpad_c = (c->Hc - 1) + 2 * c->cg.pad_y - c->cg.mb_h;      // image.c (size_x_cr_pad)
  M.csx = c->cg.shift_x - 2; M.csy = c->cg.shift_y - 2;
  M.cur_c[0] = c->cur_u; M.cur_c[1] = c->cur_v;
  M.ref_c[0] = reinterpret_cast<const uint8_t *const *>(c->ref_ptrs_dev) + 64;
  M.ref_c[1] = reinterpret_cast<const uint8_t *const *>(c->ref_ptrs_dev) + 96;
  M.wpc_round = prm->wp_chroma_round; M.wpc_denom = prm->wp_chroma_denom;
  for (int k = 0; k < 16; k++) for (int j = 0; j < 2; j++) { M.wpc_w[k][j] = prm->wp_weight_cr[k][j]; M.wpc_o[k][j] = prm->wp_offset_cr[k][j]; }
  if (M.chroma_int && !skip_int) {                      // chroma windows behind the luma window
    M.c_off = (int)((lds + 15) & ~(size_t)15);
    M.c_cw = ((P.win_pitch << 2) >> M.shift_x) + 2; M.c_ch = ((P.win_rows << 2) >> M.shift_y) + 2;
    M.nfx = M.shift_x == 3 ? 2 : 1; M.nfy = M.shift_y == 3 ? 2 : 1;
    lds = (size_t)M.c_off + (size_t)2 * M.nfx * M.nfy * M.c_cw * M.c_ch;
    if (lds > 63 * 1024) return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_me_frame: search window with the chroma term does not fit LDS (search range / spread of the predictors)");
  }
  me_metric_kernel<<<jm_xcd_grid(n), 256, lds, c->stream>>>(P, M, jobs_dev, idx_dev, res_dev, n);
  return JMHIP_OK;
}
