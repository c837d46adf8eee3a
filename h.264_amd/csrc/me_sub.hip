// me_sub.hip -- half- and quarter-pel refinement of all 41 partitions (SubPelBlockMotionSearch, lencod/src/me_fullsearch.c:341-511;
// computeSATD me_distortion.c:657, HadamardSAD4x4 :182, HadamardSAD8x8 :272). Entry: jmhip_me_subpel; jmhip_me_frame (me_int.hip)
// chains into it through jm_launch_me_sub.
#include "me_common.h"

#ifndef SUB_NT
#define SUB_NT 128    // threads per macroblock (measured at 1080p: 64 -> 0.078 ms, 128 -> 0.064, 256 -> 0.066)
#endif

namespace {

// One SATD sub-block of a partition. `mem`: all items (of any partition) on the same sub-block -- the seven block types
// tile the macroblock with the same 4x4 (8x8) blocks --, ascending, -1 padded; `memp`: their partitions.
struct SubItem { int8_t p, bx, by, bs; int8_t mem[7], memp[7]; int8_t pad[2]; };     // 20 bytes
static_assert(sizeof(SubItem) == 20, "SubItem is staged to LDS as 5 dwords");
__constant__ SubItem c_sub4[112];                   // all partitions, 4x4 sub-blocks
__constant__ SubItem c_sub8[64];                    // test8x8transform: 16 8x8 blocks (types <= 4) + 48 4x4 (types 5..7)
SubItem h_sub4[112], h_sub8[64];

void build_sub_tables()
{
  build_part_table();
  int n4 = 0, n8 = 0;
  auto put = [](SubItem *t, int &n, int p, int bx, int by, int bs) {
    SubItem s{};
    s.p = (int8_t)p; s.bx = (int8_t)bx; s.by = (int8_t)by; s.bs = (int8_t)bs;
    t[n++] = s;
  };
  for (int p = 0; p < JMHIP_NPART; p++) {
    const PartInfo &q = h_part[p];
    for (int y = 0; y < q.h4; y++) for (int x = 0; x < q.w4; x++)       // computeSATD order: y outer, x inner (:673-675)
      put(h_sub4, n4, p, 4 * (q.x4 + x), 4 * (q.y4 + y), 4);
    if (q.bt <= 4) {
      for (int y = 0; y < q.h4 / 2; y++) for (int x = 0; x < q.w4 / 2; x++) put(h_sub8, n8, p, 4 * q.x4 + 8 * x, 4 * q.y4 + 8 * y, 8);
    } else {
      for (int y = 0; y < q.h4; y++) for (int x = 0; x < q.w4; x++) put(h_sub8, n8, p, 4 * (q.x4 + x), 4 * (q.y4 + y), 4);
    }
  }
  auto link = [](SubItem *it, int n) {
    for (int i = 0; i < n; i++) {
      int k = 0;
      for (int j = 0; j < 7; j++) { it[i].mem[j] = -1; it[i].memp[j] = -1; }
      for (int j = 0; j < n; j++)
        if (it[j].bx == it[i].bx && it[j].by == it[i].by && it[j].bs == it[i].bs) { it[i].mem[k] = (int8_t)j; it[i].memp[k] = it[j].p; k++; }
    }
  };
  link(h_sub4, 112);
  link(h_sub8, 64);
}

#ifdef JMHIP_STAMPS
#define SSTAMP(k) do { if (P.stamps && blockIdx.x < 512 && threadIdx.x == 0) P.stamps[blockIdx.x * 32 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SSTAMP(k) do { } while (0)
#endif

// Sub-pel refinement of all 41 partitions of one macroblock (SubPelBlockMotionSearch, me_fullsearch.c:341).
// Per phase (half-pel: 9 positions, quarter-pel: 8): the SATD of a (4x4 | 8x8) sub-block at a candidate depends only on
// the sub-block, the partition's current vector and its access method. The seven block types tile the macroblock with
// the same sub-blocks, so wherever their vectors agree the value is shared: each distinct (sub-block, vector) is
// evaluated once by a "leader" item (compacted list) that adds it to every partition it serves; the 9 candidates of
// all partitions are then costed in parallel and JM's sequential strict-< scan becomes an atomic min on (cost, position).
// The kernel is latency-bound (dependent LDS/global reads between barriers), hence the small register footprint
// (T8 = false drops the 8x8 Hadamard path) for many resident workgroups.
// LIST: the macroblocks are a device-resident list of job indices with a partition mask each, its length device-resident too (the slice
// search's relaxation sweeps, me_xslice.hip): a fixed grid strides over it; only the masked partitions are refined and written.
template <bool T8, int NT, bool WP, bool LIST>
__global__ __launch_bounds__(NT) void me_sub_kernel(MeDev P, const jmhip_me_mb *__restrict__ jobs, jmhip_me_result *__restrict__ res, int n_items,
                                                    const int *__restrict__ list, const unsigned long long *__restrict__ masks, const int *__restrict__ n_dev, int list_cap)
{
  constexpr int NSUB = T8 ? 64 : 112;
  __shared__ __attribute__((aligned(16))) uint8_t s_cur[16][16];
  __shared__ __attribute__((aligned(16))) uint32_t s_c16[16][8];           // (x, x+1) sample pairs, biased (see satd4x4_packed)
  __shared__ __attribute__((aligned(16))) uint32_t s_tab[NSUB * 5];         // the SubItem table
  __shared__ int s_mvx[JMHIP_NPART], s_mvy[JMHIP_NPART], s_umv[JMHIP_NPART], s_px[JMHIP_NPART], s_py[JMHIP_NPART];
  __shared__ unsigned s_pkey[JMHIP_NPART], s_best[JMHIP_NPART];
  __shared__ int s_satd[JMHIP_NPART][9];
  __shared__ short s_list[NSUB];
  __shared__ int s_nlead;

  const int tid = threadIdx.x;
  if (LIST) n_items = jm_shard_slots(n_dev);                          // sharded list (jmhip_internal.h): virtual slots
  for (int vblock = blockIdx.x; vblock < jm_xcd_grid(n_items); vblock += gridDim.x) {
  if (LIST && vblock != (int)blockIdx.x) __syncthreads();              // the previous trip's last readers of the shared arrays
  const int item0 = jm_xcd_item_of(vblock, n_items);
  if (item0 < 0) { if (LIST) continue; return; }
  const int item = LIST ? jm_shard_entry(n_dev, list, list_cap, item0) : item0;
  if (LIST && item < 0) continue;
  SSTAMP(0);
  const jmhip_me_mb &job = jobs[item];
  jmhip_me_result &o = res[item];
  const int mbx = job.mb_x, mby = job.mb_y;
  const unsigned long long mask = LIST ? masks[item] : P.mask;
  const uint8_t *sub = P.ref_sub[job.ref];
  const int wpw = WP ? P.wp_w[job.ref] : 0, wpo = WP ? P.wp_o[job.ref] : 0;       // WP: weighted reference ME, a separate instantiation
  const size_t plane = (size_t)P.Wp * P.Hp;
  const int width_pad = P.Wp - 1 - 16, height_pad = P.Hp - 1 - 16;      // size_x_pad / size_y_pad, mbuffer.c:421-422
  const SubItem *items = reinterpret_cast<const SubItem *>(s_tab);

  for (int d = tid; d < NSUB * 5; d += NT) s_tab[d] = reinterpret_cast<const uint32_t *>(T8 ? c_sub8 : c_sub4)[d];
  if (tid < 64) {
    const int r = tid >> 2, k = tid & 3;
    const uint32_t v = *reinterpret_cast<const uint32_t *>(P.cur + (size_t)(mby * 16 + r) * P.W + mbx * 16 + k * 4);
    *reinterpret_cast<uint32_t *>(&s_cur[r][k * 4]) = v;
    const uint32_t bias = (r & 3) ? 0u : 0x80000000u;
    s_c16[r][2 * k] = __builtin_amdgcn_perm(0u, v, 0x0c010c00u) + bias;
    s_c16[r][2 * k + 1] = __builtin_amdgcn_perm(0u, v, 0x0c030c02u);
  }
  const bool my_active = tid < JMHIP_NPART && ((mask >> tid) & 1);
  if (tid < JMHIP_NPART) {
    s_px[tid] = job.pred_mv[tid][0]; s_py[tid] = job.pred_mv[tid][1];
    s_mvx[tid] = o.mv_int[tid][0] << 2; s_mvy[tid] = o.mv_int[tid][1] << 2;      // mv-search.c:770-774
  }
  const int w16h = (P.lam_h * 16) >> 16;
  unsigned carried = 0xffffffffu;                    // lane p: running minimum as a (cost + bias) << 4 | position key

  for (int phase = 0; phase < 2; phase++) {        // 0: half-pel (positions 0..8, step 2), 1: quarter-pel (1..8, step 1)
    const int step = phase ? 1 : 2, first = phase ? 1 : 0, ncand = 9 - first;
    __syncthreads();
    if (tid < JMHIP_NPART) {
      const PartInfo q = c_part[tid];
      const int bsx = 4 * q.w4, bsy = 4 * q.h4;
      const int mx = s_mvx[tid], my = s_mvy[tid];
      const int p4x = ((mbx * 16 + 4 * q.x4 + JMHIP_PAD) << 2) + mx, p4y = ((mby * 16 + 4 * q.y4 + JMHIP_PAD) << 2) + my;
      const int max_x4 = (P.W - bsx + 2 * JMHIP_PAD) << 2, max_y4 = (P.H - bsy + 2 * JMHIP_PAD) << 2;
      const int m = phase ? 0 : 1;                   // me_fullsearch.c:412-413 vs :468-469
      const int umv = !((p4x > m) && (p4x < max_x4 - m) && (p4y > m) && (p4y < max_y4 - m));
      s_umv[tid] = umv;
      // identity of (vector, access method); an inactive partition gets a key nobody shares
      s_pkey[tid] = my_active ? ((unsigned)(mx + 16384) | ((unsigned)(my + 16384) << 15) | ((unsigned)umv << 30)) : (0x80000000u | (unsigned)tid);
      s_best[tid] = carried;                         // :785-788: INT_MAX before half-pel; the half-pel minimum is carried on
#pragma unroll
      for (int k = 0; k < 9; k++) s_satd[tid][k] = 0;
    }
    __syncthreads();
    SSTAMP(1 + 5 * phase);

    // ---- leaders: wave 0 takes items lane and lane + 64; an item leads if no earlier item on its sub-block has its key
    if (tid < 64) {
      bool lead[2];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int it = tid + 64 * h;
        lead[h] = false;
        if (it < NSUB) {
          const SubItem si = items[it];
          const unsigned key = s_pkey[si.p];
          bool l = !(key & 0x80000000u);
#pragma unroll
          for (int k = 0; k < 6; k++) if (si.mem[k] >= 0 && si.mem[k] < it && s_pkey[si.memp[k]] == key) l = false;
          lead[h] = l;
        }
      }
      const unsigned long long b0 = __ballot(lead[0]), b1 = __ballot(lead[1]);
      const unsigned long long lt = (1ull << tid) - 1;
      if (lead[0]) s_list[__popcll(b0 & lt)] = (short)tid;
      if (lead[1]) s_list[__popcll(b0) + __popcll(b1 & lt)] = (short)(tid + 64);
      if (tid == 0) s_nlead = __popcll(b0) + __popcll(b1);
    }
    __syncthreads();
    SSTAMP(2 + 5 * phase);

    const int K = s_nlead, total = K * ncand;
    // idx / K without the integer-division sequence: idx + 0.5 is never a multiple of K, so the float quotient stays at least 0.5 / K
    // (>= 0.0045) away from an integer while its error is below 1e-4 for idx < 1008
    const float rK = __builtin_amdgcn_rcpf((float)K);
    for (int idx = tid; idx < total; idx += NT) {
      const int ci = (int)(((float)idx + 0.5f) * rK), it = s_list[idx - ci * K], cand = first + ci;     // adjacent lanes: adjacent sub-blocks, same plane
      const SubItem si = items[it];
      const int p = si.p;
      // quarter-pel coordinate of the sub-block origin incl. the pad offset (me_fullsearch.c:364-365, me_distortion.c:678)
      const int xq = ((mbx * 16 + si.bx + JMHIP_PAD) << 2) + s_mvx[p] + step * c_s9x[cand];
      const int yq = ((mby * 16 + si.by + JMHIP_PAD) << 2) + s_mvy[p] + step * c_s9y[cand];
      int xpos = xq >> 2, ypos = yq >> 2;
      if (s_umv[p]) { xpos = clampi(xpos, 0, width_pad); ypos = clampi(ypos, 0, height_pad); }     // UMVLine4X, refbuf.c:37
      const uint8_t *rp = sub + (size_t)((yq & 3) * 4 + (xq & 3)) * plane + (size_t)ypos * P.Wp + xpos;
      int v;
      if (!T8 || si.bs == 4) {
        uint32_t ref[4], c01[4], c23[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          uint32_t hi;
          fetch_row(rp + (size_t)r * P.Wp, 4, &ref[r], &hi);
          if (WP) ref[r] = wp_apply4(ref[r], wpw, wpo, P.wp_round, P.wp_denom);             // computeSATDWP, me_distortion.c:734
          const uint2 c = *reinterpret_cast<const uint2 *>(&s_c16[si.by + r][si.bx >> 1]);
          c01[r] = c.x; c23[r] = c.y;
        }
        v = satd4x4_packed(c01, c23, ref);
      } else {
        int m2[8][8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
          uint32_t lo, hi;
          fetch_row(rp + (size_t)r * P.Wp, 8, &lo, &hi);
          if (WP) { lo = wp_apply4(lo, wpw, wpo, P.wp_round, P.wp_denom); hi = wp_apply4(hi, wpw, wpo, P.wp_round, P.wp_denom); }
          const uint32_t c0 = *reinterpret_cast<const uint32_t *>(&s_cur[si.by + r][si.bx]);
          const uint32_t c1 = *reinterpret_cast<const uint32_t *>(&s_cur[si.by + r][si.bx + 4]);
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
        v = (s + 2) >> 2;                            // HadamardSAD8x8, me_distortion.c:342
      }
      // the value serves every item on this sub-block whose partition has the same key (itself included)
      const unsigned key = s_pkey[p];
#pragma unroll
      for (int k = 0; k < 7; k++) if (si.mem[k] >= 0 && s_pkey[si.memp[k]] == key) atomicAdd(&s_satd[si.memp[k]][cand], v);
    }
    __syncthreads();
    SSTAMP(3 + 5 * phase);

    // ---- all (partition, position) costs in parallel; strict-< in scan order == min over (cost, position)
    const int lam = phase ? P.lam_q : P.lam_h;
    for (int idx = tid; idx < JMHIP_NPART * ncand; idx += NT) {
      const int ci = idx / JMHIP_NPART, p = idx - ci * JMHIP_NPART, pos = first + ci;
      if (!((mask >> p) & 1)) continue;
      const int mvx = s_mvx[p], mvy = s_mvy[p];
      const int cxm = mvx + step * c_s9x[pos], cym = mvy + step * c_s9y[pos];
      int mcost = mv_cost(lam, cxm - s_px[p], cym - s_py[p]) + s_satd[p][pos];
      // check_position0, me_fullsearch.c:361, :439-442 (half-pel position 0 only; the bias keeps the key unsigned)
      if (pos == 0 && !P.rdopt && !P.is_b && job.ref_is_0 && p == 0 && mvx == 0 && mvy == 0) mcost -= w16h;
      atomicMin(&s_best[p], ((unsigned)(mcost + w16h) << 4) | (unsigned)pos);
    }
    __syncthreads();
    SSTAMP(4 + 5 * phase);
    if (my_active) {
      const unsigned k = s_best[tid];
      const int best = (int)(k & 15u);
      s_mvx[tid] += step * c_s9x[best]; s_mvy[tid] += step * c_s9y[best];
      carried = (k & ~15u);                          // start_me_refinement_qp == 1: the minimum competes as position 0
    }
    SSTAMP(5 + 5 * phase);
  }
  if (my_active) {
    o.mv[tid][0] = (int16_t)s_mvx[tid]; o.mv[tid][1] = (int16_t)s_mvy[tid]; o.cost[tid] = (int)(carried >> 4) - w16h;
  }
  SSTAMP(11);
  if (!LIST) break;
  }
}

}  // namespace

// partition + sub-block tables of this translation unit
int jm_me_sub_tables(jmhip_ctx *c)
{
  static bool uploaded[64] = {false};
  const int dev = c->cfg.device;
  if (dev < 64 && uploaded[dev]) return JMHIP_OK;
  build_sub_tables();
  JM_HIP_CHECK(c, hipMemcpyToSymbol(HIP_SYMBOL(c_part), h_part, sizeof(h_part)));
  JM_HIP_CHECK(c, hipMemcpyToSymbol(HIP_SYMBOL(c_sub4), h_sub4, sizeof(h_sub4)));
  JM_HIP_CHECK(c, hipMemcpyToSymbol(HIP_SYMBOL(c_sub8), h_sub8, sizeof(h_sub8)));
  if (dev < 64) uploaded[dev] = true;
  return JMHIP_OK;
}

void jm_launch_me_sub(jmhip_ctx *c, const MeDev &P, const jmhip_me_mb *jobs_dev, jmhip_me_result *res_dev, int n)
{
  if (P.wp_on) {
    if (P.t8x8) me_sub_kernel<true, 128, true, false><<<jm_xcd_grid(n), 128, 0, c->stream>>>(P, jobs_dev, res_dev, n, nullptr, nullptr, nullptr, 0);
    else me_sub_kernel<false, SUB_NT, true, false><<<jm_xcd_grid(n), SUB_NT, 0, c->stream>>>(P, jobs_dev, res_dev, n, nullptr, nullptr, nullptr, 0);
  } else if (P.t8x8) me_sub_kernel<true, 128, false, false><<<jm_xcd_grid(n), 128, 0, c->stream>>>(P, jobs_dev, res_dev, n, nullptr, nullptr, nullptr, 0);
  else me_sub_kernel<false, SUB_NT, false, false><<<jm_xcd_grid(n), SUB_NT, 0, c->stream>>>(P, jobs_dev, res_dev, n, nullptr, nullptr, nullptr, 0);
}

// the list form (4x4 Hadamard only: the slice search's exhaustive path runs without the 8x8 transform): list[i] = job index, masks[job index] = partitions
void jm_launch_me_sub_list(jmhip_ctx *c, const MeDev &P, const jmhip_me_mb *jobs_dev, jmhip_me_result *res_dev, const int *list_dev, const unsigned long long *masks_dev, const int *cnt_dev, int cap, int grid)
{
  if (P.wp_on) me_sub_kernel<false, SUB_NT, true, true><<<grid, SUB_NT, 0, c->stream>>>(P, jobs_dev, res_dev, 0, list_dev, masks_dev, cnt_dev, cap);
  else me_sub_kernel<false, SUB_NT, false, true><<<grid, SUB_NT, 0, c->stream>>>(P, jobs_dev, res_dev, 0, list_dev, masks_dev, cnt_dev, cap);
}

extern "C" int jmhip_me_subpel(jmhip_ctx *c, const jmhip_me_params *prm, const jmhip_me_mb *mbs, int n, jmhip_me_result *results)
{
  if (!c || !prm || !mbs || !results || n <= 0) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_me_subpel: NULL/empty arguments") : JMHIP_ERR_ARG;
  // run the integer stage's validation/upload with the search itself switched off: reuse jmhip_me_frame_async with an empty
  // candidate set is not possible, so validate here
  if (!c->has_cur) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_me_subpel: current picture not uploaded");
  for (int i = 0; i < n; i++) {
    const jmhip_me_mb &m = mbs[i];
    if (m.mb_x < 0 || m.mb_x >= c->mbw || m.mb_y < 0 || m.mb_y >= c->mbh) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_me_subpel: macroblock outside the picture");
    if (m.ref < 0 || m.ref >= (int)c->refs.size() || !c->refs[m.ref].has_luma_sub) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_me_subpel: sub-pel planes of the reference not built");
    for (int p = 0; p < JMHIP_NPART; p++)
      if (((prm->partition_mask >> p) & 1) && (results[i].mv_int[p][0] < -2048 || results[i].mv_int[p][0] > 2048 || results[i].mv_int[p][1] < -2048 || results[i].mv_int[p][1] > 2048))
        return jm_fail(c, JMHIP_ERR_ARG, "jmhip_me_subpel: integer vector out of range");
  }
  for (int k = 0; k < 3; k++)
    if (prm->lambda[k] < 0 || prm->lambda[k] > 30000000) return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_me_subpel: lambda factor out of the 32-bit cost range");
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  int rc = jm_me_sub_tables(c);
  if (rc) return rc;
  if ((rc = jm_ensure_ref_table(c))) return rc;
  const bool metric_path = jm_me_metric_path(prm);
  if (metric_path) {
    unsigned ref_mask = 0;
    for (int i = 0; i < n; i++) ref_mask |= 1u << mbs[i].ref;
    if ((rc = jm_me_metric_check(c, prm, ref_mask, "jmhip_me_subpel"))) return rc;
  }
  void *dj = nullptr, *dr = nullptr;
  if (hipMalloc(&dj, sizeof(jmhip_me_mb) * (size_t)n) != hipSuccess || hipMalloc(&dr, sizeof(jmhip_me_result) * (size_t)n) != hipSuccess) {
    (void)hipFree(dj); return jm_fail(c, JMHIP_ERR_NOMEM, "sub-pel arrays");
  }
  MeDev P{};
  P.mode = prm->search_mode; P.R = prm->search_range; P.rdopt = prm->rdopt; P.is_b = prm->is_b_slice;
  P.lam_f = prm->lambda[0]; P.lam_h = prm->lambda[1]; P.lam_q = prm->lambda[2];
  P.t8x8 = prm->transform8x8_mode ? 1 : 0; P.subpel = 1;
  P.wp_on = prm->wp_enable ? 1 : 0; P.wp_round = prm->wp_round; P.wp_denom = prm->wp_denom;
  for (int k = 0; k < 16; k++) { P.wp_w[k] = prm->wp_weight[k]; P.wp_o[k] = prm->wp_offset[k]; }
  P.mask = prm->partition_mask & ((1ull << JMHIP_NPART) - 1);
  P.W = c->W; P.H = c->H; P.Wp = c->Wp; P.Hp = c->Hp; P.cur = c->cur_y;
  P.ref_y = reinterpret_cast<const uint8_t *const *>(c->ref_ptrs_dev);
  P.ref_sub = reinterpret_cast<const uint8_t *const *>(c->ref_ptrs_dev) + 32;
  hipError_t e = hipMemcpyAsync(dj, mbs, sizeof(jmhip_me_mb) * (size_t)n, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dr, results, sizeof(jmhip_me_result) * (size_t)n, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    jm_stage_begin(c, JMHIP_STAGE_ME_SUB);
    if (metric_path) jm_launch_me_metric(c, prm, P, (const jmhip_me_mb *)dj, nullptr, (jmhip_me_result *)dr, n, 0, 1);
    else jm_launch_me_sub(c, P, (const jmhip_me_mb *)dj, (jmhip_me_result *)dr, n);
    jm_stage_end(c, JMHIP_STAGE_ME_SUB);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(results, dr, sizeof(jmhip_me_result) * (size_t)n, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(dj); (void)hipFree(dr);
  if (e != hipSuccess) { c->err = std::string("jmhip_me_subpel: ") + hipGetErrorString(e); return JMHIP_ERR_DEVICE; }
  return JMHIP_OK;
}
