// me_common.h -- what the motion-estimation translation units share: the partition table, the kernel parameter block and the
// small device helpers (costs, spiral order, search centre, Hadamard SATDs). Everything here has internal linkage: each
// translation unit that includes it gets its own copy of the __constant__ tables and uploads it itself.
#pragma once
#include "jmhip_internal.h"
#include <type_traits>
#include <cstring>
#include <cstdlib>

// kernel parameter block of the search / distortion kernels (a plain named struct: it crosses translation units)
struct MeDev {
  int mode, R, rdopt, is_b, lvl_min, lvl_max, lam_f, lam_h, lam_q, t8x8, subpel;
  unsigned long long mask;
  int W, H, Wp, Hp;
  int win_pitch, win_rows;            // LDS window geometry (bytes per row, rows)
  int win_copy_stride;                // fast path: dwords between the byte-shifted window copies
  unsigned long long *stamps;         // diagnostic build (-DJMHIP_STAMPS) only: per-wave section clocks
  const uint8_t *cur;
  const uint8_t *const *ref_y;        // [slot] integer recon
  const uint8_t *const *ref_sub;      // [slot] 16 quarter-pel planes
  int wp_on, wp_round, wp_denom;       // weighted reference ME: sample -> clip255(((w * p + round) >> denom) + o), per reference slot
  short wp_w[16], wp_o[16];
};

namespace {

struct PartInfo { int8_t bt, x4, y4, w4, h4; };
__constant__ PartInfo c_part[JMHIP_NPART];
PartInfo h_part[JMHIP_NPART];
bool h_part_ready = false;

void build_part_table()
{
  if (h_part_ready) return;
  int p = 0;
  h_part[p++] = {1, 0, 0, 4, 4};
  for (int k = 0; k < 2; k++) h_part[p++] = {2, 0, (int8_t)(2 * k), 4, 2};
  for (int k = 0; k < 2; k++) h_part[p++] = {3, (int8_t)(2 * k), 0, 2, 4};
  for (int b8 = 0; b8 < 4; b8++) h_part[p++] = {4, (int8_t)(2 * (b8 & 1)), (int8_t)(2 * (b8 >> 1)), 2, 2};
  for (int b8 = 0; b8 < 4; b8++) for (int k = 0; k < 2; k++) h_part[p++] = {5, (int8_t)(2 * (b8 & 1)), (int8_t)(2 * (b8 >> 1) + k), 2, 1};
  for (int b8 = 0; b8 < 4; b8++) for (int k = 0; k < 2; k++) h_part[p++] = {6, (int8_t)(2 * (b8 & 1) + k), (int8_t)(2 * (b8 >> 1)), 1, 2};
  for (int b8 = 0; b8 < 4; b8++) for (int k = 0; k < 4; k++) h_part[p++] = {7, (int8_t)(2 * (b8 & 1) + (k & 1)), (int8_t)(2 * (b8 >> 1) + (k >> 1)), 1, 1};
  h_part_ready = true;
}



__device__ __forceinline__ int clampi(int x, int lo, int hi) { return min(max(x, lo), hi); }
__device__ __forceinline__ int iabs(int x) { return x < 0 ? -x : x; }
// mvbits[d] (mv-search.c:333-341): 1 for 0, 2*floor(log2|d|)+3 otherwise == 65 - 2*clz(|d|) with clz(0) = 32
__device__ __forceinline__ int mvbits(int d) { return 65 - 2 * __clz(iabs(d)); }
__device__ __forceinline__ int mv_cost(int lambda, int dx, int dy) { return (lambda * (mvbits(dx) + mvbits(dy))) >> 16; }

// spiral_search index of offset (dx,dy), mv-search.c:366-393
__device__ __forceinline__ int spiral_pos(int dx, int dy)
{
  const int adx = iabs(dx), ady = iabs(dy), l = max(adx, ady);
  if (l == 0) return 0;
  const int k0 = (2 * l - 1) * (2 * l - 1);
  if (ady == l && adx < l) return k0 + 2 * (dx + l - 1) + (dy > 0 ? 1 : 0);
  return k0 + 2 * (2 * l - 1) + 2 * (dy + l) + (dx > 0 ? 1 : 0);
}
__device__ void spiral_offset(int pos, int *dx, int *dy)
{
  if (pos == 0) { *dx = 0; *dy = 0; return; }
  int l = (int)((__builtin_sqrtf((float)pos) + 1.0f) * 0.5f);       // ring: (2l-1)^2 <= pos < (2l+1)^2; float guess, exact fix-up
  l = max(l, 1);
  while ((2 * l + 1) * (2 * l + 1) <= pos) l++;
  while ((2 * l - 1) * (2 * l - 1) > pos) l--;
  int k = pos - (2 * l - 1) * (2 * l - 1);
  if (k < 2 * (2 * l - 1)) { *dx = (k >> 1) - l + 1; *dy = (k & 1) ? l : -l; }
  else { k -= 2 * (2 * l - 1); *dy = (k >> 1) - l; *dx = (k & 1) ? l : -l; }
}

// search centre from the predictor: mv-search.c:752-762 / me_fullfast.c:552-563
__device__ __forceinline__ void search_center(const MeDev &P, int pmx, int pmy, int *cx, int *cy)
{
  int mx = pmx / 4, my = pmy / 4;
  if (!P.rdopt) { mx = clampi(mx, -P.R, P.R); my = clampi(my, -P.R, P.R); }
  mx = clampi(mx, -2047 + P.R, 2047 - P.R);
  my = clampi(my, P.lvl_min + P.R, P.lvl_max - P.R);
  *cx = mx; *cy = my;
}

constexpr int TIE_BITS = 13;            // (2*64+1)^2 + 1 < 2^15 would need 15; R <= 44 fits 13 bits: checked on the host
constexpr unsigned KEY_INVALID = 0xffffffffu;

// Hadamard SATD of a 4x4 difference block held as four rows of four ints: (sum|H D H| + 1) >> 1, me_distortion.c:182
__device__ __forceinline__ int satd4x4(const int d[4][4])
{
  int m[4][4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {        // columns
    const int a = d[0][k] + d[3][k], b = d[1][k] + d[2][k], c = d[1][k] - d[2][k], e = d[0][k] - d[3][k];
    m[0][k] = a + b; m[1][k] = e + c; m[2][k] = a - b; m[3][k] = e - c;
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {        // rows
    const int a = m[k][0] + m[k][3], b = m[k][1] + m[k][2], c = m[k][1] - m[k][2], e = m[k][0] - m[k][3];
    s += iabs(a + b) + iabs(a - b) + iabs(c + e) + iabs(e - c);
  }
  return (s + 1) >> 1;
}

// one 8-point Hadamard butterfly network (me_distortion.c:280-305 per row, :311-336 per column)
__device__ __forceinline__ void had8(int v[8])
{
  int a[8], b[8];
#pragma unroll
  for (int i = 0; i < 4; i++) { a[i] = v[i] + v[i + 4]; a[i + 4] = v[i] - v[i + 4]; }
  b[0] = a[0] + a[2]; b[1] = a[1] + a[3]; b[2] = a[0] - a[2]; b[3] = a[1] - a[3];
  b[4] = a[4] + a[6]; b[5] = a[5] + a[7]; b[6] = a[4] - a[6]; b[7] = a[5] - a[7];
#pragma unroll
  for (int i = 0; i < 8; i += 2) { v[i] = b[i] + b[i + 1]; v[i + 1] = b[i] - b[i + 1]; }
}

// fetch `n` (4 or 8) samples of one row at byte address p (any alignment) from a quarter-pel plane
// four packed reference samples through the explicit-weight formula (computeSADWP, me_distortion.c:431)
__device__ __forceinline__ uint32_t wp_apply4(uint32_t v, int w, int o, int rnd, int den)
{
  uint32_t r = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int q = (((w * (int)((v >> (8 * k)) & 255u)) + rnd) >> den) + o;
    r |= (uint32_t)min(max(q, 0), 255) << (8 * k);
  }
  return r;
}

__device__ __forceinline__ void fetch_row(const uint8_t *p, int n, uint32_t *lo, uint32_t *hi)
{
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  const uint32_t *q = reinterpret_cast<const uint32_t *>(a & ~uintptr_t(3));
  const unsigned sh = (unsigned)(a & 3);
  const uint32_t d0 = q[0], d1 = q[1];
  *lo = __builtin_amdgcn_alignbyte(d1, d0, sh);
  if (n == 8) { const uint32_t d2 = q[2]; *hi = __builtin_amdgcn_alignbyte(d2, d1, sh); }
}

__constant__ int c_s9x[9] = {0, 0, 0, -1, 1, -1, 1, -1, 1};    // spiral positions 0..8, mv-search.c:366-393
__constant__ int c_s9y[9] = {0, -1, 1, -1, -1, 0, 0, 1, 1};

// HadamardSAD4x4 (me_distortion.c:182) on packed 16-bit lanes. c01/c23: the current block's rows as (x0,x1)/(x2,x3)
// pairs of 16-bit samples, with 0x8000 added to x1 of row 0 (done once when the macroblock is staged); ref: the four
// reference rows as packed bytes. The bias rides through every butterfly into all sixteen outputs -- each output holds
// exactly one biased input with a + sign, or the difference of two -- so |coef| = |biased - 0x8000| is one v_sad_u16 per
// output pair. Intermediate values are at most 16*255 in magnitude: no 16-bit overflow.
typedef short v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s as_v2s(uint32_t u) { return __builtin_bit_cast(v2s, u); }
__device__ __forceinline__ uint32_t as_u32(v2s v) { return __builtin_bit_cast(uint32_t, v); }

__device__ __forceinline__ int satd4x4_packed(const uint32_t c01[4], const uint32_t c23[4], const uint32_t ref[4])
{
  v2s d01[4], d23[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const uint32_t rlo = __builtin_amdgcn_perm(0u, ref[r], 0x0c010c00u), rhi = __builtin_amdgcn_perm(0u, ref[r], 0x0c030c02u);
    d01[r] = as_v2s(c01[r]) - as_v2s(rlo);
    d23[r] = as_v2s(c23[r]) - as_v2s(rhi);
  }
  v2s m01[4], m23[4];
  {
    const v2s a = d01[0] + d01[3], b = d01[1] + d01[2], c = d01[1] - d01[2], e = d01[0] - d01[3];
    m01[0] = a + b; m01[1] = e + c; m01[2] = a - b; m01[3] = e - c;
  }
  {
    const v2s a = d23[0] + d23[3], b = d23[1] + d23[2], c = d23[1] - d23[2], e = d23[0] - d23[3];
    m23[0] = a + b; m23[1] = e + c; m23[2] = a - b; m23[3] = e - c;
  }
  uint32_t acc = 0;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const uint32_t hi = as_u32(m23[r]);
    const v2s sw = as_v2s(__builtin_amdgcn_alignbit(hi, hi, 16));           // (x3, x2)
    const uint32_t s = as_u32(m01[r] + sw), t = as_u32(m01[r] - sw);       // (x0+x3, x1+x2+B), (x0-x3, x1-x2+B)
    const v2s u = as_v2s(__builtin_amdgcn_perm(t, s, 0x05040100u));        // (s.lo, t.lo)
    const v2s v = as_v2s(__builtin_amdgcn_perm(t, s, 0x07060302u));        // (s.hi, t.hi), both biased
    acc = __builtin_amdgcn_sad_u16(as_u32(u + v), 0x80008000u, acc);
    acc = __builtin_amdgcn_sad_u16(as_u32(u - v), 0x80008000u, acc);
  }
  return (int)((acc + 1) >> 1);
}

}  // namespace
