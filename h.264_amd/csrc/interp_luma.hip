// interp_luma.hip -- the 16 quarter-pel luma planes of one reference picture.
//
// Replaces getSubImagesLuma (lencod/src/img_luma.c:45-104) and its helpers:
//   [0][0] getSubImageInteger :120   [0][2] getHorSubImageSixTap :181   [2][0] getVerSubImageSixTap :293
//   [2][2] getVerSubImageSixTapTmp :390 (vertical taps on the RAW horizontal sums, (is+512)>>10)
//   12 bilinear planes :483-639 paired as :78-103.
// Semantics kept bit-exact: every tap coordinate is clamped to the PADDED plane (W+40 x H+40) first
// (:207-267, :308-367), then the padded plane maps to the picture by edge replication (20 >= 3 taps, so
// the two clamps compose); bilinear "+1 column / +1 row" operands clamp to the last column / row.
//
// Roofline: HBM-bound streaming. Algorithmic bytes per padded pel: 1 read + 16 written (8-bit samples).
// One workgroup = 256 columns x TY rows of the padded plane; the integer tile (+3/2 halo) and the raw
// horizontal sums (int16: range [-2550, 10710]) are staged in LDS, so the picture is read once from HBM
// (halo re-reads hit L2); each lane owns 4 adjacent pels and stores one dword per plane per row, a wave
// store covers 256 contiguous bytes.
#include "jmhip_internal.h"
#include <algorithm>
using std::max; using std::min;

namespace {

constexpr int TX = 256;          // tile width in pels (64 lanes x 4)
constexpr int TY = 8;            // tile height in rows (4 waves x 2)
constexpr int S00_W = TX + 8;    // cols i0-4 .. i0+TX+3
constexpr int ROWS = TY + 6;     // rows j0-2 .. j0+TY+3

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return min(max(x, lo), hi); }
// iClip1(255, v). The empty asm hides the preceding shift from hipcc: ROCm 7.2 otherwise fuses pairs of
// clip((x >> s), 0, 255) into gfx950's v_ashr_pk_u8_i32 and ORs further bytes into its result assuming the
// upper 16 bits are zero -- on MI355X they are not (bytes 2/3 of the packed dword came out corrupted).
__device__ __forceinline__ uint32_t clip255(int v) { asm volatile("" : "+v"(v)); return (uint32_t)min(max(v, 0), 255); }
__device__ __forceinline__ uint32_t pack4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return a | (b << 8) | (c << 16) | (d << 24); }
// per-byte (a + b + 1) >> 1 : v_lerp_u8 with all carry-in bits set
__device__ __forceinline__ uint32_t avg4(uint32_t a, uint32_t b) { return __builtin_amdgcn_lerp(a, b, 0x01010101u); }

__global__ __launch_bounds__(256) void interp_luma_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ out,
                                                         int W, int H, int Wp, int Hp, int tile_row0,
                                                         const uint8_t **fix_table, int fix_idx, const uint8_t *fix_ptr)
{
  // a pending entry of the reference pointer table (jmhip_recon_to_ref) rides along: later kernels of the stream read the table
  if (fix_table && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && threadIdx.y == 0) fix_table[fix_idx] = fix_ptr;
  __shared__ __attribute__((aligned(16))) uint8_t s00[ROWS][S00_W];
  __shared__ __attribute__((aligned(16))) int16_t stmp[ROWS][TX];

  const int tid = threadIdx.y * 64 + threadIdx.x;
  const int i0 = blockIdx.x * TX, j0 = (blockIdx.y + tile_row0) * TY;

  // ---- stage the integer tile: padded coords (row j0-2+r, col i0-4+c), clamped to the padded plane, then to the picture
  for (int d = tid; d < ROWS * (S00_W / 4); d += 256) {
    const int r = d / (S00_W / 4), cw = d - r * (S00_W / 4);
    const int pj = clampi(j0 - 2 + r, 0, Hp - 1);
    const int sy = clampi(pj - JMHIP_PAD, 0, H - 1);
    const int pc = i0 - 4 + cw * 4;                 // padded column of byte 0 (multiple of 4)
    const uint8_t *row = src + (size_t)sy * W;
    uint32_t v;
    const int sx = pc - JMHIP_PAD;                  // multiple of 4
    if (pc >= 0 && pc + 3 <= Wp - 1 && sx >= 0 && sx + 3 <= W - 1) {
      v = *reinterpret_cast<const uint32_t *>(row + sx);
    } else {
      uint32_t b[4];
#pragma unroll
      for (int k = 0; k < 4; k++) b[k] = row[clampi(clampi(pc + k, 0, Wp - 1) - JMHIP_PAD, 0, W - 1)];
      v = pack4(b[0], b[1], b[2], b[3]);
    }
    *reinterpret_cast<uint32_t *>(&s00[r][cw * 4]) = v;
  }
  __syncthreads();

  // ---- raw horizontal 6-tap sums for all staged rows: out col li..li+3 needs s00 cols li+2 .. li+10
  for (int d = tid; d < ROWS * (TX / 4); d += 256) {
    const int r = d / (TX / 4), g = d - r * (TX / 4);
    const uint32_t *p = reinterpret_cast<const uint32_t *>(&s00[r][g * 4]);
    const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
    int px[12];
#pragma unroll
    for (int k = 0; k < 4; k++) { px[k] = (w0 >> (8 * k)) & 255; px[4 + k] = (w1 >> (8 * k)) & 255; px[8 + k] = (w2 >> (8 * k)) & 255; }
    // pel i sits at px[4 + k]
    int16_t t[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int c = 4 + k;
      t[k] = (int16_t)(20 * (px[c] + px[c + 1]) - 5 * (px[c - 1] + px[c + 2]) + (px[c - 2] + px[c + 3]));
    }
    *reinterpret_cast<uint2 *>(&stmp[r][g * 4]) =
        make_uint2((uint32_t)(uint16_t)t[0] | ((uint32_t)(uint16_t)t[1] << 16), (uint32_t)(uint16_t)t[2] | ((uint32_t)(uint16_t)t[3] << 16));
  }
  __syncthreads();

  const size_t plane = (size_t)Wp * Hp;
  const int li = threadIdx.x * 4;                   // local column of this lane's 4 pels
  const int gi = i0 + li;
  if (gi >= Wp) return;                             // Wp % 4 == 0: a 4-pel group is all in or all out

#pragma unroll
  for (int rr = 0; rr < TY / 4; rr++) {
    const int jr = threadIdx.y + 4 * rr;            // local output row
    const int gj = j0 + jr;
    if (gj >= Hp) break;
    // staged row index of padded row (gj + k) is jr + 2 + k
    int p00[6][5];                                  // rows gj-2..gj+3, cols gi..gi+4
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const uint32_t a = *reinterpret_cast<const uint32_t *>(&s00[jr + k][li + 4]);
      const uint32_t b = s00[jr + k][li + 8];
#pragma unroll
      for (int x = 0; x < 4; x++) p00[k][x] = (a >> (8 * x)) & 255;
      p00[k][4] = b;
    }
    int tm[6][4];
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const uint2 t = *reinterpret_cast<const uint2 *>(&stmp[jr + k][li]);
      tm[k][0] = (int16_t)(t.x & 0xffff); tm[k][1] = (int16_t)(t.x >> 16);
      tm[k][2] = (int16_t)(t.y & 0xffff); tm[k][3] = (int16_t)(t.y >> 16);
    }
    uint32_t b02[4], b02n[4], b20[5], b22[4];
#pragma unroll
    for (int x = 0; x < 4; x++) {
      b02[x]  = clip255((tm[2][x] + 16) >> 5);
      b02n[x] = clip255((tm[3][x] + 16) >> 5);
      const int v22 = 20 * (tm[2][x] + tm[3][x]) - 5 * (tm[1][x] + tm[4][x]) + (tm[0][x] + tm[5][x]);
      b22[x] = clip255((v22 + 512) >> 10);
    }
#pragma unroll
    for (int x = 0; x < 5; x++) {
      const int v20 = 20 * (p00[2][x] + p00[3][x]) - 5 * (p00[1][x] + p00[4][x]) + (p00[0][x] + p00[5][x]);
      b20[x] = clip255((v20 + 16) >> 5);
    }
    const uint32_t q00  = pack4(p00[2][0], p00[2][1], p00[2][2], p00[2][3]);
    const uint32_t q00r = pack4(p00[2][1], p00[2][2], p00[2][3], p00[2][4]);     // column +1
    const uint32_t q00d = pack4(p00[3][0], p00[3][1], p00[3][2], p00[3][3]);     // row +1
    const uint32_t q02  = pack4(b02[0], b02[1], b02[2], b02[3]);
    const uint32_t q02d = pack4(b02n[0], b02n[1], b02n[2], b02n[3]);             // row +1
    const uint32_t q20  = pack4(b20[0], b20[1], b20[2], b20[3]);
    const uint32_t q20r = pack4(b20[1], b20[2], b20[3], b20[4]);                 // column +1
    const uint32_t q22  = pack4(b22[0], b22[1], b22[2], b22[3]);

    uint32_t *o = reinterpret_cast<uint32_t *>(out + (size_t)gj * Wp + gi);
    const size_t ps = plane / 4;                    // plane stride in dwords (Wp*Hp % 4 == 0)
    o[0 * ps]  = q00;                               // [0][0]
    o[1 * ps]  = avg4(q00, q02);                    // [0][1] img_luma.c:78
    o[2 * ps]  = q02;                               // [0][2]
    o[3 * ps]  = avg4(q02, q00r);                   // [0][3] :89
    o[4 * ps]  = avg4(q00, q20);                    // [1][0] :80
    o[5 * ps]  = avg4(q02, q20);                    // [1][1] :82
    o[6 * ps]  = avg4(q02, q22);                    // [1][2] :84
    o[7 * ps]  = avg4(q02, q20r);                   // [1][3] :91
    o[8 * ps]  = q20;                               // [2][0]
    o[9 * ps]  = avg4(q20, q22);                    // [2][1] :86
    o[10 * ps] = q22;                               // [2][2]
    o[11 * ps] = avg4(q22, q20r);                   // [2][3] :93
    o[12 * ps] = avg4(q20, q00d);                   // [3][0] :96
    o[13 * ps] = avg4(q20, q02d);                   // [3][1] :98
    o[14 * ps] = avg4(q22, q02d);                   // [3][2] :100
    o[15 * ps] = avg4(q02d, q20r);                  // [3][3] :103
  }
}

}  // namespace

// rows [prow0, prow1) of the padded plane, widened to whole tiles; the full plane when prow1 <= prow0
int jm_launch_interp_luma(jmhip_ctx *c, int ref, int prow0, int prow1)
{
  // operand shapes the kernel assumes
  if ((c->Wp & 3) || (c->W & 3) || ((size_t)c->Wp * c->Hp) % 4) return jm_fail(c, JMHIP_ERR_ARG, "interp_luma: width must be a multiple of 4");
  RefSlot &r = c->refs[ref];
  int t0 = 0, t1 = (c->Hp + TY - 1) / TY;
  if (prow1 > prow0) { t0 = max(0, prow0) / TY; t1 = min(t1, (min(c->Hp, prow1) + TY - 1) / TY); }
  if (t1 <= t0) return JMHIP_OK;
  dim3 grid((c->Wp + TX - 1) / TX, t1 - t0), block(64, 4);
  const bool fix = c->table_fix.idx >= 0 && c->ref_ptrs_dev;
  interp_luma_kernel<<<grid, block, 0, c->stream>>>(r.y, r.luma_sub, c->W, c->H, c->Wp, c->Hp, t0,
                                                   fix ? reinterpret_cast<const uint8_t **>(c->ref_ptrs_dev) : nullptr, c->table_fix.idx, c->table_fix.ptr);
  c->table_fix.idx = -1;
  JM_HIP_CHECK(c, hipGetLastError());
  return JMHIP_OK;
}
