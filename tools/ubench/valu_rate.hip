// Issue-rate microbenchmark for the integer VALU ops the ME kernel is made of (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP>
__global__ void k(unsigned *out, unsigned a, unsigned b, int iters, unsigned long long *clk)
{
  unsigned v[16];
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = threadIdx.x * 17 + i;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (OP == 0) v[i] = __builtin_amdgcn_sad_hi_u8(v[i], b, v[(i + 1) & 15]);
        else if (OP == 1) v[i] = v[i] + v[(i + 5) & 15] + a;              // v_add3_u32
        else if (OP == 2) v[i] = min(v[i], v[(i + 5) & 15] ^ a);           // v_xor + v_min
        else if (OP == 3) v[i] = (v[i] + v[(i + 3) & 15]) | a;               // v_add_u32 + v_or (or v_add_or)
        else if (OP == 4) { float f = __uint_as_float(v[i]); f = __builtin_fmaf(f, 1.0001f, __uint_as_float(a)); v[i] = __float_as_uint(f); }
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
int main()
{
  unsigned *out; unsigned long long *clk;
  hipMalloc(&out, 1 << 24); hipMalloc(&clk, 8);
  const char *names[5] = {"v_sad_hi_u8", "v_add3_u32", "v_xor+v_min_u32", "v_add+v_or", "v_fma_f32"};
  const int iters = 2000;
  for (int op = 0; op < 5; op++)
    for (int wps = 1; wps <= 8; wps *= 2) {        // waves per SIMD: block of 256*wps threads on every CU
      dim3 grid(wps == 8 ? 512 : 256), block(wps == 8 ? 1024 : 256 * wps);
      unsigned long long h = 0;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        switch (op) {
        case 0: k<0><<<grid, block>>>(out, 3, 5, iters, clk); break;
        case 1: k<1><<<grid, block>>>(out, 3, 5, iters, clk); break;
        case 2: k<2><<<grid, block>>>(out, 3, 5, iters, clk); break;
        case 3: k<3><<<grid, block>>>(out, 3, 5, iters, clk); break;
        default: k<4><<<grid, block>>>(out, 3, 5, iters, clk); break;
        }
        hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
      }
      const double ninstr = (double)iters * 64 * ((op == 2) ? 2 : 1);
      const double waves = (double)grid.x * block.x / 64; const double tot = waves * ninstr;
      printf("%-16s waves/SIMD=%d  ticks/instr(one wave)=%.2f  wall=%.3f ms  chip rate=%.3e wave-instr/s  => %.2f ns per instr per SIMD\n", names[op], wps, h / ninstr, ms, tot / (ms * 1e-3), (ms * 1e-3) / (tot / 1024) * 1e9);
    }
  return 0;
}
