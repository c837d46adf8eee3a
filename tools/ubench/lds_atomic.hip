// LDS atomic-min throughput (gfx950): can the 41 running minima of the ME kernel live in LDS (ds_min_u32, no return)
// instead of v_min_u32 on the VALU? Each lane owns 41 slots [p][tid]; per "candidate" it issues 41 ds_min_u32 with
// immediate offsets. MODE 0: atomics only; 1: atomics + 100 independent VALU ops per candidate (overlap test);
// 2: the VALU ops only; 3: v_min_u32 in registers (the current scheme) + the same VALU ops.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned a, int iters)
{
  __shared__ unsigned best[41 * 256];
  const int tid = threadIdx.x;
  for (int p = 0; p < 41; p++) best[p * 256 + tid] = 0xffffffffu;
  __syncthreads();
  unsigned v[16], reg[41];
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = tid * 17 + i;
#pragma unroll
  for (int p = 0; p < 41; p++) reg[p] = 0xffffffffu;
  for (int it = 0; it < iters; it++) {
    if (MODE == 1 || MODE == 2 || MODE == 3) {
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = __builtin_amdgcn_sad_u8(v[i], a, v[(i + 1) & 15]);
    }
    if (MODE == 0 || MODE == 1) {
#pragma unroll
      for (int p = 0; p < 41; p++) {
        const unsigned key = v[p & 15] + (unsigned)(p * 77 + it);
        __hip_atomic_fetch_min(&best[p * 256 + tid], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    if (MODE == 3) {
#pragma unroll
      for (int p = 0; p < 41; p++) { const unsigned key = v[p & 15] + (unsigned)(p * 77 + it); reg[p] = min(reg[p], key); }
    }
  }
  __syncthreads();
  unsigned s = 0;
  for (int p = 0; p < 41; p++) s += best[p * 256 + tid] + reg[p];
#pragma unroll
  for (int i = 0; i < 16; i++) s += v[i];
  out[blockIdx.x * 256 + tid] = s;
}
template <int MODE> void run(const char *name, unsigned *out)
{
  const int iters = 2000;
  for (int bpc = 1; bpc <= 2; bpc++) {           // blocks per CU (1 or 2 waves per SIMD)
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
      (void)hipEventRecord(e0);
      k<MODE><<<256 * bpc, 256>>>(out, 3, iters);
      (void)hipEventRecord(e1); (void)hipDeviceSynchronize(); (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-34s blocks/CU=%d  wall=%.3f ms  => %.1f ns per candidate-iteration per wave-slot (per SIMD: %.1f ns)\n", name, bpc, ms, ms * 1e6 / iters, ms * 1e6 / iters / bpc);
  }
}
int main()
{
  unsigned *out; (void)hipMalloc(&out, 1 << 24);
  run<0>("41 ds_min_u32 only", out);
  run<2>("96 v_sad_u8 only", out);
  run<1>("96 v_sad_u8 + 41 add + 41 ds_min", out);
  run<3>("96 v_sad_u8 + 41 add + 41 v_min", out);
  return 0;
}
