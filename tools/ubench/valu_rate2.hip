// Issue-rate microbenchmark, round 2: candidate replacements for the integer key ops (gfx950). Inline asm so the measured
// instruction is exactly the named one. One block of 256*wps threads per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void k(unsigned *out, unsigned a, unsigned b, int iters)
{
  unsigned v[16];
  unsigned long long w[8];
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = threadIdx.x * 17 + i;
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = ((unsigned long long)(threadIdx.x + i) << 32) | (unsigned)(i * 3 + 1);
  unsigned long long x = ((unsigned long long)a << 32) | b;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (OP == 0) asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
        else if (OP == 1) { if (i < 8) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(w[i]) : "v"(x), "v"(x)); else asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(w[i - 8]) : "v"(x), "v"(x)); }
        else if (OP == 2) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[i & 7]) : "v"(x)); }
        else if (OP == 3) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(v[i]));
        else if (OP == 4) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(v[i]) : "v"(a));
        else if (OP == 5) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(v[i]) : "v"(a));
        else if (OP == 6) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a), "v"(b));
        else if (OP == 7) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
        else if (OP == 8) asm volatile("v_min_u32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
        else if (OP == 9) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
        else if (OP == 10) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(v[i]) : "v"(a));
        else if (OP == 11) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a), "v"(b));
        else if (OP == 12) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
        else if (OP == 13) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(v[i]) : "v"(a));
        else if (OP == 14) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a), "v"(b));
        else if (OP == 15) asm volatile("v_min_u16 %0, %0, %1" : "+v"(v[i]) : "v"(a));
        else if (OP == 16) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
        else if (OP == 17) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
        else if (OP == 18) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a), "v"(b));
        else if (OP == 19) asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(w[i & 7]) : "v"(x), "v"(b));
        else if (OP == 20) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
        else if (OP == 21) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a), "v"(b));
        else if (OP == 22) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(v[i]) : "v"(a));
        else if (OP == 23) asm volatile("v_mqsad_u32_u8 %0, %1, %2, %0" : "+v"(*(__uint128_t *)&w[(i & 3) * 2]) : "v"(x), "v"(b));
      }
    }
  }
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += v[i];
#pragma unroll
  for (int i = 0; i < 8; i++) s += (unsigned)w[i] + (unsigned)(w[i] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name, unsigned *out)
{
  const int iters = 2000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    dim3 grid(256), block(256 * wps);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      k<OP><<<grid, block>>>(out, 3, 5, iters);
      hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1);
    }
    const double ninstr = (double)iters * 64;
    const double waves = (double)grid.x * block.x / 64;
    printf("%-18s waves/SIMD=%d  wall=%.3f ms  => %.2f ns per wave-instr per SIMD\n", name, wps, ms, (ms * 1e-3) / (waves * ninstr / 1024) * 1e9);
  }
}
int main()
{
  unsigned *out; hipMalloc(&out, 1 << 24);
  run<8>("v_min_u32", out); run<12>("v_add3_u32", out); run<20>("v_add_u32", out); run<9>("v_min3_u32", out);
  run<0>("v_min_f32", out); run<16>("v_min3_f32", out); run<17>("v_add_f32", out); run<11>("v_fma_f32", out);
  run<1>("v_pk_fma_f32", out); run<2>("v_pk_add_f32", out); run<3>("v_cvt_f32_u32", out);
  run<4>("v_pk_add_u16", out); run<5>("v_pk_min_u16", out); run<13>("v_pk_sub_i16", out); run<10>("v_pk_max_i16", out); run<15>("v_min_u16", out);
  run<21>("v_pk_fma_f16", out); run<22>("v_pk_min_f16", out);
  run<6>("v_sad_u16", out); run<18>("v_sad_u8", out); run<19>("v_qsad_pk_u16_u8", out); run<23>("v_mqsad_u32_u8", out);
  run<7>("v_perm_b32", out); run<14>("v_dot4_u32_u8", out);
  return 0;
}
