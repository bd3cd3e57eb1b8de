// Microbenchmark: cycles per v_mfma_f32_32x32x2_f32 at one wave per SIMD, for (a) one dependent
// chain, (b) four independent accumulators, (c) chain with one ds_read_b32 A operand per MFMA,
// (d) chain + 6 independent VALU per MFMA.  Dev tool.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
#define SB() __builtin_amdgcn_sched_barrier(0)
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k(float *out, int iters, long long *cyc) {
  __shared__ float lds[8192];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += NT) lds[i] = 1e-3f * (i & 63);
  __syncthreads();
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = 1e-3f * tid, y = 2e-3f * tid, v0 = x, v1 = y, v2 = x, v3 = y, v4 = x, v5 = y;
  bf16x8 bx, by;
  for (int q = 0; q < 8; ++q) { bx[q] = (short)(0x3f80 + ((tid + q) & 15)); by[q] = (short)(0x3f00 + ((tid * 3 + q) & 31)); }
  int off = tid & 63;
  asm volatile("" : "+v"(off));
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0) { a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); }
      if (MODE == 1) {
        if ((u & 3) == 0) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        if ((u & 3) == 1) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        if ((u & 3) == 2) a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        if ((u & 3) == 3) a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
      }
      if (MODE == 2) { const float a = lds[off + 64 * u + (it & 7) * 1024]; a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, y, a0, 0, 0, 0); }
      if (MODE == 3) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f);
        v3 = fmaf(v3, 1.0001f, 0.5f); v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f);
        SB();
      }
      if (MODE == 5) { a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a0, 0, 0, 0); }
      if (MODE == 6) {   // bf16 chain + 6 VALU
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a0, 0, 0, 0);
        v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f);
        v3 = fmaf(v3, 1.0001f, 0.5f); v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f);
        SB();
      }
      if (MODE == 7) {   // bf16 chain interleaved 2:1 with an independent fp32 MFMA chain
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a0, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a2, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        SB();
      }
      if (MODE == 8) {   // bf16 chain + 12 VALU + LDS
        const float a = lds[off + 64 * u + (it & 7) * 1024];
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a0, 0, 0, 0);
        v0 = fmaf(v0, 1.0001f, a); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f);
        v3 = fmaf(v3, 1.0001f, 0.5f); v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f);
        v0 = fmaf(v0, 0.9999f, 0.5f); v1 = fmaf(v1, 0.9999f, 0.5f); v2 = fmaf(v2, 0.9999f, 0.5f);
        v3 = fmaf(v3, 0.9999f, 0.5f); v4 = fmaf(v4, 0.9999f, 0.5f); v5 = fmaf(v5, 0.9999f, 0.5f);
        SB();
      }
      if (MODE == 4) {   // chain + 12 VALU + 1 LDS read per MFMA
        const float a = lds[off + 64 * u + (it & 7) * 1024];
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        v0 = fmaf(v0, 1.0001f, a); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f);
        v3 = fmaf(v3, 1.0001f, 0.5f); v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f);
        v0 = fmaf(v0, 0.9999f, 0.5f); v1 = fmaf(v1, 0.9999f, 0.5f); v2 = fmaf(v2, 0.9999f, 0.5f);
        v3 = fmaf(v3, 0.9999f, 0.5f); v4 = fmaf(v4, 0.9999f, 0.5f); v5 = fmaf(v5, 0.9999f, 0.5f);
        SB();
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * NT + tid] = s + v0 + v1 + v2 + v3 + v4 + v5;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE, int NT> void run(const char *name) {
  float *out; long long *cyc; hipMalloc(&out, 256 * NT * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, NT><<<256, NT>>>(out, 100, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE, NT><<<256, NT>>>(out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double n = 16.0 * iters;
  printf("%-34s wall %.3f ms  -> %.1f ns/MFMA = %.1f cyc@2.4GHz ; s_memtime %.1f ticks/MFMA ; %.1f TFLOP/s chip\n", name, ms,
         ms * 1e6 / n, ms * 1e6 / n * 2.4, (double)h[0] / n, (NT / 64) * 256.0 * 4096.0 * n / (ms * 1e-3) / 1e12);
}
int main() {
  run<0, 256>("1w/SIMD dependent chain"); run<3, 256>("1w/SIMD chain + 6 VALU/MFMA"); run<4, 256>("1w/SIMD chain + 12 VALU + LDS");
  run<0, 512>("2w/SIMD dependent chain"); run<3, 512>("2w/SIMD chain + 6 VALU/MFMA"); run<4, 512>("2w/SIMD chain + 12 VALU + LDS");
  run<5, 256>("1w/SIMD bf16 chain"); run<6, 256>("1w/SIMD bf16 chain + 6 VALU"); run<8, 256>("1w/SIMD bf16 chain + 12 VALU + LDS");
  run<7, 256>("1w/SIMD 2 bf16 + 1 fp32 MFMA");
  run<5, 512>("2w/SIMD bf16 chain"); run<6, 512>("2w/SIMD bf16 chain + 6 VALU"); run<8, 512>("2w/SIMD bf16 chain + 12 VALU + LDS");
  run<7, 512>("2w/SIMD 2 bf16 + 1 fp32 MFMA");
  return 0;
}
