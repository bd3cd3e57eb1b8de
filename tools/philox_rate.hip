// Microbenchmark: cost of Philox4x32-10 written with mul_hi+mul_lo vs a 64-bit product (v_mad_u64_u32).
// hipcc -O3 --offload-arch=gfx950 -o philox_rate tools/philox_rate.hip && ./philox_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int V>
__device__ __forceinline__ void round1(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t &k0, uint32_t &k1) {
  uint32_t hi0, lo0, hi1, lo1;
  if (V == 0) {
    hi0 = __umulhi(0xD2511F53u, c0); lo0 = 0xD2511F53u * c0;
    hi1 = __umulhi(0xCD9E8D57u, c2); lo1 = 0xCD9E8D57u * c2;
  } else {
    uint64_t p0, p1;
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p0) : "v"(c0), "s"(0xD2511F53u) : "vcc");
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p1) : "v"(c2), "s"(0xCD9E8D57u) : "vcc");
    hi0 = (uint32_t)(p0 >> 32); lo0 = (uint32_t)p0; hi1 = (uint32_t)(p1 >> 32); lo1 = (uint32_t)p1;
  }
  const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
  c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
  k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
}
template <int V>
__global__ __launch_bounds__(1024) void k(uint32_t *o, int iters, uint32_t seed) {
  uint32_t c0 = threadIdx.x, c1 = blockIdx.x, c2 = seed, c3 = 7, acc = 0;
  for (int it = 0; it < iters; ++it) {
    uint32_t a0 = c0 + it, a1 = c1, a2 = c2, a3 = c3, k0 = seed, k1 = 1;
#pragma unroll
    for (int r = 0; r < 10; ++r) round1<V>(a0, a1, a2, a3, k0, k1);
    acc ^= a0 ^ a1 ^ a2 ^ a3;
  }
  o[blockIdx.x * 1024 + threadIdx.x] = acc;
}
int main() {
  uint32_t *o; hipMalloc(&o, 256 * 1024 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  uint32_t h[2][4];
  for (int v = 0; v < 2; ++v) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      if (v == 0) k<0><<<256, 1024>>>(o, 1000, 42); else k<1><<<256, 1024>>>(o, 1000, 42);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (rep) printf("variant %d: %.3f ms for 1000 Philox/thread, 4 waves/SIMD -> %.1f cycles per Philox per wave @2.1GHz\n", v, ms, ms * 1e-3 * 2.1e9 / 1000 / 4);
    }
    hipMemcpy(h[v], o, 16, hipMemcpyDeviceToHost);
  }
  printf("match: %d\n", h[0][0] == h[1][0] && h[0][1] == h[1][1]);
  return 0;
}
