// Dev harness (round 3): the three k_mm3 forms of B4's layer-wise path alone (K = N = 256 layer, M data rows per chunk,
// batch = particles), HIP-event timing.  -DMILE_LAB_MM_NO_SPLIT / -DMILE_LAB_NO_MFMA price the on-the-fly split / the products.
#include "../../../mile_amd/csrc/mile_mm3.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <typename K> static float run(K kern, const MMParams &p, dim3 grid, int lds, int reps) {
  CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  kern<<<grid, 256, lds>>>(p); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) kern<<<grid, 256, lds>>>(p);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}
int main(int argc, char **argv) {
  const int E = argc > 1 ? atoi(argv[1]) : 128, M = argc > 2 ? atoi(argv[2]) : 16384, reps = argc > 3 ? atoi(argv[3]) : 3;
  const int W = 256;
  const size_t act = (size_t)E * M * W;
  float *H0, *H1, *dZ, *dW, *bias, *cs; bf16 *Wt;
  CK(hipMalloc(&H0, act * 4)); CK(hipMalloc(&H1, act * 4)); CK(hipMalloc(&dZ, act * 4));
  CK(hipMalloc(&dW, (size_t)E * W * W * 4)); CK(hipMalloc(&bias, (size_t)E * W * 4)); CK(hipMalloc(&cs, (size_t)E * W * 4));
  CK(hipMalloc(&Wt, (size_t)E * 3 * W * W * 2));
  {   // small finite values everywhere
    std::vector<float> h(1 << 20);
    srand(1);
    for (auto &v : h) v = (float)rand() / (float)RAND_MAX - 0.5f;
    for (size_t o = 0; o < act; o += h.size()) {
      const size_t n = std::min(h.size(), act - o);
      CK(hipMemcpy(H0 + o, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dZ + o, h.data(), n * 4, hipMemcpyHostToDevice));
    }
    CK(hipMemset(bias, 0, (size_t)E * W * 4)); CK(hipMemset(dW, 0, (size_t)E * W * W * 4));
    std::vector<unsigned short> wb((size_t)3 * W * W);
    for (auto &v : wb) v = 0x3c00 + (rand() & 0xff);
    for (int e = 0; e < E; ++e) CK(hipMemcpy(Wt + (size_t)e * 3 * W * W, wb.data(), wb.size() * 2, hipMemcpyHostToDevice));
  }
  const double flop = 2.0 * E * M * W * W;
  MMParams p{};
  p.M = M; p.N = W; p.K = W; p.c_vec = 1; p.xcd_remap = 1;
  // forward: H1 = relu(H0 W + b)
  p.A = H0; p.sA = (long long)M * W; p.lda = W; p.B = Wt; p.sB = 3LL * W * W; p.ldb = W; p.tB = (long long)W * W;
  p.C = H1; p.sC = (long long)M * W; p.ldc = W; p.bias = bias; p.sBias = W; p.act = MILE_ACT_RELU; p.apply_act = 1;
  using LF = MMLayout<MM_A_MK, MM_B_T3_KN, 3, 32>;
  float t = run(k_mm3<MM_A_MK, MM_B_T3_KN, MM_EPI_BIAS_ACT, 3, 32, MILE_ACT_RELU, false, false, true>, p, dim3(2, M / 128, E), LF::BYTES, reps);
  printf("%-14s forward  M=%d E=%d: %8.3f ms  %6.1f TFLOP/s (fp32-equivalent)  mix-bound frac %.3f\n", LAB_NAME, M, E, t, flop / t / 1e9, flop / t / 1e9 / 419.5);
  // dH: dZ_prev = (dZ W^T) * relu'(H0)
  p.A = dZ; p.C = H1; p.Hprev = H0; p.sH = (long long)M * W; p.ldh = W; p.bias = nullptr;
  using LD = MMLayout<MM_A_MK, MM_B_T3_NK, 3, 32>;
  t = run(k_mm3<MM_A_MK, MM_B_T3_NK, MM_EPI_ACT_GRAD, 3, 32, MILE_ACT_RELU, false, false, true>, p, dim3(2, M / 128, E), LD::BYTES, reps);
  printf("%-14s dH       M=%d E=%d: %8.3f ms  %6.1f TFLOP/s\n", LAB_NAME, M, E, t, flop / t / 1e9);
  // dW = H0^T dZ  (K = data rows), column sums of dZ
  MMParams q{};
  q.M = W; q.N = W; q.K = M; q.c_vec = 1; q.xcd_remap = 2;
  q.A = H0; q.sA = (long long)M * W; q.lda = W; q.B = dZ; q.sB = (long long)M * W; q.ldb = W;
  q.C = dW; q.sC = (long long)W * W; q.ldc = W; q.colsum = cs; q.sColsum = W; q.accumulate = 0; q.act = -1;
  using LW = MMLayout<MM_A_KM, MM_B_F32_KN, 3, 32>;
  t = run(k_mm3<MM_A_KM, MM_B_F32_KN, MM_EPI_STORE, 3, 32, -1, false, true, true>, q, dim3(2, 2, E), LW::BYTES, reps);
  printf("%-14s dW       M=%d E=%d: %8.3f ms  %6.1f TFLOP/s\n", LAB_NAME, M, E, t, flop / t / 1e9);
  return 0;
}
