// Dev harness (round 3): k_grad_w64<3,1,true> alone on the B2 shape with synthetic operands, HIP-event timing, optional
// per-phase stamps (-DMILE_LAB_W64_TIMING) and ablation hooks (-DMILE_LAB_NO_MFMA ...).
#include "../../../mile_amd/csrc/mile_grad_w64.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char **argv) {
  const int E = argc > 1 ? atoi(argv[1]) : 128, N = argc > 2 ? atoi(argv[2]) : 1052, reps = argc > 3 ? atoi(argv[3]) : 50, S = argc > 4 ? atoi(argv[4]) : 2;
  const int F = 5, NH = 3, FP = 8;
  GradParams gp{};
  DevSpec &sp = gp.spec;
  sp.n_layers = NH + 1; sp.in_features = F;
  int widths[4] = {64, 64, 64, 2}, fin = F, off = 0;
  for (int l = 0; l < 4; ++l) { sp.widths[l] = widths[l]; sp.b_off[l] = off; off += widths[l]; sp.w_off[l] = off; off += fin * widths[l]; fin = widths[l]; }
  sp.d = off; sp.max_width = 64;
  const int d = off, dp = (d + 3) & ~3, Npad = (N + 31) & ~31;
  srand(1);
  auto rnd = []() { return (float)rand() / (float)RAND_MAX * 2.0f - 1.0f; };
  std::vector<float> th((size_t)E * d), y(Npad + 64), Xp((size_t)(Npad + 64) * FP, 0.0f);
  for (auto &v : th) v = 0.1f * rnd();
  for (auto &v : y) v = rnd();
  for (int i = 0; i < N; ++i) for (int f = 0; f < F; ++f) Xp[(size_t)i * FP + f] = rnd();
  float *dth, *dy, *dXp, *dsl, *dll; long long *dbg;
  CK(hipMalloc(&dth, th.size() * 4)); CK(hipMalloc(&dy, y.size() * 4)); CK(hipMalloc(&dXp, Xp.size() * 4));
  CK(hipMalloc(&dsl, (size_t)E * S * dp * 4)); CK(hipMalloc(&dll, E * S * 4)); CK(hipMalloc(&dbg, 4096));
  CK(hipMemcpy(dth, th.data(), th.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, y.data(), y.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dXp, Xp.data(), Xp.size() * 4, hipMemcpyHostToDevice)); CK(hipMemset(dbg, 0, 4096));
  gp.theta = dth; gp.Xp = dXp; gp.y = dy; gp.slabs = dsl; gp.llpart = dll; gp.N = N; gp.Npad = Npad; gp.Fp = FP; gp.S = S; gp.dp = dp; gp.dbg_buf = dbg; gp.dbg = 0;
  using LY = W64Layout<3, 1, true>;
  auto kern = k_grad_w64<3, 1, true>;
  CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LY::BYTES));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  kern<<<dim3(S, E), 256, LY::BYTES>>>(gp, W64NoFuse{0});
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) kern<<<dim3(S, E), 256, LY::BYTES>>>(gp, W64NoFuse{0});
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<float> sl((size_t)dp);
  CK(hipMemcpy(sl.data(), dsl, sl.size() * 4, hipMemcpyDeviceToHost));
  double cs = 0; for (size_t i = 0; i < sl.size(); ++i) cs += fabs((double)sl[i]);
  printf("%-24s E=%d N=%d S=%d lds=%d us/launch %7.2f  checksum %.6e", LAB_NAME, E, N, S, LY::BYTES, ms / reps * 1e3, cs);
#ifdef MILE_LAB_W64_TIMING
  long long t[11]; CK(hipMemcpy(t, dbg, 88, hipMemcpyDeviceToHost));
#ifdef MILE_LAB_W64_TOTALS
  const double nb = 1;                                 // totals over the workgroup's blocks (difference of two workgroups = the shared block)
#else
  const double nb = t[10] > 0 ? (double)t[10] : 1;   // the buffer holds the last launch only
#endif
  printf("\n   cycles per block (wave 0 of WG 0, %lld full rounds/launch): top %.0f L0 %.0f F1 %.0f F2 %.0f head %.0f dH2 %.0f dW2 %.0f dH1 %.0f dW1 %.0f first %.0f",
         t[10], t[9] / nb, t[0] / nb, t[1] / nb, t[2] / nb, t[3] / nb, t[6] / nb, t[7] / nb, t[4] / nb, t[5] / nb, t[8] / nb);
#endif
  printf("\n");
  return 0;
}
