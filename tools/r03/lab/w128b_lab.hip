// Dev harness (round 3): k_grad_w128b<3,2> alone on the B3 shape with synthetic operands, timed with HIP events.
// Built per experiment with -DMILE_LAB_* hooks (see tools/r03/lab/build.sh); prints ms per launch and a checksum of the slabs.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I include tools/r03/lab/w128b_lab.hip -o ...
#include "../../../mile_amd/csrc/mile_grad_w128b.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
int main(int argc, char **argv) {
  const int E = argc > 1 ? atoi(argv[1]) : 512, N = argc > 2 ? atoi(argv[2]) : 36000, reps = argc > 3 ? atoi(argv[3]) : 10;
  const int F = 9, NH = 3;
  GradParams gp{};
  DevSpec &sp = gp.spec;
  sp.n_layers = NH + 1; sp.in_features = F;
  int widths[4] = {128, 128, 128, 2}, fin = F, off = 0;
  for (int l = 0; l < 4; ++l) { sp.widths[l] = widths[l]; sp.b_off[l] = off; off += widths[l]; sp.w_off[l] = off; off += fin * widths[l]; fin = widths[l]; }
  sp.d = off; sp.max_width = 128;
  const int d = off, dp = (d + 3) & ~3, Npb = (N + 63) & ~63;
  srand(1);
  auto rnd = []() { return (float)rand() / RAND_MAX * 2.0f - 1.0f; };
  std::vector<float> th((size_t)E * d), y(Npb + 64);
  for (auto &v : th) v = 0.1f * rnd();
  for (auto &v : y) v = rnd();
  std::vector<unsigned short> Xb((size_t)(Npb + 64) * 16, 0), Xt((size_t)32 * Npb, 0);
  for (int i = 0; i < N; ++i) for (int f = 0; f < F; ++f) { const unsigned short b = f2bf(rnd()); Xb[(size_t)i * 16 + f] = b; Xt[(size_t)f * Npb + i] = b; }
  float *dth, *dy, *dsl, *dll; void *dXb, *dXt; long long *dbg;
  CK(hipMalloc(&dth, th.size() * 4)); CK(hipMalloc(&dy, y.size() * 4)); CK(hipMalloc(&dXb, Xb.size() * 2)); CK(hipMalloc(&dXt, Xt.size() * 2));
  CK(hipMalloc(&dsl, (size_t)E * dp * 4)); CK(hipMalloc(&dll, E * 4)); CK(hipMalloc(&dbg, 64));
  CK(hipMemcpy(dth, th.data(), th.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, y.data(), y.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dXb, Xb.data(), Xb.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dXt, Xt.data(), Xt.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(dbg, 0, 64));
  gp.theta = dth; gp.Xb = dXb; gp.Xt = dXt; gp.y = dy; gp.slabs = dsl; gp.llpart = dll; gp.N = N; gp.Npad = Npb; gp.S = 1; gp.Npb = Npb; gp.dp = dp; gp.dbg_buf = dbg;
  using LY = W128Layout<3, 2>;
#ifdef MILE_LAB_TIMING
  auto kern = k_grad_w128b<3, 2, true>;
#else
  auto kern = k_grad_w128b<3, 2, false>;
#endif
  CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LY::BYTES));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  kern<<<dim3(1, E), 256, LY::BYTES>>>(gp);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) kern<<<dim3(1, E), 256, LY::BYTES>>>(gp);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<float> sl((size_t)E * dp);
  CK(hipMemcpy(sl.data(), dsl, sl.size() * 4, hipMemcpyDeviceToHost));
  double cs = 0; for (size_t i = 0; i < (size_t)dp * 4 && i < sl.size(); ++i) cs += fabs((double)sl[i]);
  const double W = 9 * 128 + 2 * 128 * 128 + 128 * 2, flop = (double)E * N * (6 * W - 2 * 9 * 128);
  printf("%-28s ms/launch %7.3f  %6.1f TFLOP/s  frac %.3f  checksum %.6e", LAB_NAME, ms / reps, flop / (ms / reps * 1e-3) / 1e12, flop / (ms / reps * 1e-3) / 2.5e15, cs);
#ifdef MILE_LAB_TIMING
  long long t[8]; CK(hipMemcpy(t, dbg, 64, hipMemcpyDeviceToHost));
  printf("  | top %lld F1 %lld F2 %lld F3 %lld head %lld L2 %lld L1 %lld (pairs %lld)", t[6], t[0], t[1], t[2], t[3], t[5], t[4], t[7]);
#endif
  printf("\n");
  return 0;
}
