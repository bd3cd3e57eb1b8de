// Probe: checks the bf16 MFMA operand maps, the swizzled LDS image and the transposed reads of
// mile_amd/csrc/mile_bf16_frag.h with exact integer data.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bf16_probe tools/bf16_probe.hip && ./tools/bf16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../mile_amd/csrc/mile_bf16_frag.h"

__device__ void fill_image(char *img, const float *src, int rows, int tid, int nt) {
  for (int i = tid; i < rows * 128; i += nt) {
    const int row = i / 128, col = i % 128;
    *reinterpret_cast<bf16 *>(img + img_off(row, col >> 3) + 2 * (col & 7)) = (bf16)src[i];
  }
}

__device__ void dump_tile(float *out, const f32x16 &acc, int lane) {   // out[m][n]
  const int r = lane & 31, h = lane >> 5;
  for (int j = 0; j < 16; ++j) out[acc_m(j, h) * 32 + r] = acc[j];
}

__global__ __launch_bounds__(64) void probe(const float *W, const float *P, const float *Q, float *out, float *img_dump,
                                           int cb, int ib) {
  extern __shared__ char lds[];
  char *Wimg = lds, *Pimg = lds + 32768, *Qimg = Pimg + 8192, *Timg = Qimg + 8192;
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  fill_image(Wimg, W, 128, lane, 64);
  fill_image(Pimg, P, 32, lane, 64);
  fill_image(Qimg, Q, 32, lane, 64);
  for (int i = lane; i < 8192 / 4; i += 64) reinterpret_cast<float *>(Timg)[i] = 0.0f;
  __syncthreads();
  f32x16 acc;
  // A: forward  D[m][n] = sum_in W[in][cb+m] P[n][in]
  for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
  for (int s = 0; s < 8; ++s) acc = mfma_bf16(tr_frag(Wimg, 16 * s, cb, lane), row_frag(Pimg, r, 2 * s + h), acc);
  dump_tile(out, acc, lane);
  // B: backward D[m][n] = sum_out W[cb+m][out] Q[n][out]
  for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
  for (int s = 0; s < 8; ++s) acc = mfma_bf16(row_frag(Wimg, cb + r, 2 * s + h), row_frag(Qimg, r, 2 * s + h), acc);
  dump_tile(out + 1024, acc, lane);
  // C: weight gradient D[m][n] = sum_row P[row][32 ib + m] Q[row][cb + n]
  for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
  for (int s = 0; s < 2; ++s) acc = mfma_bf16(tr_frag(Pimg, 16 * s, 32 * ib, lane), tr_frag(Qimg, 16 * s, cb, lane), acc);
  dump_tile(out + 2048, acc, lane);
  // D: store_tile round trip: tile value = m + 32 * (n & 3) written at columns cb.., read back raw
  for (int j = 0; j < 16; ++j) acc[j] = (float)(acc_m(j, h) + 32 * (r & 3));
  store_tile(Timg, cb, acc, lane);
  __syncthreads();
  for (int i = lane; i < 32 * 128; i += 64) {
    const int row = i / 128, col = i % 128;
    img_dump[i] = (float)*reinterpret_cast<bf16 *>(Timg + img_off(row, col >> 3) + 2 * (col & 7));
  }
}

int main() {
  std::vector<float> W(128 * 128), P(32 * 128), Q(32 * 128), out(3 * 1024), img(32 * 128);
  srand(1);
  for (auto &v : W) v = (float)(rand() % 7 - 3);
  for (auto &v : P) v = (float)(rand() % 7 - 3);
  for (auto &v : Q) v = (float)(rand() % 7 - 3);
  float *dW, *dP, *dQ, *dO, *dI;
  hipMalloc(&dW, W.size() * 4); hipMalloc(&dP, P.size() * 4); hipMalloc(&dQ, Q.size() * 4);
  hipMalloc(&dO, out.size() * 4); hipMalloc(&dI, img.size() * 4);
  hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dP, P.data(), P.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dQ, Q.data(), Q.size() * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  int fails = 0;
  for (int cb = 0; cb < 128; cb += 32)
    for (int ib = 0; ib < 4; ib += 3) {
      probe<<<1, 64, 32768 + 3 * 8192>>>(dW, dP, dQ, dO, dI, cb, ib);
      if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
      hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost);
      hipMemcpy(img.data(), dI, img.size() * 4, hipMemcpyDeviceToHost);
      int bad[4] = {0, 0, 0, 0};
      for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
          float a = 0, b = 0, c = 0;
          for (int k = 0; k < 128; ++k) a += W[k * 128 + cb + m] * P[n * 128 + k];
          for (int k = 0; k < 128; ++k) b += W[(cb + m) * 128 + k] * Q[n * 128 + k];
          for (int k = 0; k < 32; ++k) c += P[k * 128 + 32 * ib + m] * Q[k * 128 + cb + n];
          bad[0] += out[m * 32 + n] != a;
          bad[1] += out[1024 + m * 32 + n] != b;
          bad[2] += out[2048 + m * 32 + n] != c;
        }
      for (int row = 0; row < 32; ++row)
        for (int col = 0; col < 128; ++col) {
          const float want = (col >= cb && col < cb + 32) ? (float)((col - cb) + 32 * (row & 3)) : 0.0f;
          bad[3] += img[row * 128 + col] != want;
        }
      printf("cb=%3d ib=%d  fwd(tr W, row P): %s  bwd(row W, row Q): %s  dW(tr P, tr Q): %s  store_tile: %s\n", cb, ib,
             bad[0] ? "FAIL" : "ok", bad[1] ? "FAIL" : "ok", bad[2] ? "FAIL" : "ok", bad[3] ? "FAIL" : "ok");
      fails += bad[0] + bad[1] + bad[2] + bad[3];
    }
  printf(fails ? "PROBE FAILED (%d mismatches)\n" : "PROBE PASSED\n", fails);
  return fails != 0;
}
