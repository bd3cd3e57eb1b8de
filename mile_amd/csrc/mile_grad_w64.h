// MFMA grad-log-likelihood kernel for FCNs whose hidden layers are all 64 wide
// (BASELINE configs B1/B2: [F -> 64 -> 64 -> 64 -> 2], ReLU, Gaussian regression head).
//
// CDNA4 design (not a tiling borrowed from a 32-wide-warp GPU):
//   * one workgroup = 4 waves = one wave per SIMD, all for ONE particle; the particle's
//     weights are staged once in LDS (padded [in][68] images, conflict-free for both the
//     forward (lanes along `out`) and the backward (lanes along `in`, ds_read_b128 along
//     `out`) operand reads);
//   * each wave owns whole 32-row blocks of the training set and carries them through the
//     ENTIRE forward and backward pass in registers.  Activations live in the MFMA
//     accumulator layout transposed ("T layout": lane = data row, register = feature):
//     Z^T[out][row] = W^T . H^T, so the 32x32 accumulator tile of one layer is, register
//     for register, the B operand of the next layer's v_mfma_f32_32x32x2_f32 -- no LDS
//     round trip, no shuffles between layers (fp32 MFMA: one VGPR per operand);
//   * only the products that contract over DATA ROWS (dW = H^T dZ, db, dW_in, dW_out)
//     need the other orientation; the wave transposes dZ / H through a private 32x68
//     LDS image (ds_write_b128 in, ds_read_b32 out, both conflict-free);
//   * dW accumulators (2 x 64x64 = 128 VGPRs), the three activation tiles (96 VGPRs)
//     and the working tiles stay resident: ~350 of the 512 VGPRs one wave per SIMD may use;
//   * the skinny products (F x 64, 64 x 2, biases) run on the VALU in the shadow of the
//     64-cycle MFMAs instead of wasting 32x32 tiles on them;
//   * waves reduce their accumulators through LDS once, at the end, and the workgroup
//     writes one coalesced partial-gradient slab per (particle, row split).
//
// Work per 32-row block and wave: 392 v_mfma_f32_32x32x2_f32 (8 + 2*64 forward,
// 4*64 backward) = 1.606 MFLOP of the 1.638 MFLOP the algorithm needs for 32 rows.
#pragma once
#include "mile_device.h"
#include "mile_grad_generic.h"

#define W64_RS 68  // row stride (floats) of every padded LDS image

__device__ __forceinline__ int tfeat(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ int nrow(int s, int h) { return 8 * (s >> 2) + 4 * h + (s & 3); }

__device__ __forceinline__ void wave_lds_sync() {
  // LDS instructions of one wave execute in order; this only stops the COMPILER from
  // moving a lane's reads above other lanes' writes.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// T layout (lane = data row j, reg r of tile kb = feature 32kb + tfeat(r,h)) -> img[row][feature]
__device__ __forceinline__ void write_image(float *img, const f32x16 (&T)[2], int j, int h) {
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v = {T[kb][4 * g], T[kb][4 * g + 1], T[kb][4 * g + 2], T[kb][4 * g + 3]};
      *(f32x4 *)(img + j * W64_RS + 32 * kb + 8 * g + 4 * h) = v;
    }
}

template <int NH, int FQ>
struct W64Layout {
  static constexpr int NW = (NH > 1 ? NH - 1 : 1);
  static constexpr int FP = 8 * FQ;
  static constexpr int WIMG = 0;                                  // [NH-1][64][68]
  static constexpr int W1IMG = WIMG + (NH - 1) * 64 * W64_RS;     // [FP][68]
  static constexpr int BIAS = W1IMG + FP * W64_RS;                // [NH][64]
  static constexpr int WO = BIAS + NH * 64;                       // [2][64]
  static constexpr int BO = WO + 128;                             // [4]
  static constexpr int WAVE0 = BO + 4;
  static constexpr int IMG = 0;                                   // per wave: [32][68]
  static constexpr int XT = IMG + 32 * W64_RS;                    // [32][FP]
  static constexpr int DOUT = XT + 32 * FP;                       // [32][2]
  static constexpr int WAVE_SZ = DOUT + 64;
  static constexpr int MAIN = WAVE0 + 4 * WAVE_SZ;
  static constexpr int RED = NW * 4 * 64 * W64_RS;                // end-of-kernel reduction alias
  static constexpr int TOTAL = (MAIN > RED ? MAIN : RED);
  static constexpr int BYTES = TOTAL * 4;
};

template <int NH, int FQ>
__global__ __launch_bounds__(256, 1) void k_grad_w64(const GradParams p) {
  using LY = W64Layout<NH, FQ>;
  constexpr int FP = LY::FP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const DevSpec &sp = p.spec;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int e = blockIdx.y, s = blockIdx.x;
  const int d = sp.d, F = sp.in_features;
  const float *th = p.theta + (size_t)e * d;
  float *slab = p.slabs + ((size_t)e * p.S + s) * p.dp;

  float *WIMG = lds + LY::WIMG, *W1IMG = lds + LY::W1IMG, *BIAS = lds + LY::BIAS;
  float *WO = lds + LY::WO, *BO = lds + LY::BO;
  float *wv = lds + LY::WAVE0 + wave * LY::WAVE_SZ;
  float *img = wv + LY::IMG, *xt = wv + LY::XT, *dout_l = wv + LY::DOUT;

  // ---- stage this particle's weights in LDS --------------------------------------
  if (!(p.dbg & 2))
  for (int l = 1; l < NH; ++l) {
    const float *W = th + sp.w_off[l];
    for (int idx = tid; idx < 4096; idx += 256) WIMG[(l - 1) * 64 * W64_RS + (idx >> 6) * W64_RS + (idx & 63)] = W[idx];
  }
  {
    const float *W = th + sp.w_off[0];
    for (int idx = tid; idx < FP * 64; idx += 256) {
      const int r = idx >> 6;
      W1IMG[r * W64_RS + (idx & 63)] = r < F ? W[idx] : 0.0f;
    }
    for (int idx = tid; idx < NH * 64; idx += 256) BIAS[idx] = th[sp.b_off[idx >> 6] + (idx & 63)];
    const float *Wo = th + sp.w_off[NH];
    if (tid < 128) WO[(tid & 1) * 64 + (tid >> 1)] = Wo[tid];      // wo[c][f] = Wout[f][c]
    if (tid < 2) BO[tid] = th[sp.b_off[NH] + tid];
  }
  __syncthreads();

  // ---- per-wave accumulators ------------------------------------------------------
  f32x16 dWacc[LY::NW][2][2];   // [hidden->hidden layer][ob][ib]: dW^T[out 32ob+row][in 32ib+col]
#pragma unroll
  for (int l = 0; l < LY::NW; ++l)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) dWacc[l][a][b][r] = 0.0f;
  float bacc[NH][2];
#pragma unroll
  for (int l = 0; l < NH; ++l) bacc[l][0] = bacc[l][1] = 0.0f;
  float w1acc[2][FP];
#pragma unroll
  for (int c = 0; c < FP; ++c) w1acc[0][c] = w1acc[1][c] = 0.0f;
  float woacc[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
  float boacc[2] = {0.0f, 0.0f};
  float llacc = 0.0f;

  const int NB = p.Npad / 32;
  const int b0 = (int)(((long long)s * NB) / p.S);
  const int b1 = (p.dbg & 1) ? b0 : (int)(((long long)(s + 1) * NB) / p.S);

  for (int blk = b0 + wave; blk < b1; blk += 4) {
    const int row0 = blk * 32;
    // ---- X tile: lane (j,h) holds features 8q+4h+(0..3) of row j ------------------
    f32x4 xv[FQ];
#pragma unroll
    for (int q = 0; q < FQ; ++q) {
      xv[q] = *(const f32x4 *)(p.Xp + (size_t)(row0 + j) * FP + 8 * q + 4 * h);
      *(f32x4 *)(xt + j * FP + 8 * q + 4 * h) = xv[q];
    }
    f32x16 H[NH][2];
    // ---- layer 0: Z1^T = W1^T . X^T ----------------------------------------------
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
      f32x16 acc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *(const f32x4 *)(BIAS + 32 * ob + 8 * g + 4 * h);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[4 * g + m] = bv[m];
      }
#pragma unroll
      for (int q = 0; q < FQ; ++q)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float a = W1IMG[(8 * q + 4 * h + m) * W64_RS + 32 * ob + j];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xv[q][m], acc, 0, 0, 0);
        }
#pragma unroll
      for (int r = 0; r < 16; ++r) H[0][ob][r] = fmaxf(acc[r], 0.0f);
    }
    // ---- hidden layers ------------------------------------------------------------
#pragma unroll
    for (int l = 1; l < NH; ++l) {
      const float *Wl = WIMG + (l - 1) * 64 * W64_RS;
#pragma unroll
      for (int ob = 0; ob < 2; ++ob) {
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 bv = *(const f32x4 *)(BIAS + l * 64 + 32 * ob + 8 * g + 4 * h);
#pragma unroll
          for (int m = 0; m < 4; ++m) acc[4 * g + m] = bv[m];
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            const float a = Wl[(32 * kb + tfeat(t, h)) * W64_RS + 32 * ob + j];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, H[l - 1][kb][t], acc, 0, 0, 0);
          }
#pragma unroll
        for (int r = 0; r < 16; ++r) H[l][ob][r] = fmaxf(acc[r], 0.0f);
      }
    }
    // ---- output layer (64 -> 2) on the VALU + Gaussian head ----------------------
    float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 w0 = *(const f32x4 *)(WO + 32 * kb + 8 * g + 4 * h);
        const f32x4 w1 = *(const f32x4 *)(WO + 64 + 32 * kb + 8 * g + 4 * h);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          p0 = fmaf(w0[m], H[NH - 1][kb][4 * g + m], p0);
          p1 = fmaf(w1[m], H[NH - 1][kb][4 * g + m], p1);
        }
      }
    p0 += __shfl_xor(p0, 32);
    p1 += __shfl_xor(p1, 32);
    const float mu = p0 + BO[0], sr = p1 + BO[1];
    float dmu, ds;
    float ll = row_loss_regr(mu, sr, ((const float *)p.y)[row0 + j], dmu, ds);
    if (row0 + j >= p.N) { ll = 0.0f; dmu = 0.0f; ds = 0.0f; }
    if (h == 0) {
      llacc += ll;
      boacc[0] += dmu;
      boacc[1] += ds;
      *(f32x2 *)(dout_l + 2 * j) = f32x2{dmu, ds};
    }
    // ---- dZ of the last hidden layer ----------------------------------------------
    f32x16 dZ[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 w0 = *(const f32x4 *)(WO + 32 * kb + 8 * g + 4 * h);
        const f32x4 w1 = *(const f32x4 *)(WO + 64 + 32 * kb + 8 * g + 4 * h);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float dh = fmaf(w0[m], dmu, w1[m] * ds);
          dZ[kb][4 * g + m] = H[NH - 1][kb][4 * g + m] > 0.0f ? dh : 0.0f;
        }
      }
    // ---- dW_out[f][c] += sum_rows H_last[f][row] dout[row][c]  (row-contracting, VALU)
    wave_lds_sync();
    write_image(img, H[NH - 1], j, h);
    wave_lds_sync();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ss = 0; ss < 16; ++ss) {
        const int n = nrow(ss, h);
        const float v = img[n * W64_RS + 32 * kb + j];
        const f32x2 dd = *(const f32x2 *)(dout_l + 2 * n);
        woacc[kb][0] = fmaf(v, dd[0], woacc[kb][0]);
        woacc[kb][1] = fmaf(v, dd[1], woacc[kb][1]);
      }
    // ---- hidden layers, backward ---------------------------------------------------
#pragma unroll
    for (int l = NH - 1; l >= 1; --l) {
      const float *Wl = WIMG + (l - 1) * 64 * W64_RS;
      wave_lds_sync();
      write_image(img, dZ, j, h);
      wave_lds_sync();
      f32x16 afr[2];
#pragma unroll
      for (int ob = 0; ob < 2; ++ob) {
        float bs = 0.0f;
#pragma unroll
        for (int ss = 0; ss < 16; ++ss) {
          afr[ob][ss] = img[nrow(ss, h) * W64_RS + 32 * ob + j];
          bs += afr[ob][ss];
        }
        bacc[l][ob] += bs;
      }
      wave_lds_sync();
      write_image(img, H[l - 1], j, h);
      wave_lds_sync();
      // dW_l^T[out][in] += dZ_l^T[out][rows] . H_{l-1}[rows][in]
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int ss = 0; ss < 16; ++ss) {
          const float b = img[nrow(ss, h) * W64_RS + 32 * ib + j];
#pragma unroll
          for (int ob = 0; ob < 2; ++ob)
            dWacc[l - 1][ob][ib] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[ob][ss], b, dWacc[l - 1][ob][ib], 0, 0, 0);
        }
      // dH_{l-1}^T[in][row] = W_l[in][out] . dZ_l^T[out][row]
      f32x16 dHn[2];
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 a4 = *(const f32x4 *)(Wl + (32 * ib + j) * W64_RS + 32 * kb + 8 * g + 4 * h);
#pragma unroll
            for (int m = 0; m < 4; ++m)
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m], dZ[kb][4 * g + m], acc, 0, 0, 0);
          }
        dHn[ib] = acc;
      }
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) dZ[ib][r] = H[l - 1][ib][r] > 0.0f ? dHn[ib][r] : 0.0f;
    }
    // ---- first layer: db_0, dW_0[c][out] += sum_rows X[row][c] dZ_0[out][row] (VALU) -
    wave_lds_sync();
    write_image(img, dZ, j, h);
    wave_lds_sync();
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
      float bs = 0.0f;
#pragma unroll
      for (int ss = 0; ss < 16; ++ss) {
        const int n = nrow(ss, h);
        const float v = img[n * W64_RS + 32 * ob + j];
        bs += v;
#pragma unroll
        for (int q4 = 0; q4 < FP / 4; ++q4) {
          const f32x4 xr = *(const f32x4 *)(xt + n * FP + 4 * q4);
#pragma unroll
          for (int m = 0; m < 4; ++m) w1acc[ob][4 * q4 + m] = fmaf(v, xr[m], w1acc[ob][4 * q4 + m]);
        }
      }
      bacc[0][ob] += bs;
    }
  }

  // ---- reduce the four waves' accumulators through LDS, write the slab -------------
  __syncthreads();  // weights and images are dead from here on; LDS is reused
  float *RED = lds;
  if (p.dbg & 4) return;
  // one round for all hidden->hidden matrices: RED[l][wave][in][68]
  // D[o][i]: lane column = in-feature (32ib + j), register r = out-feature 32ob + tfeat(r,h)
#pragma unroll
  for (int l = 0; l < NH - 1; ++l)
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x16 &D = dWacc[l][ob][ib];
          f32x4 v = {D[4 * g], D[4 * g + 1], D[4 * g + 2], D[4 * g + 3]};
          *(f32x4 *)(RED + (l * 4 + wave) * 64 * W64_RS + (32 * ib + j) * W64_RS + 32 * ob + 8 * g + 4 * h) = v;
        }
  __syncthreads();
#pragma unroll
  for (int l = 0; l < NH - 1; ++l) {
    float *out = slab + sp.w_off[l + 1];
    const float *R0 = RED + l * 4 * 64 * W64_RS;
    if ((sp.w_off[l + 1] & 3) == 0) {     // 128-bit path: the slab row is 16-byte aligned (dp % 4 == 0)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int q = tid + 256 * k;        // float4 index: row q >> 4, columns 4 (q & 15) ..
        const int o = (q >> 4) * W64_RS + 4 * (q & 15);
        const f32x4 v = (*(const f32x4 *)(R0 + o) + *(const f32x4 *)(R0 + 64 * W64_RS + o)) +
                        (*(const f32x4 *)(R0 + 2 * 64 * W64_RS + o) + *(const f32x4 *)(R0 + 3 * 64 * W64_RS + o));
        *(f32x4 *)(out + 4 * q) = v;
      }
    } else {
#pragma unroll 4
      for (int idx = tid; idx < 4096; idx += 256) {
        const int o = (idx >> 6) * W64_RS + (idx & 63);
        out[idx] = (R0[o] + R0[64 * W64_RS + o]) + (R0[2 * 64 * W64_RS + o] + R0[3 * 64 * W64_RS + o]);
      }
    }
  }
  __syncthreads();
  // small accumulators: SM[wave][k]
  constexpr int SM_B = 0;                 // biases of hidden layers [NH][64]
  constexpr int SM_W1 = SM_B + NH * 64;   // [FP][64]
  constexpr int SM_WO = SM_W1 + FP * 64;  // [64][2]
  constexpr int SM_BO = SM_WO + 128;      // [2]
  constexpr int SM_LL = SM_BO + 2;
  constexpr int SM_SZ = SM_LL + 2;
  float *SM = RED + wave * SM_SZ;
#pragma unroll
  for (int l = 0; l < NH; ++l)
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
      const float v = bacc[l][ob] + __shfl_xor(bacc[l][ob], 32);
      if (h == 0) SM[SM_B + l * 64 + 32 * ob + j] = v;
    }
#pragma unroll
  for (int ob = 0; ob < 2; ++ob)
#pragma unroll
    for (int c = 0; c < FP; ++c) {
      const float v = w1acc[ob][c] + __shfl_xor(w1acc[ob][c], 32);
      if (h == 0) SM[SM_W1 + c * 64 + 32 * ob + j] = v;
    }
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float v = woacc[kb][c] + __shfl_xor(woacc[kb][c], 32);
      if (h == 0) SM[SM_WO + (32 * kb + j) * 2 + c] = v;
    }
  {
    const float v0 = wave_sum(boacc[0]), v1 = wave_sum(boacc[1]), vl = wave_sum(llacc);
    if (lane == 0) { SM[SM_BO] = v0; SM[SM_BO + 1] = v1; SM[SM_LL] = vl; }
  }
  __syncthreads();
  for (int k = tid; k < SM_LL + 1; k += 256) {
    const float v = (RED[k] + RED[SM_SZ + k]) + (RED[2 * SM_SZ + k] + RED[3 * SM_SZ + k]);
    if (k < SM_W1) slab[sp.b_off[k >> 6] + (k & 63)] = v;
    else if (k < SM_WO) { if ((k - SM_W1) < F * 64) slab[sp.w_off[0] + (k - SM_W1)] = v; }
    else if (k < SM_BO) slab[sp.w_off[NH] + (k - SM_WO)] = v;
    else if (k < SM_LL) slab[sp.b_off[NH] + (k - SM_BO)] = v;
    else p.llpart[(size_t)e * p.S + s] = v;
  }
}
