// k_grad_w128b<NH>: full-batch log-likelihood gradient of a ReLU regression FCN with NH (1..3)
// hidden layers of width 128 (config B3: [9 -> 128 -> 128 -> 128 -> 2]), bf16 MFMA operands with
// fp32 accumulation (v_mfma_f32_32x32x16_bf16).  Parameters stay fp32 in HBM; weights, inputs,
// activations and back-propagated signals are rounded to bf16 where they enter a matrix product --
// the mixed-precision recipe BASELINE config 3 ("bf16") names.  Selected explicitly
// (MILE_GRAD_MFMA_W128_BF16); MILE_GRAD_AUTO never picks a reduced-precision kernel.
//
// Follows the same maths as k_grad_generic (src/flax_building_blocks/basic.py:42-61 Dense stack,
// src/training/probabilistic.py:92-100 Gaussian head with nansum).
//
// Work split: one workgroup = 4 waves = one particle x one row range.  All four waves walk the
// same 32-row tile; wave w owns feature block w (32 of the 128 features) of every hidden layer:
// its slice of the activations, of dZ and -- the reason for the split -- its 128 x 32 column slice
// of every weight gradient, which lives in accumulator registers for the whole kernel (a full
// 128 x 128 fp32 gradient per layer does not fit one wave).  Tiles are exchanged through swizzled
// [32][128] bf16 LDS images (mile_bf16_frag.h) that serve both the row reads (forward / dH) and the
// transposed reads (dW contracts over rows) without a second copy; the bf16 weight images serve
// forward (transposed read) and backward (row read) the same way.  2*NH+1 barriers per tile.
#pragma once
#include "mile_bf16_frag.h"
#include "mile_device.h"
#include "mile_grad_generic.h"

template <int NH>
struct W128Layout {
  static constexpr int WIMG = 0;                              // W_2..W_NH: [128 in][128 out] bf16, 32 KiB each
  static constexpr int W1IMG = WIMG + (NH - 1) * 32768;       // W_1: [16 in (zero padded)][128 out]
  static constexpr int WOT = W1IMG + 4096;                    // head weights transposed: [32 (k, zero padded)][128 in]
  static constexpr int HIMG = WOT + 8192;                     // H_1..H_NH: [32 rows][128]
  static constexpr int DZ = HIMG + NH * 8192;                 // dZ ping-pong
  static constexpr int DO = DZ + 2 * 8192;                    // d(out): [32 rows][first 32 columns used]
  static constexpr int BYTES = DO + 8192;
};

template <int NH>
__global__ __launch_bounds__(256) void k_grad_w128b(const GradParams p) {
  using LY = W128Layout<NH>;
  extern __shared__ __attribute__((aligned(16))) char lds128[];
  char *lds = lds128;
  const DevSpec &sp = p.spec;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const int e = blockIdx.y, sidx = blockIdx.x;
  const int d = sp.d, F = sp.in_features;
  const float *th = p.theta + (size_t)e * d;
  float *slab = p.slabs + ((size_t)e * p.S + sidx) * p.dp;
  const bf16 *Xb = reinterpret_cast<const bf16 *>(p.Xb);
  const bf16 *Xt = reinterpret_cast<const bf16 *>(p.Xt);
  const float *yv = reinterpret_cast<const float *>(p.y);

  // ---- stage the weights as bf16 images -----------------------------------------------------
#pragma unroll
  for (int li = 1; li < NH; ++li) {
    char *img = lds + LY::WIMG + (li - 1) * 32768;
    const float *W = th + sp.w_off[li];
    for (int c = tid; c < 128 * 16; c += 256) {
      const int row = c >> 4, ch = c & 15;
      const float *src = W + row * 128 + ch * 8;
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (bf16)src[j];
      *reinterpret_cast<bf16x8 *>(img + img_off(row, ch)) = v;
    }
  }
  {
    const float *W1 = th + sp.w_off[0];
    for (int c = tid; c < 16 * 16; c += 256) {
      const int row = c >> 4, ch = c & 15;
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (bf16)(row < F ? W1[row * 128 + ch * 8 + j] : 0.0f);
      *reinterpret_cast<bf16x8 *>(lds + LY::W1IMG + img_off(row, ch)) = v;
    }
    const float *Wo = th + sp.w_off[NH];
    for (int c = tid; c < 32 * 16; c += 256) {
      const int row = c >> 4, ch = c & 15;   // row = head output k
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (bf16)(row < 2 ? Wo[(ch * 8 + j) * 2 + row] : 0.0f);
      *reinterpret_cast<bf16x8 *>(lds + LY::WOT + img_off(row, ch)) = v;
    }
  }
  float bias[NH][16];
#pragma unroll
  for (int l = 0; l < NH; ++l)
#pragma unroll
    for (int j = 0; j < 16; ++j) bias[l][j] = th[sp.b_off[l] + 32 * w + acc_m(j, h)];
  const float bo0 = th[sp.b_off[NH]], bo1 = th[sp.b_off[NH] + 1];
  __syncthreads();

  f32x16 dW[NH > 1 ? NH - 1 : 1][4], dW1, dWo, db[NH], dbo;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    dW1[j] = 0.0f; dWo[j] = 0.0f; dbo[j] = 0.0f;
#pragma unroll
    for (int l = 0; l < NH; ++l) db[l][j] = 0.0f;
#pragma unroll
    for (int l = 0; l < NH - 1; ++l)
#pragma unroll
      for (int ib = 0; ib < 4; ++ib) dW[l][ib][j] = 0.0f;
  }
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (bf16)1.0f;
  float ll_acc = 0.0f;

  const int NB = p.Npad / 32;
  const int nb0 = (int)((long long)sidx * NB / p.S), nb1 = (int)((long long)(sidx + 1) * NB / p.S);
  char *const DOimg = lds + LY::DO;

  for (int t = nb0; t < nb1; ++t) {
    const int row0 = 32 * t;
    uint32_t mask[NH];
    f32x16 acc;
    // ---- forward ---------------------------------------------------------------------------
#pragma unroll
    for (int l = 0; l < NH; ++l) {
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
      if (l == 0) {
        const bf16x8 xb = *reinterpret_cast<const bf16x8 *>(Xb + (size_t)(row0 + r) * 16 + 8 * h);
        acc = mfma_bf16(tr_frag(lds + LY::W1IMG, 0, 32 * w, lane), xb, acc);
      } else {
        const char *Wimg = lds + LY::WIMG + (l - 1) * 32768, *Hin = lds + LY::HIMG + (l - 1) * 8192;
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = mfma_bf16(tr_frag(Wimg, 16 * s, 32 * w, lane), row_frag(Hin, r, 2 * s + h), acc);
      }
      uint32_t mk = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float z = acc[j] + bias[l][j];
        mk |= (z > 0.0f ? 1u : 0u) << j;
        acc[j] = relu1(z);
      }
      mask[l] = mk;
      store_tile(lds + LY::HIMG + l * 8192, 32 * w, acc, lane);
      __syncthreads();
    }
    // ---- head (every wave computes it; wave 0 publishes d(out)) ---------------------------------
    {
      const char *Hin = lds + LY::HIMG + (NH - 1) * 8192;
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
#pragma unroll
      for (int s = 0; s < 8; ++s) acc = mfma_bf16(row_frag(lds + LY::WOT, r, 2 * s + h), row_frag(Hin, r, 2 * s + h), acc);
      float dmu = 0.0f, dsg = 0.0f;
      if (h == 0 && row0 + r < p.N) {
        const float ll = row_loss_regr(acc[0] + bo0, acc[1] + bo1, yv[row0 + r], dmu, dsg);
        ll_acc += ll;
      }
      if (w == 0) {
        bf16x8 c0, z8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { c0[j] = (bf16)0.0f; z8[j] = (bf16)0.0f; }
        c0[0] = (bf16)dmu; c0[1] = (bf16)dsg;
        *reinterpret_cast<bf16x8 *>(DOimg + img_off(r, 2 * h)) = h == 0 ? c0 : z8;
        *reinterpret_cast<bf16x8 *>(DOimg + img_off(r, 2 * h + 1)) = z8;
      }
      __syncthreads();
    }
    // ---- backward: head ---------------------------------------------------------------------------
    int pp = 0;
    {
      const char *Hin = lds + LY::HIMG + (NH - 1) * 8192;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 bq = tr_frag(DOimg, 16 * s, 0, lane);
        dWo = mfma_bf16(tr_frag(Hin, 16 * s, 32 * w, lane), bq, dWo);
        dbo = mfma_bf16(ones, bq, dbo);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
      acc = mfma_bf16(tr_frag(lds + LY::WOT, 0, 32 * w, lane), row_frag(DOimg, r, h), acc);
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = (mask[NH - 1] >> j) & 1u ? acc[j] : 0.0f;
      store_tile(lds + LY::DZ, 32 * w, acc, lane);
      __syncthreads();
    }
    // ---- backward: hidden layers NH .. 2 ------------------------------------------------------------
#pragma unroll
    for (int l = NH - 1; l >= 1; --l) {   // l = index of the layer whose dZ is in DZ[pp]; its input is H_l (image l-1)
      const char *dz = lds + LY::DZ + pp * 8192, *Hin = lds + LY::HIMG + (l - 1) * 8192;
      const char *Wimg = lds + LY::WIMG + (l - 1) * 32768;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 bq = tr_frag(dz, 16 * s, 32 * w, lane);
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) dW[l - 1][ib] = mfma_bf16(tr_frag(Hin, 16 * s, 32 * ib, lane), bq, dW[l - 1][ib]);
        db[l] = mfma_bf16(ones, bq, db[l]);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
#pragma unroll
      for (int s = 0; s < 8; ++s) acc = mfma_bf16(row_frag(Wimg, 32 * w + r, 2 * s + h), row_frag(dz, r, 2 * s + h), acc);
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = (mask[l - 1] >> j) & 1u ? acc[j] : 0.0f;
      pp ^= 1;
      store_tile(lds + LY::DZ + pp * 8192, 32 * w, acc, lane);
      __syncthreads();
    }
    // ---- backward: first layer ------------------------------------------------------------------------
    {
      const char *dz = lds + LY::DZ + pp * 8192;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 bq = tr_frag(dz, 16 * s, 32 * w, lane);
        const bf16x8 xt = *reinterpret_cast<const bf16x8 *>(Xt + (size_t)r * p.Npad + row0 + 16 * s + 8 * h);
        dW1 = mfma_bf16(xt, bq, dW1);
        db[0] = mfma_bf16(ones, bq, db[0]);
      }
    }
  }

  // ---- write this workgroup's slab: every parameter is owned by exactly one lane ----------------------
#pragma unroll
  for (int l = 1; l < NH; ++l)
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int j = 0; j < 16; ++j) slab[sp.w_off[l] + (32 * ib + acc_m(j, h)) * 128 + 32 * w + r] = dW[l - 1][ib][j];
#pragma unroll
  for (int l = 0; l < NH; ++l)
    if (h == 0) slab[sp.b_off[l] + 32 * w + r] = db[l][0];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int f = acc_m(j, h);
    if (f < F) slab[sp.w_off[0] + f * 128 + 32 * w + r] = dW1[j];
    if (r < 2) slab[sp.w_off[NH] + (32 * w + f) * 2 + r] = dWo[j];
  }
  if (w == 0) {
    if (h == 0 && r < 2) slab[sp.b_off[NH] + r] = dbo[0];
    ll_acc = wave_sum(ll_acc);
    if (lane == 0) p.llpart[(size_t)e * p.S + sidx] = ll_acc;
  }
}
