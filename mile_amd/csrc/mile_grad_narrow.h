// k_grad_narrow<NH, TH, TF> -- grad of the log-likelihood for NARROW FCNs on the matrix pipe (round 3).
//
// The nets the reference actually ships are 16 and 32 wide: experiments/replicate_uci/mclmc.yaml [16,16,2] (d = 402),
// the README example [16,16,16,2], experiments/datasize_ablation/protein_mclmc.yaml [16,16,16,2] (ReLU regression) and
// experiments/tabluar_classif/covertype.yaml [32,7] (sigmoid, 7-class softmax head).  Until round 3 they all took
// k_grad_generic (VALU, one FMA chain per output element, weights through L1).  This kernel covers 1-3 hidden layers of width
// <= 64 (<= 32: weights in registers; 33..64: weights in LDS, template flag WL) or 4-10 hidden layers of width <= 16 (the
// reference's depth ablations), F <= 64 inputs, <= 16 outputs, ReLU / tanh / sigmoid (src/config/models/base.py:25-39), both heads
// (src/training/probabilistic.py:92-109) in one fused forward + backward pass per 16-row tile, every Dense product
// (src/flax_building_blocks/basic.py:42-61) on v_mfma_f32_16x16x4_f32: fp32 operands, a k-ordered fp32 fmaf chain per output --
// fp32 arithmetic proper, no bf16 split needed (at these widths the matrix pipe is nowhere near the bound; the kernel is
// latency / issue bound and the MFMA is simply the cheapest way to move 256 multiply-adds per instruction).
//
// CDNA4 design.  A 16x16 fp32 MFMA tile has its column on the lane (c = lane & 15) and its rows in the lane group / register
// (row = 4 g + j, g = lane >> 4, j = 0..3); the A / B operands have (M or N index) = c and k = g, ONE value per lane.  The
// "k slot" of lane group g in the j-th MFMA of a product can stand for ANY contraction index as long as A and B agree, so with
// slot g of MFMA j := index 4 g + j an accumulator tile is, register j for MFMA j, directly an operand of the next product:
//   L2 layout ("row on the lane"):   reg j of lane (g, c) = M[feature 4 g + j][data row c]   = C of  Z^T = W^T H^T
//   L1 layout ("feature on the lane"): reg j of lane (g, c) = M[data row 4 g + j][feature c]
//   * forward,   Z_l^T   = W_l^T H_{l-1}^T : A = W_l regs,  B = H_{l-1} in L2  ->  Z_l^T in L2       (chain, no data movement)
//   * backward, dH_{l-1}^T = W_l dZ_l^T     : A = W_l regs', B = dZ_l in L2     ->  dH_{l-1}^T in L2   (chain, no data movement)
//   * dW_l = H_{l-1}^T dZ_l (contraction over the 16 rows): A = H_{l-1} in L1, B = dZ_l in L1 -> accumulates in registers over
//     all the wave's tiles; the bias gradient is the per-lane sum of the same dZ_l registers.
// Only the L2 -> L1 copies (one per activation tile and per dZ tile) go through a private 16 x 20 LDS image of the wave
// (4 ds_write_b32 + 1 ds_read_b128, conflict-free, no barrier: one wave's LDS operations complete in order).  X is loaded from
// the zero-padded copy Xp in both layouts straight from global memory; weights are loaded ONCE per workgroup into registers in
// the two operand arrangements (forward / dH).  One wave = one 16-row tile at a time; a workgroup = 1-4 waves (chosen at launch
// so that small ensembles still fill the chip: E = 12, N = 1052 -> 66 one-wave workgroups per particle); grid (S, E), partial
// gradients into slab[e][s] as every other grad kernel, reduced over the workgroup's waves through LDS in fixed order.
#pragma once
#include "mile_grad_generic.h"

#define NRW_MAXW 4              // waves per workgroup (upper bound)
#define NRW_TS 20               // row stride (floats) of the per-wave transposition image

// WL = true (hidden widths 33..64, TH = 3 or 4): the weights do not fit the register file next to the accumulators any more and
// live in LDS instead, as padded fp32 images in BOTH operand orientations (row stride = 16 T + 4 floats: the four lane groups of
// an A-operand read then fall on four different quarters of the banks, conflict-free): forward images [in][out], dH images
// [out][in].  The cross-wave reduction buffer aliases them after the tile loop.
template <int NH, int TH, int TF, bool WL = false>
struct NarrowLayout {
  static constexpr int TRANS = NRW_MAXW * 16 * NRW_TS;                         // per-wave 16 x 20 images
  // cross-wave reduction image: one wave's accumulators, 64 lanes x NACC floats
  static constexpr int NACC = 4 * (TF * TH + (NH - 1) * TH * TH + TH) + (NH * TH + 1) + 1;
  static constexpr int RED = 64 * NACC;
  static constexpr int HP = 16 * TH, SH = HP + 4, SL = 20;                     // padded hidden width and the image row strides
  static constexpr int W0F = 0;                                                // [16 TF][SH]
  static constexpr int WHF = W0F + 16 * TF * SH;                               // (NH - 1) x [HP][SH]
  static constexpr int WHB = WHF + (NH - 1) * HP * SH;                         // (NH - 1) x [HP][SH], transposed
  static constexpr int WLF = WHB + (NH - 1) * HP * SH;                         // [HP][SL]
  static constexpr int WLB = WLF + HP * SL;                                    // [16][SH], transposed
  static constexpr int WTOT = WLB + 16 * SH;
  static constexpr int FLOATS = TRANS + (WL ? (WTOT > RED ? WTOT : RED) : RED);
  static constexpr int BYTES = FLOATS * 4;
};

__device__ __forceinline__ f32x4 nrw_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// L2 -> L1 through the wave's private image: v[j] = M[feat 4 g + j][row c]  ->  w[j] = M[feat c][row 4 g + j]
__device__ __forceinline__ f32x4 nrw_transpose(float *img, int g, int c, const f32x4 v) {
#pragma unroll
  for (int j = 0; j < 4; ++j) img[(4 * g + j) * NRW_TS + c] = v[j];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const f32x4 w = *reinterpret_cast<const f32x4 *>(img + c * NRW_TS + 4 * g);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return w;
}

__device__ __forceinline__ float nrw_xor16(float v) { return __shfl_xor(v, 16, 64); }
__device__ __forceinline__ float nrw_xor32(float v) { return __shfl_xor(v, 32, 64); }

template <int NH, int TH, int TF, bool WL = false>
static __global__ __launch_bounds__(64 * NRW_MAXW) void k_grad_narrow(const GradParams p) {
  extern __shared__ __attribute__((aligned(16))) float nrw_lds[];
  using LY = NarrowLayout<NH, TH, TF, WL>;
  const DevSpec &sp = p.spec;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int e = blockIdx.y, s = blockIdx.x;
  const int F = sp.in_features, K = sp.widths[NH], act = sp.activation;
  const bool regr = sp.task == MILE_TASK_REGRESSION;
  const float *th = p.theta + (size_t)e * sp.d;
  float *img = nrw_lds + wave * 16 * NRW_TS;

  // ---- weights -> registers, both operand arrangements (zero beyond the real widths) ---------------------------------
  // forward (A of Z^T = W^T H^T):  wf[ti][t][j] = W[in 16 ti + 4 g + j][out 16 t + c]
  // dH      (A of dH^T = W dZ^T):  wb[ti][t][j] = W[in 16 ti + c][out 16 t + 4 g + j]
  constexpr int WR = WL ? 1 : TH, WRF = WL ? 1 : TF;             // (register weight arrays collapse in the LDS form)
  float w0f[WRF][WR][4];
  float whf[NH > 1 ? NH - 1 : 1][WR][WR][4], whb[NH > 1 ? NH - 1 : 1][WR][WR][4];
  float wlf[WR][4], wlb[WR][4];
  float b0[TH][4], bh[NH > 1 ? NH - 1 : 1][TH][4], bl[4];        // biases in L2: reg j <-> feature 4 g + j
  float *wimg = nrw_lds + LY::TRANS;                              // WL: the weight images
  if constexpr (WL) {
    const int nt = blockDim.x;
    {  // first layer, forward image [in][out]
      const int o0 = sp.widths[0];
      const float *W = th + sp.w_off[0];
      for (int i = tid; i < 16 * TF * LY::SH; i += nt) {
        const int in = i / LY::SH, out = i - in * LY::SH;
        wimg[LY::W0F + i] = (in < F && out < o0) ? W[in * o0 + out] : 0.0f;
      }
    }
#pragma unroll
    for (int l = 1; l < NH; ++l) {
      const int wi = sp.widths[l - 1], wo = sp.widths[l];
      const float *W = th + sp.w_off[l];
      float *Wf = wimg + LY::WHF + (l - 1) * LY::HP * LY::SH, *Wb = wimg + LY::WHB + (l - 1) * LY::HP * LY::SH;
      for (int i = tid; i < LY::HP * LY::SH; i += nt) {
        const int r = i / LY::SH, q = i - r * LY::SH;
        Wf[i] = (r < wi && q < wo) ? W[r * wo + q] : 0.0f;          // [in r][out q]
        Wb[i] = (q < wi && r < wo) ? W[q * wo + r] : 0.0f;          // [out r][in q]
      }
    }
    {
      const int wi = sp.widths[NH - 1];
      const float *W = th + sp.w_off[NH];
      for (int i = tid; i < LY::HP * LY::SL; i += nt) {
        const int in = i / LY::SL, k = i - in * LY::SL;
        wimg[LY::WLF + i] = (in < wi && k < K) ? W[in * K + k] : 0.0f;
      }
      for (int i = tid; i < 16 * LY::SH; i += nt) {
        const int k = i / LY::SH, in = i - k * LY::SH;
        wimg[LY::WLB + i] = (in < wi && k < K) ? W[in * K + k] : 0.0f;
      }
    }
    __syncthreads();
  }
  // A operands of the five kinds of product: from registers, or (WL) one ds_read_b32 per MFMA from the images
  auto a_f0 = [&](int ti, int t, int j) -> float {
    if constexpr (WL) return wimg[LY::W0F + (16 * ti + 4 * g + j) * LY::SH + 16 * t + c];
    else return w0f[ti][t][j];
  };
  auto a_fh = [&](int l, int ti, int t, int j) -> float {
    if constexpr (WL) return wimg[LY::WHF + (l - 1) * LY::HP * LY::SH + (16 * ti + 4 * g + j) * LY::SH + 16 * t + c];
    else return whf[l - 1][ti][t][j];
  };
  auto a_bh = [&](int l, int ti, int t, int j) -> float {
    if constexpr (WL) return wimg[LY::WHB + (l - 1) * LY::HP * LY::SH + (16 * t + 4 * g + j) * LY::SH + 16 * ti + c];
    else return whb[l - 1][ti][t][j];
  };
  auto a_fl = [&](int ti, int j) -> float {
    if constexpr (WL) return wimg[LY::WLF + (16 * ti + 4 * g + j) * LY::SL + c];
    else return wlf[ti][j];
  };
  auto a_bl = [&](int ti, int j) -> float {
    if constexpr (WL) return wimg[LY::WLB + (4 * g + j) * LY::SH + 16 * ti + c];
    else return wlb[ti][j];
  };
  {
    const int o0 = sp.widths[0];
    const float *W = th + sp.w_off[0], *B = th + sp.b_off[0];
    if constexpr (!WL) {
#pragma unroll
    for (int ti = 0; ti < TF; ++ti)
#pragma unroll
      for (int t = 0; t < TH; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int in = 16 * ti + 4 * g + j, out = 16 * t + c;
          w0f[ti][t][j] = (in < F && out < o0) ? W[in * o0 + out] : 0.0f;
        }
    }
#pragma unroll
    for (int t = 0; t < TH; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) b0[t][j] = (16 * t + 4 * g + j < o0) ? B[16 * t + 4 * g + j] : 0.0f;
  }
#pragma unroll
  for (int l = 1; l < NH; ++l) {
    const int wi = sp.widths[l - 1], wo = sp.widths[l];
    const float *W = th + sp.w_off[l], *B = th + sp.b_off[l];
    if constexpr (!WL) {
#pragma unroll
    for (int ti = 0; ti < TH; ++ti)
#pragma unroll
      for (int t = 0; t < TH; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int inf = 16 * ti + 4 * g + j, outf = 16 * t + c;
          whf[l - 1][ti][t][j] = (inf < wi && outf < wo) ? W[inf * wo + outf] : 0.0f;
          const int inb = 16 * ti + c, outb = 16 * t + 4 * g + j;
          whb[l - 1][ti][t][j] = (inb < wi && outb < wo) ? W[inb * wo + outb] : 0.0f;
        }
    }
    (void)wi;
#pragma unroll
    for (int t = 0; t < TH; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) bh[l - 1][t][j] = (16 * t + 4 * g + j < wo) ? B[16 * t + 4 * g + j] : 0.0f;
  }
  {
    const int wi = sp.widths[NH - 1];
    const float *W = th + sp.w_off[NH], *B = th + sp.b_off[NH];
    if constexpr (!WL) {
#pragma unroll
    for (int ti = 0; ti < TH; ++ti)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int inf = 16 * ti + 4 * g + j;
        wlf[ti][j] = (inf < wi && c < K) ? W[inf * K + c] : 0.0f;
        const int inb = 16 * ti + c, outb = 4 * g + j;
        wlb[ti][j] = (inb < wi && outb < K) ? W[inb * K + outb] : 0.0f;
      }
    }
    (void)wi; (void)W;
#pragma unroll
    for (int j = 0; j < 4; ++j) bl[j] = (4 * g + j < K) ? B[4 * g + j] : 0.0f;
  }

  // ---- accumulators (live over all the wave's tiles) ---------------------------------------------------------------------
  f32x4 dw0[TF][TH], dwh[NH > 1 ? NH - 1 : 1][TH][TH], dwl[TH];
  float db0[TH], dbh[NH > 1 ? NH - 1 : 1][TH], dbl = 0.0f, ll_acc = 0.0f;
#pragma unroll
  for (int ti = 0; ti < TF; ++ti)
#pragma unroll
    for (int t = 0; t < TH; ++t) dw0[ti][t] = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int l = 0; l < (NH > 1 ? NH - 1 : 1); ++l)
#pragma unroll
    for (int ti = 0; ti < TH; ++ti) {
#pragma unroll
      for (int t = 0; t < TH; ++t) dwh[l][ti][t] = f32x4{0, 0, 0, 0};
      dbh[l][ti] = 0.0f;
    }
#pragma unroll
  for (int t = 0; t < TH; ++t) { dwl[t] = f32x4{0, 0, 0, 0}; db0[t] = 0.0f; }

  // ---- the wave's tiles --------------------------------------------------------------------------------------------------
  const int rows_per = ((p.N + p.S - 1) / p.S + 15) / 16 * 16;      // whole tiles per split
  const int r_begin = s * rows_per, r_end = min(p.N, r_begin + rows_per);
  const int Fp = p.Fp;
  for (int r0 = r_begin + 16 * wave; r0 < r_end; r0 += 16 * nw) {
    // X in both layouts from the zero-padded copy (rows up to Npad + 32 exist; columns up to Fp)
    f32x4 x2[TF], x1[TF];
#pragma unroll
    for (int ti = 0; ti < TF; ++ti) {
      const int f2 = 16 * ti + 4 * g;
      x2[ti] = (f2 < Fp) ? *reinterpret_cast<const f32x4 *>(p.Xp + (size_t)(r0 + c) * Fp + f2) : f32x4{0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < 4; ++j) x1[ti][j] = (16 * ti + c < Fp) ? p.Xp[(size_t)(r0 + 4 * g + j) * Fp + 16 * ti + c] : 0.0f;
    }
    const bool row_ok = r0 + c < r_end;
    float yv = 0.0f;
    int yi = 0;
    if (regr) yv = ((const float *)p.y)[r0 + c];
    else yi = ((const int32_t *)p.y)[r0 + c];

    // ---- forward: hidden layers in L2, an L1 copy of every activation tile for the dW products ----
    f32x4 h2[NH][TH], h1[NH][TH];
#pragma unroll
    for (int t = 0; t < TH; ++t) {
      f32x4 z = {b0[t][0], b0[t][1], b0[t][2], b0[t][3]};
#pragma unroll
      for (int ti = 0; ti < TF; ++ti)
#pragma unroll
        for (int j = 0; j < 4; ++j) z = nrw_mfma(a_f0(ti, t, j), x2[ti][j], z);
#pragma unroll
      for (int j = 0; j < 4; ++j) z[j] = act_fwd(act, z[j]);
      h2[0][t] = z;
    }
#pragma unroll
    for (int l = 1; l < NH; ++l)
#pragma unroll
      for (int t = 0; t < TH; ++t) {
        f32x4 z = {bh[l - 1][t][0], bh[l - 1][t][1], bh[l - 1][t][2], bh[l - 1][t][3]};
#pragma unroll
        for (int ti = 0; ti < TH; ++ti)
#pragma unroll
          for (int j = 0; j < 4; ++j) z = nrw_mfma(a_fh(l, ti, t, j), h2[l - 1][ti][j], z);
#pragma unroll
        for (int j = 0; j < 4; ++j) z[j] = act_fwd(act, z[j]);
        h2[l][t] = z;
      }
    f32x4 zo = {bl[0], bl[1], bl[2], bl[3]};          // out^T in L2: reg j of lane (g, c) = out[class 4 g + j][row c]
#pragma unroll
    for (int ti = 0; ti < TH; ++ti)
#pragma unroll
      for (int j = 0; j < 4; ++j) zo = nrw_mfma(a_fl(ti, j), h2[NH - 1][ti][j], zo);
#pragma unroll
    for (int l = 0; l < NH; ++l)
#pragma unroll
      for (int t = 0; t < TH; ++t) h1[l][t] = nrw_transpose(img, g, c, h2[l][t]);

    // ---- head: log-likelihood of row c and d(out), probabilistic.py:92-109, NaN rows contribute nothing (nansum) ----
    f32x4 dz2 = {0, 0, 0, 0};                          // dZ_last^T in L2
    if (regr) {
      float dmu = 0.0f, ds = 0.0f, ll = 0.0f;
      if (g == 0) ll = row_loss_regr(zo[0], zo[1], yv, dmu, ds);
      if (g == 0 && row_ok) { dz2[0] = dmu; dz2[1] = ds; ll_acc += ll; }
    } else {
      float m = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) m = (4 * g + j < K) ? fmaxf(m, zo[j]) : m;
      m = fmaxf(m, nrw_xor16(m));
      m = fmaxf(m, nrw_xor32(m));
      float se = 0.0f, zy = 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        se += (4 * g + j < K) ? expf(zo[j] - m) : 0.0f;
        zy += (4 * g + j == yi) ? zo[j] : 0.0f;
      }
      se += nrw_xor16(se); se += nrw_xor32(se);
      zy += nrw_xor16(zy); zy += nrw_xor32(zy);
      const float lse = m + logf(se);
      const float ll = zy - lse;
      const bool bad = isnan(ll) || !row_ok;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        dz2[j] = (bad || 4 * g + j >= K) ? 0.0f : ((4 * g + j == yi ? 1.0f : 0.0f) - expf(zo[j] - lse));
      if (g == 0 && !bad) ll_acc += ll;
    }

    // ---- backward ----
    f32x4 dz1 = nrw_transpose(img, g, c, dz2);        // dZ_last in L1: reg j = dZ[row 4 g + j][class c]
    dbl += dz1[0] + dz1[1] + dz1[2] + dz1[3];
#pragma unroll
    for (int ti = 0; ti < TH; ++ti)
#pragma unroll
      for (int j = 0; j < 4; ++j) dwl[ti] = nrw_mfma(h1[NH - 1][ti][j], dz1[j], dwl[ti]);
    // dH of the last hidden layer (L2), then dZ = dH * act'(H)
    f32x4 dzh2[TH], dzh1[TH];
#pragma unroll
    for (int ti = 0; ti < TH; ++ti) {
      f32x4 dh = {0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < 4; ++j) dh = nrw_mfma(a_bl(ti, j), dz2[j], dh);
#pragma unroll
      for (int j = 0; j < 4; ++j) dh[j] *= act_bwd(act, h2[NH - 1][ti][j]);
      dzh2[ti] = dh;
    }
#pragma unroll
    for (int l = NH - 1; l >= 1; --l) {
#pragma unroll
      for (int t = 0; t < TH; ++t) {
        dzh1[t] = nrw_transpose(img, g, c, dzh2[t]);
        dbh[l - 1][t] += dzh1[t][0] + dzh1[t][1] + dzh1[t][2] + dzh1[t][3];
      }
#pragma unroll
      for (int ti = 0; ti < TH; ++ti)
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) dwh[l - 1][ti][t] = nrw_mfma(h1[l - 1][ti][j], dzh1[t][j], dwh[l - 1][ti][t]);
      f32x4 nx[TH];
#pragma unroll
      for (int ti = 0; ti < TH; ++ti) {
        f32x4 dh = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) dh = nrw_mfma(a_bh(l, ti, t, j), dzh2[t][j], dh);
#pragma unroll
        for (int j = 0; j < 4; ++j) dh[j] *= act_bwd(act, h2[l - 1][ti][j]);
        nx[ti] = dh;
      }
#pragma unroll
      for (int ti = 0; ti < TH; ++ti) dzh2[ti] = nx[ti];
    }
    // first layer: dW_0 = X^T dZ_0, db_0
#pragma unroll
    for (int t = 0; t < TH; ++t) {
      dzh1[t] = nrw_transpose(img, g, c, dzh2[t]);
      db0[t] += dzh1[t][0] + dzh1[t][1] + dzh1[t][2] + dzh1[t][3];
    }
#pragma unroll
    for (int ti = 0; ti < TF; ++ti)
#pragma unroll
      for (int t = 0; t < TH; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) dw0[ti][t] = nrw_mfma(x1[ti][j], dzh1[t][j], dw0[ti][t]);
  }

  // ---- reduction over the workgroup's waves (fixed order), then the slab ----------------------------------------------------
  // bias gradients: per-lane partial sums over rows 4 g + j -> add the four lane groups
#pragma unroll
  for (int t = 0; t < TH; ++t) { db0[t] += nrw_xor16(db0[t]); db0[t] += nrw_xor32(db0[t]); }
#pragma unroll
  for (int l = 0; l < (NH > 1 ? NH - 1 : 1); ++l)
#pragma unroll
    for (int t = 0; t < TH; ++t) { dbh[l][t] += nrw_xor16(dbh[l][t]); dbh[l][t] += nrw_xor32(dbh[l][t]); }
  dbl += nrw_xor16(dbl); dbl += nrw_xor32(dbl);
  ll_acc = wave_sum(ll_acc);

  float *red = nrw_lds + LY::TRANS;                 // (WL: aliases the weight images -- every wave is past its tile loop first)
  if constexpr (WL) __syncthreads();
  float *slab = p.slabs + ((size_t)e * p.S + s) * p.dp;
  // every wave but 0 parks its accumulators in turn; wave 0 adds them in wave order
  for (int w = 1; w < nw; ++w) {
    __syncthreads();
    if (wave == w) {
      int k = 0;
#pragma unroll
      for (int ti = 0; ti < TF; ++ti)
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) red[(k++) * 64 + lane] = dw0[ti][t][j];
#pragma unroll
      for (int l = 0; l < NH - 1; ++l)
#pragma unroll
        for (int ti = 0; ti < TH; ++ti)
#pragma unroll
          for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[(k++) * 64 + lane] = dwh[l][ti][t][j];
#pragma unroll
      for (int ti = 0; ti < TH; ++ti)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[(k++) * 64 + lane] = dwl[ti][j];
#pragma unroll
      for (int t = 0; t < TH; ++t) red[(k++) * 64 + lane] = db0[t];
#pragma unroll
      for (int l = 0; l < NH - 1; ++l)
#pragma unroll
        for (int t = 0; t < TH; ++t) red[(k++) * 64 + lane] = dbh[l][t];
      red[(k++) * 64 + lane] = dbl;
      red[(k++) * 64 + lane] = ll_acc;
    }
    __syncthreads();
    if (wave == 0) {
      int k = 0;
#pragma unroll
      for (int ti = 0; ti < TF; ++ti)
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) dw0[ti][t][j] += red[(k++) * 64 + lane];
#pragma unroll
      for (int l = 0; l < NH - 1; ++l)
#pragma unroll
        for (int ti = 0; ti < TH; ++ti)
#pragma unroll
          for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) dwh[l][ti][t][j] += red[(k++) * 64 + lane];
#pragma unroll
      for (int ti = 0; ti < TH; ++ti)
#pragma unroll
        for (int j = 0; j < 4; ++j) dwl[ti][j] += red[(k++) * 64 + lane];
#pragma unroll
      for (int t = 0; t < TH; ++t) db0[t] += red[(k++) * 64 + lane];
#pragma unroll
      for (int l = 0; l < NH - 1; ++l)
#pragma unroll
        for (int t = 0; t < TH; ++t) dbh[l][t] += red[(k++) * 64 + lane];
      dbl += red[(k++) * 64 + lane];
      ll_acc += red[(k++) * 64 + lane];
    }
  }
  if (wave != 0) return;
  // C layout of a dW tile: reg j of lane (g, c) = dW[in 16 ti + 4 g + j][out 16 t + c]
  {
    const int o0 = sp.widths[0];
    float *G = slab + sp.w_off[0], *GB = slab + sp.b_off[0];
#pragma unroll
    for (int ti = 0; ti < TF; ++ti)
#pragma unroll
      for (int t = 0; t < TH; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int in = 16 * ti + 4 * g + j, out = 16 * t + c;
          if (in < F && out < o0) G[in * o0 + out] = dw0[ti][t][j];
        }
#pragma unroll
    for (int t = 0; t < TH; ++t)
      if (g == 0 && 16 * t + c < o0) GB[16 * t + c] = db0[t];
  }
#pragma unroll
  for (int l = 1; l < NH; ++l) {
    const int wi = sp.widths[l - 1], wo = sp.widths[l];
    float *G = slab + sp.w_off[l], *GB = slab + sp.b_off[l];
#pragma unroll
    for (int ti = 0; ti < TH; ++ti)
#pragma unroll
      for (int t = 0; t < TH; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int in = 16 * ti + 4 * g + j, out = 16 * t + c;
          if (in < wi && out < wo) G[in * wo + out] = dwh[l - 1][ti][t][j];
        }
#pragma unroll
    for (int t = 0; t < TH; ++t)
      if (g == 0 && 16 * t + c < wo) GB[16 * t + c] = dbh[l - 1][t];
  }
  {
    const int wi = sp.widths[NH - 1];
    float *G = slab + sp.w_off[NH], *GB = slab + sp.b_off[NH];
#pragma unroll
    for (int ti = 0; ti < TH; ++ti)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int in = 16 * ti + 4 * g + j;
        if (in < wi && c < K) G[in * K + c] = dwl[ti][j];
      }
    if (g == 0 && c < K) GB[c] = dbl;
  }
  if (lane == 0) p.llpart[(size_t)e * p.S + s] = ll_acc;
}
