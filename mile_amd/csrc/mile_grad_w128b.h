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
// forward (transposed read) and backward (row read) the same way.
//
// One wave per SIMD leaves nobody to hide a wave's own LDS latency, MFMA drain and barrier waits, so
// each iteration walks RT = 2 independent row tiles between the same 2*NH+1 barriers: the second
// tile's loads and MFMAs fill the first tile's epilogue and vice versa, and the barrier cost per
// tile halves.  Bias gradients are column sums of the same transposed dZ fragments the dW products
// read (v_dot2c_f32_bf16 against ones), which keeps 64 accumulator registers free for that.
#pragma once
#include "mile_bf16_frag.h"
#include "mile_device.h"
#include "mile_grad_generic.h"

template <int NH, int RT>
struct W128Layout {
  static constexpr int WIMG = 0;                              // W_2..W_NH: [128 in][128 out] bf16, 32 KiB each
  static constexpr int WOT = WIMG + (NH - 1) * 32768;         // head weights transposed: [16 (k, zero padded)][128 in]
  static constexpr int TILE = WOT + 4096;                     // per row tile: H_1..H_NH, dZ ping-pong ([32][128] each)
  static constexpr int HIMG = 0;                              //   offsets inside a tile set
  static constexpr int DZ = NH * 8192;                        //   d(out) aliases DZ[1] (free while it is live)
  static constexpr int TILE_BYTES = (NH + 2) * 8192;
  static constexpr int BIAS = TILE + RT * TILE_BYTES;         // bias tiles in accumulator layout: [NH][4 waves][2 halves][16] fp32
  static constexpr int BYTES = BIAS + NH * 4 * 2 * 16 * 4;
};

__device__ __forceinline__ float bf16_colsum(const bf16x8 v, float acc) {   // acc + sum of the 8 elements
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bf16x2 pr = {v[2 * i], v[2 * i + 1]}, one = {(bf16)1.0f, (bf16)1.0f};
    acc = __builtin_amdgcn_fdot2_f32_bf16(pr, one, acc, false);
  }
  return acc;
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

// H = relu(z) as bf16 into the image: rounding and ReLU commute, and on bf16 bit patterns ReLU is a signed
// 16-bit max with 0 (v_pk_max_i16), two elements per instruction.
__device__ __forceinline__ void store_tile_relu(char *img, int col0, const f32x16 &z, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4_t v = {z[4 * g], z[4 * g + 1], z[4 * g + 2], z[4 * g + 3]};
    const s16x4 b = __builtin_bit_cast(s16x4, __builtin_convertvector(v, bf16x4));
    const s16x4 zero = {0, 0, 0, 0};
    *reinterpret_cast<s16x4 *>(img + img_off(r, (col0 >> 3) + g) + 8 * h) = __builtin_elementwise_max(b, zero);
  }
}

// dZ = dH * relu'(z) as bf16 into dzimg; relu'(z) = (H != 0) is read back from this wave's own slice of the
// H image (bit patterns: min(H, 1) is 0 or 1, times the dZ bits: v_pk_min_u16 + v_pk_mul_lo_u16).
__device__ __forceinline__ void store_tile_masked(char *dzimg, const char *himg, int col0, const f32x16 &dh, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const uint32_t ones = 0x00010001u;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int off = img_off(r, (col0 >> 3) + g) + 8 * h;
    const f32x4_t v = {dh[4 * g], dh[4 * g + 1], dh[4 * g + 2], dh[4 * g + 3]};
    const f32x2_t b = __builtin_bit_cast(f32x2_t, __builtin_convertvector(v, bf16x4));   // two packed pairs
    const f32x2_t hb = *reinterpret_cast<const f32x2_t *>(himg + off);
    f32x2_t o;
#pragma unroll
    for (int k = 0; k < 2; ++k) {   // inline asm: the compiler would turn x * min(h, 1) back into compare + select
      float m, t;
      asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(hb[k]), "v"(ones));
      asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(t) : "v"(b[k]), "v"(m));
      o[k] = t;
    }
    *reinterpret_cast<f32x2_t *>(dzimg + off) = o;
  }
}

template <int NH, int RT>
__global__ __launch_bounds__(256) void k_grad_w128b(const GradParams p) {
  using LY = W128Layout<NH, RT>;
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
    const float *Wo = th + sp.w_off[NH];
    for (int c = tid; c < 16 * 16; c += 256) {
      const int row = c >> 4, ch = c & 15;   // row = head output k
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (bf16)(row < 2 ? Wo[(ch * 8 + j) * 2 + row] : 0.0f);
      *reinterpret_cast<bf16x8 *>(lds + LY::WOT + img_off(row, ch)) = v;
    }
  }
  // first-layer weights: one A fragment per wave, A[m = out 32w + r][k = in 8h + j], kept in registers
  bf16x8 w1frag;
  {
    const float *W1 = th + sp.w_off[0];
#pragma unroll
    for (int j = 0; j < 8; ++j) w1frag[j] = (bf16)(8 * h + j < F ? W1[(8 * h + j) * 128 + 32 * w + r] : 0.0f);
  }
  // bias tiles in accumulator layout, read back with the other fragments of a phase: the first MFMA of a
  // layer takes the bias tile as its C operand, so the add is free
  for (int i = tid; i < NH * 128; i += 256) {
    const int l = i >> 7, c = i & 127, ww = c >> 5, hh = (c >> 4) & 1, j = c & 15;
    reinterpret_cast<float *>(lds + LY::BIAS)[i] = th[sp.b_off[l] + 32 * ww + acc_m(j, hh)];
  }
  auto bias_tile = [&](int l) {
    const f32x4_t *bp = reinterpret_cast<const f32x4_t *>(lds + LY::BIAS + ((l * 4 + w) * 2 + h) * 64);
    f32x16 b;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4_t v = bp[g];
      b[4 * g] = v[0]; b[4 * g + 1] = v[1]; b[4 * g + 2] = v[2]; b[4 * g + 3] = v[3];
    }
    return b;
  };
  const float bo0 = th[sp.b_off[NH]], bo1 = th[sp.b_off[NH] + 1];
  __syncthreads();

  f32x16 dW[NH > 1 ? NH - 1 : 1][4], dW1, dWo;
  float db[NH], dbo = 0.0f;   // per-lane partial column sums (this lane's 8-row halves)
#pragma unroll
  for (int l = 0; l < NH; ++l) db[l] = 0.0f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    dW1[j] = 0.0f; dWo[j] = 0.0f;
#pragma unroll
    for (int l = 0; l < NH - 1; ++l)
#pragma unroll
      for (int ib = 0; ib < 4; ++ib) dW[l][ib][j] = 0.0f;
  }
  float ll_acc = 0.0f;

  const int NBS = p.Npb / (32 * RT);   // super tiles of RT row tiles
  const int nb0 = (int)((long long)sidx * NBS / p.S), nb1 = (int)((long long)(sidx + 1) * NBS / p.S);

  for (int t = nb0; t < nb1; ++t) {
    f32x16 acc[RT];
    // ---- forward ---------------------------------------------------------------------------
#pragma unroll
    for (int l = 0; l < NH; ++l) {
      if (l == 0) {
        const f32x16 b0 = bias_tile(0);
#pragma unroll
        for (int q = 0; q < RT; ++q) {
          const int row0 = 32 * (t * RT + q);
          const bf16x8 xb = *reinterpret_cast<const bf16x8 *>(Xb + (size_t)(row0 + r) * 16 + 8 * h);
          acc[q] = mfma_bf16(w1frag, xb, b0);
        }
      } else {
        const char *Wimg = lds + LY::WIMG + (l - 1) * 32768;
        // all fragment reads of the phase are issued before the first MFMA: with one wave per SIMD nothing
        // else hides the LDS latency, and read-then-use pairs would pay it once per MFMA
        bf16x8 af[8], bfr[RT][8];
        const f32x16 bl = bias_tile(l);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          af[s] = tr_frag(Wimg, 16 * s, 32 * w, lane);
#pragma unroll
          for (int q = 0; q < RT; ++q)
            bfr[q][s] = row_frag(lds + LY::TILE + q * LY::TILE_BYTES + LY::HIMG + (l - 1) * 8192, r, 2 * s + h);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the reads above, the MFMAs below
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
          for (int q = 0; q < RT; ++q) acc[q] = mfma_bf16(af[s], bfr[q][s], s == 0 ? bl : acc[q]);
      }
#pragma unroll
      for (int q = 0; q < RT; ++q)
        store_tile_relu(lds + LY::TILE + q * LY::TILE_BYTES + LY::HIMG + l * 8192, 32 * w, acc[q], lane);
      __syncthreads();
    }
    // ---- head (every wave computes it; wave 0 publishes d(out) into DZ[1]) ----------------------------
    {
      f32x16 zero16;
#pragma unroll
      for (int j = 0; j < 16; ++j) zero16[j] = 0.0f;
      bf16x8 af[8], bfr[RT][8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        af[s] = row_frag(lds + LY::WOT, r & 15, 2 * s + h);
#pragma unroll
        for (int q = 0; q < RT; ++q)
          bfr[q][s] = row_frag(lds + LY::TILE + q * LY::TILE_BYTES + LY::HIMG + (NH - 1) * 8192, r, 2 * s + h);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int q = 0; q < RT; ++q) acc[q] = mfma_bf16(af[s], bfr[q][s], s == 0 ? zero16 : acc[q]);
    }
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      char *DOimg = lds + LY::TILE + q * LY::TILE_BYTES + LY::DZ + 8192;
      const int row0 = 32 * (t * RT + q);
      float dmu = 0.0f, dsg = 0.0f;
      if (h == 0 && row0 + r < p.N) {
        const float ll = row_loss_regr(acc[q][0] + bo0, acc[q][1] + bo1, yv[row0 + r], dmu, dsg);
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
    }
    __syncthreads();
    // ---- backward: head ---------------------------------------------------------------------------
    {
      f32x16 zero16;
#pragma unroll
      for (int j = 0; j < 16; ++j) zero16[j] = 0.0f;
      const bf16x8 a = tr_frag(lds + LY::WOT, 0, 32 * w, lane);
#pragma unroll
      for (int q = 0; q < RT; ++q) {
        char *ts = lds + LY::TILE + q * LY::TILE_BYTES;
        const char *Hin = ts + LY::HIMG + (NH - 1) * 8192, *DOimg = ts + LY::DZ + 8192;
        acc[q] = mfma_bf16(a, row_frag(DOimg, r, h), zero16);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const bf16x8 bq = tr_frag(DOimg, 16 * s, 0, lane);
          dWo = mfma_bf16(tr_frag(Hin, 16 * s, 32 * w, lane), bq, dWo);
          dbo = bf16_colsum(bq, dbo);
        }
        store_tile_masked(ts + LY::DZ, Hin, 32 * w, acc[q], lane);
      }
    }
    __syncthreads();
    // ---- backward: hidden layers NH .. 2 ------------------------------------------------------------
    int pp = 0;
#pragma unroll
    for (int l = NH - 1; l >= 1; --l) {   // dZ of layer l is in DZ[pp]; its input is H_l (image l-1)
      const char *Wimg = lds + LY::WIMG + (l - 1) * 32768;
      {
        f32x16 zero16;
#pragma unroll
        for (int j = 0; j < 16; ++j) zero16[j] = 0.0f;
        bf16x8 af[8], bfr[RT][8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          af[s] = row_frag(Wimg, 32 * w + r, 2 * s + h);
#pragma unroll
          for (int q = 0; q < RT; ++q)
            bfr[q][s] = row_frag(lds + LY::TILE + q * LY::TILE_BYTES + LY::DZ + pp * 8192, r, 2 * s + h);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
          for (int q = 0; q < RT; ++q) acc[q] = mfma_bf16(af[s], bfr[q][s], s == 0 ? zero16 : acc[q]);
      }
#pragma unroll
      for (int q = 0; q < RT; ++q) {
        char *ts = lds + LY::TILE + q * LY::TILE_BYTES;
        const char *dz = ts + LY::DZ + pp * 8192, *Hin = ts + LY::HIMG + (l - 1) * 8192;
        bf16x8 bq[2], ah[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bq[s] = tr_frag(dz, 16 * s, 32 * w, lane);
#pragma unroll
          for (int ib = 0; ib < 4; ++ib) ah[s][ib] = tr_frag(Hin, 16 * s, 32 * ib, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int ib = 0; ib < 4; ++ib) dW[l - 1][ib] = mfma_bf16(ah[s][ib], bq[s], dW[l - 1][ib]);
          db[l] = bf16_colsum(bq[s], db[l]);
        }
        store_tile_masked(ts + LY::DZ + (pp ^ 1) * 8192, Hin, 32 * w, acc[q], lane);
      }
      pp ^= 1;
      __syncthreads();
    }
    // ---- backward: first layer ------------------------------------------------------------------------
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      const char *dz = lds + LY::TILE + q * LY::TILE_BYTES + LY::DZ + pp * 8192;
      const int row0 = 32 * (t * RT + q);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 bq = tr_frag(dz, 16 * s, 32 * w, lane);
        const bf16x8 xt = *reinterpret_cast<const bf16x8 *>(Xt + (size_t)r * p.Npb + row0 + 16 * s + 8 * h);
        dW1 = mfma_bf16(xt, bq, dW1);
        db[0] = bf16_colsum(bq, db[0]);
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
  for (int l = 0; l < NH; ++l) {
    const float tot = db[l] + __shfl_xor(db[l], 32);
    if (h == 0) slab[sp.b_off[l] + 32 * w + r] = tot;
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int f = acc_m(j, h);
    if (f < F) slab[sp.w_off[0] + f * 128 + 32 * w + r] = dW1[j];
    if (r < 2) slab[sp.w_off[NH] + (32 * w + f) * 2 + r] = dWo[j];
  }
  {
    const float tot = dbo + __shfl_xor(dbo, 32);
    if (w == 0 && h == 0 && r < 2) slab[sp.b_off[NH] + r] = tot;
  }
  if (w == 0) {
    ll_acc = wave_sum(ll_acc);
    if (lane == 0) p.llpart[(size_t)e * p.S + sidx] = ll_acc;
  }
}
