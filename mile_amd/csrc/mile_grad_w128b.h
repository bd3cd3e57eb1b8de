// k_grad_w128b<NH, RT>: full-batch log-likelihood gradient of a ReLU regression FCN with NH (1..3)
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
//  * each iteration walks RT = 2 row tiles between the same 2*NH barriers; they share every weight
//    fragment, and the barrier cost per tile halves;
//  * all fragment reads of a phase are issued before its first MFMA (sched_barrier fences: left alone,
//    the compiler sinks every read next to its use and pays the LDS latency once per MFMA); in the
//    backward layers the transposed dW operands are read between the dH MFMAs and the dZ epilogue
//    runs between the dW MFMAs;
//  * epilogues are packed: the bias tile is the first MFMA's C operand, ReLU and its derivative are
//    16-bit integer ops on bf16 bit patterns, bias gradients are v_dot2c_f32_bf16 column sums of the
//    transposed dZ fragments the dW products read anyway;
//  * the head never touches a full H: each wave multiplies its own H tile straight from the
//    accumulator registers into partial (mu, log sigma), the partials meet in LDS behind the barrier
//    the forward pass needs anyway, and d(out) reaches dH from registers and the head-weight product
//    through a private 128-byte transposed buffer.
// Measured history and counters: profiles/r01/07_b3_bf16_notes.md.
#pragma once
#include "mile_bf16_frag.h"
#include "mile_device.h"
#include "mile_grad_generic.h"

template <int NH, int RT>
struct W128Layout {
  static constexpr int IMG = 32 * PIM_STRIDE;                 // one [32 rows][128] bf16 tile image (padded rows, mile_bf16_frag.h)
  static constexpr int WBYTES = 128 * PIM_STRIDE;             // one [128 in][128 out] bf16 weight image
  static constexpr int TILE = 0;                              // per row tile: H_1..H_NH, dZ ping-pong
  static constexpr int HIMG = 0;                              //   offsets inside a tile set
  static constexpr int DZ = NH * IMG;
  static constexpr int TILE_BYTES = (NH + 2) * IMG;
  static constexpr int WIMG = TILE + RT * TILE_BYTES;         // W_2..W_NH
  static constexpr int BIAS = WIMG + (NH - 1) * WBYTES;       // bias tiles in accumulator layout: [NH][4 waves][2 halves][16] fp32
  static constexpr int PART = BIAS + NH * 4 * 2 * 16 * 4;     // head partial sums: [RT][4 waves][32 rows][2] fp32
  static constexpr int DOP = PART + RT * 4 * 32 * 2 * 4;      // per-wave transposed d(out): [4 waves][RT][2][32 rows] bf16
  static constexpr int ZERO = DOP + 4 * RT * 2 * 32 * 2;      // 16 zero bytes
  static constexpr int BYTES = ZERO + 16;
  static_assert(BYTES <= 160 * 1024, "k_grad_w128b: LDS budget");
};

// Workgroup barrier for LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait for the global
// prefetches (X, y of the next tile pair) that are deliberately in flight across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// row_loss_regr with the hardware exp / log / reciprocal (1 ulp-class, ~1e-6 relative): the head sits alone on
// the critical path between two barriers, and next to bf16 operands (2^-9) the difference is invisible.
__device__ __forceinline__ float row_loss_regr_fast(float mu, float sr, float yv, float &dmu, float &ds) {
  const float es = __expf(sr);
  const float sig = fminf(fmaxf(es, 1e-6f), 1e6f);
  const bool unclipped = (es > 1e-6f) && (es < 1e6f);
  const float isig = __frcp_rn(sig);
  const float r = (yv - mu) * isig;
  float ll = -0.5f * r * r - __logf(sig) - 0.91893853320467274f;
  dmu = r * isig;
  ds = unclipped ? (r * r - 1.0f) : 0.0f;
  if (isnan(ll) || isnan(es) || isnan(mu)) { ll = 0.0f; dmu = 0.0f; ds = 0.0f; }
  return ll;
}

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

typedef short s16x2 __attribute__((ext_vector_type(2)));

// H = relu(z) as bf16 into the image: rounding and ReLU commute, and on bf16 bit patterns ReLU is a signed
// 16-bit max with 0 (v_pk_max_i16), two elements per instruction.  dst = image + this lane's store base
// (row pim_row(r), column 32 w + 4 h); group g of 4 features sits 16 g bytes further.
__device__ __forceinline__ void store_tile_relu(char *dst, const f32x16 &z, uint32_t packed[8]) {
  const s16x2 zero = {0, 0};
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const uint32_t b = cvt_pk_bf16(z[4 * g + 2 * k], z[4 * g + 2 * k + 1]);
      packed[2 * g + k] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, b), zero));
    }
    const u32x2_t o = {packed[2 * g], packed[2 * g + 1]};
    *reinterpret_cast<u32x2_t *>(dst + 16 * g) = o;
  }
}

// one 4-feature group of a masked dZ store; hb = this lane's 4 H values (bit patterns) of that group
__device__ __forceinline__ void store_group_masked(char *dst, const f32x4_t v, const f32x2_t hb) {
  const uint32_t ones = 0x00010001u;
  f32x2_t o;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const uint32_t b = cvt_pk_bf16(v[2 * k], v[2 * k + 1]);
    float m, t;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(hb[k]), "v"(ones));
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(t) : "v"(b), "v"(m));
    o[k] = t;
  }
  *reinterpret_cast<f32x2_t *>(dst) = o;
}

template <int NH, int RT, bool TIMING = false>
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
  // per-lane LDS bases; everything else in an address is a compile-time constant (mile_bf16_frag.h, padded images)
  const char *rb = lds + PIM_STRIDE * pim_row(r) + 16 * h;              // row reads of a tile image: + image + 32 s
  char *sb = lds + PIM_STRIDE * pim_row(r) + 64 * w + 8 * h;            // stores / mask reads, this wave's features: + image + 16 g
  const char *tb = lds + pim_tr_base(lane);                             // transposed reads: + image + PIM_STRIDE row0 + 2 col0
  const char *tbw = tb + 64 * w;                                        //   ... of this wave's column block
  const char *wb = rb + PIM_STRIDE * 32 * w;                            // weight rows 32 w + r (backward form): + image + 32 s

  // ---- stage the weights as bf16 images -----------------------------------------------------
#pragma unroll
  for (int li = 1; li < NH; ++li) {
    char *img = lds + LY::WIMG + (li - 1) * LY::WBYTES;
    const float *W = th + sp.w_off[li];
    for (int c = tid; c < 128 * 16; c += 256) {
      const int row = c >> 4, ch = c & 15;
      const float *src = W + row * 128 + ch * 8;
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (bf16)src[j];
      *reinterpret_cast<bf16x8 *>(img + pim_off(row, ch)) = v;
    }
  }
  // head weights as register fragments.  Forward: the wave's own H tile is used straight from the
  // accumulator registers as the B operand of a K = 32 product over its feature block (element j of
  // lane half h of k-step s is feature 16s + 8(j>>2) + 4h + (j&3)), so A[m = k][that feature] is
  // gathered to match; the four waves' partial sums meet in LDS.  Backward: A[m = in][k = out].
  bf16x8 woF[2], woB;
  {
    const float *Wo = th + sp.w_off[NH];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        woF[s2][j] = (bf16)(r < 2 ? Wo[(32 * w + 16 * s2 + 8 * (j >> 2) + 4 * h + (j & 3)) * 2 + r] : 0.0f);
#pragma unroll
    for (int j = 0; j < 8; ++j) woB[j] = (bf16)((h == 0 && j < 2) ? Wo[(32 * w + r) * 2 + j] : 0.0f);
    if (tid < 4) reinterpret_cast<float *>(lds + LY::ZERO)[tid] = 0.0f;
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
  unsigned tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;   // TIMING: cycles per phase (dev builds only; wave-uniform -> SGPRs)
  auto now_cycles = [&]() -> unsigned {   // s_memtime into an SGPR pair: the stamps must not cost vector registers
    unsigned long long tm;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm)::"memory");
    return (unsigned)tm;
  };
  auto tick = [&](int k) {
    if (TIMING) {
      const unsigned now = now_cycles();
      tph[k] += now - tlast;
      tlast = now;
    }
  };
  if (TIMING) tlast = now_cycles();

  const int NBS = p.Npb / (32 * RT);   // super tiles of RT row tiles
  const int nb0 = (int)((long long)sidx * NBS / p.S), nb1 = (int)((long long)(sidx + 1) * NBS / p.S);

  // Global operands are fetched a phase (X: an iteration) ahead of their use: with one wave per SIMD a global
  // load issued where it is needed costs its full ~2000-cycle latency every tile pair.
  bf16x8 xb_next[RT];
#pragma unroll
  for (int q = 0; q < RT; ++q) {
    const int rowc = 32 * (nb0 < nb1 ? nb0 * RT + q : 0);
    xb_next[q] = *reinterpret_cast<const bf16x8 *>(Xb + (size_t)(rowc + r) * 16 + 8 * h);
  }
  for (int t = nb0; t < nb1; ++t) {
    bf16x8 xb_cur[RT], xt_cur[RT][2];
    float y_cur[RT];
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      xb_cur[q] = xb_next[q];
      y_cur[q] = 0.0f;
    }
    // X of the next tile pair and X^T of this one are requested two phases before their use (not at the top:
    // 24 registers held for a whole iteration spill)
    auto prefetch_x = [&]() {
#pragma unroll
      for (int q = 0; q < RT; ++q) {
        const int row0 = 32 * (t * RT + q);
        const int rown = 32 * ((t + 1 < nb1 ? t + 1 : t) * RT + q);
        xb_next[q] = *reinterpret_cast<const bf16x8 *>(Xb + (size_t)(rown + r) * 16 + 8 * h);
#pragma unroll
        for (int s = 0; s < 2; ++s)
          xt_cur[q][s] = *reinterpret_cast<const bf16x8 *>(Xt + (size_t)r * p.Npb + row0 + 16 * s + 8 * h);
      }
    };
    f32x16 acc[RT];
    bf16x8 afp[8];   // weight fragments of the NEXT phase, read before the barrier that precedes it (they do not depend on it)
    // ---- forward ---------------------------------------------------------------------------
#pragma unroll
    for (int l = 0; l < NH; ++l) {
      if (l == 0) {
        const f32x16 b0 = bias_tile(0);
#pragma unroll
        for (int q = 0; q < RT; ++q) {
          acc[q] = mfma_bf16(w1frag, xb_cur[q], b0);
        }
#pragma unroll
        for (int q = 0; q < RT; ++q) {   // targets: needed three phases from here
          const int row0 = 32 * (t * RT + q);
          y_cur[q] = yv[row0 + r < p.N ? row0 + r : 0];
        }
      } else {
        // all fragment reads of the phase are issued before the first MFMA: with one wave per SIMD nothing
        // else hides the LDS latency, and read-then-use pairs would pay it once per MFMA
        bf16x8 bfr[RT][8];
        const f32x16 bl = bias_tile(l);
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
          for (int q = 0; q < RT; ++q)
            bfr[q][s] = *reinterpret_cast<const bf16x8 *>(rb + LY::TILE + q * LY::TILE_BYTES + LY::HIMG + (l - 1) * LY::IMG + 32 * s);
        __builtin_amdgcn_sched_barrier(0);   // keep the reads above, the MFMAs below
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
          for (int q = 0; q < RT; ++q) acc[q] = mfma_bf16(afp[s], bfr[q][s], s == 0 ? bl : acc[q]);
      }
#pragma unroll
      for (int q = 0; q < RT; ++q) {
        uint32_t pk[8];
        store_tile_relu(sb + LY::TILE + q * LY::TILE_BYTES + LY::HIMG + l * LY::IMG, acc[q], pk);
        if (l == NH - 1) {   // head, this wave's 32 of the 128 features: partial (mu, log sigma) per row
          f32x16 zero16, part;
#pragma unroll
          for (int j = 0; j < 16; ++j) zero16[j] = 0.0f;
          const u32x4_t p0 = {pk[0], pk[1], pk[2], pk[3]}, p1 = {pk[4], pk[5], pk[6], pk[7]};
          const bf16x8 b0 = __builtin_bit_cast(bf16x8, p0), b1 = __builtin_bit_cast(bf16x8, p1);
          part = mfma_bf16(woF[0], b0, zero16);
          part = mfma_bf16(woF[1], b1, part);
          if (h == 0) {
            const f32x2_t pv = {part[0], part[1]};
            *reinterpret_cast<f32x2_t *>(lds + LY::PART + ((q * 4 + w) * 32 + r) * 8) = pv;
          }
        }
      }
      if (l + 1 < NH) {   // weights of layer l + 1 (image l), forward form
#pragma unroll
        for (int s = 0; s < 8; ++s) afp[s] = pim_tr_frag(tbw + LY::WIMG + l * LY::WBYTES + PIM_STRIDE * 16 * s);
        __builtin_amdgcn_sched_barrier(0);
      }
      lds_barrier();
      tick(l);
    }
    // ---- head + backward through it: every wave sums the partials and evaluates the likelihood for its
    // lanes' rows; d(out) feeds dH straight from registers, and, transposed through a private 128-byte
    // buffer (no workgroup barrier), the head weight gradient
    {
      f32x16 zero16;
#pragma unroll
      for (int j = 0; j < 16; ++j) zero16[j] = 0.0f;
      f32x2_t pr[RT][4], hm[RT][4];
      bf16x8 ah[RT][2];
#pragma unroll
      for (int q = 0; q < RT; ++q) {
        constexpr int HOFF = LY::TILE + LY::HIMG + (NH - 1) * LY::IMG;
        const int Hin = HOFF + q * LY::TILE_BYTES;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) pr[q][ww] = *reinterpret_cast<const f32x2_t *>(lds + LY::PART + ((q * 4 + ww) * 32 + r) * 8);
#pragma unroll
        for (int g = 0; g < 4; ++g) hm[q][g] = *reinterpret_cast<const f32x2_t *>(sb + Hin + 16 * g);
#pragma unroll
        for (int s = 0; s < 2; ++s) ah[q][s] = pim_tr_frag(tbw + Hin + PIM_STRIDE * 16 * s);
      }
      if (NH == 1) prefetch_x();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < RT; ++q) {
        const int row0 = 32 * (t * RT + q);
        const float mu = ((pr[q][0][0] + pr[q][1][0]) + (pr[q][2][0] + pr[q][3][0])) + bo0;
        const float sr = ((pr[q][0][1] + pr[q][1][1]) + (pr[q][2][1] + pr[q][3][1])) + bo1;
        float dmu = 0.0f, dsg = 0.0f;
        if (h == 0 && row0 + r < p.N) {
          const float ll = row_loss_regr_fast(mu, sr, y_cur[q], dmu, dsg);
          ll_acc += ll;
        }
        bf16x8 bdo;
#pragma unroll
        for (int j = 0; j < 8; ++j) bdo[j] = (bf16)0.0f;
        bdo[0] = (bf16)dmu; bdo[1] = (bf16)dsg;          // lanes h = 1 and rows >= N carry zeros
        acc[q] = mfma_bf16(woB, bdo, zero16);
        char *dop = lds + LY::DOP + (w * RT + q) * 128;
        if (h == 0) {
          reinterpret_cast<bf16 *>(dop)[r] = bdo[0];
          reinterpret_cast<bf16 *>(dop)[32 + r] = bdo[1];
        }
      }
#pragma unroll
      for (int q = 0; q < RT; ++q) {
        const char *dop = lds + LY::DOP + (w * RT + q) * 128;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const char *src = r < 2 ? dop + r * 64 + 32 * s + 16 * h : lds + LY::ZERO;
          const bf16x8 bq = *reinterpret_cast<const bf16x8 *>(src);
          dWo = mfma_bf16(ah[q][s], bq, dWo);
          dbo = bf16_colsum(bq, dbo);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4_t v = {acc[q][4 * g], acc[q][4 * g + 1], acc[q][4 * g + 2], acc[q][4 * g + 3]};
          store_group_masked(sb + LY::TILE + q * LY::TILE_BYTES + LY::DZ + 16 * g, v, hm[q][g]);
        }
      }
      if (NH >= 2) {   // weights of the last hidden layer, backward form
#pragma unroll
        for (int s = 0; s < 8; ++s) afp[s] = *reinterpret_cast<const bf16x8 *>(wb + LY::WIMG + (NH - 2) * LY::WBYTES + 32 * s);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    lds_barrier();
    tick(3);
    // ---- backward: hidden layers NH .. 2 ------------------------------------------------------------
    // Pinned order (sched_barrier fences): all dH operands are read first; the transposed dW operands
    // are read between the dH MFMAs as those release registers; the dZ epilogue of the dH result runs
    // between the dW MFMAs, which do not depend on it.
    int pp = 0;
#pragma unroll
    for (int l = NH - 1; l >= 1; --l) {   // dZ of layer l is in DZ[pp]; its input is H_l (image l-1)
      f32x16 zero16;
#pragma unroll
      for (int j = 0; j < 16; ++j) zero16[j] = 0.0f;
      bf16x8 bfr[RT][8], bq[RT][2], ah[RT][2][4];
      f32x2_t hm[RT][4];
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int q = 0; q < RT; ++q)
          bfr[q][s] = *reinterpret_cast<const bf16x8 *>(rb + LY::TILE + q * LY::TILE_BYTES + LY::DZ + pp * LY::IMG + 32 * s);
#pragma unroll
      for (int q = 0; q < RT; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          hm[q][g] = *reinterpret_cast<const f32x2_t *>(sb + LY::TILE + q * LY::TILE_BYTES + LY::HIMG + (l - 1) * LY::IMG + 16 * g);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int q = 0; q < RT; ++q) acc[q] = mfma_bf16(afp[s], bfr[q][s], s == 0 ? zero16 : acc[q]);
        {
          const int q = (s * RT) / 8, sub = RT == 2 ? (s & 3) : (s >> 1);   // RT = 1: four steps carry reads
          const int ts = LY::TILE + q * LY::TILE_BYTES;
          const int dz = ts + LY::DZ + pp * LY::IMG, Hin = ts + LY::HIMG + (l - 1) * LY::IMG, R16 = PIM_STRIDE * 16;
          if (RT == 2 || (s & 1) == 0) {
            if (sub == 0) { bq[q][0] = pim_tr_frag(tbw + dz); ah[q][0][0] = pim_tr_frag(tb + Hin); ah[q][0][1] = pim_tr_frag(tb + Hin + 64); }
            if (sub == 1) { ah[q][0][2] = pim_tr_frag(tb + Hin + 128); ah[q][0][3] = pim_tr_frag(tb + Hin + 192); bq[q][1] = pim_tr_frag(tbw + dz + R16); }
            if (sub == 2) { ah[q][1][0] = pim_tr_frag(tb + Hin + R16); ah[q][1][1] = pim_tr_frag(tb + Hin + R16 + 64); }
            if (sub == 3) { ah[q][1][2] = pim_tr_frag(tb + Hin + R16 + 128); ah[q][1][3] = pim_tr_frag(tb + Hin + R16 + 192); }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (l == 1) {   // the dH operands are consumed: their registers take the global prefetch
        prefetch_x();
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < RT; ++q) {
        char *dzo = sb + LY::TILE + q * LY::TILE_BYTES + LY::DZ + (pp ^ 1) * LY::IMG;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int ib = 0; ib < 4; ++ib) {
            dW[l - 1][ib] = mfma_bf16(ah[q][s][ib], bq[q][s], dW[l - 1][ib]);
            if (ib & 1) {
              const int g = 2 * s + (ib >> 1);
              const f32x4_t v = {acc[q][4 * g], acc[q][4 * g + 1], acc[q][4 * g + 2], acc[q][4 * g + 3]};
              store_group_masked(dzo + 16 * g, v, hm[q][g]);
            }
          }
          db[l] = bf16_colsum(bq[q][s], db[l]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      pp ^= 1;
      if (l >= 2) {   // weights of the next layer down, backward form
#pragma unroll
        for (int s = 0; s < 8; ++s) afp[s] = *reinterpret_cast<const bf16x8 *>(wb + LY::WIMG + (l - 2) * LY::WBYTES + 32 * s);
        __builtin_amdgcn_sched_barrier(0);
      }
      lds_barrier();
      tick(3 + l);
    }
    // ---- backward: first layer ------------------------------------------------------------------------
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      const char *dz = tbw + LY::TILE + q * LY::TILE_BYTES + LY::DZ + pp * LY::IMG;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 bq = pim_tr_frag(dz + PIM_STRIDE * 16 * s);
        dW1 = mfma_bf16(xt_cur[q][s], bq, dW1);
        db[0] = bf16_colsum(bq, db[0]);
      }
    }
    tick(6);
  }
  if (TIMING && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && p.dbg_buf) {
#pragma unroll
    for (int k = 0; k < 7; ++k) p.dbg_buf[k] = tph[k] / (unsigned)(nb1 - nb0 > 0 ? nb1 - nb0 : 1);
    p.dbg_buf[7] = nb1 - nb0;
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
