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
// 128 x 128 fp32 gradient per layer does not fit one wave).  Tiles are exchanged through padded
// [32][128] bf16 LDS images (mile_bf16_frag.h: 272-byte rows, 16-row groups permuted) that serve both the
// row reads (forward / dH) and the transposed reads (dW contracts over rows) without a second copy and
// without bank conflicts, every address one of five per-lane bases plus an immediate; the bf16 weight
// images serve forward (transposed read) and backward (row read) the same way.
//
// One wave per SIMD leaves nobody to hide a wave's own LDS latency, MFMA drain and barrier waits.  The matrix
// products themselves are hidden already (removing every MFMA buys 3 %: profiles/r03/18_w128b_lab_ablation.md);
// what the kernel is made of is operand traffic, epilogue VALU and barrier skew, so
//  * each iteration walks RT = 2 row tiles between the same 2*NH - 1 barriers (2*NH with one hidden layer); they share
//    every weight fragment, and the barrier cost per tile halves;
//  * a phase issues only the reads that wait for its barrier (tile 0's dH / forward operand) before its first
//    MFMA, tile 1's in the gaps of tile 0's products; everything that does NOT depend on the barrier -- the
//    weights, the transposed H of the layer below, this wave's own dZ columns and ReLU masks -- is read in the
//    gaps of the PREVIOUS phase's products, into the fragment registers those products have just consumed
//    (sched_barrier fences pin the order: left alone, the compiler sinks every read next to its use and pays
//    the LDS latency once per MFMA);
//  * tile 0's epilogue runs in the gaps of tile 1's products, tile 1's in the gaps of the dW products;
//  * epilogues are packed: the bias tile is the first MFMA's C operand, one v_cvt_pk_bf16_f32 per pair, ReLU
//    and its derivative are 16-bit integer ops on bf16 bit patterns, bias gradients are v_dot2c_f32_bf16
//    column sums of the transposed dZ fragments the dW products read anyway;
//  * the head never touches a full H: each wave multiplies its own H tile straight from the
//    accumulator registers into partial (mu, log sigma), the partials meet in LDS behind the barrier
//    the forward pass needs anyway; lane half h evaluates the likelihood of tile h (one pass for both tiles),
//    d(out) reaches dH from registers and the head-weight product through a private 128-byte transposed buffer;
//  * global operands (X, X^T, targets) are requested a whole tile pair ahead; with hidden->hidden layers the first
//    layer of the NEXT tile pair runs inside this pair's last backward phase (its H_1 images are dead by then), and a
//    pair's first-layer weight gradient is taken right behind that phase's barrier -- one phase and one barrier less.
// Measured history and counters: profiles/r01/07_b3_bf16_notes.md (round 1), profiles/r03/16_* (round 2 kernel),
// profiles/r03/17_b3_*, 18_w128b_lab_ablation.md (this form: 5.78 -> 4.2 ms at B3).
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
#ifdef MILE_LAB_NO_BARRIER   // dev experiment (tools/r03/lab): wrong results, shows what the synchronisation costs
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#else
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

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

// H = relu(z) as bf16 into the image, one 4-feature group: rounding and ReLU commute, and on bf16 bit patterns ReLU
// is a signed 16-bit max with 0 (v_pk_max_i16), two elements per instruction.  dst = image + this lane's store base
// (row pim_row(r), column 32 w + 4 h) + 16 g.
__device__ __forceinline__ void store_group_relu(char *dst, float z0, float z1, float z2, float z3, uint32_t &p0, uint32_t &p1) {
  const s16x2 zero = {0, 0};
  p0 = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, cvt_pk_bf16(z0, z1)), zero));
  p1 = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, cvt_pk_bf16(z2, z3)), zero));
  const u32x2_t o = {p0, p1};
#ifdef MILE_LAB_NO_STORE   // dev experiment (tools/r03/lab)
  if (p0 != 0x12345u) return;
#endif
  *reinterpret_cast<u32x2_t *>(dst) = o;
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
#ifdef MILE_LAB_NO_STORE
  if (o[0] != 123.0f) return;
#endif
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
    for (int j = 0; j < 8; ++j) woB[j] = (bf16)(j < 2 ? Wo[(32 * w + r) * 2 + j] : 0.0f);   // both lane halves: see the head
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

  // Global operands are fetched a whole tile pair ahead of their use, right after the registers' previous contents are
  // consumed: with one wave per SIMD nothing hides a global load, and these took > 2000 cycles to land even from L2
  // (measured: requested one phase ahead, the consumer still waited ~1000 cycles per pair).
  static_assert(RT == 2, "the head maps row tile q to lane half h = q");
  constexpr bool MERGE = NH >= 2;   // the first layer of the NEXT pair runs inside this pair's last backward phase
  constexpr int PPF = (NH & 1) ? 0 : 1;   // the dZ buffer the last hidden backward layer leaves dZ_1 in
  static_assert(PPF == ((NH - 1) & 1), "dZ_1 buffer parity");
  const int t_end = nb1 - 1;
  bf16x8 xb_next[RT];      // X of the pair whose first layer comes next
  bf16x8 xt_cur[RT][2];    // X^T and ...
  bf16x8 bz[RT][2];        // ... this wave's own dZ_1 columns (transposed) of the pair whose first-layer weight gradient is owed
  float y_next;            // lane (r, h): target of row r of tile h
  auto load_xb = [&](int tt) {
#pragma unroll
    for (int q = 0; q < RT; ++q) xb_next[q] = *reinterpret_cast<const bf16x8 *>(Xb + (size_t)(32 * (tt * RT + q) + r) * 16 + 8 * h);
  };
  auto load_xt = [&](int tt) {
#pragma unroll
    for (int q = 0; q < RT; ++q)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        xt_cur[q][s] = *reinterpret_cast<const bf16x8 *>(Xt + (size_t)r * p.Npb + 32 * (tt * RT + q) + 16 * s + 8 * h);
  };
  auto load_y = [&](int tt) {
    const int rowg = 32 * (tt * RT + h) + r;
    y_next = yv[rowg < p.N ? rowg : 0];
  };
  {
    const u32x4_t z4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int q = 0; q < RT; ++q)
#pragma unroll
      for (int s = 0; s < 2; ++s) { xt_cur[q][s] = __builtin_bit_cast(bf16x8, z4); bz[q][s] = __builtin_bit_cast(bf16x8, z4); }
  }
  load_xb(nb0 < nb1 ? nb0 : 0);
  load_y(nb0 < nb1 ? nb0 : 0);
  // First-layer weight gradient of a finished pair: X^T from global memory times this wave's own dZ_1 columns (no other
  // wave's data).  Owed until the next pair: the dZ_1 fragments are read at the end of the last backward phase (before its
  // barrier), the four products run right behind that barrier while the next phase's operands are in flight -- done where
  // the data is produced it cost ~1000 cycles per pair waiting for X^T and the LDS round trip.
  auto dz1_reads = [&]() {
#pragma unroll
    for (int q = 0; q < RT; ++q)
#pragma unroll
      for (int s = 0; s < 2; ++s) bz[q][s] = pim_tr_frag(tbw + LY::TILE + q * LY::TILE_BYTES + LY::DZ + PPF * LY::IMG + PIM_STRIDE * 16 * s);
  };
  auto first_layer_products = [&]() {
#pragma unroll
    for (int q = 0; q < RT; ++q)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        dW1 = mfma_bf16(xt_cur[q][s], bz[q][s], dW1);
        db[0] = bf16_colsum(bz[q][s], db[0]);
      }
  };
  f32x16 acc[RT], accf[RT], zero16;
#pragma unroll
  for (int j = 0; j < 16; ++j) zero16[j] = 0.0f;
  // Operands that do not depend on the barrier in front of the phase that uses them are read BEFORE that barrier,
  // among the previous phase's MFMAs, into the fragment registers those MFMAs have just consumed:
  bf16x8 afp[8];           // weight fragments of the next phase
  bf16x8 ahh[RT][2];       // head: this wave's own columns of H_NH, transposed (own data)
  f32x2_t hmh[RT][4];      // head: this lane's own H_NH values (ReLU mask), kept from the forward epilogue
  bf16x8 ah[RT][2][4];     // backward layer: H of the layer below, transposed (written in the forward pass)
  bf16x8 bq[RT][2];        // backward layer: this wave's own dZ columns, transposed (own data)
  f32x2_t hm[RT][4];       // backward layer: this lane's own H values of the layer below (ReLU mask)
  uint32_t pk[RT][8];
  auto relu_group = [&](const f32x16 (&z)[RT], int l, int q, int g) {   // H_l = relu(z): one 4-feature group of tile q into the image
#ifdef MILE_LAB_NO_EPI
    pk[q][2 * g] = __float_as_uint(z[q][4 * g]); pk[q][2 * g + 1] = __float_as_uint(z[q][4 * g + 1]);
    return;
#endif
    store_group_relu(sb + LY::TILE + q * LY::TILE_BYTES + LY::HIMG + l * LY::IMG + 16 * g, z[q][4 * g], z[q][4 * g + 1],
                     z[q][4 * g + 2], z[q][4 * g + 3], pk[q][2 * g], pk[q][2 * g + 1]);
  };
  auto head_partial = [&](int q) {   // this wave's 32 of the 128 features: partial (mu, log sigma) per row of tile q
    const u32x4_t p0 = {pk[q][0], pk[q][1], pk[q][2], pk[q][3]}, p1 = {pk[q][4], pk[q][5], pk[q][6], pk[q][7]};
    f32x16 part = mfma_bf16(woF[0], __builtin_bit_cast(bf16x8, p0), zero16);
    part = mfma_bf16(woF[1], __builtin_bit_cast(bf16x8, p1), part);
    if (h == 0) {
      const f32x2_t pv = {part[0], part[1]};
      *reinterpret_cast<f32x2_t *>(lds + LY::PART + ((q * 4 + w) * 32 + r) * 8) = pv;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x2_t m = {pk[q][2 * g], pk[q][2 * g + 1]};
      hmh[q][g] = __builtin_bit_cast(f32x2_t, m);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
      ahh[q][s] = pim_tr_frag(tbw + LY::TILE + q * LY::TILE_BYTES + LY::HIMG + (NH - 1) * LY::IMG + PIM_STRIDE * 16 * s);
  };
  auto dout_frag = [&](int q, int s) {   // d(out) of tile q, rows 16 s .., transposed: B[k = row][n = output] from the private buffer
    const char *dop = lds + LY::DOP + (w * RT + q) * 128;
    const char *src = r < 2 ? dop + r * 64 + 32 * s + 16 * h : lds + LY::ZERO;
    return *reinterpret_cast<const bf16x8 *>(src);
  };
  auto masked_group = [&](int q, int g, int ppo, const f32x2_t m) {   // dZ = dH * (H > 0): one group of tile q
#ifdef MILE_LAB_NO_EPI
    if (acc[q][4 * g] == 123.0f) *reinterpret_cast<f32x2_t *>(sb) = m;
    return;
#endif
    const f32x4_t v = {acc[q][4 * g], acc[q][4 * g + 1], acc[q][4 * g + 2], acc[q][4 * g + 3]};
    store_group_masked(sb + LY::TILE + q * LY::TILE_BYTES + LY::DZ + ppo * LY::IMG + 16 * g, v, m);
  };
  if (MERGE && nb0 < nb1) {   // H_1 of the first pair; every later pair's is formed inside its predecessor's last backward phase
    const f32x16 b0 = bias_tile(0);
#pragma unroll
    for (int q = 0; q < RT; ++q) accf[q] = mfma_bf16(w1frag, xb_next[q], b0);
#pragma unroll
    for (int s = 0; s < 8; ++s) afp[s] = pim_tr_frag(tbw + LY::WIMG + PIM_STRIDE * 16 * s);
    load_xb(nb0 < t_end ? nb0 + 1 : t_end);
#pragma unroll
    for (int q = 0; q < RT; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g) relu_group(accf, 0, q, g);
    lds_barrier();
  }
  for (int t = nb0; t < nb1; ++t) {
    const float y_cur = y_next;
    const int tn = t < t_end ? t + 1 : t_end, tnn = t + 2 <= t_end ? t + 2 : t_end;
    // ---- forward ---------------------------------------------------------------------------
    // Tile 0's products first, then tile 1's with tile 0's epilogue (and the next phase's weight reads) in their gaps.
#pragma unroll
    for (int l = 0; l < NH; ++l) {
      if (l == 0) {
        if (MERGE) continue;   // done a pair ago
        // one hidden layer: the first layer is the whole forward pass
        const f32x16 b0 = bias_tile(0);
#pragma unroll
        for (int q = 0; q < RT; ++q) acc[q] = mfma_bf16(w1frag, xb_next[q], b0);
        if (t > nb0) {
          dz1_reads();
          __builtin_amdgcn_sched_barrier(0);
          first_layer_products();
        }
        __builtin_amdgcn_sched_barrier(0);
        load_xb(tn);   // into the registers just consumed
        load_xt(t);
        __builtin_amdgcn_sched_barrier(0);
        tick(6);
#pragma unroll
        for (int q = 0; q < RT; ++q) {
#pragma unroll
          for (int g = 0; g < 4; ++g) relu_group(acc, 0, q, g);
          head_partial(q);
        }
      } else {
        bf16x8 bfr[RT][8];
        const f32x16 bl = bias_tile(l);
#pragma unroll
        for (int s = 0; s < 8; ++s) bfr[0][s] = pim_row_read(rb + LY::TILE + LY::HIMG + (l - 1) * LY::IMG + 32 * s);
        __builtin_amdgcn_sched_barrier(0);   // keep the reads above, the MFMAs below
        if (MERGE && l == 1) {   // the previous pair's first-layer weight gradient (operands in registers) covers the wait
          first_layer_products();
          __builtin_amdgcn_sched_barrier(0);
          load_xt(t);            // into the registers just consumed
          __builtin_amdgcn_sched_barrier(0);
          tick(6);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {   // tile 1's operands are requested in the gaps of tile 0's products
          acc[0] = mfma_bf16(afp[s], bfr[0][s], s == 0 ? bl : acc[0]);
          bfr[1][s] = pim_row_read(rb + LY::TILE + LY::TILE_BYTES + LY::HIMG + (l - 1) * LY::IMG + 32 * s);
          if (NH >= 2 && l == NH - 1) {   // operands of the first backward layer that are already final: H_{NH-1}, transposed, tile 0
            ah[0][s >> 2][s & 3] = pim_tr_frag(tb + LY::TILE + LY::HIMG + (NH - 2) * LY::IMG + PIM_STRIDE * 16 * (s >> 2) + 64 * (s & 3));
            hm[s >> 2][s & 3] = *reinterpret_cast<const f32x2_t *>(sb + LY::TILE + (s >> 2) * LY::TILE_BYTES + LY::HIMG + (NH - 2) * LY::IMG + 16 * (s & 3));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          acc[1] = mfma_bf16(afp[s], bfr[1][s], s == 0 ? bl : acc[1]);
          if (l + 1 < NH) afp[s] = pim_tr_frag(tbw + LY::WIMG + l * LY::WBYTES + PIM_STRIDE * 16 * s);                 // layer l + 1, forward form
          else if (NH >= 2) afp[s] = pim_row_read(wb + LY::WIMG + (NH - 2) * LY::WBYTES + 32 * s);   // last hidden layer, backward form
          if (s >= 2 && s < 6) relu_group(acc, l, 0, s - 2);
          if (NH >= 2 && l == NH - 1)   // ... tile 1
            ah[1][s >> 2][s & 3] = pim_tr_frag(tb + LY::TILE + LY::TILE_BYTES + LY::HIMG + (NH - 2) * LY::IMG + PIM_STRIDE * 16 * (s >> 2) + 64 * (s & 3));
          __builtin_amdgcn_sched_barrier(0);
        }
        if (l == NH - 1) head_partial(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) relu_group(acc, l, 1, g);
        if (l == NH - 1) head_partial(1);
      }
      lds_barrier();
      tick(l);
    }
    // ---- head + backward through it: lane (r, h) evaluates row r of tile q = h (one pass of the likelihood for both
    // tiles); d(out) feeds dH straight from registers, and, transposed through a private 128-byte buffer (no workgroup
    // barrier), the head weight gradient
    {
      f32x2_t pr[4];
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) pr[ww] = *reinterpret_cast<const f32x2_t *>(lds + LY::PART + ((h * 4 + ww) * 32 + r) * 8);
      __builtin_amdgcn_sched_barrier(0);
      {
        const int rowg = 32 * (t * RT + h) + r;
        const float mu = ((pr[0][0] + pr[1][0]) + (pr[2][0] + pr[3][0])) + bo0;
        const float sr = ((pr[0][1] + pr[1][1]) + (pr[2][1] + pr[3][1])) + bo1;
        float dmu = 0.0f, dsg = 0.0f;
        if (rowg < p.N) {
          const float ll = row_loss_regr_fast(mu, sr, y_cur, dmu, dsg);
          ll_acc += ll;
        }
        load_y(tn);   // the next pair's targets, a pair ahead
        const uint32_t dpk = cvt_pk_bf16(dmu, dsg);   // rows >= N carry zeros
        // d(out) of tile q sits in lane half h = q.  woB holds the two output columns in the first two k-slots of BOTH
        // halves, so tile q's product takes them from k-slots 8q, 8q + 1 with the other half's operand zeroed.
#pragma unroll
        for (int q = 0; q < RT; ++q) {
          const u32x4_t bv = {h == q ? dpk : 0u, 0u, 0u, 0u};
          acc[q] = mfma_bf16(woB, __builtin_bit_cast(bf16x8, bv), zero16);
        }
        uint16_t *dop = reinterpret_cast<uint16_t *>(lds + LY::DOP + (w * RT + h) * 128);
        dop[r] = (uint16_t)(dpk & 0xffffu);
        dop[32 + r] = (uint16_t)(dpk >> 16);
      }
#pragma unroll
      for (int q = 0; q < RT; ++q) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const bf16x8 bo = dout_frag(q, s);
          dWo = mfma_bf16(ahh[q][s], bo, dWo);
          dbo = bf16_colsum(bo, dbo);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) masked_group(q, g, 0, hmh[q][g]);
      }
      if (NH >= 2) {   // this wave's own dZ columns, just stored
#pragma unroll
        for (int q = 0; q < RT; ++q)
#pragma unroll
          for (int s = 0; s < 2; ++s) bq[q][s] = pim_tr_frag(tbw + LY::TILE + q * LY::TILE_BYTES + LY::DZ + PIM_STRIDE * 16 * s);
      }
    }
    lds_barrier();
    tick(3);
    // ---- backward: hidden layers NH .. 2 ------------------------------------------------------------
    // Entered with afp (weights, backward form), ah, bq and hm of this layer in registers; only the dH operand (all
    // waves' dZ columns) waits for the barrier.  Pinned order (sched_barrier fences): tile 0's dH products, tile 1's with
    // tile 0's dZ epilogue and the next layer's weight reads in the gaps, then the dW products with tile 1's epilogue
    // and the next layer's transposed H reads in the gaps.  The LAST backward phase has no next layer to read for: its
    // dW gaps carry the whole first layer of the NEXT tile pair instead (two products from X in registers, the ReLU
    // epilogue into the H_1 images -- dead since their transposed reads a phase ago -- and the second layer's forward
    // weight fragments), which removes one phase and one barrier per pair.
    int pp = 0;
#pragma unroll
    for (int l = NH - 1; l >= 1; --l) {   // dZ of layer l is in DZ[pp]; its input is H_l (image l-1)
      bf16x8 bfr[RT][8];
      f32x16 b0f;
#pragma unroll
      for (int s = 0; s < 8; ++s) bfr[0][s] = pim_row_read(rb + LY::TILE + LY::DZ + pp * LY::IMG + 32 * s);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        acc[0] = mfma_bf16(afp[s], bfr[0][s], s == 0 ? zero16 : acc[0]);
        bfr[1][s] = pim_row_read(rb + LY::TILE + LY::TILE_BYTES + LY::DZ + pp * LY::IMG + 32 * s);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        acc[1] = mfma_bf16(afp[s], bfr[1][s], s == 0 ? zero16 : acc[1]);
        if (l >= 2) afp[s] = pim_row_read(wb + LY::WIMG + (l - 2) * LY::WBYTES + 32 * s);
        if (s >= 2 && s < 6) masked_group(0, s - 2, pp ^ 1, hm[0][s - 2]);
        if (l == 1 && s == 7) b0f = bias_tile(0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < RT; ++q)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int ib = 0; ib < 4; ++ib) {
            const int k = (q * 2 + s) * 4 + ib;
            dW[l - 1][ib] = mfma_bf16(ah[q][s][ib], bq[q][s], dW[l - 1][ib]);
            if (l >= 2)
              ah[q][s][ib] = pim_tr_frag(tb + LY::TILE + q * LY::TILE_BYTES + LY::HIMG + (l - 2) * LY::IMG + PIM_STRIDE * 16 * s + 64 * ib);
            if (k >= 2 && k < 6) masked_group(1, k - 2, pp ^ 1, hm[1][k - 2]);
            if (ib == 3) db[l] = bf16_colsum(bq[q][s], db[l]);
            if (l == 1) {   // the next pair's first layer (MERGE is implied: l >= 1 needs NH >= 2)
              if (k == 4 || k == 5) accf[k - 4] = mfma_bf16(w1frag, xb_next[k - 4], b0f);
              if (k == 6) load_xb(tnn);
              if (k >= 8) relu_group(accf, 0, (k - 8) >> 2, (k - 8) & 3);
              if (k < 8) afp[k] = pim_tr_frag(tbw + LY::WIMG + PIM_STRIDE * 16 * k);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      pp ^= 1;
      if (l >= 2) {   // the next layer's mask and this wave's own dZ columns, just stored
#pragma unroll
        for (int q = 0; q < RT; ++q) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            hm[q][g] = *reinterpret_cast<const f32x2_t *>(sb + LY::TILE + q * LY::TILE_BYTES + LY::HIMG + (l - 2) * LY::IMG + 16 * g);
#pragma unroll
          for (int s = 0; s < 2; ++s) bq[q][s] = pim_tr_frag(tbw + LY::TILE + q * LY::TILE_BYTES + LY::DZ + pp * LY::IMG + PIM_STRIDE * 16 * s);
        }
      } else {
        dz1_reads();   // own dZ_1 columns for the first-layer weight gradient, taken behind the barrier
      }
      lds_barrier();
      tick(3 + l);
    }
  }
  if (nb1 > nb0) {   // the last pair's first-layer weight gradient
    if (!MERGE) {
      dz1_reads();
      __builtin_amdgcn_sched_barrier(0);
    }
    first_layer_products();
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
