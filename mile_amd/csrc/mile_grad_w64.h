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
//
// SPLIT = true (mfma_w64_bf16x3, what AUTO selects for 2-3 hidden layers): the same kernel with every
// hidden->hidden product (forward, dH, dW) as six v_mfma_f32_32x32x16_bf16 products of exact three-term bf16
// splits of the fp32 operands -- fp32-faithful, 8 + 288 MFMAs = 9.7k instead of 25.1k matrix-pipe cycles per block.
// See the comments at split3_pk / W64_PIPE_A / W64Layout below and DESIGN.md section 3.1b.
#pragma once
#include <type_traits>

#include "mile_bf16_frag.h"
#include "mile_device.h"
#include "mile_grad_generic.h"
#include "mile_update.h"

// Fused integrator epilogue of the SPLIT kernel: when `enabled`, the workgroup of a particle that
// finishes LAST runs everything between this gradient and the next one (upd_fast_body: the B / O chain, record point, tuner,
// position update) right here instead of in a separate k_update_fast launch.  Hand-off of the S partial-gradient slabs =
// the counter form of cdna_hip_programming.md section 6 Guideline 16, placement-independent: every slab / llpart byte is
// stored sc1 (write-through), every storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier, one lane draws a ticket with
// a relaxed agent-scope fetch_add; the workgroup that draws S-1 reads all slabs with sc1 loads (ld4_slab<.., COH = true>).
struct W64Fuse {
  UpdParams upd;
  int32_t *arrive;    // [E] tickets, zero before the launch; the last arriver re-zeroes its particle's word
  int32_t enabled;
  int32_t kind;       // upd_kind(upd): UPD_KIND_MID / UPD_KIND_REC select the compile-time-specialised body, -1 the general one
};
// Kernels built without the epilogue take this instead (the fp32-MFMA form, and F > 8: with the epilogue in -- or even just
// the larger argument block -- hipcc 7.2 crashes in its AGPR-copy rewrite on the heavily spilling k_grad_w64<3,2,true>).
struct W64NoFuse {
  int32_t enabled;
};
// MEASURED AND NOT ENABLED BY DEFAULT (build with MILE_HIPCC_FLAGS=-DMILE_W64_EPILOGUE to compile it in): on B2 the
// epilogue costs 8.6 us (mid-step) / 15.8 us (record point) inside the grad launch -- 4.7 us of it the last arriver's sc1
// reads of 140 KB with four waves, and its Philox chains run at one wave per SIMD -- against 14.8 us for BOTH stand-alone
// k_update_fast launches (768 threads per particle, boundaries included): 154.1 vs 150.6 us per MCLMC step
// (profiles/r02/03_fused_epilogue_experiment.md).
#ifdef MILE_W64_EPILOGUE
#define MILE_W64_EPILOGUE_ON 1
#else
#define MILE_W64_EPILOGUE_ON 0
#endif
template <int FQ, bool SPLIT>
struct W64FuseArg {
  static constexpr bool FUSABLE = MILE_W64_EPILOGUE_ON && SPLIT && FQ == 1;
  using type = std::conditional_t<FUSABLE, W64Fuse, W64NoFuse>;
};

#define W64_RS 68  // row stride (floats) of every padded LDS image
#ifndef W64_OVERLAP
#define W64_OVERLAP 1   // SPLIT main loop: form the next backward layer's dZ (mask + split) inside the dW MFMA groups
#endif

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

// ---- SPLIT variant: fp32 operands as exact sums of three bf16 terms ------------------------------------------
// x = x1 + x2 + x3 with x1 = the top 8 significant bits of x, x2 the next 8, x3 the last 8 (each a bf16, the
// subtractions are exact), so a fp32 product sum a.b becomes the six bf16 MFMA products
//   a3.b1 + a1.b3 + a2.b2 + a2.b1 + a1.b2 + a1.b1        (accumulated in fp32, small terms first)
// -- every partial product is exact in fp32; the dropped terms a2.b3, a3.b2, a3.b3 are below 2^-23 of a1.b1, the
// size of one fp32 rounding.  v_mfma_f32_32x32x16_bf16 does 8x the work of v_mfma_f32_32x32x2_f32 in half the
// cycles, so the six products cost 3/8 of the fp32 MFMA time.
__device__ __forceinline__ uint32_t hi16_pair(float x1, float x0) {   // {bf16 bits of x1 : bf16 bits of x0}, truncating
  return __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

// One 32x32 T-layout tile (lane = data row j, reg r = feature tfeat(r,h)) -> packed bf16 pairs of its three terms:
// pk[t][q] = {term t of reg 2q+1 : term t of reg 2q}.
// MILE_SPLIT_DOT2 (experiment, OFF): the residual x - top(x) straight from the PACKED term with one v_dot2c_f32_bf16 per
// element (x + pk.lo * -1 + pk.hi * 0) instead of v_and + half a v_pk_add, 7 VALU per pair instead of 9.  Not usable: on
// gfx950 the instruction does not return the exact difference (round 2: log-posterior off by 1e-3 on the golden vectors).
#ifndef MILE_SPLIT_DOT2
#define MILE_SPLIT_DOT2 0
#endif
__device__ __forceinline__ float sub_lo(uint32_t pk, float x) {   // x - (low bf16 of pk)
  const bf16x2 m = {(bf16)-1.0f, (bf16)0.0f};
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pk), m, x, false);
}
__device__ __forceinline__ float sub_hi(uint32_t pk, float x) {   // x - (high bf16 of pk)
  const bf16x2 m = {(bf16)0.0f, (bf16)-1.0f};
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pk), m, x, false);
}
__device__ __forceinline__ void split3_pk(const f32x16 &T, uint32_t (&pk)[3][8]) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x2 x = {T[2 * q], T[2 * q + 1]};
    pk[0][q] = hi16_pair(x[1], x[0]);
#if MILE_SPLIT_DOT2
    const f32x2 r = {sub_lo(pk[0][q], x[0]), sub_hi(pk[0][q], x[1])};
    pk[1][q] = hi16_pair(r[1], r[0]);
    const f32x2 s = {sub_lo(pk[1][q], r[0]), sub_hi(pk[1][q], r[1])};
#else
    const f32x2 r = x - f32x2{trunc_bf16(x[0]), trunc_bf16(x[1])};
    pk[1][q] = hi16_pair(r[1], r[0]);
    const f32x2 s = r - f32x2{trunc_bf16(r[0]), trunc_bf16(r[1])};
#endif
    pk[2][q] = hi16_pair(s[1], s[0]);
  }
}

// -> the B operands of v_mfma_f32_32x32x16_bf16 for the tile's two 16-feature K chunks, in NATURAL feature order
// (element i of lane (j,h) = feature 16c + 8h + i).  The lane halves trade four features per chunk (v_permlane32_swap).
__device__ __forceinline__ void natk_from_pk(const uint32_t (&pk)[3][8], bf16x8 (&fr)[3][2]) {
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      // regs 8c..8c+7 = features 16c + {4h..4h+3, 8+4h..8+4h+3}: half 1's first quad <-> half 0's second quad
      const auto s0 = __builtin_amdgcn_permlane32_swap(pk[t][4 * c], pk[t][4 * c + 2], false, false);
      const auto s1 = __builtin_amdgcn_permlane32_swap(pk[t][4 * c + 1], pk[t][4 * c + 3], false, false);
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
      fr[t][c] = __builtin_bit_cast(bf16x8, v);
    }
}

__device__ __forceinline__ void split3_natk(const f32x16 &T, bf16x8 (&fr)[3][2]) {
  uint32_t pk[3][8];
  split3_pk(T, pk);
  natk_from_pk(pk, fr);
}

// The wave's private transposition buffer in the SPLIT variant: two padded bf16 images of the block's 32 rows
// (mile_bf16_frag.h: every address = one per-lane base + an immediate).  Image 0, [32][128] at 272-byte rows: term 0 of a
// 64-feature row in columns 0..63, term 1 in columns 64..127.  Image 1, [32][64] at 144-byte rows: term 2.
#define W64_S2 144                                  // row pitch of image 1
#define W64_SIMG0 (32 * PIM_STRIDE)                 // bytes of image 0 = offset of image 1
#define W64_SIMG (W64_SIMG0 + 32 * W64_S2)          // both images
#define W64_TIMG (64 * PIM_STRIDE)                  // one term image of a pair of weight layers: W^T[64 out][2 x 64 in], bytes
struct W64ImgBases {   // per-lane bases into the wave's two term images
  char *s01, *s2;              // stores: PIM_STRIDE * pim_row(j) + 8 h   /   W64_SIMG0 + W64_S2 * pim_row(j) + 8 h
  const char *t01, *t2;        // transposed reads: pim_tr_base(lane)     /   W64_SIMG0 + pim_tr_base<W64_S2>(lane)
};
// Stores tile kb (features 32kb..32kb+31) of all three terms: img[row j][feature]
__device__ __forceinline__ void store_terms(const W64ImgBases &ib, int kb, const uint32_t (&pk)[3][8]) {
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      const u32x2 v = {pk[t][2 * g], pk[t][2 * g + 1]};
      *reinterpret_cast<u32x2 *>((t == 2 ? ib.s2 : ib.s01) + 16 * ((t == 1 ? 8 : 0) + 4 * kb + g)) = v;
    }
}
// transposed fragment of term t: element i of lane (r,h) = term t of img[row 16c + 8h + i][feature 32fb + r]
__device__ __forceinline__ bf16x8 terms_tr_frag(const W64ImgBases &ib, int t, int c, int fb) {
  if (t == 2) return pim_tr_frag<W64_S2>(ib.t2 + W64_S2 * 16 * c + 2 * (32 * fb));
  return pim_tr_frag(ib.t01 + PIM_STRIDE * 16 * c + 2 * ((t == 1 ? 64 : 0) + 32 * fb));
}

// 8 fp32 values -> three bf16x8 terms (weight staging)
__device__ __forceinline__ void split3_vec(const float (&x)[8], bf16x8 (&o)[3]) {
  uint32_t pk[3][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float x0 = x[2 * q], x1 = x[2 * q + 1];
    pk[0][q] = hi16_pair(x1, x0);
    const float r0 = x0 - trunc_bf16(x0), r1 = x1 - trunc_bf16(x1);
    pk[1][q] = hi16_pair(r1, r0);
    pk[2][q] = hi16_pair(r1 - trunc_bf16(r1), r0 - trunc_bf16(r0));
  }
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const u32x4 v = {pk[t][0], pk[t][1], pk[t][2], pk[t][3]};
    o[t] = __builtin_bit_cast(bf16x8, v);
  }
}

// The six products of one K chunk for both output tiles (COOP: only tile w), streamed operand = the weight / image
// fragments a[tile][term] (single-buffered), fixed operand b[term].  The fragments of the NEXT chunk are requested as soon
// as the last MFMA that reads the register has been issued: term 2 after the first pair, term 0 after four pairs, term 1
// at the end -- each at least ~190 cycles before its first use in the next group.  W64_SB lets VALU / SALU work (the
// split of the next tile) float across but pins LDS reads and MFMAs.
#define W64_SB __builtin_amdgcn_sched_barrier(0x6);
#define W64_MF2(acc, a, ta, b, tb)                                                   \
  if (!COOP || w == 0) acc[0] = mfma_bf16((a)[0][ta], (b)[tb], acc[0]);              \
  if (!COOP || w == 1) acc[1] = mfma_bf16((a)[1][ta], (b)[tb], acc[1]);
#define W64_PIPE_A(acc, a, b, LOAD)                                                   \
  W64_MF2(acc, a, 2, b, 0) W64_SB LOAD(2) W64_SB                                      \
  W64_MF2(acc, a, 0, b, 2) W64_MF2(acc, a, 0, b, 1) W64_MF2(acc, a, 0, b, 0) W64_SB LOAD(0) W64_SB \
  W64_MF2(acc, a, 1, b, 1) W64_MF2(acc, a, 1, b, 0) W64_SB LOAD(1) W64_SB

template <int NH, int FQ, bool SPLIT = false>
struct W64Layout {
  static constexpr int NW = (NH > 1 ? NH - 1 : 1);
  static constexpr int FP = 8 * FQ;
  static constexpr int WIMG = 0;                                  // [NH-1][64][68]
  // SPLIT: instead, bf16 images W^T[out][in] of two layers side by side ([64][128], padded rows: mile_bf16_frag.h),
  // one per split term: [NSET][3][W64_TIMG bytes]
  static constexpr int NSET = NH / 2;
  // The row-contracting dW products run on bf16 terms as well (SPLIT_DW; split_dw(l) selects it per layer).  With three
  // hidden layers that only fits the register file because (a) every fragment address is one of four per-lane bases plus an
  // immediate (padded images; the XOR-swizzled images of rounds 1-2 needed ~30 address registers, re-derived per layer),
  // and (b) the first layer's activations wait in LDS between forward and backward (STASH: each wave's quarter of the
  // tile-exchange buffer, idle in the main loop).
  static constexpr bool SPLIT_DW = SPLIT;
  __host__ __device__ static constexpr bool split_dw(int l) { return SPLIT; }
  static constexpr bool STASH = SPLIT && NH >= 3;
  static constexpr int W1IMG = WIMG + (SPLIT ? NSET * 3 * (W64_TIMG / 4) : (NH - 1) * 64 * W64_RS);   // [FP][68]
  static constexpr int BIAS = W1IMG + FP * W64_RS;                // [NH][64]
  static constexpr int WO = BIAS + NH * 64;                       // [2][64]
  static constexpr int BO = WO + 128;                             // [4]
  static constexpr int WAVE0 = BO + 4;
  static constexpr int IMG = 0;                                   // per wave: [32][68]; SPLIT_DW: aliased by two bf16 images, 13 KB
  static constexpr int XT = IMG + (SPLIT_DW ? W64_SIMG / 4 : 32 * W64_RS);   // [32][FP]
  static constexpr int DOUT = XT + 32 * FP;                       // [32][2]
  static constexpr int WAVE_SZ = DOUT + 64;
  static constexpr int XB = WAVE0 + 4 * WAVE_SZ;                  // [2 pairs][2][2][16][64] tile exchange (COOP)
  static constexpr int MAIN = XB + 2 * 4 * 1024;
  static constexpr int RED = NW * 4 * 64 * W64_RS;                // end-of-kernel reduction alias
  static constexpr int TOTAL = (MAIN > RED ? MAIN : RED);
  static constexpr int BYTES = TOTAL * 4;
};

template <int NH, int FQ, bool SPLIT = false>
__global__ __launch_bounds__(256, 1) void k_grad_w64(const GradParams p, const typename W64FuseArg<FQ, SPLIT>::type fz) {
  using LY = W64Layout<NH, FQ, SPLIT>;
  static_assert(!SPLIT || NH >= 2, "SPLIT needs a hidden->hidden layer");
  constexpr int FP = LY::FP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const DevSpec &sp = p.spec;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int e = blockIdx.y, s = blockIdx.x;
  const int d = sp.d, F = sp.in_features;
  const float *th = p.theta + (size_t)e * d;
  float *slab = p.slabs + ((size_t)e * p.S + s) * p.dp;
  // slab / llpart stores: plain, or sc1 (write-through) when the update runs as this launch's epilogue
  constexpr bool FUSABLE = W64FuseArg<FQ, SPLIT>::FUSABLE;
  const bool fuse = FUSABLE && fz.enabled;
  auto st_quad = [&](int off, const f32x4 v) {
    if constexpr (FUSABLE) {
      if (fuse) {
        const __amdgpu_buffer_rsrc_t slab_rs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, p.dp * 4, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(upd_u32x4, v), slab_rs, off * 4, 0, 16);   // aux 16 = sc1
        return;
      }
    }
    *(f32x4 *)(slab + off) = v;
  };
  auto st_one = [&](float *q, const float v) {
    if constexpr (FUSABLE) {
      if (fuse) {
        __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
    }
    *q = v;
  };

  float *WIMG = lds + LY::WIMG, *W1IMG = lds + LY::W1IMG, *BIAS = lds + LY::BIAS;
  float *WO = lds + LY::WO, *BO = lds + LY::BO;
  float *wv = lds + LY::WAVE0 + wave * LY::WAVE_SZ;
  float *img = wv + LY::IMG, *xt = wv + LY::XT, *dout_l = wv + LY::DOUT;

  // dev (MILE_DEBUG=32): 100 MHz timestamps per workgroup: start, main loop done, slab stored, ticket drawn, epilogue done
  const bool stamp = (p.dbg & 32) && p.dbg_buf && tid == 0;
  long long *stamps = p.dbg_buf + ((size_t)e * p.S + s) * 16;
  if (stamp) stamps[0] = wall_clock64();
  // ---- stage this particle's weights in LDS --------------------------------------
  char *BIMG = reinterpret_cast<char *>(WIMG);   // SPLIT: term t of image set s at BIMG + (3 s + t) * W64_TIMG
  if (!(p.dbg & 2))
  for (int l = 1; l < NH; ++l) {
    const float *W = th + sp.w_off[l];
    if constexpr (SPLIT) {
      // thread task (out, 8 consecutive in): 8 coalesced loads (lanes along out), one 16-byte store per term
      for (int t = tid; t < 512; t += 256) {
        const int o = t & 63, ic = t >> 6;
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = W[(8 * ic + i) * 64 + o];
        bf16x8 tv[3];
        split3_vec(x, tv);
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)
          *reinterpret_cast<bf16x8 *>(BIMG + (3 * ((l - 1) >> 1) + tt) * W64_TIMG + pim_off(o, 8 * ((l - 1) & 1) + ic)) = tv[tt];
      }
    } else {
      for (int idx = tid; idx < 4096; idx += 256) WIMG[(l - 1) * 64 * W64_RS + (idx >> 6) * W64_RS + (idx & 63)] = W[idx];
    }
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
  // (the compiler keeps these three in scratch and read-modify-writes them once per block; measured: cheaper
  // than ds_add_f32 accumulators in LDS, 90.2 vs 91.8 us per launch)
  float boacc[2] = {0.0f, 0.0f};
  float llacc = 0.0f;

  const int NB = p.Npad / 32;
  const int b0 = (int)(((long long)s * NB) / p.S);
  const int b1 = (p.dbg & 1) ? b0 : (int)(((long long)(s + 1) * NB) / p.S);

  // Block body: mile_grad_w64_block.inc.  COOP = false: this wave does the whole block.
  // COOP = true: a PAIR of waves shares the block -- wave w of the pair computes output half w of every
  // GEMM (32 of the 64 MFMAs) and accumulates only its half of dW, db (and wave 0 the head).  fp32 form: the two
  // exchange their activation / dZ tiles through LDS after each layer.  SPLIT form (round 3): the pair keeps one
  // shared set of three-term images instead -- each wave splits only its own tile and publishes the terms, the
  // partner's half arrives as ready-made fragments -- so the splits are halved as well (mile_grad_w64_block.inc, COOPS).
  float *XB = lds + LY::XB + (wave >> 1) * 4 * 1024;   // pair's tile exchange: [2 halves][2 tiles][16][64]
  auto exchange = [&](f32x16(&T)[2], const int w, const int xid) {   // w is a compile-time constant at every call
    float *xb = XB + (xid & 1) * 2048;
    if (w == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) xb[r * 64 + lane] = T[0][r];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) xb[1024 + r * 64 + lane] = T[1][r];
    }
    __syncthreads();
    if (w == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) T[1][r] = xb[1024 + r * 64 + lane];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) T[0][r] = xb[r * 64 + lane];
    }
  };
  // SPLIT: the six per-lane LDS bases of the bf16 images (everything else in an address is an immediate)
  const char *wrb = BIMG + PIM_STRIDE * pim_row(j) + 16 * h;                          // weight rows (forward): + term + 32 ob rows + chunk
  const char *wtb = BIMG + pim_tr_base(lane);                                         // weights transposed (dH)
  W64ImgBases imb;                                                                    // this wave's term images
  imb.s01 = reinterpret_cast<char *>(img) + PIM_STRIDE * pim_row(j) + 8 * h;
  imb.s2 = reinterpret_cast<char *>(img) + W64_SIMG0 + W64_S2 * pim_row(j) + 8 * h;
  imb.t01 = reinterpret_cast<const char *>(img) + pim_tr_base(lane);
  imb.t2 = reinterpret_cast<const char *>(img) + W64_SIMG0 + pim_tr_base<W64_S2>(lane);
#ifdef MILE_LAB_W64_TIMING   // dev (tools/r03/lab): cycles per phase of a block, wave 0 of workgroup (0, 0), into p.dbg_buf[0..9]
  unsigned w64_tph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, w64_tl;
  {
    unsigned long long tm;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm)::"memory");
    w64_tl = (unsigned)tm;
  }
#define W64_TICK(k)                                                                     \
  {                                                                                     \
    unsigned long long tm_;                                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm_)::"memory");         \
    w64_tph[k] += (unsigned)tm_ - w64_tl;                                               \
    w64_tl = (unsigned)tm_;                                                             \
  }
#else
#define W64_TICK(k)
#endif
  const int nblk = b1 - b0, nfull = nblk >> 2, rem = nblk & 3;
  f32x4 xv_pre[FQ];   // X tile of the next full round, requested one block ahead
#pragma unroll
  for (int q = 0; q < FQ; ++q)
    xv_pre[q] = *(const f32x4 *)(p.Xp + (size_t)((b0 + (nfull ? wave : 0)) * 32 + j) * FP + 8 * q + 4 * h);
  for (int r = 0; r < nfull; ++r) {
    const int row0 = (b0 + 4 * r + wave) * 32;
    const int row_next = (b0 + 4 * (r + 1 < nfull ? r + 1 : r) + wave) * 32;
#define W64_COOP 0
#define W64_W 0
#define W64_PREF 1
#include "mile_grad_w64_block.inc"
#undef W64_COOP
#undef W64_W
#undef W64_PREF
  }
#define W64_PREF 0
  // STASH: a wave of a pair must not start exchanging tiles through the buffer its partner still keeps activations in
  // SPLIT: a shared block's pair images PA / PB are the two waves' private images of the main loop
  if constexpr (LY::STASH || SPLIT) __syncthreads();
  if (rem == 3) {               // three leftovers: one more independent round
    if (wave < 3) {
      const int row0 = (b0 + 4 * nfull + wave) * 32;
#define W64_COOP 0
#define W64_W 0
#include "mile_grad_w64_block.inc"
#undef W64_COOP
#undef W64_W
    }
  } else if (rem) {             // one or two leftovers: wave pairs share them
    if ((wave >> 1) < rem) {
      const int row0 = (b0 + 4 * nfull + (wave >> 1)) * 32;
      if (wave & 1) {
#define W64_COOP 1
#define W64_W 1
#include "mile_grad_w64_block.inc"
#undef W64_COOP
#undef W64_W
      } else {
#define W64_COOP 1
#define W64_W 0
#include "mile_grad_w64_block.inc"
#undef W64_COOP
#undef W64_W
      }
    } else {
      for (int k = 0; k < (SPLIT ? 4 * NH - 5 : 2 * NH - 1); ++k) __syncthreads();   // the other pair's barriers
    }
  }

#ifdef MILE_LAB_W64_TIMING
#ifndef MILE_LAB_W64_TIMING_WG
#define MILE_LAB_W64_TIMING_WG 0
#endif
  if (blockIdx.x == MILE_LAB_W64_TIMING_WG && blockIdx.y == 0 && tid == 0 && p.dbg_buf) {
#pragma unroll
    for (int k = 0; k < 10; ++k) p.dbg_buf[k] = w64_tph[k];
    p.dbg_buf[10] = nfull;
  }
#endif
  // ---- reduce the four waves' accumulators through LDS, write the slab -------------
  __syncthreads();  // weights and images are dead from here on; LDS is reused
  if (stamp) stamps[1] = wall_clock64();
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
        st_quad(sp.w_off[l + 1] + 4 * q, v);
      }
    } else {
#pragma unroll 4
      for (int idx = tid; idx < 4096; idx += 256) {
        const int o = (idx >> 6) * W64_RS + (idx & 63);
        st_one(out + idx, (R0[o] + R0[64 * W64_RS + o]) + (R0[2 * 64 * W64_RS + o] + R0[3 * 64 * W64_RS + o]));
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
    if (k < SM_W1) st_one(slab + sp.b_off[k >> 6] + (k & 63), v);
    else if (k < SM_WO) { if ((k - SM_W1) < F * 64) st_one(slab + sp.w_off[0] + (k - SM_W1), v); }
    else if (k < SM_BO) st_one(slab + sp.w_off[NH] + (k - SM_WO), v);
    else if (k < SM_LL) st_one(slab + sp.b_off[NH] + (k - SM_BO), v);
    else st_one(p.llpart + (size_t)e * p.S + s, v);
  }

  // ---- fused integrator epilogue: the particle's last-arriving workgroup runs the update -------------------
  if (stamp) { stamps[2] = wall_clock64(); stamps[3] = 0; stamps[4] = 0; }
  if constexpr (FUSABLE) {
    if (fuse) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // EVERY storing wave drains its sc1 stores ...
      __syncthreads();                                       // ... before the one lane that signals for all of them
      int *lflag = reinterpret_cast<int *>(lds);             // (the reduction scratch is dead behind this barrier)
      if (tid == 0) {
        const int old = __hip_atomic_fetch_add(fz.arrive + e, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == p.S - 1;
        if (last) __hip_atomic_store(fz.arrive + e, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        *lflag = last;
      }
      __syncthreads();
      if (stamp) stamps[3] = wall_clock64();
      if (*lflag) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no instruction: keeps the compiler from hoisting loads
        // quads per thread at 256 threads for the largest d this (NH, FQ) can have
        constexpr int NKF = ((8 * FQ * 64 + 64 + (NH - 1) * 4160 + 130) / 4 + 255) / 256;
        // inlined: every UpdParams field is then a scalar load from the kernel-argument segment.  (Out of line, reached
        // through a generic pointer, each field was re-read with a flat load after every store: 24 us instead of 7.)
        float(*ured)[UPD_NSUM + 1] = reinterpret_cast<float(*)[UPD_NSUM + 1]>(lds + 16);
        float *ubc = lds + 16 + 4 * (UPD_NSUM + 1);
        long long *ust = stamp ? stamps + 5 : nullptr;
        if (fz.kind == UPD_KIND_MID) upd_fast_body<NKF, 2, false, true, UPD_KIND_MID>(fz.upd, e, tid, 256, ured, ubc, ust);
        else if (fz.kind == UPD_KIND_REC) upd_fast_body<NKF, 2, false, true, UPD_KIND_REC>(fz.upd, e, tid, 256, ured, ubc, ust);
        else upd_fast_body<NKF, 2, false, true>(fz.upd, e, tid, 256, ured, ubc, ust);
        if (stamp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamps[4] = wall_clock64(); }
      }
    }
  }
}

template <int NH, int FQ>
constexpr int w64_fuse_nk() { return ((8 * FQ * 64 + 64 + (NH - 1) * 4160 + 130) / 4 + 255) / 256; }
