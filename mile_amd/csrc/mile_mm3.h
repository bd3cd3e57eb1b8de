// k_mm3: batched C = A.B on the gfx950 bf16 matrix pipe with fp32-faithful products -- the Dense products of WIDE nets
// (BASELINE config B4: [54 -> 256 x 4 -> 7], softmax head; anything whose hidden layers do not fit the fused kernels).
//
// Why not one fused forward+backward kernel as for widths 64 / 128: at width 256 a particle's three hidden matrices are
// 786 KB as fp32 (1.2 MB as bf16 terms) and its weight-gradient accumulators 768 KB -- neither fits a CU (160 KB LDS,
// 512 KB of registers), so the row-resident design of k_grad_w64 / k_grad_w128b has nowhere to keep them.  At this width
// the Dense products are real GEMMs (K = 256, or K = all data rows for dW): the layer-wise schedule of mile_hip.hip
// (launch_grad_wide) runs them here, with everything elementwise fused into the GEMM that produces or consumes it:
//   forward   H_l  = act(H_{l-1} W_l + b_l)                 A = activations [rows][in],  B = W_l terms [in][out]
//   dH        dZ_{l-1} = (dZ_l W_l^T) * act'(H_{l-1})       A = dZ_l [rows][out],        B = W_l terms read as [in][out] rows
//   dW        dW_l (+)= H_{l-1}^T dZ_l                      A = H_{l-1} read transposed,  B = dZ_l [rows][out]; K = data rows
// (src/flax_building_blocks/basic.py:42-61; the likelihood head and its skinny products are k_wide_head.)
//
// Arithmetic: every fp32 operand is the exact sum of three bf16 terms (top / middle / bottom 8 significand bits); a product
// a.b is accumulated in fp32 from the six bf16 MFMA products a3b1 a1b3 a2b2 a2b1 a1b2 a1b1 (the three dropped ones are
// below 2^-23 of a1b1) -- the scheme of k_grad_w64<.., SPLIT>, 3/8 of the fp32-MFMA time at fp32 accuracy.
// Here the split is paid once per operand element per workgroup tile and reused by 128 output columns / rows, and the
// weights arrive pre-split (k_wide_prep_weights), so it is a few % of the MFMA time instead of half of it.
// TERMS = 1 gives the bf16-operand form (one product, operands rounded to nearest-even) for callers that ask for it.
//
// Tiling: workgroup = 4 waves, C tile 128 x 128, K chunk 32 (56-64 KB of LDS: TWO workgroups per CU, one's staging under the
// other's MFMAs); wave (wm, wn) owns a 64 x 64 quadrant = 2 x 2 MFMA tiles (64 accumulator registers).  Operand tiles live in LDS
// as swizzled bf16 term images (mile_bf16_frag.h: one layout serves row reads ds_read_b128 and transposed reads
// ds_read_b64_tr_b16, conflict-free), single-buffered: the next chunk's global loads are issued before the 48 MFMAs of the
// current chunk and are split / stored behind them.  Whole-tile shapes take a predicate-free instantiation (FULL).
// grid = (N tiles, M tiles, batch): the batch (particle) index is slowest, so concurrently running workgroups share one
// particle's weights in L2.
#pragma once
#include <type_traits>

#include "mile_bf16_frag.h"
#include "mile_device.h"
#include "mile_grad_generic.h"

enum { MM_A_MK = 0, MM_A_KM = 1 };                       // A given as [M][K] (K contiguous) or as [K][M] (read transposed)
enum { MM_B_F32_KN = 0, MM_B_T3_KN = 1, MM_B_T3_NK = 2 };   // B: fp32 [K][N]; bf16 term planes [K][N]; term planes [N][K]
enum { MM_EPI_STORE = 0, MM_EPI_BIAS_ACT = 1, MM_EPI_ACT_GRAD = 2 };

struct MMParams {
  const float *A; long long sA; int lda;     // batch stride and leading dimension in floats (lda % 4 == 0, 16-byte aligned)
  const void *B; long long sB; int ldb;      // fp32: as A.  Term planes: bf16 elements, ldb % 8 == 0, tB = plane stride
  long long tB;
  float *C; long long sC; int ldc;
  int M, N, K;
  const float *bias; long long sBias;        // MM_EPI_BIAS_ACT: bias[n] (any alignment)
  const float *Hprev; long long sH; int ldh; // MM_EPI_ACT_GRAD: activation OUTPUT whose derivative multiplies C
  int act, apply_act, accumulate;            // accumulate: C += (row chunks of dW)
  float *colsum; long long sColsum;          // COLSUM: column sums of B over K -> colsum[bz * sColsum + n] (any alignment)
  int c_vec;                                 // C (and the accumulate reads of it) may use 16-byte accesses: ldc % 4 == 0, base aligned
  int xcd_remap;                             // 0 none; 1 N tiles of a row tile share an XCD; 2 all tiles of a batch entry do
  unsigned long long *dbg;                   // dev (MILE_DEBUG=64): per-phase 100 MHz tick sums of wave 0 of every workgroup
};

typedef uint32_t mm_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t mm_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mm_hi16_pair(float x1, float x0) {   // {bf16 bits of x1 : bf16 bits of x0}, truncating
  return __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
}
__device__ __forceinline__ float mm_trunc(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
#ifndef MILE_SPLIT_DOT2
#define MILE_SPLIT_DOT2 0
#endif
__device__ __forceinline__ float mm_sub_lo(uint32_t pk, float x) {   // x - (low bf16 of pk), exact
  const bf16x2 m = {(bf16)-1.0f, (bf16)0.0f};
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pk), m, x, false);
}
__device__ __forceinline__ float mm_sub_hi(uint32_t pk, float x) {   // x - (high bf16 of pk), exact
  const bf16x2 m = {(bf16)0.0f, (bf16)-1.0f};
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pk), m, x, false);
}

// four fp32 values -> TERMS packed bf16 quadruples (8 bytes each)
template <int TERMS>
__device__ __forceinline__ void mm_split4(const f32x4 x, mm_u32x2 (&pk)[TERMS]) {
  if constexpr (TERMS == 1) {
    const bf16x4 b = __builtin_convertvector(x, bf16x4);   // round to nearest even (v_cvt_pk_bf16_f32)
    pk[0] = __builtin_bit_cast(mm_u32x2, b);
  } else {
    static_assert(TERMS == 3, "one or three terms");
#ifdef MILE_LAB_MM_NO_SPLIT   // dev experiment (tools/r03/lab/mm3_lab.hip): wrong results; what the on-the-fly split costs
    pk[0] = mm_u32x2{__float_as_uint(x[0]), __float_as_uint(x[1])};
    pk[1] = mm_u32x2{__float_as_uint(x[2]), __float_as_uint(x[3])};
    pk[2] = pk[0];
    return;
#endif
    pk[0] = mm_u32x2{mm_hi16_pair(x[1], x[0]), mm_hi16_pair(x[3], x[2])};
#if MILE_SPLIT_DOT2   // residuals from the packed term, one v_dot2c_f32_bf16 per element (mile_grad_w64.h, split3_pk)
    const f32x4 r = {mm_sub_lo(pk[0][0], x[0]), mm_sub_hi(pk[0][0], x[1]), mm_sub_lo(pk[0][1], x[2]), mm_sub_hi(pk[0][1], x[3])};
    pk[1] = mm_u32x2{mm_hi16_pair(r[1], r[0]), mm_hi16_pair(r[3], r[2])};
    const f32x4 q = {mm_sub_lo(pk[1][0], r[0]), mm_sub_hi(pk[1][0], r[1]), mm_sub_lo(pk[1][1], r[2]), mm_sub_hi(pk[1][1], r[3])};
#else
    const f32x4 r = {x[0] - mm_trunc(x[0]), x[1] - mm_trunc(x[1]), x[2] - mm_trunc(x[2]), x[3] - mm_trunc(x[3])};
    pk[1] = mm_u32x2{mm_hi16_pair(r[1], r[0]), mm_hi16_pair(r[3], r[2])};
    const f32x4 q = {r[0] - mm_trunc(r[0]), r[1] - mm_trunc(r[1]), r[2] - mm_trunc(r[2]), r[3] - mm_trunc(r[3])};
#endif
    pk[2] = mm_u32x2{mm_hi16_pair(q[1], q[0]), mm_hi16_pair(q[3], q[2])};
  }
}

// LDS image geometry (KC = K chunk, 32 or 64).  "Row-major K" tiles ([128 rows][KC k]: A_MK, B_NK) pack their terms side by
// side in 128-column images -- KC = 64: term 0 in columns 0..63 and term 1 in columns 64..127 of image 0, term 2 in columns
// 0..63 of image 1; KC = 32: term t in columns 32 t .. 32 t + 31 of one image.  "K-major" tiles ([KC k][128 cols]: A_KM, B_KN)
// take one KC-row image per term.  KC = 32 keeps a workgroup at <= 64 KB so that TWO fit a CU: one's global-load / split /
// store phases and epilogue then run under the other's MFMAs (one wave per SIMD has nothing else to hide them behind).
template <int TERMS, int KC> __host__ __device__ constexpr int mm_rowk_bytes() { return (KC == 64 && TERMS == 3 ? 2 : 1) * 128 * 256; }
template <int TERMS, int KC> __host__ __device__ constexpr int mm_kmaj_bytes() { return TERMS * KC * 256; }
template <int KC> __device__ __forceinline__ int mm_rowk_img(int t) { return (KC == 64 && t == 2) ? 128 * 256 : 0; }
template <int KC> __device__ __forceinline__ int mm_rowk_ch(int t) { return KC == 64 ? (t == 1 ? 8 : 0) : 4 * t; }

template <int ALAY, int BSRC, int TERMS, int KC>
struct MMLayout {
  static constexpr int A_BYTES = ALAY == MM_A_MK ? mm_rowk_bytes<TERMS, KC>() : mm_kmaj_bytes<TERMS, KC>();
  static constexpr int B_BYTES = BSRC == MM_B_T3_NK ? mm_rowk_bytes<TERMS, KC>() : mm_kmaj_bytes<TERMS, KC>();
  static constexpr int STAGE_BYTES = 4 * 32 * 68 * 4;   // the epilogue's per-wave [32][68] fp32 staging, aliasing the images
  static constexpr int BYTES = A_BYTES + B_BYTES > STAGE_BYTES ? A_BYTES + B_BYTES : STAGE_BYTES;
};

// COLSUM (dW form only: B = dZ as fp32 [K rows][N]): the workgroups of M tile 0 also sum the columns of B over all K rows --
// the bias gradient dZ^T 1 -- from the B tiles they stage anyway, and write it to p.colsum (fixed order: deterministic).
// Chunks of global loads in flight ahead of the MFMAs (register stages).  vmcnt retires loads IN ORDER, so every operand
// stream must have the same depth: a one-deep B stream behind a deeper A stream drains the A loads with it (measured:
// no gain).  _DW: the dW form (two fp32 streams, 32 registers per stage); _FWD: forward / dH (fp32 A + term-plane B, 40).
#ifndef MILE_MM_PF_DW
#define MILE_MM_PF_DW 1
#endif
#ifndef MILE_MM_PF_FWD
#define MILE_MM_PF_FWD 1
#endif
#ifndef MILE_MM_OCC
#define MILE_MM_OCC 2     // workgroups per CU the register allocation is sized for
#endif
// ACT (MILE_ACT_* or -1 = none) and ACCUM are template parameters: the epilogue's 64 elements per lane run branch-free
// (as a per-element run-time switch the epilogue was most of the kernel's instructions; dispatched inside the kernel its
// eight inlined copies spilled 150 registers).
// FULL: M and N are multiples of 128 (the 256-wide layers of B4 at every chunk size the host picks; any K -- a ragged last chunk
// costs two wave-uniform tests): no tile / row / column predicates and no timing stamps -- the MFMA groups of the K loop are one
// basic block the scheduler can order freely (the general form carries ~3 SALU instructions per MFMA and a branch around every
// MFMA group).
template <int ALAY, int BSRC, int EPI, int TERMS, int KC, int ACT, bool ACCUM, bool COLSUM = false, bool FULL = false>
__global__ __launch_bounds__(256, MILE_MM_OCC) void k_mm3(const MMParams p) {
  using LY = MMLayout<ALAY, BSRC, TERMS, KC>;
  static_assert(KC == 32 || KC == 64, "K chunk");
  static_assert(!COLSUM || BSRC == MM_B_F32_KN, "column sums come from an fp32 B operand");
  extern __shared__ __attribute__((aligned(16))) char mm_smem[];
  char *Aimg = mm_smem, *Bimg = mm_smem + LY::A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware placement (speed only): workgroups are dealt round-robin over the 8 XCDs by linear id, each XCD has its own L2.
  // The workgroups that read the SAME operand chunk -- the N tiles of one (M tile, batch entry), and for the dW form all
  // tiles of a batch entry -- get linear ids 8 apart, so the second reader hits the first one's L2 lines instead of HBM.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.xcd_remap) {
    const int nx = gridDim.x, ny = gridDim.y, nz = gridDim.z;
    const long long lin = (long long)blockIdx.x + (long long)nx * (blockIdx.y + (long long)ny * blockIdx.z);
    if (p.xcd_remap == 2) {            // all nx * ny tiles of a batch entry on one XCD (nz % 8 == 0)
      const int per = nx * ny;
      const long long grp = lin / (8LL * per);
      const int rem = (int)(lin % (8LL * per));
      bz = (int)(grp * 8) + (rem & 7);
      const int t = rem >> 3;
      bx = t % nx; by = t / nx;
    } else {                           // the nx N tiles of an (M tile, batch entry) on one XCD
      const long long plane = lin / ((long long)nx * ny);
      const int rem = (int)(lin % ((long long)nx * ny));
      const int nfull = ny / 8 * 8;    // M tiles beyond the last full group of 8 keep the plain order
      bz = (int)plane;
      if (rem < nfull * nx) {
        const int grp = rem / (8 * nx), r8 = rem % (8 * nx);
        by = grp * 8 + (r8 & 7);
        bx = r8 >> 3;
      } else {
        const int t = rem - nfull * nx;
        by = nfull + t / nx;
        bx = t % nx;
      }
    }
  }
  const int n0 = bx * 128;
  const int m0 = by * 128;
  const int M = p.M, N = p.N, K = p.K;
  const float *A = p.A + (size_t)bz * p.sA;
  float *C = p.C + (size_t)bz * p.sC;

  // ---- loop-invariant LDS byte offsets ---------------------------------------------------------------------------------
  // img_off(row, ch) = 256 row + 16 (ch ^ swz(row)) and swz depends on row & 15 only, so everything that moves a row by a
  // multiple of 16 (k-steps, MFMA tiles, passes) or switches the term image is a compile-time constant on top of a handful of
  // per-lane offsets.  Left to the compiler these were recomputed for every fragment of every chunk: ~150 of the ~380 VALU
  // instructions per chunk and wave (profiles/r02/05_b4_mm3_pmc_counters.txt), in a kernel where VALU time adds to MFMA time.
  auto swz = [](int row) { return ((row & 3) << 2) | ((row >> 2) & 3); };
  constexpr int NKS = KC / 16;
  // fragments, row-major-K images (A_MK / B_NK): [term][k-step]; MFMA tile q adds 32 rows = 8192 bytes
  int afr[ALAY == MM_A_MK ? TERMS : 1][NKS], bfr_o[BSRC == MM_B_T3_NK ? TERMS : 1][NKS];
  // fragments, K-major images (A_KM / B_KN), ds_read_b64_tr_b16: [tile q][first / second 4-row group]; k-step adds 4096, term KC * 256
  int atr[ALAY == MM_A_KM ? 2 : 1][2], btr[BSRC != MM_B_T3_NK ? 2 : 1][2];
  {
    const int g1 = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
    for (int t = 0; t < TERMS; ++t)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        if constexpr (ALAY == MM_A_MK) afr[t][ks] = mm_rowk_img<KC>(t) + 256 * (64 * wm + r) + 16 * ((mm_rowk_ch<KC>(t) + 2 * ks + h) ^ swz(r & 15));
        if constexpr (BSRC == MM_B_T3_NK) bfr_o[t][ks] = mm_rowk_img<KC>(t) + 256 * (64 * wn + r) + 16 * ((mm_rowk_ch<KC>(t) + 2 * ks + h) ^ swz(r & 15));
      }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if constexpr (ALAY == MM_A_KM) {
        const int ch = ((64 * wm + 32 * q) >> 3) + 2 * g1 + (pp >> 1);
        atr[q][0] = img_off(8 * h + qq, ch) + 8 * (pp & 1);
        atr[q][1] = img_off(8 * h + qq + 4, ch) + 8 * (pp & 1);
      }
      if constexpr (BSRC != MM_B_T3_NK) {
        const int ch = ((64 * wn + 32 * q) >> 3) + 2 * g1 + (pp >> 1);
        btr[q][0] = img_off(8 * h + qq, ch) + 8 * (pp & 1);
        btr[q][1] = img_off(8 * h + qq + 4, ch) + 8 * (pp & 1);
      }
    }
  }
  auto tr_read = [&](const char *img, int off0, int off1) {   // the two ds_read_b64_tr_b16 of tr_frag at precomputed offsets
    const bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(MILE_LDS_PTR(bf16x4, img + off0));
    const bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(MILE_LDS_PTR(bf16x4, img + off1));
    return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
  };

  // ---- global -> register staging of one K chunk -------------------------------------------------------------------
  // fp32 tiles, one float4 per thread and pass.  [128 rows][KC k]: KC / 4 threads per row; [KC k][128 cols]: 32 per row.
  constexpr int A_TPR = KC / 4, A_NP = ALAY == MM_A_MK ? A_TPR / 2 : KC / 8;
  constexpr int PF = BSRC == MM_B_F32_KN ? MILE_MM_PF_DW : MILE_MM_PF_FWD;
  f32x4 ra[PF][A_NP];
  // Global operand loads are raw buffer loads: a per-thread byte offset fixed for the whole workgroup (rows / columns outside
  // the matrix get an out-of-range offset and read as zero) + a SCALAR offset that walks K -- no per-load 64-bit address
  // arithmetic or predicates in the K loop (they were ~100 of the VALU instructions per chunk and wave).  Only a ragged last
  // chunk (K % KC != 0) pays a select per load.
  constexpr int MM_OOB = 0x7ffffff0;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(A), 0, (int)min((long long)(ALAY == MM_A_MK ? M : K) * p.lda * 4, 0x7fffffffLL), 0x00020000);
  int voa[A_NP];
  if constexpr (ALAY == MM_A_MK) {
    const int c4 = tid % A_TPR, r0 = tid / A_TPR;
#pragma unroll
    for (int i = 0; i < A_NP; ++i) {
      const int m = m0 + r0 + (256 / A_TPR) * i;
      voa[i] = (FULL || m < M) ? (m * p.lda + 4 * c4) * 4 : MM_OOB;
    }
  } else {
    const int c4 = tid & 31, r0 = tid >> 5;
#pragma unroll
    for (int i = 0; i < A_NP; ++i) voa[i] = (FULL || m0 + 4 * c4 < M) ? ((r0 + 8 * i) * p.lda + m0 + 4 * c4) * 4 : MM_OOB;
  }
  auto load_a = [&](const int st, const int k0) {
    const int so = ALAY == MM_A_MK ? k0 * 4 : k0 * p.lda * 4;
    const bool ragged = k0 + KC > K;
#pragma unroll
    for (int i = 0; i < A_NP; ++i) {
      int vo = voa[i];
      if (ragged) {
        const bool ok = ALAY == MM_A_MK ? k0 + 4 * (tid % A_TPR) < K : k0 + (tid >> 5) + 8 * i < K;
        vo = ok ? vo : MM_OOB;
      }
#ifdef MILE_LAB_MM_NO_LOADA   // dev experiment: A from nowhere (wrong results; what the activation stream costs)
      ra[st][i] = f32x4{0.5f, 0.25f, 0.125f, 1.0f};
      (void)vo; (void)so;
#else
      ra[st][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, vo, so, 0));
#endif
    }
  };
  // staging stores: [128 rows][KC k] tiles move 256 / A_TPR rows (a multiple of 16) per pass: one offset per term;
  // [KC k][128] tiles move 8 rows per pass: one offset per pass parity, + 4096 per pair of passes, + KC * 256 per term
  int aso[ALAY == MM_A_MK ? TERMS : 2];
  if constexpr (ALAY == MM_A_MK) {
    const int c4 = tid % A_TPR, r0 = tid / A_TPR;
#pragma unroll
    for (int t = 0; t < TERMS; ++t) aso[t] = mm_rowk_img<KC>(t) + 256 * r0 + 16 * ((mm_rowk_ch<KC>(t) + (c4 >> 1)) ^ swz(r0 & 15)) + 8 * (c4 & 1);
  } else {
    const int c4 = tid & 31, r0 = tid >> 5;
#pragma unroll
    for (int par = 0; par < 2; ++par) aso[par] = img_off(r0 + 8 * par, c4 >> 1) + 8 * (c4 & 1);
  }
  auto store_a = [&](const int st) {
#pragma unroll
    for (int i = 0; i < A_NP; ++i) {
      mm_u32x2 pk[TERMS];
      mm_split4<TERMS>(ra[st][i], pk);
#pragma unroll
      for (int t = 0; t < TERMS; ++t) {
        if constexpr (ALAY == MM_A_MK) *reinterpret_cast<mm_u32x2 *>(Aimg + aso[t] + 256 * (256 / A_TPR) * i) = pk[t];
        else *reinterpret_cast<mm_u32x2 *>(Aimg + aso[i & 1] + 4096 * (i >> 1) + t * KC * 256) = pk[t];
      }
    }
  };
  // B: fp32 [K][N] (split here) or pre-split bf16 term planes (16-byte chunks straight into the images)
  constexpr int BF_NP = KC / 8;                                   // fp32 [KC k][128 n]: passes of 8 k rows
  constexpr int BT_NP = KC / 16;                                  // term planes: KN 16 k rows per pass; NK 2048 / KC n rows per pass
  constexpr int NK_CPR = KC / 8;                                  // NK: 16-byte chunks per n row
  f32x4 rb[BSRC == MM_B_F32_KN ? PF : 1][BSRC == MM_B_F32_KN ? BF_NP : 1];
  mm_u32x4 rbt[BSRC == MM_B_F32_KN ? 1 : PF][BSRC == MM_B_F32_KN ? 1 : TERMS][BT_NP];
  float csum[4] = {0.0f, 0.0f, 0.0f, 0.0f};                        // COLSUM: this thread's four columns, its k rows
  const void *Bb = BSRC == MM_B_F32_KN ? (const void *)((const float *)p.B + (size_t)bz * p.sB) : (const void *)((const bf16 *)p.B + (size_t)bz * p.sB);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void *>(Bb), 0, (int)min(BSRC == MM_B_F32_KN ? (long long)K * p.ldb * 4 : (long long)TERMS * p.tB * 2, 0x7fffffffLL), 0x00020000);
  constexpr int B_NP = BSRC == MM_B_F32_KN ? BF_NP : BT_NP;
  int vob[B_NP];
  if constexpr (BSRC == MM_B_F32_KN) {
    const int c4 = tid & 31, r0 = tid >> 5;
#pragma unroll
    for (int i = 0; i < B_NP; ++i) vob[i] = (FULL || n0 + 4 * c4 < N) ? ((r0 + 8 * i) * p.ldb + n0 + 4 * c4) * 4 : MM_OOB;
  } else if constexpr (BSRC == MM_B_T3_KN) {
    const int c = tid & 15, r0 = tid >> 4;            // [KC k][128 n]: 16 chunks of 8 columns per row
#pragma unroll
    for (int i = 0; i < B_NP; ++i) vob[i] = (FULL || n0 + 8 * c < N) ? ((r0 + 16 * i) * p.ldb + n0 + 8 * c) * 2 : MM_OOB;
  } else {
    const int c = tid % NK_CPR, r0 = tid / NK_CPR;    // [128 n][KC k]: KC / 8 chunks of 8 k per row
#pragma unroll
    for (int i = 0; i < B_NP; ++i) {
      const int n = n0 + r0 + (256 / NK_CPR) * i;
      vob[i] = (FULL || n < N) ? (n * p.ldb + 8 * c) * 2 : MM_OOB;
    }
  }
  auto load_b = [&](const int st, const int k0) {
    const bool ragged = k0 + KC > K;
    if constexpr (BSRC == MM_B_F32_KN) {
      const int so = k0 * p.ldb * 4;
#pragma unroll
      for (int i = 0; i < B_NP; ++i) {
        int vo = vob[i];
        if (ragged) vo = k0 + (tid >> 5) + 8 * i < K ? vo : MM_OOB;
        rb[st][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, vo, so, 0));
      }
    } else {
#pragma unroll
      for (int t = 0; t < TERMS; ++t) {
        const int so = BSRC == MM_B_T3_KN ? (int)((t * p.tB + (long long)k0 * p.ldb) * 2) : (int)((t * p.tB + k0) * 2);
#pragma unroll
        for (int i = 0; i < B_NP; ++i) {
          int vo = vob[i];
          if (ragged) {
            const bool ok = BSRC == MM_B_T3_KN ? k0 + (tid >> 4) + 16 * i < K : k0 + 8 * (tid % NK_CPR) < K;
            vo = ok ? vo : MM_OOB;
          }
          rbt[st][t][i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, vo, so, 0);
        }
      }
    }
  };
  int bso[BSRC == MM_B_T3_NK ? TERMS : 2];
  if constexpr (BSRC == MM_B_F32_KN) {
    const int c4 = tid & 31, r0 = tid >> 5;
#pragma unroll
    for (int par = 0; par < 2; ++par) bso[par] = img_off(r0 + 8 * par, c4 >> 1) + 8 * (c4 & 1);
  } else if constexpr (BSRC == MM_B_T3_KN) {
    bso[0] = img_off(tid >> 4, tid & 15); bso[1] = 0;                  // rows r0 + 16 i: + 4096 i
  } else {
    const int c = tid % NK_CPR, r0 = tid / NK_CPR;
#pragma unroll
    for (int t = 0; t < TERMS; ++t) bso[t] = mm_rowk_img<KC>(t) + 256 * r0 + 16 * ((mm_rowk_ch<KC>(t) + c) ^ swz(r0 & 15));
  }
  auto store_b = [&](const int st) {
    if constexpr (BSRC == MM_B_F32_KN) {
#pragma unroll
      for (int i = 0; i < BF_NP; ++i) {
        if constexpr (COLSUM) {
#pragma unroll
          for (int q = 0; q < 4; ++q) csum[q] += rb[st][i][q];
        }
        mm_u32x2 pk[TERMS];
        mm_split4<TERMS>(rb[st][i], pk);
#pragma unroll
        for (int t = 0; t < TERMS; ++t) *reinterpret_cast<mm_u32x2 *>(Bimg + bso[i & 1] + 4096 * (i >> 1) + t * KC * 256) = pk[t];
      }
    } else if constexpr (BSRC == MM_B_T3_KN) {
#pragma unroll
      for (int t = 0; t < TERMS; ++t)
#pragma unroll
        for (int i = 0; i < BT_NP; ++i) *reinterpret_cast<mm_u32x4 *>(Bimg + bso[0] + 4096 * i + t * KC * 256) = rbt[st][t][i];
    } else {
#pragma unroll
      for (int t = 0; t < TERMS; ++t)
#pragma unroll
        for (int i = 0; i < BT_NP; ++i) *reinterpret_cast<mm_u32x4 *>(Bimg + bso[t] + 256 * (256 / NK_CPR) * i) = rbt[st][t][i];
    }
  };

  f32x16 acc[2][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.0f;
  };
  zero_acc();
  // 32 x 32 tiles of this wave that lie inside the matrix (skinny products skip the MFMAs of the rest; wave-uniform)
  bool mt_on[2], nt_on[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) nt_on[q] = FULL || n0 + 64 * wn + 32 * q < N;

  // ---- epilogue of one C tile: D[m = acc_m(reg, h)][n = r] per 32 x 32 MFMA tile ---------------------------------------
  // The wave's 64 x 64 quadrant goes through LDS in two 32-row halves ([32][68] floats = 8.7 KB per wave, aliasing the operand
  // images) so that EVERY global access of the epilogue is a 16-byte one: with 4-byte accesses in the accumulator layout the
  // C stores cost 73 ms and the dH form's H_{l-1} reads 82 ms of a 409 ms B4 gradient (MILE_MM_SKIP knobs,
  // profiles/r02/05_b4_mm3_pmc_counters.txt) -- four times the memory instructions, and the reads were 16 dependent round trips.
  auto epilogue = [&](const int m0) {
    __syncthreads();                                            // every wave has finished reading the operand images
    float *stage = reinterpret_cast<float *>(mm_smem) + wave * (32 * 68);
    const int c4 = lane & 15, rr = lane >> 4;                   // this lane's float4 column (of 16) and row within a group of 4
    const int nq = n0 + 64 * wn + 4 * c4;                       // first of its four columns
    float bias4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if constexpr (EPI == MM_EPI_BIAS_ACT) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (FULL || nq + c < N) bias4[c] = p.bias[(size_t)bz * p.sBias + nq + c];
    }
    const bool vec_ok = (FULL || nq + 3 < N) && p.c_vec;                  // whole float4 inside the matrix and 16-byte aligned
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      if (!mt_on[a]) continue;                                  // wave-uniform
      const int mr0 = m0 + 64 * wm + 32 * a;
      f32x4 hv[8], cv[8];
      if constexpr (EPI == MM_EPI_ACT_GRAD) {                   // all reads of the half tile in flight before anything waits
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int m = mr0 + rr + 4 * i;
          hv[i] = (FULL || (m < M && nq < N)) ? *(const f32x4 *)(p.Hprev + (size_t)bz * p.sH + (size_t)m * p.ldh + nq)
                                                         : f32x4{1.0f, 1.0f, 1.0f, 1.0f};
        }
      }
      if constexpr (ACCUM) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int m = mr0 + rr + 4 * i;
          const float *c = C + (size_t)m * p.ldc + nq;
          cv[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          if (FULL || m < M) {
            if (vec_ok) cv[i] = *(const f32x4 *)c;
            else {
#pragma unroll
              for (int cc = 0; cc < 4; ++cc)
                if (nq + cc < N) cv[i][cc] = c[cc];
            }
          }
        }
      }
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) stage[(4 * h + (q & 3) + 8 * (q >> 2)) * 68 + 32 * b + r] = acc[a][b][q];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");    // wave-private region: LDS ops of one wave execute in order
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = mr0 + rr + 4 * i;
        f32x4 v = *(const f32x4 *)(stage + (rr + 4 * i) * 68 + 4 * c4);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          float x = v[cc];
          if constexpr (EPI == MM_EPI_BIAS_ACT) {
            x += bias4[cc];
            if constexpr (ACT == MILE_ACT_RELU) x = fmaxf(x, 0.0f);
            else if constexpr (ACT >= 0) x = act_fwd(ACT, x);
          }
          if constexpr (EPI == MM_EPI_ACT_GRAD) {
            if constexpr (ACT == MILE_ACT_RELU) x = hv[i][cc] > 0.0f ? x : 0.0f;
            else x *= act_bwd(ACT, hv[i][cc]);
          }
          if constexpr (ACCUM) x += cv[i][cc];
          v[cc] = x;
        }
        if (FULL || m < M) {
          float *c = C + (size_t)m * p.ldc + nq;
          if (vec_ok) *(f32x4 *)c = v;
          else {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
              if (nq + cc < N) c[cc] = v[cc];
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  };
  const int nk = (K + KC - 1) / KC;
  // software pipeline: the operands of chunks kc+1 .. kc+PF are in flight (register stage = chunk % PF) while chunk kc is
  // multiplied; both streams the same depth (vmcnt retires in order)
#pragma unroll
  for (int j = 0; j < PF; ++j)
    if (j < nk) {
      load_a(j, KC * j);
      load_b(j, KC * j);
    }
#pragma unroll
  for (int q = 0; q < 2; ++q) mt_on[q] = FULL || m0 + 64 * wm + 32 * q < M;
  for (int g0 = 0; g0 < nk; g0 += PF)
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    const int kc = g0 + j;
    if (kc >= nk) break;
    long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0;
    const bool stampw = !FULL && p.dbg != nullptr && tid == 0;
    if (stampw) ts0 = wall_clock64();
    __syncthreads();          // every wave has read the previous chunk's images
    if (stampw) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ts1 = wall_clock64(); }
    store_a(j);
    store_b(j);
    if (stampw) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ts2 = wall_clock64(); }
    __syncthreads();
    if (stampw) ts3 = wall_clock64();
    if (kc + PF < nk) {       // refill this stage
      load_a(j, KC * (kc + PF));
      load_b(j, KC * (kc + PF));
    }
    const int ksteps = min(KC / 16, (K - KC * kc + 15) / 16);
    auto kstep = [&](const int ks) {
      bf16x8 af[2][TERMS], bfr[2][TERMS];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int t = 0; t < TERMS; ++t) {
          if constexpr (ALAY == MM_A_MK) af[q][t] = *reinterpret_cast<const bf16x8 *>(Aimg + afr[t][ks] + 8192 * q);
          else af[q][t] = tr_read(Aimg + t * KC * 256 + 4096 * ks, atr[q][0], atr[q][1]);
          if constexpr (BSRC == MM_B_T3_NK) bfr[q][t] = *reinterpret_cast<const bf16x8 *>(Bimg + bfr_o[t][ks] + 8192 * q);
          else bfr[q][t] = tr_read(Bimg + t * KC * 256 + 4096 * ks, btr[q][0], btr[q][1]);
        }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if (!(mt_on[a] && nt_on[b])) continue;
          if constexpr (TERMS == 3) {   // small terms first
            acc[a][b] = mfma_bf16(af[a][2], bfr[b][0], acc[a][b]);
            acc[a][b] = mfma_bf16(af[a][0], bfr[b][2], acc[a][b]);
            acc[a][b] = mfma_bf16(af[a][1], bfr[b][1], acc[a][b]);
            acc[a][b] = mfma_bf16(af[a][1], bfr[b][0], acc[a][b]);
            acc[a][b] = mfma_bf16(af[a][0], bfr[b][1], acc[a][b]);
          }
          acc[a][b] = mfma_bf16(af[a][0], bfr[b][0], acc[a][b]);
        }
    };
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {   // one k-step's fragments live at a time; a ragged last chunk skips the empty k-steps
      if (ks < ksteps) kstep(ks);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (stampw) {
      ts4 = wall_clock64();
      atomicAdd(p.dbg + 0, (unsigned long long)(ts1 - ts0)); atomicAdd(p.dbg + 1, (unsigned long long)(ts2 - ts1));
      atomicAdd(p.dbg + 2, (unsigned long long)(ts3 - ts2)); atomicAdd(p.dbg + 3, (unsigned long long)(ts4 - ts3));
      atomicAdd(p.dbg + 5, 1ull);
    }
    if (kc == nk - 1) {       // the C tile is complete
#ifdef MILE_LAB_MM_NO_EPI   // dev experiment (tools/r03/lab/mm3_lab.hip): wrong results; what the C tile's epilogue costs
      if (acc[0][0][0] == 123.456f) epilogue(m0);
#else
      epilogue(m0);
#endif
      if (stampw) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); atomicAdd(p.dbg + 4, (unsigned long long)(wall_clock64() - ts4)); atomicAdd(p.dbg + 6, 1ull); }
    }
  }
  if constexpr (COLSUM) {
    if (by == 0) {     // (FULL changes nothing here) thread (c4, r0) holds columns n0 + 4 c4 .. + 3 summed over its k rows: add the 8 row groups
      __syncthreads();
      float *red = reinterpret_cast<float *>(mm_smem);                    // [8][128]
      const int c4 = tid & 31, r0 = tid >> 5;
      *reinterpret_cast<f32x4 *>(red + r0 * 128 + 4 * c4) = f32x4{csum[0], csum[1], csum[2], csum[3]};
      __syncthreads();
      if (tid < 128 && n0 + tid < N) {
        float t = 0.0f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += red[g * 128 + tid];
        float *o = p.colsum + (size_t)bz * p.sColsum + n0 + tid;
        *o = p.accumulate ? *o + t : t;
      }
    }
  }
}

// ---- weights of one layer -> zero-padded bf16 term planes Wt[e][t][in][outp] ------------------------------------------
// theta rows are d floats apart (any alignment); outp = out rounded up to 8 so that every plane row is 16-byte aligned.
template <int TERMS>
__global__ __launch_bounds__(256) void k_wide_prep_weights(const float *theta, long long d, int w_off, int fin, int fout, int outp,
                                                          bf16 *Wt, long long sW) {
  const int e = blockIdx.y;
  const float *W = theta + (size_t)e * d + w_off;
  bf16 *o = Wt + (size_t)e * sW;
  const long long plane = (long long)fin * outp;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < plane; idx += (long long)gridDim.x * 256) {
    const int i = (int)(idx / outp), c = (int)(idx % outp);
    const float x = c < fout ? W[(size_t)i * fout + c] : 0.0f;
    if constexpr (TERMS == 1) {
      o[idx] = (bf16)x;
    } else {
      const float x1 = mm_trunc(x), r1 = x - x1, x2 = mm_trunc(r1), x3 = r1 - x2;
      o[idx] = __builtin_bit_cast(bf16, (uint16_t)(__float_as_uint(x1) >> 16));
      o[plane + idx] = __builtin_bit_cast(bf16, (uint16_t)(__float_as_uint(x2) >> 16));
      o[2 * plane + idx] = __builtin_bit_cast(bf16, (uint16_t)(__float_as_uint(x3) >> 16));
    }
  }
}

// ---- head: per-row log-likelihood and d/d(out) in place on out [E][R][ld] (ld >= K, padding untouched = 0) -------------
// src/training/probabilistic.py:92-109 (nansum: NaN rows contribute nothing); one workgroup per particle, fixed order.
__global__ __launch_bounds__(256) void k_wide_head(float *out, long long sOut, int ld, const void *y, long long r0, int R, int K, int task,
                                                   float *llacc, int first_chunk) {
  __shared__ float red[4];
  const int e = blockIdx.x, tid = threadIdx.x;
  float *o = out + (size_t)e * sOut;
  float ll = 0.0f;
  for (int rr = tid; rr < R; rr += 256) {
    float *z = o + (size_t)rr * ld;
    if (task == MILE_TASK_REGRESSION) {
      float dmu, ds;
      ll += row_loss_regr(z[0], z[1], ((const float *)y)[r0 + rr], dmu, ds);
      z[0] = dmu; z[1] = ds;
    } else {
      const int yi = ((const int32_t *)y)[r0 + rr];
      float m = z[0];
      for (int c = 1; c < K; ++c) m = fmaxf(m, z[c]);
      float se = 0.0f;
      for (int c = 0; c < K; ++c) se += expf(z[c] - m);
      const float lse = m + logf(se);
      const float l1 = z[yi] - lse;
      const bool bad = isnan(l1);
      for (int c = 0; c < K; ++c) z[c] = bad ? 0.0f : ((c == yi ? 1.0f : 0.0f) - expf(z[c] - lse));
      ll += bad ? 0.0f : l1;
    }
  }
  ll = wave_sum(ll);
  if ((tid & 63) == 0) red[tid >> 6] = ll;
  __syncthreads();
  if (tid == 0) {
    const float t = (red[0] + red[1]) + (red[2] + red[3]);
    llacc[e] = first_chunk ? t : llacc[e] + t;
  }
}

// ---- the LAST layer in one pass over the last hidden activations (K <= 8 outputs, hidden width W <= 256) ---------------------
// Per row: out = H W + b in fp32 FMAs (a wave takes a row, a lane four of its columns, DPP butterfly sums), log-likelihood and
// d(out) exactly as k_wide_head, dZ = (d(out) W^T) * act'(H) for the layer below written at once, and the row's contribution to
// the weight / bias gradient accumulated in registers (a lane: its four columns x K).  Replaces four launches per row chunk --
// the K-wide forward GEMM, k_wide_head, the K-wide dH and dW GEMMs -- which stream a rows x W activation through 128-wide MFMA
// tiles for ~3 % of a B4 gradient's FLOPs (43 of 283 ms).  part[(e * nblk + blk) * (W K + K + 1)]: dW [W][K], db [K], ll;
// k_wide_headblock_reduce sums the blocks in fixed order (deterministic).
// K and "W == 256" are template parameters: with run-time bounds every class / column test became a branch in the row loop
// (~700 instructions per row, most of them control flow); the loop is now straight-line code.
#define WH_KMAX 8
template <int K, bool WFULL>
__global__ __launch_bounds__(256) void k_wide_headblock(const float *H, long long sH, int ldh, int W, const float *theta, long long d, int w_off,
                                                        int b_off, const void *y, long long r0, int R, int task, int act, float *dZ,
                                                        long long sZ, int ldz, float *part, int rows_per_wg) {
  static_assert(K >= 1 && K <= WH_KMAX, "outputs");
  __shared__ float red[4][256 * WH_KMAX];
  __shared__ float redb[4][WH_KMAX + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, e = blockIdx.y, blk = blockIdx.x;
  const float *Hh = H + (size_t)e * sH;
  float *dz = dZ + (size_t)e * sZ;
  const float *Wk = theta + (size_t)e * d + w_off, *bk = theta + (size_t)e * d + b_off;
  float w[4][K], gW[4][K], gb[K], bias[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    bias[k] = bk[k];
    gb[k] = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = lane + 64 * j;
      w[j][k] = (WFULL || c < W) ? Wk[(size_t)c * K + k] : 0.0f;
      gW[j][k] = 0.0f;
    }
  }
  float ll = 0.0f;
  const int rbeg = blk * rows_per_wg, rend = min(R, rbeg + rows_per_wg);
  float hn[4];
  auto load_row = [&](const int r, float (&h)[4]) {   // beyond the block: the last row again (never used)
    const size_t ro = (size_t)min(r, rend - 1) * ldh;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = lane + 64 * j;
      h[j] = (WFULL || c < W) ? Hh[ro + c] : 0.0f;
    }
  };
  load_row(rbeg + wave, hn);
  for (int r = rbeg + wave; r < rend; r += 4) {
    float h[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = hn[j];
    load_row(r + 4, hn);                                    // next row of this wave in flight
    // out[k] is wave-uniform after the sum; class k's logit goes to LANE k of `o`, so that the softmax costs two expf per row and
    // wave instead of two per class (uniform values still execute on the vector ALU: 14 redundant expf were most of this kernel)
    float out01[2] = {0.0f, 0.0f};
    int o_bits = __float_as_int(-INFINITY);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float t = h[0] * w[0][k];
#pragma unroll
      for (int j = 1; j < 4; ++j) t = fmaf(h[j], w[j][k], t);
      const float tot = wave_sum(t) + bias[k];
      if (k < 2) out01[k] = tot;
      o_bits = lane == k ? __float_as_int(tot) : o_bits;   // (lane == k) is loop-invariant: one v_cndmask per class
    }
    const float o = __int_as_float(o_bits);
    float dout[K];
#pragma unroll
    for (int k = 0; k < K; ++k) dout[k] = 0.0f;
    if (task == MILE_TASK_REGRESSION) {
      float dmu, ds;
      ll += row_loss_regr(out01[0], out01[1], ((const float *)y)[r0 + r], dmu, ds);
      dout[0] = dmu;
      if constexpr (K > 1) dout[1] = ds;
    } else {
      const int yi = __builtin_amdgcn_readfirstlane(((const int32_t *)y)[r0 + r]);
      auto first8 = [](float v, auto op) {                  // op-reduction over lanes 0..7, result read from lane 0 (uniform)
        v = op(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false)));    // quad_perm [1,0,3,2]
        v = op(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false)));    // quad_perm [2,3,0,1]
        v = op(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false)));   // row_half_mirror
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
      };
      const float m = first8(o, [](float a, float b) { return fmaxf(a, b); });                 // lanes >= K hold -inf
      const float se = first8(lane < K ? expf(o - m) : 0.0f, [](float a, float b) { return a + b; });
      const float lse = m + logf(se);
      const float l1 = __int_as_float(__builtin_amdgcn_readlane(o_bits, yi)) - lse;
      const bool bad = isnan(l1);
      const float dl = (bad || lane >= K) ? 0.0f : ((lane == yi ? 1.0f : 0.0f) - expf(o - lse));
#pragma unroll
      for (int k = 0; k < K; ++k) dout[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dl), k));
      ll += bad ? 0.0f : l1;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) gb[k] += dout[k];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = lane + 64 * j;
      float t = 0.0f;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        t = fmaf(dout[k], w[j][k], t);
        gW[j][k] = fmaf(h[j], dout[k], gW[j][k]);
      }
      if (WFULL || c < W) dz[(size_t)r * ldz + c] = t * act_bwd(act, h[j]);
    }
  }
  // the four waves' sums (each over its rows) -> one, through LDS, fixed order
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < K; ++k) red[wave][(lane + 64 * j) * WH_KMAX + k] = gW[j][k];
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) redb[wave][k] = gb[k];
    redb[wave][WH_KMAX] = ll;
  }
  __syncthreads();
  float *dst = part + ((size_t)e * gridDim.x + blk) * ((size_t)W * K + K + 1);
  for (int i = tid; i < W * K; i += 256) {
    const int c = i / K, k = i - c * K;
    dst[i] = (red[0][c * WH_KMAX + k] + red[1][c * WH_KMAX + k]) + (red[2][c * WH_KMAX + k] + red[3][c * WH_KMAX + k]);
  }
  if (tid <= K) {
    const int k = tid < K ? tid : WH_KMAX;
    dst[W * K + tid] = (redb[0][k] + redb[1][k]) + (redb[2][k] + redb[3][k]);
  }
}
// slab[e][w_off + i] (+)= sum_blk part, slab[e][b_off + k] (+)=, llacc[e] (=|+=)
__global__ __launch_bounds__(256) void k_wide_headblock_reduce(const float *part, int nblk, int WK, int K, float *slab, long long dp, int w_off,
                                                               int b_off, float *llacc, int accumulate) {
  const int e = blockIdx.y, per = WK + K + 1;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < per; i += gridDim.x * 256) {
    float s = 0.0f;
    for (int b = 0; b < nblk; ++b) s += part[((size_t)e * nblk + b) * per + i];
    float *o = i < WK ? slab + (size_t)e * dp + w_off + i : (i < WK + K ? slab + (size_t)e * dp + b_off + (i - WK) : llacc + e);
    *o = accumulate ? *o + s : s;
  }
}

// ---- evaluation: per-row log-likelihood of the head outputs out [S][R][ld] -> out_ll[(s0 + s) * N + r0 + r] ------------
// (no nansum zeroing: src/inference/metrics.py:247-294 uses the distributions' log_prob directly)
__global__ __launch_bounds__(256) void k_wide_rowll(const float *out, long long sOut, int ld, const void *y, long long r0, int R, int K, int task,
                                                    float *out_ll, long long N, long long s0) {
  const int s = blockIdx.y;
  const float *o = out + (size_t)s * sOut;
  for (int rr = blockIdx.x * 256 + threadIdx.x; rr < R; rr += gridDim.x * 256) {
    const float *z = o + (size_t)rr * ld;
    float v;
    if (task == MILE_TASK_REGRESSION) {
      const float es = expf(z[1]);
      const float sig = isnan(es) ? es : fminf(fmaxf(es, 1e-6f), 1e6f);
      const float q = (((const float *)y)[r0 + rr] - z[0]) / sig;
      v = -0.5f * q * q - logf(sig) - 0.91893853320467274f;
    } else {
      const int yi = ((const int32_t *)y)[r0 + rr];
      float m = z[0];
      for (int c = 1; c < K; ++c) m = fmaxf(m, z[c]);
      float se = 0.0f;
      for (int c = 0; c < K; ++c) se += expf(z[c] - m);
      v = z[yi] - (m + logf(se));
    }
    out_ll[(size_t)(s0 + s) * N + r0 + rr] = v;
  }
}
