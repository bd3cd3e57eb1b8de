// bf16 MFMA fragment helpers for gfx950 (v_mfma_f32_32x32x16_bf16) over swizzled LDS images.
//
// An "image" is a [rows][128] bf16 array in LDS with 256-byte rows whose 16-byte chunks are XOR
// swizzled so that BOTH kinds of operand read are bank-conflict free:
//   * row reads (ds_read_b128): a lane takes 8 consecutive columns of one row;
//   * transposed reads (ds_read_b64_tr_b16): a lane takes 8 consecutive ROWS of one column.
// Operand maps (lane l: r = l & 31, h = l >> 5; fragment element j = 0..7):
//   A[m = r][k = 8h + j],  B[k = 8h + j][n = r],  D[m = (reg&3) + 8(reg>>2) + 4h][n = r].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MILE_LDS_PTR(T, p) ((__attribute__((address_space(3))) T *)(p))

// byte offset of 16-byte chunk `ch` (0..15) of row `row` inside an image
__device__ __forceinline__ int img_off(int row, int ch) {
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

// 8 consecutive columns [8*ch, 8*ch+8) of `row`
__device__ __forceinline__ bf16x8 row_frag(const char *img, int row, int ch) {
  return *reinterpret_cast<const bf16x8 *>(img + img_off(row, ch));
}

// Transposed fragment: element j of lane (r, h) = img[row0 + 8h + j][col0 + r], for a 32-column
// block starting at col0 (multiple of 32) and 16 rows starting at row0 (multiple of 16).
// ds_read_b64_tr_b16 works per 16-lane group: lane 4q+p of the group supplies the address of row q,
// columns 4p..4p+3 of a 4 x 16 block and receives column (lane & 15) of the 4 rows.  All 64 lanes
// must be active.
__device__ __forceinline__ bf16x8 tr_frag(const char *img, int row0, int col0, int lane) {
  const int h = lane >> 5, g1 = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
  const int row = row0 + 8 * h + q;
  const int ch = (col0 >> 3) + 2 * g1 + (p >> 1);
  const char *a0 = img + img_off(row, ch) + 8 * (p & 1);
  const char *a1 = img + img_off(row + 4, ch) + 8 * (p & 1);
  const bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(MILE_LDS_PTR(bf16x4, a0));
  const bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(MILE_LDS_PTR(bf16x4, a1));
  return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Store a 32x32 accumulator tile D[m = feature][n = row] as img[row = n][col0 + m] (bf16).
__device__ __forceinline__ void store_tile(char *img, int col0, const f32x16 &acc, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4_t v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    const bf16x4 b = __builtin_convertvector(v, bf16x4);
    *reinterpret_cast<bf16x4 *>(img + img_off(r, (col0 >> 3) + g) + 8 * h) = b;
  }
}

// ---- padded images (k_grad_w128b) ---------------------------------------------------------------------------------------
// Same two kinds of read, but every address is (one per-lane base) + (a compile-time constant), so the constant rides in the
// ds instruction's offset field: the XOR swizzle above needs a separate address register for every (k-step, column block)
// pair -- ~30 registers of addresses in k_grad_w128b, which the compiler parked in AGPRs / scratch and rebuilt with ~280
// v_add / v_accvgpr_read per tile pair.  Rows are 272 bytes (256 + one 16-byte chunk of padding) and logical row i of each
// 16-row group sits at physical row pim_row(i): i = 8h + 4t + q  ->  4q + 2h + t.
//   * row reads (ds_read_b128, 16-lane service groups): 272 = 68 dwords = 4 mod 64, and the 16 lanes of a group hold 16
//     distinct physical rows mod 16 -> 16 x 4 distinct banks;
//   * transposed reads (ds_read_b64_tr_b16, 32-lane groups): the 4 rows one read covers (q = 0..3 at fixed h, t) are
//     4 physical rows apart = 16 dwords mod 64 apart, each 16 dwords wide -> 64 distinct banks; the two reads of a
//     fragment (t = 0, 1) are adjacent physical rows.
#define PIM_STRIDE 272
typedef uint32_t u32x4_lab __attribute__((ext_vector_type(4)));
// 16-byte row read of a padded image
__device__ __forceinline__ bf16x8 pim_row_read(const char *a) {
#ifdef MILE_LAB_NO_ROW   // dev experiment (tools/r03/lab)
  const u32x4_lab v = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};   // one hoisted constant: no VALU added
  return __builtin_bit_cast(bf16x8, v);
#endif
  return *reinterpret_cast<const bf16x8 *>(a);
}
__device__ __forceinline__ int pim_row(int i) { return (i & ~15) | ((i & 3) << 2) | (((i >> 3) & 1) << 1) | ((i >> 2) & 1); }
__device__ __forceinline__ int pim_off(int row, int ch) { return PIM_STRIDE * pim_row(row) + 16 * ch; }
// per-lane base of a transposed read (add STRIDE * row0 + 2 * col0, row0 a multiple of 16, col0 of 32).  STRIDE: any row
// pitch that is 16 mod 64 dwords per 4 rows, e.g. 272 bytes for 128-column images, 144 for 64-column ones.
template <int STRIDE = PIM_STRIDE>
__device__ __forceinline__ int pim_tr_base(int lane) {
  static_assert((STRIDE % 16) == 0 && ((4 * STRIDE / 4) % 64) == 16, "padded image pitch");
  const int h = lane >> 5, g1 = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
  return STRIDE * (4 * q + 2 * h) + 32 * g1 + 16 * (p >> 1) + 8 * (p & 1);
}
// element j of lane (r, h) = image[row0 + 8h + j][col0 + r]; `a` = image + pim_tr_base(lane) + STRIDE * row0 + 2 * col0
template <int STRIDE = PIM_STRIDE>
__device__ __forceinline__ bf16x8 pim_tr_frag(const char *a) {
#ifdef MILE_LAB_NO_TR   // dev experiment (tools/r03/lab)
  const u32x4_lab v = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};   // one hoisted constant: no VALU added
  return __builtin_bit_cast(bf16x8, v);
#endif
  const bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(MILE_LDS_PTR(bf16x4, a));
  const bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(MILE_LDS_PTR(bf16x4, a + STRIDE));
  return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
}
// two fp32 -> one packed bf16 pair (round to nearest even) in ONE v_cvt_pk_bf16_f32: the 2-vector conversion selects the
// packed instruction, the 4-vector one lowers to one conversion per element plus a v_perm_b32 per pair.  (Not inline asm:
// the operands are usually fresh MFMA results, and the MFMA -> VALU wait states are inserted by the compiler only for
// instructions it can see.)
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ f32x16 mfma_bf16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
#ifdef MILE_LAB_NO_MFMA   // dev experiment (tools/r03/lab): what the kernel costs without its matrix products
  f32x16 o = c;
  o[0] += (float)a[0] + (float)b[0];
  return o;
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}

__device__ __forceinline__ float bf16_colsum(const bf16x8 v, float acc) {   // acc + sum of the 8 elements
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bf16x2 pr = {v[2 * i], v[2 * i + 1]}, one = {(bf16)1.0f, (bf16)1.0f};
    acc = __builtin_amdgcn_fdot2_f32_bf16(pr, one, acc, false);
  }
  return acc;
}

// feature index (within a 32-block) that accumulator register j of lane half h holds
__device__ __forceinline__ int acc_m(int j, int h) { return (j & 3) + 8 * (j >> 2) + 4 * h; }
