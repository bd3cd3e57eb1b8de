// LeNet convolutions on the gfx950 matrix pipe (MILE_GRAD_LENET_BF16; BASELINE config 5 names bf16):
// implicit-GEMM 5x5 convolutions on v_mfma_f32_16x16x32_bf16, operands rounded to bf16 where they enter a product
// (oracle/lenet_oracle.py logpost_and_grad_bf16), fp32 accumulation.  src/models/images/cnns.py:33-66.
//
// No im2col matrix anywhere: a workgroup = one particle x a range of images keeps one image's input as a bf16 tile in
// LDS, pixel-major with the channels of a pixel contiguous (8 bytes = 4 channels per "slot": conv1 has <= 4 input
// channels = 1 slot per pixel, conv2 has 6 -> 2 slots, the dZ tile of the input-gradient pass has 16 -> 4 slots).
// A kernel tap and a slot's channels are one 8-byte LDS read at (pixel base + tap offset); the GEMM K index runs over
// (tap, channel) in units of slots, so the "patch matrix" is only ever a set of addresses.
//   forward / input gradient ("F" form):  D[m][pixel] = sum_{slot, j} Kop[m][slot][j] * tile[pixel + off(slot)][j]
//       A = the particle's kernel (rows m = output channel, or input channel for the input gradient), held in registers
//       for the whole workgroup; B = 8-byte tile reads; 16 pixels per MFMA, K = 32 = 8 slots.
//   kernel gradient ("W" form):  dK[slot][j][co] = sum_{pixel} tile[pixel + off(slot)][j] * dz[pixel][co]
//       both operands contract over pixels, i.e. are read transposed: ds_read_b64_tr_b16 (a 16-lane group hands in the
//       addresses of 4 pixels x 4 slots and gets back, per lane, one (slot, channel) row of 4 pixels); 32 pixels per
//       MFMA, 16 kernel rows (4 slots) x 16 output channels per accumulator; bias gradient = a ones row.
// MFMA operand maps (cdna_hip_programming.md section 3): lane l holds A[row l & 15][k = 8 (l >> 4) + j],
// B[k = 8 (l >> 4) + j][col l & 15], D[row 4 (l >> 4) + reg][col l & 15].
// Everything that is not a convolution product (bias, activation and its derivative, pooling, the Dense layers, the
// likelihood) is the fp32 code of mile_lenet.h; dZ = unpool(dP) * act'(A) is formed while staging, as there.
#pragma once
#include <algorithm>
#include <type_traits>

#include "mile_bf16_frag.h"
#include "mile_lenet.h"

typedef float cm_f32x4 __attribute__((ext_vector_type(4)));
typedef float cm_f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t cm_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t cm_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ cm_f32x4 cm_mfma(const bf16x8 a, const bf16x8 b, const cm_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t cm_pack2(float lo, float hi) {   // two floats -> packed bf16 pair, round to nearest even
  const bf16x2 v = {(bf16)lo, (bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}

// ReLU activations kept for the backward pass only as "was it positive": two bytes per element, any non-zero value stays non-zero
// (truncated bf16 with a sticky low bit).  Halves the largest arrays the convolution half moves through HBM.
__device__ __forceinline__ uint32_t cm_sign_pack2(float lo, float hi) {
  const uint32_t a = __float_as_uint(lo), b = __float_as_uint(hi);
  return ((a >> 16) | ((a & 0xffffu) ? 1u : 0u)) | (((b >> 16) | ((b & 0xffffu) ? 1u : 0u)) << 16);
}
// COUT-vector loads of one pixel's activations: fp32 (8- / 16-byte loads) or the two-byte form above
template <int COUT, bool A16>
__device__ __forceinline__ void cm_load_act(const void *a_img, size_t pixel, float (&av)[COUT]) {
  if constexpr (A16) {
    const uint32_t *ap = reinterpret_cast<const uint32_t *>(a_img) + pixel * (COUT / 2);
    if constexpr (COUT % 8 == 0) {
#pragma unroll
      for (int c = 0; c < COUT; c += 8) {
        const cm_u32x4 t = *reinterpret_cast<const cm_u32x4 *>(ap + c / 2);
#pragma unroll
        for (int k = 0; k < 4; ++k) { av[c + 2 * k] = __uint_as_float(t[k] << 16); av[c + 2 * k + 1] = __uint_as_float(t[k] & 0xffff0000u); }
      }
    } else {
#pragma unroll
      for (int c = 0; c < COUT; c += 2) {
        const uint32_t t = ap[c / 2];
        av[c] = __uint_as_float(t << 16); av[c + 1] = __uint_as_float(t & 0xffff0000u);
      }
    }
  } else {
    const float *ap = reinterpret_cast<const float *>(a_img) + pixel * COUT;
    if constexpr (COUT % 4 == 0) {
#pragma unroll
      for (int c = 0; c < COUT; c += 4) {
        const cm_f32x4 t = *reinterpret_cast<const cm_f32x4 *>(ap + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) av[c + k] = t[k];
      }
    } else {
#pragma unroll
      for (int c = 0; c < COUT; c += 2) {
        const cm_f32x2 t = *reinterpret_cast<const cm_f32x2 *>(ap + c);
        av[c] = t[0]; av[c + 1] = t[1];
      }
    }
  }
}

// p / w for 0 <= p < 2^20 with inv = 1.0f / w (exact: the quotient's distance from an integer is >= 1 / (2 w) >> the float error)
__device__ __forceinline__ int cm_div(int p, float inv) { return (int)(((float)p + 0.5f) * inv); }

// Slot geometry.  PB = bytes per pixel of the tile; off(s, rowpix) = byte offset of slot s from the pixel base (rowpix =
// pixels per tile row); tap(s) / c0(s) = kernel tap and first channel the slot stands for; NS slots in all.
enum { CM_IN4 = 0, CM_IN8 = 1, CM_DZ16 = 2, CM_DZ16P = 3 };
template <int MODE> struct CSlot;
template <> struct CSlot<CM_IN4> {   // <= 4 input channels: slot = (kh, kw) with kw padded to 6 (kw = 5 is a zero tap)
  static constexpr int NS = 30, PB = 8;
  __device__ static int off(int s, int rowpix) { return ((s / 6) * rowpix + s % 6) * 8; }
  __device__ static bool valid(int s) { return s < NS && s % 6 < 5; }
  __device__ static int tap(int s) { return (s / 6) * 5 + s % 6; }
  __device__ static int c0(int) { return 0; }
};
template <> struct CSlot<CM_IN8> {   // <= 8 input channels: slot = (tap, half)
  static constexpr int NS = 50, PB = 16;
  __device__ static int off(int s, int rowpix) { return (((s >> 1) / 5) * rowpix + (s >> 1) % 5) * 16 + 8 * (s & 1); }
  __device__ static bool valid(int s) { return s < NS; }
  __device__ static int tap(int s) { return s >> 1; }
  __device__ static int c0(int s) { return 4 * (s & 1); }
};
template <> struct CSlot<CM_DZ16> {  // the dZ tile of the input-gradient pass, <= 16 channels: slot = (tap, quarter), taps look BACK
  static constexpr int NS = 100, PB = 32;
  __device__ static int off(int s, int rowpix) { return -(((s >> 2) / 5) * rowpix + (s >> 2) % 5) * 32 + 8 * (s & 3); }
  __device__ static bool valid(int s) { return s < NS; }
  __device__ static int tap(int s) { return s >> 2; }
  __device__ static int c0(int s) { return 4 * (s & 3); }
};

template <> struct CSlot<CM_DZ16P> { // dZ tile, pair form: slot = (kh, kw' = -1..4, quarter)
  static constexpr int NS = 120, PB = 32;
  __device__ static int off(int s, int rowpix) { return -((s / 24) * rowpix + ((s % 24) >> 2) - 1) * 32 + 8 * (s & 3); }
  __device__ static bool valid(int s) { return s < NS; }
  __device__ static int tap(int) { return 0; }     // unused: the pair forms index the kernel themselves
  __device__ static int c0(int s) { return 4 * (s & 3); }
};

// zero-padded bf16 input tile [Hp][Wp][PB / 2 channels] (+ 8 pixels of zero slack: the padded taps read past the last row).
// U pixels per thread and pass with all their global loads issued before the first conversion: as a plain loop every
// iteration waited out its own loads' latency (16 serial HBM round trips per image were most of the kernels' time).
template <int MODE>
__device__ __forceinline__ void cm_stage_input(char *tile, const float *src, long long sH, long long sW, long long sC, int CIN, int H, int W,
                                               int pad, int tid) {
  constexpr int CPX = CSlot<MODE>::PB / 2, U = 3;
  const int Hp = H + 2 * pad, Wp = W + 2 * pad, n = Hp * Wp + 8;
  const float inv_wp = 1.0f / (float)Wp;
  const bool nhwc2 = sC == 1 && sW == CIN && (CIN & 1) == 0 && (sH & 1) == 0 && ((uintptr_t)src & 7) == 0;   // channel pairs as 8-byte loads
  for (int i0 = tid; i0 < n; i0 += 256 * U) {
    float v[U][CPX];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + 256 * u;
      const int yy = cm_div(i, inv_wp), xx = i - yy * Wp;
      const int h = yy - pad, w = xx - pad;
      const bool in = i < Hp * Wp && h >= 0 && h < H && w >= 0 && w < W;
      const float *px = src + (in ? h * sH + w * sW : 0);
      if (nhwc2) {
#pragma unroll
        for (int c = 0; c < CPX; c += 2) {
          const cm_f32x2 t = (in && c < CIN) ? *reinterpret_cast<const cm_f32x2 *>(px + c) : cm_f32x2{0.0f, 0.0f};
          v[u][c] = t[0]; v[u][c + 1] = t[1];
        }
      } else {
#pragma unroll
        for (int c = 0; c < CPX; ++c) v[u][c] = (in && c < CIN) ? px[c * sC] : 0.0f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + 256 * u;
      if (i >= n) continue;
      if constexpr (CPX == 4) {
        *reinterpret_cast<cm_u32x2 *>(tile + (size_t)i * 8) = cm_u32x2{cm_pack2(v[u][0], v[u][1]), cm_pack2(v[u][2], v[u][3])};
      } else {
        *reinterpret_cast<cm_u32x4 *>(tile + (size_t)i * 16) =
            cm_u32x4{cm_pack2(v[u][0], v[u][1]), cm_pack2(v[u][2], v[u][3]), cm_pack2(v[u][4], v[u][5]), cm_pack2(v[u][6], v[u][7])};
      }
    }
  }
}

// The particle's kernel as the A operand of the F form: ka[c] = rows m (lane & 15), k-blocks 4c + (lane >> 4) = slots 2 blk, 2 blk + 1.
// DX = false: rows are output channels, K[(tap * CIN + ci) * COUT + m];  DX = true: rows are input channels, K[(tap * CIN + m) * COUT + co].
// K is read through an LDS copy (Ksh, 25 * CIN * COUT floats, aliasing the image tile): the gather below is 8 NMF scalar reads per
// lane, which as global loads cost the workgroup more than the images it then processes.  Ends with a barrier-free state: the
// caller's image loop starts with __syncthreads() before it overwrites the tile.
// KIND: CK_FWD rows = output channels; CK_DX rows = input channels; CK_FWD2 / CK_DX2 the PAIR forms: row m = 2 * channel + dxo
// stands for the output pixel x + dxo of a pixel PAIR (x even), i.e. the kernel shifted by dxo inside a 6-wide kw window:
//   out[x + dxo][co]  = sum_{kw'} in[x + kw'] K[kw' - dxo][co]            (kw' = 0..5, the padded slots of CM_IN4)
//   din[x + dxo][ci]  = sum_{kw'} dz[x - kw'] K[kw' + dxo][ci]            (kw' = -1..4, CM_DZ16P)
// -- twice the useful MFMA rows (12 of 16 instead of 6) and half the tiles for the 6-channel sides of the two convolutions.
enum { CK_FWD = 0, CK_DX = 1, CK_FWD2 = 2, CK_DX2 = 3 };
template <int MODE, int NMF, int KIND>
__device__ __forceinline__ void cm_kernel_operand(const float *Kg, float *K, int CIN, int COUT, int tid, bf16x8 (&ka)[NMF]) {
  using G = CSlot<MODE>;
  for (int i = tid; i < 25 * CIN * COUT; i += 256) K[i] = Kg[i];
  __syncthreads();
  const int lane = tid & 63;
  const int m = lane & 15, g = lane >> 4;
#pragma unroll
  for (int c = 0; c < NMF; ++c) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int s = 2 * (4 * c + g) + (j >> 2), ch = G::c0(s) + (j & 3);
      bool ok;
      int idx;
      if constexpr (KIND == CK_DX) { ok = G::valid(s) && m < CIN && ch < COUT; idx = (G::tap(s) * CIN + m) * COUT + ch; }
      else if constexpr (KIND == CK_FWD) { ok = G::valid(s) && ch < CIN && m < COUT; idx = (G::tap(s) * CIN + ch) * COUT + m; }
      else if constexpr (KIND == CK_FWD2) {   // CM_IN4 slots: s = 6 kh + kw'
        const int kh = s / 6, kw = s % 6 - (m & 1), co = m >> 1;
        ok = s < G::NS && kw >= 0 && kw < 5 && ch < CIN && co < COUT;
        idx = ((kh * 5 + kw) * CIN + ch) * COUT + co;
      } else {                                // CM_DZ16P slots: s = 24 kh + 4 (kw' + 1) + quarter
        const int kh = s / 24, kw = ((s % 24) >> 2) - 1 + (m & 1), ci = m >> 1;
        ok = s < G::NS && kw >= 0 && kw < 5 && ci < CIN && ch < COUT;
        idx = ((kh * 5 + kw) * CIN + ci) * COUT + ch;
      }
      v[j] = ok ? K[idx] : 0.0f;
    }
    const cm_u32x4 pk = {cm_pack2(v[0], v[1]), cm_pack2(v[2], v[3]), cm_pack2(v[4], v[5]), cm_pack2(v[6], v[7])};
    ka[c] = __builtin_bit_cast(bf16x8, pk);
  }
}

// All 16-pixel tiles of one image in the F form.  A wave takes tile PAIRS (t, t + 4), t = wave, wave + 8, ...: two independent
// accumulator chains, and the tile reads of the next group of four k-blocks -- or of the next pair -- are in flight while the
// current group's MFMAs run (register double buffer, parities resolved at compile time).  Left to the compiler the loop was
// read -> s_waitcnt lgkmcnt(0) -> MFMA per k-block through ONE fragment register set: an LDS round trip per MFMA.
// base_of(t) = byte offset of this lane's pixel of tile t in the LDS image; epi(t, acc) stores one tile (all lanes call it).
template <int NMF, class BaseFn, class EpiFn>
__device__ __forceinline__ void cm_image_f(const char *tile, int ntiles, int wave, const int (&so)[NMF][2], const bf16x8 (&ka)[NMF],
                                           BaseFn base_of, EpiFn epi) {
  constexpr int NG = (NMF + 3) / 4;
  cm_u32x2 lo[2][2][4], hi[2][2][4];                    // [buffer][tile of the pair][k-block of the group]
  auto load = [&](const int bA, const int bB, const int grp, cm_u32x2 (&l)[2][4], cm_u32x2 (&h)[2][4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = 4 * grp + u;
      if (c < NMF) {
        l[0][u] = *reinterpret_cast<const cm_u32x2 *>(tile + bA + so[c][0]);
        h[0][u] = *reinterpret_cast<const cm_u32x2 *>(tile + bA + so[c][1]);
        l[1][u] = *reinterpret_cast<const cm_u32x2 *>(tile + bB + so[c][0]);
        h[1][u] = *reinterpret_cast<const cm_u32x2 *>(tile + bB + so[c][1]);
      }
    }
  };
  int t0 = wave;
  if (t0 >= ntiles) return;
  int bA = base_of(t0), bB = base_of(min(t0 + 4, ntiles - 1));
  load(bA, bB, 0, lo[0], hi[0]);
  bool more = true;
  auto pair = [&](auto par_c) {
    constexpr int PAR = decltype(par_c)::value;
    const int nt0 = t0 + 8;
    more = nt0 < ntiles;
    int nA = 0, nB = 0;
    if (more) { nA = base_of(nt0); nB = base_of(min(nt0 + 4, ntiles - 1)); }
    cm_f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
      const int cur = (PAR + grp) & 1, nxt = cur ^ 1;
      if (grp + 1 < NG) load(bA, bB, grp + 1, lo[nxt], hi[nxt]);
      else if (more) load(nA, nB, 0, lo[nxt], hi[nxt]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = 4 * grp + u;
        if (c < NMF) {
          const cm_u32x4 f0 = {lo[cur][0][u][0], lo[cur][0][u][1], hi[cur][0][u][0], hi[cur][0][u][1]};
          const cm_u32x4 f1 = {lo[cur][1][u][0], lo[cur][1][u][1], hi[cur][1][u][0], hi[cur][1][u][1]};
          acc0 = cm_mfma(ka[c], __builtin_bit_cast(bf16x8, f0), acc0);
          acc1 = cm_mfma(ka[c], __builtin_bit_cast(bf16x8, f1), acc1);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    epi(t0, acc0);
    if (t0 + 4 < ntiles) epi(t0 + 4, acc1);         // wave-uniform
    bA = nA; bB = nB; t0 = nt0;
  };
  while (more) {
    pair(std::integral_constant<int, 0>{});
    if constexpr (NG % 2 == 1) {
      if (!more) break;
      pair(std::integral_constant<int, 1>{});
    }
  }
}

// out[e][b][y][x][co] = act(bias[co] + sum in[b][y+kh-pad][x+kw-pad][ci] K[kh][kw][ci][co]), NHWC fp32 (as k_conv5_fwd), and
// pool[e][b][y/2][x/2][co] = its 2 x 2 average (avg_pool VALID, k_avgpool2) from the same registers: an MFMA tile is a 2 x 8 block
// of output pixels (lane n: row n >> 3, column n & 7), so a pooling window is lanes {n, n^1, n^8, n^9} of one 16-lane group.
// out may be null (evaluation needs the pooled activations only).
template <int MODE, int COUT>
__global__ __launch_bounds__(256) void k_conv5m_fwd(const float *in, long long sE, long long sB, long long sH, long long sW, long long sC,
                                                    int CIN, int H, int W, int pad, const float *theta, int k_off, int b_off, int d, float *out,
                                                    float *pool, int R, int ipw, int activation, int dbg = 0, int a16 = 0, int ni = 1) {
  // ni images are staged per barrier pair (small images: conv2's 144 output pixels are 12 tiles -- three tiles per wave in two pair
  // rounds; four images at once fill the rounds and quarter the barriers).
  // dbg (MILE_CM_SKIP, timing experiments only -- results are wrong): 1 no full-size store, 2 no pooled store, 4 no MFMA loop, 8 no staging
  using G = CSlot<MODE>;
  constexpr int NMF = ((G::NS + 1) / 2 + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) char cm_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, e = blockIdx.y;
  const int n16 = lane & 15, g = lane >> 4;
  const int Ho = H + 2 * pad - 4, Wo = W + 2 * pad - 4, Wp = W + 2 * pad, npix = Ho * Wo;
  const int Hq = Ho / 2, Wq = Wo / 2, TX = (Wo + 7) / 8, ntiles = ((Ho + 1) / 2) * TX;
  bf16x8 ka[NMF];
  cm_kernel_operand<MODE, NMF, CK_FWD>(theta + (size_t)e * d + k_off, reinterpret_cast<float *>(cm_lds), CIN, COUT, tid, ka);
  int so[NMF][2];
#pragma unroll
  for (int c = 0; c < NMF; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int s = 2 * (4 * c + g) + u;
      so[c][u] = s < G::NS ? G::off(s, Wp) : 0;
    }
  float bias4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bias4[i] = 4 * g + i < COUT ? theta[(size_t)e * d + b_off + 4 * g + i] : 0.0f;
  const float inv_tx = 1.0f / (float)TX, inv_nt = 1.0f / (float)ntiles;
  const int dy = n16 >> 3, dx = n16 & 7;
  const int tile_bytes = ((H + 2 * pad) * Wp + 8) * G::PB;              // one image's tile (a multiple of 16 bytes)
  const int b0 = blockIdx.x * ipw, b1 = min(R, b0 + ipw);
  for (int b = b0; b < b1; b += ni) {
    const int nimg = min(ni, b1 - b);
    __syncthreads();
    if (!(dbg & 8) || b == b0)
      for (int im = 0; im < nimg; ++im)
        cm_stage_input<MODE>(cm_lds + im * tile_bytes, in + (size_t)e * sE + (size_t)(b + im) * sB, sH, sW, sC, CIN, H, W, pad, tid);
    __syncthreads();
    float *dst = out && !(dbg & 1) ? (a16 ? (float *)((uint16_t *)out + ((size_t)e * R + b) * npix * COUT) : out + ((size_t)e * R + b) * npix * COUT) : nullptr;
    float *pdst = pool + ((size_t)e * R + b) * Hq * Wq * COUT;
    if (dbg & 4) continue;
    cm_image_f<NMF>(cm_lds, nimg * ntiles, wave, so, ka,
      [&](const int tt) {
        const int im = cm_div(tt, inv_nt), t = tt - im * ntiles;
        const int ty = cm_div(t, inv_tx), tx = t - ty * TX;
        const int y = min(2 * ty + dy, Ho - 1), x = min(8 * tx + dx, Wo - 1);
        return im * tile_bytes + (y * Wp + x) * G::PB;
      },
      [&](const int tt, const cm_f32x4 acc) {
        const int im = cm_div(tt, inv_nt), t = tt - im * ntiles;
        const int ty = cm_div(t, inv_tx), tx = t - ty * TX;
        const int y = 2 * ty + dy, x = 8 * tx + dx;
        float *dst_i = dst ? (a16 ? (float *)((uint16_t *)dst + (size_t)im * npix * COUT) : dst + (size_t)im * npix * COUT) : nullptr;
        float *pdst_i = pdst + (size_t)im * Hq * Wq * COUT;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = act_fwd(activation, acc[i] + bias4[i]);
        if (dst_i && a16 && y < Ho && x < Wo && 4 * g < COUT) {          // ReLU: the backward pass needs only "was it positive"
          uint16_t *o = (uint16_t *)dst_i + (size_t)(y * Wo + x) * COUT + 4 * g;
          if constexpr (COUT % 4 == 0) {
            *reinterpret_cast<cm_u32x2 *>(o) = cm_u32x2{cm_sign_pack2(v[0], v[1]), cm_sign_pack2(v[2], v[3])};
          } else {
            *reinterpret_cast<uint32_t *>(o) = cm_sign_pack2(v[0], v[1]);
            if (4 * g + 2 < COUT) *reinterpret_cast<uint32_t *>(o + 2) = cm_sign_pack2(v[2], v[3]);
          }
        } else if (dst_i && y < Ho && x < Wo && 4 * g < COUT) {
          float *o = dst_i + (size_t)(y * Wo + x) * COUT + 4 * g;
          if constexpr (COUT % 4 == 0) {
            *reinterpret_cast<cm_f32x4 *>(o) = cm_f32x4{v[0], v[1], v[2], v[3]};
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (4 * g + i < COUT) o[i] = v[i];
          }
        }
        float s4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {            // window sum in the order of k_avgpool2: (a00 + a01) + (a10 + a11)
          const float h2 = v[i] + __shfl_xor(v[i], 1);
          s4[i] = 0.25f * (h2 + __shfl_xor(h2, 8));
        }
        if (!(dbg & 2) && dy == 0 && (dx & 1) == 0 && y < 2 * Hq && x < 2 * Wq && 4 * g < COUT) {
          float *o = pdst_i + (size_t)((y >> 1) * Wq + (x >> 1)) * COUT + 4 * g;
          if constexpr (COUT % 4 == 0) {
            *reinterpret_cast<cm_f32x4 *>(o) = cm_f32x4{s4[0], s4[1], s4[2], s4[3]};
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (4 * g + i < COUT) o[i] = s4[i];
          }
        }
      });
  }
}

// The forward convolution in the PAIR form (CK_FWD2; <= 4 input channels, 2 COUT <= 16): an MFMA column is the pixel pair
// (y, x), (y, x + 1) with x even, rows are (output channel, which pixel of the pair).  A tile = 2 rows x 8 pairs = 2 x 16 pixels:
// the horizontal half of a pooling window sits in ONE lane (registers i, i + 1), the vertical half in lane n ^ 8.
template <int COUT>
__global__ __launch_bounds__(256) void k_conv5m_fwd2x(const float *in, long long sE, long long sB, long long sH, long long sW, long long sC,
                                                      int CIN, int H, int W, int pad, const float *theta, int k_off, int b_off, int d, float *out,
                                                      float *pool, int R, int ipw, int activation, int a16 = 0, int ni = 1) {
  using G = CSlot<CM_IN4>;
  static_assert(2 * COUT <= 16 && COUT % 2 == 0, "rows = (channel, pixel of the pair)");
  constexpr int NMF = ((G::NS + 1) / 2 + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) char cm_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, e = blockIdx.y;
  const int n16 = lane & 15, g = lane >> 4;
  const int Ho = H + 2 * pad - 4, Wo = W + 2 * pad - 4, Wp = W + 2 * pad, npix = Ho * Wo;
  const int Hq = Ho / 2, Wq = Wo / 2, TX = (Wo + 15) / 16, ntiles = ((Ho + 1) / 2) * TX;
  bf16x8 ka[NMF];
  cm_kernel_operand<CM_IN4, NMF, CK_FWD2>(theta + (size_t)e * d + k_off, reinterpret_cast<float *>(cm_lds), CIN, COUT, tid, ka);
  int so[NMF][2];
#pragma unroll
  for (int c = 0; c < NMF; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int s = 2 * (4 * c + g) + u;
      so[c][u] = s < G::NS ? G::off(s, Wp) : 0;
    }
  const bool rows_on = 2 * g < COUT;                 // this lane's rows are channels 2 g, 2 g + 1
  const float bias0 = rows_on ? theta[(size_t)e * d + b_off + 2 * g] : 0.0f, bias1 = rows_on ? theta[(size_t)e * d + b_off + 2 * g + 1] : 0.0f;
  const float inv_tx = 1.0f / (float)TX, inv_nt = 1.0f / (float)ntiles;
  const int dy = n16 >> 3, px = n16 & 7;
  const int tile_bytes = ((H + 2 * pad) * Wp + 8) * G::PB;               // per image (ni images per barrier pair)
  const int b0 = blockIdx.x * ipw, b1 = min(R, b0 + ipw);
  for (int b = b0; b < b1; b += ni) {
    const int nimg = min(ni, b1 - b);
    __syncthreads();
    for (int im = 0; im < nimg; ++im)
      cm_stage_input<CM_IN4>(cm_lds + im * tile_bytes, in + (size_t)e * sE + (size_t)(b + im) * sB, sH, sW, sC, CIN, H, W, pad, tid);
    __syncthreads();
    float *dst0 = out ? (a16 ? (float *)((uint16_t *)out + ((size_t)e * R + b) * npix * COUT) : out + ((size_t)e * R + b) * npix * COUT) : nullptr;
    float *pdst0 = pool + ((size_t)e * R + b) * Hq * Wq * COUT;
    cm_image_f<NMF>(cm_lds, nimg * ntiles, wave, so, ka,
      [&](const int tt) {
        const int im = cm_div(tt, inv_nt), t = tt - im * ntiles;
        const int ty = cm_div(t, inv_tx), tx = t - ty * TX;
        const int y = min(2 * ty + dy, Ho - 1), x = min(16 * tx + 2 * px, Wo - 1);
        return im * tile_bytes + (y * Wp + x) * G::PB;
      },
      [&](const int tt, const cm_f32x4 acc) {
        const int im = cm_div(tt, inv_nt), t = tt - im * ntiles;
        const int ty = cm_div(t, inv_tx), tx = t - ty * TX;
        const int y = 2 * ty + dy, x = 16 * tx + 2 * px;
        float *dst = dst0 ? (a16 ? (float *)((uint16_t *)dst0 + (size_t)im * npix * COUT) : dst0 + (size_t)im * npix * COUT) : nullptr;
        float *pdst = pdst0 + (size_t)im * Hq * Wq * COUT;
        // registers: 0 = (channel 2g, x), 1 = (2g, x + 1), 2 = (2g + 1, x), 3 = (2g + 1, x + 1)
        const float v0 = act_fwd(activation, acc[0] + bias0), v1 = act_fwd(activation, acc[1] + bias0);
        const float v2 = act_fwd(activation, acc[2] + bias1), v3 = act_fwd(activation, acc[3] + bias1);
        if (dst && rows_on && y < Ho) {
          if (a16) {                                   // ReLU: the backward pass needs only "was it positive"
            uint16_t *o = (uint16_t *)dst + (size_t)(y * Wo + x) * COUT + 2 * g;
            if (x < Wo) *reinterpret_cast<uint32_t *>(o) = cm_sign_pack2(v0, v2);
            if (x + 1 < Wo) *reinterpret_cast<uint32_t *>(o + COUT) = cm_sign_pack2(v1, v3);
          } else {
            float *o = dst + (size_t)(y * Wo + x) * COUT + 2 * g;
            if (x < Wo) *reinterpret_cast<cm_f32x2 *>(o) = cm_f32x2{v0, v2};
            if (x + 1 < Wo) *reinterpret_cast<cm_f32x2 *>(o + COUT) = cm_f32x2{v1, v3};
          }
        }
        const float h0 = v0 + v1, h1 = v2 + v3;        // window sum in the order of k_avgpool2: (a00 + a01) + (a10 + a11)
        const float s0 = 0.25f * (h0 + __shfl_xor(h0, 8)), s1 = 0.25f * (h1 + __shfl_xor(h1, 8));
        if (rows_on && dy == 0 && y < 2 * Hq && x < 2 * Wq)
          *reinterpret_cast<cm_f32x2 *>(pdst + (size_t)((y >> 1) * Wq + (x >> 1)) * COUT + 2 * g) = cm_f32x2{s0, s1};
      });
  }
}

// bf16 dZ tile of one image with a zero halo: tile[(y + HALO) * Wt + x + HALO][16 channels], Wt = Wo + 2 HALO;
// dz = unpool(dp) * act'(a): dz[y][x][c] = (avg-pool backward: dp[y/2][x/2][c] / 4, zero on the cropped border) * act'(a[y][x][c])
// (mile_lenet.h dz_from_pool).  npix_alloc pixels are written (zeros outside the image).  One pixel per thread and pass, its
// channels as 8- / 16-byte loads, U pixels' loads in flight (see cm_stage_input); a / dp rows are 8-byte aligned (COUT even).
template <int COUT, bool A16 = false>
__device__ __forceinline__ void cm_stage_dz(char *zt, const float *dp_img, const void *a_img, int Ho, int Wo, int halo, int npix_alloc,
                                            int activation, int tid) {
  static_assert(COUT % 2 == 0 && COUT <= 16, "channel pairs");
  constexpr int U = COUT > 8 ? 2 : 4, VW = COUT % 4 == 0 ? 4 : 2;   // 2 x 16 channels x (a, dp) = 64 registers in flight
  const int Wt = Wo + 2 * halo, Ht = Ho + 2 * halo, Hq = Ho / 2, Wq = Wo / 2;
  const float inv_wt = 1.0f / (float)Wt;
  for (int i0 = tid; i0 < npix_alloc; i0 += 256 * U) {
    float av[U][COUT], gv[U][COUT];
    bool in[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int px = i0 + 256 * u;
      const int yy = cm_div(px, inv_wt), xx = px - yy * Wt;
      const int y = yy - halo, x = xx - halo;
      in[u] = px < npix_alloc && yy < Ht && y >= 0 && y < Ho && x >= 0 && x < Wo;
      const bool pin = in[u] && y < 2 * Hq && x < 2 * Wq;
      cm_load_act<COUT, A16>(a_img, (size_t)(in[u] ? y * Wo + x : 0), av[u]);
      const float *gp = dp_img + (size_t)(pin ? (y >> 1) * Wq + (x >> 1) : 0) * COUT;
#pragma unroll
      for (int c = 0; c < COUT; c += VW) {
        if constexpr (VW == 4) {
          const cm_f32x4 tg = pin ? *reinterpret_cast<const cm_f32x4 *>(gp + c) : cm_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int k = 0; k < 4; ++k) gv[u][c + k] = tg[k];
        } else {
          const cm_f32x2 tg = pin ? *reinterpret_cast<const cm_f32x2 *>(gp + c) : cm_f32x2{0.0f, 0.0f};
          gv[u][c] = tg[0]; gv[u][c + 1] = tg[1];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int px = i0 + 256 * u;
      if (px >= npix_alloc) continue;
      uint32_t pk[8];
#pragma unroll
      for (int c = 0; c < 16; c += 2) {
        float v0 = 0.0f, v1 = 0.0f;
        if (c < COUT) {
          v0 = in[u] ? 0.25f * gv[u][c] * act_bwd(activation, av[u][c]) : 0.0f;
          v1 = in[u] ? 0.25f * gv[u][c + 1] * act_bwd(activation, av[u][c + 1]) : 0.0f;
        }
        pk[c >> 1] = cm_pack2(v0, v1);
      }
      *reinterpret_cast<cm_u32x4 *>(zt + (size_t)px * 32) = cm_u32x4{pk[0], pk[1], pk[2], pk[3]};
      *reinterpret_cast<cm_u32x4 *>(zt + (size_t)px * 32 + 16) = cm_u32x4{pk[4], pk[5], pk[6], pk[7]};
    }
  }
}

// VALID 5x5 conv, gradient w.r.t. the input (as k_conv5_dx): din[e][b][yi][xi][ci] = sum_{kh,kw,co} dz[yi-kh][xi-kw][co] K[kh][kw][ci][co]
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void k_conv5m_dx(const float *dp, const float *a, int activation, const float *theta, int k_off, int d,
                                                   float *din, int R, int Ho, int Wo, int ipw, int a16 = 0) {
  using G = CSlot<CM_DZ16>;
  constexpr int NMF = ((G::NS + 1) / 2 + 3) / 4;
  static_assert(COUT <= 16 && CIN <= 16, "one slot quartet / one MFMA row block");
  extern __shared__ __attribute__((aligned(16))) char cm_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, e = blockIdx.y;
  const int n16 = lane & 15, g = lane >> 4;
  const int H = Ho + 4, W = Wo + 4, Ht = Ho + 8, Wt = Wo + 8, npix = H * W;
  bf16x8 ka[NMF];
  cm_kernel_operand<CM_DZ16, NMF, CK_DX>(theta + (size_t)e * d + k_off, reinterpret_cast<float *>(cm_lds), CIN, COUT, tid, ka);
  int so[NMF][2];
#pragma unroll
  for (int c = 0; c < NMF; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int s = 2 * (4 * c + g) + u;
      so[c][u] = s < G::NS ? G::off(s, Wt) : 0;
    }
  const float inv_w = 1.0f / (float)W;
  const int b0 = blockIdx.x * ipw, b1 = min(R, b0 + ipw);
  for (int b = b0; b < b1; ++b) {
    const size_t img = (size_t)e * R + b;
    __syncthreads();
    if (a16) cm_stage_dz<COUT, true>(cm_lds, dp + img * (Ho / 2) * (Wo / 2) * COUT, (const uint16_t *)a + img * Ho * Wo * COUT, Ho, Wo, 4, Ht * Wt, activation, tid);
    else cm_stage_dz<COUT>(cm_lds, dp + img * (Ho / 2) * (Wo / 2) * COUT, a + img * Ho * Wo * COUT, Ho, Wo, 4, Ht * Wt, activation, tid);
    __syncthreads();
    float *dst = din + img * npix * CIN;
    cm_image_f<NMF>(cm_lds, (npix + 15) / 16, wave, so, ka,
      [&](const int t) {
        const int pc = min(t * 16 + n16, npix - 1);
        const int yi = cm_div(pc, inv_w), xi = pc - yi * W;
        return ((yi + 4) * Wt + xi + 4) * 32;
      },
      [&](const int mt, const cm_f32x4 acc) {
        const int p = mt * 16 + n16;
        if (p < npix) {
          float *o = dst + (size_t)p * CIN + 4 * g;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (4 * g + i < CIN) o[i] = acc[i];
        }
      });
  }
}

// The input gradient in the PAIR form (CK_DX2, CM_DZ16P): an MFMA column is the input pixel pair (yi, xi), (yi, xi + 1), rows are
// (input channel, which pixel): 12 of 16 rows useful for conv2's 6 input channels and 15 MFMAs per 32 pixels instead of 26.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void k_conv5m_dx2x(const float *dp, const float *a, int activation, const float *theta, int k_off, int d,
                                                     float *din, int R, int Ho, int Wo, int ipw, int a16 = 0, int ni = 1) {
  using G = CSlot<CM_DZ16P>;
  constexpr int NMF = ((G::NS + 1) / 2 + 3) / 4;
  static_assert(COUT <= 16 && 2 * CIN <= 16 && CIN % 2 == 0, "one slot quartet / rows = (channel, pixel of the pair)");
  extern __shared__ __attribute__((aligned(16))) char cm_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, e = blockIdx.y;
  const int n16 = lane & 15, g = lane >> 4;
  const int H = Ho + 4, W = Wo + 4, Ht = Ho + 8, Wt = Wo + 8, npix = H * W, W2 = (W + 1) / 2, npairs = H * W2;
  bf16x8 ka[NMF];
  cm_kernel_operand<CM_DZ16P, NMF, CK_DX2>(theta + (size_t)e * d + k_off, reinterpret_cast<float *>(cm_lds), CIN, COUT, tid, ka);
  int so[NMF][2];
#pragma unroll
  for (int c = 0; c < NMF; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int s = 2 * (4 * c + g) + u;
      so[c][u] = s < G::NS ? G::off(s, Wt) : 0;
    }
  const float inv_w2 = 1.0f / (float)W2;
  const int ntiles = (npairs + 15) / 16, tile_bytes = (Ht * Wt + 8) * 32;     // per image (ni images per barrier pair)
  const float inv_nt = 1.0f / (float)ntiles;
  const int b0 = blockIdx.x * ipw, b1 = min(R, b0 + ipw);
  for (int b = b0; b < b1; b += ni) {
    const int nimg = min(ni, b1 - b);
    __syncthreads();
    for (int im = 0; im < nimg; ++im) {
      const size_t img = (size_t)e * R + b + im;
      // + 8 pixels of zero slack: the kw' = -1 slot of an odd-width image's last pair reads one pixel past the tile
      if (a16) cm_stage_dz<COUT, true>(cm_lds + im * tile_bytes, dp + img * (Ho / 2) * (Wo / 2) * COUT, (const uint16_t *)a + img * Ho * Wo * COUT, Ho, Wo, 4, Ht * Wt + 8, activation, tid);
      else cm_stage_dz<COUT>(cm_lds + im * tile_bytes, dp + img * (Ho / 2) * (Wo / 2) * COUT, a + img * Ho * Wo * COUT, Ho, Wo, 4, Ht * Wt + 8, activation, tid);
    }
    __syncthreads();
    float *dst0 = din + ((size_t)e * R + b) * npix * CIN;
    cm_image_f<NMF>(cm_lds, nimg * ntiles, wave, so, ka,
      [&](const int tt) {
        const int im = cm_div(tt, inv_nt), t = tt - im * ntiles;
        const int pc = min(t * 16 + n16, npairs - 1);
        const int yi = cm_div(pc, inv_w2), xi = 2 * (pc - yi * W2);
        return im * tile_bytes + ((yi + 4) * Wt + xi + 4) * 32;
      },
      [&](const int tt, const cm_f32x4 acc) {
        const int im = cm_div(tt, inv_nt), t = tt - im * ntiles;
        const int p = t * 16 + n16;
        const int yi = cm_div(p, inv_w2), xi = 2 * (p - yi * W2);
        float *dst = dst0 + (size_t)im * npix * CIN;
        if (p < npairs && 2 * g < CIN) {           // registers: 0 = (channel 2g, xi), 1 = (2g, xi + 1), 2 = (2g + 1, xi), 3 = (2g + 1, xi + 1)
          float *o = dst + (size_t)(yi * W + xi) * CIN + 2 * g;
          *reinterpret_cast<cm_f32x2 *>(o) = cm_f32x2{acc[0], acc[2]};
          if (xi + 1 < W) *reinterpret_cast<cm_f32x2 *>(o + CIN) = cm_f32x2{acc[1], acc[3]};
        }
      });
  }
}

// Kernel / bias gradient of one image range (as k_conv5_dw): part[(e * nwg + wg) * (25*CIN*COUT + COUT) + ...]
//   dK[kh][kw][ci][co] = sum_{b,y,x} in[b][y+kh-pad][x+kw-pad][ci] dz[b][y][x][co],  db[co] = sum dz
template <int MODE, int COUT>
__global__ __launch_bounds__(256) void k_conv5m_dw(const float *in, long long sE, long long sB, long long sH, long long sW, long long sC, int CIN,
                                                   int H, int W, int pad, const float *dp, const float *a, int activation, float *part, int R,
                                                   int ipw, int a16 = 0, int ni = 1) {
  // ni images per barrier pair, their pixels concatenated along the MFMA K index (conv2: 144 pixels = 4.5 chunks of 32 leave the four
  // waves 2 / 1 / 1 / 1 chunks per image; four images = 18 chunks, 5 / 5 / 4 / 4)
  using G = CSlot<MODE>;
  constexpr int NMT = (G::NS + 3) / 4;            // accumulator tiles of 4 slots x 4 channels = 16 kernel rows
  static_assert(COUT <= 16, "one MFMA column block");
  extern __shared__ __attribute__((aligned(16))) char cm_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, e = blockIdx.y;
  const int n16 = lane & 15, g = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
  const int Ho = H + 2 * pad - 4, Wo = W + 2 * pad - 4, Hp = H + 2 * pad, Wp = W + 2 * pad, npix = Ho * Wo;
  const int tile_bytes = ((Hp * Wp + 8) * G::PB + 15) / 16 * 16;   // one image's input tile
  char *tile = cm_lds;                                              // ni tiles
  char *zt = cm_lds + (size_t)ni * tile_bytes;                      // [roundup32(ni * npix)][16] bf16, zero beyond the last image
  int so[NMT];
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) so[mt] = 4 * mt + p4 < G::NS ? G::off(4 * mt + p4, Wp) : 0;
  cm_f32x4 acc[NMT], accb = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) acc[mt] = cm_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const float inv_wo = 1.0f / (float)Wo;
  const cm_u32x4 ones_u = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_u);
  const int b0 = blockIdx.x * ipw, b1 = min(R, b0 + ipw);
  const float inv_np = 1.0f / (float)npix;
  for (int b = b0; b < b1; b += ni) {
    const int nimg = min(ni, b1 - b), ntot = nimg * npix, ntot32 = (ntot + 31) / 32 * 32;
    __syncthreads();
    for (int im = 0; im < nimg; ++im) {
      const size_t img = (size_t)e * R + b + im;
      const int rows = im + 1 < nimg ? npix : npix + ntot32 - ntot;        // the last image also zeroes the rows up to the chunk boundary
      cm_stage_input<MODE>(tile + im * tile_bytes, in + (size_t)e * sE + (size_t)(b + im) * sB, sH, sW, sC, CIN, H, W, pad, tid);
      if (a16) cm_stage_dz<COUT, true>(zt + (size_t)im * npix * 32, dp + img * (Ho / 2) * (Wo / 2) * COUT, (const uint16_t *)a + img * npix * COUT, Ho, Wo, 0, rows, activation, tid);
      else cm_stage_dz<COUT>(zt + (size_t)im * npix * 32, dp + img * (Ho / 2) * (Wo / 2) * COUT, a + img * npix * COUT, Ho, Wo, 0, rows, activation, tid);
    }
    __syncthreads();
    // chunks of 32 pixels: ch = wave, wave + 4, ...  The fragments of the next group of four accumulator tiles -- or of the next
    // chunk's first group and its dZ fragment -- are in flight while the current group's MFMAs run (see cm_image_f).
    constexpr int NG = (NMT + 3) / 4;
    static_assert(NG % 2 == 0, "the register double buffer returns to buffer 0 at every chunk");
    bf16x4 a0[2][4], a1[2][4];
    auto addr = [&](const int ch, int &bs0, int &bs1, int &zo0, int &zo1) {
      // this lane's two rows of the transposed reads: pixels P0 = 32 ch + 8 g + q and P0 + 4 (dZ rows beyond the image are zero)
      const int P0 = 32 * ch + 8 * g + q, P1 = P0 + 4;
      const int c0 = min(P0, ntot - 1), c1 = min(P1, ntot - 1);
      const int i0 = cm_div(c0, inv_np), i1 = cm_div(c1, inv_np), r0 = c0 - i0 * npix, r1 = c1 - i1 * npix;
      const int y0 = cm_div(r0, inv_wo), x0 = r0 - y0 * Wo, y1 = cm_div(r1, inv_wo), x1 = r1 - y1 * Wo;
      bs0 = i0 * tile_bytes + (y0 * Wp + x0) * G::PB; bs1 = i1 * tile_bytes + (y1 * Wp + x1) * G::PB;
      zo0 = P0 * 32 + 8 * p4; zo1 = P1 * 32 + 8 * p4;
    };
    auto load_z = [&](const int zo0, const int zo1) {
      const bf16x4 z0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(MILE_LDS_PTR(bf16x4, zt + zo0));
      const bf16x4 z1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(MILE_LDS_PTR(bf16x4, zt + zo1));
      return __builtin_shufflevector(z0, z1, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto load_a = [&](const int bs0, const int bs1, const int grp, bf16x4 (&f0)[4], bf16x4 (&f1)[4]) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int mt = 4 * grp + u;
        if (mt < NMT) {
          f0[u] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(MILE_LDS_PTR(bf16x4, tile + bs0 + so[mt]));
          f1[u] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(MILE_LDS_PTR(bf16x4, tile + bs1 + so[mt]));
        }
      }
    };
    int ch = wave;
    if (ch * 32 < ntot) {
      int bs0, bs1, zo0, zo1;
      addr(ch, bs0, bs1, zo0, zo1);
      bf16x8 bz = load_z(zo0, zo1);
      load_a(bs0, bs1, 0, a0[0], a1[0]);
      for (bool more = true; more;) {
        const int nch = ch + 4;
        more = nch * 32 < ntot;
        int nb0 = 0, nb1 = 0, nz0 = 0, nz1 = 0;
        if (more) addr(nch, nb0, nb1, nz0, nz1);
        bf16x8 bzn = bz;
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
          const int cur = grp & 1, nxt = cur ^ 1;
          if (grp + 1 < NG) load_a(bs0, bs1, grp + 1, a0[nxt], a1[nxt]);
          else if (more) { bzn = load_z(nz0, nz1); load_a(nb0, nb1, 0, a0[nxt], a1[nxt]); }
          __builtin_amdgcn_sched_barrier(0);
          if (grp == 0) accb = cm_mfma(ones, bz, accb);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int mt = 4 * grp + u;
            if (mt < NMT) acc[mt] = cm_mfma(__builtin_shufflevector(a0[cur][u], a1[cur][u], 0, 1, 2, 3, 4, 5, 6, 7), bz, acc[mt]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        bz = bzn; bs0 = nb0; bs1 = nb1; ch = nch;
      }
    }
  }
  // the four waves' partial sums -> one: red[wave][tile][lane] through LDS (aliases the image tiles), fixed order
  __syncthreads();
  cm_f32x4 *red = reinterpret_cast<cm_f32x4 *>(cm_lds);
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) red[(wave * (NMT + 1) + mt) * 64 + lane] = acc[mt];
  red[(wave * (NMT + 1) + NMT) * 64 + lane] = accb;
  __syncthreads();
  float *dst = part + ((size_t)e * gridDim.x + blockIdx.x) * (25 * CIN * COUT + COUT);
  for (int mt = wave; mt <= NMT; mt += 4) {
    cm_f32x4 t = red[(0 * (NMT + 1) + mt) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) t += red[(w * (NMT + 1) + mt) * 64 + lane];
    if (n16 >= COUT) continue;
    if (mt == NMT) {                         // ones row: every row of the tile is the column sum of dZ
      if (g == 0) dst[25 * CIN * COUT + n16] = t[0];
    } else {                                 // D[row 4 g + i][col n16]: slot 4 mt + g, channel c0 + i, output channel n16
      const int s = 4 * mt + g;
      if (G::valid(s)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (G::c0(s) + i < CIN) dst[(G::tap(s) * CIN + G::c0(s) + i) * COUT + n16] = t[i];
      }
    }
  }
}

// dZ of one image as PAIR rows for the kernel-gradient pass: zt[pair (y, x / 2)][column 2 co + dxo] = dz[y][x + dxo][co], bf16,
// npairs_alloc rows written (zeros outside the image).  One pair per thread and pass, U pairs' loads in flight.
template <int COUT, bool A16 = false>
__device__ __forceinline__ void cm_stage_dz_pairs(char *zt, const float *dp_img, const void *a_img, int Ho, int Wo, int npairs_alloc,
                                                  int activation, int tid) {
  static_assert(COUT % 2 == 0 && 2 * COUT <= 16, "channel pairs, 16 columns");
  constexpr int U = 2;
  const int W2 = (Wo + 1) / 2, Hq = Ho / 2, Wq = Wo / 2;
  const float inv_w2 = 1.0f / (float)W2;
  for (int i0 = tid; i0 < npairs_alloc; i0 += 256 * U) {
    float av[U][2][COUT], gv[U][2][COUT];
    bool in[U][2];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pi = i0 + 256 * u;
      const int y = cm_div(pi, inv_w2), x = 2 * (pi - y * W2);
#pragma unroll
      for (int dxo = 0; dxo < 2; ++dxo) {
        in[u][dxo] = pi < npairs_alloc && y < Ho && x + dxo < Wo;
        const bool pin = in[u][dxo] && y < 2 * Hq && x + dxo < 2 * Wq;
        cm_load_act<COUT, A16>(a_img, (size_t)(in[u][dxo] ? y * Wo + x + dxo : 0), av[u][dxo]);
        const float *gp = dp_img + (size_t)(pin ? (y >> 1) * Wq + ((x + dxo) >> 1) : 0) * COUT;
#pragma unroll
        for (int c = 0; c < COUT; c += 2) {
          const cm_f32x2 tg = pin ? *reinterpret_cast<const cm_f32x2 *>(gp + c) : cm_f32x2{0.0f, 0.0f};
          gv[u][dxo][c] = tg[0]; gv[u][dxo][c + 1] = tg[1];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pi = i0 + 256 * u;
      if (pi >= npairs_alloc) continue;
      uint32_t pk[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {          // packed word c = columns 2c, 2c + 1 = channel c at x and x + 1
        float v0 = 0.0f, v1 = 0.0f;
        if (c < COUT) {
          v0 = in[u][0] ? 0.25f * gv[u][0][c] * act_bwd(activation, av[u][0][c]) : 0.0f;
          v1 = in[u][1] ? 0.25f * gv[u][1][c] * act_bwd(activation, av[u][1][c]) : 0.0f;
        }
        pk[c] = cm_pack2(v0, v1);
      }
      *reinterpret_cast<cm_u32x4 *>(zt + (size_t)pi * 32) = cm_u32x4{pk[0], pk[1], pk[2], pk[3]};
      *reinterpret_cast<cm_u32x4 *>(zt + (size_t)pi * 32 + 16) = cm_u32x4{pk[4], pk[5], pk[6], pk[7]};
    }
  }
}

// The kernel gradient in the PAIR form (<= 4 input channels, 2 COUT <= 16): the MFMA K index runs over pixel PAIRS, the B columns
// are (output channel, which pixel of the pair):  D[(kh, kw', ci)][(co, dxo)] = sum_pairs in[y + kh][x + kw'][ci] dz[y][x + dxo][co]
// is a contribution to dK[kh][kw' - dxo][ci][co], so dK[kh][kw] = D[kw][dxo = 0] + D[kw + 1][dxo = 1] -- half the MFMAs and
// transposed reads of the pixel form, 12 of 16 columns useful instead of 6.  The two halves meet in the final LDS reduction.
template <int COUT>
__global__ __launch_bounds__(256) void k_conv5m_dw2x(const float *in, long long sE, long long sB, long long sH, long long sW, long long sC,
                                                     int CIN, int H, int W, int pad, const float *dp, const float *a, int activation, float *part,
                                                     int R, int ipw, int a16 = 0) {
  using G = CSlot<CM_IN4>;
  constexpr int NMT = (G::NS + 3) / 4;            // 8 accumulator tiles of 4 slots x 4 channels
  static_assert(2 * COUT <= 16 && COUT % 2 == 0, "columns = (channel, pixel of the pair)");
  extern __shared__ __attribute__((aligned(16))) char cm_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, e = blockIdx.y;
  const int g = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
  const int Ho = H + 2 * pad - 4, Wo = W + 2 * pad - 4, Hp = H + 2 * pad, Wp = W + 2 * pad, npix = Ho * Wo;
  const int W2 = (Wo + 1) / 2, npairs = Ho * W2, npairs32 = (npairs + 31) / 32 * 32;
  char *tile = cm_lds;                                                  // (Hp * Wp + 8) pixels of 8 bytes
  char *zt = cm_lds + ((size_t)(Hp * Wp + 8) * G::PB + 15) / 16 * 16;   // [npairs32][16] bf16
  int so[NMT];
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) so[mt] = 4 * mt + p4 < G::NS ? G::off(4 * mt + p4, Wp) : 0;
  cm_f32x4 acc[NMT], accb = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) acc[mt] = cm_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const float inv_w2 = 1.0f / (float)W2;
  const cm_u32x4 ones_u = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_u);
  const int b0 = blockIdx.x * ipw, b1 = min(R, b0 + ipw);
  for (int b = b0; b < b1; ++b) {
    const size_t img = (size_t)e * R + b;
    __syncthreads();
    cm_stage_input<CM_IN4>(tile, in + (size_t)e * sE + (size_t)b * sB, sH, sW, sC, CIN, H, W, pad, tid);
    if (a16) cm_stage_dz_pairs<COUT, true>(zt, dp + img * (Ho / 2) * (Wo / 2) * COUT, (const uint16_t *)a + img * npix * COUT, Ho, Wo, npairs32, activation, tid);
    else cm_stage_dz_pairs<COUT>(zt, dp + img * (Ho / 2) * (Wo / 2) * COUT, a + img * npix * COUT, Ho, Wo, npairs32, activation, tid);
    __syncthreads();
    constexpr int NG = (NMT + 3) / 4;
    static_assert(NG % 2 == 0, "the register double buffer returns to buffer 0 at every chunk");
    bf16x4 a0[2][4], a1[2][4];
    auto addr = [&](const int ch, int &bs0, int &bs1, int &zo0, int &zo1) {
      const int P0 = 32 * ch + 8 * g + q, P1 = P0 + 4;            // this lane's two pair rows of the transposed reads
      const int c0 = min(P0, npairs - 1), c1 = min(P1, npairs - 1);
      const int y0 = cm_div(c0, inv_w2), x0 = 2 * (c0 - y0 * W2), y1 = cm_div(c1, inv_w2), x1 = 2 * (c1 - y1 * W2);
      bs0 = (y0 * Wp + x0) * G::PB; bs1 = (y1 * Wp + x1) * G::PB;
      zo0 = P0 * 32 + 8 * p4; zo1 = P1 * 32 + 8 * p4;
    };
    auto load_z = [&](const int zo0, const int zo1) {
      const bf16x4 z0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(MILE_LDS_PTR(bf16x4, zt + zo0));
      const bf16x4 z1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(MILE_LDS_PTR(bf16x4, zt + zo1));
      return __builtin_shufflevector(z0, z1, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto load_a = [&](const int bs0, const int bs1, const int grp, bf16x4 (&f0)[4], bf16x4 (&f1)[4]) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int mt = 4 * grp + u;
        if (mt < NMT) {
          f0[u] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(MILE_LDS_PTR(bf16x4, tile + bs0 + so[mt]));
          f1[u] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(MILE_LDS_PTR(bf16x4, tile + bs1 + so[mt]));
        }
      }
    };
    int ch = wave;
    if (ch * 32 < npairs) {
      int bs0, bs1, zo0, zo1;
      addr(ch, bs0, bs1, zo0, zo1);
      bf16x8 bz = load_z(zo0, zo1);
      load_a(bs0, bs1, 0, a0[0], a1[0]);
      for (bool more = true; more;) {
        const int nch = ch + 4;
        more = nch * 32 < npairs;
        int nb0 = 0, nb1 = 0, nz0 = 0, nz1 = 0;
        if (more) addr(nch, nb0, nb1, nz0, nz1);
        bf16x8 bzn = bz;
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
          const int cur = grp & 1, nxt = cur ^ 1;
          if (grp + 1 < NG) load_a(bs0, bs1, grp + 1, a0[nxt], a1[nxt]);
          else if (more) { bzn = load_z(nz0, nz1); load_a(nb0, nb1, 0, a0[nxt], a1[nxt]); }
          __builtin_amdgcn_sched_barrier(0);
          if (grp == 0) accb = cm_mfma(ones, bz, accb);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int mt = 4 * grp + u;
            if (mt < NMT) acc[mt] = cm_mfma(__builtin_shufflevector(a0[cur][u], a1[cur][u], 0, 1, 2, 3, 4, 5, 6, 7), bz, acc[mt]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        bz = bzn; bs0 = nb0; bs1 = nb1; ch = nch;
      }
    }
  }
  // four waves' partial sums -> red[0] (fixed order), then dK[kh][kw][ci][co] = D[slot (kh, kw)][ci][2 co] + D[slot (kh, kw + 1)][ci][2 co + 1]
  __syncthreads();
  cm_f32x4 *red = reinterpret_cast<cm_f32x4 *>(cm_lds);
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) red[(wave * (NMT + 1) + mt) * 64 + lane] = acc[mt];
  red[(wave * (NMT + 1) + NMT) * 64 + lane] = accb;
  __syncthreads();
  for (int mt = wave; mt <= NMT; mt += 4) {
    cm_f32x4 t = red[(0 * (NMT + 1) + mt) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) t += red[(w * (NMT + 1) + mt) * 64 + lane];
    red[mt * 64 + lane] = t;                     // wave 0's slot of tile mt: only this wave touches tile mt here
  }
  __syncthreads();
  const float *D = reinterpret_cast<const float *>(cm_lds);   // D[(mt * 64 + g * 16 + n16) * 4 + i]: slot 4 mt + g, channel i, column n16
  float *dst = part + ((size_t)e * gridDim.x + blockIdx.x) * (25 * CIN * COUT + COUT);
  for (int i = tid; i < 25 * CIN * COUT + COUT; i += 256) {
    float v;
    if (i < 25 * CIN * COUT) {
      const int co = i % COUT, ci = (i / COUT) % CIN, tap = i / (COUT * CIN);
      const int kh = tap / 5, kw = tap % 5, s0 = 6 * kh + kw, s1 = s0 + 1;
      v = D[(((s0 >> 2) * 64 + (s0 & 3) * 16 + 2 * co) * 4) + ci] + D[(((s1 >> 2) * 64 + (s1 & 3) * 16 + 2 * co + 1) * 4) + ci];
    } else {
      const int co = i - 25 * CIN * COUT;         // ones row: every row of the bias tile is the column sum; row 0 = lanes g = 0, register 0
      v = D[((NMT * 64 + 2 * co) * 4)] + D[((NMT * 64 + 2 * co + 1) * 4)];
    }
    dst[i] = v;
  }
}

// LDS bytes of the three kernels for a geometry (host side)
static inline size_t cm_lds_fwd(int mode, int H, int W, int pad, int CIN, int COUT, int ni = 1) {
  const int pb = mode == CM_IN4 ? 8 : 16;
  return std::max((size_t)ni * ((H + 2 * pad) * (W + 2 * pad) + 8) * pb, (size_t)25 * CIN * COUT * 4);
}
static inline size_t cm_lds_dx(int Ho, int Wo, int CIN, int COUT) { return std::max((size_t)(Ho + 8) * (Wo + 8) * 32, (size_t)25 * CIN * COUT * 4); }
static inline size_t cm_lds_dx2x(int Ho, int Wo, int CIN, int COUT, int ni = 1) {
  return std::max((size_t)ni * ((Ho + 8) * (Wo + 8) + 8) * 32, (size_t)25 * CIN * COUT * 4);
}
static inline size_t cm_lds_dw2x(int H, int W, int pad) {
  const int Ho = H + 2 * pad - 4, Wo = W + 2 * pad - 4, npairs = Ho * ((Wo + 1) / 2);
  const size_t tiles = ((size_t)((H + 2 * pad) * (W + 2 * pad) + 8) * 8 + 15) / 16 * 16 + (size_t)((npairs + 31) / 32 * 32) * 32;
  const size_t red = (size_t)4 * (8 + 1) * 64 * 16;
  return tiles > red ? tiles : red;
}
static inline size_t cm_lds_dw(int mode, int H, int W, int pad, int ni = 1) {
  const int pb = mode == CM_IN4 ? 8 : 16, ns = mode == CM_IN4 ? 30 : 50;
  const int Ho = H + 2 * pad - 4, Wo = W + 2 * pad - 4;
  const size_t tiles = (size_t)ni * (((size_t)((H + 2 * pad) * (W + 2 * pad) + 8) * pb + 15) / 16 * 16) + (size_t)((ni * Ho * Wo + 31) / 32 * 32) * 32;
  const size_t red = (size_t)4 * ((ns + 3) / 4 + 1) * 64 * 16;
  return tiles > red ? tiles : red;
}
