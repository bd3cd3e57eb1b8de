// LeNet target (BASELINE config 5, src/models/images/cnns.py:33-66), fp32 (MILE_GRAD_LENET_F32, what AUTO takes): the
// convolutions as hand-written direct kernels (k_conv5_fwd / k_conv5_dx / k_conv5_dw below: LDS image tiles, no im2col
// matrices), the Dense layers as strided-batched SGEMMs (mile_grad_gemm.h), pooling as an elementwise kernel.  The first
// implementation -- im2col + SGEMMs + col2im, the matrices in HBM -- is kept as second implementation / fallback for images too
// large for the tiles (MILE_LENET_GEMM=1).  The bf16 MFMA form of the convolutions is mile_lenet_mfma.h.
//   x NCHW -> Conv(6, 5x5, pad 2) -> act -> avg_pool 2 -> Conv(16, 5x5) -> act -> avg_pool 2
//   -> flatten (h, w, c) -> Dense(120) -> act -> Dense(84) -> act -> Dense(out)
// Activations are NHWC with particles folded into the batch: [E*R][H][W][C].
#pragma once
#include "mile_device.h"

struct LeNetGeom {
  int C, H, W, K;                 // image channels / size, output width
  int hp1, wp1, h2, w2, hp2, wp2, flat;
  // parameter offsets in the raveled vector (ravel_pytree order)
  int b_c1, k_c1, b_c2, k_c2, b_f1, k_f1, b_f2, k_f2, b_f3, k_f3, d;
};

// dst[(b*Ho + y)*Wo + x][(kh*5 + kw)*C + c] = src[b, y + kh - pad, x + kw - pad, c] (0 outside).
// src element (b, h, w, c) lives at b*sB + h*sH + w*sW + c*sC: NCHW input images and NHWC activations alike.
__global__ __launch_bounds__(256) void k_im2col5(const float *src, float *dst, long long B, int H, int W, int C, int Ho, int Wo, int pad,
                                                 long long sB, long long sH, long long sW, long long sC) {
  const long long total = B * Ho * Wo * 25 * C;
  const int KC = 25 * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int kc = (int)(i % KC);
    const long long row = i / KC;
    const int c = kc % C, kw = (kc / C) % 5, kh = kc / (5 * C);
    const int x = (int)(row % Wo), y = (int)((row / Wo) % Ho);
    const long long b = row / ((long long)Wo * Ho);
    const int h = y + kh - pad, w = x + kw - pad;
    dst[i] = (h >= 0 && h < H && w >= 0 && w < W) ? src[b * sB + h * sH + w * sW + c * sC] : 0.0f;
  }
}

// avg_pool 2x2 stride 2 VALID: src [B][H][W][C] -> dst [B][H/2][W/2][C]
__global__ __launch_bounds__(256) void k_avgpool2(const float *src, float *dst, long long B, int H, int W, int C) {
  const int Hp = H / 2, Wp = W / 2;
  const long long total = B * Hp * Wp * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int x = (int)((i / C) % Wp), y = (int)((i / ((long long)C * Wp)) % Hp);
    const long long b = i / ((long long)C * Wp * Hp);
    const float *s = src + ((b * H + 2 * y) * W + 2 * x) * C + c;
    dst[i] = 0.25f * ((s[0] + s[C]) + (s[(long long)W * C] + s[(long long)W * C + C]));
  }
}

// dz[b,h,w,c] = (pool gradient spread back: dp[b,h/2,w/2,c] / 4, 0 on the cropped border) * act'(a[b,h,w,c])
__global__ __launch_bounds__(256) void k_unpool_actgrad(const float *dp, const float *a, float *dz, long long B, int H, int W, int C,
                                                        int activation) {
  const int Hp = H / 2, Wp = W / 2;
  const long long total = B * H * W * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int w = (int)((i / C) % W), h = (int)((i / ((long long)C * W)) % H);
    const long long b = i / ((long long)C * W * H);
    float g = 0.0f;
    if (h < 2 * Hp && w < 2 * Wp) g = 0.25f * dp[((b * Hp + h / 2) * Wp + w / 2) * C + c];
    dz[i] = g * act_bwd(activation, a[i]);
  }
}

// col2im for the VALID 5x5 convolution, as a gather: dst[b,y,x,c] = sum over the patches that cover (y, x)
// of dcol[(b, y-kh, x-kw)][(kh, kw, c)];  dcol [B*Ho*Wo][25*C], dst [B][Ho+4][Wo+4][C]
__global__ __launch_bounds__(256) void k_col2im5(const float *dcol, float *dst, long long B, int Ho, int Wo, int C) {
  const int H = Ho + 4, W = Wo + 4, KC = 25 * C;
  const long long total = B * H * W * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int x = (int)((i / C) % W), y = (int)((i / ((long long)C * W)) % H);
    const long long b = i / ((long long)C * W * H);
    float s = 0.0f;
    for (int kh = 0; kh < 5; ++kh) {
      const int yy = y - kh;
      if (yy < 0 || yy >= Ho) continue;
      for (int kw = 0; kw < 5; ++kw) {
        const int xx = x - kw;
        if (xx < 0 || xx >= Wo) continue;
        s += dcol[((b * Ho + yy) * Wo + xx) * KC + (kh * 5 + kw) * C + c];
      }
    }
    dst[i] = s;
  }
}

// ------------------------------------------------------------------------------------------------------
// Direct 5x5 convolutions (second implementation of the conv half: no im2col matrices in HBM, no skinny
// GEMMs).  One workgroup = one particle x a range of images; the particle's kernel sits in LDS, each image's
// input (zero-padded) and, for the gradients, its dZ are staged in LDS tiles.  NHWC activations
// [E][R][H][W][C]; the first layer's input is the shared NCHW image chunk (general strides, particle stride 0).
// ------------------------------------------------------------------------------------------------------
// dZ of a conv layer's output, formed on the fly from the pooled gradient and the activation output:
// dz[y][x][c] = (avg-pool backward: dp[y/2][x/2][c] / 4, zero on the cropped border) * act'(a[y][x][c])
__device__ __forceinline__ float dz_from_pool(const float *dp_img, const float *a_img, int y, int x, int c, int Ho, int Wo, int C,
                                              int activation) {
  const int Hq = Ho / 2, Wq = Wo / 2;
  float g = 0.0f;
  if (y < 2 * Hq && x < 2 * Wq) g = 0.25f * dp_img[((y >> 1) * Wq + (x >> 1)) * C + c];
  return g * act_bwd(activation, a_img[(y * Wo + x) * C + c]);
}

// ---- staging helpers of the direct kernels ---------------------------------------------------------------------------
// U elements / pixels per thread and pass with ALL their global loads issued before the first LDS store: as plain loops
// (load, store, next) every iteration waited out its own loads' latency -- a dozen serial HBM round trips per image, which
// was most of these kernels' time (found on the MFMA forms, mile_lenet_mfma.h; round 2).
typedef float cv_f32x2 __attribute__((ext_vector_type(2)));
typedef float cv_f32x4 __attribute__((ext_vector_type(4)));

// zero-padded input tile [CIN][Hp][Wp] of one image
__device__ __forceinline__ void conv_stage_input(float *tile, const float *src, long long sH, long long sW, long long sC, int CIN, int H, int W,
                                                 int pad, int Hp, int Wp, int tid, int nt) {
  constexpr int U = 4;
  const int n = CIN * Hp * Wp;
  for (int i0 = tid; i0 < n; i0 += nt * U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + nt * u;
      const int xx = i % Wp, yy = (i / Wp) % Hp, ci = i / (Wp * Hp);
      const int h = yy - pad, w = xx - pad;
      v[u] = (i < n && h >= 0 && h < H && w >= 0 && w < W) ? src[h * sH + w * sW + ci * sC] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i0 + nt * u < n) tile[i0 + nt * u] = v[u];
  }
}

// dZ tile(s): dst[(img * Ht + y + halo) * Wt + x + halo][COUT] for ni images, Ht = Ho + 2 halo, Wt = Wo + 2 halo, zero halo;
// dz = unpool(dp) * act'(a) as dz_from_pool.  One pixel per thread and pass, its channels as 8- / 16-byte loads (the a / dp
// arrays are 16-byte aligned and COUT is even).
template <int COUT>
__device__ __forceinline__ void conv_stage_dz(float *dst, const float *dp0, const float *a0, int ni, int Ho, int Wo, int halo, int activation,
                                              int tid, int nt) {
  static_assert(COUT % 2 == 0, "channel pairs");
  constexpr int U = COUT > 8 ? 2 : 4, VW = COUT % 4 == 0 ? 4 : 2;
  const int Wt = Wo + 2 * halo, Ht = Ho + 2 * halo, Hq = Ho / 2, Wq = Wo / 2, per = Ht * Wt, n = ni * per;
  for (int i0 = tid; i0 < n; i0 += nt * U) {
    float av[U][COUT], gv[U][COUT];
    bool in[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + nt * u;
      const int im = i / per, r = i - im * per, yy = r / Wt, xx = r - yy * Wt, y = yy - halo, x = xx - halo;
      in[u] = i < n && y >= 0 && y < Ho && x >= 0 && x < Wo;
      const bool pin = in[u] && y < 2 * Hq && x < 2 * Wq;
      const float *ap = a0 + ((size_t)(in[u] ? im : 0) * Ho * Wo + (in[u] ? y * Wo + x : 0)) * COUT;
      const float *gp = dp0 + ((size_t)(pin ? im : 0) * Hq * Wq + (pin ? (y >> 1) * Wq + (x >> 1) : 0)) * COUT;
#pragma unroll
      for (int c = 0; c < COUT; c += VW) {
        if constexpr (VW == 4) {
          const cv_f32x4 ta = *(const cv_f32x4 *)(ap + c);
          const cv_f32x4 tg = pin ? *(const cv_f32x4 *)(gp + c) : cv_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int k = 0; k < 4; ++k) { av[u][c + k] = ta[k]; gv[u][c + k] = tg[k]; }
        } else {
          const cv_f32x2 ta = *(const cv_f32x2 *)(ap + c);
          const cv_f32x2 tg = pin ? *(const cv_f32x2 *)(gp + c) : cv_f32x2{0.0f, 0.0f};
          av[u][c] = ta[0]; av[u][c + 1] = ta[1]; gv[u][c] = tg[0]; gv[u][c + 1] = tg[1];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + nt * u;
      if (i >= n) continue;
      float *o = dst + (size_t)i * COUT;
#pragma unroll
      for (int c = 0; c < COUT; ++c) o[c] = in[u] ? 0.25f * gv[u][c] * act_bwd(activation, av[u][c]) : 0.0f;
    }
  }
}

// Compile-time geometries of the two common inputs (GEO = 0: everything at run time).  With the extents known the
// tap offsets inside a tile become immediates of the LDS instructions instead of per-tap VALU address arithmetic
// (the counters showed 4-6x more VALU instructions than FMAs in the run-time form).
template <int GEO> struct ConvGeo { static constexpr int CIN = 0, H = 0, W = 0, PAD = 0, NCHW = 0; };
template <> struct ConvGeo<1> { static constexpr int CIN = 3, H = 32, W = 32, PAD = 2, NCHW = 1; };   // CIFAR conv1
template <> struct ConvGeo<2> { static constexpr int CIN = 6, H = 16, W = 16, PAD = 0, NCHW = 0; };   // CIFAR conv2
template <> struct ConvGeo<3> { static constexpr int CIN = 1, H = 28, W = 28, PAD = 2, NCHW = 1; };   // MNIST conv1
template <> struct ConvGeo<4> { static constexpr int CIN = 6, H = 14, W = 14, PAD = 0, NCHW = 0; };   // MNIST conv2
#define CONV_FOLD_GEOMETRY()                                                                          \
  if (GEO) {                                                                                          \
    CIN = ConvGeo<GEO>::CIN; H = ConvGeo<GEO>::H; W = ConvGeo<GEO>::W; pad = ConvGeo<GEO>::PAD;       \
    if (ConvGeo<GEO>::NCHW) { sH = W; sW = 1; sC = H * W; sB = CIN * H * W; }                         \
    else { sH = W * CIN; sW = CIN; sC = 1; sB = H * W * CIN; }                                        \
  }

template <int COUT>
struct ConvPad { static constexpr int P = (COUT + 3) / 4 * 4; };

// out[e][b][y][x][co] = act(bias[co] + sum_{kh,kw,ci} in[b][y+kh-pad][x+kw-pad][ci] K[kh][kw][ci][co])
template <int COUT, int GEO>
__global__ __launch_bounds__(256) void k_conv5_fwd(const float *in, long long sE, long long sB, long long sH, long long sW, long long sC,
                                                   int CIN, int H, int W, int pad, const float *theta, int k_off, int b_off, int d, float *out,
                                                   int R, int ipw, int activation) {
  CONV_FOLD_GEOMETRY()
  constexpr int CP = ConvPad<COUT>::P;
  extern __shared__ __attribute__((aligned(16))) float cl[];
  const int tid = threadIdx.x, e = blockIdx.y;
  const int Ho = H + 2 * pad - 4, Wo = W + 2 * pad - 4, Hp = H + 2 * pad, Wp = W + 2 * pad;
  float *Ksh = cl;                       // [25*CIN][CP]
  float *bsh = Ksh + 25 * CIN * CP;      // [CP]
  float *tile = bsh + CP;                // [CIN][Hp][Wp]
  const float *K = theta + (size_t)e * d + k_off, *bias = theta + (size_t)e * d + b_off;
  for (int i = tid; i < 25 * CIN * CP; i += 256) Ksh[i] = (i % CP) < COUT ? K[(i / CP) * COUT + (i % CP)] : 0.0f;
  if (tid < CP) bsh[tid] = tid < COUT ? bias[tid] : 0.0f;
  const int b0 = blockIdx.x * ipw, b1 = min(R, b0 + ipw);
  for (int b = b0; b < b1; ++b) {
    __syncthreads();
    conv_stage_input(tile, in + (size_t)e * sE + (size_t)b * sB, sH, sW, sC, CIN, H, W, pad, Hp, Wp, tid, 256);
    __syncthreads();
    float *dst = out + ((size_t)e * R + b) * Ho * Wo * COUT;
    for (int p = tid; p < Ho * Wo; p += 256) {
      const int y = p / Wo, x = p % Wo;
      float acc[CP];
#pragma unroll
      for (int c = 0; c < CP; ++c) acc[c] = bsh[c];
      for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int kh = 0; kh < 5; ++kh)
#pragma unroll
          for (int kw = 0; kw < 5; ++kw) {
            const float v = tile[(ci * Hp + y + kh) * Wp + x + kw];
            const float *kr = Ksh + ((kh * 5 + kw) * CIN + ci) * CP;
#pragma unroll
            for (int c = 0; c < CP; ++c) acc[c] = fmaf(v, kr[c], acc[c]);
          }
#pragma unroll
      for (int c = 0; c < COUT; ++c) dst[(size_t)p * COUT + c] = act_fwd(activation, acc[c]);
    }
  }
}

// VALID 5x5 conv, gradient w.r.t. the input: din[e][b][yi][xi][ci] = sum_{kh,kw,co} dz[e][b][yi-kh][xi-kw][co] K[kh][kw][ci][co]
// The dZ tiles of NI images share LDS; per kernel tap the CIN x COUT weights are read into registers once and used for
// every pixel the thread owns (the first version re-read them per pixel and was LDS-issue bound).
template <int CIN, int COUT, int NI, int HO_T, int WO_T>
__global__ __launch_bounds__(256) void k_conv5_dx(const float *dp, const float *a, int activation, const float *theta, int k_off, int d,
                                                  float *din, int R, int Ho, int Wo, int ipw) {
  if (HO_T) { Ho = HO_T; Wo = WO_T; }
  extern __shared__ __attribute__((aligned(16))) float cl[];
  const int tid = threadIdx.x, e = blockIdx.y;
  const int H = Ho + 4, W = Wo + 4, Ht = Ho + 8, Wt = Wo + 8;
  float *Ksh = cl;                         // [25*CIN][COUT]
  float *tile = Ksh + 25 * CIN * COUT;     // [NI][Ht][Wt][COUT], zero halo of 4
  const float *K = theta + (size_t)e * d + k_off;
  for (int i = tid; i < 25 * CIN * COUT; i += 256) Ksh[i] = K[i];
  const int npix = H * W;
  constexpr int PPT = NI;                  // pixels per thread and pass (NI images x ~200 pixels over 256 threads)
  const int b0 = blockIdx.x * ipw, b1 = min(R, b0 + ipw);
  for (int bb = b0; bb < b1; bb += NI) {
    const int ni = min(NI, b1 - bb);
    __syncthreads();
    {
      const size_t img0 = (size_t)e * R + bb;
      conv_stage_dz<COUT>(tile, dp + img0 * (Ho / 2) * (Wo / 2) * COUT, a + img0 * Ho * Wo * COUT, ni, Ho, Wo, 4, activation, tid, 256);
    }
    __syncthreads();
    const int total = ni * npix;
    for (int p0 = tid; p0 < total; p0 += 256 * PPT) {
      float acc[PPT][CIN];
      int toff[PPT];
      bool ok[PPT];
#pragma unroll
      for (int u = 0; u < PPT; ++u) {
        const int p = p0 + u * 256;
        ok[u] = p < total;
        const int pc = ok[u] ? p : 0, im = pc / npix, q = pc % npix;
        toff[u] = ((im * Ht + q / W + 4) * Wt + q % W + 4) * COUT;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) acc[u][ci] = 0.0f;
      }
      for (int kh = 0; kh < 5; ++kh)
        for (int kw = 0; kw < 5; ++kw) {
          float kr[CIN][COUT];
          const float *kp = Ksh + (kh * 5 + kw) * CIN * COUT;
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int c = 0; c < COUT; ++c) kr[ci][c] = kp[ci * COUT + c];
#pragma unroll
          for (int u = 0; u < PPT; ++u) {
            const float *zv = tile + toff[u] - (kh * Wt + kw) * COUT;
#pragma unroll
            for (int c = 0; c < COUT; ++c) {
              const float z = zv[c];
#pragma unroll
              for (int ci = 0; ci < CIN; ++ci) acc[u][ci] = fmaf(z, kr[ci][c], acc[u][ci]);
            }
          }
        }
#pragma unroll
      for (int u = 0; u < PPT; ++u)
        if (ok[u]) {
          const int p = p0 + u * 256, im = p / npix, q = p % npix;
          float *dst = din + (((size_t)e * R + bb + im) * npix + q) * CIN;
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci) dst[ci] = acc[u][ci];
        }
    }
  }
}

// Kernel / bias gradient of one image range: part[(e * nwg + wg) * (25*CIN*COUT + COUT) + ...]
//   dK[kh][kw][ci][co] = sum_{b,y,x} in[b][y+kh-pad][x+kw-pad][ci] dz[b][y][x][co],  db[co] = sum dz
// Thread (k5, g): k5 = (kh, ci) owns the five kw taps of a kernel row -- per pixel 5 input reads and one dZ row
// feed 5*COUT FMAs; g = pixel group; the groups are reduced through LDS at the end.
template <int COUT, int GEO>
__global__ __launch_bounds__(256) void k_conv5_dw(const float *in, long long sE, long long sB, long long sH, long long sW, long long sC, int CIN,
                                                  int H, int W, int pad, const float *dp, const float *a, int activation, float *part, int R,
                                                  int ipw) {
  CONV_FOLD_GEOMETRY()
  extern __shared__ __attribute__((aligned(16))) float cl[];
  const int tid = threadIdx.x, nt = 256, e = blockIdx.y;
  const int Ho = H + 2 * pad - 4, Wo = W + 2 * pad - 4, Hp = H + 2 * pad, Wp = W + 2 * pad;
  const int K5 = 5 * CIN, G = nt / K5;                  // CIN <= 51
  float *tile = cl;                                      // [CIN][Hp][Wp]
  float *zt = tile + CIN * Hp * Wp;                      // [Ho*Wo][COUT]
  const bool active = tid < K5 * G;
  const int k5 = active ? tid % K5 : 0, g = active ? tid / K5 : 0;
  const int ci = k5 % CIN, kh = k5 / CIN;
  float acc[5][COUT], bsum[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) {
    bsum[c] = 0.0f;
#pragma unroll
    for (int kw = 0; kw < 5; ++kw) acc[kw][c] = 0.0f;
  }
  const int b0 = blockIdx.x * ipw, b1 = min(R, b0 + ipw);
  for (int b = b0; b < b1; ++b) {
    __syncthreads();
    conv_stage_input(tile, in + (size_t)e * sE + (size_t)b * sB, sH, sW, sC, CIN, H, W, pad, Hp, Wp, tid, nt);
    const size_t img = (size_t)e * R + b;
    conv_stage_dz<COUT>(zt, dp + img * (Ho / 2) * (Wo / 2) * COUT, a + img * Ho * Wo * COUT, 1, Ho, Wo, 0, activation, tid, nt);
    __syncthreads();
    if (active)
      for (int p = g; p < Ho * Wo; p += G) {
        const int y = p / Wo, x = p % Wo;
        const float *tv = tile + (ci * Hp + y + kh) * Wp + x;
        const float *zv = zt + p * COUT;
        float v[5];
#pragma unroll
        for (int kw = 0; kw < 5; ++kw) v[kw] = tv[kw];
#pragma unroll
        for (int c = 0; c < COUT; ++c) {
          const float z = zv[c];
          if (k5 == 0) bsum[c] += z;
#pragma unroll
          for (int kw = 0; kw < 5; ++kw) acc[kw][c] = fmaf(v[kw], z, acc[kw][c]);
        }
      }
  }
  // reduce the G pixel groups into group 0, RG groups per round through a small LDS buffer (aliases the tiles):
  // red[RG][k5][5][COUT] + red_b[RG][COUT].  A buffer for all groups at once would cost the kernel its occupancy.
  constexpr int RG = 3;
  const int per_g = K5 * 5 * COUT;
  float *red = cl, *red_b = cl + RG * per_g;
  for (int base = 1; base < G; base += RG) {
    __syncthreads();
    if (active && g >= base && g < base + RG) {
#pragma unroll
      for (int kw = 0; kw < 5; ++kw)
#pragma unroll
        for (int c = 0; c < COUT; ++c) red[(g - base) * per_g + (k5 * 5 + kw) * COUT + c] = acc[kw][c];
      if (k5 == 0)
#pragma unroll
        for (int c = 0; c < COUT; ++c) red_b[(g - base) * COUT + c] = bsum[c];
    }
    __syncthreads();
    if (active && g == 0) {
      const int ng = min(RG, G - base);
      for (int gg = 0; gg < ng; ++gg) {
#pragma unroll
        for (int kw = 0; kw < 5; ++kw)
#pragma unroll
          for (int c = 0; c < COUT; ++c) acc[kw][c] += red[gg * per_g + (k5 * 5 + kw) * COUT + c];
        if (k5 == 0)
#pragma unroll
          for (int c = 0; c < COUT; ++c) bsum[c] += red_b[gg * COUT + c];
      }
    }
  }
  if (active && g == 0) {
    float *dst = part + ((size_t)e * gridDim.x + blockIdx.x) * (25 * CIN * COUT + COUT);
#pragma unroll
    for (int kw = 0; kw < 5; ++kw)
#pragma unroll
      for (int c = 0; c < COUT; ++c) dst[((kh * 5 + kw) * CIN + ci) * COUT + c] = acc[kw][c];
    if (k5 == 0)
#pragma unroll
      for (int c = 0; c < COUT; ++c) dst[25 * CIN * COUT + c] = bsum[c];
  }
}

// slab[e][k_off + i] (+)= sum_wg part[e][wg][i] for the kernel entries, slab[e][b_off + c] for the bias entries
__global__ __launch_bounds__(256) void k_conv_reduce(const float *part, int nwg, int nk, int ncout, float *slab, long long dp, int k_off, int b_off,
                                                     int accumulate) {
  const int e = blockIdx.y, per = nk + ncout;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < per; i += gridDim.x * 256) {
    float s = 0.0f;
    for (int wgi = 0; wgi < nwg; ++wgi) s += part[((size_t)e * nwg + wgi) * per + i];
    float *p = slab + (size_t)e * dp + (i < nk ? k_off + i : b_off + (i - nk));
    *p = accumulate ? *p + s : s;
  }
}
