// LeNet target (BASELINE config 5, src/models/images/cnns.py:33-66): convolutions as im2col + the
// strided-batched SGEMMs of the layer-wise path (mile_grad_gemm.h), pooling / col2im / im2col as the
// elementwise HIP kernels below.  fp32, correctness-first: the im2col matrices live in HBM.
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
