// libmile_hip.so -- host side of the C ABI declared in include/mile_hip.h.
// gfx950 (MI355X) only.  No torch types: plain pointers, sizes and a hipStream_t.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <dlfcn.h>

#include "mile_device.h"
#include "mile_grad_generic.h"
#include "mile_grad_narrow.h"
#include "mile_grad_w64.h"
#include "mile_grad_w128b.h"
#include "mile_grad_gemm.h"
#include "mile_mm3.h"
#include "mile_lenet.h"
#include "mile_lenet_mfma.h"
#include "mile_predict.h"
#include "mile_update.h"

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return fail(MILE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));           \
  } while (0)

static const double MCLACHLAN_B1 = 0.1931833275037836;

struct mile_sampler {
  mile_model_spec spec{};
  DevSpec ds{};
  int device = 0;
  int n_cu = 256;
  // data
  float *X = nullptr, *Xp = nullptr;
  void *Xb = nullptr, *Xt = nullptr;   // bf16 copies for k_grad_w128b
  void *y = nullptr;
  int N = 0, Npad = 0, Fp = 0, Npb = 0;
  int win_begin = 0, win_count = 0;     // mile_set_row_window: rows [begin, begin + count) only (count 0 = all)
  // workspace
  int E_cap = 0;
  size_t ES_cap = 0;   // slab rows (particles x row splits) the workspace holds: fewer particles may split the rows further
  float *slabs = nullptr, *llpart = nullptr, *dK = nullptr, *lold = nullptr;
  float *upart = nullptr;               // segmented update (d beyond the one-workgroup forms): partial sums per segment
  int32_t *arrive = nullptr;            // [E_cap] arrival tickets of the fused integrator epilogue (k_grad_w64 SPLIT)
  float *ev_X = nullptr, *ev_Xp = nullptr; void *ev_y = nullptr; int ev_cap = 0;   // evaluation (test) set staging
  float *alt_x = nullptr, *alt_u = nullptr, *alt_g = nullptr, *alt_logp = nullptr;   // ping-pong state of mile_tune
  int grad_kernel = MILE_GRAD_AUTO;
  LeNetGeom lg{};                       // MILE_MODEL_LENET geometry and parameter offsets
  // layer-wise GEMM path (MILE_GRAD_GEMM_F32): rocBLAS handle and activation workspace
  void *blas = nullptr;
  long long *dbg_buf = nullptr;         // dev instrumentation (MILE_DEBUG=16)
  float *tune_info = nullptr;           // [E_cap, 3] scratch MCLMCInfo of mile_tune's large-d path
  float *gemm_ws = nullptr, *gemm_ones = nullptr;
  int gemm_ones_n = 0;
  size_t gemm_ws_floats = 0;
  int gemm_R = 0, gemm_E = 0;
  // layer-wise MFMA path (MILE_GRAD_MFMA_WIDE_*): activation workspace (zero-initialised: padding columns stay 0) and
  // the pre-split weight term planes
  float *wide_ws = nullptr; size_t wide_ws_floats = 0; int wide_R = 0, wide_E = 0;
  void *wide_wt = nullptr; size_t wide_wt_bytes = 0;
  float *wide_hb = nullptr; size_t wide_hb_floats = 0;   // k_wide_headblock's per-workgroup partial sums
  // timing of grad launches
  bool timing = false;
  std::vector<hipEvent_t> ev;
  size_t ev_used = 0;
};

static bool w64_supported(const mile_model_spec &sp) {
  if (sp.model != MILE_MODEL_FCN) return false;
  if (sp.task != MILE_TASK_REGRESSION || sp.activation != MILE_ACT_RELU) return false;
  const int nh = sp.n_layers - 1;
  if (nh < 1 || nh > 3) return false;
  for (int l = 0; l < nh; ++l)
    if (sp.widths[l] != 64) return false;
  if (sp.widths[nh] != 2) return false;
  if (sp.in_features > 16) return false;
  return true;
}

// the split-bf16 variant replaces the hidden->hidden products, so it needs at least one
// k_grad_narrow: 1-3 hidden layers of width <= 64 and F <= 64 (<= 32: weights in registers; 33..64: weights in LDS), or 4-10
// hidden layers of width <= 16 and F <= 16 (the depth ablations of the reference: experiments/**: [16]*4 .. [16]*9 + [2],
// [8]*6 + [2]); <= 16 outputs (regression: exactly mu, log sigma), any activation
static int narrow_tiles_hidden(const mile_model_spec &sp) {
  int mw = 0;
  for (int l = 0; l + 1 < sp.n_layers; ++l) mw = std::max(mw, sp.widths[l]);
  return (mw + 15) / 16;
}
static bool narrow_supported(const mile_model_spec &sp) {
  if (sp.model != MILE_MODEL_FCN || !sp.use_bias) return false;
  const int nh = sp.n_layers - 1;
  if (nh < 1 || nh > 10 || sp.in_features > 64) return false;
  for (int l = 0; l < nh; ++l)
    if (sp.widths[l] < 1 || sp.widths[l] > (nh > 3 ? 16 : 64)) return false;
  if (nh > 3 && sp.in_features > 16) return false;
  const int K = sp.widths[nh];
  if (K < 1 || K > 16) return false;
  if (sp.task == MILE_TASK_REGRESSION && K != 2) return false;
  return true;
}
static bool w64x3_supported(const mile_model_spec &sp) { return w64_supported(sp) && sp.n_layers - 1 >= 2; }
static bool is_w64(int kernel) { return kernel == MILE_GRAD_MFMA_W64 || kernel == MILE_GRAD_MFMA_W64_BF16X3; }

// ---- rocBLAS, resolved at first use: libmile_hip.so has no link-time dependency on it -----------
typedef int (*rb_create_t)(void **);
typedef int (*rb_destroy_t)(void *);
typedef int (*rb_set_stream_t)(void *, hipStream_t);
typedef int (*rb_sgemm_sb_t)(void *, int, int, int, int, int, const float *, const float *, int, long long, const float *, int,
                             long long, const float *, float *, int, long long, int);
static struct {
  void *lib = nullptr;
  rb_create_t create = nullptr;
  rb_destroy_t destroy = nullptr;
  rb_set_stream_t set_stream = nullptr;
  rb_sgemm_sb_t sgemm_sb = nullptr;
  bool tried = false;
} g_rb;
enum { RB_OP_N = 111, RB_OP_T = 112 };   // rocblas_operation_none / _transpose

static bool rocblas_load() {
  if (g_rb.tried) return g_rb.sgemm_sb != nullptr;
  g_rb.tried = true;
  for (const char *name : {"librocblas.so.5", "librocblas.so", "/opt/rocm/lib/librocblas.so"}) {
    g_rb.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (g_rb.lib) break;
  }
  if (!g_rb.lib) return false;
  g_rb.create = (rb_create_t)dlsym(g_rb.lib, "rocblas_create_handle");
  g_rb.destroy = (rb_destroy_t)dlsym(g_rb.lib, "rocblas_destroy_handle");
  g_rb.set_stream = (rb_set_stream_t)dlsym(g_rb.lib, "rocblas_set_stream");
  g_rb.sgemm_sb = (rb_sgemm_sb_t)dlsym(g_rb.lib, "rocblas_sgemm_strided_batched");
  if (!g_rb.create || !g_rb.destroy || !g_rb.set_stream || !g_rb.sgemm_sb) { g_rb.sgemm_sb = nullptr; return false; }
  return true;
}

// library GEMMs pay off once the hidden layers are wide; below that the single-launch generic kernel wins
static bool gemm_preferred(const mile_model_spec &sp) {
  if (sp.model != MILE_MODEL_FCN) return false;
  int mw = 0;
  for (int l = 0; l + 1 < sp.n_layers; ++l) mw = std::max(mw, sp.widths[l]);
  return sp.n_layers >= 2 && mw >= 96;
}

static bool w128b_supported(const mile_model_spec &sp) {
  if (sp.model != MILE_MODEL_FCN) return false;
  if (sp.task != MILE_TASK_REGRESSION || sp.activation != MILE_ACT_RELU) return false;
  const int nh = sp.n_layers - 1;
  if (nh < 1 || nh > 3) return false;
  for (int l = 0; l < nh; ++l)
    if (sp.widths[l] != 128) return false;
  if (sp.widths[nh] != 2) return false;
  if (sp.in_features > 16) return false;
  return true;
}

static int resolved_kernel(const mile_sampler *s) {
  if (s->spec.model == MILE_MODEL_LENET) return s->grad_kernel == MILE_GRAD_LENET_BF16 ? MILE_GRAD_LENET_BF16 : MILE_GRAD_LENET_F32;
  if (s->grad_kernel == MILE_GRAD_AUTO) {
    if (w64x3_supported(s->spec)) return MILE_GRAD_MFMA_W64_BF16X3;   // fp32-faithful and never slower than MFMA_W64
    if (w64_supported(s->spec)) return MILE_GRAD_MFMA_W64;
    // wide nets: the hand-written layer-wise MFMA GEMMs (fp32-faithful); rocBLAS (GEMM_F32) is the cross-check, never AUTO
    if (gemm_preferred(s->spec)) return MILE_GRAD_MFMA_WIDE_BF16X3;
    // the reference's own nets (16 / 32 wide, any activation, either head): fused fp32-MFMA kernel
    if (narrow_supported(s->spec) && getenv("MILE_NO_NARROW") == nullptr) return MILE_GRAD_MFMA_NARROW_F32;
    return MILE_GRAD_GENERIC;
  }
  return s->grad_kernel;
}

// rows per LDS tile of the generic kernel: fit the activation record in ~56 KiB
static int generic_R(const DevSpec &ds) {
  const int per_row = ds.act_stride + 2 * ds.max_width;
  int R = (56 * 1024 / 4 - 16) / per_row;
  R = std::max(1, std::min(R, 64));
  return R;
}

// k_grad_narrow: 16-row tiles, workgroups of up to 4 waves.  About four waves per SIMD over the whole grid (16 x CUs waves) and
// NO MORE row splits than that needs: every split is one more slab the update kernels sum -- with S = 64 splits of a d = 402
// net the two update launches of a step took 27-31 us each against 6.8 us for the gradient itself (profiles/r03/12_*), the
// slab / llpart loops being 64 dependent global loads long.  E = 12, N = 1052 (66 tiles): 17 splits of 4 waves; E = 128: 8.
static int narrow_S(const mile_sampler *s, int E) {
  const int tiles = (s->N + 15) / 16;
  if (narrow_tiles_hidden(s->spec) >= 3)      // LDS-weight form: ~100 KB of images staged per workgroup -> one workgroup per CU
    return std::max(1, std::min({64, s->n_cu / std::max(E, 1), std::max(1, tiles / NRW_MAXW)}));
  const int waves_per_particle = std::max(1, std::min(tiles, (16 * s->n_cu + std::max(E, 1) - 1) / std::max(E, 1)));
  return std::max(1, std::min(64, (waves_per_particle + NRW_MAXW - 1) / NRW_MAXW));
}
static int narrow_waves(const mile_sampler *s, int S, int N) {
  const int tiles = (N + 15) / 16;
  return std::max(1, std::min(NRW_MAXW, (tiles + S - 1) / S));
}

static int choose_S(const mile_sampler *s, int E, int kernel) {
  if (kernel == MILE_GRAD_MFMA_NARROW_F32) return narrow_S(s, E);
  if (is_w64(kernel)) {
    const int NB = s->Npad / 32;
    int S = std::max(1, s->n_cu / std::max(E, 1));
    S = std::min(S, std::max(1, NB / 4));  // keep >= 4 row blocks (one per wave) per workgroup
    return S;
  }
  if (kernel == MILE_GRAD_GEMM_F32 || kernel == MILE_GRAD_LENET_F32 || kernel == MILE_GRAD_LENET_BF16 || kernel == MILE_GRAD_MFMA_WIDE_BF16X3 ||
      kernel == MILE_GRAD_MFMA_WIDE_BF16)
    return 1;
  if (kernel == MILE_GRAD_MFMA_W128_BF16) {
    const int NBS = s->Npb / 64;              // iterations of two 32-row tiles
    int S = std::max(1, s->n_cu / std::max(E, 1));
    return std::min(S, std::max(1, NBS / 4));  // amortise weight staging over >= 8 row tiles
  }
  int S = std::max(1, (2 * s->n_cu) / std::max(E, 1));
  S = std::min(S, std::max(1, s->N / 64));
  return std::min(S, 64);
}

extern "C" {

const char *mile_last_error(void) { return g_err.c_str(); }
int32_t mile_abi_version(void) { return MILE_ABI_VERSION; }

int32_t mile_create(const mile_model_spec *spec, int32_t device, mile_sampler **out) {
  if (!spec || !out) return fail(MILE_ERR_INVALID, "mile_create: null argument");
  if (spec->n_layers < 1 || spec->n_layers > MILE_MAX_LAYERS) return fail(MILE_ERR_INVALID, "n_layers out of range");
  if (spec->in_features < 1) return fail(MILE_ERR_INVALID, "in_features must be >= 1");
  if (!spec->use_bias) return fail(MILE_ERR_INVALID, "use_bias=false is not supported");
  if (spec->activation < 0 || spec->activation > MILE_ACT_SIGMOID) return fail(MILE_ERR_INVALID, "unknown activation");
  if (spec->task != MILE_TASK_REGRESSION && spec->task != MILE_TASK_CLASSIFICATION) return fail(MILE_ERR_INVALID, "unknown task");
  if (spec->prior != MILE_PRIOR_NORMAL && spec->prior != MILE_PRIOR_LAPLACE) return fail(MILE_ERR_INVALID, "unknown prior");
  if (!(spec->prior_scale > 0.0f)) return fail(MILE_ERR_INVALID, "prior_scale must be > 0");
  for (int l = 0; l < spec->n_layers; ++l)
    if (spec->widths[l] < 1) return fail(MILE_ERR_INVALID, "layer width must be >= 1");
  if (spec->task == MILE_TASK_REGRESSION && spec->widths[spec->n_layers - 1] != 2)
    return fail(MILE_ERR_INVALID, "regression needs an output layer of width 2 (mu, log sigma)");
  if (spec->model != MILE_MODEL_FCN && spec->model != MILE_MODEL_LENET) return fail(MILE_ERR_INVALID, "unknown model");
  if (spec->model == MILE_MODEL_LENET) {
    if (spec->n_layers != 1) return fail(MILE_ERR_INVALID, "LeNet: n_layers must be 1 (widths[0] = out_dim)");
    if (spec->img_c < 1 || spec->img_h < 1 || spec->img_w < 1 ||
        (long long)spec->img_c * spec->img_h * spec->img_w != spec->in_features)
      return fail(MILE_ERR_INVALID, "LeNet: in_features must equal img_c * img_h * img_w");
    if ((spec->img_h / 2 - 4) / 2 < 1 || (spec->img_w / 2 - 4) / 2 < 1) return fail(MILE_ERR_INVALID, "LeNet: image too small");
    if (!rocblas_load()) return fail(MILE_ERR_HIP, "LeNet needs librocblas.so, which could not be loaded");
  }

  auto *s = new mile_sampler();
  s->spec = *spec;
  s->device = device;
  DevSpec &ds = s->ds;
  ds.n_layers = spec->n_layers;
  ds.in_features = spec->in_features;
  ds.activation = spec->activation;
  ds.task = spec->task;
  ds.prior = spec->prior;
  ds.prior_loc = spec->prior_loc;
  ds.prior_scale = spec->prior_scale;
  if (spec->model == MILE_MODEL_LENET) {   // ravel_pytree order of {'core': {conv1, conv2, fc1, fc2, fc3}}: bias, kernel each
    LeNetGeom &g = s->lg;
    g.C = spec->img_c; g.H = spec->img_h; g.W = spec->img_w; g.K = spec->widths[0];
    g.hp1 = g.H / 2; g.wp1 = g.W / 2; g.h2 = g.hp1 - 4; g.w2 = g.wp1 - 4; g.hp2 = g.h2 / 2; g.wp2 = g.w2 / 2;
    g.flat = g.hp2 * g.wp2 * 16;
    long long o = 0;
    g.b_c1 = (int)o; o += 6;   g.k_c1 = (int)o; o += 25LL * g.C * 6;
    g.b_c2 = (int)o; o += 16;  g.k_c2 = (int)o; o += 150LL * 16;
    g.b_f1 = (int)o; o += 120; g.k_f1 = (int)o; o += (long long)g.flat * 120;
    g.b_f2 = (int)o; o += 84;  g.k_f2 = (int)o; o += 120LL * 84;
    g.b_f3 = (int)o; o += g.K; g.k_f3 = (int)o; o += 84LL * g.K;
    if (o > 0x7fffffffLL) { delete s; return fail(MILE_ERR_INVALID, "parameter count exceeds int32"); }
    g.d = (int)o;
    ds.d = g.d;
    ds.widths[0] = g.K;
    ds.b_off[0] = g.b_c1; ds.w_off[0] = g.k_c1;
    ds.max_width = 150; ds.act_stride = 0;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) == hipSuccess && device < cnt) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, device) == hipSuccess) s->n_cu = prop.multiProcessorCount;
    }
    *out = s;
    return MILE_OK;
  }
  // ravel_pytree order: layers sorted by NAME ("layer10" < "layer2"), bias before kernel
  std::vector<int> order(spec->n_layers);
  for (int i = 0; i < spec->n_layers; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [](int a, int b) {
    return std::string("layer") + std::to_string(a) < std::string("layer") + std::to_string(b);
  });
  long long off = 0;
  for (int li : order) {
    const int fin = li == 0 ? spec->in_features : spec->widths[li - 1], fout = spec->widths[li];
    ds.b_off[li] = (int)off; off += fout;
    ds.w_off[li] = (int)off; off += (long long)fin * fout;
  }
  if (off > 0x7fffffffLL) { delete s; return fail(MILE_ERR_INVALID, "parameter count exceeds int32"); }
  if (off < 2) { delete s; return fail(MILE_ERR_INVALID, "The target distribution must have more than 1 dimension for MCLMC."); }
  ds.d = (int)off;
  ds.act_off[0] = 0;
  int a = spec->in_features, mw = spec->in_features;
  for (int l = 0; l < spec->n_layers; ++l) {
    ds.widths[l] = spec->widths[l];
    ds.act_off[l + 1] = a;
    a += spec->widths[l];
    mw = std::max(mw, spec->widths[l]);
  }
  ds.act_stride = a;
  ds.max_width = mw;
  if (generic_R(ds) < 1) { delete s; return fail(MILE_ERR_INVALID, "network too wide for the generic kernel"); }

  int cnt = 0;
  hipError_t e = hipGetDeviceCount(&cnt);
  if (e == hipSuccess && device < cnt) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) s->n_cu = prop.multiProcessorCount;
  }
  *out = s;
  return MILE_OK;
}

static void free_data(mile_sampler *s) {
  if (s->X) (void)hipFree(s->X);
  if (s->Xp) (void)hipFree(s->Xp);
  if (s->y) (void)hipFree(s->y);
  if (s->Xb) (void)hipFree(s->Xb);
  if (s->Xt) (void)hipFree(s->Xt);
  s->X = s->Xp = nullptr; s->y = nullptr; s->Xb = s->Xt = nullptr;
}
static void free_ws(mile_sampler *s) {
  if (s->slabs) (void)hipFree(s->slabs);
  if (s->llpart) (void)hipFree(s->llpart);
  if (s->dK) (void)hipFree(s->dK);
  if (s->upart) (void)hipFree(s->upart);
  s->upart = nullptr;
  if (s->lold) (void)hipFree(s->lold);
  if (s->alt_x) (void)hipFree(s->alt_x);
  if (s->alt_u) (void)hipFree(s->alt_u);
  if (s->alt_g) (void)hipFree(s->alt_g);
  if (s->alt_logp) (void)hipFree(s->alt_logp);
  s->alt_x = s->alt_u = s->alt_g = s->alt_logp = nullptr;
  if (s->tune_info) (void)hipFree(s->tune_info);
  s->tune_info = nullptr;
  if (s->arrive) (void)hipFree(s->arrive);
  s->arrive = nullptr;
  if (s->dbg_buf) (void)hipFree(s->dbg_buf);   // sized from the slab-row capacity
  s->dbg_buf = nullptr;
  s->slabs = s->llpart = s->dK = s->lold = nullptr;
  s->E_cap = 0; s->ES_cap = 0;
}

int32_t mile_destroy(mile_sampler *s) {
  if (!s) return MILE_OK;
  if (s->ev_X) (void)hipFree(s->ev_X);
  if (s->ev_Xp) (void)hipFree(s->ev_Xp);
  if (s->ev_y) (void)hipFree(s->ev_y);
  free_data(s);
  free_ws(s);
  if (s->gemm_ws) (void)hipFree(s->gemm_ws);
  if (s->gemm_ones) (void)hipFree(s->gemm_ones);
  if (s->wide_ws) (void)hipFree(s->wide_ws);
  if (s->wide_wt) (void)hipFree(s->wide_wt);
  if (s->wide_hb) (void)hipFree(s->wide_hb);
  if (s->dbg_buf) (void)hipFree(s->dbg_buf);
  if (s->tune_info) (void)hipFree(s->tune_info);
  if (s->blas && g_rb.destroy) (void)g_rb.destroy(s->blas);
  for (auto ev : s->ev) (void)hipEventDestroy(ev);
  delete s;
  return MILE_OK;
}

int64_t mile_param_count(const mile_sampler *s) { return s ? s->ds.d : -1; }

int32_t mile_param_offsets(const mile_sampler *s, int32_t layer, int64_t *bias_off, int64_t *kernel_off) {
  if (s && s->spec.model == MILE_MODEL_LENET) {   // layers 0..4 = conv1, conv2, fc1, fc2, fc3
    const LeNetGeom &g = s->lg;
    const int bo[5] = {g.b_c1, g.b_c2, g.b_f1, g.b_f2, g.b_f3}, ko[5] = {g.k_c1, g.k_c2, g.k_f1, g.k_f2, g.k_f3};
    if (layer < 0 || layer >= 5) return fail(MILE_ERR_INVALID, "mile_param_offsets: bad layer");
    if (bias_off) *bias_off = bo[layer];
    if (kernel_off) *kernel_off = ko[layer];
    return MILE_OK;
  }
  if (!s || layer < 0 || layer >= s->ds.n_layers) return fail(MILE_ERR_INVALID, "mile_param_offsets: bad layer");
  if (bias_off) *bias_off = s->ds.b_off[layer];
  if (kernel_off) *kernel_off = s->ds.w_off[layer];
  return MILE_OK;
}

__global__ void k_pad_x(const float *X, float *Xp, int N, int Npad, int F, int Fp) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)Npad * Fp) return;
  const int r = (int)(idx / Fp), c = (int)(idx % Fp);
  Xp[idx] = (r < N && c < F) ? X[(long long)r * F + c] : 0.0f;
}

__global__ void k_prep_bf16(const float *X, bf16 *Xb, bf16 *Xt, int N, int Npad, int F) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)Npad * 32) return;
  const int r = (int)(idx >> 5), c = (int)(idx & 31);
  const bf16 v = (bf16)((r < N && c < F) ? X[(long long)r * F + c] : 0.0f);
  if (c < 16) Xb[(long long)r * 16 + c] = v;
  Xt[(long long)c * Npad + r] = v;
}

int32_t mile_set_data(mile_sampler *s, const float *X, const void *y, int64_t N, void *stream) {
  if (!s || !X || !y) return fail(MILE_ERR_INVALID, "mile_set_data: null argument");
  if (N < 1 || N > 0x3fffffff) return fail(MILE_ERR_INVALID, "mile_set_data: N out of range");
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(s->device));
  const int F = s->spec.in_features;
  // Same number of rows as before (the warm-start stage hands over a reshuffled copy of the training set every epoch,
  // src/dataset/tabular.py:170-212): every buffer and the workspace keep their sizes -- overwrite in place, no
  // free / malloc cycle of ~15 allocations per epoch.
  const bool same_shape = s->X && s->N == (int)N;
  if (!same_shape) {
    free_data(s);
    free_ws(s);
  }
  s->N = (int)N;
  s->Npad = ((int)N + 31) / 32 * 32;
  s->Fp = (F + 7) / 8 * 8;
  if (!same_shape) {
    HIP_TRY(hipMalloc(&s->X, (size_t)N * F * 4));
    // (32 rows of slack: a row window that ends at the last row still reads whole 32-row blocks from its own first row)
    HIP_TRY(hipMalloc(&s->Xp, (size_t)(s->Npad + 32) * s->Fp * 4));
    HIP_TRY(hipMalloc(&s->y, (size_t)(s->Npad + 32) * 4));
  }
  HIP_TRY(hipMemsetAsync(s->Xp, 0, (size_t)(s->Npad + 32) * s->Fp * 4, st));
  HIP_TRY(hipMemcpyAsync(s->X, X, (size_t)N * F * 4, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemsetAsync(s->y, 0, (size_t)(s->Npad + 32) * 4, st));
  s->win_begin = s->win_count = 0;
  HIP_TRY(hipMemcpyAsync(s->y, y, (size_t)N * 4, hipMemcpyDeviceToDevice, st));
  const long long tot = (long long)s->Npad * s->Fp;
  k_pad_x<<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(s->X, s->Xp, s->N, s->Npad, F, s->Fp);
  HIP_TRY(hipGetLastError());
  if (w128b_supported(s->spec)) {
    s->Npb = ((int)N + 63) / 64 * 64;
    if (!same_shape) {
      HIP_TRY(hipMalloc(&s->Xb, (size_t)s->Npb * 16 * 2));
      HIP_TRY(hipMalloc(&s->Xt, (size_t)s->Npb * 32 * 2));
    }
    const long long tb = (long long)s->Npb * 32;
    k_prep_bf16<<<(unsigned)((tb + 255) / 256), 256, 0, st>>>(s->X, (bf16 *)s->Xb, (bf16 *)s->Xt, s->N, s->Npb, F);
    HIP_TRY(hipGetLastError());
  }
  return MILE_OK;
}

int32_t mile_set_row_window(mile_sampler *s, int64_t begin, int64_t count) {
  if (!s) return fail(MILE_ERR_INVALID, "null handle");
  if (!s->X) return fail(MILE_ERR_STATE, "mile_set_row_window: call mile_set_data first");
  if (begin < 0 || count < 0 || begin + count > s->N) return fail(MILE_ERR_INVALID, "mile_set_row_window: window outside the data");
  s->win_begin = count ? (int)begin : 0;
  s->win_count = (int)count;
  return MILE_OK;
}

int32_t mile_reserve(mile_sampler *s, int32_t E) {
  if (!s || E < 1) return fail(MILE_ERR_INVALID, "mile_reserve: bad argument");
  if (!s->X) return fail(MILE_ERR_STATE, "mile_reserve: call mile_set_data first");
  HIP_TRY(hipSetDevice(s->device));
  // capacity must cover whichever grad kernel is selected later
  int S = std::max(choose_S(s, E, MILE_GRAD_GENERIC), w64_supported(s->spec) ? choose_S(s, E, MILE_GRAD_MFMA_W64) : 1);
  if (w128b_supported(s->spec)) S = std::max(S, choose_S(s, E, MILE_GRAD_MFMA_W128_BF16));
  if (narrow_supported(s->spec)) S = std::max(S, choose_S(s, E, MILE_GRAD_MFMA_NARROW_F32));
  // A smaller ensemble splits the rows of a particle over MORE workgroups (S grows as E shrinks): capacity is counted in
  // slab rows E * S, and a later call with fewer particles must neither fail nor shrink what a larger one reserved.
  if (E <= s->E_cap && (size_t)E * S <= s->ES_cap) return MILE_OK;
  const int En = std::max(E, s->E_cap);
  const size_t ESn = std::max((size_t)E * S, s->ES_cap);
  free_ws(s);
  HIP_TRY(hipMalloc(&s->slabs, ESn * ((s->ds.d + 3) / 4 * 4) * 4));
  HIP_TRY(hipMalloc(&s->llpart, ESn * 4));
  HIP_TRY(hipMalloc(&s->dK, (size_t)En * 4));
  HIP_TRY(hipMalloc(&s->lold, (size_t)En * 4));
  HIP_TRY(hipMalloc(&s->upart, ((size_t)En * ((s->ds.d + UPD_SEG - 1) / UPD_SEG) * UPD_NSUM + (size_t)En * 8) * 4));
  HIP_TRY(hipMalloc(&s->alt_x, (size_t)En * s->ds.d * 4));
  HIP_TRY(hipMalloc(&s->alt_u, (size_t)En * s->ds.d * 4));
  HIP_TRY(hipMalloc(&s->alt_g, (size_t)En * s->ds.d * 4));
  HIP_TRY(hipMalloc(&s->alt_logp, (size_t)En * 4));
  HIP_TRY(hipMalloc(&s->arrive, ((size_t)En * 4 + 15) / 16 * 16));
  HIP_TRY(hipMemset(s->arrive, 0, ((size_t)En * 4 + 15) / 16 * 16));
  s->E_cap = En;
  s->ES_cap = ESn;
  return MILE_OK;
}

int32_t mile_set_grad_kernel(mile_sampler *s, int32_t which) {
  if (!s) return fail(MILE_ERR_INVALID, "null handle");
  if (which < MILE_GRAD_AUTO || which > MILE_GRAD_MFMA_NARROW_F32) return fail(MILE_ERR_INVALID, "unknown grad kernel");
  if (which == MILE_GRAD_MFMA_NARROW_F32 && !narrow_supported(s->spec))
    return fail(MILE_ERR_INVALID, "MFMA_NARROW_F32 needs an FCN with 1-3 hidden layers of width <= 64 and F <= 64 (or 4-10 of width <= 16 and F <= 16) and <= 16 outputs");
  if ((which == MILE_GRAD_MFMA_WIDE_BF16X3 || which == MILE_GRAD_MFMA_WIDE_BF16) && s->spec.model != MILE_MODEL_FCN)
    return fail(MILE_ERR_INVALID, "MFMA_WIDE_* are FCN kernels");
  if ((s->spec.model == MILE_MODEL_LENET) != (which == MILE_GRAD_LENET_F32 || which == MILE_GRAD_LENET_BF16) && which != MILE_GRAD_AUTO)
    return fail(MILE_ERR_INVALID, "LENET_F32 / LENET_BF16 are the kernels of MILE_MODEL_LENET, and its only ones");
  if (which == MILE_GRAD_LENET_BF16 && s->lg.C > 4) return fail(MILE_ERR_INVALID, "LENET_BF16 needs <= 4 image channels");
  if (which == MILE_GRAD_GEMM_F32 && !rocblas_load()) return fail(MILE_ERR_HIP, "GEMM_F32 needs librocblas.so, which could not be loaded");
  if (which == MILE_GRAD_MFMA_W128_BF16 && !w128b_supported(s->spec))
    return fail(MILE_ERR_INVALID, "MFMA_W128_BF16 needs ReLU regression with 1-3 hidden layers of width 128 and F <= 16");
  if (which == MILE_GRAD_MFMA_W64 && !w64_supported(s->spec))
    return fail(MILE_ERR_INVALID, "MFMA_W64 needs ReLU regression with 1-3 hidden layers of width 64 and F <= 16");
  if (which == MILE_GRAD_MFMA_W64_BF16X3 && !w64x3_supported(s->spec))
    return fail(MILE_ERR_INVALID, "MFMA_W64_BF16X3 needs ReLU regression with 2-3 hidden layers of width 64 and F <= 16");
  s->grad_kernel = which;
  return MILE_OK;
}
int32_t mile_get_grad_kernel(const mile_sampler *s) { return s ? resolved_kernel(s) : MILE_ERR_INVALID; }

}  // extern "C"

// ------------------------------------------------------------------------------------
// grad launch
// ------------------------------------------------------------------------------------
static int ptr_align(const void *q) {   // alignment in floats (4, 2 or 1) of a device pointer
  const uintptr_t a = (uintptr_t)q;
  return (a & 15) == 0 ? 4 : ((a & 7) == 0 ? 2 : 1);
}

template <int AL, bool SDC>
static void launch_update_al(const UpdParams &u, int E, int nk, hipStream_t st) {
  // fewest waves that still leave <= nk quads per thread: padded lanes are wasted VALU time in a kernel that only fills
  // half the chip (d = 8834: 2208 quads, nk = 3 -> 768 threads instead of 1024)
  const int nqf = u.d >> 2;
  const int nt = std::min(UPD_NT, ((nqf + nk - 1) / nk + 63) / 64 * 64);
  const int kind = SDC ? -1 : upd_kind(u);
#define MILE_UPD_CASE(NK_)                                                                       \
  if constexpr (!SDC) {                                                                          \
    if (kind == UPD_KIND_MID) { k_update_fast<NK_, AL, SDC, UPD_KIND_MID><<<E, nt, 0, st>>>(u); break; } \
    if (kind == UPD_KIND_REC) { k_update_fast<NK_, AL, SDC, UPD_KIND_REC><<<E, nt, 0, st>>>(u); break; } \
    if (kind == UPD_KIND_TUNE) {                                                                 \
      if (nt <= 768) k_update_fast<NK_, AL, SDC, UPD_KIND_TUNE, 768><<<E, nt, 0, st>>>(u);       \
      else k_update_fast<NK_, AL, SDC, UPD_KIND_TUNE><<<E, nt, 0, st>>>(u);                      \
      break;                                                                                     \
    }                                                                                            \
  }                                                                                              \
  if (nt <= 768) k_update_fast<NK_, AL, SDC, -1, 768><<<E, nt, 0, st>>>(u);                      \
  else k_update_fast<NK_, AL, SDC><<<E, nt, 0, st>>>(u);                                         \
  break;
  switch (nk) {
    case 1: MILE_UPD_CASE(1)
    case 2: MILE_UPD_CASE(2)
    case 3: MILE_UPD_CASE(3)
    default: MILE_UPD_CASE(4)
  }
#undef MILE_UPD_CASE
}

// d beyond the register cache, up to 4 * UPD_NT * UPD_QMAX_BIG: k_update_big (x, u in registers, g in LDS)
template <int AL, bool SDC>
static bool launch_update_big_al(const UpdParams &u, int E, int nk, hipStream_t st) {
  const int nqf = u.d >> 2;
  const int nt = std::min(UPD_NT, ((nqf + nk - 1) / nk + 63) / 64 * 64);
  const size_t lds = (size_t)nk * nt * 16;
  // no compile-time launch kinds here: with the flag tests folded the unrolled quads become one basic block, the compiler
  // interleaves them and spills 109 registers (NK = 9, record kind) against 7 with the run-time flags
#define MILE_UPD_BIG(NK_, CF_)                                                                                     \
  {                                                                                                                \
    static bool attr = false;                                                                                      \
    if (!attr) {                                                                                                   \
      if (hipFuncSetAttribute((const void *)k_update_big<NK_, AL, SDC, CF_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              UPD_NT * NK_ * 16) != hipSuccess) return false;                                      \
      attr = true;                                                                                                 \
    }                                                                                                              \
    k_update_big<NK_, AL, SDC, CF_><<<E, nt, lds, st>>>(u);                                                        \
    return true;                                                                                                   \
  }
#define MILE_UPD_BIG_CASE(NK_) \
  case NK_:                    \
    MILE_UPD_BIG(NK_, -1)
  switch (nk) {
    MILE_UPD_BIG_CASE(5)
    MILE_UPD_BIG_CASE(6)
    MILE_UPD_BIG_CASE(7)
    MILE_UPD_BIG_CASE(8)
    MILE_UPD_BIG_CASE(9)
  }
#undef MILE_UPD_BIG_CASE
#undef MILE_UPD_BIG
  return false;
}

static bool launch_update_big(const UpdParams &u, int E, hipStream_t st) {
  const int nqf = u.d >> 2;
  const int nk = (nqf + UPD_NT - 1) / UPD_NT;
  if (nk <= UPD_QMAX || nk > UPD_QMAX_BIG || (u.flags & UPD_TUNE) || u.zA || u.zB || getenv("MILE_NO_UPD_BIG")) return false;
  int al = (u.d % 4 == 0) ? 4 : ((u.d % 2 == 0) ? 2 : 1);
  const void *ptrs[] = {u.x, u.u, u.g, u.slabs, u.sdc, u.zA, u.zB, u.out_sample, u.x_in, u.u_in, u.g_in};
  for (const void *q : ptrs)
    if (q) al = std::min(al, ptr_align(q));
  if (u.sdc) return al == 4 ? launch_update_big_al<4, true>(u, E, nk, st) : (al == 2 ? launch_update_big_al<2, true>(u, E, nk, st) : launch_update_big_al<1, true>(u, E, nk, st));
  return al == 4 ? launch_update_big_al<4, false>(u, E, nk, st) : (al == 2 ? launch_update_big_al<2, false>(u, E, nk, st) : launch_update_big_al<1, false>(u, E, nk, st));
}

static void launch_update(const UpdParams &u, int E, hipStream_t st) {
  const int nqf = u.d >> 2;
  if (nqf < 1 || nqf > UPD_NT * UPD_QMAX) {
    if (nqf >= 1 && launch_update_big(u, E, st)) return;
    if ((u.d + 3) / 4 <= UPD_NT * UPD_QMAX) { k_update<true><<<E, UPD_NT, 0, st>>>(u); return; }
    if (u.upart && getenv("MILE_NO_UPD_SEG") == nullptr) {   // any d, all CUs: segments of UPD_SEG elements, sums -> chain + apply -> scalars
      const int nseg = (u.d + UPD_SEG - 1) / UPD_SEG;
      k_update_seg<1><<<dim3(nseg, E), UPD_NT, 0, st>>>(u, u.upart, nseg);
      k_update_seg<2><<<dim3(nseg, E), UPD_NT, 0, st>>>(u, u.upart, nseg);
      k_update_seg_scalars<<<(E + 255) / 256, 256, 0, st>>>(u, u.upart, nseg, E);
      return;
    }
    k_update<false><<<E, UPD_NT, 0, st>>>(u);
    return;
  }
  // every row base is (pointer + e*d): vector width allowed by d and by the pointers
  int al = (u.d % 4 == 0) ? 4 : ((u.d % 2 == 0) ? 2 : 1);
  const void *ptrs[] = {u.x, u.u, u.g, u.slabs, u.sdc, u.zA, u.zB, u.out_sample, u.x_in, u.u_in, u.g_in, u.t_avg, u.u_rec};
  for (const void *q : ptrs)
    if (q) al = std::min(al, ptr_align(q));
  const int nk = (nqf + UPD_NT - 1) / UPD_NT;
  if (u.sdc) {
    if (al == 4) launch_update_al<4, true>(u, E, nk, st);
    else if (al == 2) launch_update_al<2, true>(u, E, nk, st);
    else launch_update_al<1, true>(u, E, nk, st);
  } else {
    if (al == 4) launch_update_al<4, false>(u, E, nk, st);
    else if (al == 2) launch_update_al<2, false>(u, E, nk, st);
    else launch_update_al<1, false>(u, E, nk, st);
  }
}

// The F > 8 split kernels live in their own translation unit, built WITHOUT -amdgpu-mfma-vgpr-form: with that option
// hipcc 7.2's 'AMDGPU Rewrite AGPR-Copy-MFMA' pass crashes on the heavily spilling k_grad_w64<3,2,true> (mile_amd/_build.py).
hipError_t mile_launch_w64_split_fq2(int nh, const GradParams &gp, int E, hipStream_t st);

template <int NH, int FQ, bool SPLIT = false>
static hipError_t launch_w64(const GradParams &gp, const W64Fuse &fz, int E, hipStream_t st) {
  using LY = W64Layout<NH, FQ, SPLIT>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)k_grad_w64<NH, FQ, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, LY::BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  if constexpr (W64FuseArg<FQ, SPLIT>::FUSABLE) {
    k_grad_w64<NH, FQ, SPLIT><<<dim3(gp.S, E), 256, LY::BYTES, st>>>(gp, fz);
  } else {
    if (fz.enabled) return hipErrorInvalidValue;
    k_grad_w64<NH, FQ, SPLIT><<<dim3(gp.S, E), 256, LY::BYTES, st>>>(gp, W64NoFuse{0});
  }
  return hipGetLastError();
}

// Can the update that follows this gradient run as the grad launch's epilogue (k_grad_w64 SPLIT, W64Fuse)?
// 8-byte aligned rows everywhere (the epilogue is the AL = 2 form), no preconditioner, d within its register cache.
static bool fuse_ok(const mile_sampler *s, int kernel, const UpdParams &u) {
  if (!MILE_W64_EPILOGUE_ON || kernel != MILE_GRAD_MFMA_W64_BF16X3 || getenv("MILE_NO_FUSE")) return false;
  if (u.sdc || (u.d & 1) || u.u_rec) return false;   // (the merged warm-up launch exists as a stand-alone kernel only)
  const int nh = s->spec.n_layers - 1, fq = s->Fp / 8;
  if (fq != 1) return false;                       // the F > 8 kernels are built without the epilogue (mile_grad_w64.h)
  const int nk = nh == 2 ? w64_fuse_nk<2, 1>() : w64_fuse_nk<3, 1>();
  if ((u.d >> 2) > 256 * nk) return false;
  const void *ptrs[] = {u.x, u.u, u.g, u.zA, u.zB, u.out_sample, u.x_in, u.u_in, u.g_in, u.t_avg, u.bk_x, u.bk_u, u.bk_g};
  for (const void *q : ptrs)
    if (q && ptr_align(q) < 2) return false;
  return true;
}

template <int NH, int TH, int TF>
static hipError_t launch_narrow_t(const GradParams &gp, int E, int nw, hipStream_t st) {
  constexpr bool WL = TH >= 3;
  using LY = NarrowLayout<NH, TH, TF, WL>;
  if constexpr (WL) {
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute((const void *)k_grad_narrow<NH, TH, TF, WL>, hipFuncAttributeMaxDynamicSharedMemorySize, LY::BYTES);
      if (e != hipSuccess) return e;
      attr_set = true;
    }
  }
  k_grad_narrow<NH, TH, TF, WL><<<dim3(gp.S, E), 64 * nw, LY::BYTES, st>>>(gp);
  return hipGetLastError();
}
static hipError_t launch_narrow(const mile_sampler *s, const GradParams &gp, int E, hipStream_t st) {
  const int nh = s->spec.n_layers - 1;
  const int th = narrow_tiles_hidden(s->spec), tf = s->spec.in_features <= 16 ? 1 : 4;
  const int nw = th >= 3 ? NRW_MAXW : narrow_waves(s, gp.S, gp.N);
#define MILE_NRW(NH_, TH_, TF_) if (nh == NH_ && th == TH_ && tf == TF_) return launch_narrow_t<NH_, TH_, TF_>(gp, E, nw, st);
  MILE_NRW(1, 1, 1) MILE_NRW(1, 2, 1) MILE_NRW(1, 1, 4) MILE_NRW(1, 2, 4)
  MILE_NRW(2, 1, 1) MILE_NRW(2, 2, 1) MILE_NRW(2, 1, 4) MILE_NRW(2, 2, 4)
  MILE_NRW(3, 1, 1) MILE_NRW(3, 2, 1) MILE_NRW(3, 1, 4) MILE_NRW(3, 2, 4)
  MILE_NRW(4, 1, 1) MILE_NRW(5, 1, 1) MILE_NRW(6, 1, 1) MILE_NRW(7, 1, 1) MILE_NRW(8, 1, 1) MILE_NRW(9, 1, 1) MILE_NRW(10, 1, 1)
  MILE_NRW(1, 3, 1) MILE_NRW(1, 4, 1) MILE_NRW(1, 3, 4) MILE_NRW(1, 4, 4)      // hidden widths 33..64: weights in LDS
  MILE_NRW(2, 3, 1) MILE_NRW(2, 4, 1) MILE_NRW(2, 3, 4) MILE_NRW(2, 4, 4)
  MILE_NRW(3, 3, 1) MILE_NRW(3, 4, 1) MILE_NRW(3, 3, 4) MILE_NRW(3, 4, 4)
#undef MILE_NRW
  return hipErrorInvalidValue;
}

template <int NH>
static hipError_t launch_w128b(const GradParams &gp, int E, hipStream_t st) {
  using LY = W128Layout<NH, 2>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)k_grad_w128b<NH, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LY::BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  if (gp.dbg & 16) {   // dev: per-phase cycle counts from workgroup 0 (MILE_DEBUG=16)
    static bool attr_t = false;
    if (!attr_t) {
      hipError_t e = hipFuncSetAttribute((const void *)k_grad_w128b<NH, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LY::BYTES);
      if (e != hipSuccess) return e;
      attr_t = true;
    }
    k_grad_w128b<NH, 2, true><<<dim3(gp.S, E), 256, LY::BYTES, st>>>(gp);
    long long hb[8];
    if (gp.dbg_buf && hipMemcpy(hb, gp.dbg_buf, 64, hipMemcpyDeviceToHost) == hipSuccess)
      fprintf(stderr, "w128b cycles per tile pair (%lld pairs): F1 %lld F2 %lld F3 %lld headbwd %lld L(1) %lld L(2) %lld first %lld\n", hb[7],
              hb[0], hb[1], hb[2], hb[3], hb[4], hb[5], hb[6]);
    return hipGetLastError();
  }
  k_grad_w128b<NH, 2><<<dim3(gp.S, E), 256, LY::BYTES, st>>>(gp);
  return hipGetLastError();
}

template <int NH, int FQ, bool SPLIT = false>
static int w64_lds_bytes() { return W64Layout<NH, FQ, SPLIT>::BYTES; }

__global__ void k_fill(float *p, float v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// LeNet: forward (+ backward when slab != nullptr) over image chunks; batch = particles (or samples).
// out_ll != nullptr: evaluation, per-row log-likelihoods out_ll[(s0 + e) * N + r]; otherwise the gradient goes
// to slab[e * dp + offset] and the log-likelihood sums to llacc[e].
// Dense half of LeNet on the hand-written MFMA GEMMs (k_mm3 with fp32-faithful three-term products; defined below the k_mm3
// launchers): LENET_BF16 uses no library kernel.  lenet_dense_prep once per gradient, lenet_dense_chunk per image chunk.
struct LeNetDenseBufs { const float *p2; float *f1, *f2, *out, *df2, *df1, *dp2; };
static int lenet_dense_prep(mile_sampler *s, const float *theta, int E, hipStream_t st);
static int lenet_dense_chunk(mile_sampler *s, const float *theta, int E, int Rc, int r0, int chunk, const LeNetDenseBufs &b, const void *y,
                             float *slab, long long dp, float *llacc, float *out_ll, long long Ntot, long long s0, hipStream_t st);

static int run_lenet(mile_sampler *s, const float *theta, int E, const float *X, const void *y, int N, float *slab, long long dp,
                     float *llacc, float *out_ll, long long s0, hipStream_t st, bool mfma = false) {
  if (!rocblas_load()) return fail(MILE_ERR_HIP, "librocblas.so could not be loaded");
  const LeNetGeom &g = s->lg;
  const int d = g.d, act = s->ds.activation, task = s->ds.task;
  const bool grad = slab != nullptr;
  if (!s->blas && g_rb.create(&s->blas) != 0) { s->blas = nullptr; return fail(MILE_ERR_HIP, "rocblas_create_handle failed"); }
  if (g_rb.set_stream(s->blas, st) != 0) return fail(MILE_ERR_HIP, "rocblas_set_stream failed");
  const size_t HW = (size_t)g.H * g.W, P1 = (size_t)g.hp1 * g.wp1, HW2 = (size_t)g.h2 * g.w2;
  const size_t n_a1 = 6 * HW, n_p1 = 6 * P1, n_col2 = 150 * HW2, n_a2 = 16 * HW2, n_p2 = g.flat;
  // direct convolution kernels (LDS tiles per image) unless the image is too large for them or the im2col + SGEMM
  // form is asked for (MILE_LENET_GEMM=1: kept as the second implementation / fallback)
  const int ipw = getenv("MILE_LENET_IPW") ? std::max(1, atoi(getenv("MILE_LENET_IPW"))) : (mfma ? 16 : 8);   // images per workgroup (measured: 53.0 / 54.7 ms at 16 / 8 on the MFMA forms)
  const size_t lds_f1 = (size_t)(25 * g.C * 8 + 8 + g.C * (g.H + 4) * (g.W + 4)) * 4;
  const int KT1 = 25 * g.C;
  // k_conv5_dw: pixel groups are reduced 3 at a time through LDS (buffer aliases the tiles)
  const size_t lds_w1 = std::max((size_t)(g.C * (g.H + 4) * (g.W + 4)) + HW * 6, 3 * (size_t)(KT1 * 6 + 6)) * 4;
  const size_t lds_f2 = (size_t)(150 * 16 + 16 + 6 * P1) * 4, lds_x2 = (size_t)(2400 + 1 * (g.h2 + 8) * (g.w2 + 8) * 16) * 4;   // 2 images' dZ per pass: two workgroups per CU
  const size_t lds_w2 = std::max(6 * P1 + HW2 * 16, 3 * (size_t)(2400 + 16)) * 4;
  const size_t lds_max = std::max({lds_f1, lds_w1, lds_f2, lds_x2, lds_w2});
  // compile-time geometry for CIFAR- / MNIST-sized inputs (their LDS needs are below the 64 KiB default limit)
  const int geo = (g.C == 3 && g.H == 32 && g.W == 32) ? 1 : (g.C == 1 && g.H == 28 && g.W == 28) ? 3 : 0;
  const bool direct = getenv("MILE_LENET_GEMM") == nullptr && g.C <= 16 && lds_max <= 150 * 1024;
  // MILE_GRAD_LENET_BF16: the five convolution launches on the bf16 matrix pipe (mile_lenet_mfma.h), everything else as below
  // small images: several per barrier pair (as many as keep the tiles under ~56 KB, at most 4)
  const size_t per_img2 = (size_t)(g.hp1 * g.wp1 + 8) * 16 + (size_t)g.h2 * g.w2 * 32;   // input tile + dZ rows of one image
  const int ni2 = getenv("MILE_CM_NI") ? std::max(1, std::min(8, atoi(getenv("MILE_CM_NI")))) : std::max(1, std::min(4, (int)(57344 / per_img2)));
  const size_t ldm_f1 = cm_lds_fwd(CM_IN4, g.H, g.W, 2, g.C, 6), ldm_f2 = cm_lds_fwd(CM_IN8, g.hp1, g.wp1, 0, 6, 16, ni2), ldm_x2 = cm_lds_dx(g.h2, g.w2, 6, 16);
  const size_t ldm_w1 = cm_lds_dw(CM_IN4, g.H, g.W, 2), ldm_w2 = cm_lds_dw(CM_IN8, g.hp1, g.wp1, 0, ni2);
  // the 6-channel sides of the two convolutions in the PAIR forms (two pixels per MFMA row / column); MILE_CM_NO_PAIR: the pixel forms
  const bool pair = mfma && getenv("MILE_CM_NO_PAIR") == nullptr;
  // ReLU: the full-size conv activations are kept for the backward pass only as two-byte "was it positive" values
  const int a16 = mfma && act == MILE_ACT_RELU && getenv("MILE_CM_A32") == nullptr;
  const auto ni_for = [](size_t per_img) { return getenv("MILE_CM_NI") ? std::max(1, std::min(8, atoi(getenv("MILE_CM_NI")))) : std::max(1, std::min(4, (int)(57344 / per_img))); };
  const int ni_x2 = ni_for((size_t)((g.h2 + 8) * (g.w2 + 8) + 8) * 32), ni_f1 = ni_for((size_t)((g.H + 4) * (g.W + 4) + 8) * 8);
  const size_t ldm_x2p = cm_lds_dx2x(g.h2, g.w2, 6, 16, ni_x2), ldm_w1p = cm_lds_dw2x(g.H, g.W, 2), ldm_f1p = cm_lds_fwd(CM_IN4, g.H, g.W, 2, g.C, 6, ni_f1);
  if (mfma && (!direct || g.C > 4 || std::max({ldm_f1, ldm_f2, ldm_x2, ldm_w1, ldm_w2, ldm_x2p, ldm_w1p, ldm_f1p}) > 150 * 1024))
    return fail(MILE_ERR_INVALID, "LENET_BF16 needs <= 4 image channels and an image that fits the LDS tiles");
  // Dense activations: exact widths on the rocBLAS path; k_mm3 wants rows of 8 k floats (zero padding columns)
  const size_t w_f2 = mfma ? 88 : 84, w_out = mfma ? (size_t)(g.K + 7) / 8 * 8 : (size_t)g.K;
  size_t per = n_a1 + n_p1 + (direct ? 0 : n_col2) + n_a2 + n_p2 + 120 + w_f2 + w_out;
  if (grad) per += w_f2 + 120 + n_p2 + n_p1 + (direct ? 0 : n_a2 + n_a1);
  const size_t shared = direct ? 0 : 25 * (size_t)g.C * HW;
  size_t R = ((size_t)1 << 30) / ((size_t)E * per + shared);
  R = std::max<size_t>(1, std::min<size_t>(R, (size_t)N));
  if (const char *rv = getenv("MILE_GEMM_ROWS")) R = std::max<size_t>(1, std::min<size_t>((size_t)atoll(rv), (size_t)N));
  const size_t nwg_max = (R + ipw - 1) / ipw;
  const size_t n_part1 = direct && grad ? (size_t)E * nwg_max * (25 * g.C * 6 + 6) : 0;
  const size_t n_part2 = direct && grad ? (size_t)E * nwg_max * (2400 + 16) : 0;
  const size_t need = R * ((size_t)E * per + shared) + n_part1 + n_part2 + 128;   // + the arrays' alignment padding
  if (need > s->gemm_ws_floats) {
    if (s->gemm_ws) (void)hipFree(s->gemm_ws);
    s->gemm_ws = nullptr; s->gemm_ws_floats = 0;
    HIP_TRY(hipMalloc(&s->gemm_ws, need * 4));
    s->gemm_ws_floats = need;
  }
  s->gemm_E = 0;
  const size_t n_ones = R * HW;
  if ((size_t)s->gemm_ones_n < n_ones) {
    if (s->gemm_ones) (void)hipFree(s->gemm_ones);
    s->gemm_ones = nullptr; s->gemm_ones_n = 0;
    HIP_TRY(hipMalloc(&s->gemm_ones, n_ones * 4));
    k_fill<<<(unsigned)((n_ones + 255) / 256), 256, 0, st>>>(s->gemm_ones, 1.0f, (int)n_ones);
    s->gemm_ones_n = (int)n_ones;
  }
  const size_t ER = (size_t)E * R;
  float *q = s->gemm_ws;
  auto take = [&](size_t n) { float *r = q; q += (n + 3) / 4 * 4; return r; };   // every array 16-byte aligned (vector loads in the conv kernels)
  float *col1 = take(R * shared), *a1 = take(ER * n_a1), *p1 = take(ER * n_p1), *col2 = take(direct ? 0 : ER * n_col2), *a2 = take(ER * n_a2);
  float *p2 = take(ER * n_p2), *f1 = take(ER * 120), *f2 = take(ER * w_f2), *out = take(ER * w_out);
  float *df2 = nullptr, *df1 = nullptr, *dp2 = nullptr, *dz2 = nullptr, *dp1 = nullptr, *dz1 = nullptr;
  if (grad) { df2 = take(ER * w_f2); df1 = take(ER * 120); dp2 = take(ER * n_p2); dz2 = take(direct ? 0 : ER * n_a2); dp1 = take(ER * n_p1); dz1 = take(direct ? 0 : ER * n_a1); }
  float *part1 = take(n_part1), *part2 = take(n_part2);
  if (direct) {
    static bool attr_done = false;
    if (!attr_done) {
      HIP_TRY(hipFuncSetAttribute((const void *)k_conv5_fwd<6, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void *)k_conv5_fwd<16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void *)k_conv5_dw<6, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void *)k_conv5_dw<16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void *)k_conv5_dx<6, 16, 1, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void *)k_conv5m_fwd<CM_IN4, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void *)k_conv5m_fwd<CM_IN8, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void *)k_conv5m_dw<CM_IN4, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void *)k_conv5m_dw<CM_IN8, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void *)k_conv5m_dx<6, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void *)k_conv5m_fwd2x<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void *)k_conv5m_dx2x<6, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void *)k_conv5m_dw2x<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      attr_done = true;
    }
  }
  const int cm_dbg = getenv("MILE_CM_SKIP") ? atoi(getenv("MILE_CM_SKIP")) : 0;   // dev: timing knobs of k_conv5m_fwd
  const float one = 1.0f, zero = 0.0f;
  auto blocks = [](long long n) { return (unsigned)std::min<long long>((n + 255) / 256, 65535); };
  // row-major C[rows x fout] = A[rows x fin] W[fin x fout] per batch entry; W from theta (+ offset), batch stride d
  auto fwd = [&](int w_off, int fin, int fout, const float *in, long long sin, long long rows, float *o) {
    return g_rb.sgemm_sb(s->blas, RB_OP_N, RB_OP_N, fout, (int)rows, fin, &one, theta + w_off, fout, d, in, fin, sin, &zero, o, fout,
                         rows * fout, E);
  };
  auto bias_act = [&](float *z, int b_off, int W, long long rows, int apply) {
    k_gemm_bias_act<<<dim3(blocks(rows * W), E), 256, 0, st>>>(z, theta, b_off, d, W, rows * W, act, apply);
  };
  if (mfma) {   // padding columns of the Dense activations are read as operands: zero them (never written afterwards)
    HIP_TRY(hipMemsetAsync(f1, 0, (ER * 120 + 4 + ER * w_f2 + 4 + ER * w_out) * 4, st));
    if (grad) HIP_TRY(hipMemsetAsync(df2, 0, (ER * w_f2 + 4 + ER * 120) * 4, st));
    const int rc = lenet_dense_prep(s, theta, E, st);
    if (rc) return rc;
  }
  for (int r0 = 0, chunk = 0; r0 < N; r0 += (int)R, ++chunk) {
    const long long Rc = std::min<long long>((long long)R, N - r0), B = (long long)E * Rc;
    const long long M1 = Rc * (long long)HW, M2 = Rc * (long long)HW2;
    // ---- forward
    const float *Xc = X + (size_t)r0 * g.C * HW;
    const unsigned nwg = (unsigned)((Rc + ipw - 1) / ipw);
    if (direct) {
#define LAUNCH_FWD1(GEO_) k_conv5_fwd<6, GEO_><<<dim3(nwg, E), 256, lds_f1, st>>>(Xc, 0, (long long)g.C * HW, g.W, 1, (long long)HW, g.C, g.H, g.W, 2, theta, g.k_c1, g.b_c1, d, a1, (int)Rc, ipw, act)
#define LAUNCH_FWD2(GEO_) k_conv5_fwd<16, GEO_><<<dim3(nwg, E), 256, lds_f2, st>>>(p1, Rc * (long long)n_p1, (long long)n_p1, g.wp1 * 6, 6, 1, 6, g.hp1, g.wp1, 0, theta, g.k_c2, g.b_c2, d, a2, (int)Rc, ipw, act)
      // MFMA form: the pooled activations come out of the convolution's own epilogue; evaluation skips the full-size ones
      if (pair) k_conv5m_fwd2x<6><<<dim3(nwg, E), 256, ldm_f1p, st>>>(Xc, 0, (long long)g.C * HW, g.W, 1, (long long)HW, g.C, g.H, g.W, 2, theta, g.k_c1, g.b_c1, d, grad ? a1 : nullptr, p1, (int)Rc, ipw, act, a16, ni_f1);
      else if (mfma) k_conv5m_fwd<CM_IN4, 6><<<dim3(nwg, E), 256, ldm_f1, st>>>(Xc, 0, (long long)g.C * HW, g.W, 1, (long long)HW, g.C, g.H, g.W, 2, theta, g.k_c1, g.b_c1, d, grad ? a1 : nullptr, p1, (int)Rc, ipw, act, cm_dbg, a16);
      else if (geo == 1) LAUNCH_FWD1(1); else if (geo == 3) LAUNCH_FWD1(3); else LAUNCH_FWD1(0);
      if (!mfma) k_avgpool2<<<blocks(B * (long long)n_p1), 256, 0, st>>>(a1, p1, B, g.H, g.W, 6);
      if (mfma) k_conv5m_fwd<CM_IN8, 16><<<dim3(nwg, E), 256, ldm_f2, st>>>(p1, Rc * (long long)n_p1, (long long)n_p1, g.wp1 * 6, 6, 1, 6, g.hp1, g.wp1, 0, theta, g.k_c2, g.b_c2, d, grad ? a2 : nullptr, p2, (int)Rc, ipw, act, cm_dbg, a16, ni2);
      else if (geo == 1) LAUNCH_FWD2(2); else if (geo == 3) LAUNCH_FWD2(4); else LAUNCH_FWD2(0);
#undef LAUNCH_FWD1
#undef LAUNCH_FWD2
    } else {
      k_im2col5<<<blocks(M1 * 25 * g.C), 256, 0, st>>>(Xc, col1, Rc, g.H, g.W, g.C, g.H, g.W, 2, (long long)g.C * HW, g.W, 1, (long long)HW);
      if (fwd(g.k_c1, 25 * g.C, 6, col1, 0, M1, a1)) return fail(MILE_ERR_HIP, "rocblas sgemm (conv1) failed");
      bias_act(a1, g.b_c1, 6, M1, 1);
      k_avgpool2<<<blocks(B * (long long)n_p1), 256, 0, st>>>(a1, p1, B, g.H, g.W, 6);
      k_im2col5<<<blocks(B * (long long)n_col2), 256, 0, st>>>(p1, col2, B, g.hp1, g.wp1, 6, g.h2, g.w2, 0, (long long)n_p1, g.wp1 * 6, 6, 1);
      if (fwd(g.k_c2, 150, 16, col2, M2 * 150, M2, a2)) return fail(MILE_ERR_HIP, "rocblas sgemm (conv2) failed");
      bias_act(a2, g.b_c2, 16, M2, 1);
    }
    if (!mfma) k_avgpool2<<<blocks(B * (long long)n_p2), 256, 0, st>>>(a2, p2, B, g.h2, g.w2, 16);
    if (mfma) {   // Dense layers, head and their backward pass on k_mm3; leaves dp2 for the convolution backward pass below
      const LeNetDenseBufs db{p2, f1, f2, out, df2, df1, dp2};
      const int rc = lenet_dense_chunk(s, theta, E, (int)Rc, r0, chunk, db, y, slab, dp, llacc, out_ll, N, s0, st);
      if (rc) return rc;
      if (!grad) continue;
    } else {
      if (fwd(g.k_f1, g.flat, 120, p2, Rc * g.flat, Rc, f1)) return fail(MILE_ERR_HIP, "rocblas sgemm (fc1) failed");
      bias_act(f1, g.b_f1, 120, Rc, 1);
      if (fwd(g.k_f2, 120, 84, f1, Rc * 120, Rc, f2)) return fail(MILE_ERR_HIP, "rocblas sgemm (fc2) failed");
      bias_act(f2, g.b_f2, 84, Rc, 1);
      if (fwd(g.k_f3, 84, g.K, f2, Rc * 84, Rc, out)) return fail(MILE_ERR_HIP, "rocblas sgemm (fc3) failed");
      bias_act(out, g.b_f3, g.K, Rc, 0);
      if (!grad) {
        k_gemm_rowll<<<dim3((unsigned)((Rc + 255) / 256), E), 256, 0, st>>>(out, y, r0, (int)Rc, g.K, task, out_ll, N, s0);
        continue;
      }
      k_gemm_head<<<E, 256, 0, st>>>(out, y, r0, (int)Rc, g.K, task, llacc, chunk == 0);
    }
    // ---- backward
    const float *beta = chunk == 0 ? &zero : &one;
    // dW[fin x fout] (+)= in^T dz into the slab; bias gradient = dz^T 1
    auto dW = [&](int w_off, int b_off, int fin, int fout, const float *in, long long sin, const float *dz, long long rows) -> int {
      if (g_rb.sgemm_sb(s->blas, RB_OP_N, RB_OP_T, fout, fin, (int)rows, &one, dz, fout, rows * fout, in, fin, sin, beta, slab + w_off, fout,
                        dp, E))
        return 1;
      return g_rb.sgemm_sb(s->blas, RB_OP_N, RB_OP_N, fout, 1, (int)rows, &one, dz, fout, rows * fout, s->gemm_ones, (int)rows, 0, beta,
                           slab + b_off, fout, dp, E);
    };
    // dX[rows x fin] = dz[rows x fout] W^T
    auto dX = [&](int w_off, int fin, int fout, const float *dz, long long rows, float *o) {
      return g_rb.sgemm_sb(s->blas, RB_OP_T, RB_OP_N, fin, (int)rows, fout, &one, theta + w_off, fout, d, dz, fout, rows * fout, &zero, o, fin,
                           rows * fin, E);
    };
    auto act_grad = [&](float *dh, const float *h, long long n) {
      k_gemm_act_grad<<<dim3(blocks(n), E), 256, 0, st>>>(dh, h, n, act);
    };
    if (!mfma) {
      if (dW(g.k_f3, g.b_f3, 84, g.K, f2, Rc * 84, out, Rc)) return fail(MILE_ERR_HIP, "rocblas sgemm (d fc3) failed");
      if (dX(g.k_f3, 84, g.K, out, Rc, df2)) return fail(MILE_ERR_HIP, "rocblas sgemm (d fc3 in) failed");
      act_grad(df2, f2, Rc * 84);
      if (dW(g.k_f2, g.b_f2, 120, 84, f1, Rc * 120, df2, Rc)) return fail(MILE_ERR_HIP, "rocblas sgemm (d fc2) failed");
      if (dX(g.k_f2, 120, 84, df2, Rc, df1)) return fail(MILE_ERR_HIP, "rocblas sgemm (d fc2 in) failed");
      act_grad(df1, f1, Rc * 120);
      if (dW(g.k_f1, g.b_f1, g.flat, 120, p2, Rc * g.flat, df1, Rc)) return fail(MILE_ERR_HIP, "rocblas sgemm (d fc1) failed");
      if (dX(g.k_f1, g.flat, 120, df1, Rc, dp2)) return fail(MILE_ERR_HIP, "rocblas sgemm (d fc1 in) failed");
    }
    if (!direct) k_unpool_actgrad<<<blocks(B * (long long)n_a2), 256, 0, st>>>(dp2, a2, dz2, B, g.h2, g.w2, 16, act);
    if (direct) {   // dZ = unpool(dP) * act'(A) is formed inside the kernels' tile loads: no dz2 / dz1 arrays
      const int acc = chunk != 0;
#define LAUNCH_DW2(GEO_) k_conv5_dw<16, GEO_><<<dim3(nwg, E), 256, lds_w2, st>>>(p1, Rc * (long long)n_p1, (long long)n_p1, g.wp1 * 6, 6, 1, 6, g.hp1, g.wp1, 0, dp2, a2, act, part2, (int)Rc, ipw)
#define LAUNCH_DX(HO_, WO_) k_conv5_dx<6, 16, 1, HO_, WO_><<<dim3(nwg, E), 256, lds_x2, st>>>(dp2, a2, act, theta, g.k_c2, d, dp1, (int)Rc, g.h2, g.w2, ipw)
#define LAUNCH_DW1(GEO_) k_conv5_dw<6, GEO_><<<dim3(nwg, E), 256, lds_w1, st>>>(Xc, 0, (long long)g.C * HW, g.W, 1, (long long)HW, g.C, g.H, g.W, 2, dp1, a1, act, part1, (int)Rc, ipw)
      if (mfma) k_conv5m_dw<CM_IN8, 16><<<dim3(nwg, E), 256, ldm_w2, st>>>(p1, Rc * (long long)n_p1, (long long)n_p1, g.wp1 * 6, 6, 1, 6, g.hp1, g.wp1, 0, dp2, a2, act, part2, (int)Rc, ipw, a16, ni2);
      else if (geo == 1) LAUNCH_DW2(2); else if (geo == 3) LAUNCH_DW2(4); else LAUNCH_DW2(0);
      k_conv_reduce<<<dim3(10, E), 256, 0, st>>>(part2, (int)nwg, 2400, 16, slab, dp, g.k_c2, g.b_c2, acc);
      if (pair) k_conv5m_dx2x<6, 16><<<dim3(nwg, E), 256, ldm_x2p, st>>>(dp2, a2, act, theta, g.k_c2, d, dp1, (int)Rc, g.h2, g.w2, ipw, a16, ni_x2);
      else if (mfma) k_conv5m_dx<6, 16><<<dim3(nwg, E), 256, ldm_x2, st>>>(dp2, a2, act, theta, g.k_c2, d, dp1, (int)Rc, g.h2, g.w2, ipw, a16);
      else if (geo == 1) LAUNCH_DX(12, 12); else if (geo == 3) LAUNCH_DX(10, 10); else LAUNCH_DX(0, 0);
      if (pair) k_conv5m_dw2x<6><<<dim3(nwg, E), 256, ldm_w1p, st>>>(Xc, 0, (long long)g.C * HW, g.W, 1, (long long)HW, g.C, g.H, g.W, 2, dp1, a1, act, part1, (int)Rc, ipw, a16);
      else if (mfma) k_conv5m_dw<CM_IN4, 6><<<dim3(nwg, E), 256, ldm_w1, st>>>(Xc, 0, (long long)g.C * HW, g.W, 1, (long long)HW, g.C, g.H, g.W, 2, dp1, a1, act, part1, (int)Rc, ipw, a16);
      else if (geo == 1) LAUNCH_DW1(1); else if (geo == 3) LAUNCH_DW1(3); else LAUNCH_DW1(0);
#undef LAUNCH_DW2
#undef LAUNCH_DX
#undef LAUNCH_DW1
      k_conv_reduce<<<dim3(2, E), 256, 0, st>>>(part1, (int)nwg, KT1 * 6, 6, slab, dp, g.k_c1, g.b_c1, acc);
    } else {
      if (dW(g.k_c2, g.b_c2, 150, 16, col2, M2 * 150, dz2, M2)) return fail(MILE_ERR_HIP, "rocblas sgemm (d conv2) failed");
      if (dX(g.k_c2, 150, 16, dz2, M2, col2)) return fail(MILE_ERR_HIP, "rocblas sgemm (d conv2 in) failed");   // d(col2) over col2
      k_col2im5<<<blocks(B * (long long)n_p1), 256, 0, st>>>(col2, dp1, B, g.h2, g.w2, 6);
      k_unpool_actgrad<<<blocks(B * (long long)n_a1), 256, 0, st>>>(dp1, a1, dz1, B, g.H, g.W, 6, act);
      if (dW(g.k_c1, g.b_c1, 25 * g.C, 6, col1, 0, dz1, M1)) return fail(MILE_ERR_HIP, "rocblas sgemm (d conv1) failed");
    }
  }
  HIP_TRY(hipGetLastError());
  return MILE_OK;
}

// Layer-wise path: per row chunk, forward GEMM + bias/activation per layer, head, then per layer
// dW (accumulated in place in the slab), bias column sums, dH GEMM + activation derivative.
// Row-major products through rocBLAS's column-major interface: C = A B  <=>  C^T = B^T A^T.
static int launch_grad_gemm(mile_sampler *s, const GradParams &gp, int E, hipStream_t st) {
  if (!rocblas_load()) return fail(MILE_ERR_HIP, "librocblas.so could not be loaded");
  const DevSpec &ds = s->ds;
  const int L = ds.n_layers, F = ds.in_features, d = ds.d, N = s->N;
  if (!s->blas) {
    if (g_rb.create(&s->blas) != 0) { s->blas = nullptr; return fail(MILE_ERR_HIP, "rocblas_create_handle failed"); }
  }
  if (g_rb.set_stream(s->blas, st) != 0) return fail(MILE_ERR_HIP, "rocblas_set_stream failed");
  size_t per_row = 2 * (size_t)ds.max_width;
  for (int l = 0; l < L; ++l) per_row += ds.widths[l];
  if (E != s->gemm_E || !s->gemm_ws) {   // (re)size the activation workspace: <= ~4 GiB, whole data set if it fits
    const size_t budget = (size_t)1 << 30;   // floats
    size_t R = budget / ((size_t)E * per_row);
    R = std::min<size_t>(R, (size_t)N);
    if (R < (size_t)N) R = std::max<size_t>(256, R / 256 * 256);
    R = std::min<size_t>(R, (size_t)N);
    if (const char *rv = getenv("MILE_GEMM_ROWS")) R = std::max<size_t>(1, std::min<size_t>((size_t)atoll(rv), (size_t)N));   // test hook
    const size_t need = (size_t)E * R * per_row;
    if (need > s->gemm_ws_floats) {
      if (s->gemm_ws) (void)hipFree(s->gemm_ws);
      s->gemm_ws = nullptr; s->gemm_ws_floats = 0;
      HIP_TRY(hipMalloc(&s->gemm_ws, need * 4));
      s->gemm_ws_floats = need;
    }
    s->gemm_R = (int)R; s->gemm_E = E;
  }
  const int R = s->gemm_R;
  if (s->gemm_ones_n < R) {   // ones[R]: bias gradients are dZ^T 1 (a skinny GEMM, memory bound like the sum it replaces)
    if (s->gemm_ones) (void)hipFree(s->gemm_ones);
    s->gemm_ones = nullptr; s->gemm_ones_n = 0;
    HIP_TRY(hipMalloc(&s->gemm_ones, (size_t)R * 4));
    k_fill<<<(R + 255) / 256, 256, 0, st>>>(s->gemm_ones, 1.0f, R);
    s->gemm_ones_n = R;
  }
  float *H[MILE_MAX_LAYERS], *tmp[2];
  {
    float *q = s->gemm_ws;
    for (int l = 0; l < L; ++l) { H[l] = q; q += (size_t)E * R * ds.widths[l]; }
    tmp[0] = q; q += (size_t)E * R * ds.max_width;
    tmp[1] = q;
  }
  const float one = 1.0f, zero = 0.0f;
  float *slab = gp.slabs;        // S = 1: [E][dp]
  const long long dp = gp.dp;
  auto gemm = [&](int ta, int tb, int m, int n, int k, const float *A, int lda, long long sa, const float *B, int ldb, long long sb,
                  const float *beta, float *C, int ldc, long long sc) -> int {
    return g_rb.sgemm_sb(s->blas, ta, tb, m, n, k, &one, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, E);
  };
  auto blocks = [](long long n) { return (unsigned)std::min<long long>((n + 255) / 256, 4096); };
  for (int r0 = 0, chunk = 0; r0 < N; r0 += R, ++chunk) {
    const int Rc = std::min(R, N - r0);
    // ---- forward
    for (int l = 0; l < L; ++l) {
      const int fin = l == 0 ? F : ds.widths[l - 1], fout = ds.widths[l];
      const float *in = l == 0 ? gp.X + (size_t)r0 * F : H[l - 1];
      const long long sin = l == 0 ? 0 : (long long)Rc * fin;
      if (gemm(RB_OP_N, RB_OP_N, fout, Rc, fin, gp.theta + ds.w_off[l], fout, d, in, fin, sin, &zero, H[l], fout, (long long)Rc * fout))
        return fail(MILE_ERR_HIP, "rocblas sgemm (forward) failed");
      const long long RW = (long long)Rc * fout;
      k_gemm_bias_act<<<dim3(blocks(RW), E), 256, 0, st>>>(H[l], gp.theta, ds.b_off[l], d, fout, RW, ds.activation, l + 1 < L);
    }
    // ---- head: log-likelihood and d(out), in place
    k_gemm_head<<<E, 256, 0, st>>>(H[L - 1], gp.y, r0, Rc, ds.widths[L - 1], ds.task, gp.llpart, chunk == 0);
    // ---- backward
    float *dz = H[L - 1];
    int pp = 0;
    const float *beta = chunk == 0 ? &zero : &one;
    for (int l = L - 1; l >= 0; --l) {
      const int fin = l == 0 ? F : ds.widths[l - 1], fout = ds.widths[l];
      const float *in = l == 0 ? gp.X + (size_t)r0 * F : H[l - 1];
      const long long sin = l == 0 ? 0 : (long long)Rc * fin;
      // dW[in][out] (+)= in^T dz, written straight into the slab at the kernel's offset
      if (gemm(RB_OP_N, RB_OP_T, fout, fin, Rc, dz, fout, (long long)Rc * fout, in, fin, sin, beta, slab + ds.w_off[l], fout, dp))
        return fail(MILE_ERR_HIP, "rocblas sgemm (dW) failed");
      if (gemm(RB_OP_N, RB_OP_N, fout, 1, Rc, dz, fout, (long long)Rc * fout, s->gemm_ones, Rc, 0, beta, slab + ds.b_off[l], fout, dp))
        return fail(MILE_ERR_HIP, "rocblas sgemm (bias gradient) failed");
      if (l > 0) {
        if (gemm(RB_OP_T, RB_OP_N, fin, Rc, fout, gp.theta + ds.w_off[l], fout, d, dz, fout, (long long)Rc * fout, &zero, tmp[pp], fin,
                 (long long)Rc * fin))
          return fail(MILE_ERR_HIP, "rocblas sgemm (dH) failed");
        const long long RW = (long long)Rc * fin;
        k_gemm_act_grad<<<dim3(blocks(RW), E), 256, 0, st>>>(tmp[pp], H[l - 1], RW, ds.activation);
        dz = tmp[pp];
        pp ^= 1;
      }
    }
  }
  HIP_TRY(hipGetLastError());
  return MILE_OK;
}

static int launch_grad(mile_sampler *s, const float *theta, int E, hipStream_t st, const UpdParams *fused_update = nullptr);

// ------------------------------------------------------------------------------------
// Layer-wise MFMA path (MILE_GRAD_MFMA_WIDE_BF16X3 / _BF16): mile_mm3.h.  Per row chunk: forward GEMM per layer with bias +
// activation in its epilogue, head, then per layer dW (K = the chunk's rows, accumulated in place in the slab), bias
// column sums, and the dH GEMM with the activation derivative in its epilogue.
// ------------------------------------------------------------------------------------
#ifndef MILE_MM_KC
#define MILE_MM_KC 32     // K chunk of k_mm3: 32 = two workgroups per CU (mile_mm3.h); 64 = one, measured slower
#endif
template <int ALAY, int BSRC, int EPI, int TERMS, int ACT, bool ACCUM, bool COLSUM, bool FULL = false>
static hipError_t launch_mm3_k(const MMParams &p, int batch, hipStream_t st) {
  constexpr int KC = MILE_MM_KC;
  using LY = MMLayout<ALAY, BSRC, TERMS, KC>;
  static const bool dbg = getenv("MILE_DEBUG") && (atoi(getenv("MILE_DEBUG")) & 64);
  if constexpr (!FULL) {   // whole tiles everywhere: the predicate-free instantiation (timing stamps need the general one)
    static const bool no_full = getenv("MILE_MM_NO_FULL") != nullptr;
    if (!dbg && !no_full && p.M % 128 == 0 && p.N % 128 == 0)
      return launch_mm3_k<ALAY, BSRC, EPI, TERMS, ACT, ACCUM, COLSUM, true>(p, batch, st);
  }
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)k_mm3<ALAY, BSRC, EPI, TERMS, KC, ACT, ACCUM, COLSUM, FULL>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LY::BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  MMParams q = p;
  const int mtiles = (p.M + 127) / 128;
  const dim3 grid((p.N + 127) / 128, mtiles, batch);
  q.c_vec = (p.ldc % 4 == 0) && (p.sC % 4 == 0) && (((uintptr_t)p.C & 15) == 0);
  static const bool no_remap = getenv("MILE_MM_NO_XCD") != nullptr;
  q.xcd_remap = 0;
  if (!no_remap) {
    if (COLSUM && batch % 8 == 0) q.xcd_remap = 2;                     // dW: the few tiles of a particle share both operands
    else if (!COLSUM && grid.x > 1 && grid.y >= 8) q.xcd_remap = 1;
  }
  static unsigned long long *dbuf = nullptr;
  if (dbg) {   // dev: mean time per phase of wave 0 (100 MHz ticks -> us), per launch
    if (!dbuf && hipMalloc(&dbuf, 64) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipMemsetAsync(dbuf, 0, 64, st);
    q.dbg = dbuf;
  }
  k_mm3<ALAY, BSRC, EPI, TERMS, KC, ACT, ACCUM, COLSUM, FULL><<<grid, 256, LY::BYTES, st>>>(q);
  if (dbg) {
    unsigned long long h[8];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h, dbuf, 64, hipMemcpyDeviceToHost);
    const double nc = (double)std::max<unsigned long long>(h[5], 1), nt = (double)std::max<unsigned long long>(h[6], 1);
    fprintf(stderr, "k_mm3<A%d,B%d,E%d> M=%d N=%d K=%d grid %ux%ux%u: per chunk us: wait loads+barrier %.2f  split+store %.2f  barrier %.2f  loads issue+frags+MFMA %.2f | per tile: epilogue %.2f  (chunks/tile %.1f)\n",
            ALAY, BSRC, EPI, p.M, p.N, p.K, grid.x, grid.y, grid.z, h[0] / nc * 0.01, h[1] / nc * 0.01, h[2] / nc * 0.01, h[3] / nc * 0.01, h[4] / nt * 0.01, nc / nt);
  }
  return hipGetLastError();
}
// the three products of the layer-wise path, dispatched on the launch constants the kernel takes as template parameters
template <int TERMS>
static hipError_t launch_mm3_fwd(const MMParams &p, int batch, hipStream_t st) {
  const int act = p.apply_act ? p.act : -1;
#define MILE_MM_FWD(A_) launch_mm3_k<MM_A_MK, MM_B_T3_KN, MM_EPI_BIAS_ACT, TERMS, A_, false, false>(p, batch, st)
  return act == MILE_ACT_RELU ? MILE_MM_FWD(MILE_ACT_RELU) : act == MILE_ACT_TANH ? MILE_MM_FWD(MILE_ACT_TANH)
       : act == MILE_ACT_SIGMOID ? MILE_MM_FWD(MILE_ACT_SIGMOID) : MILE_MM_FWD(-1);
#undef MILE_MM_FWD
}
template <int TERMS>
static hipError_t launch_mm3_dh(const MMParams &p, int batch, hipStream_t st) {
#define MILE_MM_DH(A_) launch_mm3_k<MM_A_MK, MM_B_T3_NK, MM_EPI_ACT_GRAD, TERMS, A_, false, false>(p, batch, st)
  return p.act == MILE_ACT_RELU ? MILE_MM_DH(MILE_ACT_RELU) : p.act == MILE_ACT_TANH ? MILE_MM_DH(MILE_ACT_TANH) : MILE_MM_DH(MILE_ACT_SIGMOID);
#undef MILE_MM_DH
}
template <int TERMS>
static hipError_t launch_mm3_dw(const MMParams &p, int batch, hipStream_t st) {
  return p.accumulate ? launch_mm3_k<MM_A_KM, MM_B_F32_KN, MM_EPI_STORE, TERMS, -1, true, true>(p, batch, st)
                      : launch_mm3_k<MM_A_KM, MM_B_F32_KN, MM_EPI_STORE, TERMS, -1, false, true>(p, batch, st);
}

// ---- Dense half of LeNet (flat -> 120 -> 84 -> K) on k_mm3, TERMS = 3: the weights as zero-padded bf16 term planes in
// s->wide_wt [E][layer][term][in][outp], activations [E][Rc][ld] with ld = 120 / 88 / up8(K)
static void lenet_dense_dims(const LeNetGeom &g, int (&fin)[3], int (&fout)[3], int (&ld)[3], int (&woff)[3], int (&boff)[3], size_t (&pl)[3],
                             size_t &elems) {
  fin[0] = g.flat; fin[1] = 120; fin[2] = 84;
  fout[0] = 120; fout[1] = 84; fout[2] = g.K;
  woff[0] = g.k_f1; woff[1] = g.k_f2; woff[2] = g.k_f3;
  boff[0] = g.b_f1; boff[1] = g.b_f2; boff[2] = g.b_f3;
  elems = 0;
  for (int l = 0; l < 3; ++l) { ld[l] = (fout[l] + 7) / 8 * 8; pl[l] = elems; elems += (size_t)3 * fin[l] * ld[l]; }
}
static int lenet_dense_prep(mile_sampler *s, const float *theta, int E, hipStream_t st) {
  int fin[3], fout[3], ld[3], woff[3], boff[3]; size_t pl[3], elems;
  lenet_dense_dims(s->lg, fin, fout, ld, woff, boff, pl, elems);
  const size_t bytes = (size_t)E * elems * 2;
  if (bytes > s->wide_wt_bytes) {
    if (s->wide_wt) (void)hipFree(s->wide_wt);
    s->wide_wt = nullptr; s->wide_wt_bytes = 0;
    HIP_TRY(hipMalloc(&s->wide_wt, bytes));
    s->wide_wt_bytes = bytes;
  }
  for (int l = 0; l < 3; ++l) {
    const long long plane = (long long)fin[l] * ld[l];
    k_wide_prep_weights<3><<<dim3((unsigned)std::min<long long>((plane + 255) / 256, 1024), E), 256, 0, st>>>(
        theta, s->lg.d, woff[l], fin[l], fout[l], ld[l], (bf16 *)s->wide_wt + pl[l], (long long)elems);
  }
  HIP_TRY(hipGetLastError());
  return MILE_OK;
}
static int lenet_dense_chunk(mile_sampler *s, const float *theta, int E, int Rc, int r0, int chunk, const LeNetDenseBufs &b, const void *y,
                             float *slab, long long dp, float *llacc, float *out_ll, long long Ntot, long long s0, hipStream_t st) {
  const LeNetGeom &g = s->lg;
  int fin[3], fout[3], ld[3], woff[3], boff[3]; size_t pl[3], elems;
  lenet_dense_dims(g, fin, fout, ld, woff, boff, pl, elems);
  const bf16 *Wt = (const bf16 *)s->wide_wt;
  const int act = s->ds.activation, task = s->ds.task, d = g.d;
  const float *in[3] = {b.p2, b.f1, b.f2};
  const int ldin[3] = {g.flat, ld[0], ld[1]};
  float *H[3] = {b.f1, b.f2, b.out};
  for (int l = 0; l < 3; ++l) {   // forward: H_l = act(in W_l + b_l), no activation on the head
    MMParams p{};
    p.A = in[l]; p.sA = (long long)Rc * ldin[l]; p.lda = ldin[l]; p.K = fin[l];
    p.B = Wt + pl[l]; p.sB = (long long)elems; p.ldb = ld[l]; p.tB = (long long)fin[l] * ld[l];
    p.C = H[l]; p.sC = (long long)Rc * ld[l]; p.ldc = ld[l];
    p.M = Rc; p.N = fout[l];
    p.bias = theta + boff[l]; p.sBias = d;
    p.act = act; p.apply_act = l < 2;
    HIP_TRY(launch_mm3_fwd<3>(p, E, st));
  }
  if (!slab) {   // evaluation: per-row log-likelihood
    k_wide_rowll<<<dim3((unsigned)((Rc + 255) / 256), E), 256, 0, st>>>(b.out, (long long)Rc * ld[2], ld[2], y, r0, Rc, g.K, task, out_ll, Ntot, s0);
    HIP_TRY(hipGetLastError());
    return MILE_OK;
  }
  k_wide_head<<<E, 256, 0, st>>>(b.out, (long long)Rc * ld[2], ld[2], y, r0, Rc, g.K, task, llacc, chunk == 0);
  float *dz[3] = {b.df1, b.df2, b.out};      // dZ of layer l (the head's d(out) is in place)
  for (int l = 2; l >= 0; --l) {
    {  // dW_l (+)= in^T dZ_l into the slab, bias gradient from the same B tiles
      MMParams p{};
      p.A = in[l]; p.sA = (long long)Rc * ldin[l]; p.lda = ldin[l];
      p.B = dz[l]; p.sB = (long long)Rc * ld[l]; p.ldb = ld[l];
      p.C = slab + woff[l]; p.sC = dp; p.ldc = fout[l];
      p.M = fin[l]; p.N = fout[l]; p.K = Rc;
      p.accumulate = chunk != 0;
      p.colsum = slab + boff[l]; p.sColsum = dp;
      HIP_TRY(launch_mm3_dw<3>(p, E, st));
    }
    MMParams p{};   // d(in) = dZ_l W_l^T, times act'(in) for the Dense inputs; the pooled conv output takes it as is
    p.A = dz[l]; p.sA = (long long)Rc * ld[l]; p.lda = ld[l]; p.K = fout[l];
    p.B = Wt + pl[l]; p.sB = (long long)elems; p.ldb = ld[l]; p.tB = (long long)fin[l] * ld[l];
    p.M = Rc; p.N = fin[l];
    if (l > 0) {
      p.C = dz[l - 1]; p.sC = (long long)Rc * ldin[l]; p.ldc = ldin[l];
      p.Hprev = in[l]; p.sH = (long long)Rc * ldin[l]; p.ldh = ldin[l];
      p.act = act;
      HIP_TRY(launch_mm3_dh<3>(p, E, st));
    } else {
      p.C = b.dp2; p.sC = (long long)Rc * g.flat; p.ldc = g.flat;
      HIP_TRY((launch_mm3_k<MM_A_MK, MM_B_T3_NK, MM_EPI_STORE, 3, -1, false, false>(p, E, st)));
    }
  }
  HIP_TRY(hipGetLastError());
  return MILE_OK;
}

template <int TERMS>
static int launch_grad_wide(mile_sampler *s, const GradParams &gp, int E, hipStream_t st) {
  const DevSpec &ds = s->ds;
  const int L = ds.n_layers, d = ds.d, N = gp.N, Fp = s->Fp;   // gp.N < s->N under a row window (workspace sized for the whole set)
  auto up8 = [](int v) { return (v + 7) / 8 * 8; };
  int wp[MILE_MAX_LAYERS], fin[MILE_MAX_LAYERS], finp[MILE_MAX_LAYERS];
  size_t per_row = 0, wt_elems = 0, wt_off[MILE_MAX_LAYERS];
  int maxwp = 0;
  for (int l = 0; l < L; ++l) {
    wp[l] = up8(ds.widths[l]);
    fin[l] = l == 0 ? ds.in_features : ds.widths[l - 1];
    finp[l] = l == 0 ? Fp : wp[l - 1];
    per_row += wp[l];
    maxwp = std::max(maxwp, wp[l]);
    wt_off[l] = wt_elems;
    wt_elems += (size_t)TERMS * fin[l] * wp[l];
  }
  per_row += 2 * (size_t)maxwp;
  if (E != s->wide_E || !s->wide_ws) {   // activation workspace: MILE_WIDE_WS_GB (default 16 GiB of the 288), whole data set if it fits
    double gb = 16.0;
    if (const char *ev = getenv("MILE_WIDE_WS_GB")) gb = std::max(0.001, atof(ev));
    const size_t budget = (size_t)(gb * (double)(1ull << 30) / 4.0);
    const size_t Nall = (size_t)s->N;   // sized for the whole data set, also when the first call comes under a row window
    size_t R = budget / ((size_t)E * per_row);
    R = std::min<size_t>(std::max<size_t>(R, 128), Nall);
    if (R < Nall) R = std::max<size_t>(128, R / 128 * 128);
    if (const char *rv = getenv("MILE_GEMM_ROWS")) R = std::max<size_t>(1, std::min<size_t>((size_t)atoll(rv), Nall));   // test hook
    const size_t need = (size_t)E * R * per_row;
    if (need > s->wide_ws_floats) {
      if (s->wide_ws) (void)hipFree(s->wide_ws);
      s->wide_ws = nullptr; s->wide_ws_floats = 0;
      HIP_TRY(hipMalloc(&s->wide_ws, need * 4));
      s->wide_ws_floats = need;
    }
    HIP_TRY(hipMemsetAsync(s->wide_ws, 0, s->wide_ws_floats * 4, st));   // padding columns are read as operands: zero, once per layout
    s->wide_R = (int)R; s->wide_E = E;
  }
  const size_t wt_bytes = (size_t)E * wt_elems * 2;
  if (wt_bytes > s->wide_wt_bytes) {
    if (s->wide_wt) (void)hipFree(s->wide_wt);
    s->wide_wt = nullptr; s->wide_wt_bytes = 0;
    HIP_TRY(hipMalloc(&s->wide_wt, wt_bytes));
    s->wide_wt_bytes = wt_bytes;
  }
  const int R = s->wide_R;
  float *H[MILE_MAX_LAYERS], *tmp[2];
  {
    float *q = s->wide_ws;
    for (int l = 0; l < L; ++l) { H[l] = q; q += (size_t)E * R * wp[l]; }
    tmp[0] = q; q += (size_t)E * R * maxwp;
    tmp[1] = q;
  }
  // The two dZ buffers are shared by layers of different padded widths: a layer's padding columns (real width .. wp) are not
  // written by the dH epilogue and would otherwise show what an earlier use with another leading dimension -- possibly an
  // earlier CALL with a diverged, non-finite theta -- left there; k_mm3 reads operands in 4- / 8-element granules, and
  // NaN x 0 is NaN.  Re-zeroed per call whenever the hidden layers pad to different widths (nets of equal hidden
  // widths -- B4 -- keep one layout per buffer, whose padding stays zero from the workspace's own memset).
  {
    bool mixed = false;                                    // dZ of layers 0 .. L-2 lives in tmp with leading dimension wp[l]
    for (int l = 1; l + 1 < L; ++l) mixed = mixed || wp[l] != wp[0];
    if (mixed) HIP_TRY(hipMemsetAsync(tmp[0], 0, (size_t)E * R * maxwp * 2 * 4, st));
  }
  bf16 *Wt = (bf16 *)s->wide_wt;
  // ---- this gradient's weights as zero-padded bf16 term planes [E][layer][term][in][outp]
  for (int l = 0; l < L; ++l) {
    const long long plane = (long long)fin[l] * wp[l];
    k_wide_prep_weights<TERMS><<<dim3((unsigned)std::min<long long>((plane + 255) / 256, 1024), E), 256, 0, st>>>(
        gp.theta, d, ds.w_off[l], fin[l], ds.widths[l], wp[l], Wt + wt_off[l], (long long)wt_elems);
  }
  float *slab = gp.slabs;        // S = 1: [E][dp]
  const long long dp = gp.dp;
  // The last layer (K <= 8 outputs on a hidden width <= 256) as one pass over the last hidden activations -- forward, head, dH and
  // dW of that layer in k_wide_headblock -- instead of four launches on 128-wide MFMA tiles (fp32-faithful form only: the
  // bf16-operand recipe rounds this product's operands)
  const bool headblock = TERMS == 3 && L >= 2 && ds.widths[L - 2] <= 256 && ds.widths[L - 1] <= WH_KMAX &&
                         getenv("MILE_WIDE_NO_HEADBLOCK") == nullptr;
  constexpr int HB_ROWS = 512;                      // rows per workgroup of k_wide_headblock
  if (headblock) {
    const size_t need = (size_t)E * ((R + HB_ROWS - 1) / HB_ROWS) * ((size_t)ds.widths[L - 2] * ds.widths[L - 1] + ds.widths[L - 1] + 1);
    if (need > s->wide_hb_floats) {
      if (s->wide_hb) (void)hipFree(s->wide_hb);
      s->wide_hb = nullptr; s->wide_hb_floats = 0;
      HIP_TRY(hipMalloc(&s->wide_hb, need * 4));
      s->wide_hb_floats = need;
    }
  }
  for (int r0 = 0, chunk = 0; r0 < N; r0 += R, ++chunk) {
    const int Rc = std::min(R, N - r0);
    // ---- forward
    for (int l = 0; l < L - (headblock ? 1 : 0); ++l) {
      MMParams p{};
      if (l == 0) { p.A = gp.Xp + (size_t)r0 * Fp; p.sA = 0; p.lda = Fp; }   // X zero-padded to Fp columns
      else { p.A = H[l - 1]; p.sA = (long long)R * wp[l - 1]; p.lda = wp[l - 1]; }
      p.K = fin[l];
      p.B = Wt + wt_off[l]; p.sB = (long long)wt_elems; p.ldb = wp[l]; p.tB = (long long)fin[l] * wp[l];
      p.C = H[l]; p.sC = (long long)R * wp[l]; p.ldc = wp[l];
      p.M = Rc; p.N = ds.widths[l];
      p.bias = gp.theta + ds.b_off[l]; p.sBias = d;
      p.act = ds.activation; p.apply_act = l + 1 < L;
      HIP_TRY(launch_mm3_fwd<TERMS>(p, E, st));
    }
    // ---- head: log-likelihood and d(out), in place
    float *dz = H[L - 1];
    int pp = 0, ltop = L - 1;
    if (headblock) {
      const int Wl = ds.widths[L - 2], K = ds.widths[L - 1], nblk = (Rc + HB_ROWS - 1) / HB_ROWS;
#define MILE_HB(K_, WF_) k_wide_headblock<K_, WF_><<<dim3(nblk, E), 256, 0, st>>>(H[L - 2], (long long)R * wp[L - 2], wp[L - 2], Wl, gp.theta, d, \
      ds.w_off[L - 1], ds.b_off[L - 1], gp.y, r0, Rc, ds.task, ds.activation, tmp[0], (long long)R * wp[L - 2], wp[L - 2], s->wide_hb, HB_ROWS)
#define MILE_HB_K(K_) case K_: if (Wl == 256) MILE_HB(K_, true); else MILE_HB(K_, false); break;
      switch (K) { MILE_HB_K(1) MILE_HB_K(2) MILE_HB_K(3) MILE_HB_K(4) MILE_HB_K(5) MILE_HB_K(6) MILE_HB_K(7) MILE_HB_K(8) }
#undef MILE_HB_K
#undef MILE_HB
      k_wide_headblock_reduce<<<dim3(8, E), 256, 0, st>>>(s->wide_hb, nblk, Wl * K, K, slab, dp, ds.w_off[L - 1], ds.b_off[L - 1], gp.llpart,
                                                          chunk != 0);
      dz = tmp[0]; pp = 1; ltop = L - 2;
    } else {
      k_wide_head<<<E, 256, 0, st>>>(H[L - 1], (long long)R * wp[L - 1], wp[L - 1], gp.y, r0, Rc, ds.widths[L - 1], ds.task, gp.llpart, chunk == 0);
    }
    // ---- backward
    for (int l = ltop; l >= 0; --l) {
      {  // dW_l[in][out] (+)= in^T dz, straight into the slab at the kernel's offset
        MMParams p{};
        if (l == 0) { p.A = gp.Xp + (size_t)r0 * Fp; p.sA = 0; p.lda = Fp; }
        else { p.A = H[l - 1]; p.sA = (long long)R * wp[l - 1]; p.lda = wp[l - 1]; }
        p.B = dz; p.sB = (long long)R * wp[l]; p.ldb = wp[l];
        p.C = slab + ds.w_off[l]; p.sC = dp; p.ldc = ds.widths[l];
        p.M = fin[l]; p.N = ds.widths[l]; p.K = Rc;
        p.accumulate = chunk != 0;
        p.colsum = slab + ds.b_off[l]; p.sColsum = dp;      // bias gradient dz^T 1 from the B tiles of M tile 0
        HIP_TRY(launch_mm3_dw<TERMS>(p, E, st));
      }
      if (l > 0) {   // dZ_{l-1} = (dz W_l^T) * act'(H_{l-1})
        MMParams p{};
        p.A = dz; p.sA = (long long)R * wp[l]; p.lda = wp[l]; p.K = ds.widths[l];
        p.B = Wt + wt_off[l]; p.sB = (long long)wt_elems; p.ldb = wp[l]; p.tB = (long long)fin[l] * wp[l];
        p.C = tmp[pp]; p.sC = (long long)R * wp[l - 1]; p.ldc = wp[l - 1];
        p.M = Rc; p.N = fin[l];
        p.Hprev = H[l - 1]; p.sH = (long long)R * wp[l - 1]; p.ldh = wp[l - 1];
        p.act = ds.activation;
        HIP_TRY(launch_mm3_dh<TERMS>(p, E, st));
        dz = tmp[pp];
        pp ^= 1;
      }
    }
  }
  HIP_TRY(hipGetLastError());
  return MILE_OK;
}

static int launch_grad(mile_sampler *s, const float *theta, int E, hipStream_t st, const UpdParams *fused_update) {
  if (!s->X) return fail(MILE_ERR_STATE, "no data: call mile_set_data first");
  const int kernel = resolved_kernel(s);
  const int S = choose_S(s, E, kernel);
  if (E > s->E_cap || (size_t)E * S > s->ES_cap) return fail(MILE_ERR_STATE, "workspace too small: call mile_reserve(E) first");
  GradParams gp;
  gp.spec = s->ds;
  gp.theta = theta;
  gp.X = s->X; gp.Xp = s->Xp; gp.y = s->y; gp.Xb = s->Xb; gp.Xt = s->Xt;
  gp.slabs = s->slabs; gp.llpart = s->llpart;
  gp.N = s->N; gp.Npad = s->Npad; gp.Npb = s->Npb; gp.Fp = s->Fp; gp.S = S; gp.R = generic_R(s->ds); gp.dp = (s->ds.d + 3) / 4 * 4;
  if (s->win_count) {   // minibatch: the same kernels on a shifted view of the rows
    const bool chunked = kernel == MILE_GRAD_MFMA_WIDE_BF16X3 || kernel == MILE_GRAD_MFMA_WIDE_BF16 || kernel == MILE_GRAD_LENET_F32 ||
                         kernel == MILE_GRAD_LENET_BF16;   // these walk the rows in chunks anyway: a window is a shorter walk
    if (kernel != MILE_GRAD_GENERIC && kernel != MILE_GRAD_MFMA_NARROW_F32 && !is_w64(kernel) && !chunked)
      return fail(MILE_ERR_STATE, "a row window needs the generic, the MFMA_NARROW, an MFMA_W64, an MFMA_WIDE or a LENET grad kernel");
    if (fused_update) return fail(MILE_ERR_STATE, "row windows are for mile_logpost_grad only");
    const int F = s->spec.in_features;
    gp.X = s->X + (size_t)s->win_begin * F;
    gp.Xp = s->Xp + (size_t)s->win_begin * s->Fp;
    gp.y = (const char *)s->y + (size_t)s->win_begin * 4;
    gp.N = s->win_count;
    gp.Npad = (s->win_count + 31) / 32 * 32;
  }
  { const char *dv = getenv("MILE_DEBUG"); gp.dbg = dv ? atoi(dv) : 0; }
  gp.dbg_buf = nullptr;
  if ((gp.dbg & 16) && !s->dbg_buf) HIP_TRY(hipMalloc(&s->dbg_buf, 64));
  if ((gp.dbg & 32) && !s->dbg_buf) HIP_TRY(hipMalloc(&s->dbg_buf, s->ES_cap * 128));
  gp.dbg_buf = s->dbg_buf;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (s->timing) {
    if (s->ev_used + 2 > s->ev.size()) {
      for (int k = 0; k < 2; ++k) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreate(&ev));
        s->ev.push_back(ev);
      }
    }
    e0 = s->ev[s->ev_used]; e1 = s->ev[s->ev_used + 1];
    s->ev_used += 2;
    HIP_TRY(hipEventRecord(e0, st));
  }
  W64Fuse fz{};
  if (fused_update) {
    if (kernel != MILE_GRAD_MFMA_W64_BF16X3) return fail(MILE_ERR_STATE, "fused update needs the mfma_w64_bf16x3 grad kernel");
    fz.upd = *fused_update; fz.arrive = s->arrive; fz.enabled = 1; fz.kind = upd_kind(*fused_update);
  }
  if (kernel == MILE_GRAD_MFMA_W64) {
    const int nh = s->spec.n_layers - 1, fq = s->Fp / 8;
    hipError_t e = hipErrorInvalidValue;
    if (nh == 1 && fq == 1) e = launch_w64<1, 1>(gp, fz, E, st);
    else if (nh == 2 && fq == 1) e = launch_w64<2, 1>(gp, fz, E, st);
    else if (nh == 3 && fq == 1) e = launch_w64<3, 1>(gp, fz, E, st);
    else if (nh == 1 && fq == 2) e = launch_w64<1, 2>(gp, fz, E, st);
    else if (nh == 2 && fq == 2) e = launch_w64<2, 2>(gp, fz, E, st);
    else if (nh == 3 && fq == 2) e = launch_w64<3, 2>(gp, fz, E, st);
    HIP_TRY(e);
  } else if (kernel == MILE_GRAD_MFMA_W64_BF16X3) {
    const int nh = s->spec.n_layers - 1, fq = s->Fp / 8;
    hipError_t e = hipErrorInvalidValue;
    if (nh == 2 && fq == 1) e = launch_w64<2, 1, true>(gp, fz, E, st);
    else if (nh == 3 && fq == 1) e = launch_w64<3, 1, true>(gp, fz, E, st);
    else if (fq == 2 && !fz.enabled) e = mile_launch_w64_split_fq2(nh, gp, E, st);   // mile_w64_fq2.hip
    HIP_TRY(e);
  } else if (kernel == MILE_GRAD_MFMA_NARROW_F32) {
    HIP_TRY(launch_narrow(s, gp, E, st));
  } else if (kernel == MILE_GRAD_LENET_F32 || kernel == MILE_GRAD_LENET_BF16) {
    const int rc = run_lenet(s, theta, E, gp.X, gp.y, gp.N, gp.slabs, gp.dp, gp.llpart, nullptr, 0, st, kernel == MILE_GRAD_LENET_BF16);
    if (rc) return rc;
  } else if (kernel == MILE_GRAD_GEMM_F32) {
    const int rc = launch_grad_gemm(s, gp, E, st);
    if (rc) return rc;
  } else if (kernel == MILE_GRAD_MFMA_WIDE_BF16X3 || kernel == MILE_GRAD_MFMA_WIDE_BF16) {
    const int rc = kernel == MILE_GRAD_MFMA_WIDE_BF16X3 ? launch_grad_wide<3>(s, gp, E, st) : launch_grad_wide<1>(s, gp, E, st);
    if (rc) return rc;
  } else if (kernel == MILE_GRAD_MFMA_W128_BF16) {
    if (!s->Xb) return fail(MILE_ERR_STATE, "bf16 data copies missing: call mile_set_data");
    const int nh = s->spec.n_layers - 1;
    hipError_t e = hipErrorInvalidValue;
    if (nh == 1) e = launch_w128b<1>(gp, E, st);
    else if (nh == 2) e = launch_w128b<2>(gp, E, st);
    else if (nh == 3) e = launch_w128b<3>(gp, E, st);
    HIP_TRY(e);
  } else {
    const size_t lds = ((size_t)gp.R * (s->ds.act_stride + 2 * s->ds.max_width) + 16) * 4;
    k_grad_generic<<<dim3(S, E), 256, lds, st>>>(gp);
    HIP_TRY(hipGetLastError());
  }
  if (s->timing) HIP_TRY(hipEventRecord(e1, st));
  if ((gp.dbg & 32) && is_w64(kernel)) {   // dev: phase timestamps of the w64 kernels (100 MHz wall clock -> us)
    std::vector<long long> hb((size_t)E * S * 16);
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(hb.data(), s->dbg_buf, hb.size() * 8, hipMemcpyDeviceToHost));
    long long t00 = hb[0], tend = 0;
    for (int w = 0; w < E * S; ++w) { t00 = std::min(t00, hb[w * 16]); tend = std::max({tend, hb[w * 16 + 2], hb[w * 16 + 4]}); }
    double a[4] = {0, 0, 0, 0}, mx[4] = {0, 0, 0, 0}, ep[4] = {0, 0, 0, 0}, sub[4] = {0, 0, 0, 0}; int nl = 0;
    for (int w = 0; w < E * S; ++w) {
      const long long *q = &hb[w * 16];
      const double v0 = (q[1] - q[0]) * 0.01, v1 = (q[2] - q[1]) * 0.01;
      a[0] += v0; a[1] += v1; mx[0] = std::max(mx[0], v0); mx[1] = std::max(mx[1], v1);
      if (q[4]) { const double v2 = (q[3] - q[2]) * 0.01, v3 = (q[4] - q[3]) * 0.01; a[2] += v2; a[3] += v3; mx[2] = std::max(mx[2], v2); mx[3] = std::max(mx[3], v3); ++nl;
        ep[0] += (q[5] - q[3]) * 0.01; ep[1] += (q[6] - q[5]) * 0.01; ep[2] += (q[7] - q[6]) * 0.01; ep[3] += (q[4] - q[7]) * 0.01;
        sub[0] += (q[8] - q[5]) * 0.01; sub[1] += (q[9] - q[8]) * 0.01; sub[2] += (q[10] - q[9]) * 0.01; sub[3] += (q[6] - q[10]) * 0.01; }
    }
    fprintf(stderr, "w64 phases (us, mean/max over %d WGs): main %.1f/%.1f reduce+store %.1f/%.1f | last arrivers (%d): drain+ticket %.1f/%.1f epilogue %.1f/%.1f | first start -> last end %.1f\n",
            E * S, a[0] / (E * S), mx[0], a[1] / (E * S), mx[1], nl, nl ? a[2] / nl : 0.0, mx[2], nl ? a[3] / nl : 0.0, mx[3], (tend - t00) * 0.01);
    if (nl) fprintf(stderr, "    epilogue: loads %.1f  noise+sums+reduce %.1f  chain %.1f  pass2+stores %.1f   [sums: quads %.1f tail %.1f wave_sum %.1f lds+barrier %.1f]\n", ep[0] / nl, ep[1] / nl, ep[2] / nl, ep[3] / nl, sub[0] / nl, sub[1] / nl, sub[2] / nl, sub[3] / nl);
  }
  return MILE_OK;
}

// gradient at `theta`, then the update `u` that consumes it: one launch when the update can be the grad kernel's epilogue
static int grad_then_update(mile_sampler *s, const float *theta, int E, const UpdParams &u, bool fused, hipStream_t st) {
  if (fused) return launch_grad(s, theta, E, st, &u);
  const int rc = launch_grad(s, theta, E, st);
  if (rc) return rc;
  launch_update(u, E, st);
  return MILE_OK;
}

template <int NH, int FQ>
static hipError_t launch_fwd_w64(const PredParams &pp, int S, hipStream_t st) {
  using LY = W64Layout<NH, FQ>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)k_fwd_w64<NH, FQ>, hipFuncAttributeMaxDynamicSharedMemorySize, LY::BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  k_fwd_w64<NH, FQ><<<dim3(pp.SB, S), 256, LY::BYTES, st>>>(pp);
  return hipGetLastError();
}

// Evaluation forward for wide nets: the same strided-batched SGEMMs, samples as the batch.
static int launch_fwd_gemm(mile_sampler *s, const float *theta, int S, const float *X, const void *y, int N, float *out, hipStream_t st) {
  if (!rocblas_load()) return fail(MILE_ERR_HIP, "librocblas.so could not be loaded");
  const DevSpec &ds = s->ds;
  const int L = ds.n_layers, F = ds.in_features, d = ds.d;
  if (!s->blas && g_rb.create(&s->blas) != 0) { s->blas = nullptr; return fail(MILE_ERR_HIP, "rocblas_create_handle failed"); }
  if (g_rb.set_stream(s->blas, st) != 0) return fail(MILE_ERR_HIP, "rocblas_set_stream failed");
  // two ping-pong activation buffers of [Sc][Rc][max_width]; <= 1 GiB of floats
  const size_t budget = (size_t)1 << 28;
  const int Sc = std::min(S, 1024);
  size_t Rr = budget / (2 * (size_t)Sc * ds.max_width);
  Rr = std::max<size_t>(1, std::min<size_t>(Rr, (size_t)N));
  if (const char *rv = getenv("MILE_GEMM_ROWS")) Rr = std::max<size_t>(1, std::min<size_t>((size_t)atoll(rv), (size_t)N));
  const int R = (int)Rr;
  const size_t need = 2 * (size_t)Sc * R * ds.max_width;
  if (need > s->gemm_ws_floats) {
    if (s->gemm_ws) (void)hipFree(s->gemm_ws);
    s->gemm_ws = nullptr; s->gemm_ws_floats = 0; s->gemm_E = 0;
    HIP_TRY(hipMalloc(&s->gemm_ws, need * 4));
    s->gemm_ws_floats = need;
  }
  s->gemm_E = 0;   // the gradient path re-derives its layout on its next call
  float *buf[2] = {s->gemm_ws, s->gemm_ws + (size_t)Sc * R * ds.max_width};
  const float one = 1.0f, zero = 0.0f;
  for (int s0 = 0; s0 < S; s0 += Sc) {
    const int Sn = std::min(Sc, S - s0);
    const float *th = theta + (size_t)s0 * d;
    for (int r0 = 0; r0 < N; r0 += R) {
      const int Rc = std::min(R, N - r0);
      int pp = 0;
      for (int l = 0; l < L; ++l) {
        const int fin = l == 0 ? F : ds.widths[l - 1], fout = ds.widths[l];
        const float *in = l == 0 ? X + (size_t)r0 * F : buf[pp ^ 1];
        const long long sin = l == 0 ? 0 : (long long)Rc * fin;
        if (g_rb.sgemm_sb(s->blas, RB_OP_N, RB_OP_N, fout, Rc, fin, &one, th + ds.w_off[l], fout, d, in, fin, sin, &zero, buf[pp], fout,
                          (long long)Rc * fout, Sn))
          return fail(MILE_ERR_HIP, "rocblas sgemm (evaluation) failed");
        const long long RW = (long long)Rc * fout;
        k_gemm_bias_act<<<dim3((unsigned)std::min<long long>((RW + 255) / 256, 4096), Sn), 256, 0, st>>>(buf[pp], th, ds.b_off[l], d, fout, RW,
                                                                                                   ds.activation, l + 1 < L);
        pp ^= 1;
      }
      k_gemm_rowll<<<dim3((Rc + 255) / 256, Sn), 256, 0, st>>>(buf[pp ^ 1], y, r0, Rc, ds.widths[L - 1], ds.task, out, N, s0);
    }
  }
  HIP_TRY(hipGetLastError());
  return MILE_OK;
}

// Evaluation forward for wide nets on the MFMA GEMMs (fp32-faithful, whatever the sampling kernel was): samples as the batch.
static int launch_fwd_wide(mile_sampler *s, const float *theta, int S, const float *Xp, int Fp, const void *y, int N, float *out,
                           hipStream_t st) {
  const DevSpec &ds = s->ds;
  const int L = ds.n_layers, d = ds.d;
  auto up8 = [](int v) { return (v + 7) / 8 * 8; };
  int wp[MILE_MAX_LAYERS], fin[MILE_MAX_LAYERS];
  size_t wt_elems = 0, wt_off[MILE_MAX_LAYERS];
  int maxwp = 0;
  for (int l = 0; l < L; ++l) {
    wp[l] = up8(ds.widths[l]);
    fin[l] = l == 0 ? ds.in_features : ds.widths[l - 1];
    maxwp = std::max(maxwp, wp[l]);
    wt_off[l] = wt_elems;
    wt_elems += (size_t)3 * fin[l] * wp[l];
  }
  const int Sc = std::min(S, 512);
  size_t Rr = ((size_t)1 << 28) / (2 * (size_t)Sc * maxwp);           // two ping-pong buffers, <= 1 GiB of floats
  Rr = std::max<size_t>(1, std::min<size_t>(Rr, (size_t)N));
  if (const char *rv = getenv("MILE_GEMM_ROWS")) Rr = std::max<size_t>(1, std::min<size_t>((size_t)atoll(rv), (size_t)N));
  const int R = (int)Rr;
  const size_t need = 2 * (size_t)Sc * R * maxwp;
  if (need > s->wide_ws_floats) {
    if (s->wide_ws) (void)hipFree(s->wide_ws);
    s->wide_ws = nullptr; s->wide_ws_floats = 0;
    HIP_TRY(hipMalloc(&s->wide_ws, need * 4));
    s->wide_ws_floats = need;
  }
  s->wide_E = 0;   // the gradient path re-derives (and re-zeroes) its layout on its next call
  const size_t wt_bytes = (size_t)Sc * wt_elems * 2;
  if (wt_bytes > s->wide_wt_bytes) {
    if (s->wide_wt) (void)hipFree(s->wide_wt);
    s->wide_wt = nullptr; s->wide_wt_bytes = 0;
    HIP_TRY(hipMalloc(&s->wide_wt, wt_bytes));
    s->wide_wt_bytes = wt_bytes;
  }
  float *buf[2] = {s->wide_ws, s->wide_ws + (size_t)Sc * R * maxwp};
  bf16 *Wt = (bf16 *)s->wide_wt;
  for (int s0 = 0; s0 < S; s0 += Sc) {
    const int Sn = std::min(Sc, S - s0);
    const float *th = theta + (size_t)s0 * d;
    for (int l = 0; l < L; ++l) {
      const long long plane = (long long)fin[l] * wp[l];
      k_wide_prep_weights<3><<<dim3((unsigned)std::min<long long>((plane + 255) / 256, 1024), Sn), 256, 0, st>>>(
          th, d, ds.w_off[l], fin[l], ds.widths[l], wp[l], Wt + wt_off[l], (long long)wt_elems);
    }
    for (int r0 = 0; r0 < N; r0 += R) {
      const int Rc = std::min(R, N - r0);
      // layers of different widths share the two buffers: padding columns an operand read may touch must be zero
      HIP_TRY(hipMemsetAsync(s->wide_ws, 0, need * 4, st));
      int pp = 0;
      for (int l = 0; l < L; ++l) {
        MMParams p{};
        if (l == 0) { p.A = Xp + (size_t)r0 * Fp; p.sA = 0; p.lda = Fp; }
        else { p.A = buf[pp ^ 1]; p.sA = (long long)R * wp[l - 1]; p.lda = wp[l - 1]; }
        p.K = fin[l];
        p.B = Wt + wt_off[l]; p.sB = (long long)wt_elems; p.ldb = wp[l]; p.tB = (long long)fin[l] * wp[l];
        p.C = buf[pp]; p.sC = (long long)R * wp[l]; p.ldc = wp[l];
        p.M = Rc; p.N = ds.widths[l];
        p.bias = th + ds.b_off[l]; p.sBias = d;
        p.act = ds.activation; p.apply_act = l + 1 < L;
        HIP_TRY(launch_mm3_fwd<3>(p, Sn, st));
        pp ^= 1;
      }
      k_wide_rowll<<<dim3((Rc + 255) / 256, Sn), 256, 0, st>>>(buf[pp ^ 1], (long long)R * wp[L - 1], wp[L - 1], y, r0, Rc, ds.widths[L - 1], ds.task, out, N, s0);
    }
  }
  HIP_TRY(hipGetLastError());
  return MILE_OK;
}

extern "C" int32_t mile_pointwise_loglik(mile_sampler *s, const float *theta, int32_t S, const float *X, const void *y,
                                         int64_t N, float *out, void *stream) {
  if (!s || !theta || !X || !y || !out || S < 1) return fail(MILE_ERR_INVALID, "mile_pointwise_loglik: bad argument");
  if (N < 1 || N > 0x3fffffff) return fail(MILE_ERR_INVALID, "mile_pointwise_loglik: N out of range");
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(s->device));
  const int F = s->spec.in_features, Npad = ((int)N + 31) / 32 * 32, Fp = (F + 7) / 8 * 8;
  if (Npad > s->ev_cap) {   // evaluation is off the stepping path: (re)allocate its staging here
    if (s->ev_X) (void)hipFree(s->ev_X);
    if (s->ev_Xp) (void)hipFree(s->ev_Xp);
    if (s->ev_y) (void)hipFree(s->ev_y);
    s->ev_X = s->ev_Xp = nullptr; s->ev_y = nullptr; s->ev_cap = 0;
    HIP_TRY(hipMalloc(&s->ev_X, (size_t)Npad * F * 4));
    HIP_TRY(hipMalloc(&s->ev_Xp, (size_t)Npad * Fp * 4));
    HIP_TRY(hipMalloc(&s->ev_y, (size_t)Npad * 4));
    s->ev_cap = Npad;
  }
  HIP_TRY(hipMemcpyAsync(s->ev_X, X, (size_t)N * F * 4, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemsetAsync(s->ev_y, 0, (size_t)Npad * 4, st));
  HIP_TRY(hipMemcpyAsync(s->ev_y, y, (size_t)N * 4, hipMemcpyDeviceToDevice, st));
  const long long tot = (long long)Npad * Fp;
  k_pad_x<<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(s->ev_X, s->ev_Xp, (int)N, Npad, F, Fp);
  PredParams pp;
  pp.spec = s->ds; pp.theta = theta; pp.X = s->ev_X; pp.Xp = s->ev_Xp; pp.y = s->ev_y; pp.out = out;
  pp.N = (int)N; pp.Npad = Npad; pp.Fp = Fp; pp.R = generic_R(s->ds);
  const int kernel = resolved_kernel(s);
  if (kernel == MILE_GRAD_LENET_F32 || kernel == MILE_GRAD_LENET_BF16) {
    for (int s0 = 0; s0 < S; s0 += 256) {
      const int rc = run_lenet(s, theta + (size_t)s0 * s->ds.d, std::min(256, S - s0), s->ev_X, s->ev_y, (int)N, nullptr, 0, nullptr, out, s0, st,
                               kernel == MILE_GRAD_LENET_BF16);
      if (rc) return rc;
    }
    return MILE_OK;
  }
  if (kernel == MILE_GRAD_GEMM_F32) return launch_fwd_gemm(s, theta, S, s->ev_X, s->ev_y, (int)N, out, st);
  if (kernel == MILE_GRAD_MFMA_WIDE_BF16X3 || kernel == MILE_GRAD_MFMA_WIDE_BF16 || kernel == MILE_GRAD_MFMA_W128_BF16) {
    // wide nets: evaluation stays fp32-faithful whatever the sampling kernel was
    return launch_fwd_wide(s, theta, S, s->ev_Xp, Fp, s->ev_y, (int)N, out, st);
  }
  if (is_w64(kernel)) {
    const int NB = Npad / 32;
    pp.SB = std::max(1, std::min(std::max(1, (2 * s->n_cu) / S), std::max(1, NB / 4)));
    const int nh = s->spec.n_layers - 1, fq = Fp / 8;
    hipError_t e = hipErrorInvalidValue;
    if (nh == 1 && fq == 1) e = launch_fwd_w64<1, 1>(pp, S, st);
    else if (nh == 2 && fq == 1) e = launch_fwd_w64<2, 1>(pp, S, st);
    else if (nh == 3 && fq == 1) e = launch_fwd_w64<3, 1>(pp, S, st);
    else if (nh == 1 && fq == 2) e = launch_fwd_w64<1, 2>(pp, S, st);
    else if (nh == 2 && fq == 2) e = launch_fwd_w64<2, 2>(pp, S, st);
    else if (nh == 3 && fq == 2) e = launch_fwd_w64<3, 2>(pp, S, st);
    HIP_TRY(e);
  } else {
    pp.SB = std::max(1, std::min(std::max(1, (4 * s->n_cu) / S), std::max(1, (int)N / 64)));
    const size_t lds = ((size_t)pp.R * s->ds.act_stride + 16) * 4;
    k_fwd_generic<<<dim3(pp.SB, S), 256, lds, st>>>(pp);
    HIP_TRY(hipGetLastError());
  }
  return MILE_OK;
}

extern "C" {

int32_t mile_grad_launch_info(const mile_sampler *s, int32_t E, int32_t *grid_x, int32_t *grid_y,
                              int32_t *block, int32_t *lds_bytes, char *name, int32_t name_len) {
  if (!s || !s->X) return fail(MILE_ERR_STATE, "mile_grad_launch_info: no data set");
  const int kernel = resolved_kernel(s);
  const int S = choose_S(s, E, kernel);
  if (grid_x) *grid_x = S;
  if (grid_y) *grid_y = E;
  if (block) *block = 256;
  int lds = 0;
  const char *nm = "k_grad_generic";
  if (kernel == MILE_GRAD_MFMA_W64) {
    const int nh = s->spec.n_layers - 1, fq = s->Fp / 8;
    nm = "k_grad_w64";
    lds = nh == 1 ? (fq == 1 ? w64_lds_bytes<1, 1>() : w64_lds_bytes<1, 2>())
        : nh == 2 ? (fq == 1 ? w64_lds_bytes<2, 1>() : w64_lds_bytes<2, 2>())
                  : (fq == 1 ? w64_lds_bytes<3, 1>() : w64_lds_bytes<3, 2>());
  } else if (kernel == MILE_GRAD_MFMA_W64_BF16X3) {
    const int nh = s->spec.n_layers - 1, fq = s->Fp / 8;
    nm = "k_grad_w64";
    lds = nh == 2 ? (fq == 1 ? w64_lds_bytes<2, 1, true>() : w64_lds_bytes<2, 2, true>())
                  : (fq == 1 ? w64_lds_bytes<3, 1, true>() : w64_lds_bytes<3, 2, true>());
  } else if (kernel == MILE_GRAD_MFMA_NARROW_F32) {
    nm = "k_grad_narrow";
    if (block) *block = 64 * (narrow_tiles_hidden(s->spec) >= 3 ? NRW_MAXW : narrow_waves(s, S, s->N));
    lds = narrow_tiles_hidden(s->spec) >= 3 ? NarrowLayout<3, 4, 4, true>::BYTES      // upper bounds over the instantiations
                                            : std::max(NarrowLayout<3, 2, 4>::BYTES, NarrowLayout<10, 1, 1>::BYTES);
  } else if (kernel == MILE_GRAD_LENET_BF16) {
    nm = "k_conv5m_fwd/dx/dw (implicit-GEMM bf16 MFMA) + k_mm3 (Dense, fp32-faithful three-term products)";
    lds = (int)cm_lds_dw(CM_IN8, s->lg.hp1, s->lg.wp1, 0);
  } else if (kernel == MILE_GRAD_LENET_F32) {
    nm = "rocblas_sgemm_strided_batched+k_im2col5/k_col2im5/k_avgpool2";
    lds = 0;
  } else if (kernel == MILE_GRAD_MFMA_WIDE_BF16X3 || kernel == MILE_GRAD_MFMA_WIDE_BF16) {
    nm = "k_mm3 (layer-wise MFMA GEMMs)";
    lds = kernel == MILE_GRAD_MFMA_WIDE_BF16X3 ? MMLayout<MM_A_MK, MM_B_T3_NK, 3, MILE_MM_KC>::BYTES : MMLayout<MM_A_MK, MM_B_T3_NK, 1, MILE_MM_KC>::BYTES;
  } else if (kernel == MILE_GRAD_GEMM_F32) {
    nm = "rocblas_sgemm_strided_batched+k_gemm_*";
    lds = 0;
  } else if (kernel == MILE_GRAD_MFMA_W128_BF16) {
    const int nh = s->spec.n_layers - 1;
    nm = "k_grad_w128b";
    lds = nh == 1 ? W128Layout<1, 2>::BYTES : nh == 2 ? W128Layout<2, 2>::BYTES : W128Layout<3, 2>::BYTES;
  } else {
    lds = (int)(((size_t)generic_R(s->ds) * (s->ds.act_stride + 2 * s->ds.max_width) + 16) * 4);
  }
  if (lds_bytes) *lds_bytes = lds;
  if (name && name_len > 0) { std::strncpy(name, nm, name_len - 1); name[name_len - 1] = 0; }
  return MILE_OK;
}

int32_t mile_grad_timing_begin(mile_sampler *s) {
  if (!s) return fail(MILE_ERR_INVALID, "null handle");
  s->timing = true;
  s->ev_used = 0;
  return MILE_OK;
}

int32_t mile_grad_timing_end(mile_sampler *s, float *total_ms, int32_t *n_launches) {
  if (!s) return fail(MILE_ERR_INVALID, "null handle");
  s->timing = false;
  double tot = 0.0;
  const size_t n = s->ev_used / 2;
  if (n) HIP_TRY(hipEventSynchronize(s->ev[s->ev_used - 1]));
  for (size_t k = 0; k < n; ++k) {
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev[2 * k], s->ev[2 * k + 1]));
    tot += ms;
  }
  if (total_ms) *total_ms = (float)tot;
  if (n_launches) *n_launches = (int32_t)n;
  s->ev_used = 0;
  return MILE_OK;
}

int32_t mile_logpost_grad(mile_sampler *s, const float *theta, int32_t E, float *logp, float *grad, void *stream) {
  if (!s || !theta || !logp || !grad || E < 1) return fail(MILE_ERR_INVALID, "mile_logpost_grad: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(s->device));
  int rc = launch_grad(s, theta, E, st);
  if (rc) return rc;
  const int S = choose_S(s, E, resolved_kernel(s));
  k_finalize<<<E, AUX_NT, 0, st>>>(s->ds.d, (s->ds.d + 3) / 4 * 4, S, s->ds.prior, s->ds.prior_loc, s->ds.prior_scale, theta,
                                   s->slabs, s->llpart, grad, logp);
  HIP_TRY(hipGetLastError());
  return MILE_OK;
}

int32_t mile_warmstart_step(mile_sampler *s, float *theta, int32_t E, const mile_optim_args *a, void *stream) {
  if (!s || !theta || !a || E < 1) return fail(MILE_ERR_INVALID, "mile_warmstart_step: bad argument");
  if (a->kind < MILE_OPT_SGD || a->kind > MILE_OPT_ADAMW) return fail(MILE_ERR_INVALID, "mile_warmstart_step: unknown optimizer");
  if (a->kind != MILE_OPT_SGD && (!a->m || !a->v)) return fail(MILE_ERR_INVALID, "mile_warmstart_step: adam / adamw need m and v");
  if (a->t < 1) return fail(MILE_ERR_INVALID, "mile_warmstart_step: t counts from 1");
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(s->device));
  const int rc = launch_grad(s, theta, E, st);          // likelihood gradient of the row window -> slabs, llpart
  if (rc) return rc;
  OptimParams op{};
  op.d = s->ds.d; op.S = choose_S(s, E, resolved_kernel(s)); op.dp = (s->ds.d + 3) / 4 * 4; op.kind = a->kind;
  op.lr = a->learning_rate; op.b1 = a->b1; op.b2 = a->b2; op.eps = a->eps; op.wd = a->weight_decay;
  op.inv_batch = 1.0f / (float)(s->win_count ? s->win_count : s->N);
  op.bc1 = (float)(1.0 - std::pow((double)a->b1, (double)a->t));
  op.bc2 = (float)(1.0 - std::pow((double)a->b2, (double)a->t));
  op.theta = theta; op.m = a->m; op.v = a->v; op.slabs = s->slabs; op.llpart = s->llpart; op.active = a->active; op.out_nll = a->out_nll;
  k_optim_step<<<dim3((op.d + 255) / 256, E), 256, 0, st>>>(op);
  HIP_TRY(hipGetLastError());
  return MILE_OK;
}

int32_t mile_init(mile_sampler *s, mile_state *state, const float *noise, uint64_t seed,
                  const int32_t *particle_ids, void *stream) {
  if (!s || !state || !state->position || !state->momentum || !state->logdensity || !state->logdensity_grad)
    return fail(MILE_ERR_INVALID, "mile_init: null state field");
  const int E = state->n_particles;
  if (E < 1) return fail(MILE_ERR_INVALID, "mile_init: n_particles must be >= 1");
  hipStream_t st = (hipStream_t)stream;
  int rc = mile_logpost_grad(s, state->position, E, state->logdensity, state->logdensity_grad, stream);
  if (rc) return rc;
  k_init_momentum<<<E, AUX_NT, 0, st>>>(s->ds.d, noise, seed, particle_ids, state->momentum);
  HIP_TRY(hipGetLastError());
  return MILE_OK;
}

int32_t mile_step(mile_sampler *s, mile_state *state, const mile_step_args *a, void *stream) {
  if (!s || !state || !a) return fail(MILE_ERR_INVALID, "mile_step: null argument");
  if (!state->position || !state->momentum || !state->logdensity || !state->logdensity_grad)
    return fail(MILE_ERR_INVALID, "mile_step: null state field");
  if (!a->step_size || !a->L) return fail(MILE_ERR_INVALID, "mile_step: step_size and L are required");
  if (a->n_steps < 0) return fail(MILE_ERR_INVALID, "mile_step: n_steps < 0");
  if (a->refresh != MILE_REFRESH_O_STEP_O && a->refresh != MILE_REFRESH_STEP_O)
    return fail(MILE_ERR_INVALID, "mile_step: unknown refresh mode");
  const int E = state->n_particles, d = s->ds.d;
  if (E < 1) return fail(MILE_ERR_INVALID, "mile_step: n_particles must be >= 1");
  if (a->n_steps == 0) return MILE_OK;
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(s->device));
  const int kernel = resolved_kernel(s);
  const int S = choose_S(s, E, kernel);
  if (!s->X) return fail(MILE_ERR_STATE, "no data: call mile_set_data first");
  if (E > s->E_cap || (size_t)E * S > s->ES_cap) return fail(MILE_ERR_STATE, "workspace too small: call mile_reserve(E) first");

  UpdParams up{};
  up.d = d; up.E = E; up.S = S; up.dp = (d + 3) / 4 * 4;
  up.prior = s->ds.prior; up.prior_loc = s->ds.prior_loc; up.prior_scale = s->ds.prior_scale;
  up.x = state->position; up.u = state->momentum; up.g = state->logdensity_grad; up.logp = state->logdensity;
  up.slabs = s->slabs; up.llpart = s->llpart;
  up.eps = a->step_size; up.L = a->L; up.sdc = a->sqrt_diag_cov;
  up.seed = a->seed; up.pids = a->particle_ids;
  up.dK = s->dK; up.lold = s->lold; up.upart = s->upart;
  const float b1 = (float)MCLACHLAN_B1, b2 = (float)(1.0 - 2.0 * MCLACHLAN_B1);
  const bool oso = a->refresh == MILE_REFRESH_O_STEP_O;
  const size_t Ed = (size_t)E * d;
  auto noise_at = [&](int i, int k) -> const float * { return a->noise ? a->noise + ((size_t)i * 2 + k) * Ed : nullptr; };

  // two launches per step when the updates can run as the grad kernel's epilogue (k_grad_w64 SPLIT), else four
  UpdParams probe = up;
  probe.zA = a->noise; probe.zB = a->noise; probe.out_sample = a->out_samples;
  const bool fused = fuse_ok(s, kernel, probe);
  if (fused) HIP_TRY(hipMemsetAsync(s->arrive, 0, ((size_t)E * 4 + 15) / 16 * 16, st));   // tickets re-zeroed every call
  int kept = 0;
  for (int i = 0; i < a->n_steps; ++i) {
    const int64_t gstep = a->step_offset + i;
    if (i == 0) {  // O(z1) . B(b1) . A(1/2) from the cached gradient
      UpdParams u = up;
      u.flags = UPD_START | UPD_B2 | UPD_A | (oso ? UPD_OB : 0);
      u.zB = noise_at(i, 0); u.stepB = (uint32_t)gstep; u.stageB = 0; u.hB = 0.5f;
      u.coef_b2 = b1; u.coef_a = 0.5f;
      launch_update(u, E, st);
    }
    {  // grad . B(1 - 2 b1) . A(1/2)
      UpdParams u = up;
      u.flags = UPD_FROM_SLABS | UPD_B1 | UPD_A | UPD_NO_G;
      u.coef_b1 = b2; u.coef_a = 0.5f;
      const int rc = grad_then_update(s, state->position, E, u, fused, st);
      if (rc) return rc;
    }
    {  // grad . B(b1) . O(z2) . record  [ . O(z1') . B(b1) . A(1/2) of the next step ]
      UpdParams u = up;
      const bool last = (i == a->n_steps - 1);
      u.flags = UPD_FROM_SLABS | UPD_B1 | UPD_OA | UPD_RECORD;
      u.coef_b1 = b1;
      u.zA = noise_at(i, 1); u.stepA = (uint32_t)gstep; u.stageA = 1; u.hA = oso ? 0.5f : 1.0f;
      if (!last) {
        u.flags |= UPD_B2 | UPD_A | (oso ? UPD_OB : 0) | UPD_NO_G;
        u.zB = noise_at(i + 1, 0); u.stepB = (uint32_t)(gstep + 1); u.stageB = 0; u.hB = 0.5f;
        u.coef_b2 = b1; u.coef_a = 0.5f;
      }
      if (a->out_info) u.out_info = a->out_info + (size_t)i * E * 3;
      if (a->out_samples && a->n_thinning > 0 && (gstep % a->n_thinning) == 0) {
        u.out_sample = a->out_samples + (size_t)kept * Ed;
        ++kept;
      }
      const int rc = grad_then_update(s, state->position, E, u, fused, st);
      if (rc) return rc;
    }
  }
  HIP_TRY(hipGetLastError());
  return MILE_OK;
}

int32_t mile_tune(mile_sampler *s, mile_state *state, const mile_tune_args *a, void *stream) {
  if (!s || !state || !a) return fail(MILE_ERR_INVALID, "mile_tune: null argument");
  if (!state->position || !state->momentum || !state->logdensity || !state->logdensity_grad)
    return fail(MILE_ERR_INVALID, "mile_tune: null state field");
  if (!a->step_size || !a->L || !a->step_size_max || !a->time || !a->x_average || !a->stream_weight || !a->stream_average)
    return fail(MILE_ERR_INVALID, "mile_tune: null tuner array");
  if (a->n_steps < 0 || a->schedule_total < 1) return fail(MILE_ERR_INVALID, "mile_tune: bad step counts");
  if (a->refresh != MILE_REFRESH_O_STEP_O && a->refresh != MILE_REFRESH_STEP_O)
    return fail(MILE_ERR_INVALID, "mile_tune: unknown refresh mode");
  const int E = state->n_particles, d = s->ds.d;
  if (E < 1) return fail(MILE_ERR_INVALID, "mile_tune: n_particles must be >= 1");
  if ((d >> 2) < 1) return fail(MILE_ERR_INVALID, "mile_tune: d < 4");
  const bool big = (d >> 2) > UPD_NT * UPD_QMAX || getenv("MILE_TUNE_POST") != nullptr;   // beyond k_update_fast's register cache
  if (a->n_steps == 0) return MILE_OK;
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(s->device));
  if (!s->X) return fail(MILE_ERR_STATE, "no data: call mile_set_data first");
  const int S = choose_S(s, E, resolved_kernel(s));
  if (E > s->E_cap || (size_t)E * S > s->ES_cap) return fail(MILE_ERR_STATE, "workspace too small: call mile_reserve(E) first");

  struct Buf { float *x, *u, *g, *logp; };
  const Buf A{state->position, state->momentum, state->logdensity_grad, state->logdensity};
  const Buf B{s->alt_x, s->alt_u, s->alt_g, s->alt_logp};
  UpdParams up{};
  up.d = d; up.E = E; up.S = S; up.dp = (d + 3) / 4 * 4;
  up.prior = s->ds.prior; up.prior_loc = s->ds.prior_loc; up.prior_scale = s->ds.prior_scale;
  up.slabs = s->slabs; up.llpart = s->llpart;
  up.eps = a->step_size; up.L = a->L; up.sdc = a->sqrt_diag_cov;
  up.seed = a->seed; up.pids = a->particle_ids;
  up.dK = s->dK; up.lold = s->lold; up.upart = s->upart;
  const float b1 = (float)MCLACHLAN_B1, b2 = (float)(1.0 - 2.0 * MCLACHLAN_B1);
  const bool oso = a->refresh == MILE_REFRESH_O_STEP_O;
  const size_t Ed = (size_t)E * d;
  auto noise_at = [&](int i, int k) -> const float * { return a->noise ? a->noise + ((size_t)i * 2 + k) * Ed : nullptr; };
  auto set_state = [](UpdParams &u, const Buf &b) { u.x = b.x; u.u = b.u; u.g = b.g; u.logp = b.logp; };
  auto target_var = [&](int sp) -> float {
    const double tot = (double)a->schedule_total, vs = a->desired_energy_var_start, ve = a->desired_energy_var_end;
    if (vs > 2.0) {
      const double tau = tot / 4.0, ex = std::exp(-(double)sp / tau);
      return (float)(vs * ex + ve * (1.0 - ex));
    }
    return (float)(vs - (vs - ve) * std::min((double)sp / tot, 1.0));
  };

  if (big) {
    // Large d: an ordinary in-place kernel step (two-pass update kernels), the previous state copied aside first,
    // then k_tune_post applies the predictor, the rejection of non-finite steps and the streaming averages.
    if (!s->tune_info) HIP_TRY(hipMalloc(&s->tune_info, (size_t)s->E_cap * 3 * 4));
    for (int i = 0; i < a->n_steps; ++i) {
      const int64_t gstep = a->step_offset + i;
      HIP_TRY(hipMemcpyAsync(B.x, A.x, Ed * 4, hipMemcpyDeviceToDevice, st));
      HIP_TRY(hipMemcpyAsync(B.u, A.u, Ed * 4, hipMemcpyDeviceToDevice, st));
      HIP_TRY(hipMemcpyAsync(B.g, A.g, Ed * 4, hipMemcpyDeviceToDevice, st));
      HIP_TRY(hipMemcpyAsync(B.logp, A.logp, (size_t)E * 4, hipMemcpyDeviceToDevice, st));
      float *info_i = a->out_info ? a->out_info + (size_t)i * E * 3 : s->tune_info;
      {
        UpdParams u = up;
        set_state(u, A);
        u.flags = UPD_START | UPD_B2 | UPD_A | (oso ? UPD_OB : 0);
        u.zB = noise_at(i, 0); u.stepB = (uint32_t)gstep; u.stageB = 0; u.hB = 0.5f;
        u.coef_b2 = b1; u.coef_a = 0.5f;
        launch_update(u, E, st);
      }
      int rc = launch_grad(s, A.x, E, st);
      if (rc) return rc;
      {
        UpdParams u = up;
        set_state(u, A);
        u.flags = UPD_FROM_SLABS | UPD_B1 | UPD_A | UPD_NO_G;
        u.coef_b1 = b2; u.coef_a = 0.5f;
        launch_update(u, E, st);
      }
      rc = launch_grad(s, A.x, E, st);
      if (rc) return rc;
      {
        UpdParams u = up;
        set_state(u, A);
        u.flags = UPD_FROM_SLABS | UPD_B1 | UPD_OA | UPD_RECORD;
        u.coef_b1 = b1;
        u.zA = noise_at(i, 1); u.stepA = (uint32_t)gstep; u.stageA = 1; u.hA = oso ? 0.5f : 1.0f;
        u.out_info = info_i;
        launch_update(u, E, st);
      }
      TunePostParams tp{};
      tp.d = d; tp.x = A.x; tp.u = A.u; tp.g = A.g; tp.logp = A.logp;
      tp.bk_x = B.x; tp.bk_u = B.u; tp.bk_g = B.g; tp.bk_logp = B.logp;
      tp.info = info_i;
      tp.t_eps = a->step_size; tp.t_eps_max = a->step_size_max; tp.t_time = a->time; tp.t_xavg = a->x_average;
      tp.t_W = a->stream_weight; tp.t_avg = a->stream_average;
      const int sp = a->schedule_step0 + i;
      tp.t_mask = sp < a->n_mask_steps ? 1.0f : 0.0f;
      tp.t_var = target_var(sp);
      tp.t_trust = a->trust_in_estimate; tp.t_decay = a->decay_rate;
      k_tune_post<<<E, AUX_NT, 0, st>>>(tp);
    }
    HIP_TRY(hipGetLastError());
    return MILE_OK;
  }

  UpdParams probe = up;
  probe.x = A.x; probe.u = A.u; probe.g = A.g; probe.x_in = B.x; probe.u_in = B.u; probe.g_in = B.g;
  probe.zA = a->noise; probe.t_avg = a->stream_average;
  if (a->n_steps > 1) probe.u_rec = A.u;
  const bool fused = fuse_ok(s, resolved_kernel(s), probe);
  if (fused) HIP_TRY(hipMemsetAsync(s->arrive, 0, ((size_t)E * 4 + 15) / 16 * 16, st));
  // Per step: grad . update(B, A) . grad . update(B, O, record + TUNE [. O, B, A of the next step with the NEW step size]) --
  // four launches, as a sampling step (mile_step).  Step i works in buffer W_i and leaves the accepted state there; its last
  // launch writes the next step's working position / momentum into the other buffer, which until then held the state before
  // step i (what handle_nans reverts to): W_{i+1} = K_i, K_{i+1} = W_i, no copies.
  const bool no_merge = getenv("MILE_TUNE_NO_MERGE") != nullptr;      // dev / test: the five-launch form of rounds 1-2
  for (int i = 0; i < a->n_steps; ++i) {
    const Buf &cur = (i & 1) ? B : A, &nxt = (i & 1) ? A : B;
    const int64_t gstep = a->step_offset + i;
    if (i == 0 || no_merge) {  // O(z1) . B(b1) . A(1/2): reads the current state, writes the other buffer (free backup)
      UpdParams u = up;
      set_state(u, nxt);
      u.x_in = cur.x; u.u_in = cur.u; u.g_in = cur.g; u.logp_in = cur.logp;
      u.flags = UPD_START | UPD_B2 | UPD_A | (oso ? UPD_OB : 0);
      u.zB = noise_at(i, 0); u.stepB = (uint32_t)gstep; u.stageB = 0; u.hB = 0.5f;
      u.coef_b2 = b1; u.coef_a = 0.5f;
      launch_update(u, E, st);
    }
    {
      UpdParams u = up;
      set_state(u, nxt);
      u.flags = UPD_FROM_SLABS | UPD_B1 | UPD_A | UPD_NO_G;
      u.coef_b1 = b2; u.coef_a = 0.5f;
      const int rc = grad_then_update(s, nxt.x, E, u, fused, st);
      if (rc) return rc;
    }
    {  // B(b1) . O(z2) . record + tuner (step-size predictor, handle_nans, streaming averages) [+ the next step's O, B, A]
      UpdParams u = up;
      set_state(u, nxt);
      u.flags = UPD_FROM_SLABS | UPD_B1 | UPD_OA | UPD_RECORD | UPD_TUNE;
      u.coef_b1 = b1;
      u.zA = noise_at(i, 1); u.stepA = (uint32_t)gstep; u.stageA = 1; u.hA = oso ? 0.5f : 1.0f;
      u.bk_x = cur.x; u.bk_u = cur.u; u.bk_g = cur.g; u.bk_logp = cur.logp;
      u.t_eps = a->step_size; u.t_eps_max = a->step_size_max; u.t_time = a->time; u.t_xavg = a->x_average;
      u.t_W = a->stream_weight; u.t_avg = a->stream_average;
      const int sp = a->schedule_step0 + i;
      u.t_mask = sp < a->n_mask_steps ? 1.0f : 0.0f;
      u.t_var = target_var(sp);
      u.t_trust = a->trust_in_estimate; u.t_decay = a->decay_rate;
      if (a->out_info) u.out_info = a->out_info + (size_t)i * E * 3;
      if (i + 1 < a->n_steps && !no_merge) {
        u.flags |= UPD_B2 | UPD_A | (oso ? UPD_OB : 0);
        u.zB = noise_at(i + 1, 0); u.stepB = (uint32_t)(gstep + 1); u.stageB = 0; u.hB = 0.5f;
        u.coef_b2 = b1; u.coef_a = 0.5f;
        u.x_in = nxt.x; u.u_in = nxt.u;          // the step ran in nxt; its accepted state stays there ...
        u.x_acc = nxt.x; u.u_rec = nxt.u;
        u.x = cur.x; u.u = cur.u;                // ... and the next step's working state goes to the other buffer
        u.force_restart = getenv("MILE_TUNE_FORCE_RESTART") != nullptr;
      }
      const int rc = grad_then_update(s, nxt.x, E, u, fused, st);
      if (rc) return rc;
    }
  }
  if (a->n_steps & 1) {   // the final state sits in the library's buffer
    HIP_TRY(hipMemcpyAsync(A.x, B.x, Ed * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(A.u, B.u, Ed * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(A.g, B.g, Ed * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(A.logp, B.logp, (size_t)E * 4, hipMemcpyDeviceToDevice, st));
  }
  HIP_TRY(hipGetLastError());
  return MILE_OK;
}

int32_t mile_debug_noise(mile_sampler *s, uint64_t seed, const int32_t *particle_ids, int32_t E,
                         int64_t step, int32_t stage, float *out, void *stream) {
  if (!s || !out || E < 1) return fail(MILE_ERR_INVALID, "mile_debug_noise: bad argument");
  HIP_TRY(hipSetDevice(s->device));
  k_debug_noise<<<E, AUX_NT, 0, (hipStream_t)stream>>>(s->ds.d, seed, particle_ids, (uint32_t)step, (uint32_t)stage, out);
  HIP_TRY(hipGetLastError());
  return MILE_OK;
}

}  // extern "C"
