// k_grad_w64<NH, 2, true>: the split-bf16 width-64 grad kernel for 9..16 input features, in a translation unit of its own
// because it must be built without -mllvm -amdgpu-mfma-vgpr-form (see mile_amd/_build.py and the note in mile_hip.hip).
#include <hip/hip_runtime.h>

#include "mile_grad_w64.h"

template <int NH>
static hipError_t launch(const GradParams &gp, int E, hipStream_t st) {
  using LY = W64Layout<NH, 2, true>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)k_grad_w64<NH, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LY::BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  k_grad_w64<NH, 2, true><<<dim3(gp.S, E), 256, LY::BYTES, st>>>(gp, W64NoFuse{0});
  return hipGetLastError();
}

hipError_t mile_launch_w64_split_fq2(int nh, const GradParams &gp, int E, hipStream_t st) {
  if (nh == 2) return launch<2>(gp, E, st);
  if (nh == 3) return launch<3>(gp, E, st);
  return hipErrorInvalidValue;
}
