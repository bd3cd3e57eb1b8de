"""Build libmile_hip.so in-tree with hipcc for gfx950 (MI355X).

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / 'csrc'
LIB_PATH = PKG_DIR / 'libmile_hip.so'
# translation units and their extra flags.  -amdgpu-mfma-vgpr-form: MFMA results that the VALU consumes next land in VGPRs
# instead of AGPRs (fewer v_accvgpr moves and spills in the register-bound grad kernels; measured +0.5 % w64, +2 % w128b);
# mile_w64_fq2.hip is built without it (hipcc 7.2 crashes on k_grad_w64<3,2,true> with it).
VGPR_FORM = ['-mllvm', '-amdgpu-mfma-vgpr-form']
SOURCES = {'mile_hip.hip': VGPR_FORM, 'mile_w64_fq2.hip': []}
HEADERS = ['mile_device.h', 'mile_grad_generic.h', 'mile_grad_narrow.h', 'mile_grad_w64.h', 'mile_grad_w64_block.inc', 'mile_bf16_frag.h',
           'mile_grad_w128b.h', 'mile_grad_gemm.h', 'mile_mm3.h', 'mile_lenet.h', 'mile_lenet_mfma.h', 'mile_predict.h', 'mile_update.h']
OBJ_DIR = PKG_DIR / 'csrc' / '_obj'


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError('hipcc not found: libmile_hip.so cannot be built')


def needs_build() -> bool:
    if not LIB_PATH.exists():
        return True
    t = LIB_PATH.stat().st_mtime
    deps = [CSRC / f for f in list(SOURCES) + HEADERS] + [PKG_DIR.parent / 'include' / 'mile_hip.h']
    return any(p.stat().st_mtime > t for p in deps)


def build_library(force: bool = False, verbose: bool = False) -> Path:
    """hipcc --offload-arch=gfx950: one object per translation unit (compiled concurrently), then
    -shared -> mile_amd/libmile_hip.so."""
    if not force and not needs_build():
        return LIB_PATH
    extra = os.environ.get('MILE_HIPCC_FLAGS', '').split()          # dev: extra compiler flags for experiments
    OBJ_DIR.mkdir(exist_ok=True)
    base = [_hipcc(), '-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC']
    jobs = []
    for src, flags in SOURCES.items():
        obj = OBJ_DIR / (Path(src).stem + '.o')
        cmd = base + flags + extra + ['-c', str(CSRC / src), '-o', str(obj)]
        if verbose:
            print(' '.join(cmd))
        jobs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    objs = []
    for src, obj, proc in jobs:
        out, err = proc.communicate()
        if proc.returncode != 0:
            for _, _, other in jobs:
                if other.poll() is None:
                    other.kill()
            raise RuntimeError(f'hipcc failed on {src}:\n{out}\n{err}')
        objs.append(str(obj))
    link = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(LIB_PATH)] + objs + ['-ldl']
    if verbose:
        print(' '.join(link))
    res = subprocess.run(link, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f'hipcc link failed:\n{res.stdout}\n{res.stderr}')
    return LIB_PATH


if __name__ == '__main__':
    print(build_library(force=True, verbose=True))
