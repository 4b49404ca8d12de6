"""Builds elektronn3_amd/libe3unet.so (hand-written HIP for gfx950) with hipcc, in-tree.

    python -m elektronn3_amd.build        # or: from elektronn3_amd.build import build; build()

The .so is git-ignored but travels with the tree (e.g. to the GPU box); it is rebuilt only when a source is newer.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libe3unet.so')
OBJDIR = os.path.join(HERE, 'build')
SOURCES = ['conv_mfma.hip', 'conv_v3.hip', 'conv_wino.hip', 'conv_wino4.hip', 'conv_wino2d.hip', 'upconv_gemm.hip', 'wgrad_mfma.hip', 'wgrad_wino.hip', 'wgrad_wino2d.hip', 'conv_small.hip', 'elementwise.hip', 'loss.hip', 'optim.hip', 'api.cpp', 'unet_plan.cpp',
           'attention.hip', 'bf16_conv.hip', 'bf16_wgrad.hip', 'bf16_upconv.hip', 'bf16_ew.hip', 'bf16_first.hip', 'api_bf16.cpp', 'unet_bf16.cpp']
# the 16-bit path is compiled a second time for IEEE half (-DE3_F16, external symbols renamed by f16_names.h; see csrc/bf16.h)
F16_SOURCES = ['bf16_conv.hip', 'bf16_wgrad.hip', 'bf16_upconv.hip', 'bf16_ew.hip', 'bf16_first.hip', 'api_bf16.cpp', 'unet_bf16.cpp']
HEADERS = ['common.h', 'kernels.h', 'brick_order.h', 'bf16.h', 'f16_names.h', 'plan_internal.h', os.path.join('..', '..', 'include', 'e3unet.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function', '-x', 'hip'] + os.environ.get('E3_HIPCC_EXTRA', '').split()
# per-file extras.  wgrad_wino: the SLP vectoriser turns its scalar transforms into v_pk_* ops plus ~200 v_mov per brick
# to pair the operands up; packed fp32 is not faster than two scalar ops on gfx950 (7 vs 2 x 4.5 cycles), the moves are pure loss.
EXTRA_FLAGS = {'wgrad_wino.hip': ['-fno-slp-vectorize'], 'wgrad_wino2d.hip': ['-fno-slp-vectorize']}


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: libe3unet.so cannot be built on this machine')
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    if not force and not _stale(OUT, srcs + hdrs + [os.path.abspath(__file__)]):
        return OUT
    os.makedirs(OBJDIR, exist_ok=True)
    # one builder at a time (several ranks of a torch.distributed launch may import the package at once): the others wait for
    # the lock and then find the library fresh
    import fcntl
    with open(os.path.join(OBJDIR, '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale(OUT, srcs + hdrs + [os.path.abspath(__file__)]):
            return OUT
        return _build_locked(srcs, hdrs, force, verbose)


def _build_locked(srcs, hdrs, force, verbose):
    hipcc = _hipcc()

    def compile_one(job):
        src, f16 = job
        obj = os.path.join(OBJDIR, os.path.basename(src) + ('.f16.o' if f16 else '.o'))
        if force or _stale(obj, [src] + hdrs):
            extra = ['-DE3_F16', '-include', os.path.join(CSRC, 'f16_names.h')] if f16 else []
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + extra + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f'hipcc failed on {src}:\n{r.stderr}')
            if verbose and r.stderr.strip():
                print(r.stderr, file=sys.stderr)
        return obj

    jobs = [(src, False) for src in srcs] + [(os.path.join(CSRC, f), True) for f in F16_SOURCES]
    with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        objs = list(ex.map(compile_one, jobs))
    tmp = OUT + f'.tmp{os.getpid()}'
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stderr}')
    os.replace(tmp, OUT)              # atomic: a concurrent dlopen never sees a half-written library
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
