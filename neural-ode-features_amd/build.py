"""Build recipe for libnode_hip.so (gfx950 only, in-tree).

    python neural-ode-features_amd/build.py [--force]

hipcc cross-compiles for gfx950 without a GPU; the built .so stays next to the
sources (git-ignored, but it travels to the GPU box with the gpurun snapshot).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
ROOT = os.path.dirname(HERE)
LIB = os.path.join(CSRC, 'libnode_hip.so')
SOURCES = ['kernels_layout.hip', 'kernels_pointwise.hip', 'kernels_conv.hip', 'kernels_wgrad.hip', 'kernels_head.hip', 'kernels_optim.hip', 'kernels_w4.hip', 'kernels_w4s.hip', 'kernels_stem.hip', 'stem_api.hip', 'node_api.hip']
HEADERS = [os.path.join(CSRC, 'node_internal.h'), os.path.join(CSRC, 'wino4.h'), os.path.join(CSRC, 'stem.h'), os.path.join(ROOT, 'include', 'node_hip.h')]
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function',
         '-ffp-contract=fast']


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(CSRC, src.replace('.hip', '.o'))
    path = os.path.join(CSRC, src)
    if _stale(obj, [path] + HEADERS):
        cmd = [HIPCC] + FLAGS + ['-c', path, '-o', obj]
        print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return obj


def build(force=False, verbose=True):
    """Compile every HIP translation unit for gfx950 and link libnode_hip.so."""
    if force:
        for s in SOURCES:
            o = os.path.join(CSRC, s.replace('.hip', '.o'))
            if os.path.exists(o):
                os.remove(o)
        if os.path.exists(LIB):
            os.remove(LIB)
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if _stale(LIB, objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        _check_asm_loads(verbose)
    return LIB


def _check_asm_loads(verbose):
    """After a relink: the in-flight-register guard for the kernels with inline-asm loads (tools/check_asm_loads.py)."""
    tool = os.path.join(os.path.dirname(HERE), 'tools', 'check_asm_loads.py')
    if not os.path.exists(tool):
        return
    r = subprocess.run([sys.executable, tool], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout.strip(), flush=True)
    if r.returncode:
        raise RuntimeError('inline-asm load guard failed:\n' + r.stdout + r.stderr)


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
