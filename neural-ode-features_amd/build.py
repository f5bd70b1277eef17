"""Build recipe for libnode_hip.so (gfx950 only, in-tree).

    python neural-ode-features_amd/build.py [--force] [--diag]

`--diag` also links libnode_hip_diag.so: the same objects with kernels_w4.hip compiled -DNODE_DIAG, i.e. WITH the timing
ablations (results wrong by design), the in-kernel stamps and the measured-and-rejected kernel variants that tools/ and the
`-m diag` tests use (NODE_HIP_DIAG=1 makes _lib.load() take it).  The product library contains none of them.

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
LIB_DIAG = os.path.join(CSRC, 'libnode_hip_diag.so')
DIAG_SOURCES = ['kernels_w4.hip', 'kernels_tiny_solve.hip']       # translation units that hold `#ifdef NODE_DIAG` code
SOURCES = ['kernels_layout.hip', 'kernels_pointwise.hip', 'kernels_conv.hip', 'kernels_wgrad.hip', 'kernels_head.hip', 'kernels_loss.hip', 'kernels_optim.hip', 'kernels_w4.hip', 'kernels_w4s.hip', 'kernels_stem.hip', 'kernels_tiny.hip', 'kernels_tiny_solve.hip', 'stem_api.hip', 'node_api.hip']
HEADERS = [os.path.join(CSRC, 'node_internal.h'), os.path.join(CSRC, 'wino4.h'), os.path.join(CSRC, 'stem.h'), os.path.join(CSRC, 'step_control.h'), os.path.join(ROOT, 'include', 'node_hip.h')]
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function',
         '-ffp-contract=fast']


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, diag=False):
    obj = os.path.join(CSRC, src.replace('.hip', '.diag.o' if diag else '.o'))
    path = os.path.join(CSRC, src)
    if _stale(obj, [path] + HEADERS):
        cmd = [HIPCC] + FLAGS + (['-DNODE_DIAG'] if diag else []) + ['-c', path, '-o', obj]
        print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return obj


def _link(lib, objs, verbose):
    tmp = lib + '.tmp'
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, lib)


def build(force=False, verbose=True, diag=False):
    """Compile every HIP translation unit for gfx950 and link libnode_hip.so (and, with `diag`, libnode_hip_diag.so)."""
    if force:
        for s in SOURCES:
            for suffix in ('.o', '.diag.o'):
                o = os.path.join(CSRC, s.replace('.hip', suffix))
                if os.path.exists(o):
                    os.remove(o)
        for lib in (LIB, LIB_DIAG):
            if os.path.exists(lib):
                os.remove(lib)
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if _stale(LIB, objs):
        # the guard runs BEFORE the link and a stale library is removed first: a violation never leaves a libnode_hip.so
        # behind that a later, non-forced build() or load would take as up to date
        if os.path.exists(LIB):
            os.remove(LIB)
        _check_asm_loads(verbose)
        _link(LIB, objs, verbose)
    if diag:
        dobjs = [_compile(s, diag=True) if s in DIAG_SOURCES else o for s, o in zip(SOURCES, objs)]
        if _stale(LIB_DIAG, dobjs):
            _link(LIB_DIAG, dobjs, verbose)
    return LIB


def _check_asm_loads(verbose):
    """Before a (re)link: the in-flight-register guard for the kernels with inline-asm loads (tools/check_asm_loads.py,
    which compiles with this module's FLAGS)."""
    tool = os.path.join(os.path.dirname(HERE), 'tools', 'check_asm_loads.py')
    if not os.path.exists(tool):
        return
    r = subprocess.run([sys.executable, tool], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout.strip(), flush=True)
    if r.returncode:
        raise RuntimeError('inline-asm load guard failed:\n' + r.stdout + r.stderr)


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, diag='--diag' in sys.argv))
