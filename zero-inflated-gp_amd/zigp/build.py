"""Build libzigp.so (HIP, gfx950) in-tree: zero-inflated-gp_amd/lib/libzigp.so."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'csrc')
LIBDIR = os.path.join(ROOT, 'lib')
LIB = os.path.join(LIBDIR, 'libzigp.so')
SOURCES = ['zigp_lib.hip']
HEADERS = ['zigp_dense.hip', 'zigp_kron.hip', 'zigp_gemm.h', 'zigp_ctx.h', 'zigp_kernels.h', 'zigp_host.h', '../../include/zigp.h']


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: cannot build libzigp.so')
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [hipcc, '-O3'] + os.environ.get('ZIGP_EXTRA_FLAGS', '').split() + [ '-DZIGP_NSTAGE=%d' % int(os.environ.get('ZIGP_NSTAGE', '2')), '-DZIGP_WAVES_DEFAULT=%d' % int(os.environ.get('ZIGP_WAVES_DEFAULT', '4')), '--offload-arch=gfx950', '-std=c++17', '-shared', '-fPIC', '-o', LIB] + srcs
    if verbose:
        print(' '.join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + r.stdout + r.stderr)
    return LIB


def ensure():
    """Build only when the library file is absent (a git checkout; a gpurun snapshot carries the built file)."""
    return LIB if os.path.exists(LIB) else build(force=True)


if __name__ == '__main__':
    print(build(force=True, verbose=True))
