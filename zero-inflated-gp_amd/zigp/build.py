"""Build libzigp.so (HIP, gfx950) in-tree: zero-inflated-gp_amd/lib/libzigp.so."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'csrc')
LIBDIR = os.path.join(ROOT, 'lib')
LIB = os.path.join(LIBDIR, 'libzigp.so')
SOURCES = ['zigp_lib.hip']
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(('.h', '.hip')) and f not in SOURCES) + ['../../include/zigp.h', '../../include/zigp_diag.h']


HASHFILE = LIB + '.srchash'


DENSE_FILES = ['zigp_gemm.h', 'zigp_kernels.h', 'zigp_host.h', 'zigp_ctx.h', 'zigp_dense.hip']   # what the cfg3 step's kernels are built from


def source_hash(files=None):
    """sha256 over every file the library is built from (file mtimes do not survive a snapshot copy; contents do);
    files=DENSE_FILES: the dense path only (provenance of the counter summaries under profiles/)"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(files if files is not None else SOURCES + HEADERS):
        p = os.path.join(CSRC, f)
        if os.path.exists(p):
            h.update(f.encode())
            h.update(open(p, 'rb').read())
    for k in ('ZIGP_EXTRA_FLAGS',):
        h.update(('%s=%s' % (k, os.environ.get(k, ''))).encode())
    return h.hexdigest()


def needs_build():
    """True when the library is missing or was built from other sources than the ones on disk"""
    if not os.path.exists(LIB) or not os.path.exists(HASHFILE):
        return True
    return open(HASHFILE).read().strip() != source_hash()


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: cannot build libzigp.so')
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [hipcc, '-O3'] + os.environ.get('ZIGP_EXTRA_FLAGS', '').split() + ['--offload-arch=gfx950', '-std=c++17', '-shared', '-fPIC', '-o', LIB] + srcs
    if verbose:
        print(' '.join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + r.stdout + r.stderr)
    with open(HASHFILE, 'w') as f:
        f.write(source_hash() + '\n')
    return LIB


def ensure():
    """Make sure lib/libzigp.so matches the sources on disk: builds when the file is absent (a git checkout) or stale (an edit
    to csrc/ or include/zigp.h after the last build); a gpurun snapshot carries the built file and its source hash."""
    return build(force=False)


if __name__ == '__main__':
    print(build(force=True, verbose=True))
