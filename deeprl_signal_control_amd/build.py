"""Builds the C-ABI library (deeprl_signal_control_amd/libtsc.so) with hipcc for gfx950.

In-tree on purpose: the built .so is git-ignored but travels to the GPU box with the repo
snapshot.  -ffp-contract=off keeps one rounding per fp32 operation in the microsimulator
(bit-exact vehicle state vs the CPU oracle); the MFMA kernels do not depend on contraction."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libtsc.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off',
         '-fno-fast-math', '-Wall', '-Wno-unused-function']


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, '..', 'include', 'tsc.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compiles every .hip source (one hipcc process per file, side by side) and links libtsc.so.  force=True always
    recompiles from source -- what __graft_entry__.build() does, so that a box that runs the tests provably built what
    it tests; force=False reuses a library that is newer than every source (the import-time path of _lib.py)."""
    if not force and not stale():
        return LIB
    extra = os.environ.get('TSC_BUILD_DEFS', '').split()          # measurement builds, e.g. TSC_BUILD_DEFS=-DTSC_STREAM_SC1=1
    # object files of THIS build only (two concurrent force builds -- ranks, pytest workers -- wrote the same
    # _obj/<name>.o and could link a torn object): a per-process directory, removed after linking
    import shutil
    import tempfile
    os.makedirs(os.path.join(HERE, '_obj'), exist_ok=True)
    objdir = tempfile.mkdtemp(prefix='build%d_' % os.getpid(), dir=os.path.join(HERE, '_obj'))
    try:
        return _build_in(objdir, extra, verbose)
    finally:
        shutil.rmtree(objdir, ignore_errors=True)


def _build_in(objdir, extra, verbose):
    cflags = [f for f in FLAGS if f != '-shared'] + extra + ['-I', os.path.join(HERE, '..', 'include')]
    jobs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + '.o')
        cmd = [HIPCC] + cflags + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        jobs.append((cmd, obj, subprocess.Popen(cmd)))
    for cmd, _, p in jobs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    tmp = LIB + '.tmp%d' % os.getpid()
    link = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp] + [o for _, o, _ in jobs]
    if verbose:
        print(' '.join(link), file=sys.stderr)
    subprocess.check_call(link)
    os.replace(tmp, LIB)                                         # a concurrent reader never sees a half-written library
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True)
    print(LIB)
