"""Build libdcsnet_hip.so (the C-ABI library of include/dcsnet_hip.h) for gfx950 with hipcc.

In-tree build: objects under dcs-net_amd/build/, library at dcs-net_amd/lib/libdcsnet_hip.so
(git-ignored, but it travels to the GPU box with the gpurun snapshot).  hipcc cross-compiles
without a GPU.  Usage: python dcs-net_amd/build.py [--force]
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libdcsnet_hip.so')
ARCH = 'gfx950'
FLAGS = ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-Wall', '-Wno-unused-function']
FLAGS += os.environ.get('DCS_EXTRA_HIPCC_FLAGS', '').split()      # diagnostic builds (e.g. -DDCS_PIPE_DIAG)


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _stamp(paths):
    h = hashlib.sha256()
    for p in paths:
        with open(p, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = _sources()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(HERE, '..', 'include', 'dcsnet_hip.h'))
    stamp_file = os.path.join(OBJ, 'stamp.txt')
    stamp = _stamp([os.path.join(CSRC, s) for s in srcs] + sorted(headers))
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return LIB

    def compile_one(src):
        obj = os.path.join(OBJ, src[:-4] + '.o')
        cmd = [hipcc] + FLAGS + ['-c', os.path.join(CSRC, src), '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{r.stdout}\n{r.stderr}')
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [hipcc, '-shared', '-fPIC', f'--offload-arch={ARCH}', '-o', LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    with open(stamp_file, 'w') as f:
        f.write(stamp)
    if verbose:
        print(f'built {LIB}')
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
