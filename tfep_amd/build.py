"""Build libtfep_hip.so (gfx950) in-tree with hipcc.  ``python -m tfep_amd.build``."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_DIR = os.path.join(HERE, 'lib')
LIB_PATH = os.path.join(LIB_DIR, 'libtfep_hip.so')
SOURCES = ['transformers.hip', 'masked_linear.hip', 'split_gemm.hip', 'inverse_block.hip', 'reduce.hip', 'backward.hip']


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), 'include', 'tfep_hip.h'))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 into one shared library."""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, src.replace('.hip', '.o'))
        cmd = [_hipcc(), '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-fno-gpu-rdc',
               '-Wno-unused-result', '-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f'hipcc failed on {src}')
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB_PATH)
