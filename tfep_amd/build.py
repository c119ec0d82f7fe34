"""Build libtfep_hip.so (gfx950) in-tree with hipcc.  ``python -m tfep_amd.build``.

Safe under the one-process-per-GPU launch: the whole build runs under an exclusive file lock, objects are only
recompiled when their source (or any header) is newer, and the library is linked to a temporary name and moved into
place with ``os.replace`` -- a concurrent ``dlopen`` sees the old file or the new one, never a half-written one.
"""
import fcntl
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_DIR = os.path.join(HERE, 'lib')
LIB_PATH = os.path.join(LIB_DIR, 'libtfep_hip.so')
SOURCES = ['transformers.hip', 'masked_linear.hip', 'split_gemm.hip', 'split_gemm_layouts.hip', 'inverse_block.hip',
           'reduce.hip', 'backward.hip', 'egnn.hip', 'maf_layer.hip']
# Per-file flags.  egnn.hip: the edge kernels are bound by vector-instruction issue, and on gfx950 a packed fp32
# instruction (v_pk_fma_f32 ...) is no cheaper than the two scalar ones it replaces (the fp32 vector peak is reached
# without packing; beside MFMAs a packed op costs more) -- keep clang's SLP vectoriser from forming them.
EXTRA_FLAGS = {'egnn.hip': ['-fno-slp-vectorize']}


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def _headers():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    deps.append(os.path.join(os.path.dirname(HERE), 'include', 'tfep_hip.h'))
    return deps


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES] + _headers()
    return any(os.path.getmtime(d) > t for d in deps)


def build_probe(out_path, extra_flags, verbose=True):
    """A separate library with extra compiler flags (timing probes such as -DTFEP_PROBE_KWINDOW=128), built into its
    own directory; select it with TFEP_HIP_LIB=<out_path>.  The shipped library is not touched."""
    import tempfile
    tmp = tempfile.mkdtemp(prefix='tfep_probe_')
    objs, procs = [], []
    for src in SOURCES:
        obj = os.path.join(tmp, src.replace('.hip', '.o'))
        objs.append(obj)
        cmd = [_hipcc(), '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-fno-gpu-rdc', '-Wno-unused-result',
               *EXTRA_FLAGS.get(src, []), *extra_flags, '-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f'hipcc failed on {src}')
    subprocess.check_call([_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out_path] + objs)
    return out_path


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 into one shared library."""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    with open(os.path.join(LIB_DIR, '.build.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():      # another process built it while this one waited for the lock
                return LIB_PATH
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    t_hdr = max(os.path.getmtime(h) for h in _headers())
    objs, procs = [], []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(LIB_DIR, src.replace('.hip', '.o'))
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), t_hdr):
            continue
        tmp = obj + f'.{os.getpid()}.tmp'
        cmd = [_hipcc(), '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-fno-gpu-rdc',
               '-Wno-unused-result', *EXTRA_FLAGS.get(src, []), '-c', path, '-o', tmp]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, tmp, obj, subprocess.Popen(cmd)))
    failed = []
    for src, tmp, obj, p in procs:
        if p.wait() != 0:
            failed.append(src)
            if os.path.exists(tmp):
                os.remove(tmp)
        else:
            os.replace(tmp, obj)
    if failed:
        raise RuntimeError(f'hipcc failed on {", ".join(failed)}')
    tmp = LIB_PATH + f'.{os.getpid()}.tmp'
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == '__main__':
    if '--probe' in sys.argv:            # python -m tfep_amd.build --probe /tmp/libprobe.so -DTFEP_PROBE_KWINDOW=128
        i = sys.argv.index('--probe')
        print(build_probe(sys.argv[i + 1], sys.argv[i + 2:]))
    else:
        build(force='--force' in sys.argv)
        print(LIB_PATH)
