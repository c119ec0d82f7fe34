"""Repository rules that keep the parity claims honest (checked on CPU)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py_files(sub):
    for d, _, files in os.walk(os.path.join(ROOT, sub)):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                yield os.path.join(d, f)


def test_product_never_touches_the_oracle():
    pat = re.compile(r'^\s*(from|import)\s+oracle\b', re.M)
    for path in _py_files('tfep_amd'):
        assert not pat.search(open(path).read()), f'{path} imports the oracle'


def test_oracle_is_only_used_by_tests_smoke_and_cpu_baseline():
    bench = open(os.path.join(ROOT, 'bench.py')).read()
    for m in re.finditer(r'^\s*(from|import)\s+oracle\b.*$', bench, re.M):
        fn = re.findall(r'^def (\w+)', bench[:m.start()], re.M)[-1]
        assert fn == 'cpu_baseline', f'bench.py imports the oracle in {fn}()'
    entry = open(os.path.join(ROOT, '__graft_entry__.py')).read()
    for m in re.finditer(r'^\s*(from|import)\s+oracle\b.*$', entry, re.M):
        fn = re.findall(r'^def (\w+)', entry[:m.start()], re.M)[-1]
        assert fn == 'smoke'


def test_nothing_reads_the_reference_at_run_time():
    """/root/reference does not exist on the GPU box: only tools/ (dev-only generators) may name it."""
    for sub in ('tfep_amd', 'oracle', 'tests'):
        for path in _py_files(sub):
            if path.endswith('test_layout.py'):
                continue
            assert '/root/reference' not in open(path).read(), path
    for f in ('bench.py', '__graft_entry__.py'):
        assert '/root/reference' not in open(os.path.join(ROOT, f)).read(), f


def test_no_compat_layers_in_kernels():
    for path in _py_files(os.path.join('tfep_amd', 'csrc')):
        src = open(path).read()
        assert '__HIP_PLATFORM_AMD__' not in src and 'cuda_runtime' not in src and 'hipify' not in src.lower(), path
