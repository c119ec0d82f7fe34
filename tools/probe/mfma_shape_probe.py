#!/usr/bin/env python
"""Sustained fp16 MFMA throughput, clock and socket power of the 16x16x32 and 32x32x16 shapes (see mfma_shape_probe.hip).
Build here (`python tools/probe/mfma_shape_probe.py --build`, hipcc cross-compiles), run on an MI355X."""
import ctypes
import os
import re
import subprocess
import sys
import threading
import time

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, 'libmfma_shape_probe.so')


def build():
    subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-shared', '-fPIC', '-Wno-unused-result', '-Wno-unused-value',
                    os.path.join(HERE, 'mfma_shape_probe.hip'), '-o', SO], check=True)


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(['/opt/rocm/bin/rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True,
                                 timeout=10).stdout
            sclk = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', txt)
            pw = re.search(r'Power \(W\): ([\d.]+)', txt)
            if sclk and pw:
                out.append((int(sclk.group(1)), float(pw.group(1))))
        except Exception:
            pass
        time.sleep(0.3)


def main():
    if '--build' in sys.argv:
        build()
        return
    lib = ctypes.CDLL(SO)
    lib.probe_run.restype = ctypes.c_float
    lib.probe_run.argtypes = [ctypes.c_int] * 5
    lib.probe_flops.restype = ctypes.c_double
    lib.probe_flops.argtypes = [ctypes.c_int] * 2
    blocks, iters, launches = 256, 400000, 12         # one workgroup (4 waves, 1 per SIMD) per CU; ~0.3 s per launch
    for zero in (0, 1):
        for shape in (16, 32):
            lib.probe_run(shape, zero, blocks, 200, 1)
            stop, samples = threading.Event(), []
            th = threading.Thread(target=sample, args=(stop, samples))
            th.start()
            ms = lib.probe_run(shape, zero, blocks, iters, launches)
            stop.set()
            th.join()
            s = samples[1:] or samples
            sclk = sum(a for a, _ in s) / max(len(s), 1)
            pw = sum(b for _, b in s) / max(len(s), 1)
            tf = lib.probe_flops(blocks, iters) * launches / (ms * 1e-3) / 1e12
            print(f'{"zero" if zero else "random"} operands, v_mfma_f32_{"16x16x32" if shape == 16 else "32x32x16"}_f16: {tf:7.0f} TFLOP/s f16 '
                  f'({ms / launches:.1f} ms per launch); rocm-smi over {len(s)} samples: sclk {sclk:.0f} MHz, {pw:.0f} W', flush=True)


if __name__ == '__main__':
    main()
