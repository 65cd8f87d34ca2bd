"""Time the secondary-spectrum and conjugate-spectrum FFT paths on the GPU (hipEvents)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scintools_amd import ththmod as thth
from scintools_amd.dynspec import sspec_device
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
# usage: time_fft.py [size ...] [op ...]   ops: sspec prewhite cs cs3   (default: everything)
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [2048, 4096, 8192]
ops = [a for a in sys.argv[1:] if not a.isdigit()] or ['sspec', 'prewhite', 'cs', 'cs3']
for size in sizes:
    x = torch.randn(size, size, dtype=torch.float64, device='cuda')
    R = C = 2 * size
    alg = 8 * size * size + 8 * (R // 2) * C
    if 'sspec' in ops:
        ms = t(lambda: sspec_device(x))
        print(f'sspec {size}^2: {ms:.3f} ms  algorithmic {alg/1e6:.0f} MB -> {alg/ms/1e6:.0f} GB/s')
    if 'prewhite' in ops:
        ms = t(lambda: sspec_device(x, prewhite=True))
        print(f'sspec prewhite {size}^2: {ms:.3f} ms -> {alg/ms/1e6:.0f} GB/s')
    if 'cs' in ops:
        ms = t(lambda: thth.conjugate_spectrum(x, 0, pad_value=0.0))
        alg = 8 * size * size + 16 * size * size
        print(f'CS npad=0 {size}^2: {ms:.3f} ms  algorithmic {alg/1e6:.0f} MB -> {alg/ms/1e6:.0f} GB/s')
    if size <= 4096 and 'cs3' in ops:
        ms = t(lambda: thth.conjugate_spectrum(x, 3, pad_value=0.0), n=3)
        alg = 8 * size * size + 16 * 16 * size * size
        print(f'CS npad=3 {size}^2 -> {4*size}^2: {ms:.3f} ms  algorithmic {alg/1e6:.0f} MB -> {alg/ms/1e6:.0f} GB/s')
