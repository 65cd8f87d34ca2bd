"""Context for DESIGN.md section 9: what the vendor FFT library (rocFFT through torch.fft) takes for the transforms of calc_sspec 4096^2
(padded to 8192 x 8192, kept half 4096 x 8192) on the same GPU.  Development tool: nothing in the product uses torch.fft or rocFFT."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scintools_amd.dynspec import sspec_device
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x = torch.randn(size, size, dtype=torch.float64, device='cuda')
print(f'calc_sspec (this library) {size}^2: {t(lambda: sspec_device(x)):.3f} ms')
print(f'torch.fft.rfft2(x, s=(2n, 2n)) [rocFFT, transform only]: {t(lambda: torch.fft.rfft2(x, s=(2*size, 2*size))):.3f} ms')
print(f'torch.fft.fft2(x, s=(2n, 2n)) complex out: {t(lambda: torch.fft.fft2(x, s=(2*size, 2*size)), n=5):.3f} ms')
y = torch.randn(size, size, dtype=torch.complex128, device='cuda')
print(f'row pass only: torch.fft.fft(y[{size} rows x {size}], n=2n, dim=1): {t(lambda: torch.fft.fft(y, n=2*size, dim=1)):.3f} ms')
print(f'column pass only: torch.fft.fft(y, n=2n, dim=0): {t(lambda: torch.fft.fft(y, n=2*size, dim=0)):.3f} ms')
