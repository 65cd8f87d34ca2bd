"""Experiment (round 6): shader clocks per phase of sspec_rows2_kernel, from the kernel's own clock reads.
    SCINT_SSPEC_ABL=32 python tools/experiments/rows2_phase_times.py [size]
The kernel (abl & 32) overwrites the first elements of its output with eight per-workgroup sums (thread 0, all its rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scintools_amd.dynspec import sspec_device
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x = torch.randn(size, size, dtype=torch.float64, device='cuda')
sspec_device(x); torch.cuda.synchronize()
sec = sspec_device(x); torch.cuda.synchronize()
g = 512 if size == 4096 else 256
t = sec.flatten()[: g * 8].reshape(g, 8).cpu().numpy()
names = ["wait for the row", "issue stores", "even transform", "even logs", "odd input + prefetch issue", "odd transform", "odd logs", "-"]
if int(os.environ.get("SCINT_SSPEC_ABL", "0")) & 64:     # the column kernel's clocks (the row kernel does not run)
    names = ["wait for the pair + window", "even input", "even transform", "even separation + stores", "odd input + prefetch issue", "odd transform", "odd separation + stores", "-"]
rows = (size // g) if not int(os.environ.get("SCINT_SSPEC_ABL", "0")) & 64 else size // 2 // g
tot = t.sum(axis=1)
print(f"size {size}: workgroups {g}, rows per workgroup {rows}; clocks per row (mean over workgroups, min..max):")
for k, nme in enumerate(names[:7]):
    c = t[:, k] / rows
    print(f"  {nme:28s} {c.mean():9.0f}  ({c.min():7.0f} .. {c.max():7.0f})  {100 * t[:, k].sum() / tot.sum():5.1f} %")
print(f"  total per row {tot.mean() / rows:9.0f}")
