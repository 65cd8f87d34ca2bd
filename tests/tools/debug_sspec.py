import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes
from oracle import sspec_oracle as so
from scintools_amd.dynspec import sspec_device
from scintools_amd import _lib
from scintools_amd.device import to_device, ptr, stream_ptr
g = np.load('tests/golden/sim_sspec.npz')
dyn = g['dyn']
lib = _lib.load()
m = ctypes.c_double()
dt = to_device(dyn, torch.float64)
lib.scint_mean(ptr(dt), dyn.size, ctypes.byref(m), stream_ptr())
print('mean gpu', m.value, 'np', dyn.mean(), 'diff', m.value - dyn.mean())
for kw in [dict(window=None), dict(), dict(window=None, halve=False)]:
    sec = sspec_device(dt, **kw).cpu().numpy()
    ref = so.calc_sspec(dyn, 30., 1., **kw)[2]
    lin, lref = 10**(sec/10), 10**(ref/10)
    i = np.unravel_index(np.abs(lin-lref).argmax(), lin.shape)
    print(kw, 'max abs lin err', np.abs(lin-lref).max(), 'at', i, lref[i], 'peak', lref.max(),
          'max rel', np.nanmax(np.abs(lin-lref)/lref), 'median rel', np.nanmedian(np.abs(lin-lref)/lref))
# random data
rng = np.random.default_rng(0)
x = rng.standard_normal((96,128))
sec = sspec_device(to_device(x, torch.float64), window=None).cpu().numpy()
ref = so.calc_sspec(x, 30., 1., window=None)[2]
lin, lref = 10**(sec/10), 10**(ref/10)
print('random: max rel', np.max(np.abs(lin-lref)/lref), 'median rel', np.median(np.abs(lin-lref)/lref))
x = rng.standard_normal((96,128)) + 100.0
sec = sspec_device(to_device(x, torch.float64), window=None).cpu().numpy()
ref = so.calc_sspec(x, 30., 1., window=None)[2]
lin, lref = 10**(sec/10), 10**(ref/10)
print('random+100: max rel', np.max(np.abs(lin-lref)/lref), 'median rel', np.median(np.abs(lin-lref)/lref))
