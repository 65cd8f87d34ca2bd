"""Wall time of Dynspec.thetatheta_chunks on the tutorial data of the parity tests (tests/golden/fit_thetatheta.npz,
7 chunks of 64 x 150): the batched path (one stack, one eigenpair sweep) against the chunk-by-chunk loop the
reference runs (pool.map over single_chunk_retrieval, here a serial `map`), and the difference of their results."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from scintools_amd.dynspec import Dynspec  # noqa: E402

g = np.load(os.path.join(REPO, "tests", "golden", "retrieval.npz"))
f = np.load(os.path.join(REPO, "tests", "golden", "fit_thetatheta.npz"))
n = int(g["nchan"])


class B:
    dyn, freqs, times, dt, df = f["dspec"][:n], f["freq"][:n], f["time"], float(f["dt"]), float(f["df"])


class SerialPool:
    def map(self, fn, it):
        return [fn(x) for x in it]


d = Dynspec(dyn=B(), verbose=False)
d.prep_thetatheta(cwf=64, edges_lim=.3, eta_min=30, eta_max=50, nedge=128)
d.fit_thetatheta()
res = {}
for name, pool in (("batched", None), ("chunk by chunk", SerialPool()), ("batched", None), ("chunk by chunk", SerialPool())):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d.thetatheta_chunks(pool=pool)
    torch.cuda.synchronize()
    res.setdefault(name, []).append(time.perf_counter() - t0)
    res[name + " chunks"] = d.chunks.copy()
a, b = res["batched chunks"], res["chunk by chunk chunks"]
dev = max(np.abs(x * np.exp(-1j * np.angle(np.vdot(y, x))) - y).max() / np.abs(y).max() for x, y in zip(a[:, 0], b[:, 0]))
print(f"thetatheta_chunks, {a.shape[0]}x{a.shape[1]} chunks of {a.shape[2]}x{a.shape[3]}: batched {min(res['batched'])*1e3:.1f} ms, "
      f"chunk by chunk {min(res['chunk by chunk'])*1e3:.1f} ms; max deviation up to the per-chunk phase {dev:.2e}")
