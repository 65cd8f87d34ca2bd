"""CPU model of the sweep's block Lanczos: matrix passes per curvature against the block size.

    python tools/models/block_lanczos_passes.py <size> [eta-index stride]

Same start vectors (rows of theta-theta around n/2), same recurrence and the same stopping rule
(err = resid^2 / (theta1 - theta2) <= tol |theta1| and a settled theta1) as csrc/eigen_packed.hip,
in NumPy, on the bench workload (scintools_amd.synth.arc_dynspec, 256 curvatures).  Development
tool behind DESIGN.md section 9 item 1; uses the oracle to build the matrices, so it is not part
of the product.
"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import thth_oracle as O
from scintools_amd.synth import arc_dynspec
from scintools_amd.ththmod import fft_axis
size = int(sys.argv[1]); neta = 256
if "--sim" in sys.argv:        # a reference Simulation screen (oracle/sim_oracle.py) instead of the analytic arc: round 5, VERDICT r4 next 3
    sys.argv.remove("--sim")
    from oracle import sim_oracle
    sim = sim_oracle.baseline_dynspec(size, 3)      # (serial: this script has no __main__ guard for spawned workers)
    dyn, freqs, times, eta_true = np.array(sim.dyn, dtype=float), sim.freqs, sim.times, float(sim.eta)
else:
    dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=3, nimg=64)
dyn -= dyn.mean()
fd = fft_axis(times, 1000.0, 0); tau = fft_axis(freqs, 1.0, 0)
edges = np.linspace(-fd.max() / 2, fd.max() / 2, size)
etas = np.geomspace(0.25, 4.0, neta) * eta_true
CS = O.conjugate_spectrum(dyn, 0)
if isinstance(CS, tuple): CS = CS[0]
def block_lanczos(A, X0, tol=1e-12, maxit=300):
    b = X0.shape[1]
    Q, _ = np.linalg.qr(X0)
    Qs = [Q]; As = []; Bs = []
    Qprev = np.zeros_like(Q); Bprev = np.zeros((b, b), complex)
    prev = -np.inf
    for k in range(1, maxit + 1):
        W = A @ Qs[-1] - Qprev @ Bprev.conj().T
        Ak = Qs[-1].conj().T @ W
        W = W - Qs[-1] @ Ak
        As.append((Ak + Ak.conj().T) / 2)
        Qn, Bk = np.linalg.qr(W)
        m = b * k
        T = np.zeros((m, m), complex)
        for j in range(k):
            T[b*j:b*j+b, b*j:b*j+b] = As[j]
            if j < k - 1:
                T[b*j+b:b*j+2*b, b*j:b*j+b] = Bs[j]
                T[b*j:b*j+b, b*j+b:b*j+2*b] = Bs[j].conj().T
        w, S = np.linalg.eigh(T)
        th1, th2 = w[-1], (w[-2] if m > 1 else -np.inf)
        s = S[:, -1]
        resid = np.linalg.norm(Bk @ s[-b:])
        at = abs(th1)
        err = resid**2 / max(th1 - th2, 1e-300)
        settled = (th1 - prev) <= 1e3 * tol * at
        if k >= 3 and err <= tol * at and settled:
            return th1, k
        prev = th1
        Bs.append(Bk); Qprev = Qs[-1]; Bprev = Bk; Qs.append(Qn)
    return th1, maxit
tot = {1: 0, 2: 0, 3: 0, 4: 0, 6: 0, 8: 0}
bytes_rel = {b: 0.0 for b in tot}
idx = list(range(4, 256, int(sys.argv[2]) if len(sys.argv) > 2 else 17))
for i in idx:
    A, _ = O.thth_redmap(CS, tau, fd, etas[i], edges)
    n = A.shape[0]
    rows = [n // 2, n // 2 + 7, n // 2 - 7, n // 2 + 14, n//2 - 14, n//2 + 21, n//2-21, n//2+28]
    out = []
    for b in tot:
        X0 = np.stack([A[r % n, :].conj() for r in rows[:b]], axis=1)
        th, k = block_lanczos(A, X0)
        tot[b] += k; out.append((b, k)); bytes_rel[b] += k * n * (n + 1.0)
    print(i, n, out, flush=True)
print("passes per curvature", {b: round(tot[b] / len(idx), 2) for b in tot})
print("matrix bytes relative to two vectors", {b: round(bytes_rel[b] / bytes_rel[2], 3) for b in tot})
