"""CPU study for DESIGN.md section 9, item 1: Krylov iterations on a complex64 copy of the
theta-theta matrix, certified by one fp64 Rayleigh quotient.

    python tests/tools/mixed_precision_study.py [size] [neta]

For a few curvatures of the synthetic arc workload it prints the Lanczos steps and the relative
eigenvalue error against ARPACK (the reference's eigsh) for
  (a) the fp64 matrix (what the shipped sweep does),
  (b) the complex64-rounded matrix alone,
  (c) (b) followed by rho = x^H A x on the fp64 matrix, with the bound |r|^2 / gap.
Uses the oracle for the matrices -- test infrastructure, not product code."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
from scipy.sparse.linalg import eigsh  # noqa: E402

from oracle import thth_oracle as to  # noqa: E402
from scintools_amd.synth import arc_dynspec  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
neta = int(sys.argv[2]) if len(sys.argv) > 2 else 6
TOL = 1e-12


def lanczos(matvec, v0, tol, max_steps=300, check_every=4, first_check=8):
    """Hermitian Lanczos without re-orthogonalisation and the sweep's a-posteriori stop:
    err = min(resid, resid^2 / (theta1 - theta2)) <= tol * |theta1|.  Returns (theta, steps, x, gap)."""
    n = len(v0)
    q_prev = np.zeros(n, complex)
    q = v0 / np.linalg.norm(v0)
    Q, alpha, beta = [q], [], [0.0]
    b = 0.0
    for k in range(1, max_steps + 1):
        u = matvec(q) - b * q_prev
        a = np.vdot(q, u).real
        u = u - a * q
        b = np.linalg.norm(u)
        alpha.append(a)
        if k >= first_check and k % check_every == 0:
            T = np.diag(alpha) + np.diag(beta[1:], 1) + np.diag(beta[1:], -1)
            w, y = np.linalg.eigh(T)
            resid = abs(b * y[-1, -1])
            gap = w[-1] - w[-2]
            if min(resid, resid**2 / gap) <= tol * abs(w[-1]):
                x = np.array(Q).T @ y[:, -1]
                return w[-1], k, x / np.linalg.norm(x), gap
        beta.append(b)
        q_prev, q = q, u / b
        Q.append(q)
    raise RuntimeError("no convergence")


dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=3, nimg=64)
dyn = dyn - dyn.mean()
fd = to.fft_axis(times, 1000.0, 0)
tau = to.fft_axis(freqs, 1.0, 0)
CS = np.fft.fftshift(np.fft.fft2(dyn))
edges = np.linspace(-fd.max() / 2, fd.max() / 2, size)
print(f"# {size}x{size}, N up to {size - 1}, tol {TOL}")
print("# eta/eta_true  N   steps64 err64      steps32 err32(alone)  err32+rayleigh  bound |r|^2/gap/lam")
for eta in np.geomspace(0.5, 2.0, neta) * eta_true:
    A, _ = to.thth_redmap(CS, tau, fd, eta, edges)
    n = A.shape[0]
    v0 = A[n // 2, :].copy()
    lam = eigsh(A, 1, which="LA", v0=v0 / np.linalg.norm(v0))[0][0]
    th64, k64, _, _ = lanczos(lambda v: A @ v, v0, TOL)
    A32 = A.astype(np.complex64)
    th32, k32, x, gap = lanczos(lambda v: A32 @ v, v0, TOL)      # fp64 vectors, complex64 matrix entries
    y = A @ x
    rho = np.vdot(x, y).real
    r = np.linalg.norm(y - rho * x)
    print(f"{eta / eta_true:8.3f} {n:5d}   {k64:4d}  {abs(th64 - lam) / lam:9.2e}   {k32:4d}  "
          f"{abs(th32 - lam) / lam:9.2e}     {abs(rho - lam) / lam:9.2e}      {r * r / gap / lam:9.2e}")
