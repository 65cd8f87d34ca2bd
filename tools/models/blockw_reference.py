"""Executable specification of the W-vector block Lanczos step algebra, written the way the device
code computes it (the specification the opt-in wide-block kernels are checked against; not part of the product).

Conventions (W = block width; every matrix below is W x W complex unless noted):
  W_j   = A Q_j - Q_{j-1} B_{j-1}^H            (N x W; the mat-vec + the reduce kernel)
  A_j   = Q_j^H W_j                            (Hermitian; from fixed-order partial sums)
  G_j   = W_j^H W_j                            (Gram matrix; same partial sums)
  G'_j  = G_j - A_j^H A_j = B_j^H B_j          (B_j upper triangular: Cholesky factor)
  Q_{j+1} = (W_j - Q_j A_j) B_j^{-1}           (row by row, forward substitution over columns)
  T     = block tridiagonal, T[j][j] = A_j, T[j+1][j] = B_j  -> Hermitian band matrix, half width W

A direction whose pivot is not positive (the Krylov space is exhausted in it) gets a zero row in
B_j and a zero column in Q_{j+1} (inverse pivot 0), as the two-vector device code does.
"""
import numpy as np


PIVOT_FLOOR = 1e-13      # kBwPivotFloor of blockw.hpp


def step_block(Asum, Gsum, room=None):
    """From the summed partials A = Q^H W and G = W^H W: (A, B upper triangular, inverse pivots).
    A pivot at the rounding level of its Gram entry (PIVOT_FLOOR, relative) is a direction that is
    exhausted; so is every direction beyond `room`, the dimensions the Krylov space has left (n minus
    the rank of the earlier blocks)."""
    W = Asum.shape[0]
    A = Asum.copy()
    H = Gsum - A.conj().T @ A
    B = np.zeros((W, W), complex)
    inv = np.zeros(W)
    room = W if room is None else room
    for c in range(W):                         # Cholesky, H = B^H B, row c of B at a time
        d = H[c, c].real - sum(abs(B[m, c]) ** 2 for m in range(c))
        keep = room > 0 and d > PIVOT_FLOOR * Gsum[c, c].real
        room -= keep
        B[c, c] = np.sqrt(d) if keep else 0.0
        inv[c] = 1.0 / B[c, c].real if keep else 0.0
        for j in range(c + 1, W):
            s = H[c, j] - sum(np.conj(B[m, c]) * B[m, j] for m in range(c))
            B[c, j] = s * inv[c]
    return A, B, inv


def q_row(A, B, inv, u, q):
    """Row of Q_{j+1} from the rows u of W_j and q of Q_j."""
    W = len(u)
    y = u - q @ A
    x = np.zeros(W, complex)
    for c in range(W):
        x[c] = (y[c] - sum(x[m] * B[m, c] for m in range(c))) * inv[c]
    return x


def band_from_blocks(As, Bs, W):
    """band[i][k] = T[i + k][i], k = 0..W, for the block tridiagonal T (len(As) blocks)."""
    k = len(As)
    n = W * k
    T = np.zeros((n, n), complex)
    for j in range(k):
        T[W * j:W * j + W, W * j:W * j + W] = As[j]
        if j + 1 < k:
            T[W * j + W:W * j + 2 * W, W * j:W * j + W] = Bs[j]
            T[W * j:W * j + W, W * j + W:W * j + 2 * W] = Bs[j].conj().T
    band = np.zeros((n, W + 1), complex)
    for i in range(n):
        for d in range(W + 1):
            if i + d < n:
                band[i, d] = T[i + d, i]
    return band, T


def band_factor(band, x, tiny):
    """LDL^H of T - x for a Hermitian band matrix (half width W): pivots d and M = L D below the
    diagonal, M[i][k-1] = M_{i+k, i}.  Returns (count of negative pivots, d, M)."""
    n, W1 = band.shape
    W = W1 - 1
    d = np.zeros(n)
    M = np.zeros((n, W), complex)
    cnt = 0
    for i in range(n):
        # d_i = T_ii - x - sum_m |M_{i,i-m}|^2 / d_{i-m}
        di = band[i, 0].real - x
        for m in range(1, W + 1):
            if i - m >= 0:
                di -= abs(M[i - m, m - 1]) ** 2 / d[i - m]
        if abs(di) < tiny:
            di = -tiny
        d[i] = di
        cnt += di < 0.0
        # M_{i+k,i} = T_{i+k,i} - sum_{m>=1, m+k<=W} M_{i+k,i-m} conj(M_{i,i-m}) / d_{i-m}
        for k in range(1, W + 1):
            if i + k >= n:
                break
            v = band[i, k]
            for m in range(1, W - k + 1):
                if i - m >= 0:
                    v -= M[i - m, m + k - 1] * np.conj(M[i - m, m - 1]) / d[i - m]
            M[i, k - 1] = v
    return cnt, d, M


def band_count(band, x, tiny):
    return band_factor(band, x, tiny)[0]


def band_inverse_iteration(band, sigma, tiny, iters=2):
    """Eigenvector of T nearest sigma: solves with the LDL^H factor of T - sigma."""
    n, W1 = band.shape
    W = W1 - 1
    _, d, M = band_factor(band, sigma, tiny)
    s = np.ones(n, complex)
    for _ in range(iters):
        for i in range(n):                       # L y = s, L_{i,i-m} = M_{i,i-m} / d_{i-m}
            y = s[i]
            for m in range(1, W + 1):
                if i - m >= 0:
                    y -= M[i - m, m - 1] * s[i - m] / d[i - m]
            s[i] = y
        s = s / d
        for i in range(n - 1, -1, -1):           # L^H z = y
            z = s[i]
            for m in range(1, W + 1):
                if i + m < n:
                    z -= np.conj(M[i, m - 1]) * s[i + m] / d[i]
            s[i] = z
        s = s / np.abs(s).max()
    return s / np.linalg.norm(s)


def lanczos(A, X0, tol=1e-12, first_check=4, check_every=3, max_steps=128):
    """The sweep's recurrence + stopping rule with the algebra above.  Returns (theta, passes, ritz vector)."""
    n, W = X0.shape
    # step 0: Q_0 from the start block by the same Cholesky route (A-part zero)
    Az = np.zeros((W, W), complex)
    A0, B0, inv0 = step_block(Az, X0.conj().T @ X0)
    Q = np.stack([q_row(A0, B0, inv0, X0[r], np.zeros(W, complex)) for r in range(n)])
    Qprev = np.zeros_like(Q)
    Bprev = np.zeros((W, W), complex)
    As, Bs, Qs = [], [], [Q]
    prev = -np.inf
    for k in range(1, max_steps + 1):
        Wj = A @ Q - Qprev @ Bprev.conj().T
        Aj, Bj, invj = step_block(Q.conj().T @ Wj, Wj.conj().T @ Wj)
        Aj = (Aj + Aj.conj().T) / 2
        As.append(Aj)
        if k >= first_check and (k - first_check) % check_every == 0 or k == max_steps:
            band, T = band_from_blocks(As, Bs, W)
            nn = band.shape[0]
            scale = np.abs(T).sum(axis=1).max()
            tiny = scale * 1e-300 + 1e-300
            lo, hi = -scale * 1.001, scale * 1.001
            theta = bisect(band, nn, lo, hi, tiny)
            theta2 = bisect(band, nn - 1, lo, theta, tiny)
            s = band_inverse_iteration(band, theta + 8e-16 * max(abs(theta), scale * 1e-3), tiny)
            resid = np.linalg.norm(Bj @ s[-W:])
            gap = theta - theta2
            err = resid * resid / gap if gap > resid else resid
            settled = (theta - prev) <= 1e3 * tol * abs(theta)
            if (err <= tol * abs(theta) and settled) or k == max_steps:
                V = np.concatenate(Qs, axis=1) @ s
                return theta, k, V / np.linalg.norm(V)
            prev = theta
        Qn = np.stack([q_row(Aj, Bj, invj, Wj[r], Q[r]) for r in range(n)])
        Bs.append(Bj)
        Qprev, Bprev, Q = Q, Bj, Qn
        Qs.append(Q)
    raise RuntimeError("not reached")


def bisect(band, target, lo, hi, tiny):
    """Smallest x with count(T < x) >= target, i.e. the target-th eigenvalue (1-based from below)."""
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if band_count(band, mid, tiny) >= target:
            hi = mid
        else:
            lo = mid
        if hi - lo <= 2e-16 * max(abs(lo), abs(hi)):
            break
    return 0.5 * (lo + hi)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for W in (1, 2, 3, 4):
        for n in (40, 130):
            Mx = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
            Amat = Mx + Mx.conj().T
            np.fill_diagonal(Amat, 0.0)
            # a dominant pair so that the run converges in few passes
            v = rng.standard_normal(n) + 1j * rng.standard_normal(n)
            Amat = Amat + 3.0 * np.outer(v, v.conj())
            rows = [(n // 2 + 7 * t) % n for t in range(W)]
            X0 = np.stack([Amat[r].conj() for r in rows], axis=1)
            theta, k, V = lanczos(Amat, X0)
            w, U = np.linalg.eigh(Amat)
            print(f"W={W} n={n}: passes {k:3d}  rel err {abs(theta - w[-1]) / abs(w[-1]):.1e}  "
                  f"1-|<V,u>| {1 - abs(np.vdot(U[:, -1], V)):.1e}")
        # band factor against dense inertia
        As = [np.diag(rng.standard_normal(W)).astype(complex) for _ in range(6)]
        Bs = [np.triu(rng.standard_normal((W, W)) + 1j * rng.standard_normal((W, W))) for _ in range(5)]
        band, T = band_from_blocks(As, Bs, W)
        ev = np.linalg.eigvalsh(T)
        for x in (ev[2] + 1e-3, ev[-1] - 1e-3, 0.0):
            assert band_count(band, x, 1e-300) == int((ev < x).sum()), (W, x)
    print("band counts agree with dense inertia")
