"""W-vector block Lanczos algebra (scintools_amd/csrc/blockw.hpp, compiled for the HOST) against its
NumPy specification tools/models/blockw_reference.py.  CPU only; the kernels that use the algebra are
opt-in, run on the host interpreter (tests/test_emu_cpu.py) and have an opt-in GPU test."""
import ctypes
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "models"))
import blockw_reference as ref  # noqa: E402


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("bw") / "libbw.so")
    subprocess.run([hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950",
                    os.path.join(ROOT, "tests", "tools", "blockw_host_check.hip"), "-o", out], check=True,
                   stderr=subprocess.DEVNULL)
    L = ctypes.CDLL(out)
    L.bw_count_c.restype = ctypes.c_int
    L.bw_invit_c.restype = ctypes.c_double
    return L


def dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def pack(M, W):
    """Hermitian / upper-triangular W x W -> packed W*W doubles (diagonal, then strict upper (re, im))."""
    out = [M[r, r].real for r in range(W)]
    for r in range(W):
        for c in range(r + 1, W):
            out += [M[r, c].real, M[r, c].imag]
    return np.array(out)


def cview(a):
    return a.view(np.complex128)


@pytest.mark.parametrize("W", [2, 3, 4])
def test_step_coefficients_and_rows(lib, W):
    rng = np.random.default_rng(W)
    n = 50
    Q, _ = np.linalg.qr(rng.standard_normal((n, W)) + 1j * rng.standard_normal((n, W)))
    Wj = rng.standard_normal((n, W)) + 1j * rng.standard_normal((n, W))
    A = Q.conj().T @ Wj
    A = (A + A.conj().T) / 2          # the device sums are of a Hermitian quantity
    Wj = Wj - Q @ (Q.conj().T @ Wj) + Q @ A
    G = Wj.conj().T @ Wj
    sa, sg = pack(A, W), pack(G, W)
    a = np.zeros(2 * W * W); b = np.zeros(2 * W * W); inv = np.zeros(W)
    lib.bw_from_sums_c(W, dp(sa), dp(sg), dp(a), dp(b), dp(inv))
    Ar, Br, invr = ref.step_block(A, G)
    np.testing.assert_allclose(cview(a).reshape(W, W), Ar, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(cview(b).reshape(W, W), Br, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(inv, invr, rtol=1e-11)
    # B^H B reproduces G - A^H A, and the new block is orthonormal
    Bc = cview(b).reshape(W, W)
    np.testing.assert_allclose(Bc.conj().T @ Bc, G - A.conj().T @ A, rtol=1e-10, atol=1e-11)
    X = np.zeros((n, W), complex)
    for r in range(n):
        u = np.ascontiguousarray(Wj[r]).view(np.float64).copy()
        q = np.ascontiguousarray(Q[r]).view(np.float64).copy()
        x = np.zeros(2 * W); h = np.zeros(2 * W)
        lib.bw_q_row_c(W, dp(sa), dp(sg), dp(u), dp(q), dp(x), dp(h))
        X[r] = cview(x)
        np.testing.assert_allclose(cview(x), ref.q_row(Ar, Br, invr, Wj[r], Q[r]), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(cview(h), Q[r] @ Br.conj().T, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(X.conj().T @ X, np.eye(W), atol=1e-9)
    np.testing.assert_allclose(Q.conj().T @ X, 0, atol=1e-9)


def test_exhausted_direction_gives_zero_pivot(lib):
    W = 3
    A = np.diag([1.0, 2.0, 3.0]).astype(complex)
    G = A.conj().T @ A + np.diag([0.5, 0.0, 0.25])      # second direction has no new component
    a = np.zeros(2 * W * W); b = np.zeros(2 * W * W); inv = np.zeros(W)
    lib.bw_from_sums_c(W, dp(pack(A, W)), dp(pack(G, W)), dp(a), dp(b), dp(inv))
    assert inv[1] == 0.0 and inv[0] > 0 and inv[2] > 0


def test_numerically_dependent_column_is_dropped(lib):
    """A duplicated column of W_j leaves a pivot of pure rounding noise (either sign): the relative
    floor must drop it, or its normalised column would be amplified noise in Q_{j+1}."""
    W, n = 4, 60
    rng = np.random.default_rng(7)
    Q, _ = np.linalg.qr(rng.standard_normal((n, W)) + 1j * rng.standard_normal((n, W)))
    Wj = rng.standard_normal((n, W)) + 1j * rng.standard_normal((n, W))
    Wj[:, 2] = Wj[:, 0] * (0.3 - 0.4j) + Wj[:, 1] * 1.7            # in the span of the first two
    A = Q.conj().T @ Wj
    G = Wj.conj().T @ Wj
    a = np.zeros(2 * W * W); b = np.zeros(2 * W * W); inv = np.zeros(W)
    lib.bw_from_sums_c(W, dp(pack((A + A.conj().T) / 2, W)), dp(pack(G, W)), dp(a), dp(b), dp(inv))
    assert inv[2] == 0.0 and inv[0] > 0 and inv[1] > 0 and inv[3] > 0
    Ar, Br, invr = ref.step_block((A + A.conj().T) / 2, G)
    assert invr[2] == 0.0
    np.testing.assert_allclose(inv, invr, rtol=1e-9)


@pytest.mark.parametrize("W", [2, 3, 4])
def test_band_assembly_counts_and_inverse_iteration(lib, W):
    rng = np.random.default_rng(10 + W)
    nblk = 7
    As = []
    for _ in range(nblk):
        M = rng.standard_normal((W, W)) + 1j * rng.standard_normal((W, W))
        As.append((M + M.conj().T) / 2)
    Bs = [np.triu(rng.standard_normal((W, W)) + 1j * rng.standard_normal((W, W))) for _ in range(nblk - 1)]
    for B in Bs:
        B[np.diag_indices(W)] = np.abs(B[np.diag_indices(W)].real) + 0.1      # Cholesky factors: real positive pivots
    pa = np.concatenate([pack(A, W) for A in As])
    pb = np.concatenate([pack(B, W) for B in Bs] + [np.zeros(W * W)])
    n = W * nblk
    band = np.zeros(2 * n * (W + 1))
    lib.bw_band_c(W, dp(pa), dp(pb), nblk, dp(band))
    band_ref, T = ref.band_from_blocks(As, Bs, W)
    np.testing.assert_array_equal(cview(band).reshape(n, W + 1), band_ref)
    ev = np.linalg.eigvalsh(T)
    for x in list(ev[:-1] + np.diff(ev) / 2) + [ev[0] - 1.0, ev[-1] + 1.0]:
        assert lib.bw_count_c(W, dp(band), n, ctypes.c_double(x), ctypes.c_double(1e-300)) == int((ev < x).sum())
    s = np.zeros(2 * n)
    work = np.zeros(n + 1 + 2 * n * W + 8)
    sigma = ev[-1] + 8e-16 * abs(ev[-1])
    nrm = lib.bw_invit_c(W, dp(band), n, ctypes.c_double(sigma), ctypes.c_double(1e-300), dp(s), dp(work))
    v = cview(s) / np.sqrt(nrm)
    w, U = np.linalg.eigh(T)
    assert 1 - abs(np.vdot(U[:, -1], v)) <= 1e-12
    np.testing.assert_allclose(np.linalg.norm(T @ v - ev[-1] * v), 0, atol=1e-10 * abs(ev[-1]))
