"""TEST INFRASTRUCTURE ONLY: back-map (rev_map) outputs of the host-interpreted kernels for a fixed set of
grids -- rank-1 and explicit, Hermitian or not, uniform and irregular theta grids, several delay slabs,
curvatures that push pairs off the delay axis -- saved to argv[1] (.npz).  Used to check that a rewrite of
rev_gather_kernel leaves every bit of the image unchanged (the sums are order-independent by construction):
run it before and after and compare the files (python tests/emu/revmap_probe.py --compare a.npz b.npz).
tests/test_emu_cpu.py::test_back_map_bits_are_pinned holds the SHA-256 of every image (tests/golden/revmap_bits.json,
written by `python tests/emu/revmap_probe.py tests/golden/revmap_bits.json` from the round-2 kernel and unchanged by
the round-3 rewrite)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)


def cases():
    rng = np.random.default_rng(11)
    out = []
    # (name, ntau, nfd, nedge, eta factor, irregular grid, edge span as a fraction of fd.max / 2)
    for name, ntau, nfd, nedge, ef, irregular, span in (
            ("small", 96, 80, 100, 1.0, False, 1.0),
            ("two_slabs", 1100, 64, 130, 1.0, False, 1.0),
            ("three_chunks", 200, 150, 601, 0.7, False, 1.0),
            ("off_axis", 128, 128, 300, 6.0, False, 1.0),
            ("flat", 128, 128, 300, 0.05, False, 1.0),
            ("irregular", 160, 120, 280, 1.0, True, 1.0),
            ("wide_bins", 300, 40, 400, 1.0, False, 0.9),
            ("sparse_theta", 256, 256, 60, 1.0, False, 1.0),
            ("chunks_and_slabs", 2200, 48, 1200, 1.0, False, 1.0),
            ("chunks_and_slabs_steep", 2500, 40, 1100, 3.0, True, 1.0),
            # round 4: odd axis lengths, three delay slabs on an irregular grid, axes that are NOT symmetric about 0
            ("odd_axes", 97, 81, 120, 1.0, False, 1.0),
            ("odd_axes_slabs", 1301, 37, 500, 1.3, True, 1.0),
            ("shifted_axes", 96, 80, 100, 1.0, False, 1.0)):
        tau = (np.arange(ntau) - ntau // 2) * 0.0137
        fd = (np.arange(nfd) - nfd // 2) * 0.211
        if name == "shifted_axes":
            tau = tau + 0.3 * 0.0137
            fd = fd - 0.4 * 0.211
        edges = np.linspace(-span * fd.max() / 2, span * fd.max() / 2, nedge)
        if irregular:
            edges = np.sort(edges + rng.uniform(-0.3, 0.3, nedge) * (edges[1] - edges[0]))
        eta = ef * np.abs(tau).max() / (fd.max() / 2) ** 2
        out.append((name, tau, fd, edges, eta))
    return out


def images(ththmod):
    """Every case's back-map image through the (already interpreted) ththmod wrappers."""
    import torch
    res = {}
    rng = np.random.default_rng(5)
    for name, tau, fd, edges, eta in cases():
        grid = ththmod._Grid(tau, fd, edges)
        n = grid.M
        th_t = ththmod.to_device(grid.th_cents, torch.float64)
        v = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.exp(-np.linspace(-2, 2, n) ** 2)
        w = np.array([-3.7 if name == "flat" else 2.9])
        rec = ththmod._rev_map_dev(grid.geom, th_t, n, eta, True, vec_t=ththmod.to_device(v),
                                   w_t=ththmod.to_device(w, torch.float64))
        res[name + "_rank1"] = rec.cpu().numpy()
        if n <= 300:
            a = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
            for tag, mat, herm in (("herm", a + a.conj().T, True), ("plain", a, False)):
                rec = ththmod._rev_map_dev(grid.geom, th_t, n, eta, herm, thth_t=ththmod.to_device(mat))
                res[f"{name}_{tag}"] = rec.cpu().numpy()
    return res


def digests(res):
    import hashlib
    return {k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() for k, v in sorted(res.items())}


def main(out_path):
    from _pytest.monkeypatch import MonkeyPatch
    import emulated
    patch = MonkeyPatch()
    emulated.install(patch)
    from scintools_amd import ththmod
    res = images(ththmod)
    patch.undo()
    if out_path.endswith(".json"):          # the pinned digests of tests/golden/revmap_bits.json
        import json
        with open(out_path, "w") as fh:
            json.dump(digests(res), fh, indent=1)
    else:
        np.savez(out_path, **res)


def compare(a_path, b_path):
    a, b = np.load(a_path), np.load(b_path)
    assert set(a.files) == set(b.files), (a.files, b.files)
    bad = [k for k in a.files if not np.array_equal(a[k], b[k], equal_nan=True)]
    for k in a.files:
        nz = np.count_nonzero(np.nan_to_num(a[k]))
        print(f"{k:24s} {a[k].shape}  nonzero {nz:8d}  {'DIFFERENT' if k in bad else 'same bits'}")
    return 1 if bad else 0


if __name__ == "__main__":
    if sys.argv[1] == "--compare":
        sys.exit(compare(sys.argv[2], sys.argv[3]))
    main(sys.argv[1])
