"""Host-side tooling that the numbers in DESIGN.md / profiles/ go through (CPU only)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)


def test_interval_union_and_co_residency():
    """tools/rocpd_summary.py: the union of launch intervals (what bytes / time must use when two
    streams run the same kernel) and the time spent at each overlap depth."""
    from rocpd_summary import union_and_depth
    # two streams: [0,10) and [5,15) overlap on [5,10); a third launch [20,30) is alone
    union, hist = union_and_depth([(0, 10), (5, 15), (20, 30)])
    assert union == 25
    assert hist == {1: 20, 2: 5}
    # back-to-back launches on one stream: union == plain sum
    spans = [(i * 10, i * 10 + 10) for i in range(7)]
    union, hist = union_and_depth(spans)
    assert union == 70 and hist == {1: 70}
    # nested launch
    union, hist = union_and_depth([(0, 100), (10, 20)])
    assert union == 100 and hist == {1: 90, 2: 10}


def test_bench_pool_size_is_bounded_by_memory_and_cores():
    import bench
    n = bench.pool_size(-1, 4096)
    assert 1 <= n <= 64
    assert bench.pool_size(5, 4096) == 5          # an explicit request is honoured
    # a bigger problem never gets more workers than a smaller one
    assert bench.pool_size(-1, 16384) <= bench.pool_size(-1, 1024)


def test_bench_workload_is_the_baseline_config():
    import bench
    dyn, freqs, times, fd, tau, edges, etas, eta_true = bench.make_workload(256, 32, 256, seed=3)
    assert dyn.shape == (256, 256) and abs(dyn.mean()) < 1e-9      # mean-subtracted chunk (dynspec.py:1692)
    assert edges.shape == (256,) and np.allclose(edges, -edges[::-1])
    assert etas.shape == (32,) and np.isclose(etas[0], 0.25 * eta_true) and np.isclose(etas[-1], 4.0 * eta_true)
    assert np.all(np.diff(fd) > 0) and np.all(np.diff(tau) > 0)


def test_experiment_patches_still_apply():
    """tools/experiments/*.patch are variants queued for the next GPU call (tools/build_variant.sh): each must still apply to
    the tree it is meant to be measured against."""
    import glob
    import shutil
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which("git") or not os.path.isdir(os.path.join(repo, ".git")):
        pytest.skip("not a git checkout")
    patches = sorted(glob.glob(os.path.join(repo, "tools", "experiments", "*.patch")))
    assert patches
    for p in patches:
        out = subprocess.run(["git", "apply", "--check", p], cwd=repo, capture_output=True, text=True)
        assert out.returncode == 0, (os.path.basename(p), out.stderr[-500:])


def test_bench_refuses_a_pmc_summary_taken_from_other_kernel_sources(monkeypatch, tmp_path):
    """VERDICT r4 (weak 10): `roofline.traffic` is a committed PMC ratio times live bytes, so a kernel change without a new PMC
    pass carried a stale ratio.  The summaries now record the SHA-256 of the kernel sources they were measured on
    (bench.library_fingerprint, copied by tools/pmc_summary.py from the profiled run's own bench line) and bench.py quotes a
    summary only while the sources hash to that value."""
    import json
    import bench
    fp = bench.library_fingerprint()
    assert len(fp["csrc_sha256"]) == 64 and fp == bench.library_fingerprint()           # deterministic
    prof = tmp_path / "profiles"
    prof.mkdir()
    summ = {"dominant_kernel": "scint::pk2_matvec_kernel", "csrc_sha256": fp["csrc_sha256"],
            "kernels": {"scint::pk2_matvec_kernel": {"traffic_over_algorithmic": 1.04}}}
    (prof / "r99_pmc_summary.json").write_text(json.dumps(summ))
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    monkeypatch.setattr(bench, "library_fingerprint", lambda: fp)
    ratio, src, stale = bench.pmc_traffic_ratio()
    assert ratio == 1.04 and src.endswith("r99_pmc_summary.json") and stale is None
    monkeypatch.setattr(bench, "library_fingerprint", lambda: dict(fp, csrc_sha256="0" * 64))   # the sources changed
    ratio, src, stale = bench.pmc_traffic_ratio()
    assert ratio is None and "other kernel sources" in stale
    (prof / "r99_pmc_summary.json").write_text(json.dumps({k: v for k, v in summ.items() if k != "csrc_sha256"}))   # a pre-round-5 file
    monkeypatch.setattr(bench, "library_fingerprint", lambda: fp)
    assert bench.pmc_traffic_ratio()[0] is None
    assert bench.pmc_modeler_ratio() == (None, None, "no PMC summary committed")
