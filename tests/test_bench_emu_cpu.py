"""bench.py's own Python paths on the host interpreter (tests/emu/bench_probe.py), tiny sizes: the JSON contract of the
default line incl. its mixed-precision leg, and of `--precision mixed`.  The rates are the interpreter's and mean nothing;
tests/test_gpu_bench_contract.py is the same contract on the GPU."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(REPO, "tests", "emu", "bench_probe.py")
SMALL = ["--size", "128", "--neta", "6", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--modeler-steps", "0"]


def _line(extra):
    try:
        sys.path.insert(0, os.path.join(REPO, "tests", "emu"))
        import emulated
        emulated.load()
    except (RuntimeError, OSError, subprocess.CalledProcessError) as exc:
        pytest.skip(f"host interpreter could not be built: {exc}")
    out = subprocess.run([sys.executable, PROBE] + SMALL + extra, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_default_line_with_its_mixed_leg():
    d = _line(["--mixed-steps", "1"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["dtype"] == "f64" and d["config"]["sweep_precision"] == "f64" and d["config"]["failed_etas"] == 0
    assert d["roofline"]["kernel"].startswith("pk2_matvec_kernel") and d["roofline"]["achieved"] > 0
    mp = d["mixed_precision"]
    assert "error" not in mp, mp
    assert mp["failed_etas"] == 0 and mp["max_rel_diff_vs_f64_curve"] < 1e-11 and mp["value"] > 0
    assert mp["certified_per_step"] == 6 and mp["certificate_passes_mean"] >= 1.0
    assert mp["complex64_bytes_per_step"] > 0 and mp["complex128_bytes_per_step"] > 0
    assert mp["matvec32"]["launches"] > 0


@pytest.mark.timeout(900)
def test_precision_mixed_line():
    d = _line(["--precision", "mixed"])
    assert d["config"]["sweep_precision"] == "mixed" and d["config"]["failed_etas"] == 0
    r = d["roofline"]
    assert r["kernel"] == "pk2_matvec32_kernel" and r["achieved"] > 0 and r["traffic"] is None
    assert r["mixed"]["certified_per_step"] == 6 and r["algorithmic_bytes_per_step"] == r["mixed"]["complex64_bytes_per_step"]
    assert "mixed_precision" not in d and "one_slot_group" not in r
