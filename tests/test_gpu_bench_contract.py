"""The driver-facing contract: bench.py prints one JSON line with the agreed fields (plus the
roofline and cpu_baseline objects), and __graft_entry__.smoke() passes.  Small sizes: this is a
contract check, not a measurement."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--size", "512", "--neta", "16",
                          "--steps", "2", "--warmup", "1", "--cpu-sample", "2", "--cpu-reps", "2", "--cpu-pool", "0"],
                         capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["scaling"] == "weak"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 16 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"]
    assert "workload" in d["config"] and d["config"]["failed_etas"] == 0
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and 0 < r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    one = r["one_slot_group"]            # the mat-vec with the GPU to itself (same sweep, one slot group)
    assert one["unit"] == "GB/s" and 0 < one["frac"] == pytest.approx(one["achieved"] / r["peak"]) and one["eta_per_s"] > 0
    mp = d["mixed_precision"]            # the mixed-precision sweep on the same workload: a reported leg, never `value`
    assert d["config"]["sweep_precision"] == "f64"
    assert mp["failed_etas"] == 0 and mp["max_rel_diff_vs_f64_curve"] < 1e-11 and mp["value"] > 0
    assert mp["certified_per_step"] == 16 and mp["certificate_passes_mean"] >= 1.0
    assert mp["complex64_bytes_per_step"] > 0 and mp["complex128_bytes_per_step"] > 0
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["max_rel_diff_vs_gpu"] < 1e-9
    # round 6: flat scalars after every nested object (what the driver's tail of the line shows), the value last
    keys = list(d)
    tail = keys[keys.index("tail"):]
    for k in ("matvec_frac_of_hbm_peak", "gather_frac_in_sweep", "gather_frac_alone", "sim_screen_eta_per_s", "sim_screen_passes",
              "modeler_eta_per_s", "mixed_eta_per_s", "strong_scaling_pred_8", "sspec_ms", "sspec_frac_of_hbm_peak",
              "workload_fit_thetatheta_s", "workload_wavefield_s", "workload_tutorial_fit_s", "workload_fit_arc_s",
              "cpu_baseline_eta_per_s", "eta_per_s"):
        assert k in tail, k
    assert keys[-1] == "eta_per_s" and d["eta_per_s"] == d["value"] and d["modeler_eta_per_s"] == d["modeler"]["value"]
    assert d["gather_frac_alone"] == d["gather"]["one_slot_group"]["frac"] and d["sspec_ms"] == d["sspec"]["512x512"]["ms"]


def test_bench_default_line_times_the_workloads():
    """At a size with 4 x 4 chunks the default line also carries `workloads`: Dynspec.fit_thetatheta, calc_wavefield, the tutorial
    recipe and fit_arc timed on the GPU alone (their CPU samples stay behind --workload X)."""
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--size", "1024", "--neta", "16", "--steps", "1", "--warmup", "1",
                          "--no-cpu-baseline", "--modeler-steps", "0", "--mixed-steps", "0", "--sim-steps", "0", "--share-steps", "0",
                          "--workload-steps", "1"], capture_output=True, text=True, timeout=900, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    w = d["workloads"]
    for name in ("fit_thetatheta", "wavefield", "tutorial_fit", "fit_arc"):
        assert "error" not in w[name], w[name]
        assert w[name]["seconds"] > 0 and d[f"workload_{name}_s"] == w[name]["seconds"]
    assert w["tutorial_fit"]["parity"]["max_rel_diff_eta_evo_vs_reference_run"] < 1e-6


def test_graft_entry_smoke():
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"],
                         capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "smoke ok" in out.stdout
