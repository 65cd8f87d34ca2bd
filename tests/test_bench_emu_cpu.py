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
    # round 5: the legs that describe what a user's observation and a rank of --shard eta get
    sim = d["simulation_screen"]
    assert "error" not in sim, sim
    assert sim["value"] > 0 and sim["failed_etas"] == 0 and sum(sim["passes_per_eta_histogram"].values()) == 6
    assert sim["ratio_to_headline_input"] == pytest.approx(sim["value"] / d["value"])
    pred = d["config"]["predicted_strong_scaling"]
    assert "error" not in pred, pred
    assert set(pred) - {"T1_ms", "note"} == {"2"} and pred["2"]["efficiency"] > 0          # 6 curvatures: only two ranks get >= 2 each
    assert len(d["library"]["csrc_sha256"]) == 64
    assert d["roofline"]["traffic"] is None or "csrc_sha256" not in (d["roofline"]["traffic_note"] or "")
    # round 6: the figures a reader of the END of the line needs are flat scalars after every nested object
    keys = list(d)
    tail = keys[keys.index("tail"):]
    for k in ("matvec_frac_of_hbm_peak", "gather_frac_in_sweep", "gather_frac_alone", "sim_screen_eta_per_s", "sim_screen_passes",
              "modeler_eta_per_s", "mixed_eta_per_s", "strong_scaling_pred_8", "strong_scaling_measured", "sspec_ms",
              "workload_fit_thetatheta_s", "workload_wavefield_s", "workload_tutorial_fit_s", "workload_fit_arc_s", "eta_per_s"):
        assert k in tail, k
    assert keys[-1] == "eta_per_s" and d["eta_per_s"] == d["value"]
    assert d["sim_screen_eta_per_s"] == sim["value"] and d["gather_frac_in_sweep"] == d["gather"]["frac"]
    assert d["mixed_eta_per_s"] == mp["value"] and d["modeler_eta_per_s"] is None            # (--modeler-steps 0 here)
    assert "skipped" in d["workloads"]["note"] and d["workload_wavefield_s"] is None          # (128^2: fewer than 4 x 4 chunks)


@pytest.mark.timeout(900)
def test_precision_mixed_line():
    d = _line(["--precision", "mixed"])
    assert d["config"]["sweep_precision"] == "mixed" and d["config"]["failed_etas"] == 0
    r = d["roofline"]
    assert r["kernel"] == "pk2_matvec32_kernel" and r["achieved"] > 0 and r["traffic"] is None
    assert r["mixed"]["certified_per_step"] == 6 and r["algorithmic_bytes_per_step"] == r["mixed"]["complex64_bytes_per_step"]
    assert "mixed_precision" not in d and "one_slot_group" not in r


@pytest.mark.timeout(900)
def test_modeler_leg_has_its_own_roofline():
    """The `modeler` object of the line: rate, a roofline of its own whose parts add up, and the
    mixed-all leg against the float64 chi^2 curve."""
    args = SMALL[:-2] + ["--modeler-steps", "1", "--mixed-steps", "0"]
    try:
        sys.path.insert(0, os.path.join(REPO, "tests", "emu"))
        import emulated
        emulated.load()
    except (RuntimeError, OSError, subprocess.CalledProcessError) as exc:
        pytest.skip(f"host interpreter could not be built: {exc}")
    out = subprocess.run([sys.executable, PROBE] + args, capture_output=True, text=True, timeout=800, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    m = d["modeler"]
    r = m["roofline"]
    assert m["value"] > 0 and m["failed_etas"] == 0 and r["bound"] == "hbm" and 0 < r["frac"] == r["achieved"] / r["peak"]
    parts = r["algorithmic_bytes_per_eta_by_part"]
    assert abs(sum(parts.values()) - r["algorithmic_bytes_per_eta"]) <= 1e-9 * r["algorithmic_bytes_per_eta"]
    chi_step = "chi^2 step (edge terms + final sum beside the fused back-map; chisq_parseval_batch_kernel when the image is written; model transform + sink when cropped or masked)"
    if m["chisq_route"]["from_back_map_accumulators"]:      # (the probe's axes are fft_axis of even lengths: the fused route of round 6)
        assert parts["back_map_reads_fft2_dspec"] == 16.0 * 128 * 128 and parts["chisq_step"] == 0.0
        assert r["parts"][chi_step]["launches_per_step"] > 0 and r["parts"][chi_step]["algorithmic_bytes_per_step"] == 0.0
    else:
        assert parts["back_map_write"] == 16.0 * 128 * 128 and parts["model_read_plus_dspec"] == 24.0 * 128 * 128
        assert r["parts"][chi_step]["launches_per_step"] > 0 and r["parts"][chi_step]["achieved"] > 0
    assert m["chisq_route"]["curvatures_redone_from_a_written_image"] >= 0
    for k in ("pk2_matvec_kernel", "back-map (rev_diag_batch_kernel; rank-1)"):
        assert r["parts"][k]["launches_per_step"] > 0 and r["parts"][k]["achieved"] > 0, k
    mx = m["mixed_all"]
    assert "error" not in mx, mx
    assert mx["failed_etas"] == 0 and mx["max_rel_diff_vs_f64_chisq_curve"] < 1e-9 and mx["value"] > 0


@pytest.mark.timeout(900)
@pytest.mark.parametrize("extra", [[], ["--shard", "eta", "--precision", "mixed"]])
def test_two_ranks(extra):
    """bench.py as the driver launches it for N = 2 (one process per rank, RANK / WORLD_SIZE / MASTER_* in the environment;
    gloo here, the ranks' "GPUs" are the interpreter): the weak-scaling default, and ONE observation's curvatures in blocks
    over the ranks with the mixed sweep -- whose gathered curve must equal rank 0's own sweep of all curvatures, bit for bit."""
    import socket
    try:
        sys.path.insert(0, os.path.join(REPO, "tests", "emu"))
        import emulated
        emulated.load()
    except (RuntimeError, OSError, subprocess.CalledProcessError) as exc:
        pytest.skip(f"host interpreter could not be built: {exc}")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    args = ["--gpus", "2", "--size", "128", "--neta", "7", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
            "--modeler-steps", "0"] + extra
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, PROBE] + args, env=env, cwd=REPO, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]      # rank 0 prints the line
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 2 and c["ranks_seen"] == 2 and c["backend"] == "gloo" and c["failed_etas"] == 0
    assert c["per_rank_eta_per_s"]["min"] > 0
    if extra:
        assert d["scaling"] == "strong" and c["sweep_precision"] == "mixed" and c["gathered_equals_one_gpu"] is True
        assert d["roofline"]["kernel"] == "pk2_matvec32_kernel"
        assert "strong" not in d
    else:
        assert d["scaling"] == "weak" and c["sweep_precision"] == "f64" and "mixed_precision" not in d
        # round 6: the plain N-rank command also yields a strong-scaling point (the same ranks split observation 0's curvatures)
        st = d["strong"]
        assert st["scaling"] == "strong" and st["ranks_seen"] == 2 and st["backend"] == "gloo" and st["gathered_equals_one_gpu"] is True
        assert st["value"] > 0 and st["efficiency"] == pytest.approx(st["T1_ms"] / (2 * st["ms_per_step"]))
        assert d["strong_scaling_measured"] == st["efficiency"]
