#!/usr/bin/env python
"""Headline benchmark: eta-curvature-sweep points/s on a 4096x4096 dynamic spectrum.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 4096] [--neta 256]

One *step* is one pass of the hot path over one observation per GPU: device-resident
real dynamic spectrum [4096, 4096] -> conjugate spectrum (2-D FFT) -> for each of 256
curvatures: theta-theta gather (nedge=4096) + dominant 'LA' eigenvalue -> eigs[256] on the
host -> parabola fit (the body of ththmod.single_search, ththmod.py:773-859).  This is
BASELINE.json configs[2], the configuration the metric is quoted on.

Multi-GPU: one rank per GPU over RCCL.  The driver launches the ranks with
torch.distributed.run; `python bench.py --gpus N` with no WORLD_SIZE in the environment
launches them itself (same command line).  Three partitionings, none with a data-path collective
(the eigenvalue curves are all-gathered at the end of each step, the only collective):
  default           one observation per GPU (per-GPU work fixed: weak scaling);
  --shard eta       ONE observation, its eta range in contiguous blocks over the ranks
                    (sweep.sharded_eval_sweep; the reference's pool.map pattern, dynspec.py:1706-1723;
                    total work fixed: strong scaling) -- every rank FFTs the same dynspec locally;
  --obs-total T     T observations dealt round-robin (BASELINE config 4: `--size 2048 --obs-total 64`;
                    strong scaling).
`value` is the whole-job eta-points/s.  For N > 1 the line proves itself: `config.ranks_seen` is
dist.get_world_size(), `config.backend` the transport, `config.per_rank_eta_per_s` the slowest / fastest
rank's own rate, and with --shard eta `config.gathered_equals_one_gpu` says whether the gathered curve is
bit-identical to rank 0 sweeping all curvatures alone (checked after the timed region).

Besides the contract fields the JSON line carries
  roofline      the dominant kernel (eigen mat-vec on the Hermitian tile-packed matrix):
                algorithmic bytes 8 N (N+1) per matrix pass per eta -- every upper-triangle
                element once; SURVEY.md 8d's 16 N^2 assumed the full matrix -- summed over
                every (two-vector) Lanczos pass of the timed region, divided by the time that kernel was
                running (hipEvents on the launch streams, union of the launch intervals);
  gather        the same for the theta-theta gather kernel (16 N^2 bytes per eta);
  modeler       (N=1) the OTHER objective of BASELINE configs[2]: the ththmod.modeler /
                chisq_calc sweep over the same 256 curvatures, timed after the headline region,
                with one curvature checked against the oracle's chisq_calc;
  cpu_baseline  (N=1) the NumPy/SciPy oracle (a restatement of the reference) timed on this
                host by SURVEY.md 8d's protocol: the conjugate-spectrum FFT plus a 16-eta subset spanning
                the sweep, median of 3 repetitions, scaled to the 256-eta sweep; and eta-parallel over a
                process pool.  A floor for context, not the target.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def library_fingerprint():
    """What the loaded library was built from: SHA-256 over the kernel sources and the C header (sorted by name), and over
    the shared library file itself.  The PMC summaries under profiles/ record the first (the tools copy it from the bench
    line of the profiled run); a summary taken from other sources is refused below (a kernel change
    without a new PMC pass carried a stale ratio)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(REPO, "scintools_amd", "csrc")
    names = sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".hpp")))
    for path in [os.path.join(csrc, f) for f in names] + [os.path.join(REPO, "include", "scint_hip.h")]:
        with open(path, "rb") as fh:
            h.update(os.path.basename(path).encode() + b"\0" + fh.read())
    so = hashlib.sha256()
    try:
        with open(os.path.join(REPO, "scintools_amd", "libscint_hip.so"), "rb") as fh:
            so.update(fh.read())
        so_hex = so.hexdigest()
    except OSError:
        so_hex = None
    return {"csrc_sha256": h.hexdigest(), "so_sha256": so_hex}


def _newest_summary(pattern, key):
    """(ratio, source, note) from the newest profiles/<pattern> whose recorded source fingerprint is THIS library's."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", pattern)))
    if not files:
        return None, None, "no PMC summary committed"
    mine = library_fingerprint()["csrc_sha256"]
    with open(files[-1]) as fh:
        summ = json.load(fh)
    rel = os.path.relpath(files[-1], REPO)
    theirs = summ.get("csrc_sha256")
    if theirs != mine:
        return None, rel, (f"{rel} was measured on other kernel sources (csrc_sha256 {str(theirs)[:12]} != {mine[:12]}): "
                           "no traffic figure is quoted until `tools/gpu_run.sh pmc` has been re-run on this library")
    return key(summ), rel, None


def pmc_traffic_ratio():
    """HBM bytes / algorithmic bytes of the dominant kernel, from the newest committed PMC
    summary (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same script,
    gfx950 FETCH x2 correction -- tools/pmc_summary.py) -- if it was measured on this library's sources."""
    return _newest_summary("*_pmc_summary.json", lambda summ: summ["kernels"].get(
        summ.get("dominant_kernel", "scint::pk_matvec_kernel"), {}).get("traffic_over_algorithmic"))


def pmc_modeler_ratio():
    """HBM bytes / algorithmic bytes of ONE WHOLE STEP of the modeler / chi^2 objective (every kernel of the step), from the
    newest committed summary of `tools/gpu_run.sh pmc_modeler` (separate FETCH_SIZE / WRITE_SIZE passes of
    `bench.py --objective chisq`, tools/pmc_modeler_summary.py) -- if it was measured on this library's sources."""
    return _newest_summary("*_pmc_modeler_summary.json", lambda summ: summ.get("traffic_over_algorithmic"))


def pmc_modeler_bytes_per_eta():
    """The measured HBM bytes of that summary per curvature (its absolute figure: the line's own algorithmic bytes -- which depend on
    the route chi^2 took -- are divided into it, not the summary's)."""
    def per_eta(summ):
        n = summ.get("steps", 1) * summ.get("curvatures_per_step", 0)
        return summ["hbm_bytes"] / n if n and summ.get("hbm_bytes") else None
    return _newest_summary("*_pmc_modeler_summary.json", per_eta)


def port_vs_reference():
    """Time of the oracle ("port") over the time of the unmodified reference on the same conjugate spectrum, recorded in the
    build container by tests/golden/time_port_vs_reference.py (the bench box has no /root/reference)."""
    try:
        with open(os.path.join(REPO, "tests", "golden", "port_vs_reference_timing.json")) as fh:
            t = json.load(fh)
        return {"port_over_reference_time": t["port_over_reference_time"], "max_rel_diff": t["max_rel_diff"],
                "size": t["size"], "host_cores": t["host_cores"],
                "source": "tests/golden/port_vs_reference_timing.json (tests/golden/time_port_vs_reference.py, build container)"}
    except (OSError, KeyError, ValueError):
        return None


def modeler_objects(mod, msteps, neta, etas, eta_true, geom_bytes, dspec_bytes):
    """The `modeler` object of the line from a timed chi^2 run: rate, pass count, and a roofline of its own.

    Algorithmic bytes per curvature: eigenPAIR passes x 8 N (N + 1) (upper triangle of the packed Hermitian
    theta-theta once per two-vector pass) + 16 N^2 for the gather + 16 ntau nfd written by the rank-1 back-map (recov) +
    16 ntau nfd (recov) read back + 8 nf nt of the dynamic spectrum for chi^2 (the Parseval kernel reads fft2(dspec), 16 ntau nfd,
    in its place: its traffic is above this floor by that difference); when chi^2 came from the back-map's accumulators (round 6:
    `chisq_route`) no image is written or read back -- the back-map reads fft2(dspec), 16 ntau nfd, and that is all.  `achieved` is those
    bytes over the WALL time of the step -- the mat-vecs, back-maps and model transforms of different curvatures share the
    GPU on four streams, so the objective as a whole, not one kernel, is what the fraction describes; `parts` gives each
    kernel's own bytes over the union of its launch intervals."""
    chis, minfo = mod["curves"][0], mod["info"]
    n_ = minfo["N"].astype(float)
    mv = float(np.sum(8.0 * n_ * (n_ + 1.0) * minfo["iters"]))
    ga = float(np.sum(16.0 * n_ * n_))
    rv, mt = neta * geom_bytes, neta * (geom_bytes + dspec_bytes)
    fused = bool(minfo.get("fused", False))
    if fused:
        # chi^2 from the back-map's accumulators (include/scint_hip.h, version 107): no image is written or read back; the back-map
        # workgroups read fft2(dspec) (16 ntau nfd) once, the chi^2 step proper is the edge terms and a sum of 16 387 numbers
        rv, mt = neta * geom_bytes, 0.0
    per_step = mv + ga + rv + mt
    el = mod["elapsed"] / msteps
    hbm_per_eta, src, stale = pmc_modeler_bytes_per_eta()
    ratio = hbm_per_eta / (per_step / neta) if hbm_per_eta else None
    images = neta * msteps

    def part(bytes_per_step, k):
        busy = mod["busy_ms"][k] / 1e3 / msteps
        return {"algorithmic_bytes_per_step": bytes_per_step, "busy_ms_per_step": 1e3 * busy,
                "launches_per_step": mod["launches"][k] / msteps,
                "avg_launch_ms": mod["sum_ms"][k] / max(1, mod["launches"][k]),
                "achieved": bytes_per_step / busy / 1e9 if busy > 0 else 0.0, "unit": "GB/s",
                "frac": bytes_per_step / busy / 1e9 / HBM_PEAK_GBS if busy > 0 else 0.0,
                "share_of_step_time": busy / el}
    return {
        "workload": f"ththmod.modeler / chisq_calc over the same {neta} curvatures (ththmod.py:274-368): "
                    "eigenPAIR (Ritz-residual stop), rank-1 rev_map, inverse FFT, chi^2",
        "value": neta / el, "unit": "eta-points/s", "steps": msteps, "ms_per_step": 1e3 * el,
        "lanczos_steps_mean": float(minfo["iters"].mean()),
        "failed_etas": int(np.sum(minfo["status"] != 0)),
        "chisq_route": {"from_back_map_accumulators": bool(minfo.get("fused", False)), "curvatures_redone_from_a_written_image": int(minfo.get("redone", 0)),
                        "note": "include/scint_hip.h (version 107): on symmetric axes the back-map workgroups add |recov - fft2(dspec)|^2 of their "
                                "interior pixels themselves; the image is neither written nor read back, and the algorithmic bytes below say so "
                                "(back-map: fft2(dspec) read once, 16 ntau nfd; chi^2 step: none)"},
        "matvec_share_of_step_time": mod["busy_ms"][1] / 1e3 / mod["elapsed"],
        "matvec_GBs": mod["mv_bytes"] / (mod["busy_ms"][1] / 1e3) / 1e9 if mod["busy_ms"][1] > 0 else 0.0,
        "eta_at_min_chisq_over_true": float(etas[np.nanargmin(chis)] / eta_true),
        "roofline": {"bound": "hbm", "scope": "the whole objective: algorithmic bytes of every kernel of a step / wall time of the step",
                     "achieved": per_step / el / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": per_step / el / 1e9 / HBM_PEAK_GBS,
                     "algorithmic_bytes_per_eta": per_step / neta,
                     "algorithmic_bytes_per_eta_by_part": ({"eigenpair_passes": mv / neta, "gather": ga / neta,
                                                            "back_map_reads_fft2_dspec": rv / neta, "chisq_step": mt / neta} if fused else
                                                           {"eigenpair_passes": mv / neta, "gather": ga / neta,
                                                            "back_map_write": rv / neta, "model_read_plus_dspec": mt / neta}),
                     "traffic": ratio * per_step / neta if ratio else None,
                     "traffic_note": (f"HBM bytes per eta = {ratio:.3f} x algorithmic (every kernel of a chi^2 step; rocprofv3 "
                                      f"PMC, FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes), {src}") if ratio else stale,
                     "parts": {"pk2_matvec_kernel": part(mv, 1), "thth_gather_packed_kernel": part(ga, 0),
                               "back-map (rev_diag_batch_kernel; rank-1)": dict(
                                   part(rv, 3), images_per_launch=images / max(1, mod["launches"][3]),
                                   avg_ms_per_image=mod["sum_ms"][3] / max(1, images),
                                   note="launches cover the <= 8 curvatures one chunk retired (tail batches); only the delay band "
                                        "a curvature reaches is computed and written, the algorithmic bytes are the full image's"),
                               "chi^2 step (edge terms + final sum beside the fused back-map; chisq_parseval_batch_kernel when the image is written; model transform + sink when cropped or masked)": dict(
                                   part(mt, 4), images_per_launch=images / max(1, mod["launches"][4]),
                                   avg_ms_per_image=mod["sum_ms"][4] / max(1, images))}}}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=4096, help="dynspec is size x size")
    ap.add_argument("--neta", type=int, default=256)
    ap.add_argument("--nedge", type=int, default=None, help="default: size")
    ap.add_argument("--batch", type=int, default=None, help="etas resident per launch")
    ap.add_argument("--npad", type=int, default=0, help="zero-padding multiple of the CS (reference default 3)")
    ap.add_argument("--obs", type=int, default=1, help="observations swept per GPU per step (weak scaling)")
    ap.add_argument("--obs-total", type=int, default=0,
                    help="total observations per step, dealt round-robin to the ranks (strong scaling; "
                         "BASELINE config 4: --size 2048 --obs-total 64)")
    ap.add_argument("--shard", choices=["obs", "eta"], default="obs",
                    help="multi-GPU partitioning: obs = whole observations per rank (default); eta = ONE observation, "
                         "curvatures dealt interleaved, etas[rank::world] (sweep.sharded_eval_sweep, strong scaling)")
    ap.add_argument("--objective", choices=["eig", "chisq"], default="eig",
                    help="objective of the headline timed region: eig = Eval_calc sweep of single_search; "
                         "chisq = modeler/chisq_calc sweep")
    ap.add_argument("--modeler-steps", type=int, default=None,
                    help="steps of the modeler/chisq objective timed after the headline region at N=1 "
                         "(default min(steps, 3); 0 = skip)")
    ap.add_argument("--dyn-npz", default=None,
                    help="observation 0 from a file (keys dyn[nf, nt], freqs [MHz], times [s], eta [s^3]) instead "
                         "of the analytic arc -- e.g. a reference-Simulation screen written by "
                         "tests/tools/make_sim_input.py; --size must match")
    ap.add_argument("--precision", choices=["f64", "mixed"], default="f64",
                    help="operand of the Lanczos passes of the eigenvalue sweep: complex128 throughout (f64), or a "
                         "complex64 copy with a complex128 certificate pass per curvature (mixed; ththmod.sweep_precision)")
    ap.add_argument("--mixed-steps", type=int, default=3,
                    help="N=1, --precision f64: also time this many steps of the mixed sweep on the same workload and "
                         "compare its curve with the float64 one (object 'mixed_precision' of the line)")
    ap.add_argument("--sim-steps", type=int, default=3,
                    help="N=1: also time this many steps of the same sweep on a reference-`Simulation` screen of the same size "
                         "(oracle/sim_oracle.py, SURVEY.md 8d settings, generated outside every timed region; object "
                         "'simulation_screen' of the line); 0 = skip")
    ap.add_argument("--share-steps", type=int, default=3,
                    help="N=1: also time rank 0's and the last rank's interleaved share etas[R::W] of the sweep alone on this GPU for "
                         "W = 2, 4, 8 (config.predicted_strong_scaling: what --shard eta would give if nothing but the shares' own "
                         "time mattered); 0 = skip")
    ap.add_argument("--workload", choices=["sweep", "fit_thetatheta", "tutorial_fit", "wavefield", "fit_arc"], default="sweep",
                    help="sweep (default): the headline line above.  The others time the user entry points either side of the path "
                         "(SURVEY.md 8f) end to end on ONE GPU, each with the CPU oracle timed on a bounded sample beside it, and print "
                         "a line of their own: fit_thetatheta / wavefield = Dynspec.fit_thetatheta / calc_wavefield of a --size^2 "
                         "observation in --chunk^2 chunks (dynspec.py:1657-1856); tutorial_fit = the reference's tutorial recipe on "
                         "its Sample_Data (tests/golden/fit_thetatheta.npz); fit_arc = Dynspec.fit_arc(lamsteps=True) "
                         "(dynspec.py:970-1346)")
    ap.add_argument("--chunk", type=int, default=256, help="--workload fit_thetatheta / wavefield: cwf = cwt")
    ap.add_argument("--workload-steps", type=int, default=2,
                    help="N=1, default line: also time this many calls (after one warm-up) of each --workload entry point on the GPU alone "
                         "(no CPU sample: those stay behind --workload X) -- object 'workloads' of the line; 0 = skip; skipped below "
                         "--size 1024 (the chunked workloads need 4 x 4 chunks of --chunk)")
    ap.add_argument("--strong-steps", type=int, default=3,
                    help="N>1, default partitioning: after the weak-scaling region also time this many steps of --shard eta on observation 0 "
                         "(every rank holds it): object 'strong' of the line (value, efficiency against rank 0's own one-GPU step of the "
                         "weak region, gathered_equals_one_gpu); 0 = skip")
    ap.add_argument("--tol", type=float, default=None,
                    help="Ritz tolerance of the eigenvalue sweeps (default ththmod.DEFAULT_TOL = 1e-12, 1000x inside the 1e-9 parity "
                         "bar); for the tolerance A/Bs of profiles/ -- the headline is quoted at the default")
    ap.add_argument("--side-stream", action="store_true",
                    help="run the timed regions on a non-default torch stream (the sweeps queue on torch's CURRENT stream; the default is "
                         "the legacy null stream, which blocking streams synchronise with implicitly)")
    ap.add_argument("--no-share-walk", action="store_true",
                    help="chi^2 objective: every back-map walks the theta centres itself (the A/B of the partner table that same-crop "
                         "curvatures share; results are bit-identical either way)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed region's own sweep: no CPU baseline, no modeler / mixed / one-slot-group legs "
                         "(profiling runs: every kernel launch of the process then belongs to the headline schedule)")
    ap.add_argument("--cpu-sample", type=int, default=16, help="etas timed on the CPU oracle (spread over the sweep)")
    ap.add_argument("--cpu-reps", type=int, default=3, help="repetitions of the CPU sample (the median is reported)")
    ap.add_argument("--cpu-pool", type=int, default=-1,
                    help="workers of the eta-parallel oracle baseline (multiprocessing.Pool, one BLAS thread "
                         "each: how a user parallelises the reference, dynspec.py:1715-1719); -1 = every "
                         "core the host memory allows, 0 = skip")
    args = ap.parse_args()
    if args.headline_only:
        args.no_cpu_baseline, args.modeler_steps, args.mixed_steps, args.sim_steps, args.share_steps = True, 0, 0, 0, 0
        args.workload_steps = 0
    return args


NPROF = 8      # kernels scint_profile_end reports on (include/scint_hip.h): 0-4 the sweeps', 5-7 calc_sspec's three


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n):
    """Re-run this command line under torch.distributed.run with n ranks; relay its output."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def make_workload(size, neta, nedge, seed, npad=0, npz=None):
    from scintools_amd.synth import arc_dynspec
    from scintools_amd.ththmod import fft_axis
    if npz:
        with np.load(npz) as z:
            dyn, freqs, times, eta_true = np.array(z["dyn"], dtype=np.float64), z["freqs"], z["times"], float(z["eta"])
        if dyn.shape != (size, size):
            raise SystemExit(f"--dyn-npz holds {dyn.shape}, --size says {size}")
    else:
        dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=seed, nimg=64)
    dyn -= dyn.mean()                # as Dynspec.fit_thetatheta hands chunks over (dynspec.py:1692)
    fd = fft_axis(times, 1000.0, npad)     # s -> mHz
    tau = fft_axis(freqs, 1.0, npad)       # MHz -> us
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, nedge)
    etas = np.geomspace(0.25, 4.0, neta) * eta_true
    return dyn, freqs, times, fd, tau, edges, etas, eta_true


def mixed_summary(run, steps, netas):
    """What a run of the mixed sweep streamed, from the library's count (scint_sweep_stats)."""
    s = run["stats"]
    return {"what": "Lanczos passes on a complex64 copy of theta-theta (float64 vectors and sums); the eigenvalue returned is "
                    "the Ritz value of a certificate pass on the complex128 tiles under the float64 sweep's a-posteriori bound",
            "complex64_bytes_per_step": s[0] / steps, "complex128_bytes_per_step": s[1] / steps,
            "certified_per_step": s[2] / steps, "certificate_passes_mean": s[3] / max(1.0, s[2]),
            "lanczos_steps_mean": float(run["info"]["iters"].mean()), "curvatures_per_step": netas}


def mixed_leg(mx, steps, neta, ref_curve, f64_value):
    """The `mixed_precision` object of the line: the mixed sweep timed on the headline workload, against the float64 curve."""
    got = mx["curves"][0]
    return dict(
        mixed_summary(mx, steps, neta),
        value=neta * steps / mx["elapsed"], unit="eta-points/s", steps=steps,
        ms_per_step=1e3 * mx["elapsed"] / steps,
        speedup_vs_f64=(neta * steps / mx["elapsed"]) / f64_value,
        failed_etas=int(np.sum(mx["info"]["status"] != 0)),
        max_rel_diff_vs_f64_curve=float(np.nanmax(np.abs(got - ref_curve) / np.abs(ref_curve))),
        matvec32={"achieved": mx["stats"][0] / (mx["busy_ms"][2] / 1e3) / 1e9 if mx["busy_ms"][2] > 0 else 0.0,
                  "unit": "GB/s", "avg_launch_ms": mx["sum_ms"][2] / max(1, mx["launches"][2]),
                  "launches": int(mx["launches"][2]), "busy_ms": mx["busy_ms"][2],
                  "share_of_step_time": mx["busy_ms"][2] / 1e3 / mx["elapsed"],
                  "note": "complex64 bytes / time a complex64 (or combined) mat-vec launch is in flight; the certificate strips "
                          "ride in the same launches, their complex128 bytes are not counted here"},
        matvec64={"launches": int(mx["launches"][1]), "busy_ms": mx["busy_ms"][1],
                  "note": "complex128-only launches (when no complex64 strip is left in the group)"})


def lanczos_block():
    """(vectors per Lanczos pass, name of the mat-vec kernel): the one recurrence the library ships
    (eigen_packed.hip); the single-vector and the four- / eight-vector families were measured
    and removed (profiles/r03_wide_blocks_ab.json)."""
    return 2, "pk2_matvec_kernel (two-vector block Lanczos mat-vec; eight block rows x <= 12 column tiles per workgroup)"


def blas_threads():
    try:
        from threadpoolctl import threadpool_info
        return max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline(dyn, tau, fd, edges, etas, nsample, npad=0, reps=3):
    """Oracle (port of the reference) by SURVEY.md 8d's protocol: one repetition = the conjugate-spectrum
    FFT (ththmod.py:777-787) + Eval_calc on `nsample` curvatures spread over the sweep; `reps` repetitions,
    the median is kept; the sweep rate is neta / (t_fft + neta * t_eta) with the per-eta time of the sample."""
    from oracle import thth_oracle
    idx = np.unique(np.linspace(0, len(etas) - 1, nsample + 2).astype(int)[1:-1])
    runs, vals = [], None
    for _ in range(max(1, reps)):
        t0 = time.perf_counter()
        CS = thth_oracle.conjugate_spectrum(dyn, npad)
        t1 = time.perf_counter()
        vals = [thth_oracle.Eval_calc(CS, tau, fd, etas[i], edges) for i in idx]
        t2 = time.perf_counter()
        runs.append((t1 - t0, t2 - t1))
    runs.sort(key=lambda r: r[0] + r[1])
    t_fft, t_eta = runs[len(runs) // 2]
    per_eta = t_eta / len(idx)
    sweep_s = t_fft + len(etas) * per_eta
    return {"value": len(etas) / sweep_s, "unit": "eta-points/s", "cores": int(blas_threads()), "kind": "port",
            "host_cores": os.cpu_count(),
            "sample": f"oracle CS FFT + Eval_calc (NumPy gather + ARPACK eigsh) on {len(idx)} of {len(etas)} etas "
                      f"(indices {idx.tolist()}) of the same {dyn.shape[0]}x{dyn.shape[1]} workload; median of "
                      f"{len(runs)} repetitions: FFT {t_fft:.2f} s + {per_eta:.2f} s per eta, scaled to the "
                      f"{len(etas)}-eta sweep ({sweep_s:.0f} s); one process, {int(blas_threads())} BLAS threads "
                      f"(as shipped) on a {os.cpu_count()}-core host",
            "repetitions_s": [round(a + b, 2) for a, b in runs]}, dict(zip(idx.tolist(), vals))


def _pool_worker(job):
    """One eta of the oracle sweep in a worker process (spawned: no GPU state is inherited)."""
    path, i = job
    import numpy as _np
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except Exception:
        ctx = None
    from oracle import thth_oracle
    z = _np.load(path, mmap_mode="r")
    meta = _np.load(path.replace("_cs.npy", "_meta.npz"))
    val = thth_oracle.Eval_calc(_np.asarray(z), meta["tau"], meta["fd"], float(meta["etas"][i]), meta["edges"])
    del ctx
    return i, val


def _pool_warm(path):
    """Worker warm-up: the imports and the page cache of the shared conjugate spectrum, no eta."""
    import numpy as _np
    from oracle import thth_oracle  # noqa: F401
    z = _np.load(path, mmap_mode="r")
    return float(_np.abs(z[::64, ::64]).sum())


def pool_size(requested, nedge):
    """Workers for the eta-parallel baseline: every core, capped by host memory (one oracle
    gather holds about 80 M^2 bytes of temporaries) and by 64."""
    if requested >= 0:
        return requested
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    avail = None
    try:
        with open("/proc/meminfo") as fh:
            for line in fh:
                if line.startswith("MemAvailable:"):
                    avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            with open(path) as fh:
                v = fh.read().strip()
            if v.isdigit():
                avail = min(avail, int(v)) if avail else int(v)
        except OSError:
            pass
    per_worker = 80 * nedge * nedge + (1 << 30)
    by_mem = int((avail or (64 << 30)) * 0.5 // per_worker)
    return int(max(1, min(cores, by_mem, 64)))


def cpu_baseline_pool(dyn, tau, fd, edges, etas, nproc, npad=0):
    """eta-parallel oracle: Pool(nproc).map over nproc curvatures spread over the sweep (one wave of
    equal-cost jobs: about 15 s at 4096^2, as the bounded-sample rule of the bench contract asks)."""
    import multiprocessing as mp
    import shutil
    import tempfile
    from oracle import thth_oracle
    CS = thth_oracle.conjugate_spectrum(dyn, npad)
    tmp = tempfile.mkdtemp(prefix="scint_bench_")
    path = os.path.join(tmp, "w_cs.npy")
    np.save(path, CS)
    np.savez(path.replace("_cs.npy", "_meta.npz"), tau=tau, fd=fd, etas=etas, edges=edges)
    idx = np.unique(np.linspace(0, len(etas) - 1, nproc + 2).astype(int)[1:-1]).tolist()
    ctx = mp.get_context("spawn")
    try:
        with ctx.Pool(nproc) as pool:
            # warm the workers (imports, page cache); bounded waits: a pool whose workers die at start-up
            # respawns them for ever, and a baseline must never hang the benchmark
            pool.map_async(_pool_warm, [path] * nproc, chunksize=1).get(timeout=180)
            t0 = time.perf_counter()
            pool.map_async(_pool_worker, [(path, i) for i in idx]).get(timeout=300)
            dt = time.perf_counter() - t0
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {"value": len(idx) / dt, "unit": "eta-points/s", "cores": int(nproc), "kind": "port",
            "host_cores": os.cpu_count(),
            "sample": f"oracle Eval_calc over multiprocessing.Pool({nproc}) (spawn, 1 BLAS thread per worker) on "
                      f"{len(idx)} of {len(etas)} etas, {dt:.1f} s"}


def simulation_screen_leg(args, size, neta, nedge, wl, main_wl, timed, ththmod):
    """The `simulation_screen` object: the headline sweep on the input SURVEY.md 8d's config 3 actually specifies -- a reference
    `Simulation` screen (scint_sim.py:23-415; mb2=20, ar=10, Kolmogorov, seed 3), restated bit for bit by oracle/sim_oracle.py
    and generated here OUTSIDE every timed region (an input generator; nothing of it is measured).  On such a screen
    lambda_2 / lambda_1 -> 0.99 on the flat ends of the curve and a curvature needs ~1.65x the passes of the analytic arc: this
    is the rate a user's observation gets.  The values at the golden indices are compared with the reference's own Eval_calc
    run (tests/golden/sim_sweep_4096.npz, committed; generated by tests/golden/make_golden.py from the unmodified reference)."""
    try:
        from oracle import sim_oracle
        from scintools_amd.ththmod import fft_axis
        import torch
        t0 = time.perf_counter()
        workers = sim_oracle.default_workers(64)
        sim = sim_oracle.baseline_dynspec(size, 3, workers=workers)
        gen_s = time.perf_counter() - t0
        d = np.array(sim.dyn, dtype=np.float64)
        d -= d.mean()
        fd = fft_axis(sim.times, 1000.0, args.npad)
        tau = fft_axis(sim.freqs, 1.0, args.npad)
        edges = np.linspace(-fd.max() / 2, fd.max() / 2, nedge)
        etas = np.geomspace(0.25, 4.0, neta) * float(sim.eta)
        k = args.sim_steps
        wl.update(main_wl, dyns=[ththmod.to_device(d, torch.float64)], tau=tau, fd=fd, edges=edges, etas=etas)
        try:
            r = timed("eig", k, 1)
        finally:
            wl.update(main_wl)
        info, eigs = r["info"], r["curves"][0]
        passes = info["iters"]
        hist = {str(int(p)): int(c) for p, c in zip(*np.unique(passes, return_counts=True))}
        leg = {"what": f"the same {neta}-eta sweep on a reference Simulation screen {size}x{size} (oracle/sim_oracle.baseline_dynspec({size}, 3): "
                       "mb2=20, ar=10, psi=0, alpha=5/3, dlam=0.25, ny=128), nedge and eta range as the headline",
               "value": neta * k / r["elapsed"], "unit": "eta-points/s", "steps": k, "ms_per_step": 1e3 * r["elapsed"] / k,
               "ratio_to_headline_input": None,
               "lanczos_steps_mean": float(passes.mean()), "passes_per_eta_histogram": hist,
               "failed_etas": int(np.sum(info["status"] != 0)), "N_min": int(info["N"].min()), "N_max": int(info["N"].max()),
               "matvec_GBs": r["mv_bytes"] / (r["busy_ms"][1] / 1e3) / 1e9 if r["busy_ms"][1] > 0 else 0.0,
               "matvec_share_of_step_time": r["busy_ms"][1] / 1e3 / r["elapsed"],
               "eta_fit_over_sim_eta": float(r["fit"][0] / sim.eta) if np.isfinite(r["fit"][0]) else None,
               "input_generation_s": gen_s, "input_workers": workers, "input_sha256": sim_oracle.checksum(sim.dyn)[:16]}
        gpath = os.path.join(REPO, "tests", "golden", f"sim_sweep_{size}.npz")
        if os.path.exists(gpath) and args.npad == 0 and nedge == size:
            with np.load(gpath) as g:
                same_input = sim_oracle.checksum(sim.dyn) == str(g["sha256"])
                idx = g["idx"]
                if same_input and neta == 256 and np.array_equal(g["etas"], etas[idx]):
                    leg["max_rel_diff_vs_reference_values"] = float(np.max(np.abs(eigs[idx] - g["eigs"]) / np.abs(g["eigs"])))
                    leg["reference_values"] = f"tests/golden/sim_sweep_{size}.npz: the unmodified reference's Eval_calc at eta indices {idx.tolist()}"
                leg["input_is_the_reference_screen"] = bool(same_input)
        return leg
    except Exception as exc:                       # a reported leg must never take the headline down
        return {"error": repr(exc)}


def sspec_roofline(n, ms, parts_ms):
    """`roofline` of calc_sspec at n x n: which roof binds each of its three kernels, from the committed per-kernel HBM bytes and
    instruction counts of THIS library's kernels (profiles/*_sspec_roofline_<n>.json: rocprofv3 PMC, tools/sspec_roofline.py)
    and the kernel times measured live (hipEvent brackets inside scint_sspec).  A kernel's floor is the larger of its HBM floor
    (measured bytes at the ~6.3 TB/s the guide calls achievable) and its issue floor (VALU instructions x wavefronts x 4
    cycles over 1024 SIMDs at 2.4 GHz); `frac` = floor / measured time says how far the kernel is from the roof that binds it."""
    summ_ratio, src, stale = _newest_summary(f"*_sspec_roofline_{n}.json", lambda summ: summ)
    if summ_ratio is None:
        return {"note": stale}
    summ = summ_ratio
    kern, floor_total = {}, 0.0
    for name, live in zip(("prep", "cols", "rows"), parts_ms):
        k = summ["kernels"].get(name)
        if not k:
            continue
        hb, iss = k["hbm_floor_us"] / 1e3, k["issue_floor_us"] / 1e3
        extra = summ["kernels"].get("prep_means", {}).get("duration_under_counters_us", 0.0) / 1e3 if name == "prep" else 0.0
        floor = max(hb, iss)
        floor_total += floor
        kern[name] = {"ms": live, "hbm_bytes": k["hbm_bytes_per_launch"], "hbm_floor_ms": hb, "valu_insts_per_wavefront": k["valu_insts_per_wavefront"],
                      "wavefronts": k["wavefronts"], "issue_floor_ms": iss, "bound": "valu-issue" if iss > hb else "hbm",
                      "frac_of_binding_roof": floor / live if live > 0 else None}
        if extra:
            kern[name]["includes_the_one_block_sum_kernel_ms"] = extra
    alg = 8.0 * n * n + 8.0 * n * (2 * n)
    return {"source": src, "traffic": summ["hbm_bytes"], "traffic_over_algorithmic": summ["traffic_over_algorithmic"],
            "algorithmic_floor_ms": 1e3 * alg / 6.3e12, "two_trip_floor_ms": 1e3 * (alg + 2 * 8.0 * n * (2 * n)) / 6.3e12,
            "sum_of_binding_floors_ms": floor_total, "frac": floor_total / ms if ms > 0 else None,
            "bound": "valu-issue" if sum(1 for v in kern.values() if v["bound"] == "valu-issue") >= 2 else "hbm", "kernels": kern,
            "note": "frac = sum over the three kernels of max(HBM floor at 6.3 TB/s of its MEASURED bytes, VALU-issue floor of its MEASURED "
                    "instruction count) / time of a call; the algorithmic and the two-trip byte floors are given beside it"}


def sspec_timing(torch, size, lib=None):
    """Dynspec.calc_sspec (window, zero-padded real-to-complex 2-D FFT, |.|^2, shift, dB;
    dynspec.py:3665-3721) on a size^2 and a (2 size)^2 dynamic spectrum resident in HBM.
    Algorithmic bytes: 8 nf nt read + 8 (nrfft/2) ncfft written."""
    from scintools_amd.dynspec import sspec_device
    res = {}
    for n in (size, 2 * size):
        try:
            x = torch.randn(n, n, dtype=torch.float64, device="cuda")      # input creation, not the product path
            sspec_device(x)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10 if n <= 4096 else 4
            a.record()
            for _ in range(reps):
                sspec_device(x)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / reps
            alg = 8.0 * n * n + 8.0 * n * (2 * n)          # nrfft/2 = n rows kept, ncfft = 2n columns
            res[f"{n}x{n}"] = {"ms": ms, "algorithmic_bytes": alg, "GBs": alg / ms / 1e6,
                               "frac_of_hbm_peak": alg / ms / 1e6 / HBM_PEAK_GBS}
            if lib is not None:
                # the same calls again with the library's own brackets around its three kernels (their events would sit between
                # the kernels of the timing loop above)
                lib.scint_profile_begin()
                for _ in range(reps):
                    sspec_device(x)
                pm, ps, pl = (ctypes.c_double * NPROF)(), (ctypes.c_double * NPROF)(), (ctypes.c_int64 * NPROF)()
                lib.scint_profile_end(pm, ps, pl, NPROF)
                parts = [ps[k] / max(1, pl[k]) for k in (5, 6, 7)]
                if all(pl[k] > 0 for k in (5, 6, 7)):
                    res[f"{n}x{n}"]["kernels_ms"] = dict(zip(("prep", "cols", "rows"), parts))
                    res[f"{n}x{n}"]["roofline"] = sspec_roofline(n, ms, parts)
            del x
        except Exception as exc:
            res[f"{n}x{n}"] = {"error": repr(exc)}
    return res


class _Obs:
    """The attribute set Dynspec.load_dyn_obj copies (dynspec.py:378-419)."""

    def __init__(self, dyn, freqs, times, name):
        self.dyn, self.freqs, self.times, self.name = dyn, freqs, times, name
        self.dt, self.df = float(times[1] - times[0]), float(freqs[1] - freqs[0])


def _median_time(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), ts


def workload_main(args):
    """`--workload X`: the (f) rows of SURVEY.md 8 end to end on one GPU.  One JSON line each:
    seconds per call (median of --steps calls after --warmup), the work it covers, where the GPU time goes (the library's
    hipEvent brackets: gather, mat-vec, back-map), a parity figure against the oracle on a sample, and `cpu_baseline` -- the
    oracle (kind "port") timed here on a bounded sample of the same work and scaled, sample stated."""
    print(json.dumps(run_workload(args, with_cpu=True)))


def run_workload(args, with_cpu=True):
    """One --workload entry point timed on the GPU; with_cpu also times the oracle on a bounded sample (parity + cpu_baseline)."""
    import torch
    from scintools_amd import _lib, ththmod
    from scintools_amd.device import require_gpu
    from scintools_amd.dynspec import Dynspec
    from scintools_amd.synth import arc_dynspec
    torch.cuda.set_device(0)
    require_gpu()
    lib = _lib.load()
    size, cw, steps, warm = args.size, args.chunk, max(1, args.steps), max(1, args.warmup)
    out = {"metric": f"{args.workload}_seconds", "unit": "s", "higher_is_better": False, "n_gpus": 1, "steps": steps, "warmup": warm,
           "dtype": "f64", "data": "synthetic", "library": library_fingerprint()}

    def profiled(fn):
        """median seconds of `steps` calls + the library's per-kernel busy time over them"""
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        lib.scint_profile_begin()
        med, ts = _median_time(lambda: (fn(), torch.cuda.synchronize()), steps)
        ms, ms_sum, launches = (ctypes.c_double * NPROF)(), (ctypes.c_double * NPROF)(), (ctypes.c_int64 * NPROF)()
        lib.scint_profile_end(ms, ms_sum, launches, NPROF)
        tot = sum(ts)
        names = ("thth_gather_packed_kernel", "pk2_matvec_kernel", "pk2_matvec32_kernel", "back-map (rev_diag / rev_gather / rev_row kernels)", "model / chi^2 step")
        return med, ts, {n: {"busy_share_of_wall": ms[k] / 1e3 / tot, "launches_per_call": launches[k] / steps,
                             "avg_launch_us": 1e3 * ms_sum[k] / max(1, launches[k])} for k, n in enumerate(names) if launches[k]}

    if args.workload in ("fit_thetatheta", "wavefield"):
        if with_cpu:                     # the oracle is the checker and the CPU baseline of these lines, nothing else
            from oracle import thth_oracle
        dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=3, nimg=64)
        d = Dynspec(dyn=_Obs(dyn, freqs, times, f"arc {size}x{size}"), process=False, verbose=False)
        kw = dict(cwf=cw, cwt=cw, eta_min=0.5 * eta_true, eta_max=2.0 * eta_true, npad=args.npad if args.npad else 3)
        if args.nedge:
            kw["nedge"] = args.nedge
        d.prep_thetatheta(**kw)
        nchunk = d.ncf_fit * d.nct_fit
        cfg = {"workload": f"Dynspec.{'fit_thetatheta' if args.workload == 'fit_thetatheta' else 'calc_wavefield'} on a {size}x{size} "
                           f"observation, cwf = cwt = {cw}, npad = {d.npad} (chunk CS {(d.npad + 1) * cw}^2), {d.edges.shape[0]} edges, "
                           f"{d.neta} curvatures per chunk (eta_min .. eta_max = 0.5 .. 2 eta_true, fw = {d.fw})",
               "fit_chunks": nchunk, "neta": int(d.neta), "nedge": int(d.edges.shape[0]), "chunk_cs": [(d.npad + 1) * cw] * 2}
        if args.workload == "fit_thetatheta":
            med, ts, kern = profiled(lambda: d.fit_thetatheta())
            jobs = nchunk * d.neta
            st = (ctypes.c_double * 4)()
            lib.scint_sweep_stats(st)                  # the last call's sweep (all chunks are one sweep while their stack fits 8 GiB)
            mv = kern.get("pk2_matvec_kernel")
            if mv and st[1] > 0:
                mv["algorithmic_GB_per_call"] = st[1] / 1e9
                mv["GBs_in_flight"] = st[1] / 1e9 / (mv["busy_share_of_wall"] * med)
                mv["frac_of_hbm_peak"] = mv["GBs_in_flight"] / HBM_PEAK_GBS
                mv["note"] = ("8 N (N + 1) bytes per pass summed by the library (scint_sweep_stats) over the time a mat-vec launch is in "
                              "flight; matrices this small (N ~ 1200: 11 MB) stay in the 256 MiB Infinity Cache between the passes of a "
                              "chunk of launches, so the rate may exceed what HBM alone delivers")
            # parity sample + CPU port: single_search of the oracle on a few chunks
            sample = [(0, 0), (d.ncf_fit // 2, d.nct_fit // 2)][: max(1, args.cpu_sample // 8)] if with_cpu else []
            t_cpu, diffs = [], []
            for cf, ct in sample:
                p_ = d._search_params(cf, ct)
                t0 = time.perf_counter()
                r = thth_oracle.single_search(p_[0], p_[1], p_[2], p_[3], p_[4], fw=d.fw, npad=d.npad)
                t_cpu.append(time.perf_counter() - t0)
                both_nan = not np.isfinite(d.eta_evo[cf, ct]) and not np.isfinite(r[0])        # (the parabola fit fails on both sides alike)
                diffs.append(0.0 if both_nan else abs(d.eta_evo[cf, ct] - r[0]) / abs(r[0]))
                diffs.append(float(np.nanmax(np.abs(d.thth_eigs[cf, ct] - r[4]) / np.abs(r[4]))))
            out.update(value=med, seconds_all=ts, config=dict(cfg, chunk_eta_jobs=jobs),
                       chunk_eta_points_per_s=jobs / med, kernels=kern,
                       eta_fit_over_true_median=float(np.nanmedian(d.eta_evo) / eta_true))
            if with_cpu:
                out.update(parity_sample={"chunks": sample, "max_rel_diff_eta_fit_vs_oracle": float(max(diffs[0::2])),
                                      "max_rel_diff_eigs_vs_oracle": float(max(diffs[1::2]))},
                       cpu_baseline={"value": float(np.median(t_cpu)) * nchunk, "unit": "s", "kind": "port", "cores": int(blas_threads()),
                                     "host_cores": os.cpu_count(),
                                     "sample": f"oracle.thth_oracle.single_search on {len(sample)} of {nchunk} chunks "
                                               f"({[round(t, 1) for t in t_cpu]} s), median x {nchunk}"})
                out["speedup_vs_cpu_baseline"] = out["cpu_baseline"]["value"] / med
        else:
            d.fit_thetatheta()
            nret = d.ncf_ret * d.nct_ret

            def once():
                if type(d).chunks.present(d):              # (hasattr would copy the parked gigabyte to the host)
                    del d.chunks
                d.calc_wavefield()
            med, ts, kern = profiled(once)
            sample = [(0, 0), (d.ncf_ret // 2, d.nct_ret // 2)][: max(1, args.cpu_sample // 8)] if with_cpu else []
            t_cpu, diffs = [], []
            for cf, ct in sample:
                fs = slice(cf * (cw // 2), cf * (cw // 2) + cw)
                tsl = slice(ct * (cw // 2), ct * (cw // 2) + cw)
                freq2, time2 = d.freqs[fs], d.times[tsl]
                dspec2 = np.nan_to_num(d.dyn[fs, tsl] - np.nanmean(d.dyn[fs, tsl]))
                eta = d.ththeta * (d.fref / freq2.mean()) ** 2
                t0 = time.perf_counter()
                ref = thth_oracle.single_chunk_retrieval(dspec2, d.edges * (freq2.mean() / d.fref), time2, freq2, eta, d.npad)
                t_cpu.append(time.perf_counter() - t0)
                got = d.chunks[cf, ct]
                ph = np.vdot(ref, got)
                ph /= abs(ph)
                diffs.append(float(np.abs(got / ph - ref).max() / np.abs(ref).max()))
            out.update(value=med, seconds_all=ts, config=dict(cfg, retrieval_chunks=nret), chunks_per_s=nret / med, kernels=kern)
            if with_cpu:
                out.update(parity_sample={"chunks": sample, "max_rel_diff_vs_oracle_modulo_global_phase": float(max(diffs))},
                           cpu_baseline={"value": float(np.median(t_cpu)) * nret, "unit": "s", "kind": "port", "cores": int(blas_threads()),
                                         "host_cores": os.cpu_count(),
                                         "sample": f"oracle.thth_oracle.single_chunk_retrieval on {len(sample)} of {nret} chunks "
                                                   f"({[round(t, 1) for t in t_cpu]} s), median x {nret} (mosaic not included: host NumPy in both)"})
                out["speedup_vs_cpu_baseline"] = out["cpu_baseline"]["value"] / med
    elif args.workload == "tutorial_fit":
        if with_cpu:                     # the oracle is the checker and the CPU baseline of these lines, nothing else
            from oracle import thth_oracle
        with np.load(os.path.join(REPO, "tests", "golden", "fit_thetatheta.npz")) as g:
            obs = _Obs(np.array(g["dspec"], dtype=float), g["freq"], g["time"], "Sample_Data (tutorial)")
            ref_evo, ref_ththeta = np.array(g["eta_evo"]), float(g["ththeta"])
        d = Dynspec(dyn=obs, process=False, verbose=False)

        def once():
            d.prep_thetatheta(cwf=64, edges_lim=.3, eta_min=30, eta_max=50)        # the recipe of dynspec_thth.rst:146-170
            d.fit_thetatheta()
        med, ts, kern = profiled(once)
        nchunk = d.ncf_fit * d.nct_fit
        out.update(value=med, seconds_all=ts, kernels=kern,
                   config={"workload": "the reference's tutorial: prep_thetatheta(cwf=64, edges_lim=.3, eta_min=30, eta_max=50) + "
                                       f"fit_thetatheta on Sample_Data ({obs.dyn.shape[0]} x {obs.dyn.shape[1]}), npad = 3", "fit_chunks": nchunk, "neta": int(d.neta),
                           "nedge": int(d.edges.shape[0])},
                   parity={"max_rel_diff_eta_evo_vs_reference_run": float(np.nanmax(np.abs(d.eta_evo - ref_evo) / np.abs(ref_evo))),
                           "rel_diff_ththeta_vs_reference_run": abs(d.ththeta - ref_ththeta) / abs(ref_ththeta)})
        if with_cpu:
            p_ = d._search_params(0, 0)
            idx = np.unique(np.linspace(0, d.neta - 1, 8).astype(int))
            fd_, tau_ = thth_oracle.fft_axis(p_[2], 1000.0, d.npad), thth_oracle.fft_axis(p_[1], 1.0, d.npad)
            t0 = time.perf_counter()
            CS = thth_oracle.conjugate_spectrum(p_[0], d.npad, tau_, 0.0)
            vals = [thth_oracle.Eval_calc(CS, tau_, fd_, p_[3][i], p_[4]) for i in idx]
            dt_cpu = time.perf_counter() - t0
            timing = None
            try:
                with open(os.path.join(REPO, "tests", "golden", "reference_workload_timing.json")) as fh:
                    timing = json.load(fh).get("tutorial_fit_thetatheta")
            except (OSError, ValueError):
                pass
            out["parity"]["max_rel_diff_eigs_vs_oracle_sample"] = float(max(abs(d.thth_eigs[0, 0][i] - v) / abs(v) for i, v in zip(idx, vals)))
            out["cpu_baseline"] = {"value": dt_cpu / len(idx) * d.neta * nchunk, "unit": "s", "kind": "port", "cores": int(blas_threads()),
                                   "host_cores": os.cpu_count(),
                                   "sample": f"oracle CS + Eval_calc on {len(idx)} of {d.neta} curvatures of chunk 0 ({dt_cpu:.1f} s), "
                                             f"x {d.neta} / {len(idx)} x {nchunk} chunks",
                                   "reference_itself": timing}
            out["speedup_vs_cpu_baseline"] = out["cpu_baseline"]["value"] / med
    else:   # fit_arc
        if with_cpu:                     # the oracle is the checker and the CPU baseline of these lines, nothing else
            from oracle import arcfit_oracle
        dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=3, nimg=64)
        obs = _Obs(dyn, freqs, times, f"arc {size}x{size}")
        d = Dynspec(dyn=obs, process=False, verbose=False)

        def once():
            # forget the previous call's products WITHOUT looking at them: hasattr() on a device-parked attribute is a host read, i.e. a
            # 134-MB copy down that the timed region then paid for (50 of the 60-80 ms this line reported until the round's last closing call)
            for a in ("lamsspec", "lamdyn", "betaeta"):
                slot = getattr(type(d), a, None)
                if hasattr(slot, "present"):
                    if slot.present(d):
                        delattr(d, a)
                else:
                    d.__dict__.pop(a, None)
            d.fit_arc(lamsteps=True, numsteps=1e4)
        med, ts, kern = profiled(once)
        out.update(value=med, seconds_all=ts, kernels=kern,
                   config={"workload": f"Dynspec.fit_arc(lamsteps=True, numsteps=1e4) from the raw {size}x{size} dynspec: cubic-spline "
                                       "resample to equal wavelength steps, secondary spectrum, norm_sspec, parabola fit"},
                   betaeta=float(d.betaeta), betaetaerr=float(d.betaetaerr))
        if with_cpu:
            # CPU port on the top-left quarter (size/2)^2 of the same observation: the chain is O(pixels log pixels)
            h = size // 2
            t0 = time.perf_counter()
            o = arcfit_oracle.calc_sspec_lam(dyn[:h, :h], freqs[:h], obs.dt, obs.df)
            t1 = time.perf_counter()
            fa = arcfit_oracle.fit_arc(o["lamsspec"], o["beta"], o["tdel"], o["beta"], o["fdop"], float(np.mean(freqs[:h])),
                                       lamsteps=True, numsteps=1e4)
            t2 = time.perf_counter()
            out["cpu_baseline"] = {"value": 4.0 * (t2 - t0), "unit": "s", "kind": "port", "cores": int(blas_threads()), "host_cores": os.cpu_count(),
                                   "sample": f"oracle scale_dyn + calc_sspec ({t1 - t0:.1f} s) + fit_arc ({t2 - t1:.1f} s) on the top-left "
                                             f"{h}x{h} quarter of the same observation, x 4 (pixel count)",
                                   "betaeta_of_the_quarter": float(fa["sides"][0]["eta"])}
            out["speedup_vs_cpu_baseline"] = out["cpu_baseline"]["value"] / med
    return out


def main():
    args = parse()
    if args.workload != "sweep":
        return workload_main(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    from scintools_amd import _lib, ththmod
    from scintools_amd.device import require_gpu

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = max(1, torch.cuda.device_count())
    # one rank per GPU over RCCL; with fewer GPUs than ranks (the 1-GPU test box) the ranks share
    # devices and the curves travel over gloo -- the JSON line says so (ranks_per_gpu)
    backend = os.environ.get("SCINT_BENCH_BACKEND", "nccl" if ndev >= world else "gloo")
    ranks_per_gpu = -(-world // ndev)
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)
    comm_dev = "cuda" if backend == "nccl" else "cpu"
    require_gpu()
    lib = _lib.load()
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    size, neta = args.size, args.neta
    nedge = args.nedge or size
    shard_eta = args.shard == "eta"
    if shard_eta and (args.obs_total > 0 or args.obs != 1 or args.objective != "eig"):
        raise SystemExit("--shard eta splits the eigenvalue sweep of ONE observation: not with --obs / --obs-total / chisq")
    if shard_eta:
        obs_ids, n_obs_job, scaling = [0], 1, "strong"        # every rank holds the same observation
    elif args.obs_total > 0:
        obs_ids = list(range(rank, args.obs_total, world))       # round-robin (dynspec.py:1706-1723 is the pattern)
        n_obs_job, scaling = args.obs_total, "strong"
    else:
        obs_ids = [rank * args.obs + k for k in range(args.obs)]
        n_obs_job, scaling = world * args.obs, "weak"
    per_rank_max = 1 if shard_eta else -(-n_obs_job // world)
    dyn, freqs, times, fd, tau, edges, etas, eta_true = make_workload(size, neta, nedge, seed=3, npad=args.npad,
                                                                      npz=args.dyn_npz)
    dyns = []
    for i in obs_ids:                      # every observation resident in HBM before the clock starts
        d_i = dyn if i == 0 else make_workload(size, neta, nedge, seed=3 + 97 * i, npad=args.npad)[0]
        dyns.append(ththmod.to_device(d_i, torch.float64))
    R, C = (args.npad + 1) * size, (args.npad + 1) * size
    stack = torch.empty((max(1, len(dyns)), R, C), dtype=torch.complex128, device="cuda") if len(dyns) > 1 else None
    gathered = [torch.empty((per_rank_max, neta), dtype=torch.float64, device=comm_dev) for _ in range(world)]

    # the workload a step runs on: the headline's, or -- for the reported legs below -- a dict that overrides parts of it
    # (another observation with its axes: the simulation-screen leg; a rank's share of the curvatures: the rank-share leg)
    main_wl = {"dyns": dyns, "tau": tau, "fd": fd, "edges": edges, "etas": etas, "batch": args.batch}
    tol = args.tol if args.tol else ththmod.DEFAULT_TOL
    wl = dict(main_wl)

    def step(objective):
        """One pass of the hot path over this rank's observations: the body of single_search
        (ththmod.py:773-859) -- CS once per observation, the eta loop, the peak fit."""
        dyns, tau, fd, edges, etas, batch = (wl[k] for k in ("dyns", "tau", "fd", "edges", "etas", "batch"))
        curves = np.full((per_rank_max, len(etas)), np.nan)
        fit, info = (np.nan, np.nan, None), None
        if objective == "chisq":
            for k, d_t in enumerate(dyns):
                cs_t = ththmod.conjugate_spectrum(d_t, args.npad, tau, 0.0, True)
                curves[k], info = ththmod.chisq_sweep(d_t, cs_t, tau, fd, etas, edges, 1.0, return_info=True, tol=tol, batch=batch,
                                                      share_walk=not args.no_share_walk)
                fit = (etas[np.nanargmin(curves[k])], np.nan, None)
        elif shard_eta:
            # ONE observation, this rank's interleaved share of the curvatures; the all-gather is inside
            from scintools_amd import sweep
            cs_t = ththmod.conjugate_spectrum(dyns[0], args.npad, tau, 0.0, True)
            curves[0], info = sweep.sharded_eval_sweep(cs_t, tau, fd, etas, edges, batch=batch, tol=tol, return_info=True)
            fit = ththmod.fit_eig_peak(etas, curves[0], 0.1)
            return curves, info, fit
        elif len(dyns) == 1:
            cs_t = ththmod.conjugate_spectrum(dyns[0], args.npad, tau, 0.0, True)
            curves[0], info = ththmod.eval_sweep(cs_t, tau, fd, etas, edges, batch=batch, tol=tol, return_info=True)
            fit = ththmod.fit_eig_peak(etas, curves[0], 0.1) if len(etas) >= 8 else (np.nan, np.nan, None)
        elif dyns:
            # several observations on this GPU: all conjugate spectra in one stack, all
            # (observation, eta) pairs in ONE continuously batched sweep
            for k, d_t in enumerate(dyns):
                ththmod.conjugate_spectrum(d_t, args.npad, tau, 0.0, True, out=stack[k])
            eig_list, info = ththmod.eval_sweep_multi(stack, [(tau, fd, edges)] * len(dyns), [etas] * len(dyns),
                                                      batch=batch, tol=tol, return_info=True)
            for k, e in enumerate(eig_list):
                curves[k] = e
                fit = ththmod.fit_eig_peak(etas, e, 0.1)
        if world > 1:
            dist.all_gather(gathered, torch.from_numpy(curves).to(comm_dev))
        return curves, info, fit

    def timed(objective, steps, warmup):
        for _ in range(warmup):
            step(objective)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        lib.scint_profile_begin()
        t0 = time.perf_counter()
        mv_bytes = 0.0
        local_etas = 0
        stats = np.zeros(4)
        st = (ctypes.c_double * 4)()
        for _ in range(steps):
            curves, info, fit = step(objective)
            if info is not None:
                local_etas += int(info["N"].shape[0])
                n_ = info["N"].astype(float)
                if objective == "eig" and ththmod.sweep_precision() in ("mixed", "mixed-all"):
                    # bytes by operand from the library's own count (the per-eta step counts do not split by phase)
                    lib.scint_sweep_stats(st)
                    stats += np.array(list(st))
                    mv_bytes += st[0] + st[1]
                else:
                    mv_bytes += float(np.sum(8.0 * n_ * (n_ + 1.0) * info["iters"]))   # Hermitian: upper triangle once
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        # union of each kernel's launch intervals: gather, complex128 mat-vec, complex64 mat-vec, rank-1 back-map, model transform
        ms = (ctypes.c_double * NPROF)()
        ms_sum = (ctypes.c_double * NPROF)()    # plain sum of the individual launch spans
        launches = (ctypes.c_int64 * NPROF)()
        lib.scint_profile_end(ms, ms_sum, launches, NPROF)
        rank_rates = [local_etas / elapsed]
        if world > 1:
            # every rank's own rate (its curvatures / its wall time between the two barriers), then the MAX time
            rr = [torch.zeros(1, dtype=torch.float64, device=comm_dev) for _ in range(world)]
            dist.all_gather(rr, torch.tensor([local_etas / elapsed], dtype=torch.float64, device=comm_dev))
            rank_rates = [float(x.item()) for x in rr]
            tt = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return dict(elapsed=elapsed, curves=curves, info=info, fit=fit, mv_bytes=mv_bytes, rank_rates=rank_rates,
                    busy_ms=list(ms), sum_ms=list(ms_sum), launches=list(launches), stats=stats)

    mixed = args.precision == "mixed" and args.objective == "eig"
    if mixed:
        ththmod.sweep_precision("mixed")
    if args.side_stream:
        torch.cuda.set_stream(torch.cuda.Stream())
    head = timed(args.objective, args.steps, args.warmup)

    share_balance = None
    if shard_eta and world > 1:
        # outside the timed region: N, pass counts and status of EVERY curvature (a rank's info covers its own share),
        # so that the line's N_min / N_max / lanczos_steps_mean / failed_etas describe the whole sweep
        from scintools_amd import sweep
        width = -(-neta // world)
        mine = torch.full((3, width), -1, dtype=torch.int64, device=comm_dev)
        li = head["info"]
        if li is not None:
            k = int(li["N"].shape[0])
            mine[0, :k] = torch.from_numpy(li["N"].astype(np.int64)).to(comm_dev)
            mine[1, :k] = torch.from_numpy(li["iters"].astype(np.int64)).to(comm_dev)
            mine[2, :k] = torch.from_numpy(li["status"].astype(np.int64)).to(comm_dev)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        whole = {k: np.zeros(neta, dtype=np.int64) for k in ("N", "iters", "status")}
        for r in range(world):
            idx = sweep.eta_share(neta, world, r)
            pr = parts[r].cpu().numpy()
            for j, k in enumerate(("N", "iters", "status")):
                whole[k][idx] = pr[j, : idx.shape[0]]
        cost = 8.0 * whole["N"] * (whole["N"] + 1.0) * whole["iters"]
        share_balance = {"max_over_mean_matvec_bytes_per_rank": sweep.share_imbalance(cost, world),
                         "rank0_share_of_matvec_bytes": float(cost[sweep.eta_share(neta, world, 0)].sum() / cost.sum())}
        head["info"] = dict(whole, batch=(li or {}).get("batch", 0))

    gathered_equal = None
    if shard_eta and world > 1 and rank == 0:
        # outside the timed region: rank 0 sweeps ALL curvatures alone; the gathered curve must be the same bits
        cs_t = ththmod.conjugate_spectrum(dyns[0], args.npad, tau, 0.0, True)
        alone = ththmod.eval_sweep(cs_t, tau, fd, etas, edges, batch=args.batch)
        gathered_equal = bool(np.array_equal(alone, head["curves"][0], equal_nan=True))

    strong = None
    if (world > 1 and not shard_eta and args.obs_total == 0 and args.obs == 1 and args.objective == "eig" and args.precision == "f64"
            and args.strong_steps > 0):
        # The plain multi-GPU command is weak scaling (one observation per GPU: near-linear by construction).  So that whatever N the
        # driver picks also yields a STRONG-scaling point, the same ranks now split ONE observation's curvatures (--shard eta's
        # path: sweep.sharded_eval_sweep, interleaved shares, one all-gather of float64 per step).  Every rank holds observation 0
        # (the analytic arc of seed 3 is generated on every rank); T_1 is rank 0's own step of the weak region -- the same
        # observation, all curvatures, one GPU, in this very process.
        from scintools_amd import sweep
        d0 = dyns[0] if rank == 0 else ththmod.to_device(dyn, torch.float64)

        def strong_step():
            cs_t = ththmod.conjugate_spectrum(d0, args.npad, tau, 0.0, True)
            return sweep.sharded_eval_sweep(cs_t, tau, fd, etas, edges, batch=args.batch, tol=tol)
        strong_step()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.strong_steps):
            curve = strong_step()
        torch.cuda.synchronize()
        dist.barrier()
        tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_w = float(tt.item()) / args.strong_steps
        if rank == 0:
            t_1 = neta / head["rank_rates"][0]                       # rank 0's own seconds per step of the weak region
            strong = {"what": "--shard eta on observation 0 with the same ranks: one observation's curvatures dealt interleaved "
                              "(etas[rank::world]), one all-gather of float64 per step",
                      "scaling": "strong", "value": neta / t_w, "unit": "eta-points/s", "steps": args.strong_steps, "ms_per_step": 1e3 * t_w,
                      "T1_ms": 1e3 * t_1, "speedup_vs_rank0_alone": t_1 / t_w, "efficiency": t_1 / (world * t_w),
                      "gathered_equals_one_gpu": bool(np.array_equal(curve, head["curves"][0], equal_nan=True)),
                      "ranks_seen": int(dist.get_world_size()), "backend": backend,
                      "note": "efficiency = T_1 / (N x T_N), T_1 = rank 0's own step in the weak-scaling region of this run (same observation, all "
                              "curvatures); the replicated part is the conjugate-spectrum FFT every rank repeats"}

    if rank == 0:
        elapsed, info, fit = head["elapsed"], head["info"], head["fit"]
        eigs = head["curves"][0]
        n_ = info["N"].astype(float)
        # packed gather: one CS read per strict-upper element + the upper-triangle tiles written
        gather_bytes = float(np.sum(8.0 * n_ * (n_ - 1.0) + 8.0 * n_ * (n_ + 1.0))) * args.steps
        # the dominant kernel: the complex128 mat-vec, or -- mixed sweep -- the complex64 one
        mvk = 2 if mixed else 1
        mv_s, ga_s = head["busy_ms"][mvk] / 1e3, head["busy_ms"][0] / 1e3
        alg_bytes = head["stats"][0] if mixed else head["mv_bytes"]
        achieved = alg_bytes / mv_s / 1e9 if mv_s > 0 else 0.0
        ratio, ratio_src, ratio_stale = pmc_traffic_ratio() if not mixed else (None, None, None)
        launches = head["launches"]
        alg_per_launch = alg_bytes / max(1, launches[mvk])
        what = ("theta-theta eigenvalue sweep (Eval_calc loop of single_search)" if args.objective == "eig"
                else "modeler/chisq_calc sweep")
        out = {
            "metric": "eta_curvature_sweep_points_per_sec",
            "value": n_obs_job * neta * args.steps / elapsed,     # --shard eta: one observation's 256 eta, whoever computed them
            "unit": "eta-points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" if not args.dyn_npz else f"synthetic ({os.path.basename(args.dyn_npz)})",
            "library": library_fingerprint(),
            "config": {"workload": f"{size}x{size} dynspec, {neta}-eta {what}, nedge={nedge}, npad={args.npad}, "
                                   + ("ONE observation, curvatures dealt interleaved to the GPUs (sweep.sharded_eval_sweep)" if shard_eta
                                      else f"{args.obs_total} observations dealt round-robin to the GPUs"
                                      if args.obs_total > 0 else f"{args.obs} observation(s) per GPU"),
                       "observations_per_step": n_obs_job, "ranks_per_gpu": ranks_per_gpu,
                       "shard": args.shard, "ranks_seen": int(dist.get_world_size()) if world > 1 else 1,
                       "backend": backend if world > 1 else None,
                       "per_rank_eta_per_s": {"min": min(head["rank_rates"]), "max": max(head["rank_rates"])},
                       "gathered_equals_one_gpu": gathered_equal, "eta_share_balance": share_balance,
                       "collective": (f"all_gather of float64 [{-(-neta // world)}] per rank per step ({backend})" if shard_eta else
                                      f"all_gather of float64 [{per_rank_max}, {neta}] per rank per step ({backend})")
                       if world > 1 else None,
                       "sweep_precision": args.precision if args.objective == "eig" else "f64",
                       "eta_range": "geomspace(0.25, 4.0) * eta_true", "tol": tol,
                       "N_min": int(info["N"].min()), "N_max": int(info["N"].max()),
                       "lanczos_steps_mean": float(info["iters"].mean()),
                       "lanczos_vectors_per_step": lanczos_block()[0],
                       "batch": int(info["batch"]), "failed_etas": int(np.sum(info["status"] != 0)),
                       "eta_fit_over_true": float(fit[0] / eta_true) if np.isfinite(fit[0]) else None},
            "roofline": {"kernel": "pk2_matvec32_kernel" if mixed else lanczos_block()[1],
                         "scope": "rank 0's own launches and bytes" if world > 1 else "the one GPU's launches",
                         "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": alg_per_launch,
                         "traffic": (ratio * alg_per_launch) if ratio else None,
                         "traffic_note": (f"HBM bytes per launch = {ratio:.3f} x algorithmic bytes; ratio measured "
                                          f"with rocprofv3 PMC (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, "
                                          f"separate passes), {ratio_src}") if ratio else ratio_stale,
                         "avg_launch_ms": head["sum_ms"][mvk] / max(1, launches[mvk]), "launches": int(launches[mvk]),
                         "busy_ms": head["busy_ms"][mvk],
                         "timing_note": "achieved = algorithmic bytes / busy_ms; busy_ms = union of this kernel's "
                                        "hipEvent launch intervals on its launch streams (equal to launches x "
                                        "avg_launch_ms when the sweep runs on one stream); tools/rocpd_summary.py "
                                        "computes the same union from a rocprofv3 kernel trace",
                         "algorithmic_bytes_per_step": alg_bytes / args.steps,
                         "share_of_step_time": mv_s / elapsed},
            "gather": {"kernel": "thth_gather_packed_kernel", "bound": "hbm",
                       "achieved_GBs": gather_bytes / ga_s / 1e9 if ga_s > 0 else 0.0,
                       "avg_launch_ms": head["sum_ms"][0] / max(1, launches[0]), "launches": int(launches[0]),
                       "busy_ms": head["busy_ms"][0],
                       "algorithmic_bytes_per_eta": "16 N^2 (one 16-B CS read per strict-upper element + the "
                                                    "upper-triangle tiles written)",
                       "frac": (gather_bytes / ga_s / 1e9 / HBM_PEAK_GBS) if ga_s > 0 else 0.0},
        }
        if mixed:
            out["roofline"]["mixed"] = mixed_summary(head, args.steps, neta * n_obs_job)
            ththmod.sweep_precision("f64")        # every other leg of the line is the float64 library
        if world == 1 and args.objective == "eig" and len(dyns) == 1 and not shard_eta and not mixed and args.mixed_steps > 0:
            try:                                   # a reported leg must never take the headline down
                ththmod.sweep_precision("mixed")
                try:
                    mx = timed("eig", args.mixed_steps, 1)
                finally:
                    ththmod.sweep_precision("f64")
                out["mixed_precision"] = mixed_leg(mx, args.mixed_steps, neta, eigs, out["value"])
            except Exception as exc:
                out["mixed_precision"] = {"error": repr(exc)}
        if world == 1 and args.objective == "eig" and len(dyns) == 1 and not shard_eta and not mixed and not args.headline_only:
            # the same sweep with ONE slot group (scint_sweep_schedule): every mat-vec launch has the
            # GPU to itself, so this is the kernel's own rate; in the headline schedule two groups' launches and the
            # small kernels share the GPU and `achieved` above is bytes / (time any mat-vec launch is in flight)
            lib.scint_sweep_schedule(-1, -1, 1)
            try:
                one = timed("eig", min(args.steps, 3), 1)
            finally:
                lib.scint_sweep_schedule(-1, -1, 0)
            one_s = one["busy_ms"][1] / 1e3
            if one_s > 0:
                out["roofline"]["one_slot_group"] = {
                    "achieved": one["mv_bytes"] / one_s / 1e9, "unit": "GB/s",
                    "frac": one["mv_bytes"] / one_s / 1e9 / HBM_PEAK_GBS,
                    "avg_launch_ms": one["sum_ms"][1] / max(1, one["launches"][1]),
                    "eta_per_s": neta * min(args.steps, 3) / one["elapsed"],
                    "note": "the mat-vec kernel with the GPU to itself: same sweep, one slot group on one stream "
                            "(launches do not overlap; achieved = algorithmic bytes / sum of the launch durations)"}
            one_ga_s = one["busy_ms"][0] / 1e3
            if one_ga_s > 0:
                ga_one = gather_bytes / args.steps * min(args.steps, 3)
                out["gather"]["one_slot_group"] = {
                    "achieved_GBs": ga_one / one_ga_s / 1e9, "frac": ga_one / one_ga_s / 1e9 / HBM_PEAK_GBS,
                    "avg_launch_ms": one["sum_ms"][0] / max(1, one["launches"][0]),
                    "note": "the gather kernel with the GPU to itself (the one-slot-group sweep above: nothing runs beside "
                            "it); `frac` of this object's parent is its rate while the other group's mat-vec and the "
                            "reduce blocks share the GPU with it"}
        legs_ok = world == 1 and args.objective == "eig" and len(dyns) == 1 and not shard_eta and not mixed and not args.headline_only
        if legs_ok and args.share_steps > 0:
            # Strong scaling of --shard eta, as far as ONE GPU can show it: a rank of a W-rank job sweeps etas[R::W] (sweep.eta_share)
            # and nothing else, so T_1 / (W x T_share) is the efficiency the job reaches if the ranks do not disturb each
            # other (they share nothing but the host).  What it exposes is the small-share regime: 32 curvatures for 107
            # slots, no refill, the step is mostly its own low-occupancy tail.  Both ends of the rank range are timed (the
            # shares are balanced in bytes, tests/test_sharding_cpu.py), with the slot groups sweep.share_schedule picks
            # (two, as measured: profiles/r05_rank_share_ab.json).
            try:
                from scintools_amd import sweep
                k = args.share_steps
                t1 = elapsed / args.steps
                pred = {"T1_ms": 1e3 * t1, "note": "predicted efficiency = T1 / (W x slowest timed share), one GPU, unmeasured on a multi-GPU node"}
                for W in (2, 4, 8):
                    if neta < 2 * W:
                        continue
                    per = {}
                    for rk in sorted({0, W - 1}):
                        idx = sweep.eta_share(neta, W, rk)
                        groups = sweep.share_schedule(len(idx), int(info["batch"]))
                        wl.update(main_wl, etas=etas[idx])
                        lib.scint_sweep_schedule(-1, -1, groups)
                        try:
                            r = timed("eig", k, 1)
                        finally:
                            lib.scint_sweep_schedule(-1, -1, 0)
                            wl.update(main_wl)
                        per[f"rank{rk}"] = {"etas": int(len(idx)), "ms_per_step": 1e3 * r["elapsed"] / k, "slot_groups": groups or 2,
                                           "matvec_GBs": r["mv_bytes"] / (r["busy_ms"][1] / 1e3) / 1e9 if r["busy_ms"][1] > 0 else 0.0}
                    slow = max(v["ms_per_step"] for v in per.values()) / 1e3
                    pred[str(W)] = dict(per, efficiency=t1 / (W * slow), eta_per_s=neta / slow)
                out["config"]["predicted_strong_scaling"] = pred
            except Exception as exc:               # a reported leg must never take the headline down
                out["config"]["predicted_strong_scaling"] = {"error": repr(exc)}
        if legs_ok and args.sim_steps > 0 and not args.dyn_npz:
            out["simulation_screen"] = simulation_screen_leg(args, size, neta, nedge, wl, main_wl, timed, ththmod)
            if "value" in out["simulation_screen"]:
                out["simulation_screen"]["ratio_to_headline_input"] = out["simulation_screen"]["value"] / out["value"]
        if world == 1 and args.objective == "chisq" and len(dyns) == 1:
            # the chi^2 objective as the headline region (tools/gpu_run.sh modeler / pmc_modeler): the same object
            out["modeler"] = modeler_objects(head, args.steps, neta, etas, eta_true, 16.0 * R * C, 8.0 * size * size)
        msteps = args.modeler_steps if args.modeler_steps is not None else min(args.steps, 3)
        if world == 1 and args.objective == "eig" and msteps > 0 and len(dyns) == 1:
            mod = timed("chisq", msteps, 1)
            chis = mod["curves"][0]
            out["modeler"] = modeler_objects(mod, msteps, neta, etas, eta_true, 16.0 * R * C, 8.0 * size * size)
            try:                                   # the same objective with the eigenpair iteration on the complex64 copy
                ththmod.sweep_precision("mixed-all")
                try:
                    mxa = timed("chisq", msteps, 1)
                finally:
                    ththmod.sweep_precision("f64")
                got = mxa["curves"][0]
                out["modeler"]["mixed_all"] = {
                    "what": "scint_sweep_precision(2): the eigenPAIR iteration streams the complex64 copy to the eigenvalue rule, "
                            "the vector is finished on the complex128 tiles to the float64 sweep's residual rule; a reported leg, "
                            "not modeler.value",
                    "value": neta * msteps / mxa["elapsed"], "unit": "eta-points/s", "ms_per_step": 1e3 * mxa["elapsed"] / msteps,
                    "speedup_vs_f64": (neta * msteps / mxa["elapsed"]) / out["modeler"]["value"],
                    "lanczos_steps_mean": float(mxa["info"]["iters"].mean()),
                    "failed_etas": int(np.sum(mxa["info"]["status"] != 0)),
                    "max_rel_diff_vs_f64_chisq_curve": float(np.nanmax(np.abs(got - chis) / np.abs(chis)))}
            except Exception as exc:               # a reported leg must never take the line down
                out["modeler"]["mixed_all"] = {"error": repr(exc)}
        if world == 1 and args.objective == "eig" and msteps > 0:
            out["sspec"] = sspec_timing(torch, size, lib)
        if world == 1 and not args.no_cpu_baseline and args.objective == "eig":
            cb, ref_vals = cpu_baseline(dyn, tau, fd, edges, etas, args.cpu_sample, args.npad, args.cpu_reps)
            out["cpu_baseline"] = cb
            pr = port_vs_reference()
            if pr:
                cb["port_vs_reference"] = pr
                cb["sample"] += (f"; the port takes {pr['port_over_reference_time']:.2f} x the time of the unmodified reference's "
                                 f"Eval_calc on the same {pr['size']}^2 conjugate spectrum (values identical, rel. diff "
                                 f"{pr['max_rel_diff']:.1e}; {pr['source']})")
            out["cpu_baseline"]["max_rel_diff_vs_gpu"] = float(
                max(abs(eigs[i] - v) / abs(v) for i, v in ref_vals.items()))
            out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
            nproc = pool_size(args.cpu_pool, nedge)
            if nproc > 0:
                try:
                    out["cpu_baseline_pool"] = cpu_baseline_pool(dyn, tau, fd, edges, etas, nproc, args.npad)
                    out["speedup_vs_cpu_baseline_pool"] = out["value"] / out["cpu_baseline_pool"]["value"]
                except Exception as exc:          # a baseline must never take the benchmark down
                    out["cpu_baseline_pool"] = {"error": repr(exc)}
            if "modeler" in out:
                # one curvature of the modeler objective against the oracle's chisq_calc
                from oracle import thth_oracle
                i = int(np.argmin(np.abs(etas - eta_true)))
                t0 = time.perf_counter()
                ref = thth_oracle.chisq_calc(dyn, thth_oracle.conjugate_spectrum(dyn, args.npad), tau, fd,
                                             etas[i], edges, 1.0)
                dt = time.perf_counter() - t0
                out["modeler"]["parity_sample"] = {"eta_index": i, "rel_diff_vs_oracle_chisq_calc":
                                                   float(abs(chis[i] - ref) / abs(ref)), "oracle_seconds": dt}
                out["modeler"]["cpu_baseline_value"] = 1.0 / dt
            # practical read ceiling of this GPU for the roofline context (a 2 GiB torch.sum; not
            # part of the timed region, not part of the product path)
            try:
                probe = torch.empty(2**28, dtype=torch.float64, device="cuda").normal_()
                for _ in range(2):
                    probe.sum()
                torch.cuda.synchronize()
                tp = time.perf_counter()
                for _ in range(5):
                    probe.sum()
                torch.cuda.synchronize()
                ceil_gbs = probe.numel() * 8 * 5 / (time.perf_counter() - tp) / 1e9
                out["roofline"]["measured_read_ceiling_GBs"] = ceil_gbs
                out["roofline"]["frac_of_measured_read_ceiling"] = achieved / ceil_gbs
                del probe
            except Exception:
                pass
        if strong is not None:
            out["strong"] = strong
        if legs_ok and args.workload_steps > 0:
            # the (f) rows end to end, GPU alone (their CPU samples and parity figures stay behind --workload X)
            if size >= 1024 and size % args.chunk == 0 and size // args.chunk >= 4:
                import argparse as _ap
                wls = {}
                for name in ("fit_thetatheta", "wavefield", "tutorial_fit", "fit_arc"):
                    try:
                        ns = _ap.Namespace(**dict(vars(args), workload=name, steps=args.workload_steps, warmup=1, npad=0, nedge=None))
                        r = run_workload(ns, with_cpu=False)
                        wls[name] = {"seconds": r["value"], "seconds_all": r["seconds_all"], "workload": r["config"]["workload"],
                                     "kernels_busy_share_of_wall": {k: v["busy_share_of_wall"] for k, v in r.get("kernels", {}).items()}}
                        for k in ("chunk_eta_points_per_s", "chunks_per_s", "parity", "betaeta"):
                            if k in r:
                                wls[name][k] = r[k]
                    except Exception as exc:           # a reported leg must never take the headline down
                        wls[name] = {"error": repr(exc)}
                out["workloads"] = wls
            else:
                out["workloads"] = {"note": f"skipped: --size {size} gives fewer than 4 x 4 chunks of --chunk {args.chunk}"}
        # The numbers a reader of the END of this line needs, as flat scalars (the nested objects above hold what each means)
        def _get(*path):
            o = out
            for k in path:
                if not isinstance(o, dict) or k not in o:
                    return None
                o = o[k]
            return o
        out["tail"] = "flat copies of the figures the nested objects above define"
        out["matvec_frac_of_hbm_peak"] = _get("roofline", "frac")
        out["gather_frac_in_sweep"] = _get("gather", "frac")
        out["gather_frac_alone"] = _get("gather", "one_slot_group", "frac")
        out["sim_screen_eta_per_s"] = _get("simulation_screen", "value")
        out["sim_screen_passes"] = _get("simulation_screen", "lanczos_steps_mean")
        out["modeler_eta_per_s"] = _get("modeler", "value")
        out["mixed_eta_per_s"] = _get("mixed_precision", "value")
        out["strong_scaling_pred_8"] = _get("config", "predicted_strong_scaling", "8", "efficiency")
        out["strong_scaling_measured"] = _get("strong", "efficiency")
        out["sspec_ms"] = _get("sspec", f"{size}x{size}", "ms")
        out["sspec_frac_of_hbm_peak"] = _get("sspec", f"{size}x{size}", "frac_of_hbm_peak")
        out["sspec_traffic_over_algorithmic"] = _get("sspec", f"{size}x{size}", "roofline", "traffic_over_algorithmic")
        for name in ("fit_thetatheta", "wavefield", "tutorial_fit", "fit_arc"):
            out[f"workload_{name}_s"] = _get("workloads", name, "seconds")
        out["cpu_baseline_eta_per_s"] = _get("cpu_baseline", "value")
        out["eta_per_s"] = out["value"]
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
