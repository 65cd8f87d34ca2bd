#!/usr/bin/env python
"""Headline benchmark: eta-curvature-sweep points/s on a 4096x4096 dynamic spectrum.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 4096] [--neta 256]

One *step* is one pass of the hot path over one observation per GPU: device-resident
real dynamic spectrum [4096, 4096] -> conjugate spectrum (2-D FFT) -> for each of 256
curvatures: theta-theta gather (nedge=4096) + dominant 'LA' eigenvalue -> eigs[256] on the
host -> parabola fit (the body of ththmod.single_search, ththmod.py:773-859).  This is
BASELINE.json configs[2], the configuration the metric is quoted on.

Multi-GPU (launched by torch.distributed.run, one rank per GPU, RCCL): the path shards by
observation -- every rank sweeps its own observation (different seed) and the eigenvalue
curves are all-gathered at the end of each step.  Per-GPU work is fixed: weak scaling;
`value` is the whole-job eta-points/s.

Besides the contract fields the JSON line carries
  roofline      the dominant kernel (eigen mat-vec on the Hermitian tile-packed matrix):
                algorithmic bytes 8 N (N+1) per mat-vec per eta -- every upper-triangle
                element once; SURVEY.md 8d's 16 N^2 assumed the full matrix -- summed over
                every Lanczos step of the timed region, divided by the summed hipEvent time
                of that kernel's launches;
  cpu_baseline  the NumPy/SciPy oracle (a restatement of the reference) timed on this host
                on a 3-eta sample of the same workload (rank 0, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def pmc_traffic_ratio():
    """HBM bytes / algorithmic bytes of the dominant kernel, from the committed PMC summary
    (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same script, gfx950
    FETCH x2 correction -- tools/pmc_summary.py).  None if no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*_pmc_summary.json")))
    if not files:
        return None, None
    with open(files[-1]) as fh:
        k = json.load(fh)["kernels"].get("scint::pk_matvec_kernel", {})
    return k.get("traffic_over_algorithmic"), os.path.relpath(files[-1], REPO)


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
    ap.add_argument("--obs", type=int, default=1,
                    help="observations swept per GPU per step (BASELINE config 4: --size 2048 --obs 8 on 8 GPUs)")
    ap.add_argument("--objective", choices=["eig", "chisq"], default="eig",
                    help="eig: Eval_calc sweep of single_search (headline); chisq: modeler/chisq_calc sweep")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=3, help="etas timed on the CPU oracle")
    ap.add_argument("--cpu-pool", type=int, default=8,
                    help="also time the oracle eta-parallel with multiprocessing.Pool(P), one BLAS thread "
                         "per worker (how a user parallelises the reference, dynspec.py:1715-1719); 0 = skip")
    return ap.parse_args()


def make_workload(size, neta, nedge, seed, npad=0):
    from scintools_amd.synth import arc_dynspec
    from scintools_amd.ththmod import fft_axis
    dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=seed, nimg=64)
    dyn -= dyn.mean()                # as Dynspec.fit_thetatheta hands chunks over (dynspec.py:1692)
    fd = fft_axis(times, 1000.0, npad)     # s -> mHz
    tau = fft_axis(freqs, 1.0, npad)       # MHz -> us
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, nedge)
    etas = np.geomspace(0.25, 4.0, neta) * eta_true
    return dyn, freqs, times, fd, tau, edges, etas, eta_true


def cpu_baseline(dyn, tau, fd, edges, etas, nsample, npad=0):
    """Oracle (port of the reference) on a bounded sample: `nsample` curvatures spread over
    the sweep; the FFT is done once and not counted (it is amortised over 256 etas)."""
    from oracle import thth_oracle
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    CS = thth_oracle.conjugate_spectrum(dyn, npad)
    idx = np.unique(np.linspace(0, len(etas) - 1, nsample + 2).astype(int)[1:-1])
    t0 = time.perf_counter()
    vals = [thth_oracle.Eval_calc(CS, tau, fd, etas[i], edges) for i in idx]
    dt = time.perf_counter() - t0
    return {"value": len(idx) / dt, "unit": "eta-points/s", "cores": int(cores), "kind": "port",
            "sample": f"oracle Eval_calc (NumPy gather + ARPACK eigsh) on {len(idx)} of {len(etas)} etas "
                      f"(indices {idx.tolist()}) of the same {dyn.shape[0]}x{dyn.shape[1]} workload, "
                      f"{dt:.1f} s; BLAS threads as shipped; CS FFT excluded"}, dict(zip(idx.tolist(), vals))


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


def cpu_baseline_pool(dyn, tau, fd, edges, etas, nproc, npad=0):
    """eta-parallel oracle: Pool(nproc).map over 2*nproc curvatures spread over the sweep."""
    import multiprocessing as mp
    import tempfile
    from oracle import thth_oracle
    CS = thth_oracle.conjugate_spectrum(dyn, npad)
    tmp = tempfile.mkdtemp(prefix="scint_bench_")
    path = os.path.join(tmp, "w_cs.npy")
    np.save(path, CS)
    np.savez(path.replace("_cs.npy", "_meta.npz"), tau=tau, fd=fd, etas=etas, edges=edges)
    idx = np.unique(np.linspace(0, len(etas) - 1, 2 * nproc + 2).astype(int)[1:-1]).tolist()
    ctx = mp.get_context("spawn")
    try:
        with ctx.Pool(nproc) as pool:
            pool.map(_pool_worker, [(path, idx[0])] * nproc)          # warm the workers (imports, page cache)
            t0 = time.perf_counter()
            res = pool.map_async(_pool_worker, [(path, i) for i in idx]).get(timeout=600)
            dt = time.perf_counter() - t0
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    return {"value": len(idx) / dt, "unit": "eta-points/s", "cores": int(nproc), "kind": "port",
            "sample": f"oracle Eval_calc over multiprocessing.Pool({nproc}) (spawn, 1 BLAS thread per worker) on "
                      f"{len(idx)} of {len(etas)} etas, {dt:.1f} s"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from scintools_amd import _lib, ththmod
    from scintools_amd.device import require_gpu

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("SCINT_BENCH_BACKEND", "nccl")   # "gloo": 2 ranks on one GPU (tests only)
    if world > 1:
        dev_index = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)
    else:
        torch.cuda.set_device(0)
    comm_dev = "cuda" if backend == "nccl" else "cpu"
    require_gpu()
    lib = _lib.load()
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    size, neta = args.size, args.neta
    nedge = args.nedge or size
    dyn, freqs, times, fd, tau, edges, etas, eta_true = make_workload(size, neta, nedge, seed=3 + rank, npad=args.npad)
    dyn_t = ththmod.to_device(dyn, torch.float64)      # resident in HBM before the clock starts
    extra_obs = [ththmod.to_device(make_workload(size, neta, nedge, seed=1000 + 97 * rank + k, npad=args.npad)[0], torch.float64)
                 for k in range(1, args.obs)]
    gathered = [torch.empty(neta, dtype=torch.float64, device=comm_dev) for _ in range(world)]

    def step():
        for other in extra_obs:                      # further observations of this rank's share
            cs_o = ththmod.conjugate_spectrum(other, args.npad, tau, 0.0, True)
            e_o = ththmod.eval_sweep(cs_o, tau, fd, etas, edges, batch=args.batch)
            ththmod.fit_eig_peak(etas, e_o, 0.1)
        # body of single_search (ththmod.py:773-859): CS once, the eta loop, the peak fit
        cs_t = ththmod.conjugate_spectrum(dyn_t, args.npad, tau, 0.0, True)
        if args.objective == "chisq":
            # the other objective of BASELINE config 3: chisq_calc(modeler(...)) for every eta
            chis, info = ththmod.chisq_sweep(dyn_t, cs_t, tau, fd, etas, edges, 1.0, return_info=True)
            if world > 1:
                dist.all_gather(gathered, torch.from_numpy(chis).to(comm_dev))
            return chis, info, (etas[np.nanargmin(chis)], np.nan, None)
        eigs, info = ththmod.eval_sweep(cs_t, tau, fd, etas, edges, batch=args.batch, return_info=True)
        if world > 1:
            dist.all_gather(gathered, torch.from_numpy(eigs).to(comm_dev))
        fit = ththmod.fit_eig_peak(etas, eigs, 0.1)
        return eigs, info, fit

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    lib.scint_profile_begin()
    t0 = time.perf_counter()
    alg_bytes = 0.0
    for _ in range(args.steps):
        eigs, info, fit = step()
        n_ = info["N"].astype(float)
        alg_bytes += float(np.sum(8.0 * n_ * (n_ + 1.0) * info["iters"])) * args.obs   # Hermitian: upper triangle once
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ms = (ctypes.c_double * 2)()        # union of each kernel's launch intervals (two streams overlap)
    ms_sum = (ctypes.c_double * 2)()    # plain sum of the individual launch spans
    launches = (ctypes.c_int64 * 2)()
    lib.scint_profile_end(ms, ms_sum, launches)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        n_ = info["N"].astype(float)
        # packed gather: one CS read per strict-upper element + the upper-triangle tiles written
        gather_bytes = float(np.sum(8.0 * n_ * (n_ - 1.0) + 8.0 * n_ * (n_ + 1.0))) * args.steps
        mv_s = ms[1] / 1e3
        achieved = alg_bytes / mv_s / 1e9 if mv_s > 0 else 0.0
        ratio, ratio_src = pmc_traffic_ratio()
        alg_per_launch = alg_bytes / max(1, launches[1])
        out = {
            "metric": "eta_curvature_sweep_points_per_sec",
            "value": world * args.obs * neta * args.steps / elapsed,
            "unit": "eta-points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": (f"{size}x{size} dynspec, {neta}-eta theta-theta eigenvalue sweep "
                                    f"(Eval_calc loop of single_search), nedge={nedge}, npad={args.npad}, "
                                    f"one observation per GPU") if args.objective == "eig" else
                                   (f"{size}x{size} dynspec, {neta}-eta modeler/chisq_calc sweep, nedge={nedge}, "
                                    f"npad={args.npad}, one observation per GPU"),
                       "observations_per_gpu_per_step": args.obs,
                       "eta_range": "geomspace(0.25, 4.0) * eta_true", "tol": ththmod.DEFAULT_TOL,
                       "N_min": int(info["N"].min()), "N_max": int(info["N"].max()),
                       "lanczos_steps_mean": float(info["iters"].mean()),
                       "batch": int(info["batch"]), "failed_etas": int(np.sum(info["status"] != 0)),
                       "eta_fit_over_true": float(fit[0] / eta_true) if np.isfinite(fit[0]) else None},
            "roofline": {"kernel": "pk_matvec_kernel", "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": alg_per_launch,
                         "traffic": (ratio * alg_per_launch) if ratio else None,
                         "traffic_note": (f"HBM bytes per launch = {ratio:.3f} x algorithmic bytes; ratio measured "
                                          f"with rocprofv3 PMC (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, "
                                          f"separate passes), {ratio_src}") if ratio else None,
                         "avg_launch_ms": ms_sum[1] / max(1, launches[1]), "launches": int(launches[1]),
                         "busy_ms": ms[1],
                         "timing_note": "achieved = algorithmic bytes / busy_ms, busy_ms = union of this kernel's "
                                        "hipEvent launch intervals (the sweep runs two streams whose launches "
                                        "overlap); avg_launch_ms = mean individual launch span, the figure a "
                                        "rocprofv3 kernel trace averages",
                         "algorithmic_bytes_per_step": alg_bytes / args.steps,
                         "share_of_step_time": mv_s / elapsed},
            "gather": {"kernel": "thth_gather_packed_kernel", "achieved_GBs": gather_bytes / (ms[0] / 1e3) / 1e9
                       if ms[0] > 0 else 0.0, "avg_launch_ms": ms_sum[0] / max(1, launches[0]),
                       "launches": int(launches[0]),
                       "frac": (gather_bytes / (ms[0] / 1e3) / 1e9 / HBM_PEAK_GBS) if ms[0] > 0 else 0.0},
        }
        if world == 1 and not args.no_cpu_baseline and args.objective == "eig":
            cb, ref_vals = cpu_baseline(dyn, tau, fd, edges, etas, args.cpu_sample, args.npad)
            out["cpu_baseline"] = cb
            out["cpu_baseline"]["max_rel_diff_vs_gpu"] = float(
                max(abs(eigs[i] - v) / abs(v) for i, v in ref_vals.items()))
            out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
            if args.cpu_pool > 0:
                try:
                    out["cpu_baseline_pool"] = cpu_baseline_pool(dyn, tau, fd, edges, etas, args.cpu_pool, args.npad)
                    out["speedup_vs_cpu_baseline_pool"] = out["value"] / out["cpu_baseline_pool"]["value"]
                except Exception as exc:          # a baseline must never take the benchmark down
                    out["cpu_baseline_pool"] = {"error": repr(exc)}
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
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
