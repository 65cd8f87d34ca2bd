"""Multi-GPU sharding of the eta sweep: one process per GPU, ``torch.distributed``
(backend "nccl" = RCCL over xGMI on the node; "gloo" in the CPU tests).

The path shards embarrassingly -- every curvature and every observation is independent
(the reference maps chunks over a pool, dynspec.py:1715-1719) -- so there is no data-path
collective: each rank computes its share and the only communication is one all-gather of
float64 eigenvalues at the end (256 eta -> 2 KiB in total).

Two partitionings:
  * :func:`sharded_eval_sweep` -- one observation, its curvatures dealt to the ranks INTERLEAVED
    (rank r takes etas[r::world], :func:`eta_share`).  The cost of a curvature is bytes x passes
    = 8 N (N + 1) x Lanczos passes, and both vary smoothly along a sweep (the crop takes N from
    4095 down to 2447 across the headline sweep): contiguous blocks -- what this module dealt
    until round 4 -- differ by up to 2.5x in bytes per rank and cap the strong-scaling efficiency
    at 0.86 for 2, 4 and 8 ranks; interleaved shares are within 1 % of each other
    (:func:`share_imbalance`, tests/test_sharding_cpu.py).  The reference never had the
    problem: its ``pool.map`` deals whole chunks of equal shape (dynspec.py:1706-1723).  Every
    rank holds the conjugate spectrum (each rank FFTs the same dynspec locally: cheaper than
    broadcasting a 0.25-4 GiB complex plane).
  * :func:`sharded_observations` -- a batch of observations dealt round-robin to the ranks,
    each rank running whole sweeps (BASELINE config 4; what ``bench.py --gpus N`` times).

  * :func:`sharded_chunks` -- the fitting chunks of ``Dynspec.fit_thetatheta`` dealt
    round-robin; what the reference does with ``pool.map`` (dynspec.py:1715-1719).
  * :func:`gpu_pool` -- a ``multiprocessing`` pool whose workers each bind one GPU, for callers
    that keep the reference's ``fit_thetatheta(pool=...)`` idiom instead of ``torchrun``.

``local_fn`` is the per-rank compute; it defaults to the HIP path and exists so that the CPU
tests can exercise the sharding/gather logic with the oracle in its place (the HIP path under
ranks is covered by ``tests/test_gpu_multirank.py``).
"""
import multiprocessing as mp

import numpy as np
import torch
import torch.distributed as dist


def block_bounds(n, world, rank):
    """Contiguous block [lo, hi) of `n` items for `rank` (sizes differ by at most one)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def eta_share(n, world, rank):
    """Indices of the `n` curvatures of one sweep that `rank` computes: every world-th one, starting at `rank`.
    Neighbouring curvatures cost nearly the same (N and the pass count vary smoothly with eta), so interleaved
    shares carry equal work whatever the cost profile is; sizes differ by at most one."""
    return np.arange(rank, n, world)


def share_schedule(n_share, batch):
    """Slot groups for a rank's share of a sweep (argument of ``scint_sweep_schedule``: 0 = the library's default of two
    groups on two streams, 1 = one group).  Measured on one MI355X with the headline sweep's shares etas[R::W] (round 5, call 1,
    profiles/r05_rank_share_ab.json): two groups win at every share size, also where a share no longer fills the resident
    slots once -- 32 curvatures (W = 8): 22.9 ms per share with two groups against 26.0 with one (predicted efficiency 0.908 /
    0.799); 64 (W = 4): 0.970 / 0.920; 128 (W = 2): 0.971 / 0.946.  The second group's launches fall into the first one's
    check / reduce / refill gaps whatever the share's size, so the rule is: keep the default.  (One group is only forced for
    shares too small to split: fewer than four curvatures, where run_sweep does the same.)"""
    return 1 if n_share < 4 else 0


def share_imbalance(cost, world, shares=eta_share):
    """max over ranks / mean over ranks of the summed `cost` (one number per curvature, e.g. 8 N (N + 1) x passes)
    under the partition `shares(n, world, rank)`: 1.0 is perfect, and 1 / it bounds the strong-scaling efficiency."""
    cost = np.asarray(cost, dtype=float)
    per_rank = np.array([cost[shares(cost.shape[0], world, r)].sum() for r in range(world)])
    return float(per_rank.max() / per_rank.mean())


def _device_for_backend(group=None):
    backend = dist.get_backend(group)
    if str(backend) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _all_gather_shares(local, n_total, group=None):
    """All-gather the ranks' float64 shares (laid out by eta_share) -> full array, every value where its curvature is."""
    world = dist.get_world_size(group)
    dev = _device_for_backend(group)
    width = -(-n_total // world)
    buf = torch.full((width,), float("nan"), dtype=torch.float64, device=dev)
    buf[: local.shape[0]] = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64)).to(dev)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    full = np.empty(n_total, dtype=np.float64)
    for r in range(world):
        idx = eta_share(n_total, world, r)
        full[idx] = out[r][: idx.shape[0]].cpu().numpy()
    return full


def sharded_eval_sweep(CS, tau, fd, etas, edges, group=None, local_fn=None, return_info=False, **kw):
    """Eigenvalue curve of ONE observation, its curvatures dealt interleaved to the ranks (:func:`eta_share`).
    Returns the full curve on every rank (identical to the single-process result: each eta
    is computed by exactly one rank with the same kernels, and a curvature's arithmetic does not depend on
    which others share its launch, so not a bit changes).  With ``return_info`` also this rank's info dict
    (N, Lanczos steps, status of ITS share, plus ``"eta_index"``: which curvatures those are; None for an
    empty share) as ``local_fn(..., return_info=True)`` reports it."""
    if local_fn is None:
        from .ththmod import eval_sweep as local_fn
    etas = np.asarray(etas, dtype=float)
    if return_info:
        kw = dict(kw, return_info=True)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    mine = eta_share(etas.shape[0], world, dist.get_rank(group) if world > 1 else 0)
    info = None
    if mine.shape[0] > 0:
        local = local_fn(CS, tau, fd, etas[mine], edges, **kw)
        if return_info:
            local, info = local
            info = dict(info, eta_index=mine)
        local = np.asarray(local)
    else:
        local = np.empty(0)
    full = local if world == 1 else _all_gather_shares(local, etas.shape[0], group)
    return (full, info) if return_info else full


def sharded_observations(n_obs, sweep_fn, neta, group=None):
    """Run ``sweep_fn(i) -> eigs[neta]`` for the observations this rank owns (round-robin)
    and all-gather the curves: returns eigs[n_obs, neta] on every rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return np.stack([np.asarray(sweep_fn(i), dtype=float) for i in range(n_obs)])
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = list(range(rank, n_obs, world))
    per_rank = -(-n_obs // world)
    dev = _device_for_backend(group)
    buf = torch.full((per_rank, neta), float("nan"), dtype=torch.float64, device=dev)
    for slot, i in enumerate(mine):
        buf[slot] = torch.from_numpy(np.asarray(sweep_fn(i), dtype=np.float64)).to(dev)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    full = np.empty((n_obs, neta), dtype=np.float64)
    for r in range(world):
        for slot, i in enumerate(range(r, n_obs, world)):
            full[i] = out[r][slot].cpu().numpy()
    return full


def sharded_chunks(n_items, rows_fn, width, group=None):
    """Deal `n_items` independent work items (fitting chunks) round-robin to the ranks.

    ``rows_fn(indices) -> float64 [len(indices), 2 + width]`` computes this rank's items in one
    call (so that it can batch them on its GPU); every rank gets the full
    ``[n_items, 2 + width]`` table back (one all-gather).  Without an initialised process group
    it is a plain call."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return np.asarray(rows_fn(list(range(n_items))), dtype=np.float64)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = list(range(rank, n_items, world))
    per_rank = -(-n_items // world)
    dev = _device_for_backend(group)
    buf = torch.full((per_rank, 2 + width), float("nan"), dtype=torch.float64, device=dev)
    if mine:
        rows = np.ascontiguousarray(rows_fn(mine), dtype=np.float64)
        buf[: len(mine)] = torch.from_numpy(rows).to(dev)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    full = np.empty((n_items, 2 + width), dtype=np.float64)
    for r in range(world):
        for slot, i in enumerate(range(r, n_items, world)):
            full[i] = out[r][slot].cpu().numpy()
    return full


def _bind_worker(counter):
    """Pool initializer: worker k drives GPU k (mod the number of visible GPUs)."""
    with counter.get_lock():
        k = counter.value
        counter.value += 1
    torch.cuda.set_device(k % max(1, torch.cuda.device_count()))


def gpu_pool(n_workers=None):
    """``multiprocessing.Pool`` (spawn) with one worker per GPU, for the reference's
    ``Dynspec.fit_thetatheta(pool=pool)`` / ``pool.map(thth.single_search, pars)`` idiom: every
    worker runs the HIP path on its own device.  Close it like any pool."""
    ctx = mp.get_context("spawn")
    n = int(n_workers or max(1, torch.cuda.device_count()))
    return ctx.Pool(n, initializer=_bind_worker, initargs=(ctx.Value("i", 0),))
