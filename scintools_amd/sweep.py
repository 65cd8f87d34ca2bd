"""Multi-GPU sharding of the eta sweep: one process per GPU, ``torch.distributed``
(backend "nccl" = RCCL over xGMI on the node; "gloo" in the CPU tests).

The path shards embarrassingly -- every curvature and every observation is independent
(the reference maps chunks over a pool, dynspec.py:1715-1719) -- so there is no data-path
collective: each rank computes its share and the only communication is one all-gather of
float64 eigenvalues at the end (256 eta -> 2 KiB in total).

Two partitionings:
  * :func:`sharded_eval_sweep` -- one observation, contiguous eta blocks per rank.  Every
    rank holds the conjugate spectrum (each rank FFTs the same dynspec locally: cheaper than
    broadcasting a 0.25-4 GiB complex plane).
  * :func:`sharded_observations` -- a batch of observations dealt round-robin to the ranks,
    each rank running whole sweeps (BASELINE config 4; what ``bench.py --gpus N`` times).

``local_fn`` is the per-rank compute; it defaults to the HIP path and exists so that the CPU
tests can exercise the sharding/gather logic with the oracle in its place.
"""
import numpy as np
import torch
import torch.distributed as dist


def block_bounds(n, world, rank):
    """Contiguous block [lo, hi) of `n` items for `rank` (sizes differ by at most one)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _device_for_backend(group=None):
    backend = dist.get_backend(group)
    if str(backend) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _all_gather_blocks(local, n_total, group=None):
    """All-gather variable-length float64 blocks laid out by block_bounds -> full array."""
    world = dist.get_world_size(group)
    dev = _device_for_backend(group)
    width = -(-n_total // world)
    buf = torch.full((width,), float("nan"), dtype=torch.float64, device=dev)
    buf[: local.shape[0]] = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64)).to(dev)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    full = np.empty(n_total, dtype=np.float64)
    for r in range(world):
        lo, hi = block_bounds(n_total, world, r)
        full[lo:hi] = out[r][: hi - lo].cpu().numpy()
    return full


def sharded_eval_sweep(CS, tau, fd, etas, edges, group=None, local_fn=None, **kw):
    """Eigenvalue curve of ONE observation, eta range split across the ranks.
    Returns the full curve on every rank (identical to the single-process result: each eta
    is computed by exactly one rank with the same kernels, so not a bit changes)."""
    if local_fn is None:
        from .ththmod import eval_sweep as local_fn
    etas = np.asarray(etas, dtype=float)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return np.asarray(local_fn(CS, tau, fd, etas, edges, **kw))
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = block_bounds(etas.shape[0], world, rank)
    local = np.asarray(local_fn(CS, tau, fd, etas[lo:hi], edges, **kw)) if hi > lo else np.empty(0)
    return _all_gather_blocks(local, etas.shape[0], group)


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
