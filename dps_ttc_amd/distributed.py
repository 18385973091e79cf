"""Particle sharding across the GPUs of one node: one process per GPU, RCCL over xGMI.

Particles are independent between select points, so the DPS step itself has no
collective.  The only exchanges are:
  * best-of-N:   all-gather of the per-particle scores [N/G] (a few hundred bytes,
                 latency-bound), identical first-min argmin on every rank, and -- for
                 the per-step search -- a broadcast of the winning particle (786 KB)
                 from its owner over the direct xGMI links;
  * resampling:  all-gather of scores, identical multinomial draw on every rank from a
                 shared-seed host generator, then an all-gather of particle states.
`torch.distributed` is plumbing: backend "nccl" is RCCL on ROCm; the CPU tests run the
same control flow over "gloo".
"""
import torch
import torch.distributed as dist

from . import kernels


def _world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def shard_range(n_total, rank=None, world=None):
    """Contiguous block of particles owned by `rank` (N=256 -> 32 per GPU on 8 GPUs)."""
    r, w = _world()
    rank = r if rank is None else rank
    world = w if world is None else world
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_scores(scores_local):
    """[n_local] -> [world * n_local] on every rank (equal shards), via ncclAllGather/RCCL."""
    rank, world = _world()
    scores_local = scores_local.contiguous()
    if world == 1:
        return scores_local
    out = torch.empty(world * scores_local.numel(), dtype=scores_local.dtype, device=scores_local.device)
    dist.all_gather_into_tensor(out, scores_local)
    return out


def first_argmin(scores):
    """torch.argmin semantics (first minimum, NaN wins); HIP on the device, host control logic for gloo tests."""
    if scores.is_cuda:
        return kernels.argmin(scores)
    return torch.argmin(scores)


def global_best_of_n(scores_local, particles_local):
    """Final best-of-N over all ranks' particles (best_of_n_simple.py:32-40 moved on device).

    Returns (winner [1,C,H,W] on every rank, global index, all scores).  The index is read on
    the host once per trajectory (end of the 1000-step loop), not per step."""
    rank, world = _world()
    all_scores = gather_scores(scores_local)
    best = int(first_argmin(all_scores))
    n_local = scores_local.numel()
    owner, local = divmod(best, n_local)
    winner = particles_local[local:local + 1].contiguous() if owner == rank \
        else torch.empty_like(particles_local[:1]).contiguous()
    if world > 1:
        dist.broadcast(winner, src=owner)
    return winner, best, all_scores


class GlobalSelect:
    """`SearchDDPM.global_select` hook: per-step best-of-N across ranks (gaussian_diffusion.py:626-633
    of the reference, generalised to a sharded particle set).  No host sync: every rank broadcasts its
    local champion slot, the winner is picked on the device from the gathered (min, rank) table."""

    def __call__(self, costs_local, particles_local):
        rank, world = _world()
        if world == 1:
            return kernels.replicate(particles_local, kernels.argmin(costs_local))
        n = particles_local.shape[0]
        local_best = first_argmin(costs_local)
        champ = particles_local[local_best].unsqueeze(0).contiguous() if not particles_local.is_cuda else \
            kernels.replicate(particles_local, local_best, n_out=1)
        mins = gather_scores(costs_local[local_best].reshape(1))          # [world]
        champs = [torch.empty_like(champ) for _ in range(world)]
        dist.all_gather(champs, champ)                                    # world x 786 KB over xGMI
        win_rank = first_argmin(mins)                                     # lowest rank wins ties = first-min rule
        stacked = torch.cat(champs, dim=0)
        if stacked.is_cuda:
            return kernels.replicate(stacked, win_rank, n_out=n)
        return stacked[int(win_rank)].unsqueeze(0).repeat(n, 1, 1, 1)


def resample_ids(scores_local, temperature, generator):
    """Identical multinomial ids on every rank from the gathered scores and a shared-seed HOST generator
    (gaussian_diffusion.py:689-698: w = exp(-d / T), torch.multinomial with replacement)."""
    all_scores = gather_scores(scores_local).float().cpu()
    w = torch.exp(-all_scores / temperature)
    if w.max() == w.min():
        return None
    return torch.multinomial(w, all_scores.numel(), replacement=True, generator=generator)


def resample_particles(particles_local, ids_global):
    """Fetch the resampled particle set: all-gather of states, then a HIP gather of this rank's slots."""
    rank, world = _world()
    n_local = particles_local.shape[0]
    if world == 1:
        pool = particles_local
    else:
        pool = torch.empty((world * n_local,) + tuple(particles_local.shape[1:]), dtype=particles_local.dtype,
                           device=particles_local.device)
        dist.all_gather_into_tensor(pool, particles_local.contiguous())
    mine = ids_global[rank * n_local:(rank + 1) * n_local]
    if pool.is_cuda:
        return kernels.gather(pool, mine)
    return pool[mine]
