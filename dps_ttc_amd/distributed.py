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
same control flow over "gloo".  (SURVEY 8b sketched four `dpsx_comm_*` C symbols wrapping
RCCL; the host side of this path is Python, which already owns the process group, the
streams and the tensors, so a second communicator inside libdpsx would only duplicate
the rendezvous -- the collectives stay here, the kernels stay in the library.)

Global particle order is rank-major: rank r's particles follow rank r-1's.  Shards may be
unequal or empty (`counts`); every rank must enter every function below.
"""
import torch
import torch.distributed as dist

from . import kernels


def _world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def _solo():
    """True when there is nobody to talk to: no process group.  A process group of ONE rank still takes the collective
    path -- the exchanges then run through the backend (RCCL on a GPU) as they would with eight ranks, which is how the
    one-GPU boxes of this pool exercise the real backend (tests/test_driver_gpu.py::test_rccl_single_rank_collectives)."""
    return not (dist.is_available() and dist.is_initialized())


def shard_range(n_total, rank=None, world=None):
    """Contiguous block of particles owned by `rank` (N=256 -> 32 per GPU on 8 GPUs)."""
    r, w = _world()
    rank = r if rank is None else rank
    world = w if world is None else world
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_counts(n_total, world=None):
    """[particles owned by rank r for r in range(world)] -- what every rank can compute without talking."""
    world = _world()[1] if world is None else world
    return [hi - lo for lo, hi in (shard_range(n_total, r, world) for r in range(world))]


def exchange_counts(n_local, device):
    """Shard sizes of all ranks when they cannot be derived (one tiny all-gather + one host read)."""
    rank, world = _world()
    if _solo():
        return [int(n_local)]
    mine = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    out = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, mine)
    return [int(v) for v in out.tolist()]


def gather_scores(scores_local, counts=None):
    """[n_local] -> [sum(counts)] on every rank, rank-major, via ncclAllGather / RCCL.

    counts: per-rank shard sizes.  None = exchange them first (robust default, one host read);
    pass `shard_counts(...)` or `[n] * world` when they are known by construction (no sync).
    Unequal shards are padded with +inf to the largest one for the collective and compacted after."""
    rank, world = _world()
    scores_local = scores_local.contiguous()
    if _solo():
        return scores_local
    if counts is None:
        counts = exchange_counts(scores_local.numel(), scores_local.device)
    if len(counts) != world or counts[rank] != scores_local.numel():
        raise ValueError(f"shard sizes {counts} do not describe rank {rank}'s {scores_local.numel()} scores")
    width = max(counts)
    if width == 0:
        return scores_local
    if min(counts) == width:
        out = torch.empty(world * width, dtype=scores_local.dtype, device=scores_local.device)
        dist.all_gather_into_tensor(out, scores_local)
        return out
    padded = torch.full((width,), float("inf"), dtype=scores_local.dtype, device=scores_local.device)
    padded[:scores_local.numel()] = scores_local
    out = torch.empty(world * width, dtype=scores_local.dtype, device=scores_local.device)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[r * width:r * width + counts[r]] for r in range(world)])


def first_argmin(scores):
    """torch.argmin semantics (first minimum, NaN wins); HIP on the device, host control logic for gloo tests."""
    if scores.is_cuda:
        return kernels.argmin(scores)
    return torch.argmin(scores)


def locate(index, counts):
    """global rank-major particle index -> (owner rank, local index)"""
    for r, c in enumerate(counts):
        if index < c:
            return r, index
        index -= c
    raise IndexError("particle index beyond the sharded set")


def global_best_of_n(scores_local, particles_local, counts=None):
    """Final best-of-N over all ranks' particles (best_of_n_simple.py:32-40 moved on device).

    Returns (winner [1,C,H,W] on every rank, global rank-major index, all scores).  The index is read on
    the host once per trajectory (end of the 1000-step loop), not per step.  A rank may hold no particles
    (`particles_local` of shape [0, C, H, W]): it still enters the collectives and receives the winner."""
    rank, world = _world()
    if counts is None:
        counts = exchange_counts(scores_local.numel(), scores_local.device)
    all_scores = gather_scores(scores_local, counts)
    if all_scores.numel() == 0:
        raise ValueError("best-of-N over an empty particle set")
    best = int(first_argmin(all_scores))
    owner, local = locate(best, counts)
    if owner == rank:
        winner = particles_local[local:local + 1].contiguous()
    else:
        winner = torch.empty((1,) + tuple(particles_local.shape[1:]), dtype=particles_local.dtype,
                             device=particles_local.device)
    if not _solo():
        dist.broadcast(winner, src=owner)
    return winner, best, all_scores


_ZEROS2, _OFFSETS = {}, {}


def _offsets(counts, device):
    """exclusive prefix sums of the shard sizes on the device (uploaded once per (counts, device))"""
    key = (tuple(int(c) for c in counts), str(device))
    if key not in _OFFSETS:
        acc, out = 0, []
        for c in key[0]:
            out.append(acc)
            acc += c
        _OFFSETS[key] = torch.tensor(out, dtype=torch.int64, device=device)
    return _OFFSETS[key]


def _zeros2(device):
    key = str(device)
    if key not in _ZEROS2:
        _ZEROS2[key] = torch.zeros(2, dtype=torch.float32, device=device)
    return _ZEROS2[key]


def _exchange_champions(local_min, local_best, champ):
    """ONE collective for the per-rank champions: every rank contributes [C*H*W floats of its champion | min, local index,
    0, 0] (the header behind the image keeps the image 16-byte aligned) and receives all of them.
    -> (mins [world] fp32, local indices [world] int64, champions [world, C, H, W]).
    One all-gather instead of two: through RCCL a collective costs tens of microseconds of fixed latency however small it
    is (measured with a one-rank group on MI355X: the two-collective form of the per-step select added about 50 us to a
    59 us search step), so the scores ride with the particle.  (Host-tensor form, what the gloo tests run; on the device
    `_champion_table` builds the same record in one launch and `kernels.select_champion` reads it in one.)"""
    world = dist.get_world_size()
    shape = tuple(champ.shape[1:])
    chw = champ[0].numel()
    # (local_best as fp32 is exact below 2^24 particles per rank)
    mine = torch.cat([champ.reshape(-1), local_min.reshape(1).float(), local_best.reshape(1).float(), _zeros2(champ.device)])
    table = torch.empty(world * (chw + 4), dtype=torch.float32, device=champ.device)
    dist.all_gather_into_tensor(table, mine)
    table = table.reshape(world, chw + 4)
    return table[:, chw].contiguous(), table[:, chw + 1].long(), table[:, :chw].reshape((world,) + shape)


def _champion_table(costs_local, particles_local):
    """Device form of the exchange: pack (argmin over this rank's costs + champion + header: ONE launch), all-gather
    -> table [world, C*H*W + 4].  An empty shard contributes a +inf record."""
    world = dist.get_world_size()
    chw = particles_local[0].numel() if particles_local.shape[0] else int(torch.Size(particles_local.shape[1:]).numel())
    if particles_local.shape[0] == 0:
        mine = torch.zeros(chw + 4, dtype=torch.float32, device=particles_local.device)
        mine[chw] = float("inf")
    else:
        mine = kernels.pack_champion(particles_local, costs_local)
    table = torch.empty(world * (chw + 4), dtype=torch.float32, device=particles_local.device)     # (flat: gloo insists)
    dist.all_gather_into_tensor(table, mine)
    return table.reshape(world, chw + 4)


def global_best_of_n_device(scores_local, particles_local, counts):
    """The same select without any host read (for timed regions / graph-friendly callers): every rank contributes its
    local champion, the winner is picked on the device.  -> (winner [1,C,H,W] on every rank, global rank-major index
    as a device int64 scalar).  ONE collective: per rank the champion particle (786 KB at 3x256x256) with its (min, local
    index -- exact below 2^24) behind it, over the direct xGMI links.
    An empty shard contributes +inf (if every real score is +inf too, an empty lower rank's placeholder could win --
    torch.argmin's all-inf answer is not reproduced in that corner)."""
    rank, world = _world()
    n_local = scores_local.numel()
    dev, shape = particles_local.device, tuple(particles_local.shape[1:])
    if particles_local.is_cuda and not _solo():
        # three launches around the one collective: pack (argmin + champion + header), select (pick + copy), index
        winner, win_rank, win_local = kernels.select_champion(_champion_table(scores_local, particles_local), shape,
                                                              n_out=1, want_index=True)
        offsets = _offsets(counts, dev)
        return winner, (offsets.gather(0, win_rank.reshape(1)).reshape(()) + win_local)
    if n_local == 0:
        local_best = torch.zeros((), dtype=torch.int64, device=dev)
        local_min = torch.full((1,), float("inf"), dtype=torch.float32, device=dev)
        champ = torch.zeros((1,) + shape, dtype=particles_local.dtype, device=dev)
    elif particles_local.is_cuda:
        local_best, local_min = kernels.argmin(scores_local, want_value=True)
        champ = kernels.replicate(particles_local, local_best, n_out=1)
    else:
        local_best = torch.argmin(scores_local)
        local_min = scores_local[local_best].reshape(1).float()
        champ = particles_local[local_best].unsqueeze(0).contiguous()
    if _solo():
        return champ, local_best
    mins, local_idx, stacked = _exchange_champions(local_min, local_best, champ.float())
    win_rank = first_argmin(mins)                                        # lowest rank wins ties = first-min rule
    best = (_offsets(counts, dev) + local_idx).gather(0, win_rank.reshape(1)).reshape(())
    stacked = stacked.contiguous()
    winner = kernels.replicate(stacked, win_rank, n_out=1) if stacked.is_cuda else stacked[int(win_rank)].unsqueeze(0)
    return winner, best


class GlobalSelect:
    """`SearchDDPM.global_select` hook: per-step best-of-N across ranks (gaussian_diffusion.py:626-633
    of the reference, generalised to a sharded particle set).  No host sync: every rank contributes its
    local champion, the winner is picked on the device from the gathered (min, rank) table.
    Every rank holds at least one particle (the loop runs a batch per rank)."""

    def __call__(self, costs_local, particles_local, n_out=None):
        """-> n_out copies of the global winner (default: one per local particle)"""
        rank, world = _world()
        n = particles_local.shape[0] if n_out is None else int(n_out)
        if _solo():
            return kernels.replicate(particles_local, kernels.argmin(costs_local), n_out=n)
        if particles_local.is_cuda:
            # pack launch -> ONE collective -> select launch (pick + n copies); nothing else on the stream, no host read
            return kernels.select_champion(_champion_table(costs_local, particles_local), tuple(particles_local.shape[1:]),
                                           n_out=n)
        local_best = torch.argmin(costs_local)
        local_min = costs_local[local_best].reshape(1)
        champ = particles_local[local_best].unsqueeze(0).contiguous()
        mins, _, stacked = _exchange_champions(local_min, local_best, champ.float())     # ONE collective per step
        win_rank = first_argmin(mins)                                     # lowest rank wins ties = first-min rule
        return stacked[int(win_rank)].unsqueeze(0).repeat(n, 1, 1, 1)


class ScoreGather:
    """Per-step all-gather of the particle scores that does not stall the step (BASELINE configs[3]: "RCCL score
    all-gather"): the scores of step k are copied to a staging buffer and gathered asynchronously on the backend's own
    stream while step k + 1 runs; `submit` hands back the gathered scores of the PREVIOUS step (None at first), `flush` the
    last ones.  A collective costs tens of microseconds of fixed latency through RCCL: issued synchronously every step it
    would be a fifth of a 130 us step, pipelined it is hidden.  Equal shards (the loops run one batch size per rank)."""

    def __init__(self):
        self._stage, self._out, self._pending, self._k = [None, None], [None, None], None, 0

    def _finish(self):
        if self._pending is None:
            return None
        work, out = self._pending
        self._pending = None
        if work is not None:
            work.wait()
        return out

    def submit(self, scores_local):
        prev = self._finish()
        rank, world = _world()
        i = self._k & 1
        self._k += 1
        n = scores_local.numel()
        if self._stage[i] is None or self._stage[i].numel() != n or self._stage[i].device != scores_local.device:
            self._stage[i] = torch.empty(n, dtype=torch.float32, device=scores_local.device)
            self._out[i] = torch.empty(n * world, dtype=torch.float32, device=scores_local.device)
        self._stage[i].copy_(scores_local.reshape(-1))
        if _solo():
            self._out[i].copy_(self._stage[i])
            self._pending = (None, self._out[i])
        else:
            self._pending = (dist.all_gather_into_tensor(self._out[i], self._stage[i], async_op=True), self._out[i])
        return prev

    def flush(self):
        return self._finish()


def resample_ids(scores_local, temperature, generator, counts=None, return_scores=False):
    """Identical multinomial ids on every rank from the gathered scores and a shared-seed HOST generator
    (gaussian_diffusion.py:689-698: w = exp(-d / T), torch.multinomial with replacement).
    None when all weights are equal (the reference skips the draw, :693)."""
    all_scores = gather_scores(scores_local, counts).float().cpu()
    w = torch.exp(-all_scores / temperature)
    ids = None if w.max() == w.min() else torch.multinomial(w, all_scores.numel(), replacement=True,
                                                            generator=generator)
    return (ids, all_scores) if return_scores else ids


def fetch_plan(ids_global, n_local, rank, world):
    """Who sends what for a resample with known ids (host int64 [world * n_local], identical on every rank).
    Rank r's slots are ids[r*n_local:(r+1)*n_local]; particle g lives on rank g // n_local.  A source sends each particle
    a destination asks for ONCE, however many of the destination's slots drew it (multinomial draws repeat the heavy
    particles); the destination re-expands locally.
    -> (send_local [int64: this rank's local particle indices, grouped by destination rank, ascending within a group],
        in_splits [world], out_splits [world], slot_map [n_local int64: row of the receive buffer for each of my slots])."""
    import numpy as np
    ids = np.asarray(ids_global, dtype=np.int64).reshape(world, n_local)
    if ids.size and (ids.min() < 0 or ids.max() >= world * n_local):
        raise IndexError("resample id beyond the sharded particle set")
    send, in_splits = [], []
    for dst in range(world):                          # what I (as a source) send to dst
        u = np.unique(ids[dst])
        mine = u[(u >= rank * n_local) & (u < (rank + 1) * n_local)] - rank * n_local
        send.append(mine)
        in_splits.append(int(mine.size))
    need = np.unique(ids[rank])                       # what I (as a destination) receive: sorted, hence grouped by source
    owners = need // max(n_local, 1)
    out_splits = [int((owners == src).sum()) for src in range(world)]
    slot_map = np.searchsorted(need, ids[rank])
    send_local = np.concatenate(send) if send else np.zeros(0, dtype=np.int64)
    return send_local.astype(np.int64), in_splits, out_splits, slot_map.astype(np.int64)


def resample_particles(particles_local, ids_global, fetch="auto"):
    """Fetch the resampled particle set (equal shards: every rank runs the same batch size).
    fetch="all":      all-gather of every rank's states, then a HIP gather of this rank's slots -- world x the bytes a
                      rank needs, but no host knowledge of the ids (they may live on the device: no host read).
    fetch="selected": only the particles each rank's slots drew travel, each once per destination (`fetch_plan`): ONE
                      all-to-all with uneven splits straight over the pairwise xGMI links -- at most n_local particles
                      received per rank instead of (world - 1) * n_local (N = 512 on 8 GPUs: <= 50 MB instead of 352 MB
                      per rank and resample).  The split sizes are host integers, so device ids cost one host read.
    fetch="auto":     "selected" when the ids are on the host already or there are more than two ranks, else "all"."""
    rank, world = _world()
    n_local = particles_local.shape[0]
    if fetch not in ("auto", "all", "selected"):
        raise ValueError("fetch: 'auto', 'all' or 'selected'")
    if fetch == "auto":
        fetch = "selected" if (not ids_global.is_cuda or world > 2) else "all"
    if _solo():
        pool = particles_local
    elif fetch == "selected":
        send_local, in_splits, out_splits, slot_map = fetch_plan(ids_global.cpu().numpy(), n_local, rank, world)
        dev = particles_local.device
        send_local, slot_map = torch.from_numpy(send_local), torch.from_numpy(slot_map)
        if particles_local.is_cuda:
            outbox = kernels.gather(particles_local, send_local.to(dev, non_blocking=True), validate=False)
        else:
            outbox = particles_local[send_local].contiguous()
        inbox = torch.empty((sum(out_splits),) + tuple(particles_local.shape[1:]), dtype=outbox.dtype, device=dev)
        dist.all_to_all_single(inbox, outbox, out_splits, in_splits)
        if inbox.is_cuda:
            return kernels.gather(inbox, slot_map.to(dev, non_blocking=True), validate=False)
        return inbox[slot_map]
    else:
        pool = torch.empty((world * n_local,) + tuple(particles_local.shape[1:]), dtype=particles_local.dtype,
                           device=particles_local.device)
        dist.all_gather_into_tensor(pool, particles_local.contiguous())
    mine = ids_global[rank * n_local:(rank + 1) * n_local]
    if pool.is_cuda:
        return kernels.gather(pool, mine.to(pool.device), validate=False)      # ids drawn over the gathered set
    return pool[mine]


def global_resample(particles_local, scores_local, temperature, generator, fetch="auto"):
    """The resampling block of TTC_DDIM.p_sample_loop (gaussian_diffusion.py:685-698) over a sharded particle set:
    -> (particles_local', scores_local', ids_global or None).  Same ids as one process holding all particles would
    draw from the same generator state.

    generator: a HOST generator seeded identically on every rank (the draw is torch.multinomial on the CPU: one host
    read of the gathered scores per resample -- the form the parity tests pin), or a DEVICE generator seeded identically
    on every rank: the draw then runs on the GPU as the reference's own does (its tensors live on cuda), every rank
    computes the same ids from the same gathered weights, and nothing is read on the host -- when all weights are equal
    the draw is replaced by the identity on the device, as TTC_DDIM._resample does on one GPU.
    fetch: how the drawn particles travel (`resample_particles`); with device ids "selected" costs the one host read of
    the ids that "all" avoids, and moves world x fewer bytes -- "auto" takes it beyond two ranks."""
    rank, world = _world()
    n_local = particles_local.shape[0]
    if generator is not None and generator.device.type == "cuda" and particles_local.is_cuda:
        all_scores = gather_scores(scores_local, [n_local] * world).float()
        n = all_scores.numel()
        w = torch.exp(-all_scores / temperature)
        flat = w.max() == w.min()
        drawn = torch.multinomial(torch.where(flat, torch.ones_like(w), w), n, replacement=True, generator=generator)
        ids = torch.where(flat, torch.arange(n, device=w.device), drawn)
        fetched = resample_particles(particles_local, ids, fetch)
        mine = ids[rank * n_local:(rank + 1) * n_local]
        return fetched, kernels.gather(all_scores.reshape(n, 1), mine, validate=False).reshape(n_local), ids
    ids, all_scores = resample_ids(scores_local, temperature, generator, counts=[n_local] * world,
                                   return_scores=True)
    if ids is None:
        return particles_local, scores_local, None
    fetched = resample_particles(particles_local, ids, fetch)
    mine = ids[rank * n_local:(rank + 1) * n_local]
    return fetched, all_scores[mine].to(scores_local.device), ids
