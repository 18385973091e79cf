"""CPU suite: the multi-GPU control flow (particle sharding, score all-gather, global select, resampling) over
world-size-2 gloo.  The data-path kernels are not involved: this pins the rendezvous logic and the index rules."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dps_ttc_amd import distributed as dd
        out = {}
        n_total, n_local = 6, 3
        lo, hi = dd.shard_range(n_total)
        out["shard"] = (lo, hi)
        g = torch.Generator().manual_seed(0)
        particles = torch.randn(n_total, 3, 4, 4, generator=g)
        scores = torch.tensor([5.0, 2.0, 9.0, 2.0, 1.5, 7.0])
        mine_p, mine_s = particles[lo:hi].clone(), scores[lo:hi].clone()
        allsc = dd.gather_scores(mine_s)
        out["gathered"] = allsc.tolist()
        winner, best, _ = dd.global_best_of_n(mine_s, mine_p)
        out["best"] = best
        out["winner_ok"] = bool(torch.equal(winner[0], particles[4]))
        # first-min tie rule across ranks: equal minima -> lowest global index
        tie = torch.tensor([3.0, 1.0, 4.0]) if rank == 0 else torch.tensor([1.0, 9.0, 1.0])
        _, tbest, _ = dd.global_best_of_n(tie, mine_p)
        out["tie_best"] = tbest
        # per-step select: everybody ends up with n_local copies of the global champion
        sel = dd.GlobalSelect()(mine_s, mine_p)
        out["select_ok"] = bool(all(torch.equal(sel[i], particles[4]) for i in range(n_local)))
        # resampling: identical ids on every rank, states fetched from their owners
        ids = dd.resample_ids(mine_s, 100.0, torch.Generator().manual_seed(77))
        out["ids"] = ids.tolist()
        fetched = dd.resample_particles(mine_p, ids)
        out["fetch_ok"] = bool(all(torch.equal(fetched[i], particles[ids[rank * n_local + i]]) for i in range(n_local)))
        out["flat"] = dd.resample_ids(torch.ones(3), 100.0, torch.Generator().manual_seed(1)) is None
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=150) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert res[0]["shard"] == (0, 3) and res[1]["shard"] == (3, 6)
    for r in (0, 1):
        assert res[r]["gathered"] == [5.0, 2.0, 9.0, 2.0, 1.5, 7.0]
        assert res[r]["best"] == 4 and res[r]["winner_ok"]
        assert res[r]["tie_best"] == 1
        assert res[r]["select_ok"] and res[r]["fetch_ok"] and res[r]["flat"]
    assert res[0]["ids"] == res[1]["ids"] and len(res[0]["ids"]) == 6
    # same draw as a single process would make from the gathered scores (torch.multinomial, shared seed)
    w = torch.exp(-torch.tensor([5.0, 2.0, 9.0, 2.0, 1.5, 7.0]) / 100.0)
    assert res[0]["ids"] == torch.multinomial(w, 6, replacement=True, generator=torch.Generator().manual_seed(77)).tolist()


def test_single_process_paths():
    from dps_ttc_amd import distributed as dd
    assert dd.shard_range(10, rank=0, world=3) == (0, 4) and dd.shard_range(10, rank=2, world=3) == (7, 10)
    s = torch.tensor([3.0, 1.0, 2.0])
    assert dd.gather_scores(s) is not None and int(dd.first_argmin(s)) == 1
    w, best, alls = dd.global_best_of_n(s, torch.arange(12.0).reshape(3, 1, 2, 2))
    assert best == 1 and torch.equal(w[0], torch.arange(4.0, 8.0).reshape(1, 2, 2)) and alls.tolist() == s.tolist()
    assert int(dd.first_argmin(torch.tensor([2.0, float("nan"), 1.0]))) == 1      # NaN is the minimum (torch.argmin)
