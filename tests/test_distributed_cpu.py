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
        one = dd.GlobalSelect()(mine_s, mine_p, n_out=1)          # the single-state search loop asks for ONE copy
        out["select_one_ok"] = bool(one.shape[0] == 1 and torch.equal(one[0], particles[4]))
        # resampling: identical ids on every rank, states fetched from their owners
        ids = dd.resample_ids(mine_s, 100.0, torch.Generator().manual_seed(77))
        out["ids"] = ids.tolist()
        out["fetch_ok"] = True
        for mode in ("auto", "all", "selected"):      # both exchange forms fetch the same set
            fetched = dd.resample_particles(mine_p, ids, fetch=mode)
            out["fetch_ok"] &= bool(all(torch.equal(fetched[i], particles[ids[rank * n_local + i]]) for i in range(n_local)))
        # one-sided draws: everything from rank 1 / everything from rank 0 / each rank keeps its own (nothing travels)
        for forced in ([4, 4, 5, 3, 3, 4], [0, 0, 0, 2, 1, 0], [2, 1, 0, 5, 5, 3]):
            fi = torch.tensor(forced)
            fetched = dd.resample_particles(mine_p, fi, fetch="selected")
            out["fetch_ok"] &= bool(torch.equal(fetched, particles[fi[rank * n_local:(rank + 1) * n_local]]))
        out["flat"] = dd.resample_ids(torch.ones(3), 100.0, torch.Generator().manual_seed(1)) is None
        # the sync-free select (what bench.py closes its timed region with): same winner, index as a tensor
        w2, b2 = dd.global_best_of_n_device(mine_s, mine_p, [3, 3])
        out["dev_best"] = int(b2)
        out["dev_winner_ok"] = bool(torch.equal(w2[0], particles[4]))
        _, tb2 = dd.global_best_of_n_device(tie, mine_p, [3, 3])
        out["dev_tie_best"] = int(tb2)
        # --- the driver's layout: n_paths / batch_size = 3 particle groups of 3 on 2 ranks (rank 0: two groups,
        # rank 1: one), sharded contiguously, so the rank-major index IS the path index
        groups, batch = 3, 3
        g_lo, g_hi = dd.shard_range(groups)
        counts = [c * batch for c in dd.shard_counts(groups)]
        gen = torch.Generator().manual_seed(5)
        all_p = torch.randn(groups * batch, 3, 4, 4, generator=gen)
        all_s = torch.tensor([4.0, 3.0, 8.0, 6.0, 5.0, 9.0, 7.0, 0.5, 2.0])        # winner: path 7 (group 2, rank 1)
        my_p, my_s = all_p[g_lo * batch:g_hi * batch].clone(), all_s[g_lo * batch:g_hi * batch].clone()
        wg, bg, ag = dd.global_best_of_n(my_s, my_p, counts)
        out["groups"] = (counts, bg, ag.tolist(), bool(torch.equal(wg[0], all_p[7])))
        wg2, bg2 = dd.global_best_of_n_device(my_s, my_p, counts)
        out["groups_dev"] = (int(bg2), bool(torch.equal(wg2[0], all_p[7])))
        # counts exchanged instead of derived
        _, bg3, _ = dd.global_best_of_n(my_s, my_p)
        out["groups_exchanged"] = bg3
        # --- fewer groups than ranks: rank 1 holds nothing, still enters the collectives, still gets the winner
        e_p = all_p[:3].clone() if rank == 0 else torch.empty(0, 3, 4, 4)
        e_s = torch.tensor([2.0, 1.0, 3.0]) if rank == 0 else torch.empty(0)
        we, be, ae = dd.global_best_of_n(e_s, e_p, [3, 0])
        out["empty"] = (be, ae.tolist(), bool(torch.equal(we[0], all_p[1])))
        we2, be2 = dd.global_best_of_n_device(e_s, e_p, [3, 0])
        out["empty_dev"] = (int(be2), bool(torch.equal(we2[0], all_p[1])))
        # --- ttc_ddim's resampling block over the sharded set (control flow only: no kernels on CPU tensors)
        from dps_ttc_amd.gaussian_diffusion import create_sampler
        smp = create_sampler(sampler="ttc_ddim", steps=1000, noise_schedule="linear", model_mean_type="epsilon",
                             model_var_type="learned_range", dynamic_threshold=False, clip_denoised=True,
                             rescale_timesteps=True, timestep_respacing="20")
        smp.global_resample, smp.resample_generator = True, torch.Generator().manual_seed(11)
        d6 = torch.tensor([50.0, 400.0, 30.0, 900.0, 10.0, 250.0])
        img3, dist3 = smp._resample(particles[lo:hi].clone(), d6[lo:hi].clone(), 100)
        out["ttc"] = (smp.last_resample_ids.tolist(), img3, dist3.tolist())
        # --- pipelined per-step score gather: submit() returns the PREVIOUS step's gathered scores, flush() the last
        sg = dd.ScoreGather()
        seen = []
        for k in range(3):
            prev = sg.submit(mine_s + float(k))
            seen.append(None if prev is None else prev.tolist())
        seen.append(sg.flush().tolist())
        out["score_gather"] = seen
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def _worker4(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dps_ttc_amd import distributed as dd
        out = {}
        n_local = 3
        n = world * n_local
        particles = torch.randn(n, 3, 4, 4, generator=torch.Generator().manual_seed(2))
        scores = torch.tensor([9.0, 4.0, 7.0, 3.0, 8.0, 6.0, 5.0, 0.5, 2.0, 0.5, 1.0, 11.0])   # first minimum: 7 (rank 2)
        lo, hi = dd.shard_range(n)
        mine_p, mine_s = particles[lo:hi].clone(), scores[lo:hi].clone()
        out["shard"] = (lo, hi)
        w, b = dd.global_best_of_n_device(mine_s, mine_p, [n_local] * world)
        out["dev"] = (int(b), bool(torch.equal(w[0], particles[7])))
        sel = dd.GlobalSelect()(mine_s, mine_p)
        out["select"] = bool(all(torch.equal(sel[i], particles[7]) for i in range(n_local)))
        # uneven shards, one of them empty: counts [4, 0, 5, 3]
        counts = [4, 0, 5, 3]
        off = [0, 4, 4, 9]
        up, us = particles[off[rank]:off[rank] + counts[rank]].clone(), scores[off[rank]:off[rank] + counts[rank]].clone()
        w, b, alls = dd.global_best_of_n(us, up, counts)
        out["uneven"] = (b, alls.tolist(), bool(torch.equal(w[0], particles[7])))
        w, b = dd.global_best_of_n_device(us, up, counts)
        out["uneven_dev"] = (int(b), bool(torch.equal(w[0], particles[7])))
        # resampling over four ranks: both exchange forms fetch the set one process would hold
        gen = torch.Generator().manual_seed(21)
        xr, dr, ids = dd.global_resample(mine_p, mine_s * 60.0, 100.0, gen, fetch="selected")
        out["ids"] = ids.tolist()
        out["sel_ok"] = bool(torch.equal(xr, particles[ids[lo:hi]]) and torch.equal(dr, (scores * 60.0)[ids[lo:hi]]))
        xa = dd.resample_particles(mine_p, ids, fetch="all")
        out["all_ok"] = bool(torch.equal(xa, xr))
        skew = torch.tensor([11] * 6 + [0] * 6)            # two particles feed everybody
        out["skew_ok"] = bool(torch.equal(dd.resample_particles(mine_p, skew, fetch="selected"), particles[skew[lo:hi]]))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_world_size_4_gloo():
    """four ranks (the CPU rehearsal of a wider node): rank-major shards, the champion exchange, uneven and empty shards,
    and the resampling exchange in both forms -- all-gather of every state / one all-to-all of the drawn particles"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker4, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=150) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    scores = [9.0, 4.0, 7.0, 3.0, 8.0, 6.0, 5.0, 0.5, 2.0, 0.5, 1.0, 11.0]
    w = torch.exp(-torch.tensor(scores) * 60.0 / 100.0)
    ids1 = torch.multinomial(w, 12, replacement=True, generator=torch.Generator().manual_seed(21)).tolist()
    for r in range(4):
        assert res[r]["shard"] == (3 * r, 3 * r + 3)
        assert res[r]["dev"] == (7, True) and res[r]["select"]
        assert res[r]["uneven"] == (7, scores, True) and res[r]["uneven_dev"] == (7, True)
        assert res[r]["ids"] == ids1                       # the draw one process holding all twelve would make
        assert res[r]["sel_ok"] and res[r]["all_ok"] and res[r]["skew_ok"]


def _worker8(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dps_ttc_amd import distributed as dd
        n_local = 4
        n = world * n_local
        g = torch.Generator().manual_seed(3)
        particles = torch.randn(n, 3, 4, 4, generator=g)
        scores = (torch.rand(n, generator=g) * 300.0).round()
        scores[19] = scores.min() - 1.0                    # a unique winner, on rank 4
        lo, hi = dd.shard_range(n)
        mine_p, mine_s = particles[lo:hi].clone(), scores[lo:hi].clone()
        out = {}
        w, b = dd.global_best_of_n_device(mine_s, mine_p, [n_local] * world)
        out["dev"] = (int(b), bool(torch.equal(w[0], particles[19])))
        sel = dd.GlobalSelect()(mine_s, mine_p, n_out=2)
        out["select"] = bool(sel.shape[0] == 2 and torch.equal(sel[0], particles[19]) and torch.equal(sel[1], particles[19]))
        xr, dr, ids = dd.global_resample(mine_p, mine_s, 100.0, torch.Generator().manual_seed(5))      # fetch="auto": selected
        out["ids"] = ids.tolist()
        out["sel_ok"] = bool(torch.equal(xr, particles[ids[lo:hi]]) and torch.equal(dr, scores[ids[lo:hi]]))
        out["all_ok"] = bool(torch.equal(dd.resample_particles(mine_p, ids, fetch="all"), xr))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world_size_8_gloo():
    """eight ranks on the CPU -- the node size the BASELINE configurations name: the champion exchange and the resampling
    exchange (uneven all-to-all of the drawn particles against the all-gather of every state) agree with one process"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(3)
    torch.randn(32, 3, 4, 4, generator=g)
    scores = (torch.rand(32, generator=g) * 300.0).round()
    scores[19] = scores.min() - 1.0
    ids1 = torch.multinomial(torch.exp(-scores / 100.0), 32, replacement=True, generator=torch.Generator().manual_seed(5)).tolist()
    for r in range(8):
        assert res[r]["dev"] == (19, True) and res[r]["select"]
        assert res[r]["ids"] == ids1 and res[r]["sel_ok"] and res[r]["all_ok"]


def test_fetch_plan_is_consistent_across_ranks():
    """distributed.fetch_plan (the selected-particles exchange of a resample): what rank s plans to send to rank d is what
    rank d plans to receive from s, every particle travels at most once per destination, and expanding the receive buffer
    by the slot map yields exactly the drawn set -- for worlds 1..8, skewed draws included."""
    from dps_ttc_amd import distributed as dd
    rng = np.random.default_rng(0)
    for world, n_local in ((1, 5), (2, 3), (3, 4), (8, 64), (8, 1)):
        n = world * n_local
        for ids in (rng.integers(0, n, n), np.full(n, n - 1), np.arange(n), np.arange(n)[::-1].copy(),
                    rng.choice([0, n // 2, n - 1], n)):
            plans = [dd.fetch_plan(ids, n_local, r, world) for r in range(world)]
            particles = rng.standard_normal((n, 2)).astype(np.float32)
            moved = 0
            for d in range(world):
                send_d, in_d, out_d, slot_d = plans[d]
                assert sum(in_d) == send_d.size and len(in_d) == len(out_d) == world
                inbox = []
                for s_ in range(world):
                    send_s, in_s, _, _ = plans[s_]
                    off = sum(in_s[:d])
                    rows = send_s[off:off + in_s[d]]
                    assert rows.size == out_d[s_]                          # the two sides agree on every split
                    assert len(set(rows.tolist())) == rows.size            # once per destination
                    inbox.append(particles[s_ * n_local + rows])
                    moved += rows.size if s_ != d else 0
                inbox = np.concatenate(inbox) if inbox else np.zeros((0, 2), np.float32)
                np.testing.assert_array_equal(inbox[slot_d], particles[ids[d * n_local:(d + 1) * n_local]])
            assert moved <= n                                             # never more than one set's worth in flight
    with pytest.raises(IndexError):
        dd.fetch_plan(np.array([0, 4, 1, 1]), 2, 0, 2)


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
        assert res[r]["select_ok"] and res[r]["select_one_ok"] and res[r]["fetch_ok"] and res[r]["flat"]
    assert res[0]["ids"] == res[1]["ids"] and len(res[0]["ids"]) == 6
    # same draw as a single process would make from the gathered scores (torch.multinomial, shared seed)
    w = torch.exp(-torch.tensor([5.0, 2.0, 9.0, 2.0, 1.5, 7.0]) / 100.0)
    assert res[0]["ids"] == torch.multinomial(w, 6, replacement=True, generator=torch.Generator().manual_seed(77)).tolist()
    all_s = [4.0, 3.0, 8.0, 6.0, 5.0, 9.0, 7.0, 0.5, 2.0]
    for r in (0, 1):
        assert res[r]["dev_best"] == 4 and res[r]["dev_winner_ok"] and res[r]["dev_tie_best"] == 1
        # 3 groups on 2 ranks pick the path a single process picks: index 7 = argmin of the concatenated distances
        assert res[r]["groups"] == ([6, 3], 7, all_s, True) and int(torch.argmin(torch.tensor(all_s))) == 7
        assert res[r]["groups_dev"] == (7, True) and res[r]["groups_exchanged"] == 7
        assert res[r]["empty"] == (1, [2.0, 1.0, 3.0], True) and res[r]["empty_dev"] == (1, True)
    base = [5.0, 2.0, 9.0, 2.0, 1.5, 7.0]
    for r in (0, 1):       # pipelined score gather: step k's scores come back at step k + 1, the last ones at flush()
        assert res[r]["score_gather"] == [None, base, [v + 1.0 for v in base], [v + 2.0 for v in base]]
    # ttc_ddim: 2 ranks x 3 particles resample to what 1 rank x 6 does with the same generator
    from dps_ttc_amd.gaussian_diffusion import create_sampler
    smp = create_sampler(sampler="ttc_ddim", steps=1000, noise_schedule="linear", model_mean_type="epsilon",
                         model_var_type="learned_range", dynamic_threshold=False, clip_denoised=True,
                         rescale_timesteps=True, timestep_respacing="20")
    smp.global_resample, smp.resample_generator = True, torch.Generator().manual_seed(11)
    particles = torch.randn(6, 3, 4, 4, generator=torch.Generator().manual_seed(0))
    d6 = torch.tensor([50.0, 400.0, 30.0, 900.0, 10.0, 250.0])
    img6, dist6 = smp._resample(particles.clone(), d6.clone(), 100)
    ids6 = smp.last_resample_ids.tolist()
    assert len(set(ids6)) > 1
    for r in (0, 1):
        ids, img3, dist3 = res[r]["ttc"]
        assert ids == ids6
        assert torch.equal(img3, img6[3 * r:3 * r + 3]) and dist3 == dist6[3 * r:3 * r + 3].tolist()


def test_single_process_paths():
    from dps_ttc_amd import distributed as dd
    assert dd.shard_range(10, rank=0, world=3) == (0, 4) and dd.shard_range(10, rank=2, world=3) == (7, 10)
    s = torch.tensor([3.0, 1.0, 2.0])
    assert dd.gather_scores(s) is not None and int(dd.first_argmin(s)) == 1
    w, best, alls = dd.global_best_of_n(s, torch.arange(12.0).reshape(3, 1, 2, 2))
    assert best == 1 and torch.equal(w[0], torch.arange(4.0, 8.0).reshape(1, 2, 2)) and alls.tolist() == s.tolist()
    w, best = dd.global_best_of_n_device(s, torch.arange(12.0).reshape(3, 1, 2, 2), [3])
    assert int(best) == 1 and torch.equal(w[0], torch.arange(4.0, 8.0).reshape(1, 2, 2))
    assert dd.shard_counts(10, 3) == [4, 3, 3] and dd.locate(5, [4, 3, 3]) == (1, 1) and dd.locate(0, [0, 2]) == (1, 0)
    assert int(dd.first_argmin(torch.tensor([2.0, float("nan"), 1.0]))) == 1      # NaN is the minimum (torch.argmin)
    sg = dd.ScoreGather()                                                          # no process group: a local copy
    assert sg.submit(s) is None and sg.submit(s + 1).tolist() == s.tolist() and sg.flush().tolist() == (s + 1).tolist()
