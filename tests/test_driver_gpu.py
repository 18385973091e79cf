"""GPU suite: the driver counterpart end to end (YAML -> UNet -> fused DPS loop -> PNGs), tiny settings."""
import os
import sys

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(tmp_path, task, sampler):
    from PIL import Image
    data = tmp_path / "data"
    data.mkdir()
    rng = np.random.RandomState(0)
    for i in range(2):
        base = rng.rand(8, 8, 3)
        img = np.kron(base, np.ones((32, 32, 1)))            # 256 x 256 blocky image
        Image.fromarray((img * 255).astype(np.uint8)).save(data / f"{i:05d}.png")
    cfg = yaml.load(open(os.path.join(ROOT, "configs", task)), Loader=yaml.FullLoader)
    cfg["data"]["root"] = str(data)
    tpath = tmp_path / "task.yaml"
    yaml.dump(cfg, open(tpath, "w"))
    diff = yaml.load(open(os.path.join(ROOT, "configs", "diffusion_config.yaml")), Loader=yaml.FullLoader)
    diff["sampler"] = sampler
    dpath = tmp_path / "diffusion.yaml"
    yaml.dump(diff, open(dpath, "w"))
    return str(tpath), str(dpath)


@pytest.mark.parametrize("task,sampler", [("gaussian_deblur_config.yaml", "ddpm"),
                                          ("super_resolution_config.yaml", "search_ddpm"),
                                          ("inpainting_config.yaml", "ddpm")])
def test_driver_end_to_end(tmp_path, task, sampler):
    sys.path.insert(0, ROOT)
    import sample_condition_batched_ttc as drv
    tpath, dpath = _setup(tmp_path, task, sampler)
    out = tmp_path / "results"
    drv.main(["--model_config", os.path.join(ROOT, "configs", "model_config.yaml"), "--diffusion_config", dpath,
              "--task_config", tpath, "--save_dir", str(out), "--n_paths", "2", "--batch_size", "2",
              "--ref_image_idxs", "0", "--timestep_respacing", "3", "--seed", "0", "--gpu", "0"])
    sub = [d for d in os.listdir(out)]
    assert len(sub) == 1
    root = out / sub[0]
    assert (root / "input" / "00000.png").exists() and (root / "label" / "00000.png").exists()
    assert (root / "recon_paths" / "00000" / "path#1.png").exists() and (root / "recon_paths" / "00000" / "path#2.png").exists()
    assert (root / "best_of_n" / "00000.png").exists()
    d = np.load(root / "00000_pathwise_distances.npy")
    assert d.shape == (2,) and np.isfinite(d).all() and (d > 0).all()


def _run_driver(tmp_path, task, sampler, n_paths, batch_size, ranks=1, extra=()):
    """the driver as its own process(es); ranks > 1: one process per rank on this box's single GPU over gloo
    (the RCCL path needs one GPU per rank: the driver's round-end multi-GPU run covers it)"""
    import socket
    import subprocess
    tpath, dpath = _setup(tmp_path, task, sampler)
    out = tmp_path / "results"
    cmd = [sys.executable, os.path.join(ROOT, "sample_condition_batched_ttc.py"),
           "--model_config", os.path.join(ROOT, "configs", "model_config.yaml"), "--diffusion_config", dpath,
           "--task_config", tpath, "--save_dir", str(out), "--n_paths", str(n_paths), "--batch_size", str(batch_size),
           "--ref_image_idxs", "0", "--timestep_respacing", "3", "--seed", "0", "--gpu", "0", *extra]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(ranks):
        env = dict(os.environ)
        if ranks > 1:
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(ranks), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), DPSX_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    (sub,) = os.listdir(out)
    return out / sub, logs


def _check_best_of_n(root, n_paths):
    d = np.load(root / "00000_pathwise_distances.npy")
    assert d.shape == (n_paths,) and np.isfinite(d).all() and (d > 0).all()
    best = int(np.argmin(d))
    for i in range(n_paths):
        assert (root / "recon_paths" / "00000" / f"path#{i + 1}.png").exists(), i
    # the saved best-of-N image IS the reconstruction of the reported path (not another group's particle)
    a = open(root / "best_of_n" / "00000.png", "rb").read()
    b = open(root / "recon_paths" / "00000" / f"path#{best + 1}.png", "rb").read()
    assert a == b
    others = [open(root / "recon_paths" / "00000" / f"path#{i + 1}.png", "rb").read() for i in range(n_paths) if i != best]
    assert all(o != a for o in others)
    return best


def test_driver_more_groups_than_one(tmp_path):
    """n_paths = 2 x batch_size: the groups' final particles must not alias the sampler's persistent step buffers
    (every group ends in the same ping-pong slot), so best_of_n/*.png is the image of the reported path."""
    root, logs = _run_driver(tmp_path, "gaussian_deblur_config.yaml", "ddpm", 4, 2)
    best = _check_best_of_n(root, 4)
    assert f"best-of-4 = path#{best + 1} " in logs[0]


def test_consecutive_loops_do_not_share_storage():
    import torch
    from standin import StandInModel
    from dps_ttc_amd.condition_methods import get_conditioning_method
    from dps_ttc_amd.gaussian_diffusion import create_sampler
    from dps_ttc_amd.measurements import get_noise, get_operator
    dev = "cuda:0"
    op = get_operator("gaussian_blur", kernel_size=61, intensity=3.0, device=dev)
    cm = get_conditioning_method("ps", op, get_noise("gaussian", sigma=0.05), scale=0.3)
    smp = create_sampler(sampler="ddpm", steps=1000, noise_schedule="linear", model_mean_type="epsilon",
                         model_var_type="learned_range", dynamic_threshold=False, clip_denoised=True,
                         rescale_timesteps=True, timestep_respacing="4")
    model = StandInModel().to(dev)
    y = torch.rand(1, 3, 64, 64, device=dev)
    outs = []
    for seed in (0, 1):
        torch.manual_seed(seed)
        img, d, _ = smp.p_sample_loop(model=model, x_start=torch.randn(2, 3, 64, 64, device=dev).requires_grad_(),
                                      measurement=y, measurement_cond_fn=cm.conditioning, record=False, save_root=None)
        outs.append((img, d, img.clone(), d.clone()))
    assert outs[0][0].data_ptr() != outs[1][0].data_ptr() and outs[0][1].data_ptr() != outs[1][1].data_ptr()
    assert torch.equal(outs[0][0], outs[0][2]) and torch.equal(outs[0][1], outs[0][3])     # untouched by the 2nd loop
    assert not torch.equal(outs[0][0], outs[1][0])


@pytest.mark.parametrize("task,sampler,n_paths,batch,extra", [
    ("gaussian_deblur_config.yaml", "ddpm", 3, 1, ()),
    ("gaussian_deblur_config.yaml", "search_ddpm", 4, 2, ()),
    ("gaussian_deblur_config.yaml", "ttc_ddim", 4, 2, ()),
    # BASELINE configs[3] in its sharded form: motion deblur + ACTIVE semantic-guidance term (anneal 10x) -- the embedder is
    # a stand-in (the reference's face network is not available offline), everything else is the shipped YAML
    ("motion_deblur_config_semantic.yaml", "ddpm", 4, 2, ("--embedder", "tests.standin:toy_embedder")),
    ("motion_deblur_config_semantic.yaml", "search_ddpm", 4, 2, ("--embedder", "tests.standin:toy_embedder")),
    # BASELINE configs[4] in its sharded form: phase retrieval + global multinomial resampling over all ranks' particles
    ("phase_retrieval_config.yaml", "ttc_ddim", 4, 2, ()),
    ("phase_retrieval_config.yaml", "search_ddpm", 4, 2, ()),
])
def test_driver_two_ranks(tmp_path, task, sampler, n_paths, batch, extra):
    """two rank processes (gloo rehearsal on one GPU): uneven group shards for the independent-particle loop (3 groups
    on 2 ranks), per-step global select for search_ddpm, global resampling for ttc_ddim; path numbering is global and
    the best-of-N image is the reported path's."""
    root, logs = _run_driver(tmp_path, task, sampler, n_paths, batch, ranks=2, extra=extra)
    best = _check_best_of_n(root, n_paths) if sampler == "ddpm" else None
    d = np.load(root / "00000_pathwise_distances.npy")
    assert d.shape == (n_paths,) and np.isfinite(d).all()
    assert f"best-of-{n_paths} = path#{int(np.argmin(d)) + 1} " in logs[0]
    if sampler == "search_ddpm":
        # every particle of every rank ends as a copy of the global per-step winner: all distances agree
        assert np.allclose(d, d[0], rtol=1e-6)
    if best is not None and n_paths == 3:
        assert "Path#3 " in logs[1] and "Path#1 " in logs[0] and "Path#2 " in logs[0]
    if best is not None and n_paths == 4:
        assert "Path#3 " in logs[1] and "Path#4 " in logs[1] and "Path#1 " in logs[0]


def test_driver_particle_groups_flag(tmp_path):
    """--particle_groups 2: the batch's particles as two sub-batches on two HIP streams inside p_sample_loop
    (kernels.ParticleGroups); same outputs tree, finite distances"""
    root, logs = _run_driver(tmp_path, "gaussian_deblur_config.yaml", "ddpm", 4, 4, extra=("--particle_groups", "2"))
    _check_best_of_n(root, 4)


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` with no launcher: the parent starts the rank processes before any GPU call and
    relays one JSON line (gloo rehearsal: both ranks share this box's GPU)."""
    import json
    import subprocess
    env = dict(os.environ, DPSX_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--particles", "4", "--no-cpu-baseline"], env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_particles"] == 8 and rec["scaling"] == "weak"
    assert rec["value"] > 0 and 0 <= rec["best_of_n_index"] < 8 and "roofline" in rec


@pytest.mark.parametrize("mode", [
    # BASELINE configs[3]: motion blur, N sharded (strong scaling), semantic stand-in, per-step score all-gather
    ["--workload", "dps_scores", "--operator", "motion_blur", "--scaling", "strong", "--particles", "6", "--semantic"],
    # per-step best-of-N with the global select inside the timed region
    ["--workload", "search", "--operator", "motion_blur", "--scaling", "strong", "--particles", "6"],
    # BASELINE configs[4]: phase retrieval, global resampling every 2 steps inside the timed region
    ["--workload", "resample", "--operator", "phase_retrieval", "--particles", "3", "--resample-every", "2"],
    # the same with only the drawn particles travelling (one all-to-all with uneven splits)
    ["--workload", "resample", "--operator", "phase_retrieval", "--particles", "3", "--resample-every", "2",
     "--resample-fetch", "selected"],
])
def test_bench_sharded_workloads_two_ranks(mode):
    """the timed search / resample / per-step-score modes of bench.py, two gloo ranks on this box's one GPU: the
    collectives of distributed.py run inside the timed region and the line says so"""
    import json
    import subprocess
    env = dict(os.environ, DPSX_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--no-cpu-baseline", *mode], env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and np.isfinite(rec["ms_per_step"])
    assert rec["config"]["step"] == mode[1] and rec["config"]["global_particles"] == 6
    assert rec["scaling"] == ("strong" if "strong" in mode else "weak")
    assert rec["config"]["particles_per_gpu"] == 3 and rec["config"]["chains_per_gpu"] == 1
    assert 0 <= rec["best_of_n_index"] < 6 and "roofline" in rec
    if "selected" in mode:
        assert "all-to-all of the drawn particles" in rec["config"]["parallelism"]


def test_bench_line_carries_both_schedules():
    """one GPU, default workload: the line has the grouped step (value) AND the one-chain step, the survey-priced AND the
    moved-bytes fraction (profiles/traffic.json holds the Gaussian operator's PMC bytes)"""
    import json
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline"],
                       env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    rec = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.strip()][-1])
    rf = rec["roofline"]
    assert rec["config"]["chains_per_gpu"] == 3 and rec["config"]["x0_hat_store"] is False
    for key in ("one_chain_ms_per_step", "one_chain_value", "step_moved_bytes_per_particle", "step_moved_GBps",
                "step_frac_moved", "step_frac_of_hbm_roofline", "frac_of_copy_ceiling"):
        assert key in rf and rf[key] > 0, key
    assert rf["one_chain_ms_per_step"] > 0.5 * rec["ms_per_step"]
    for k, v in rf["per_launch"].items():
        assert v["moved_bytes"] and v["moved_GBps"] < 8000.0 and v["algorithmic_GBps"] < 8000.0, (k, v)
    assert rf["step_frac_moved"] < rf["step_frac_of_hbm_roofline"] <= 1.0


@pytest.mark.parametrize("operator", ["gaussian_blur", "motion_blur", "inpainting"])
def test_particle_groups_on_streams_match_one_chain(operator):
    """kernels.ParticleGroups (what sampler.particle_groups, the driver's --particle_groups and bench.py's timed loop
    run): the particles as independent groups on separate HIP streams, own operator handle and residual scratch each --
    per-particle results must not depend on the grouping: x_{t-1} and the distances bit for bit."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from dps_ttc_amd import kernels
    from dps_ttc_amd.gaussian_diffusion import create_sampler
    dev = torch.device("cuda", 0)
    smp = create_sampler(sampler="ddpm", steps=1000, noise_schedule="linear", model_mean_type="epsilon",
                         model_var_type="learned_range", dynamic_threshold=False, clip_denoised=True,
                         rescale_timesteps=True, timestep_respacing="")
    n, steps = 7, 4
    x_t, ring, truth, meas_noise = bench.synth_inputs(n, 2, dev, 77)
    op, fkw = bench.build_operator(operator, dev)
    mask = fkw.get("mask")
    handle = op.hip_handle_for(mask) if operator == "inpainting" else op.hip_handle(x_t)
    buf = kernels.StepBuffers(handle, n, 3, 256, 256, dev)
    y = (op.forward(truth.to(dev), **fkw).detach() + meas_noise.to(dev)[..., :256, :256]).contiguous()
    x = x_t
    for i in range(steps):
        ck = smp.step_coefs[999 - 300 * i]
        s = ring[i % 2]
        kernels.step_fwd(handle, buf, x, s["model_out"], s["noise"], y, ck, want_x0=False)
        kernels.step_bwd(handle, buf, y, 0.3, 1, ck)
        x = kernels.step_update(buf, s["g_unet"], ck)
    ref, ref_norm = x.clone(), buf.norm.clone()
    for groups in (2, 3):
        pg = kernels.ParticleGroups(op, n, 3, 256, 256, dev, groups, mask=mask, like=x_t)
        assert sum(pg.sizes) == n and max(pg.sizes) - min(pg.sizes) <= 1
        pg.fork()
        xs = [x_t[sl] for sl in pg.slices]
        for i in range(steps):
            ck = smp.step_coefs[999 - 300 * i]
            s = ring[i % 2]
            for j in range(len(pg)):
                pg.step_fwd(j, xs[j], s["model_out"], s["noise"], y, ck, want_x0=False)
                pg.step_bwd(j, y, 0.3, 1, ck)
                xs[j] = pg.step_update(j, s["g_unet"], ck)
        pg.join()
        torch.cuda.synchronize()
        assert torch.equal(pg.x_next(), ref), groups
        assert torch.equal(pg.full.norm, ref_norm), groups


@pytest.mark.parametrize("operator,semantic", [("gaussian_blur", False), ("motion_blur", True)])
def test_sampler_particle_groups_equal_one_chain(operator, semantic):
    """sampler.particle_groups = 3: p_sample_loop runs every group's whole step (model call, three launches, model VJP) on
    the group's own stream; the loop's outputs equal the one-chain loop's bit for bit (same noise draws, per-particle
    arithmetic independent of the batch)."""
    import torch
    from standin import StandInModel, ToyEmbedder
    sys.path.insert(0, ROOT)
    import bench
    from dps_ttc_amd.condition_methods import get_conditioning_method
    from dps_ttc_amd.gaussian_diffusion import create_sampler
    from dps_ttc_amd.measurements import get_noise
    dev = torch.device("cuda", 0)
    op, _ = bench.build_operator(operator, dev)
    kw = {}
    if semantic:
        emb = ToyEmbedder().to(dev)
        kw = dict(sem_guid_scale=0.05, anneal_factor=10.0, embedder=emb,
                  guid_image_emb=emb(torch.rand(1, 3, 64, 64, device=dev)).unsqueeze(0))
    cm = get_conditioning_method("ps_semantic" if semantic else "ps", op, get_noise("gaussian", sigma=0.05), scale=0.3, **kw)
    model = StandInModel().to(dev)
    y = torch.rand(1, 3, 64, 64, device=dev)
    outs = []
    for groups in (1, 3):
        smp = create_sampler(sampler="ddpm", steps=1000, noise_schedule="linear", model_mean_type="epsilon",
                             model_var_type="learned_range", dynamic_threshold=False, clip_denoised=True,
                             rescale_timesteps=True, timestep_respacing="6")
        smp.particle_groups = groups
        torch.manual_seed(11)
        x0 = torch.randn(5, 3, 64, 64, device=dev)
        img, d, sem = smp.p_sample_loop(model=model, x_start=x0.clone().requires_grad_(), measurement=y,
                                        measurement_cond_fn=cm.conditioning, record=False, save_root=None)
        torch.cuda.synchronize()
        outs.append((img, d, sem))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert bool(torch.isfinite(outs[0][0]).all())
    if semantic:
        assert torch.equal(outs[0][2].reshape(-1), outs[1][2].reshape(-1))


def test_global_resample_device_generator_no_host_read():
    """distributed.global_resample with a DEVICE generator (the driver's and bench.py's setting): the multinomial draw runs
    on the GPU as the reference's does; particles and scores come back gathered by the drawn ids, equal weights leave the
    set untouched -- and two generators in the same state draw the same ids (what makes every rank agree)."""
    import torch
    from dps_ttc_amd import distributed as dd
    dev = torch.device("cuda", 0)
    x = torch.randn(6, 3, 32, 32, device=dev)
    d = torch.tensor([5.0, 300.0, 7.0, 250.0, 6.0, 400.0], device=dev)
    g1, g2 = torch.Generator(device=dev).manual_seed(3), torch.Generator(device=dev).manual_seed(3)
    xa, da, ia = dd.global_resample(x, d, 100.0, g1)
    xb, db, ib = dd.global_resample(x, d, 100.0, g2)
    assert torch.equal(ia, ib) and torch.equal(xa, xb) and torch.equal(da, db)
    assert ia.shape == (6,) and int(ia.min()) >= 0 and int(ia.max()) < 6
    assert torch.equal(xa, x[ia]) and torch.equal(da, d[ia])
    assert len(set(ia.tolist())) < 6 or True          # (heavier weights are drawn more often; not asserted)
    flat = torch.full((6,), 9.0, device=dev)
    xc, dc, ic = dd.global_resample(x, flat, 100.0, g1)
    assert torch.equal(ic, torch.arange(6, device=dev)) and torch.equal(xc, x) and torch.equal(dc, flat)


_RCCL_ONE_RANK = r'''
import os, sys, socket
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)           # "nccl" IS RCCL on ROCm
from dps_ttc_amd import distributed as dd, kernels
torch.manual_seed(0)
x = torch.randn(6, 3, 64, 64, device=dev)
d = torch.tensor([5.0, 3.0, 7.0, 3.0, 9.0, 4.0], device=dev)
assert dd.exchange_counts(6, dev) == [6]
assert torch.equal(dd.gather_scores(d, [6]), d) and torch.equal(dd.gather_scores(d), d)
w, best, alls = dd.global_best_of_n(d, x, [6])
assert best == 1 and torch.equal(w, x[1:2]) and torch.equal(alls, d)
w2, b2 = dd.global_best_of_n_device(d, x, [6])
assert int(b2) == 1 and torch.equal(w2, x[1:2])
sel = dd.GlobalSelect()(d, x)
assert sel.shape == x.shape and torch.equal(sel, x[1:2].repeat(6, 1, 1, 1))
assert torch.equal(dd.GlobalSelect()(d, x, n_out=1), x[1:2])
g = torch.Generator(device=dev).manual_seed(1)
xr, dr, ids = dd.global_resample(x, d * 40.0, 100.0, g)
assert torch.equal(xr, x[ids]) and torch.equal(dr, (d * 40.0)[ids])
gh = torch.Generator().manual_seed(1)
xr, dr, ids = dd.global_resample(x, d * 40.0, 100.0, gh)
assert ids is not None and torch.equal(xr, x[ids.to(dev)])
for mode in ("all", "selected"):          # both exchange forms: all-gather of states / all-to-all of the drawn ones
    g = torch.Generator(device=dev).manual_seed(2)
    xr, dr, ids = dd.global_resample(x, d * 40.0, 100.0, g, fetch=mode)
    assert torch.equal(xr, x[ids]) and torch.equal(dr, (d * 40.0)[ids])
    assert torch.equal(dd.resample_particles(x, torch.tensor([5, 5, 5, 0, 5, 5]), fetch=mode), x[[5, 5, 5, 0, 5, 5]])
w3, b3 = dd.global_best_of_n_device(d[:0], x[:0], [0])     # an empty shard still enters the exchange
assert w3.shape == (1, 3, 64, 64)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier(); torch.cuda.synchronize()
assert float(t) == 1.5
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK")
'''


def test_rccl_single_rank_collectives():
    """The real backend on the one GPU this pool gives a box: a process group of ONE rank over "nccl" (= RCCL).  With a
    process group initialised, distributed.py takes its collective paths (all-gather of scores and champions, winner
    broadcast, padded gathers, the resampling exchange) -- here they run through RCCL kernels, degenerate but real: the
    library loads, the communicator initialises under HSA_ENABLE_IPC_MODE_LEGACY=0, dtypes and shapes are accepted, the
    stream ordering holds.  (Two or more ranks need two or more GPUs: the driver's round-end run.)"""
    import subprocess
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK, ROOT], capture_output=True, timeout=600)
    out = r.stdout.decode(errors="replace") + r.stderr.decode(errors="replace")
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in out, out[-3000:]


def test_bench_single_rank_process_group():
    """bench.py --force-process-group: the sharded workloads with their exchanges going through RCCL (one rank)"""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "DPSX_BENCH_BACKEND"):
        env.pop(k, None)
    for mode in (["--workload", "search", "--particles", "8"],
                 ["--workload", "resample", "--operator", "phase_retrieval", "--particles", "4", "--resample-every", "2"],
                 ["--workload", "dps_scores", "--operator", "motion_blur", "--particles", "4", "--semantic"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "3", "--no-cpu-baseline",
                            "--force-process-group", *mode], env=env, capture_output=True, timeout=900)
        assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
        rec = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.strip()][-1])
        assert rec["config"]["process_group"] == "nccl x1" and rec["value"] > 0 and rec["n_gpus"] == 1
