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


@pytest.mark.parametrize("sampler,n_paths,batch", [("ddpm", 3, 1), ("search_ddpm", 4, 2), ("ttc_ddim", 4, 2)])
def test_driver_two_ranks(tmp_path, sampler, n_paths, batch):
    """two rank processes (gloo rehearsal on one GPU): uneven group shards for the independent-particle loop (3 groups
    on 2 ranks), per-step global select for search_ddpm, global resampling for ttc_ddim; path numbering is global and
    the best-of-N image is the reported path's."""
    task = "gaussian_deblur_config.yaml"
    root, logs = _run_driver(tmp_path, task, sampler, n_paths, batch, ranks=2)
    best = _check_best_of_n(root, n_paths) if sampler == "ddpm" else None
    d = np.load(root / "00000_pathwise_distances.npy")
    assert d.shape == (n_paths,)
    assert f"best-of-{n_paths} = path#{int(np.argmin(d)) + 1} " in logs[0]
    if sampler == "search_ddpm":
        # every particle of every rank ends as a copy of the global per-step winner: all distances agree
        assert np.allclose(d, d[0], rtol=1e-6)
    if best is not None:
        assert "Path#3 " in logs[1] and "Path#1 " in logs[0] and "Path#2 " in logs[0]


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


@pytest.mark.parametrize("explicit", [False, True])
@pytest.mark.parametrize("operator", ["gaussian_blur", "motion_blur"])
def test_particle_groups_on_streams_match_one_chain(operator, explicit):
    """bench.py's timed loop runs the particles as independent groups on separate HIP streams (own operator handle and
    buffers each): per-particle results must not depend on the grouping -- x_{t-1} and the distances bit for bit."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from dps_ttc_amd import kernels
    from dps_ttc_amd.gaussian_diffusion import create_sampler
    dev = torch.device("cuda", 0)
    smp = create_sampler(sampler="ddpm", steps=1000, noise_schedule="linear", model_mean_type="epsilon",
                         model_var_type="learned_range", dynamic_threshold=False, clip_denoised=True,
                         rescale_timesteps=True, timestep_respacing="")
    n, steps = 6, 4
    x_t, ring, truth, meas_noise = bench.synth_inputs(n, 2, dev, 77)

    def make(count):
        op, fkw = bench.build_operator(operator, dev)
        handle = op.hip_handle(x_t)
        return op, handle, kernels.StepBuffers(handle, count, 3, 256, 256, dev)

    op, handle, buf = make(n)
    y = (op.forward(truth.to(dev)).detach() + meas_noise.to(dev)[..., :256, :256]).contiguous()

    def run(handle, buf, x, sl, stream):
        # explicit: the stream is handed to the launches (kernels.step_*(stream=), what bench.py does); else a stream context
        import contextlib
        kw = {"stream": stream} if explicit else {}
        with (contextlib.nullcontext() if explicit else torch.cuda.stream(stream)):
            for i in range(steps):
                ck = smp.step_coefs[999 - 300 * i]
                s = ring[i % 2]
                kernels.step_fwd(handle, buf, x, s["model_out"][sl], s["noise"][sl], y, ck, want_x0=not explicit, **kw)
                kernels.step_bwd(handle, buf, y, 0.3, 1, ck, **kw)
                x = kernels.step_update(buf, s["g_unet"][sl], ck, **kw)
        return x

    ref = run(handle, buf, x_t, slice(0, n), torch.cuda.current_stream()).clone()
    ref_norm = buf.norm.clone()
    outs = []
    for j in range(2):
        _, h2, b2 = make(n // 2)
        st = torch.cuda.Stream(device=dev)
        st.wait_stream(torch.cuda.current_stream())
        sl = slice(j * n // 2, (j + 1) * n // 2)
        outs.append((run(h2, b2, x_t[sl], sl, st), b2, st))
    for _, _, st in outs:
        torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([o[0] for o in outs]), ref)
    assert torch.equal(torch.cat([o[1].norm for o in outs]), ref_norm)
