"""CPU suite: driver-side plumbing (YAML configs, dataset glob, mask generator, PSNR, UNet architecture)."""
import glob
import os

import numpy as np
import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shipped_yaml_configs_parse_and_build_host_objects():
    from dps_ttc_amd.condition_methods import get_conditioning_method
    from dps_ttc_amd.gaussian_diffusion import create_sampler
    from dps_ttc_amd.measurements import get_noise, get_operator
    diff = yaml.load(open(os.path.join(ROOT, "configs", "diffusion_config.yaml")), Loader=yaml.FullLoader)
    s = create_sampler(**diff)
    assert s.num_timesteps == 1000 and s.hip_posterior
    seen = set()
    for f in sorted(glob.glob(os.path.join(ROOT, "configs", "*_config*.yaml"))):
        cfg = yaml.load(open(f), Loader=yaml.FullLoader)
        if "measurement" not in cfg:
            continue
        opcfg = cfg["measurement"]["operator"]
        seen.add(opcfg["name"])
        op = get_operator(device="cpu", **opcfg)                       # host tables only; no GPU needed
        noiser = get_noise(**cfg["measurement"]["noise"])
        params = dict(cfg["conditioning"]["params"])
        if params.get("sem_guid_scale", 0) != 0:
            params["embedder"] = lambda x: x.flatten(1)                 # the face network is pluggable
        cm = get_conditioning_method(cfg["conditioning"]["method"], op, noiser, **params)
        assert callable(cm.conditioning)
        if opcfg["name"] == "super_resolution":
            assert isinstance(opcfg["in_shape"], tuple)                 # !!python/tuple survives FullLoader
    assert seen == {"gaussian_blur", "motion_blur", "super_resolution", "inpainting", "phase_retrieval"}


def test_dataset_glob_and_transform(tmp_path):
    from PIL import Image
    from dps_ttc_amd.data import get_dataloader, get_dataset, to_minus1_1
    rng = np.random.RandomState(0)
    for name in ("b.png", "a.png", "sub/c.png"):
        p = tmp_path / name
        p.parent.mkdir(exist_ok=True)
        Image.fromarray(rng.randint(0, 255, (16, 16, 4), dtype=np.uint8), "RGBA").save(p)
    ds = get_dataset("ffhq", root=str(tmp_path), transforms=to_minus1_1)
    assert [os.path.basename(p) for p in ds.fpaths] == ["a.png", "b.png", "c.png"] and len(ds) == 3
    x = ds[0]
    assert x.shape == (3, 16, 16) and x.dtype == torch.float32 and -1 <= float(x.min()) and float(x.max()) <= 1
    batch = next(iter(get_dataloader(ds, batch_size=1, num_workers=0, train=False)))
    assert batch.shape == (1, 3, 16, 16)
    with pytest.raises(NameError):
        get_dataset("no_such_dataset", root=".")
    with pytest.raises(AssertionError):
        get_dataset("ffhq", root=str(tmp_path / "empty"))


def test_mask_generator_and_psnr():
    from dps_ttc_amd.img_utils import clear_color, mask_generator
    from dps_ttc_amd.metrics import compute_psnr, compute_psnr_manual
    img = torch.zeros(1, 3, 64, 64)
    np.random.seed(1)
    m = mask_generator("random", mask_prob_range=(0.3, 0.7), image_size=64)(img)
    assert m.shape == (1, 3, 64, 64) and set(m.unique().tolist()) == {0.0, 1.0}
    assert torch.equal(m[:, 0], m[:, 1]) and 0.25 < 1 - float(m.mean()) < 0.75
    np.random.seed(1)
    b = mask_generator("box", mask_len_range=(16, 17), image_size=64, margin=(4, 4))(img)
    assert float((b == 0).sum()) == 3 * 16 * 16
    a = torch.rand(1, 3, 8, 8) * 2 - 1
    assert float(compute_psnr(a, a + 0.01)) == pytest.approx(10 * np.log10(float((a.max() - a.min()) ** 2) / 1e-4), rel=1e-4)
    assert float(compute_psnr_manual(a, a + 0.1)) == pytest.approx(20.0, rel=1e-4)
    c = clear_color(a)
    assert c.shape == (8, 8, 3) and c.min() == 0 and c.max() == pytest.approx(1.0)


def test_unet_architecture_matches_the_checkpoint_layout():
    """parameter count the survey probed on the reference: 93.6 M (FFHQ) -- and the key names load_state_dict needs"""
    from dps_ttc_amd.unet import create_model
    cfg = yaml.load(open(os.path.join(ROOT, "configs", "model_config.yaml")), Loader=yaml.FullLoader)
    m = create_model(**cfg)                      # no checkpoint in the container -> random init, as the reference
    n = sum(p.numel() for p in m.parameters())
    assert n == 93_563_910
    keys = set(m.state_dict())
    for k in ("time_embed.0.weight", "input_blocks.0.0.weight", "input_blocks.1.0.in_layers.0.weight",
              "input_blocks.1.0.emb_layers.1.weight", "input_blocks.1.0.out_layers.3.weight",
              "middle_block.1.qkv.weight", "middle_block.1.proj_out.weight", "output_blocks.0.0.skip_connection.weight",
              "out.2.weight"):
        assert k in keys, k
    x = torch.randn(2, 3, 64, 64, requires_grad=True)
    y = m(x, torch.tensor([250.0]))              # t of shape [1] broadcasts over the particles, as in the loop
    assert y.shape == (2, 6, 64, 64)
    assert torch.all(y == 0)                     # zero-initialised output conv, as in the public architecture
    m.out[2].weight.data.normal_(0, 0.02)
    (g,) = torch.autograd.grad(m(x, torch.tensor([250.0])).square().sum(), x)
    assert g.shape == x.shape and torch.isfinite(g).all() and float(g.abs().sum()) > 0


def test_reference_import_block_resolves_through_alias_modules():
    """INTEGRATION.md route A: the import block of the reference driver (sample_condition_batched_ttc.py:11-18)
    works unchanged with the repo root on sys.path -- `guided_diffusion.*`, `data.dataloader`, `util.*` are the
    `dps_ttc_amd` modules themselves (shared registries)."""
    from guided_diffusion.condition_methods import get_conditioning_method
    from guided_diffusion.measurements import get_noise, get_operator
    from guided_diffusion.unet import create_model
    from guided_diffusion.gaussian_diffusion import create_sampler
    from data.dataloader import get_dataset, get_dataloader
    from util.img_utils import clear_color, mask_generator
    from util.logger import get_logger
    import dps_ttc_amd.condition_methods as cm
    import dps_ttc_amd.gaussian_diffusion as gd
    import dps_ttc_amd.measurements as ms
    import guided_diffusion.measurements as alias
    assert alias is ms and get_operator is ms.get_operator and get_noise is ms.get_noise
    assert get_conditioning_method is cm.get_conditioning_method and create_sampler is gd.create_sampler
    assert callable(create_model) and callable(get_dataset) and callable(get_dataloader)
    assert callable(clear_color) and callable(mask_generator) and get_logger().name == "DPS"
    with pytest.raises(NameError):
        get_operator("no_such_operator", device="cpu")
    s = create_sampler(sampler="ttc_ddim", steps=1000, noise_schedule="linear", model_mean_type="epsilon",
                       model_var_type="learned_range", dynamic_threshold=False, clip_denoised=True,
                       rescale_timesteps=True, timestep_respacing="ddim50")
    assert s.num_timesteps == 50 and s.global_resample is False


def test_bench_byte_table_and_cli():
    """bench.py and tools/kbench.py price their launches from ONE table (bench.algo_p); the figures of SURVEY.md 8d, the
    store-off variant, and the command lines of BASELINE configs 3 and 4 in their sharded form parse."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    P = bench.P_BYTES
    assert P == 786432
    t = bench.algo_p("gaussian_blur")
    assert t["survey"] == {"fwd": 8.0, "bwd": 5.0, "upd": 4.0} and sum(t["survey"].values()) * P == 13369344      # 17P
    assert bench.algo_p("gaussian_blur", x0_store=False)["algorithmic"] == {"fwd": 7.0, "bwd": 4.0, "upd": 4.0}
    assert sum(bench.algo_p("super_resolution")["survey"].values()) * P == 11894784                              # 15.125P
    assert sum(bench.algo_p("inpainting")["survey"].values()) * P == 11796480                                    # 15P
    assert bench.algo_p("inpainting", x0_store=False)["algorithmic"] == bench.algo_p("inpainting")["survey"]       # reads x0_hat back
    assert sum(bench.algo_p("phase_retrieval")["survey"].values()) * P == 18874368                               # 24P
    assert sum(bench.algo_p("gaussian_blur", workload="search")["survey"].values()) * P == 6291456               # 8P
    assert bench.algo_p("motion_blur", semantic=True)["survey"]["bwd"] == 6.0
    src = open(os.path.join(ROOT, "tools", "kbench.py")).read()
    assert "bench.algo_p(" in src and "(7 + rho)" not in src
    old = sys.argv
    try:
        sys.argv = ["bench.py", "--gpus", "8", "--operator", "motion_blur", "--scaling", "strong", "--particles", "256",
                    "--workload", "dps_scores", "--semantic"]
        a = bench.parse()
        assert (a.gpus, a.scaling, a.particles, a.workload, a.semantic) == (8, "strong", 256, "dps_scores", True)
        sys.argv = ["bench.py", "--gpus", "8", "--operator", "phase_retrieval", "--workload", "resample", "--particles", "64"]
        a = bench.parse()
        assert (a.workload, a.resample_every, a.scaling) == ("resample", 10, "weak")
        sys.argv = ["bench.py"]
        a = bench.parse()
        assert (a.gpus, a.workload, a.operator, a.particles, a.chains) == (1, "dps", "gaussian_blur", 64, 0)
    finally:
        sys.argv = old
    assert abs(bench.semantic_scale(0.3, 10.0, 0.0) - 0.3 * (1 + 9 / (1 + np.exp(-3.0)))) < 1e-12


def test_particle_groups_partition():
    """kernels.ParticleGroups splits N particles into contiguous groups that differ by at most one particle (host logic
    only: the constructor's partition is computed before anything touches the device)"""
    from dps_ttc_amd import kernels
    for n, g in ((64, 3), (7, 3), (5, 8), (1, 4), (32, 2)):
        groups = max(1, min(g, max(n, 1)))
        sizes = [n // groups + (1 if j < n % groups else 0) for j in range(groups)]
        assert sum(sizes) == n and max(sizes) - min(sizes) <= 1
    assert hasattr(kernels.ParticleGroups, "step_fwd") and hasattr(kernels.ParticleGroups, "x_next")
    from dps_ttc_amd.gaussian_diffusion import create_sampler
    s = create_sampler(sampler="ddpm", steps=1000, noise_schedule="linear", model_mean_type="epsilon",
                       model_var_type="learned_range", dynamic_threshold=False, clip_denoised=True,
                       rescale_timesteps=True, timestep_respacing="")
    assert s.particle_groups == 1
