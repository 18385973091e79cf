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
