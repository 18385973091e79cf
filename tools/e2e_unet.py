#!/usr/bin/env python3
"""End-to-end DPS steps with the real FFHQ UNet architecture (random init), un-gated context number
(SURVEY.md 8d): how much of a real step is the UNet (PyTorch-ROCm) and how much the HIP tail.

    python tools/e2e_unet.py [--particles 16] [--steps 5] [--operator gaussian_blur]
"""
import argparse
import os
import sys
import time

import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=16)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--operator", default="gaussian_blur")
    args = ap.parse_args()
    from dps_ttc_amd.condition_methods import get_conditioning_method
    from dps_ttc_amd.gaussian_diffusion import create_sampler
    from dps_ttc_amd.measurements import get_noise
    from dps_ttc_amd.unet import create_model
    dev = torch.device("cuda", 0)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model = create_model(**yaml.load(open(os.path.join(root, "configs", "model_config.yaml")), Loader=yaml.FullLoader))
    model = model.to(dev).eval()
    model.out[2].weight.data.normal_(0, 0.02)          # the zero-initialised head would make the VJP trivially zero
    op, fkw = bench.build_operator(args.operator, dev)
    cm = get_conditioning_method("ps", op, get_noise("gaussian", sigma=0.05), scale=0.3)
    smp = create_sampler(sampler="ddpm", steps=1000, noise_schedule="linear", model_mean_type="epsilon",
                         model_var_type="learned_range", dynamic_threshold=False, clip_denoised=True,
                         rescale_timesteps=True, timestep_respacing="")
    n = args.particles
    x = torch.randn(n, 3, 256, 256, device=dev)
    truth = torch.rand(1, 3, 256, 256, device=dev) * 2 - 1
    y = op.forward(truth, **fkw).detach()
    y = (y + 0.05 * torch.randn_like(y)).contiguous()
    handle = op.hip_handle_for(fkw["mask"]) if args.operator == "inpainting" else op.hip_handle(x)
    for i in range(2):
        x, _ = smp.dps_step(model, x, 999 - i, y, cm, fkw, handle)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        x, norm = smp.dps_step(model, x, 997 - i, y, cm, fkw, handle)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    # UNet alone (forward + VJP)
    xx = x.detach().requires_grad_()
    t = smp._model_timesteps(dev)[500:501]
    g = torch.randn(n, 6, 256, 256, device=dev)
    for _ in range(2):
        out = model(xx, t)
        torch.autograd.grad(out, xx, g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = model(xx, t)
        torch.autograd.grad(out, xx, g)
    torch.cuda.synchronize()
    du = (time.perf_counter() - t0) / args.steps
    print(f"N={n} {args.operator}: full DPS step {dt * 1e3:.1f} ms ({n / dt:.1f} particle-steps/s); "
          f"UNet fwd+VJP alone {du * 1e3:.1f} ms -> HIP tail + glue {max(dt - du, 0) * 1e3:.2f} ms "
          f"({100 * max(dt - du, 0) / dt:.1f} % of the step); norm[0]={float(norm[0]):.3f}")


if __name__ == "__main__":
    main()
