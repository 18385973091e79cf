#!/usr/bin/env python3
"""Per-step best-of-N (search_ddpm) micro-bench: S1 + score + argmin + winner replication, N particles, 256x256."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--operator", default="gaussian_blur")
    ap.add_argument("--particles", type=int, default=64)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--unfused", action="store_true", help="S1 and the scoring launch separately (the sequence other operators run)")
    ap.add_argument("--one", action="store_true",
                    help="the single-state step (dpsx_search_step_one_f32): one state particle, N proposals, winner copied once")
    args = ap.parse_args()
    from dps_ttc_amd import kernels
    from dps_ttc_amd.gaussian_diffusion import create_sampler
    dev = torch.device("cuda", 0)
    n = args.particles
    op, fkw = bench.build_operator(args.operator, dev)
    smp = create_sampler(sampler="search_ddpm", steps=1000, noise_schedule="linear", model_mean_type="epsilon",
                         model_var_type="learned_range", dynamic_threshold=False, clip_denoised=True,
                         rescale_timesteps=True, timestep_respacing="")
    x_t, ring, truth, meas_noise = bench.synth_inputs(n, 2, dev, 1234)
    yy = op.forward(truth.to(dev), **fkw).detach()
    mn = meas_noise.to(dev)
    if mn.shape[-1] < yy.shape[-1]:                      # phase retrieval measures on the oversampled grid
        mn = 0.05 * torch.randn(yy.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    y = (yy + mn[..., :yy.shape[-2], :yy.shape[-1]]).contiguous()
    handle = op.hip_handle_for(fkw["mask"]) if args.operator == "inpainting" else op.hip_handle(x_t)
    ck = smp.step_coefs[500]

    def step(i, x):
        s = ring[i % 2]
        if args.one:                # x: [1, C, H, W]; the model output of the one state particle
            return handle.search_step_one(x, s["model_out"][:1], s["noise"], y, ck)[0]
        if not args.unfused:        # what SearchDDPM.search_step runs: dpsx_search_step_f32 + dpsx_replicate_f32
            return handle.search_step(x, s["model_out"], s["noise"], y, ck)[0]
        _, sample = kernels.posterior_fwd(x, s["model_out"], s["noise"], ck, want_x0=False)
        if args.unfused:
            costs, best, _ = handle.score_argmin(sample, y)
            return kernels.replicate(sample, best)
        return None

    x = x_t[:1].contiguous() if args.one else x_t
    for i in range(3):
        x = step(i, x)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
    for i, (a, b) in enumerate(evs):
        a.record()
        x = step(i, x)
        b.record()
    torch.cuda.synchronize()
    ts = np.array([a.elapsed_time(b) for a, b in evs]) * 1e3
    if args.one:
        # priced on the bytes this form moves (3P per particle-step); the replicated form's 8P would read as more than the
        # HBM peak at this step time, which says the form skips traffic, not that it streams faster than the memory
        roof = 8000e9 / (8 * bench.P_BYTES)
        print(f"search step (one state particle) N={n} {args.operator}: avg {ts.mean():.1f} us  min {ts.min():.1f} us  "
              f"{n / ts.mean() * 1e6:.0f} particle-steps/s ({n / ts.mean() * 1e6 / roof:.2f} of the replicated form's 8P "
              f"roofline of {roof / 1e6:.2f} M/s)  {3 * bench.P_BYTES * n / ts.mean() / 1e3:.0f} GB/s on the 3P it moves")
        return
    # the launches one by one (each timed alone, back to back with itself)
    s0 = ring[0]
    _, sample = kernels.posterior_fwd(x, s0["model_out"], s0["noise"], ck, want_x0=False)
    costs, best, _ = handle.score_argmin(sample, y)
    parts = {"search_step w/o replicate": lambda: handle.search_step(x, s0["model_out"], s0["noise"], y, ck, replicate=False),
             "S1 (no x0_hat store)": lambda: kernels.posterior_fwd(x, s0["model_out"], s0["noise"], ck, want_x0=False),
             "score + finalize/select": lambda: handle.score_argmin(sample, y),
             "replicate": lambda: kernels.replicate(sample, best)}
    for name, fn in parts.items():
        for _ in range(3):
            fn()
        e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
        for a_, b_ in e:
            a_.record()
            fn()
            b_.record()
        torch.cuda.synchronize()
        tt = np.array([a_.elapsed_time(b_) for a_, b_ in e]) * 1e3
        print(f"    {name:28s} avg {tt.mean():6.1f} us  min {tt.min():6.1f} us")
    algo = 8 * bench.P_BYTES * n
    print(f"search step N={n} {args.operator}: avg {ts.mean():.1f} us  min {ts.min():.1f} us  "
          f"{n / ts.mean() * 1e6:.0f} particle-steps/s  {algo / ts.mean() / 1e3:.0f} GB/s algorithmic (8P/particle)")


if __name__ == "__main__":
    main()
