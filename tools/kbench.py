#!/usr/bin/env python3
"""Kernel micro-bench: time each launch of the fused DPS step (and the plain operator calls) in isolation.

    python tools/kbench.py [--operator gaussian_blur] [--particles 64] [--reps 30] [--only fwd,bwd,upd,op,adj]

Prints one line per launch: avg / min microseconds and achieved GB/s against the algorithmic bytes
(SURVEY.md 8d).  Used under rocprofv3 for the per-kernel profiles in profiles/.
"""
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
    ap.add_argument("--only", default="fwd,bwd,upd,op,adj,score")
    ap.add_argument("--sigma", type=float, default=3.0, help="gaussian_blur only: another radius bucket")
    ap.add_argument("--norm-in-fwd", action="store_true", help="K1 finishes the norm itself (last block of a particle)")
    ap.add_argument("--no-x0", action="store_true", help="K1 does not write x0_hat out (blur / resize; the `ps` loop's setting)")
    args = ap.parse_args()
    from dps_ttc_amd import kernels
    from dps_ttc_amd.gaussian_diffusion import create_sampler
    dev = torch.device("cuda", 0)
    n = args.particles
    op, fkw = bench.build_operator(args.operator, dev, sigma=args.sigma)
    smp = create_sampler(sampler="ddpm", steps=1000, noise_schedule="linear", model_mean_type="epsilon",
                         model_var_type="learned_range", dynamic_threshold=False, clip_denoised=True,
                         rescale_timesteps=True, timestep_respacing="")
    x_t, ring, truth, meas_noise = bench.synth_inputs(n, 2, dev, 1234)
    yy = op.forward(truth.to(dev), **fkw).detach()
    mn = meas_noise.to(dev)
    if mn.shape[-1] < yy.shape[-1]:                      # phase retrieval measures on the oversampled grid
        mn = 0.05 * torch.randn(yy.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    y = (yy + mn[..., :yy.shape[-2], :yy.shape[-1]]).contiguous()
    handle = op.hip_handle_for(fkw["mask"]) if args.operator == "inpainting" else op.hip_handle(x_t)
    buf = kernels.StepBuffers(handle, n, 3, 256, 256, dev)
    ck = smp.step_coefs[500]
    P = bench.P_BYTES
    # one byte table for bench.py and this tool (bench.algo_p): the configuration launched (x0_hat store on / off)
    tbl = bench.algo_p(args.operator, x0_store=not args.no_x0)["algorithmic"]
    rho = bench.RHO[args.operator]
    u = torch.randn((n,) + tuple(y.shape[1:]), device=dev)
    cases = {
        "fwd": (lambda i: kernels.step_fwd(handle, buf, x_t, ring[i % 2]["model_out"], ring[i % 2]["noise"], y, ck,
                                           finalize_norm=args.norm_in_fwd, want_x0=not args.no_x0), tbl["fwd"] * P),
        # as in the loop: the norm is finalised from the forward half's partials (--norm-in-fwd: by K1's own tail)
        "bwd": (lambda i: (setattr(buf, "norm_ready", args.norm_in_fwd), kernels.step_bwd(handle, buf, y, 0.3, 1, ck)),
                tbl["bwd"] * P),
        "upd": (lambda i: kernels.step_update(buf, ring[i % 2]["g_unet"], ck), tbl["upd"] * P),
        "op": (lambda i: handle.forward(x_t), (1 + rho) * P),
        "adj": (lambda i: handle.adjoint(u, x=x_t, in_hw=(256, 256)), (1 + rho) * P),
        "score": (lambda i: handle.score(x_t, y), 1 * P),
    }
    cases["fwd"][0](0)           # partial sums for a stand-alone "bwd"
    for name in args.only.split(","):
        fn, bytes_pp = cases[name]
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
        for i, (a, b) in enumerate(evs):
            a.record()
            fn(i)
            b.record()
        torch.cuda.synchronize()
        ts = np.array([a.elapsed_time(b) for a, b in evs]) * 1e3
        label = f", x0_hat store {'off' if args.no_x0 else 'on'}" if name in ("fwd", "bwd") else ""
        # the rate is on the bytes the launch MOVES (PMC, profiles/traffic.json: N = 64, x0_hat store off, sigma = 3) where
        # they are on file -- the survey's compulsory bytes include streams this design never writes (the zero variance
        # half of g_model_out, grad_x_direct), so priced on them a short launch would read as faster than the memory
        moved = bench.load_traffic(args.operator).get(name) if (args.no_x0 and args.sigma == 3.0 and not args.norm_in_fwd) else None
        if moved:
            moved = moved * n / 64.0
            print(f"{name:6s} avg {ts.mean():8.1f} us  min {ts.min():8.1f} us   {moved / ts.mean() / 1e3:8.1f} GB/s moved "
                  f"({moved / n / P:.2f} P/particle by PMC; algorithmic {bytes_pp / P:.2f} P{label})", flush=True)
        else:
            print(f"{name:6s} avg {ts.mean():8.1f} us  min {ts.min():8.1f} us   "
                  f"{bytes_pp * n / ts.mean() / 1e3:8.1f} GB/s algorithmic ({bytes_pp / P:.2f} P/particle{label})", flush=True)


if __name__ == "__main__":
    main()
