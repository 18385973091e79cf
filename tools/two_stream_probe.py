#!/usr/bin/env python3
"""Probe: the fused DPS step of N particles as ONE chain on one stream vs TWO independent half-chains (N/2 particles each,
own operator handle and buffers) on two streams.  The launches of a chain are dependent, the chains are not, so the
bandwidth-bound launch of one half can run beside the arithmetic-bound launch of the other.
    python tools/two_stream_probe.py [--operator gaussian_blur] [--particles 64] [--steps 200]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--operator", default="gaussian_blur")
    ap.add_argument("--particles", type=int, default=64)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--chains", type=int, default=2)
    ap.add_argument("--offset", action="store_true",
                    help="chain j starts j launches into its step (K1 of one group beside K2 / K3 of the other)")
    args = ap.parse_args()
    from dps_ttc_amd import kernels
    from dps_ttc_amd.gaussian_diffusion import create_sampler
    dev = torch.device("cuda", 0)
    smp = create_sampler(sampler="ddpm", steps=1000, noise_schedule="linear", model_mean_type="epsilon",
                         model_var_type="learned_range", dynamic_threshold=False, clip_denoised=True,
                         rescale_timesteps=True, timestep_respacing="")

    def make(n, seed):
        op, fkw = bench.build_operator(args.operator, dev)
        x_t, ring, truth, meas_noise = bench.synth_inputs(n, 3, dev, seed)
        yy = op.forward(truth.to(dev), **fkw).detach()
        y = (yy + meas_noise.to(dev)[..., :yy.shape[-2], :yy.shape[-1]]).contiguous()
        handle = op.hip_handle_for(fkw["mask"]) if args.operator == "inpainting" else op.hip_handle(x_t)
        buf = kernels.StepBuffers(handle, n, 3, 256, 256, dev)
        return dict(op=op, handle=handle, buf=buf, x=x_t, ring=ring, y=y)

    def launch(c, i, which):
        ck = smp.step_coefs[999 - (i % 1000)]
        s = c["ring"][i % 3]
        if which == 0:
            kernels.step_fwd(c["handle"], c["buf"], c["x"], s["model_out"], s["noise"], c["y"], ck)
        elif which == 1:
            kernels.step_bwd(c["handle"], c["buf"], c["y"], 0.3, 1, ck)
        else:
            c["x"] = kernels.step_update(c["buf"], s["g_unet"], ck)

    def step(c, i):
        # a chain with offset o enqueues launches o .. o+2 of the sequence K1 K2 K3 K1 ...: same work per call, shifted
        o = c.get("offset", 0)
        for q in range(o, o + 3):
            launch(c, i + q // 3, q % 3)

    def run(chains, streams):
        for i in range(20):
            for c, st in zip(chains, streams):
                with torch.cuda.stream(st):
                    step(c, i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            for c, st in zip(chains, streams):
                with torch.cuda.stream(st):
                    step(c, 20 + i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps

    n = args.particles
    one = [make(n, 1234)]
    t1 = run(one, [torch.cuda.current_stream()])
    print(f"1 chain  x {n:3d} particles: {t1 * 1e6:7.1f} us/step  {n / t1 / 1e3:7.1f} k particle-steps/s")
    k = args.chains
    many = [make(n // k, 1234 + j) for j in range(k)]
    if args.offset:
        for j, c in enumerate(many):
            c["offset"] = j % 3
            for q in range(c["offset"]):          # run the launches the shifted sequence skips at its start
                launch(c, 0, q)
    streams = [torch.cuda.Stream() for _ in range(k)]
    tk = run(many, streams)
    print(f"{k} chains x {n // k:3d} particles: {tk * 1e6:7.1f} us/step  {n / tk / 1e3:7.1f} k particle-steps/s")
    same = run(many, [torch.cuda.current_stream()] * k)
    print(f"{k} chains x {n // k:3d} particles on ONE stream: {same * 1e6:7.1f} us/step")


if __name__ == "__main__":
    main()
