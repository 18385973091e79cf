#!/usr/bin/env python3
"""Host cost of enqueueing one fused DPS step (three launches) -- with the current-stream lookup, inside a stream context,
and with the stream handed to the launches explicitly -- at a particle count small enough that the GPU is never the limit;
plus a cProfile of the plain form.   python tools/host_cost.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, bench
from dps_ttc_amd import kernels
from dps_ttc_amd.gaussian_diffusion import create_sampler
dev = torch.device("cuda", 0)
smp = create_sampler(sampler="ddpm", steps=1000, noise_schedule="linear", model_mean_type="epsilon", model_var_type="learned_range",
                     dynamic_threshold=False, clip_denoised=True, rescale_timesteps=True, timestep_respacing="")
n = 3
op, fkw = bench.build_operator("gaussian_blur", dev)
x_t, ring, truth, mn = bench.synth_inputs(n, 2, dev, 1)
y = (op.forward(truth.to(dev)).detach() + mn.to(dev)[..., :256, :256]).contiguous()
h = op.hip_handle(x_t); buf = kernels.StepBuffers(h, n, 3, 256, 256, dev)
ck = smp.step_coefs[500]; s = ring[0]
st = torch.cuda.Stream()
def step(x):
    kernels.step_fwd(h, buf, x, s["model_out"], s["noise"], y, ck, want_x0=False)
    kernels.step_bwd(h, buf, y, 0.3, 1, ck)
    return kernels.step_update(buf, s["g_unet"], ck)
def step_explicit(x):
    kernels.step_fwd(h, buf, x, s["model_out"], s["noise"], y, ck, want_x0=False, stream=st)
    kernels.step_bwd(h, buf, y, 0.3, 1, ck, stream=st)
    return kernels.step_update(buf, s["g_unet"], ck, stream=st)
for mode in ("plain", "with_stream", "explicit"):
    x = x_t
    for _ in range(50): x = step(x)
    torch.cuda.synchronize()
    K = 3000
    t0 = time.perf_counter()
    for _ in range(K):
        if mode == "plain": x = step(x)
        elif mode == "explicit": x = step_explicit(x)
        else:
            with torch.cuda.stream(st): x = step(x)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(mode, "enqueue %.1f us/step" % ((t1 - t0) / K * 1e6), "total %.1f us/step" % ((t2 - t0) / K * 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): x = step(x)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
