import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from dps_ttc_amd import kernels
from dps_ttc_amd.condition_methods import get_conditioning_method
from dps_ttc_amd.gaussian_diffusion import create_sampler
from dps_ttc_amd.measurements import get_noise
dev = torch.device("cuda", 0)
n = 64
op, fkw = bench.build_operator("gaussian_blur", dev)
cm = get_conditioning_method("ps", op, get_noise("gaussian", sigma=0.05), scale=0.3)
smp = create_sampler(sampler="ddpm", steps=1000, noise_schedule="linear", model_mean_type="epsilon", model_var_type="learned_range", dynamic_threshold=False, clip_denoised=True, rescale_timesteps=True, timestep_respacing="")
x_t, ring, truth, mn = bench.synth_inputs(n, 3, dev, 1234)
yy = op.forward(truth.to(dev)).detach(); y = (yy + mn.to(dev)[..., :256, :256]).contiguous()
handle = op.hip_handle(x_t); buf = kernels.StepBuffers(handle, n, 3, 256, 256, dev); spec = cm.fused_spec()
def step(i, x):
    ck = smp.step_coefs[999 - (i % 1000)]; s = ring[i % 3]
    kernels.step_fwd(handle, buf, x, s["model_out"], s["noise"], y, ck)
    kernels.step_bwd(handle, buf, y, spec["scale"], spec["power"], ck)
    return kernels.step_update(buf, s["g_unet"], ck)
x = x_t
for i in range(20): x = step(i, x)
torch.cuda.synchronize()
for K in (20, 50, 100, 200, 400, 200, 20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K): x = step(i, x)
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); tt = time.perf_counter() - t0
    print(f"K={K:4d}  host enqueue {th/K*1e6:7.1f} us/step   total {tt/K*1e6:7.1f} us/step", flush=True)
