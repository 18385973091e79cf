"""Does a collective leave the steps that follow it slower?  (one-rank RCCL group on one GPU)"""
import os, sys, socket, time
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
sys.path.insert(0, os.getcwd())
fd = os.dup(1); os.dup2(2, 1)
import torch, torch.distributed as dist
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
from dps_ttc_amd import distributed as dd, kernels
from dps_ttc_amd.gaussian_diffusion import create_sampler
smp = create_sampler(sampler="ddpm", steps=1000, noise_schedule="linear", model_mean_type="epsilon", model_var_type="learned_range",
                     dynamic_threshold=False, clip_denoised=True, rescale_timesteps=True, timestep_respacing="")
op, _ = bench.build_operator("gaussian_blur", dev)
x_t, ring, truth, mn = bench.synth_inputs(64, 2, dev, 1)
y = (op.forward(truth.to(dev)).detach() + mn.to(dev)[..., :256, :256]).contiguous()
h = op.hip_handle(x_t); buf = kernels.StepBuffers(h, 64, 3, 256, 256, dev)
def steps(x, n=100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        ck = smp.step_coefs[999 - i]; s = ring[i % 2]
        kernels.step_fwd(h, buf, x, s["model_out"], s["noise"], y, ck, want_x0=False)
        kernels.step_bwd(h, buf, y, 0.3, 1, ck)
        x = kernels.step_update(buf, s["g_unet"], ck)
    torch.cuda.synchronize(); return x, (time.perf_counter() - t0) / n * 1e6
out = []
x, t = steps(x_t); x, t = steps(x); out.append(("baseline", t))
small = torch.empty(64, device=dev); dist.all_gather_into_tensor(small, buf.norm)
x, t = steps(x); out.append(("after a 256 B all_gather of buf.norm", t))
pool = torch.empty_like(x); dist.all_gather_into_tensor(pool, x.contiguous())
x, t = steps(x); out.append(("after a 50 MB all_gather of x (x_next buffer)", t))
x2 = kernels.gather(pool, torch.arange(64, device=dev), validate=False); del pool
x, t = steps(x2); out.append(("continuing from a gathered copy", t))
g = torch.Generator(device=dev).manual_seed(0)
x3, _, _ = dd.global_resample(x, buf.norm, 100.0, g)
x, t = steps(x3); out.append(("after global_resample", t))
x, t = steps(x); out.append(("100 steps later", t))
dist.destroy_process_group()
os.dup2(fd, 1)
for k, v in out: print(f"{k:50s} {v:8.1f} us/step")
