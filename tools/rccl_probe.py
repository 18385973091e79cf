import os, sys, socket, time
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
sys.path.insert(0, os.getcwd())
fd = os.dup(1); os.dup2(2, 1)
import torch, torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
from dps_ttc_amd import distributed as dd, kernels
x = torch.randn(64, 3, 256, 256, device=dev); d = torch.rand(64, device=dev) * 100
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
out = {}
small = torch.empty(64, device=dev); big = torch.empty_like(x); one = torch.empty(1, 3, 256, 256, device=dev)
out["all_gather 256 B"] = t(lambda: dist.all_gather_into_tensor(small, d))
out["all_gather 786 KB"] = t(lambda: dist.all_gather_into_tensor(one, x[:1]))
out["all_gather 50 MB"] = t(lambda: dist.all_gather_into_tensor(big, x))
out["broadcast 786 KB"] = t(lambda: dist.broadcast(one, src=0))
out["barrier"] = t(lambda: dist.barrier())
g = torch.Generator(device=dev).manual_seed(0)
out["global_resample (device draw)"] = t(lambda: dd.global_resample(x, d, 100.0, g), 10)
out["GlobalSelect n_out=1"] = t(lambda: dd.GlobalSelect()(d, x, n_out=1))
out["global_best_of_n_device"] = t(lambda: dd.global_best_of_n_device(d, x, [64]))
sg = dd.ScoreGather()
out["ScoreGather.submit"] = t(lambda: sg.submit(d))
w = torch.exp(-d / 100)
out["torch.multinomial 64 (device gen)"] = t(lambda: torch.multinomial(w, 64, replacement=True, generator=g))
out["kernels.gather 64 particles"] = t(lambda: kernels.gather(x, torch.arange(64, device=dev), validate=False))
dist.destroy_process_group()
os.dup2(fd, 1)
for k, v in out.items(): print(f"{k:40s} {v:9.1f} us")
