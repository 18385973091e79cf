#!/usr/bin/env python3
"""Headline benchmark: particles x denoise-steps / s of the DPS hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts N rank processes itself, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): FFHQ-shaped 256x256 RGB,
Gaussian deblur sigma=3.0 (61x61 kernel), N=64 particles per GPU, 'ps' conditioning (scale 0.3).
One step = the whole non-UNet DPS step over the particle batch -- S1 posterior arithmetic, A(x0_hat) + residual
+ norm, the cotangent back through A^T / clamp to the UNet output, and the update -- three fused HIP launches.
The UNet is not in the timed region: its output (model_out) and its VJP (g_unet) are synthetic device-resident
tensors, cycled through a small ring so no step re-reads warm lines; x_t chains from step to step
as in the real loop, t cycles 999 -> 0 with the real fp32 tables.

Multi-GPU: particles shard across ranks (64 per GPU, weak scaling), no collective inside the step; the timed
region ends with the global best-of-N select over all ranks' particles (RCCL all-gather of the per-rank
champions, device-side pick, no host read).

Rank 0 prints ONE JSON line; `roofline` is for the dominant kernel from live HIP-event timing, `cpu_baseline`
is the oracle (a CPU port of the reference path) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

P_BYTES = 3 * 256 * 256 * 4                    # one fp32 particle image
HBM_PEAK_GBS = 8000.0                          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
VALU_F32_PEAK_TFLOPS = 157.3                   # MI355X_MICROARCH.md: fp32 vector peak (256 CUs x 128 FMA lanes x 2.4 GHz)

# algorithmic bytes per particle of the three launches, in units of P.
#   fwd = S1 6P + S2 (1 + rho)P      bwd = S3 (4 + rho)P      upd = S4 4P      rho = |y| / |x|
# phase retrieval stores the complex cotangent instead of r: S2 = 5.5P, S3 = 8.5P.
ALGO_P = {
    "gaussian_blur": {"fwd": 8.0, "bwd": 5.0, "upd": 4.0},
    "motion_blur": {"fwd": 8.0, "bwd": 5.0, "upd": 4.0},
    "super_resolution": {"fwd": 7.0625, "bwd": 4.0625, "upd": 4.0},
    "inpainting": {"fwd": 7.0, "bwd": 4.0, "upd": 4.0},
    "phase_retrieval": {"fwd": 11.5, "bwd": 8.5, "upd": 4.0},
}
KERNELS = {
    "gaussian_blur": {"fwd": "S1 + A(x0_hat) + residual + norm partials (k_blur_sep_fwd<3,POST,RESID>)",
                      "bwd": "A^T + clamp gate + -b*coef (k_blur_sep_adj<3,EPI>)"},
    "motion_blur": {"fwd": "S1 + tap-list A(x0_hat) + residual + norm partials (k_blur_taps<POST,RESID>)",
                    "bwd": "tap-list A^T on the image domain: plain + mirrored windows in one scan, gate, -b*coef (k_blur_taps_adj<EPI>)"},
    "super_resolution": {"fwd": "S1 + resize W,H passes + residual + norm partials (k_resize_rows_fwd<POST,RESID>)",
                         "bwd": "resize adjoint (inverse tables) + clamp gate + -b*coef (k_resize_adj<EPI>)"},
    "inpainting": {"fwd": "S1 + mask residual + norm partials (k_mask_step_fwd)",
                   "bwd": "mask A^T + clamp gate + -b*coef (k_mask_step_bwd)"},
    "phase_retrieval": {"fwd": "S1 + row FFTs, column FFT + modulus residual + inverse columns (k_pr_rows_fwd + k_pr_cols)",
                        "bwd": "inverse row FFTs + crop + clamp gate + -b*coef (k_pr_rows_inv)"},
}
WORKLOADS = {
    "gaussian_blur": "Gaussian deblur (sigma=3.0, k=61)",
    "motion_blur": "motion deblur (synthetic 61x61 path kernel, intensity 0.5)",
    "super_resolution": "x4 super-resolution (bicubic, 256->64)",
    "inpainting": "inpainting (Bernoulli(0.5) mask)",
    "phase_retrieval": "phase retrieval (oversample 2.0, 384x384 spectrum)",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--particles", type=int, default=64, help="particles per GPU")
    ap.add_argument("--operator", default="gaussian_blur", choices=sorted(ALGO_P))
    ap.add_argument("--x0-store", action="store_true",
                    help="K1 also writes the x0_hat image out.  The `ps` loop reads it nowhere after K1 (the backward half "
                         "works from the clamp gate), so p_sample_loop -- and this bench -- ask for it only when something "
                         "consumes it (a progress snapshot, the semantic term); inpainting always writes it (its backward half reads it)")
    ap.add_argument("--chains", type=int, default=0,
                    help="independent particle groups per GPU, each on its own HIP stream (1 = one chain of N; default: 3, "
                         "2 for phase retrieval, whose launches hold 52 KB of LDS per workgroup and gain nothing from a third)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-particles", type=int, default=64)
    ap.add_argument("--cpu-steps", type=int, default=5)
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------- self-launch
def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: this process starts N fresh rank processes (one per GPU, the
    same environment torch.distributed.run would give them), relays rank 0's JSON line and exits non-zero if any
    rank fails.  It never touches the GPU itself (no torch import, no HIP call, no exec)."""
    import socket
    import subprocess
    import threading
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    captured = []
    reader = threading.Thread(target=lambda: captured.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("DPSX_BENCH_TIMEOUT", "1500"))
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad or time.time() > deadline:
            failed = bad[0].returncode if bad else 124
            for p in procs:                     # the exact children started above, by handle
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.2)
    reader.join(timeout=10)
    out = (captured[0] if captured else b"").decode(errors="replace")
    if failed is None:
        failed = next((p.returncode for p in procs if p.returncode), None)
    if failed:
        sys.stderr.write(out)
        raise SystemExit(f"bench.py: a rank process failed (exit {failed})")
    # stdout carries the ONE JSON line; whatever else a rank's libraries print there (e.g. gloo's connection
    # banner) goes to stderr
    for ln in out.splitlines():
        (sys.stdout if ln.lstrip().startswith('{"metric"') else sys.stderr).write(ln + "\n")
    sys.stdout.flush()


# ---------------------------------------------------------------------------------------------- workload
def synth_inputs(n, ring, device, seed):
    """torch.manual_seed(1234) on the CPU, then copied to HBM."""
    import torch
    g = torch.Generator().manual_seed(seed)
    shape = (n, 3, 256, 256)
    x_t = torch.randn(shape, generator=g)
    sets = []
    for _ in range(ring):
        eps = torch.randn(shape, generator=g)
        v = torch.rand(shape, generator=g) * 2 - 1
        sets.append({"model_out": torch.cat([eps, v], dim=1).to(device),
                     "noise": torch.randn(shape, generator=g).to(device),
                     "g_unet": (torch.randn(shape, generator=g) * 1e-2).to(device)})
    truth = torch.rand((1, 3, 256, 256), generator=g) * 2 - 1
    meas_noise = torch.randn((1, 3, 384, 384), generator=g) * 0.05
    return x_t.to(device), sets, truth, meas_noise


def build_operator(name, device, sigma=3.0):
    """sigma: only tools/kbench.py passes another value (other radius buckets); the bench itself runs sigma = 3.0"""
    import numpy as np
    import torch
    from dps_ttc_amd.measurements import get_operator
    if name == "gaussian_blur":
        return get_operator("gaussian_blur", kernel_size=61, intensity=sigma, device=device), {}
    if name == "motion_blur":
        np.random.seed(0)
        return get_operator("motion_blur", kernel_size=61, intensity=0.5, device=device), {}
    if name == "super_resolution":
        return get_operator("super_resolution", in_shape=(1, 3, 256, 256), scale_factor=4, device=device), {}
    if name == "phase_retrieval":
        return get_operator("phase_retrieval", oversample=2.0, device=device), {}
    if name != "inpainting":
        raise SystemExit(f"unknown operator {name}")
    g = torch.Generator().manual_seed(7)
    mask = (torch.rand((1, 1, 256, 256), generator=g) < 0.5).float().to(device)
    return get_operator("inpainting", device=device), {"mask": mask}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size must match")

    # one hardware queue per HIP stream: with the runtime's default of 4 the particle groups' streams, the null stream
    # and RCCL's can end up sharing queues (seen once as a 2.4x slower step with three groups); must be set before the
    # HIP runtime initialises
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import numpy as np
    import torch
    import torch.distributed as dist
    shared_gpu = world > max(torch.cuda.device_count(), 1)
    local = local % max(torch.cuda.device_count(), 1)      # rehearsals may put several ranks on one GPU
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("DPSX_BENCH_BACKEND", "nccl")   # "nccl" is RCCL; "gloo" only to rehearse on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from dps_ttc_amd import distributed as dd
    from dps_ttc_amd import kernels
    from dps_ttc_amd.condition_methods import get_conditioning_method
    from dps_ttc_amd.gaussian_diffusion import create_sampler
    from dps_ttc_amd.measurements import get_noise

    n = args.particles
    op, fkw = build_operator(args.operator, device)
    cm = get_conditioning_method("ps", op, get_noise("gaussian", sigma=0.05), scale=0.3)
    smp = create_sampler(sampler="ddpm", steps=1000, noise_schedule="linear", model_mean_type="epsilon",
                         model_var_type="learned_range", dynamic_threshold=False, clip_denoised=True,
                         rescale_timesteps=True, timestep_respacing="")
    x_t, ring, truth, meas_noise = synth_inputs(n, 3, device, 1234 + rank)
    yy = op.forward(truth.to(device), **fkw).detach()
    y = (yy + meas_noise.to(device)[..., :yy.shape[-2], :yy.shape[-1]]).contiguous()
    handle = op.hip_handle_for(fkw["mask"]) if args.operator == "inpainting" else op.hip_handle(x_t)
    buf = kernels.StepBuffers(handle, n, 3, 256, 256, device)
    spec = cm.fused_spec()
    counts = [n] * world

    def step(i, x, timers=None):
        t = 999 - (i % 1000)
        ck = smp.step_coefs[t]
        s = ring[i % len(ring)]
        if timers is not None:
            timers[0].record()
        kernels.step_fwd(handle, buf, x, s["model_out"], s["noise"], y, ck, want_x0=args.x0_store)
        if timers is not None:
            timers[1].record()
        kernels.step_bwd(handle, buf, y, spec["scale"], spec["power"], ck)
        if timers is not None:
            timers[2].record()
        out = kernels.step_update(buf, s["g_unet"], ck)
        if timers is not None:
            timers[3].record()
        return out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the timed loop runs the N particles as `chains` independent groups, each with its own operator handle and
    # buffers on its own HIP stream: the three launches of a step are dependent, the groups are not, so the
    # bandwidth-bound launch of one group runs beside the arithmetic-bound launch of another (DESIGN.md, section 5:
    # inside one chain the load / compute / store phases of the tile kernels add up).  Per-particle results do not
    # depend on the grouping (tests/test_driver_gpu.py checks it bit for bit).
    nch = max(1, min(args.chains if args.chains > 0 else (2 if args.operator == "phase_retrieval" else 3), n))
    if shared_gpu:
        nch = 1         # rank processes time-slicing ONE GPU (gloo rehearsal): several queues per process make it crawl
    sizes = [n // nch + (1 if j < n % nch else 0) for j in range(nch)]      # groups may differ by one particle
    starts = [sum(sizes[:j]) for j in range(nch)]
    groups = []
    for j in range(nch):
        m = sizes[j]
        if nch == 1:
            gop, ghandle, gbuf, gstream = op, handle, buf, torch.cuda.current_stream()
        else:
            gop, _ = build_operator(args.operator, device)
            ghandle = gop.hip_handle_for(fkw["mask"]) if args.operator == "inpainting" else gop.hip_handle(x_t)
            gbuf = kernels.StepBuffers(ghandle, m, 3, 256, 256, device)
            gstream = torch.cuda.Stream(device=device)
        sl = slice(starts[j], starts[j] + m)
        groups.append({"op": gop, "handle": ghandle, "buf": gbuf, "stream": gstream, "x": x_t[sl],
                       "ring": [{k: v[sl] for k, v in s.items()} for s in ring]})

    def group_step(g, i, timers=None):
        # the group's stream is passed to the launches explicitly: no stream context to enter, no current-stream lookup
        # (tools/host_cost.py: 96 -> 35 us of host time per step for three groups)
        ck = smp.step_coefs[999 - (i % 1000)]
        s = g["ring"][i % len(ring)]
        st = g["stream"]
        if timers is not None:
            timers[0].record(st)
        kernels.step_fwd(g["handle"], g["buf"], g["x"], s["model_out"], s["noise"], y, ck, want_x0=args.x0_store, stream=st)
        if timers is not None:
            timers[1].record(st)
        kernels.step_bwd(g["handle"], g["buf"], y, spec["scale"], spec["power"], ck, stream=st)
        if timers is not None:
            timers[2].record(st)
        g["x"] = kernels.step_update(g["buf"], s["g_unet"], ck, stream=st)
        if timers is not None:
            timers[3].record(st)

    def run_steps(first, count, timers=None):
        for i in range(count):
            for j, g in enumerate(groups):
                group_step(g, first + i, timers[i] if timers is not None and j == 0 else None)

    def join_groups():
        cur = torch.cuda.current_stream()
        for g in groups:
            cur.wait_stream(g["stream"])
        if nch == 1:
            return groups[0]["buf"].norm, groups[0]["x"]
        return torch.cat([g["buf"].norm for g in groups]), torch.cat([g["x"] for g in groups])

    for g in groups:
        g["stream"].wait_stream(torch.cuda.current_stream())
    run_steps(0, args.warmup)
    norm_all, x_all = join_groups()
    dd.global_best_of_n_device(norm_all.clone(), x_all, counts)        # warm the select (and the collectives) too
    barrier()
    for g in groups:
        g["stream"].wait_stream(torch.cuda.current_stream())
    t0 = time.perf_counter()
    run_steps(args.warmup, args.steps)
    norm_all, x_all = join_groups()
    winner, best_dev = dd.global_best_of_n_device(norm_all, x_all, counts)   # final best-of-N over all ranks' particles
    barrier()
    elapsed = time.perf_counter() - t0
    best = int(best_dev)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    x = x_t

    # ---- per-kernel durations (HIP events on the launch stream), outside the timed region: ONE chain of all N particles,
    # launches back to back with nothing beside them
    reps = 50                       # independent of --steps: a short timed region should not mean a noisy launch average
    for i in range(10):             # (with particle groups the one-chain buffers have not been touched yet)
        x = step(i, x)
    torch.cuda.synchronize()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps)]
    for i in range(reps):
        x = step(args.warmup + args.steps + i, x, evs[i])
    torch.cuda.synchronize()
    dur = {k: float(np.mean([e[j].elapsed_time(e[j + 1]) for e in evs])) * 1e-3
           for j, k in enumerate(("fwd", "bwd", "upd"))}
    # ... and the same launches as the timed loop runs them: group 0's (N / chains particles) on its own stream while the
    # other groups' launches run beside them (longer per launch, shorter per step)
    dur_grouped = None
    if nch > 1:
        for g in groups:
            g["stream"].wait_stream(torch.cuda.current_stream())
        gev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps)]
        run_steps(args.warmup + args.steps, reps, gev)
        join_groups()
        torch.cuda.synchronize()
        dur_grouped = {k: float(np.mean([e[j].elapsed_time(e[j + 1]) for e in gev])) * 1e3
                       for j, k in enumerate(("fwd", "bwd", "upd"))}

    # ---- on-box copy ceiling (report against the vendor peak AND a measured copy kernel):
    # a 1 GiB device-to-device copy, read + write bytes over its event time
    copy_gbs = None
    if rank == 0:
        src = torch.empty(256 << 20, dtype=torch.float32, device=device).normal_()
        dst = torch.empty_like(src)
        for _ in range(2):
            dst.copy_(src)
        ce = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ce[0].record()
        for _ in range(5):
            dst.copy_(src)
        ce[1].record()
        torch.cuda.synchronize()
        copy_gbs = 5 * 2 * src.numel() * 4 / (ce[0].elapsed_time(ce[1]) * 1e-3) / 1e9
        del src, dst

    if rank == 0:
        total = n * world
        value = total * args.steps / elapsed
        algo = {k: v * P_BYTES for k, v in ALGO_P[args.operator].items()}
        step_p = sum(ALGO_P[args.operator].values())
        dom = max(dur, key=dur.get)
        names = dict(KERNELS[args.operator], upd="x_{t-1} = sample - (a g_pre + g_unet) (k_step_update)")
        achieved = algo[dom] * n / dur[dom] / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.operator, {}).get(dom)
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "algorithmic_bytes_per_launch": algo[dom] * n,
                    "avg_launch_ms": dur[dom] * 1e3,
                    "all_launches_ms": {k: v * 1e3 for k, v in dur.items()},
                    "launch_timing": "one chain of all N particles, launches back to back, HIP events on the launch "
                                     "stream, outside the timed region",
                    "grouped_launches_us": dur_grouped,
                    "step_algorithmic_bytes_per_particle": step_p * P_BYTES,
                    "step_frac_of_hbm_roofline": (step_p * P_BYTES * n / (elapsed / args.steps)) / 1e9 / HBM_PEAK_GBS,
                    "copy_ceiling": copy_gbs, "frac_of_copy_ceiling": achieved / copy_gbs}
        if args.operator == "motion_blur":
            # SURVEY 8d exception: the tap-list kernels are priced against the fp32 vector peak as well as HBM;
            # the binding limit is the one with the larger fraction
            taps = int((op.get_kernel() != 0).sum())
            flops = 2.0 * taps * 3 * 256 * 256 * n                        # one launch: fwd (= the adjoint's count)
            valu = {k: flops / dur[k] / 1e12 for k in ("fwd", "bwd")}
            roofline["valu"] = {"bound": "valu", "nonzero_taps": taps, "flop_per_launch": flops,
                                "achieved_tflops": valu, "peak_tflops": VALU_F32_PEAK_TFLOPS,
                                "frac": {k: v / VALU_F32_PEAK_TFLOPS for k, v in valu.items()}}
            hb, vb = roofline["frac"], valu.get(dom, 0.0) / VALU_F32_PEAK_TFLOPS
            roofline["binding"] = "valu" if vb > hb else "hbm"
        line = {
            "metric": "particles×denoise-steps/sec @256×256 N=64; x0_hat rel-L2 vs ref",
            "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"FFHQ-shaped 256x256 {WORKLOADS[args.operator]}, 'ps' scale 0.3, "
                                   f"best-of-N N={n}/GPU, DDPM t cycling 999->0",
                       "operator": args.operator, "particles_per_gpu": n, "global_particles": total,
                       "image": "3x256x256",
                       "chains_per_gpu": nch,
                       "x0_hat_store": bool(args.x0_store) or args.operator == "inpainting",
                       "parallelism": f"particles sharded x{world}; per GPU {nch} independent particle group(s), one HIP "
                                      f"stream each; champion all-gather at the select"},
            "roofline": roofline,
            "best_of_n_index": best,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], line["x0_hat_rel_l2"] = cpu_baseline(args, op, fkw, smp, ring, x_t, y, handle, device)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(args, op, fkw, smp, ring, x_t, y, handle, device):
    """The oracle (plain-C port of the reference step, OpenMP) on the host cores, same inputs, bounded sample.
    Also yields the parity figure of the metric: rel-L2 of the HIP x0_hat / x_{t-1} against it."""
    import numpy as np
    # threads = the cores this process may actually run on (the GPU box gives a CPU share, not the host)
    # (a 1-GPU box: 16 of the host's cores -- the affinity mask does not show the share, so cap at 16)
    cores = int(os.environ.get("OMP_NUM_THREADS", 0)) or min(len(os.sched_getaffinity(0)), 16)
    os.environ["OMP_NUM_THREADS"] = str(cores)      # read by libgomp when the oracle library loads
    import oracle
    from dps_ttc_amd import kernels
    nc, steps = min(args.cpu_particles, x_t.shape[0]), args.cpu_steps
    if args.operator == "gaussian_blur":
        orc = oracle.make_operator("gaussian_blur", kernel_size=61, intensity=3.0)
    elif args.operator == "motion_blur":
        orc = oracle.make_operator("motion_blur", kernel=op.get_kernel().reshape(61, 61).cpu().numpy())
    elif args.operator == "super_resolution":
        orc = oracle.make_operator("super_resolution", in_shape=(1, 3, 256, 256), scale_factor=4)
    elif args.operator == "inpainting":
        orc = oracle.make_operator("inpainting", mask=fkw["mask"].cpu().numpy())
    else:
        orc = oracle.make_operator("phase_retrieval", oversample=2.0)
        nc = min(nc, 8)                              # the oracle's phase retrieval is a plain DFT
    sched = oracle.tables.schedule(1000)
    x = x_t[:nc].cpu().numpy()
    yh = y.cpu().numpy()
    sets = [{k: v[:nc].cpu().numpy() for k, v in s.items()} for s in ring]
    orc.forward(x[:1])                       # page in the library outside the timed region
    t0 = time.perf_counter()
    outs = []
    for i in range(steps):
        t = 999 - i
        s = sets[i % len(sets)]
        c = oracle.tables.step_coefs(sched, t)
        r = oracle.dps_step(orc, x, s["model_out"], s["noise"], yh, c, scale=0.3, power=1,
                            g_unet_fn=lambda g, gu=s["g_unet"]: gu)
        outs.append(r)
        x = r["x_next"]
    dt = time.perf_counter() - t0
    # parity of the same steps on the GPU
    buf = kernels.StepBuffers(handle, nc, 3, 256, 256, device)
    xg = x_t[:nc].contiguous()
    worst = 0.0
    for i in range(steps):
        ck = smp.step_coefs[999 - i]
        s = ring[i % len(ring)]
        kernels.step_fwd(handle, buf, xg, s["model_out"][:nc].contiguous(), s["noise"][:nc].contiguous(), y, ck)
        kernels.step_bwd(handle, buf, y, 0.3, 1, ck)
        xg = kernels.step_update(buf, s["g_unet"][:nc].contiguous(), ck)
        for a, b in ((buf.x0_hat, outs[i]["x0_hat"]), (xg, outs[i]["x_next"])):
            a = a.cpu().numpy().astype(np.float64)
            worst = max(worst, float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel())))
    return ({"value": nc * steps / dt, "unit": "particle-steps/s", "cores": cores, "cpu": cpu_model(),
             "kind": "port",
             "sample": f"{nc} particles x {steps} steps of the same workload (oracle/dps_oracle.c, OpenMP, "
                       f"zero taps of a blur kernel skipped), {dt:.1f} s"}, worst)


if __name__ == "__main__":
    main()
