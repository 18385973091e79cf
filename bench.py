#!/usr/bin/env python3
"""Headline benchmark: particles x denoise-steps / s of the DPS hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts N rank processes itself, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Default workload (BASELINE.json configs[2], the configuration the metric is quoted on): FFHQ-shaped 256x256 RGB,
Gaussian deblur sigma=3.0 (61x61 kernel), N=64 particles per GPU, 'ps' conditioning (scale 0.3).
One step = the whole non-UNet DPS step over the particle batch -- S1 posterior arithmetic, A(x0_hat) + residual
+ norm, the cotangent back through A^T / clamp to the UNet output, and the update -- three fused HIP launches.
The UNet is not in the timed region: its output (model_out) and its VJP (g_unet) are synthetic device-resident
tensors, cycled through a small ring so no step re-reads warm lines; x_t chains from step to step
as in the real loop, t cycles 999 -> 0 with the real fp32 tables.

--workload   what one timed step is (every variant calls the package's own entry points):
    dps         the DPS step; the timed region ends with ONE global best-of-N select        (configs[2]; default)
    dps_scores  the DPS step + a per-step all-gather of the per-particle scores over all ranks and the global first-min
                argmin on the device (dps_ttc_amd.distributed.gather_scores)                 (configs[3]: RCCL score all-gather)
    search      the search_ddpm step (reference gaussian_diffusion.py:618-633): S1, scoring, per-step global select
                (distributed.GlobalSelect: champion all-gather), winner replicated            (per-step best-of-N)
    resample    the ttc_ddim step (DDIM S1) + global multinomial resampling of all ranks' particles every
                --resample-every steps (distributed.global_resample, reference :685-698)      (configs[4])
--scaling    weak: --particles is per GPU (default);  strong: --particles is the GLOBAL count, sharded over the ranks
             (configs[3]: --scaling strong --particles 256 --gpus 8 -> 32 per GPU)
--semantic   the active semantic-guidance term as a stand-in: a synthetic device-resident cotangent on x0_hat (what the
             embedder's VJP would return) scaled by the exact anneal scalar of condition_methods.py:155 enters K2 through
             dpsx_step_bwd_extra_f32 (the face networks themselves are not available offline)

Multi-GPU: particles shard across ranks, no collective inside the DPS step itself; what is exchanged is named above.

Rank 0 prints ONE JSON line; `roofline` is for the dominant kernel from live HIP-event timing, `cpu_baseline`
is the oracle (a CPU port of the reference path) on a bounded sample of the same workload.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

P_BYTES = 3 * 256 * 256 * 4                    # one fp32 particle image
HBM_PEAK_GBS = 8000.0                          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
VALU_F32_PEAK_TFLOPS = 157.3                   # MI355X_MICROARCH.md: fp32 vector peak (256 CUs x 128 FMA lanes x 2.4 GHz)

OPERATORS = ("gaussian_blur", "motion_blur", "super_resolution", "inpainting", "phase_retrieval")
# rho = |y| / |x| of each operator
RHO = {"gaussian_blur": 1.0, "motion_blur": 1.0, "super_resolution": 1.0 / 16.0, "inpainting": 0.0, "phase_retrieval": 2.25}


def algo_p(operator, x0_store=True, workload="dps", semantic=False):
    """Bytes per particle of each launch, in units of P -- the ONE table bench.py and tools/kbench.py price with.

    "survey":   SURVEY.md 8d's compulsory traffic of the reference algorithm at the boundaries the UNet forces:
                fwd = S1 6P + S2 (1 + rho)P, bwd = S3 (4 + rho)P, upd = S4 4P; inpainting recomputes r (rho -> 0);
                phase retrieval stores the complex cotangent instead of r (S2 = 5.5P, S3 = 8.5P).  This build's spectral
                step forms the cotangent and runs the inverse column transforms inside the FORWARD launches (pass B:
                the spectrum never goes back to HBM), so the 4.5P of S3 that are the cotangent's read are priced on the
                launch that does that work: fwd 16P, bwd 4P -- the step's 24P are unchanged.
    "algorithmic": the same table for the configuration actually launched -- without the x0_hat store (the `ps` step
                reads the image nowhere after K1) the forward half has no x0_hat write and the backward half no x0_hat
                read (it works from the clamp gate): fwd - 1P, bwd - 1P.  `roofline.achieved` is priced on this.
    The bytes the launches are DESIGNED to move are smaller still (the zero variance half of g_model_out and grad_x_direct
    are never written: DESIGN.md section 2); what they really move is the PMC figure (`traffic`, profiles/traffic.json)."""
    rho = RHO[operator]
    if operator == "phase_retrieval":
        survey = {"fwd": 11.5 + 4.5, "bwd": 8.5 - 4.5, "upd": 4.0}
    else:
        survey = {"fwd": 7.0 + rho, "bwd": 4.0 + rho, "upd": 4.0}
    algo = dict(survey)
    droppable = operator != "inpainting"          # inpainting's backward half reads x0_hat back (r is recomputed)
    if not x0_store and droppable:
        algo["fwd"] -= 1.0
        algo["bwd"] -= 1.0
    if semantic:                                   # one more image-sized read in K2 (the extra cotangent)
        algo["bwd"] += 1.0
        survey["bwd"] += 1.0
    if workload in ("search", "search_single"):
        # reference :618-633 -- S1 6P (5P without the x0_hat store) + score 1P + winner gather 1P (+1P written)
        survey = {"s1": 6.0, "score": 1.0, "select": 1.0}
        algo = {"s1": 5.0, "score": 1.0, "select": 1.0}
        if workload == "search_single":
            # one state particle: S1 reads the noise and writes the proposal (the state and its model output are read once
            # for all N: -> 0 per particle), the proposal is scored, the winner is copied once
            algo = {"s1": 2.0, "score": 1.0, "select": 0.0}
    return {"survey": survey, "algorithmic": algo}


KERNELS = {
    "gaussian_blur": {"fwd": "S1 + A(x0_hat) + residual + norm partials (k_blur_sep_fwd<3,POST,RESID>)",
                      "bwd": "A^T (symmetric taps: D C E, no fold terms) + clamp gate + -b*coef (k_blur_sep_adj_sym<3,EPI>)"},
    "motion_blur": {"fwd": "S1 + tap-list A(x0_hat) + residual + norm partials (k_blur_taps<POST,RESID>)",
                    "bwd": "tap-list A^T on the image domain: plain + mirrored windows in one scan, gate, -b*coef (k_blur_taps_adj<EPI>)"},
    "super_resolution": {"fwd": "S1 + resize W,H passes + residual + norm partials (k_resize_rows_fwd<POST,RESID>)",
                         "bwd": "resize adjoint (inverse tables) + clamp gate + -b*coef (k_resize_adj<EPI>)"},
    "inpainting": {"fwd": "S1 + mask residual + norm partials (k_mask_step_fwd)",
                   "bwd": "mask A^T + clamp gate + -b*coef (k_mask_step_bwd)"},
    "phase_retrieval": {"fwd": "S1 + row FFTs, column FFT + modulus residual + inverse columns (k_pr_rows_fwd + k_pr_cols)",
                        "bwd": "inverse row FFTs + crop + clamp gate + -b*coef (k_pr_rows_inv)"},
}
SEARCH_KERNELS = {"s1": "S1 without the x0_hat store (k_posterior_fwd)", "score": "scoring launch: A(sample), residual, per-tile sums",
                  "select": "costs + torch.argmin-order select + winner replication (k_finalize_select + k_gather)"}
WORKLOADS = {
    "gaussian_blur": "Gaussian deblur (sigma=3.0, k=61)",
    "motion_blur": "motion deblur (synthetic 61x61 path kernel, intensity 0.5)",
    "super_resolution": "x4 super-resolution (bicubic, 256->64)",
    "inpainting": "inpainting (Bernoulli(0.5) mask)",
    "phase_retrieval": "phase retrieval (oversample 2.0, 384x384 spectrum)",
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--particles", type=int, default=64, help="particles per GPU (--scaling weak) or in total (--scaling strong)")
    ap.add_argument("--operator", default="gaussian_blur", choices=sorted(OPERATORS))
    ap.add_argument("--workload", default="dps", choices=["dps", "dps_scores", "search", "resample"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--semantic", action="store_true", help="stand-in semantic-guidance cotangent through dpsx_step_bwd_extra_f32")
    ap.add_argument("--resample-every", type=int, default=10)
    ap.add_argument("--resample-fetch", choices=["auto", "all", "selected"], default="auto",
                    help="resample workload: how the drawn particles travel (distributed.resample_particles)")
    ap.add_argument("--search-form", default="single", choices=["single", "replicated"],
                    help="--workload search: 'single' = the loop's default (SearchDDPM.single_state: after a select all particles "
                         "are copies of the winner, so ONE state particle feeds the N proposals -- dpsx_search_step_one_f32); "
                         "'replicated' = the reference's N copies (dpsx_search_step_f32)")
    ap.add_argument("--x0-store", action="store_true",
                    help="K1 also writes the x0_hat image out.  The `ps` loop reads it nowhere after K1 (the backward half "
                         "works from the clamp gate), so p_sample_loop -- and this bench -- ask for it only when something "
                         "consumes it (a progress snapshot, the semantic term); inpainting always writes it (its backward half reads it)")
    ap.add_argument("--chains", type=int, default=0,
                    help="independent particle groups per GPU, each on its own HIP stream (kernels.ParticleGroups = "
                         "sampler.particle_groups; 1 = one chain of N; default: 3, 2 for phase retrieval, whose launches hold "
                         "52 KB of LDS per workgroup; workloads with a per-step exchange run one chain)")
    ap.add_argument("--force-process-group", action="store_true",
                    help="initialise the torch.distributed process group even for ONE rank: every exchange of the workload "
                         "then runs through the backend (RCCL) as it would with eight -- how a one-GPU box exercises it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-particles", type=int, default=64)
    ap.add_argument("--cpu-steps", type=int, default=5)
    return ap.parse_args(argv)


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
def synth_inputs(n, ring, device, seed, semantic=False):
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
    if semantic:        # what the embedder's VJP would hand back: a unit-scale cotangent on x0_hat (scaled per step below)
        for s in sets:
            s["g_sem"] = (torch.randn(shape, generator=g) * 1e-3).to(device)
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


def semantic_scale(sem_guid_scale, anneal_factor, t):
    """condition_methods.py:155 (the anneal scalar of the semantic term), exact"""
    return sem_guid_scale * (1 + (anneal_factor - 1) / (1 + math.exp(-10 * (0.3 - t))))


def load_traffic(operator):
    """PMC HBM bytes per launch at N = 64 (profiles/traffic.json: tools/pmc_traffic.sh), or {}"""
    try:
        return dict(json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(operator, {}))
    except Exception:
        return {}


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # ONE JSON line on stdout, nothing else: libraries write there too (RCCL prints a version banner to stdout when its
    # first communicator comes up), so file descriptor 1 points at stderr for the whole run and is put back for the line
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size must match")

    # Both are read when the HIP / HSA runtime initialises, i.e. at the first GPU call below -- set them before anything
    # touches the device.  One hardware queue per HIP stream: with the runtime's default of 4 the particle groups'
    # streams, the null stream and RCCL's can end up sharing queues (seen once as a 2.4x slower step with three groups).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this host driver (RCCL)
    import numpy as np
    import torch
    import torch.distributed as dist
    shared_gpu = world > max(torch.cuda.device_count(), 1)
    local = local % max(torch.cuda.device_count(), 1)      # rehearsals may put several ranks on one GPU
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    use_pg = world > 1 or args.force_process_group
    if use_pg:
        if world == 1:          # no launcher: the rendezvous of a one-rank group
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
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

    # ---- particles of this rank
    if args.scaling == "strong":
        counts = dd.shard_counts(args.particles, world)           # contiguous blocks, rank-major (distributed.shard_range)
        if min(counts) < 1:
            raise SystemExit(f"--scaling strong: {args.particles} particles do not cover {world} ranks")
    else:
        counts = [args.particles] * world
    n, total = counts[rank], sum(counts)
    if args.workload == "resample" and len(set(counts)) != 1:
        raise SystemExit("--workload resample needs equal shards (distributed.resample_particles)")

    wl = args.workload
    semantic = bool(args.semantic) and wl != "search"
    op, fkw = build_operator(args.operator, device)
    cm = get_conditioning_method("ps", op, get_noise("gaussian", sigma=0.05), scale=0.3)
    smp = create_sampler(sampler={"search": "search_ddpm", "resample": "ttc_ddim"}.get(wl, "ddpm"), steps=1000,
                         noise_schedule="linear", model_mean_type="epsilon", model_var_type="learned_range",
                         dynamic_threshold=False, clip_denoised=True, rescale_timesteps=True, timestep_respacing="")
    x_t, ring, truth, meas_noise = synth_inputs(n, 3, device, 1234 + rank, semantic=semantic)
    yy = op.forward(truth.to(device), **fkw).detach()
    y = (yy + meas_noise.to(device)[..., :yy.shape[-2], :yy.shape[-1]]).contiguous()
    mask = fkw.get("mask")
    handle = op.hip_handle_for(mask) if args.operator == "inpainting" else op.hip_handle(x_t)
    buf = kernels.StepBuffers(handle, n, 3, 256, 256, device)
    spec = cm.fused_spec()
    want_x0 = bool(args.x0_store) or semantic          # the embedder reads x0_hat (p_sample_loop asks for it then, too)
    if wl == "search":
        smp.global_select = dd.GlobalSelect() if use_pg else None
    res_gen = torch.Generator(device=device).manual_seed(0)     # same stream on every rank; device draw: no host read

    def coefs_at(i):
        return smp.sample_coefs(999 - (i % 1000))       # DDPM record; ttc_ddim: the DDIM record

    def sem_cotangent(i, s):
        """the stand-in semantic cotangent of step i: g_sem scaled by the exact anneal scalar (a [1]-sized host float)"""
        if not semantic:
            return None
        return s["g_sem"], semantic_scale(0.3, 10.0, (999 - (i % 1000)) / 1000.0)

    # ---- one chain of all N particles on torch's current stream (the product loop's default schedule)
    sem_buf = torch.empty_like(x_t) if semantic else None

    def step(i, x, timers=None):
        ck = coefs_at(i)
        s = ring[i % len(ring)]
        g_sem = None
        if semantic:        # [the embedder's forward + VJP on buf.x0_hat would run between K1 and K2: torch, not this path's work]
            g, sc = sem_cotangent(i, s)
            g_sem = torch.mul(g, sc, out=sem_buf)
        if timers is not None:
            timers[0].record()
        kernels.step_fwd(handle, buf, x, s["model_out"], s["noise"], y, ck, want_x0=want_x0)
        if timers is not None:
            timers[1].record()
        kernels.step_bwd(handle, buf, y, spec["scale"], spec["power"], ck, g_x0_extra=g_sem)
        if timers is not None:
            timers[2].record()
        out = kernels.step_update(buf, s["g_unet"], ck)
        if timers is not None:
            timers[3].record()
        return out

    def barrier():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- particle groups on streams: kernels.ParticleGroups, the object sampler.particle_groups / the driver's
    # --particle_groups run the fused loop with.  Per-particle results do not depend on the grouping
    # (tests/test_driver_gpu.py checks it bit for bit).  Workloads with an exchange every step run one chain.
    # (N = 16 is 768 tiles on 256 CUs: one short generation of workgroups per launch, nothing for a second group to fill)
    # measured on MI355X (tools/ab_chains.sh, profiles/r03_ab_chains.txt; us per step, chains 1 / 2 / 3 / 4): N = 64 Gaussian
    # 127 / - / 117 / 133, motion 172 / - / 161 / 190; N = 32 Gaussian 74.9 / 65.7 / 66.6 / 86.9, motion 103.5 / 90.4 / 90.9 /
    # 85.0, SR x4 70.6 / 62.1 / - / 85.9; N = 16 no gain (43 us either way).  Every group costs the host three launches of
    # ~6.5 us per step: with k groups a step cannot be shorter than ~20 k us, which is the cliff at 4 groups and above
    default_chains = 1 if wl != "dps" or semantic or n < 24 else (2 if args.operator == "phase_retrieval" or n < 48 else 3)
    nch = max(1, min(args.chains if args.chains > 0 else default_chains, n))
    if shared_gpu or wl in ("search", "resample", "dps_scores"):
        nch = 1         # rank processes time-slicing ONE GPU (gloo rehearsal): several queues per process make it crawl
    pg = None
    if nch > 1:
        pg = kernels.ParticleGroups(op, n, 3, 256, 256, device, nch, mask=mask, like=x_t, record_streams=False)
    single = wl == "search" and args.search_form == "single"
    if wl == "dps_scores" and len(set(counts)) != 1:
        raise SystemExit("--workload dps_scores needs equal shards (distributed.ScoreGather)")
    state = {"scores": dd.ScoreGather(), "x": x_t[:1].contiguous() if single else x_t, "gx": [x_t[sl] for sl in pg.slices] if pg else None}

    def grouped_steps(first, count, timers=None):
        for i in range(count):
            ck = coefs_at(first + i)
            s = ring[(first + i) % len(ring)]
            for j in range(len(pg)):
                tm = timers[i] if timers is not None and j == 0 else None
                st = pg.streams[j]
                if tm is not None:
                    tm[0].record(st)
                pg.step_fwd(j, state["gx"][j], s["model_out"], s["noise"], y, ck, want_x0=want_x0)
                if tm is not None:
                    tm[1].record(st)
                pg.step_bwd(j, y, spec["scale"], spec["power"], ck)
                if tm is not None:
                    tm[2].record(st)
                state["gx"][j] = pg.step_update(j, s["g_unet"], ck)
                if tm is not None:
                    tm[3].record(st)

    def chain_steps(first, count):
        x = state["x"]
        for i in range(count):
            k = first + i
            if wl == "search":
                # SearchDDPM.search_step without the model call: dpsx_search_step_f32 (+ GlobalSelect across ranks)
                s = ring[k % len(ring)]
                local_only = smp.global_select is None
                if single:      # SearchDDPM.search_step_one: the state is ONE particle, its model output [1, 2C, H, W]
                    winner, sample, costs, best, _ = handle.search_step_one(x, s["model_out"][:1], s["noise"], y,
                                                                            coefs_at(k), want_winner=local_only)
                    x = winner if local_only else smp.global_select(costs, sample, n_out=1)
                    continue
                x_next, sample, costs, best, _ = handle.search_step(x, s["model_out"], s["noise"], y, coefs_at(k),
                                                                    replicate=local_only)
                x = x_next if local_only else smp.global_select(costs, sample)
                continue
            x = step(k, x)
            if wl == "dps_scores":
                # per-step score all-gather (RCCL), pipelined: the gather of step k runs on the backend's stream under
                # step k + 1 (distributed.ScoreGather); the global first-min argmin of the finished one, on the device
                done = state["scores"].submit(buf.norm)
                if done is not None:
                    state["best"] = dd.first_argmin(done)
            elif wl == "resample" and k % args.resample_every == 0:
                # TTC_DDIM._resample over the sharded particle set: scores all-gathered, identical host draw, states fetched
                x, _, _ = dd.global_resample(x, buf.norm, 100.0, res_gen, fetch=args.resample_fetch)
        state["x"] = x

    def run_steps(first, count):
        if pg is not None:
            grouped_steps(first, count)
        else:
            chain_steps(first, count)

    def closing_select():
        """final best-of-N over all ranks' particles (best_of_n_simple.py:32-40 on the device, no host read)"""
        if wl == "dps_scores":
            last = state["scores"].flush()
            if last is not None:
                state["best"] = dd.first_argmin(last)
        if pg is not None:
            pg.join()
            return dd.global_best_of_n_device(pg.full.norm, pg.x_next(), counts)
        if wl == "search":          # every particle is already the global winner
            return state["x"][:1], torch.zeros((), dtype=torch.int64, device=device)
        return dd.global_best_of_n_device(buf.norm, state["x"], counts)

    def timed(first, count):
        barrier()
        if pg is not None:
            pg.fork()
        t0 = time.perf_counter()
        run_steps(first, count)
        winner, best_dev = closing_select()
        barrier()
        dt = time.perf_counter() - t0
        if use_pg:
            tmax = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, int(best_dev)

    if pg is not None:
        pg.fork()
    run_steps(0, args.warmup)
    closing_select()                                   # warm the select (and the collectives) too
    elapsed, best = timed(args.warmup, args.steps)

    # ---- the same steps as ONE chain of all N particles (the schedule of sampler.particle_groups = 1), timed the same
    # way, so that the line carries both figures
    one_chain_s = None
    if pg is not None and wl == "dps":
        saved, pg_saved = dict(state), pg
        pg = None
        state["x"] = x_t
        oc_steps = max(args.steps, 100)        # (an auxiliary figure: a short --steps run should not make it a noisy one)
        run_steps(0, max(args.warmup, 20))
        one_chain_s, _ = timed(args.warmup, oc_steps)
        one_chain_s *= args.steps / oc_steps   # normalised to --steps: everything below divides by args.steps
        pg = pg_saved
        state.update(saved)
    elif pg is None:
        one_chain_s = elapsed

    # ---- per-kernel durations (HIP events on the launch stream), outside the timed region: ONE chain of all N particles,
    # launches back to back with nothing beside them
    reps = 50                       # independent of --steps: a short timed region should not mean a noisy launch average
    x = x_t
    if wl == "search":
        s0 = ring[0]
        ck0 = coefs_at(0)
        _, sample0 = kernels.posterior_fwd(x, s0["model_out"], s0["noise"], ck0, want_x0=False)
        _, best0, _ = handle.score_argmin(sample0, y)
        x1 = x_t[:1].contiguous()
        parts = {"s1": lambda: kernels.posterior_fwd(x, s0["model_out"], s0["noise"], ck0, want_x0=False),
                 "score": lambda: handle.score(sample0, y),
                 "select": lambda: kernels.replicate(sample0, best0)}
        if single:      # S1 from one state has no entry point of its own: the step without the winner's copy, minus its scoring half
            parts["s1"] = lambda: handle.search_step_one(x1, s0["model_out"][:1], s0["noise"], y, ck0, want_winner=False)
            parts["score"] = lambda: handle.score_argmin(sample0, y)
            parts["select"] = lambda: kernels.replicate(sample0, best0, n_out=1)
        dur = {}
        for name, fn in parts.items():
            for _ in range(5):
                fn()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for a_, b_ in ev:
                a_.record()
                fn()
                b_.record()
            torch.cuda.synchronize()
            dur[name] = float(np.mean([a_.elapsed_time(b_) for a_, b_ in ev])) * 1e-3
        if single:
            dur["s1"] = max(dur["s1"] - dur["score"], 1e-9)
    else:
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
    if pg is not None:
        pg.fork()
        gev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps)]
        grouped_steps(args.warmup + args.steps, reps, gev)
        pg.join()
        torch.cuda.synchronize()
        dur_grouped = {k: float(np.mean([e[j].elapsed_time(e[j + 1]) for e in gev])) * 1e3
                       for j, k in enumerate(("fwd", "bwd", "upd"))}

    # ---- on-box copy ceiling (report against the vendor peak AND a measured copy kernel): 1 GiB moved device to
    # device, read + write bytes over the event time, by torch's copy_ and by the library's own particle copy
    # (dpsx_gather_f32 with identity ids: float4 per lane, one particle per block row) -- the ceiling is the faster of the two
    copy_gbs, copy_detail = None, None
    if rank == 0:
        npart = (1 << 30) // P_BYTES
        src = torch.empty((npart, 3, 256, 256), dtype=torch.float32, device=device).normal_()
        dst = torch.empty_like(src)
        ids = torch.arange(npart, device=device)

        def rate(fn):
            for _ in range(2):
                fn()
            ce = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ce[0].record()
            for _ in range(5):
                fn()
            ce[1].record()
            torch.cuda.synchronize()
            return 5 * 2 * src.numel() * 4 / (ce[0].elapsed_time(ce[1]) * 1e-3) / 1e9

        chw = src[0].numel()
        copy_detail = {"torch_copy_": rate(lambda: dst.copy_(src)),
                       "dpsx_gather_f32_identity": rate(lambda: kernels.check(kernels.lib().dpsx_gather_f32(
                           kernels.ptr(src), kernels.ptr(ids), kernels.ptr(dst), npart, npart, chw,
                           kernels.stream_of(src)), "dpsx_gather_f32"))}
        copy_gbs = max(copy_detail.values())
        del src, dst

    if rank == 0:
        x0_stored = want_x0 or args.operator == "inpainting"
        table = algo_p(args.operator, x0_store=x0_stored, workload="search_single" if single else wl, semantic=semantic)
        algo = {k: v * P_BYTES for k, v in table["algorithmic"].items()}
        survey_step_p = sum(table["survey"].values())
        algo_step_p = sum(table["algorithmic"].values())
        value = total * args.steps / elapsed
        step_s = elapsed / args.steps
        dom = max(dur, key=dur.get)
        names = dict(SEARCH_KERNELS) if wl == "search" else \
            dict(KERNELS[args.operator], upd="x_{t-1} = sample - (a g_pre + g_unet) (k_step_update)")
        achieved = algo[dom] * n / dur[dom] / 1e9
        # PMC traffic (HBM bytes per launch at N = 64, collected with the x0_hat store off): scaled to this run's N;
        # valid for the plain `ps` step only
        pmc_ok = wl != "search" and not semantic and (not x0_stored or args.operator == "inpainting")
        traffic_tbl = load_traffic(args.operator) if pmc_ok else {}
        moved = {k: (traffic_tbl[k] * n / 64.0 if k in traffic_tbl else None) for k in dur}
        roofline = {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": moved.get(dom),
                    "algorithmic_bytes_per_launch": algo[dom] * n,
                    "bytes_priced": "SURVEY 8d compulsory traffic of the configuration launched (x0_hat store "
                                    + ("on" if x0_stored else "off: fwd and bwd 1P less than the survey table") + ")",
                    "avg_launch_ms": dur[dom] * 1e3,
                    "all_launches_ms": {k: v * 1e3 for k, v in dur.items()},
                    "launch_timing": "one chain of all N particles, launches back to back, HIP events on the launch "
                                     "stream, outside the timed region",
                    "per_launch": {k: {"algorithmic_bytes": algo[k] * n, "moved_bytes": moved[k],
                                       "algorithmic_GBps": algo[k] * n / dur[k] / 1e9,
                                       "moved_GBps": (moved[k] / dur[k] / 1e9) if moved[k] else None,
                                       "frac_moved": (moved[k] / dur[k] / 1e9 / HBM_PEAK_GBS) if moved[k] else None}
                                   for k in dur},
                    "grouped_launches_us": dur_grouped,
                    # the whole step: SURVEY 8d's figure (17P for blur -- the reference algorithm's compulsory traffic),
                    # the figure of the configuration launched, and the bytes the launches really moved (PMC)
                    "step_survey_bytes_per_particle": survey_step_p * P_BYTES,
                    "step_algorithmic_bytes_per_particle": algo_step_p * P_BYTES,
                    "step_frac_of_hbm_roofline": (survey_step_p * P_BYTES * n / step_s) / 1e9 / HBM_PEAK_GBS,
                    "step_frac_algorithmic": (algo_step_p * P_BYTES * n / step_s) / 1e9 / HBM_PEAK_GBS,
                    "copy_ceiling": copy_gbs, "copy_ceiling_detail": copy_detail}
        if one_chain_s is not None:
            oc = one_chain_s / args.steps
            roofline["one_chain_ms_per_step"] = oc * 1e3
            roofline["one_chain_value"] = total * args.steps / one_chain_s
            roofline["one_chain_step_frac_of_hbm_roofline"] = (survey_step_p * P_BYTES * n / oc) / 1e9 / HBM_PEAK_GBS
        if all(moved.get(k) for k in dur):
            mv = sum(moved.values())
            roofline["step_moved_bytes_per_particle"] = mv / n
            roofline["step_moved_GBps"] = mv / step_s / 1e9
            roofline["step_frac_moved"] = mv / step_s / 1e9 / HBM_PEAK_GBS
            roofline["frac_of_copy_ceiling"] = mv / step_s / 1e9 / copy_gbs
            if one_chain_s is not None:
                roofline["one_chain_step_frac_moved"] = mv / (one_chain_s / args.steps) / 1e9 / HBM_PEAK_GBS
            if roofline["frac_of_copy_ceiling"] > 1.0:
                # concurrent streams can beat a single 1 GiB copy_ (5.0-5.3 TB/s here against the guide's 6.3 TB/s
                # float4 copy), so a little above 1 is possible; well above it the byte model is wrong
                sys.stderr.write(f"bench.py: WARNING moved bytes / step time = {roofline['frac_of_copy_ceiling']:.2f} x the "
                                 "on-box copy ceiling\n")
        if args.operator == "motion_blur" and wl != "search":
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
        what = {"dps": "'ps' scale 0.3, one global best-of-N select at the end",
                "dps_scores": "'ps' scale 0.3, per-step all-gather of the particle scores + global argmin",
                "search": "search_ddpm step: S1, scoring, per-step global select, " +
                          ("ONE state particle kept (SearchDDPM.single_state)" if single else "winner replicated to all N"),
                "resample": f"ttc_ddim step (DDIM S1) with 'ps' scale 0.3, global multinomial resampling every "
                            f"{args.resample_every} steps"}[wl]
        fetch = args.resample_fetch if args.resample_fetch != "auto" else ("selected" if world > 2 else "all")
        exchange = {"dps": "ONE champion all-gather at the closing select (1 particle + its score and index per rank)",
                    "dps_scores": "RCCL all-gather of the [N/G] scores every step",
                    "search": "per step: ONE all-gather of one champion particle + its score per rank, device-side pick",
                    "resample": f"every {args.resample_every} steps: all-gather of scores, identical device multinomial "
                                f"draw on every rank, " + ("all-to-all of the drawn particles (each once per destination)"
                                                           if fetch == "selected" else "all-gather of states") +
                                " + HIP gather"}[wl]
        line = {
            "metric": "particles×denoise-steps/sec @256×256 N=64; x0_hat rel-L2 vs ref",
            "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": step_s * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"FFHQ-shaped 256x256 {WORKLOADS[args.operator]}, {what}, "
                                   f"N={total} particles ({'/'.join(str(c) for c in sorted(set(counts)))} per GPU), "
                                   f"t cycling 999->0" + (", stand-in semantic-guidance cotangent (anneal 10x)" if semantic else ""),
                       "operator": args.operator, "step": wl, "particles_per_gpu": n, "global_particles": total,
                       "image": "3x256x256",
                       "chains_per_gpu": nch,
                       "x0_hat_store": x0_stored,
                       "semantic_stand_in": semantic,
                       "process_group": (os.environ.get("DPSX_BENCH_BACKEND", "nccl") + f" x{world}") if use_pg else None,
                       "parallelism": f"particles sharded x{world} ({args.scaling} scaling); per GPU {nch} independent "
                                      f"particle group(s) (kernels.ParticleGroups), one HIP stream each; {exchange}"},
            "roofline": roofline,
            "best_of_n_index": best,
        }
        if world == 1 and not args.no_cpu_baseline and wl == "dps" and not semantic:
            line["cpu_baseline"], line["x0_hat_rel_l2"] = cpu_baseline(args, op, fkw, smp, ring, x_t, y, handle, device)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if use_pg:
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
    base = {"value": nc * steps / dt, "unit": "particle-steps/s", "cores": cores, "cpu": cpu_model(),
            "kind": "port",
            "sample": f"{nc} particles x {steps} steps of the same workload (oracle/dps_oracle.c, OpenMP, "
                      f"zero taps of a blur kernel skipped), {dt:.1f} s"}
    # SURVEY 8d's second CPU baseline: the reference's own ATen ops on the host cores (oracle/torch_ref.py: S1 as tensor
    # arithmetic, ReflectionPad2d + depthwise conv2d / gather-sum resize / mask / fft2, linalg.norm, autograd.grad), same
    # inputs, a smaller sample (a 61 x 61 conv2d forward + backward over 16 particles takes a few hundred ms)
    try:
        import torch
        from oracle import torch_ref
        torch.set_num_threads(cores)
        tn, ts = min(nc, 16), min(steps, 3)
        if args.operator in ("gaussian_blur", "motion_blur"):
            top = torch_ref.TorchOperator(args.operator, kernel=orc.kw["kernel"])
        elif args.operator == "super_resolution":
            top = torch_ref.TorchOperator(args.operator, tables=orc.kw["tables"])
        elif args.operator == "inpainting":
            top = torch_ref.TorchOperator(args.operator, mask=orc.kw["mask"])
        else:
            top = torch_ref.TorchOperator(args.operator, pad=orc.kw["pad"])
        xt = x_t[:tn].cpu().numpy()
        torch_ref.dps_step(top, xt[:1], sets[0]["model_out"][:1], sets[0]["noise"][:1], yh, oracle.tables.step_coefs(sched, 999),
                           0.3, sets[0]["g_unet"][:1])          # page in
        t1 = time.perf_counter()
        werr = 0.0
        for i in range(ts):
            s = sets[i % len(sets)]
            r = torch_ref.dps_step(top, xt, s["model_out"][:tn], s["noise"][:tn], yh, oracle.tables.step_coefs(sched, 999 - i),
                                   0.3, s["g_unet"][:tn])
            ref = outs[i]["x_next"][:tn] if i < len(outs) else None
            if ref is not None:
                werr = max(werr, float(np.linalg.norm((r["x_next"].astype(np.float64) - ref).ravel()) / np.linalg.norm(ref.ravel())))
            xt = r["x_next"]
        dt2 = time.perf_counter() - t1
        base["torch_ops"] = {"value": tn * ts / dt2, "unit": "particle-steps/s", "cores": cores, "kind": "torch CPU ops",
                             "sample": f"{tn} particles x {ts} steps, torch {torch.__version__} CPU ops in the reference's order "
                                       f"(oracle/torch_ref.py), {dt2:.1f} s", "x_next_rel_l2_vs_port": werr}
    except Exception as e:          # a reported extra, never a reason to lose the line
        base["torch_ops"] = {"error": repr(e)[:200]}
    return base, worst


if __name__ == "__main__":
    main()
