#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (CPU) in this container.

Usage (container only -- /root/reference does not exist on the GPU box):
    python tests/golden/make_golden.py

The reference's hot-path modules import three third-party packages that are
not installed and carry none of the hot-path arithmetic:
`motionblur` (kernel generator), `facenet_pytorch` (semantic net) and
`torchvision` (used only as `from torchvision import torch`).  They are
replaced by in-memory stubs below; nothing is copied from the reference, the
fixtures hold inputs and the reference's outputs only.
"""
import contextlib
import io
import os
import re
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from standin import StandInModel, synthetic_motion_kernel  # noqa: E402

REF = "/root/reference"
sys.dont_write_bytecode = True


class _StubKernel:
    """Stands in for motionblur.motionblur.Kernel: the kernel is a hot-path INPUT."""
    next_seed = 0

    def __init__(self, size=(61, 61), intensity=0.5):
        self.kernelMatrix = synthetic_motion_kernel(size[0], seed=_StubKernel.next_seed).astype(np.float64)


def install_stubs():
    mb, mbb = types.ModuleType("motionblur"), types.ModuleType("motionblur.motionblur")
    mbb.Kernel = _StubKernel
    mb.motionblur = mbb
    sys.modules["motionblur"], sys.modules["motionblur.motionblur"] = mb, mbb
    fn = types.ModuleType("facenet_pytorch")
    fn.MTCNN = fn.InceptionResnetV1 = object
    sys.modules["facenet_pytorch"] = fn
    tv = types.ModuleType("torchvision")
    tv.torch = torch
    sys.modules["torchvision"] = tv
    sys.path.insert(0, REF)


@contextlib.contextmanager
def quiet():
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
        yield buf


def np32(t):
    return t.detach().cpu().numpy().astype(np.float32)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print("wrote", name, "%.1f KB" % (os.path.getsize(path) / 1024))


DIFF = dict(steps=1000, noise_schedule="linear", model_mean_type="epsilon",
            model_var_type="learned_range", dynamic_threshold=False, clip_denoised=True,
            rescale_timesteps=True)

SCHED_KEYS = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
              "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
              "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1", "posterior_mean_coef2",
              "posterior_variance", "posterior_log_variance_clipped"]


def main():
    install_stubs()
    with quiet():
        from guided_diffusion import condition_methods as CM
        from guided_diffusion import gaussian_diffusion as GD
        from guided_diffusion import measurements as MS

    torch.set_num_threads(8)

    # ------------------------------------------------------------- 1. tables
    out = {}
    for tag, resp in (("full", ""), ("r20", "20"), ("r100", "100"), ("ddim50", "ddim50")):
        with quiet():
            s = GD.create_sampler(sampler="ddpm", timestep_respacing=resp, **DIFF)
        for k in SCHED_KEYS:
            out[f"{tag}.{k}"] = np.asarray(getattr(s, k), dtype=np.float64)
        out[f"{tag}.timestep_map"] = np.asarray(s.timestep_map, dtype=np.int64)
        out[f"{tag}.log_betas"] = np.log(s.var_processor.betas)
    g = MS.get_operator("gaussian_blur", kernel_size=61, intensity=3.0, device="cpu")
    out["gauss61_s3.kernel_f64"] = g.get_kernel().numpy()[0, 0].astype(np.float64)
    out["gauss61_s3.weight_f32"] = np32(g.conv.seq[1].weight)[:, 0]
    for f, hw in ((4, 256), (8, 256), (4, 64)):
        sr = MS.get_operator("super_resolution", in_shape=(1, 3, hw, hw), scale_factor=f, device="cpu")
        rz = sr.down_sample
        out[f"sr{f}_{hw}.sorted_dims"] = np.asarray(rz.sorted_dims, dtype=np.int64)
        for j, d in enumerate(rz.sorted_dims):
            out[f"sr{f}_{hw}.w_dim{d}"] = np32(rz.weights[j]).reshape(rz.weights[j].shape[0], -1)
            out[f"sr{f}_{hw}.i_dim{d}"] = rz.field_of_view[j].numpy().astype(np.int64)
    save("tables.npz", **out)

    # ------------------------------------------------------------- 2. posterior step
    with quiet():
        ddpm = GD.create_sampler(sampler="ddpm", timestep_respacing="", **DIFF)
    out = {}
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(2, 3, 8, 8, generator=gen)
    out["x"] = np32(x)
    out["w_x0"] = np32(torch.randn(2, 3, 8, 8, generator=gen))
    out["w_s"] = np32(torch.randn(2, 3, 8, 8, generator=gen))
    for t in (999, 500, 1, 0):
        a = float(ddpm.sqrt_recip_alphas_cumprod[t])
        b = float(ddpm.sqrt_recipm1_alphas_cumprod[t])
        # eps chosen so x0_hat straddles the clamp interval
        target = 1.4 * torch.tanh(torch.randn(2, 3, 8, 8, generator=gen))
        eps = (a * x - target) / b
        v = torch.rand(2, 3, 8, 8, generator=gen) * 2 - 1
        mo = torch.cat([eps, v], dim=1).requires_grad_()
        xx = x.clone().requires_grad_()
        torch.manual_seed(100 + t)
        noise = torch.randn_like(xx)
        torch.manual_seed(100 + t)
        with quiet():
            res = ddpm.p_sample(model=lambda z, ts: mo, x=xx, t=torch.tensor([t]))
        loss = (res["pred_xstart"] * torch.from_numpy(out["w_x0"])).sum() + \
               (res["sample"] * torch.from_numpy(out["w_s"])).sum()
        gx, gmo = torch.autograd.grad(loss, [xx, mo])
        out[f"t{t}.model_out"] = np32(mo)
        out[f"t{t}.noise"] = np32(noise)
        out[f"t{t}.x0_hat"] = np32(res["pred_xstart"])
        out[f"t{t}.sample"] = np32(res["sample"])
        out[f"t{t}.g_x"] = np32(gx)
        out[f"t{t}.g_model_out"] = np32(gmo)
    save("posterior.npz", **out)

    # ------------------------------------------------------------- 3. operators fwd / adjoint
    out = {}

    def op_case(tag, op, shape, seed, store_x=True, **fkw):
        gen = torch.Generator().manual_seed(seed)
        x = (torch.rand(*shape, generator=gen) * 2 - 1).requires_grad_()
        y = op.forward(x, **fkw)
        u = torch.randn(*y.shape, generator=gen)
        (gx,) = torch.autograd.grad((y * u).sum(), x)
        out[f"{tag}.seed"] = np.int64(seed)
        out[f"{tag}.shape"] = np.asarray(shape, dtype=np.int64)
        if store_x:
            out[f"{tag}.x"] = np32(x)
            out[f"{tag}.u"] = np32(u)
        out[f"{tag}.y"] = np32(y)
        out[f"{tag}.adj"] = np32(gx)

    gb = MS.get_operator("gaussian_blur", kernel_size=61, intensity=3.0, device="cpu")
    op_case("gauss.small", gb, (2, 3, 64, 64), 21)
    op_case("gauss.full", gb, (1, 3, 256, 256), 22, store_x=False)
    _StubKernel.next_seed = 3
    mbo = MS.get_operator("motion_blur", kernel_size=61, intensity=0.5, device="cpu")
    out["motion.kernel"] = np32(mbo.conv.seq[1].weight)[0, 0]
    op_case("motion.small", mbo, (2, 3, 64, 64), 23)
    sr4s = MS.get_operator("super_resolution", in_shape=(1, 3, 64, 64), scale_factor=4, device="cpu")
    op_case("sr4.small", sr4s, (2, 3, 64, 64), 25)
    sr4 = MS.get_operator("super_resolution", in_shape=(1, 3, 256, 256), scale_factor=4, device="cpu")
    op_case("sr4.full", sr4, (1, 1, 256, 256), 26, store_x=False)
    sr8 = MS.get_operator("super_resolution", in_shape=(1, 3, 256, 256), scale_factor=8, device="cpu")
    op_case("sr8.full", sr8, (1, 1, 256, 256), 27, store_x=False)
    inp = MS.get_operator("inpainting", device="cpu")
    mgen = torch.Generator().manual_seed(7)
    mask = (torch.rand(1, 1, 64, 64, generator=mgen) < 0.5).float()
    out["inpaint.mask"] = np32(mask)
    op_case("inpaint.small", inp, (2, 3, 64, 64), 28, mask=mask)
    pr = MS.get_operator("phase_retrieval", oversample=2.0, device="cpu")
    op_case("phase.small", pr, (2, 3, 32, 32), 29)
    op_case("phase.full", pr, (1, 1, 256, 256), 30, store_x=False)
    save("operators.npz", **out)

    # ------------------------------------------------------------- 4. conditioning per call
    out = {}
    model = StandInModel()
    noiser = MS.get_noise("gaussian", sigma=0.05)
    sr4c = MS.get_operator("super_resolution", in_shape=(1, 3, 32, 32), scale_factor=4, device="cpu")
    mask32 = mask[..., :32, :32].contiguous()
    out["inpaint.mask"] = np32(mask32)
    ops = {"gauss": (gb, {}), "sr4": (sr4c, {}), "inpaint": (inp, {"mask": mask32}), "motion": (mbo, {}),
           "phase": (pr, {})}
    for oname, (op, fkw) in ops.items():
        hw = 32
        gen = torch.Generator().manual_seed(40)
        truth = torch.rand(1, 3, hw, hw, generator=gen) * 2 - 1
        y = op.forward(truth, **fkw)
        y = y + 0.05 * torch.randn(*y.shape, generator=gen)
        out[f"{oname}.y"] = np32(y)
        for t in (900, 500, 0):
            x_prev0 = torch.randn(2, 3, hw, hw, generator=gen)
            out[f"{oname}.t{t}.x_prev"] = np32(x_prev0)
            for method, params in (("ps", {"scale": 0.3}), ("ps_semantic", {"scale": 0.7, "sem_guid_scale": 0.0}),
                                   ("ps_semantic2", {"scale": 0.7, "sem_guid_scale": 0.0, "norm_exp": 2}),
                                   ("ps_anneal", {"scale": 0.3})):
                mname = "ps_semantic" if method.startswith("ps_semantic") else method
                with quiet():
                    cm = CM.get_conditioning_method(mname, op, noiser, **params)
                x_prev = x_prev0.clone().requires_grad_()
                torch.manual_seed(1000 + t)
                noise = torch.randn_like(x_prev)
                torch.manual_seed(1000 + t)
                with quiet():
                    res = ddpm.p_sample(model=model, x=x_prev, t=torch.tensor([t]))
                    sample0 = res["sample"].detach().clone()
                    r = cm.conditioning(x_prev=x_prev, x_t=res["sample"], x_0_hat=res["pred_xstart"],
                                        measurement=y, noisy_measurement=y,
                                        beta_scale=float(ddpm.betas[t]), t=t / 1000.0, **fkw)
                out[f"{oname}.t{t}.noise"] = np32(noise)
                out[f"{oname}.t{t}.x0_hat"] = np32(res["pred_xstart"])
                out[f"{oname}.t{t}.sample"] = np32(sample0)
                out[f"{oname}.t{t}.{method}.ret0"] = np32(r[0])
                out[f"{oname}.t{t}.{method}.ret1"] = np32(r[1])
                out[f"{oname}.t{t}.{method}.ret2"] = np32(torch.as_tensor(r[2]).float())
    save("conditioning.npz", **out)

    # ------------------------------------------------------------- 5. free-running base loop
    out = {}

    def run_base_loop(tag, op, fkw, respacing, hw, n, seed, scale, norm_exp=1):
        with quiet():
            smp = GD.create_sampler(sampler="ddpm", timestep_respacing=respacing, **DIFF)
            cm = CM.get_conditioning_method("ps_semantic", op, noiser, scale=scale, sem_guid_scale=0.0,
                                            norm_exp=norm_exp)
        gen = torch.Generator().manual_seed(seed)
        truth = torch.rand(1, 3, hw, hw, generator=gen) * 2 - 1
        y = op.forward(truth, **fkw)
        y = y + 0.05 * torch.randn(*y.shape, generator=gen)
        x_start = torch.randn(n, 3, hw, hw, generator=gen)
        norms, x0s = [], []

        def cond(**kw):
            r = cm.conditioning(**kw, **fkw)
            norms.append(np32(r[1]))
            x0s.append(np32(kw["x_0_hat"]))
            return r

        torch.manual_seed(seed + 1)
        with quiet():
            img, dist, _ = smp.p_sample_loop(model=model, x_start=x_start.clone().requires_grad_(),
                                             measurement=y, measurement_cond_fn=cond, record=False,
                                             save_root=None)
        out[f"{tag}.y"] = np32(y)
        # the reference's q_sample draws randn_like(y) every step; Resizer returns a transposed view, and
        # torch's CPU normal_() consumes the generator differently for non-contiguous tensors, so the
        # RNG stream depends on y's strides: record them for the parity replay
        out[f"{tag}.y_stride"] = np.asarray(y.stride(), dtype=np.int64)
        out[f"{tag}.x_start"] = np32(x_start)
        out[f"{tag}.rng_seed"] = np.int64(seed + 1)
        out[f"{tag}.final"] = np32(img)
        out[f"{tag}.norms"] = np.stack(norms)
        out[f"{tag}.x0_first"] = x0s[0]
        out[f"{tag}.x0_last"] = x0s[-1]

    run_base_loop("gauss.r20", gb, {}, "20", 64, 4, 50, 0.5)
    run_base_loop("sr4.r20", sr4s, {}, "20", 64, 4, 51, 1.0)
    run_base_loop("inpaint.r20", inp, {"mask": mask}, "20", 64, 4, 52, 0.5)
    run_base_loop("motion.r20", mbo, {}, "20", 64, 2, 53, 0.5, norm_exp=2)
    run_base_loop("sr4.full1000", sr4s, {}, "", 64, 2, 54, 1.0)
    save("loop.npz", **out)

    # ------------------------------------------------------------- 6. search_ddpm + resampling ids
    out = {}
    for tag, op, hw, seed in (("sr4", sr4s, 64, 60), ("gauss", gb, 64, 61)):
        with quiet():
            smp = GD.create_sampler(sampler="search_ddpm", timestep_respacing="20", **DIFF)
        gen = torch.Generator().manual_seed(seed)
        truth = torch.rand(1, 3, hw, hw, generator=gen) * 2 - 1
        y = op.forward(truth)
        y = y + 0.05 * torch.randn(*y.shape, generator=gen)
        x_start = torch.randn(5, 3, hw, hw, generator=gen)
        torch.manual_seed(seed + 1)
        with quiet() as buf:
            img = smp.p_sample_loop(model=model, x_start=x_start.clone(), measurement=y,
                                    measurement_cond_fn=None, record=False, save_root=None, operator=op)
        found = re.findall(r"Best path = (\d+), cost = ([0-9.eE+-]+)", buf.getvalue())
        out[f"{tag}.y"] = np32(y)
        out[f"{tag}.x_start"] = np32(x_start)
        out[f"{tag}.rng_seed"] = np.int64(seed + 1)
        out[f"{tag}.final"] = np32(img)
        out[f"{tag}.best"] = np.asarray([int(a) for a, _ in found], dtype=np.int64)
        out[f"{tag}.cost"] = np.asarray([float(b) for _, b in found], dtype=np.float64)
    # TTC_DDIM-style multinomial resample (gaussian_diffusion.py:689-698): ids for fixed distances
    dist = torch.tensor([3.0, 1.0, 250.0, 40.0, 0.5, 90.0, 12.0, 700.0])
    wts = torch.exp(-dist / 100.0)
    torch.manual_seed(77)
    out["resample.dist"] = np32(dist)
    out["resample.seed"] = np.int64(77)
    out["resample.ids"] = torch.multinomial(wts, 8, replacement=True).numpy().astype(np.int64)
    save("search.npz", **out)
    ddim_fixtures(GD, CM, MS)
    project_fixtures(GD, CM, MS)
    resample_fixtures(GD, CM, MS)


def ddim_fixtures(GD, CM, MS):
    """7. DDIM step (gaussian_diffusion.py:479-509) and the ttc_ddim loop (:644-707) -> ddim.npz"""
    out = {}
    with quiet():
        ddim = GD.create_sampler(sampler="ddim", timestep_respacing="", **DIFF)
    gen = torch.Generator().manual_seed(12)
    x = torch.randn(2, 3, 8, 8, generator=gen)
    out["x"] = np32(x)
    out["w_x0"] = np32(torch.randn(2, 3, 8, 8, generator=gen))
    out["w_s"] = np32(torch.randn(2, 3, 8, 8, generator=gen))
    for t, eta in ((999, 0.0), (500, 0.0), (1, 0.0), (0, 0.0), (500, 0.5), (0, 0.5)):
        tag = f"t{t}.eta{eta:g}"
        a = float(ddim.sqrt_recip_alphas_cumprod[t])
        b = float(ddim.sqrt_recipm1_alphas_cumprod[t])
        target = 1.4 * torch.tanh(torch.randn(2, 3, 8, 8, generator=gen))
        eps = (a * x - target) / b
        v = torch.rand(2, 3, 8, 8, generator=gen) * 2 - 1
        mo = torch.cat([eps, v], dim=1).requires_grad_()
        xx = x.clone().requires_grad_()
        torch.manual_seed(200 + t)
        noise = torch.randn_like(xx)
        torch.manual_seed(200 + t)
        with quiet():
            res = ddim.p_sample(model=lambda z, ts: mo, x=xx, t=torch.tensor([t]), eta=eta)
        loss = (res["pred_xstart"] * torch.from_numpy(out["w_x0"])).sum() + \
               (res["sample"] * torch.from_numpy(out["w_s"])).sum()
        gx, gmo = torch.autograd.grad(loss, [xx, mo])
        out[f"{tag}.model_out"] = np32(mo)
        out[f"{tag}.noise"] = np32(noise)
        out[f"{tag}.x0_hat"] = np32(res["pred_xstart"])
        out[f"{tag}.sample"] = np32(res["sample"])
        out[f"{tag}.g_x"] = np32(gx)
        out[f"{tag}.g_model_out"] = np32(gmo)

    # ttc_ddim free-running loop.  Of the shipped conditioning methods only the two-value ones fit the loop's
    # `img, distance = measurement_cond_fn(...)` (:672): 'mcg' on a blur operator (transpose = identity) is the
    # combination that runs end to end in the reference.
    model = StandInModel()
    noiser = MS.get_noise("gaussian", sigma=0.05)
    gb = MS.get_operator("gaussian_blur", kernel_size=61, intensity=3.0, device="cpu")
    for tag, respacing, n, seed, scale in (("gauss.r20", "20", 4, 70, 0.5), ("gauss.r50", "ddim50", 6, 71, 0.3)):
        with quiet():
            smp = GD.create_sampler(sampler="ttc_ddim", timestep_respacing=respacing, **DIFF)
            cm = CM.get_conditioning_method("mcg", gb, noiser, scale=scale)
        gen = torch.Generator().manual_seed(seed)
        truth = torch.rand(1, 3, 64, 64, generator=gen) * 2 - 1
        y = gb.forward(truth)
        y = y + 0.05 * torch.randn(*y.shape, generator=gen)
        x_start = torch.randn(n, 3, 64, 64, generator=gen)
        norms = []

        def cond(**kw):
            r = cm.conditioning(**kw)
            norms.append(np32(r[1]))
            return r

        torch.manual_seed(seed + 1)
        with quiet() as buf:
            img, dist = smp.p_sample_loop(model=model, x_start=x_start.clone().requires_grad_(), measurement=y,
                                          measurement_cond_fn=cond, record=False, save_root=None)
        ev = re.findall(r"Resampling, ids = tensor\(\[([0-9, ]+)\]\), idx = (\d+)", buf.getvalue())
        out[f"{tag}.y"] = np32(y)
        out[f"{tag}.x_start"] = np32(x_start)
        out[f"{tag}.rng_seed"] = np.int64(seed + 1)
        out[f"{tag}.final"] = np32(img)
        out[f"{tag}.distance"] = np32(dist)
        out[f"{tag}.norms"] = np.stack(norms)
        out[f"{tag}.resample_idx"] = np.asarray([int(i) for _, i in ev], dtype=np.int64)
        out[f"{tag}.resample_ids"] = np.asarray([[int(v) for v in ids.split(",")] for ids, _ in ev],
                                                dtype=np.int64).reshape(len(ev), n)
    save("ddim.npz", **out)


def project_fixtures(GD, CM, MS):
    """8. base loop with the DiffStateGrad projection (gaussian_diffusion.py:203-204, 240-251) -> project.npz"""
    out = {}
    model = StandInModel()
    noiser = MS.get_noise("gaussian", sigma=0.05)
    gb = MS.get_operator("gaussian_blur", kernel_size=61, intensity=3.0, device="cpu")
    for tag, n, period, seed in (("gauss.n1.p5", 1, 5, 80), ("gauss.n2.p4", 2, 4, 81)):
        with quiet():
            smp = GD.create_sampler(sampler="ddpm", timestep_respacing="20", **DIFF)
            cm = CM.get_conditioning_method("ps_semantic", gb, noiser, scale=0.5, sem_guid_scale=0.0)
        gen = torch.Generator().manual_seed(seed)
        truth = torch.rand(1, 3, 64, 64, generator=gen) * 2 - 1
        y = gb.forward(truth)
        y = y + 0.05 * torch.randn(*y.shape, generator=gen)
        x_start = torch.randn(n, 3, 64, 64, generator=gen)
        torch.manual_seed(seed + 1)
        with quiet() as buf:
            img, dist, _ = smp.p_sample_loop(model=model, x_start=x_start.clone().requires_grad_(), measurement=y,
                                             measurement_cond_fn=cm.conditioning, record=False, save_root=None,
                                             project=True, period=period)
        out[f"{tag}.y"] = np32(y)
        out[f"{tag}.x_start"] = np32(x_start)
        out[f"{tag}.rng_seed"] = np.int64(seed + 1)
        out[f"{tag}.period"] = np.int64(period)
        out[f"{tag}.final"] = np32(img)
        out[f"{tag}.distance"] = np32(dist)
    # the rank rule alone (diffstategrad_utils.py:4-44) on a fixed state
    from guided_diffusion.diffstategrad_utils import compute_svd_and_adaptive_rank
    gen = torch.Generator().manual_seed(82)
    z = torch.randn(2, 3, 64, 64, generator=gen) * torch.linspace(2.0, 0.05, 64).view(1, 1, 1, 64)
    for cutoff in (0.99, 0.9, 0.5):
        out[f"rank.cut{cutoff:g}"] = np.int64(compute_svd_and_adaptive_rank(z, cutoff)[3])
    out["rank.z"] = np32(z)
    save("project.npz", **out)


def resample_fixtures(GD, CM, MS):
    """9. SearchDDPM.resample_update (gaussian_diffusion.py:515-587), called directly (no loop of the reference calls
    it): all four potential types x {first call, update without resampling, resample + update} -> resample.npz"""
    out = {}
    with quiet():
        smp = GD.create_sampler(sampler="search_ddpm", timestep_respacing="20", **DIFF)
    gb = MS.get_operator("gaussian_blur", kernel_size=61, intensity=3.0, device="cpu")
    sr4 = MS.get_operator("super_resolution", in_shape=(1, 3, 64, 64), scale_factor=4, device="cpu")
    n = 6
    for tag, op, seed in (("gauss", gb, 90), ("sr4", sr4, 91)):
        gen = torch.Generator().manual_seed(seed)
        truth = torch.rand(1, 3, 64, 64, generator=gen) * 2 - 1
        y = op.forward(truth)
        y = y + 0.05 * torch.randn(*y.shape, generator=gen)
        # denoised candidates at different distances from the truth; candidates carry their index in one pixel
        spread = torch.tensor([0.02, 0.3, 0.1, 0.6, 0.05, 0.2]).view(n, 1, 1, 1)
        denoised = (truth + spread * torch.randn(n, 3, 64, 64, generator=gen)).clamp(-1, 1)
        cands = torch.randn(n, 3, 64, 64, generator=gen)
        cands[:, 0, 0, 0] = torch.arange(n, dtype=torch.float32)
        prev = torch.rand(n, generator=gen) * 60.0 + 1.0
        out[f"{tag}.y"] = np32(y)
        out[f"{tag}.denoised"] = np32(denoised)
        out[f"{tag}.candidates"] = np32(cands)
        out[f"{tag}.prev_costs"] = np32(prev)
        for pot in ("mean", "min", "diff", "curr"):
            cases = (("first", dict(prev_costs=None, resample=True)),
                     ("noresample", dict(prev_costs=prev.clone(), resample=False)),
                     ("resample", dict(prev_costs=prev.clone(), resample=True, rs_temp=0.05, steps_done=3)))
            for cname, kw in cases:
                torch.manual_seed(500 + seed)
                with quiet():
                    c2, net = smp.resample_update(cands.clone(), denoised.clone(), op, y, potential_type=pot, **kw)
                out[f"{tag}.{pot}.{cname}.net"] = np32(net)
                out[f"{tag}.{pot}.{cname}.ids"] = c2[:, 0, 0, 0].round().numpy().astype(np.int64)
        out[f"{tag}.rng_seed"] = np.int64(500 + seed)
    # equal potentials: the draw is skipped (:545)
    torch.manual_seed(1)
    with quiet():
        c2, net = smp.resample_update(cands.clone(), denoised.clone(), gb, torch.from_numpy(out["gauss.y"]),
                                      prev_costs=torch.full((n,), 7.0), resample=True, potential_type="min")
    out["gauss.flat.ids"] = c2[:, 0, 0, 0].round().numpy().astype(np.int64)
    save("resample.npz", **out)


if __name__ == "__main__":
    extra = {"--only-ddim": ddim_fixtures, "--only-project": project_fixtures, "--only-resample": resample_fixtures}
    chosen = [fn for flag, fn in extra.items() if flag in sys.argv]
    if chosen:                          # add fixtures without rewriting the others
        install_stubs()
        with quiet():
            from guided_diffusion import condition_methods as CM
            from guided_diffusion import gaussian_diffusion as GD
            from guided_diffusion import measurements as MS
        torch.set_num_threads(8)
        for fn in chosen:
            fn(GD, CM, MS)
    else:
        main()
