"""Samplers and the DPS / test-time-compute loops -- the reference's sampler API
(guided_diffusion/gaussian_diffusion.py) driving the HIP hot path.

    create_sampler(sampler, steps, noise_schedule, model_mean_type, model_var_type, dynamic_threshold,
                   clip_denoised, rescale_timesteps, timestep_respacing="") -> s        (reference :34-56)
    s.p_sample_loop(model=, x_start=, measurement=, measurement_cond_fn=, record=, save_root=, **kw)
    s.p_sample(model, x, t) -> {'sample', 'pred_xstart'};  s.q_sample(x0, t);  s.num_timesteps;  s.betas

Per step the host does: one UNet forward (PyTorch-ROCm), three fused HIP launches
(kernels.step_fwd / step_bwd / step_update), one UNet VJP (torch.autograd.grad with the
HIP-produced cotangent).  No table upload, no `.item()`, no host sync inside the loop.

Deliberate differences from the reference snapshot:
  * the base loop applies the *intended* DPS update for every conditioning method: methods that
    return the updated x_t ('ps', 'ps_anneal', ...) are taken at their word, methods that return
    the gradient ('ps_semantic') have it subtracted -- the reference loop does the latter for
    both and so zeroes the image under 'ps';
  * called with the test-time-compute driver's keyword set (operator=...), p_sample_loop returns
    the bare [N,C,H,W] tensor that driver expects (sample_condition_batched_ttc.py:181-183);
    the per-particle distances are kept in `sampler.last_measurement_distance`.
"""
import functools
import math
import os

import numpy as np
import torch

from . import kernels
from .condition_methods import ConditioningMethod
from .diffstategrad_utils import apply_diffstategrad, compute_svd_and_adaptive_rank
from .posterior_mean_variance import get_mean_processor, get_var_processor

__SAMPLER__ = {}


def register_sampler(name: str):
    def wrapper(cls):
        if __SAMPLER__.get(name, None):
            raise NameError(f"Name {name} is already registered!")
        __SAMPLER__[name] = cls
        return cls
    return wrapper


def get_sampler(name: str):
    if __SAMPLER__.get(name, None) is None:
        raise NameError(f"Name {name} is not defined!")
    return __SAMPLER__[name]


def create_sampler(sampler, steps, noise_schedule, model_mean_type, model_var_type, dynamic_threshold,
                   clip_denoised, rescale_timesteps, timestep_respacing=""):
    cls = get_sampler(name=sampler)
    betas = get_named_beta_schedule(noise_schedule, steps)
    if not timestep_respacing:
        timestep_respacing = [steps]
    return cls(use_timesteps=space_timesteps(steps, timestep_respacing), betas=betas,
               model_mean_type=model_mean_type, model_var_type=model_var_type,
               dynamic_threshold=dynamic_threshold, clip_denoised=clip_denoised,
               rescale_timesteps=rescale_timesteps)


# ------------------------------------------------------------------ schedules
def get_named_beta_schedule(schedule_name, num_diffusion_timesteps):
    if schedule_name == "linear":
        scale = 1000 / num_diffusion_timesteps
        return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)
    if schedule_name == "cosine":
        return betas_for_alpha_bar(num_diffusion_timesteps,
                                   lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    n = num_diffusion_timesteps
    return np.array([min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), max_beta) for i in range(n)])


def space_timesteps(num_timesteps, section_counts):
    """Subset of the base timesteps kept by a respacing spec ("", "250", "10,15,20", "ddim50")."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    elif isinstance(section_counts, int):
        section_counts = [section_counts]
    per, extra = divmod(num_timesteps, len(section_counts))
    start, kept = 0, []
    for i, count in enumerate(section_counts):
        size = per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            kept.append(start + round(pos))
            pos += stride
        start += size
    return set(kept)


# ------------------------------------------------------------------ base class
class GaussianDiffusion:
    def __init__(self, betas, model_mean_type, model_var_type, dynamic_threshold, clip_denoised,
                 rescale_timesteps):
        betas = np.array(betas, dtype=np.float64)           # float64 tables, fp32 at the point of use
        self.betas = betas
        assert betas.ndim == 1, "betas must be 1-D"
        assert (0 < betas).all() and (betas <= 1).all(), "betas must be in (0..1]"
        self.num_timesteps = int(betas.shape[0])
        self.rescale_timesteps = rescale_timesteps

        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1],
                                                               self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)

        self.model_mean_type, self.model_var_type = model_mean_type, model_var_type
        self.dynamic_threshold, self.clip_denoised = dynamic_threshold, clip_denoised
        self.mean_processor = get_mean_processor(model_mean_type, betas=betas, dynamic_threshold=dynamic_threshold,
                                                 clip_denoised=clip_denoised)
        self.var_processor = get_var_processor(model_var_type, betas=betas)
        # the configuration every shipped YAML uses runs S1 as one HIP kernel
        self.hip_posterior = (model_mean_type == 'epsilon' and model_var_type == 'learned_range'
                              and clip_denoised and not dynamic_threshold)
        log_betas = np.log(betas)
        self.step_coefs = [kernels.make_coefs(self.sqrt_recip_alphas_cumprod[t], self.sqrt_recipm1_alphas_cumprod[t],
                                              self.posterior_mean_coef1[t], self.posterior_mean_coef2[t],
                                              self.posterior_log_variance_clipped[t], log_betas[t], t != 0)
                           for t in range(self.num_timesteps)]
        #: draw noise on the host generator in the reference's order (parity tests); default: on the device
        self.rng_parity = False
        #: strides of the reference run's measurement tensor (only read when rng_parity is set)
        self.parity_measurement_stride = None
        self.progress = False
        self.last_measurement_distance = None
        self.last_semantic_distance = None
        self._ts_cache = {}
        self._bufs = None
        self._step_semantic = None
        #: > 1: the fused base loop runs its N particles as this many independent sub-batches, each a whole chain (model
        #: call, three HIP launches, model VJP) on its own HIP stream with its own operator handle (kernels.ParticleGroups):
        #: the tile kernels' load / compute / store phases add up inside one chain, side by side they fill each other's gaps.
        #: Per-particle results do not depend on it (the noise is still drawn for the whole batch, in the same order).
        self.particle_groups = 1
        self._pgroups = None

    # -- RNG ---------------------------------------------------------------
    def _randn(self, like, stride=None, shape=None):
        """standard-normal draw shaped like `like` (or `shape` on like's device: no dummy tensor is allocated)"""
        if shape is not None:
            if self.rng_parity:
                return torch.randn(tuple(shape), dtype=torch.float32).to(like.device)
            return torch.randn(tuple(shape), dtype=torch.float32, device=like.device)
        if self.rng_parity:
            # replay of the reference's host RNG stream (tests): torch's CPU normal_() consumes the
            # generator differently for non-contiguous tensors, so the layout is part of the stream
            proxy = torch.empty(like.shape, dtype=torch.float32) if stride is None else \
                torch.empty_strided(tuple(like.shape), tuple(stride), dtype=torch.float32)
            return torch.randn_like(proxy).contiguous().to(like.device)
        return torch.randn_like(like, dtype=torch.float32)

    # -- q ------------------------------------------------------------------
    def q_sample(self, x_start, t):
        """reference :134-151 (two scalars times tensors: not worth a kernel, result unused by ps*)"""
        noise = self._randn(x_start, self.parity_measurement_stride)
        t = int(t)
        return float(np.float32(self.sqrt_alphas_cumprod[t])) * x_start + \
            float(np.float32(self.sqrt_one_minus_alphas_cumprod[t])) * noise

    # -- model timestep -----------------------------------------------------
    def _model_timesteps(self, device):
        """device tensor [T] of what the UNet receives as t for each loop index (reference :333-336)"""
        key = str(device)
        if key not in self._ts_cache:
            ts = np.arange(self.num_timesteps, dtype=np.float64)
            ts = ts * (1000.0 / self.num_timesteps) if self.rescale_timesteps else ts
            self._ts_cache[key] = torch.tensor(ts, dtype=torch.float32 if self.rescale_timesteps else torch.int64,
                                               device=device)
        return self._ts_cache[key]

    def _call_model(self, model, x, idx):
        return model(x, self._model_timesteps(x.device)[idx:idx + 1])

    # -- S1 -------------------------------------------------------------------
    def p_mean_variance(self, model, x, t):
        idx = int(t)
        model_output = self._call_model(model, x, idx)
        if model_output.shape[1] == 2 * x.shape[1]:
            model_output, model_var_values = torch.split(model_output, x.shape[1], dim=1)
        else:
            model_var_values = model_output
        model_mean, pred_xstart = self.mean_processor.get_mean_and_xstart(x, idx, model_output)
        model_variance, model_log_variance = self.var_processor.get_variance(model_var_values, idx)
        return {'mean': model_mean, 'variance': model_variance, 'log_variance': model_log_variance,
                'pred_xstart': pred_xstart}

    def p_sample(self, model, x, t):
        raise NotImplementedError

    def sample_coefs(self, idx):
        """struct dpsx_coefs of this sampler's step at loop index idx (DDPM record; DDIM overrides)"""
        return self.step_coefs[idx]

    def _scale_timesteps(self, t):
        return t.float() * (1000.0 / self.num_timesteps) if self.rescale_timesteps else t

    # -- fusion plan ----------------------------------------------------------
    @staticmethod
    def _unwrap_cond_fn(fn):
        """-> (ConditioningMethod or None, bound keyword arguments)"""
        kw = {}
        while isinstance(fn, functools.partial):
            kw = {**fn.keywords, **kw}
            fn = fn.func
        method = getattr(fn, '__self__', None)
        if isinstance(method, ConditioningMethod) and getattr(fn, '__name__', '') == 'conditioning':
            return method, kw
        return None, kw

    def _fusion_plan(self, measurement_cond_fn, x_start):
        if not (self.hip_posterior and isinstance(self, (DDPM, DDIM))):
            return None
        method, kw = self._unwrap_cond_fn(measurement_cond_fn)
        if method is None or method.fused_spec(**kw) is None:
            return None
        op = method.operator
        if op.name == 'inpainting':
            if kw.get('mask', None) is None:
                return None
            handle = op.hip_handle_for(kw['mask'])
        elif hasattr(op, 'hip_handle'):
            handle = op.hip_handle(x_start)
        else:
            return None
        return method, kw, handle

    def _buffers(self, handle, x):
        n, c, h, w = x.shape
        b = self._bufs
        if b is None or b[0] is not handle or b[1].shape != (n, c, h, w) or b[1].x0_hat.device != x.device:
            self._bufs = (handle, kernels.StepBuffers(handle, n, c, h, w, x.device))
        return self._bufs[1]

    def dps_step(self, model, x_prev, idx, measurement, method, cond_kw, handle, noise=None, loop_kw=None,
                 want_x0=False):
        """One fused DPS step at loop index idx.  Returns (x_next, norm[N]) -- device tensors that live in the
        sampler's persistent step buffers (valid until the next step; the loops clone what they hand out).
        loop_kw: the keyword arguments the calling loop passes to measurement_cond_fn besides the tensors
        (base loop: beta_scale, t -- reference :230-236; ttc_ddim: none -- :678-682)."""
        x_prev = x_prev.detach().requires_grad_()
        with torch.enable_grad():
            model_out = self._call_model(model, x_prev, idx)
        mo = kernels.f32c(model_out.detach(), "model output")
        if mo.shape[1] != 2 * x_prev.shape[1]:
            raise ValueError("the fused DPS step needs a learned-sigma model ([N, 2C, H, W] output)")
        coefs = self.sample_coefs(idx)
        if noise is None:
            noise = self._randn(x_prev)
        buf = self._buffers(handle, x_prev)
        xp = kernels.f32c(x_prev.detach(), "x_t")
        y = kernels.f32c(measurement, "measurement")
        if loop_kw is None:
            loop_kw = {'beta_scale': self.betas[idx], 't': idx / self.num_timesteps}
        spec = method.fused_spec(**loop_kw, **cond_kw)
        # x0_hat is consumed inside K1 (A(x0_hat), the clamp gate): the image itself is written out only when something
        # reads it afterwards -- the semantic term's embedder, a progress snapshot (want_x0)
        kernels.step_fwd(handle, buf, xp, mo, noise, y, coefs, want_x0=want_x0 or "semantic" in spec)
        g_sem, self._step_semantic = None, None
        if "semantic" in spec:            # embedder forward + VJP on x0_hat (torch), between the two HIP halves
            g_sem, self._step_semantic = spec["semantic"](buf.x0_hat)
        kernels.step_bwd(handle, buf, y, spec["scale"], spec["power"], coefs, g_x0_extra=g_sem)
        g_unet = None
        if model_out.requires_grad:
            (g_unet,) = torch.autograd.grad(model_out, x_prev, grad_outputs=buf.g_model_out.to(model_out.dtype))
            g_unet = kernels.f32c(g_unet, "UNet VJP")
        x_next = kernels.step_update(buf, g_unet, coefs)
        return x_next, buf.norm

    def _particle_group_set(self, method, cond_kw, x):
        """the sampler's kernels.ParticleGroups for this operator / batch shape (built once, reused across trajectories)"""
        n, c, h, w = x.shape
        op, mask = method.operator, cond_kw.get('mask', None)
        key = (id(op), None if mask is None else (mask.data_ptr(), mask._version), (n, c, h, w), str(x.device),
               int(self.particle_groups))
        if self._pgroups is None or self._pgroups[0] != key:
            self._pgroups = (key, kernels.ParticleGroups(op, n, c, h, w, x.device, self.particle_groups, mask=mask, like=x))
        return self._pgroups[1]

    def dps_step_grouped(self, model, img, idx, measurement, method, cond_kw, pg, noise, loop_kw=None, want_x0=False):
        """dps_step over kernels.ParticleGroups: every group's whole step -- model call, K1, [semantic term], K2, model VJP,
        K3 -- is enqueued on the group's own stream; nothing is joined between steps (group j's next step reads only what
        group j wrote).  img, noise: full-batch tensors.  Returns (x_next [N, C, H, W], norm [N]) -- views of the group
        set's full-batch buffers, valid on the CALLER's stream only after pg.join()."""
        coefs = self.sample_coefs(idx)
        y = kernels.f32c(measurement, "measurement")
        if loop_kw is None:
            loop_kw = {'beta_scale': self.betas[idx], 't': idx / self.num_timesteps}
        spec = method.fused_spec(**loop_kw, **cond_kw)
        pg.fork()                 # the noise (and, on the first step, x_start) was produced on the caller's stream
        sems = []
        for j in range(len(pg)):
            with torch.cuda.stream(pg.streams[j]):
                x_prev = img[pg.slices[j]].detach().requires_grad_()
                with torch.enable_grad():
                    model_out = self._call_model(model, x_prev, idx)
                mo = kernels.f32c(model_out.detach(), "model output")
                if mo.shape[1] != 2 * x_prev.shape[1]:
                    raise ValueError("the fused DPS step needs a learned-sigma model ([N, 2C, H, W] output)")
                pg.step_fwd(j, kernels.f32c(x_prev.detach(), "x_t"), mo, noise, y, coefs,
                            want_x0=want_x0 or "semantic" in spec)
                g_sem = None
                if "semantic" in spec:
                    g_sem, sem = spec["semantic"](pg.bufs[j].x0_hat)
                    sems.append(sem)
                pg.step_bwd(j, y, spec["scale"], spec["power"], coefs, g_x0_extra=g_sem)
                g_unet = None
                if model_out.requires_grad:
                    (g_unet,) = torch.autograd.grad(model_out, x_prev,
                                                    grad_outputs=pg.bufs[j].g_model_out.to(model_out.dtype))
                    g_unet = kernels.f32c(g_unet, "UNet VJP")
                pg.step_update(j, g_unet, coefs)
        self._step_semantic = sems if sems else None
        return pg.x_next(), pg.full.norm

    # -- the base loop (reference :175-303) -------------------------------------
    def p_sample_loop(self, model, x_start, measurement, measurement_cond_fn, record, save_root, **kwargs):
        img = x_start.detach()
        kernels.require_cuda(img, "x_start")
        ttc_driver_call = 'operator' in kwargs      # sample_condition_batched_ttc.py:91-100
        plan = self._fusion_plan(measurement_cond_fn, img)
        method, _ = self._unwrap_cond_fn(measurement_cond_fn)
        returns_gradient = True if method is None else method.returns_gradient
        distance, semantic = None, torch.zeros((), device=img.device)
        steps = range(self.num_timesteps - 1, -1, -1)
        if self.progress:
            from tqdm.auto import tqdm
            steps = tqdm(list(steps))
        # DiffStateGrad (reference :203-204, 240-251): off by default; a projected step needs the gradient
        # itself, so it takes the per-op path; every other step keeps the fused launches
        period, project = kwargs.get('period', 20), kwargs.get('project', False)
        # particle groups on streams (sampler.particle_groups > 1): only where every step is a fused step
        pg = None
        if plan is not None and self.particle_groups > 1 and img.shape[0] > 1 and not (project and returns_gradient):
            pg = self._particle_group_set(plan[0], plan[1], img)
        for idx in steps:
            projecting = project and returns_gradient and period != 0 and idx % period == 0
            if plan is not None and not projecting:
                noise = self._randn(img)
                if self.rng_parity:
                    # the reference's q_sample draw (:224), result unused by ps*
                    self._randn(measurement, self.parity_measurement_stride)
                snapshot = bool(record) and idx % 100 == 0
                if pg is not None:
                    img, distance = self.dps_step_grouped(model, img, idx, measurement, plan[0], plan[1], pg, noise,
                                                          want_x0=snapshot)
                    if snapshot:
                        pg.join()
                else:
                    img, distance = self.dps_step(model, img, idx, measurement, plan[0], plan[1], plan[2], noise=noise,
                                                  want_x0=snapshot)
                if self._step_semantic is not None:
                    semantic = self._step_semantic
            else:
                img = img.detach().requires_grad_()
                time = torch.tensor([idx], device=img.device)
                out = self.p_sample(x=img, t=time, model=model)
                noisy_measurement = self.q_sample(measurement, t=idx)
                ret = measurement_cond_fn(x_t=out['sample'], measurement=measurement,
                                          noisy_measurement=noisy_measurement, x_prev=img,
                                          x_0_hat=out['pred_xstart'], beta_scale=self.betas[idx],
                                          t=idx / self.num_timesteps)
                if not isinstance(ret, tuple):
                    ret = (ret,)
                if returns_gradient:
                    grad = ret[0].detach()
                    if projecting:
                        U, sv, Vh, rank = compute_svd_and_adaptive_rank(z_t=out['sample'].detach(), var_cutoff=0.99)
                        grad = apply_diffstategrad(norm_grad=grad, iteration_count=idx, period=period, U=U, s=sv,
                                                   Vh=Vh, adaptive_rank=rank)
                        if grad.shape[0] != out['sample'].shape[0]:       # [1, C, H, W] broadcasts at :255
                            grad = grad.expand_as(out['sample']).contiguous()
                    img = kernels.update(out['sample'].detach(), grad)                # :255
                else:
                    img = ret[0].detach()
                distance = ret[1] if len(ret) > 1 else None
                semantic = ret[2] if len(ret) > 2 and returns_gradient else semantic
            if record and idx % 100 == 0:
                self._record(save_root, kwargs.get('path_curr_group_idx', 0), idx)
        if pg is not None:
            pg.join()
            if isinstance(semantic, list):
                semantic = torch.cat([v.reshape(-1) for v in semantic])
        # the fused step hands back views of the persistent ping-pong / norm buffers, which the next trajectory on
        # this sampler overwrites: what leaves the loop is a copy (one per 1000-step trajectory)
        img = img.clone()
        distance = distance.clone() if torch.is_tensor(distance) else distance
        self.last_measurement_distance, self.last_semantic_distance = distance, semantic
        if ttc_driver_call:
            return img
        return img, distance, semantic

    def _record(self, save_root, group, idx):
        """x0_hat snapshot of particle 0 every 100 steps (reference :296-301); best effort, off the hot path"""
        try:
            import matplotlib.pyplot as plt
        except ImportError:
            return
        if self._pgroups is not None and self.particle_groups > 1:
            x0 = self._pgroups[1].full.x0_hat[0]
        else:
            x0 = self._bufs[1].x0_hat[0] if self._bufs is not None else None
        if x0 is None or save_root is None:
            return
        a = x0.detach().float().cpu().numpy().transpose(1, 2, 0)
        a = (a - a.min()) / max(float(a.max() - a.min()), 1e-12)
        path = os.path.join(save_root, f"progress/path#{group + 1}")
        os.makedirs(path, exist_ok=True)
        plt.imsave(os.path.join(path, f"x_{str(idx).zfill(4)}.png"), a)


class SpacedDiffusion(GaussianDiffusion):
    """A diffusion process that keeps a subset of the base timesteps (reference :395-445)."""

    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(kwargs["betas"])
        base_abar = np.cumprod(1.0 - np.array(kwargs["betas"], dtype=np.float64))
        self.timestep_map, new_betas, last = [], [], 1.0
        for i, abar in enumerate(base_abar):
            if i in self.use_timesteps:
                new_betas.append(1 - abar / last)
                last = abar
                self.timestep_map.append(i)
        kwargs["betas"] = np.array(new_betas)
        super().__init__(**kwargs)

    def _model_timesteps(self, device):
        key = str(device)
        if key not in self._ts_cache:
            ts = np.array(self.timestep_map, dtype=np.float64)
            if self.rescale_timesteps:                                   # reference :455-463
                self._ts_cache[key] = torch.tensor(ts, dtype=torch.float32, device=device) * \
                    (1000.0 / self.original_num_steps)
            else:
                self._ts_cache[key] = torch.tensor(ts, dtype=torch.int64, device=device)
        return self._ts_cache[key]

    def _scale_timesteps(self, t):
        return t


@register_sampler(name='ddpm')
class DDPM(SpacedDiffusion):
    def p_sample(self, model, x, t, noise=None):
        """reference :466-476.  The noise is drawn whether or not it is used (t == 0), as there."""
        idx = int(t)
        if noise is None:
            noise = self._randn(x)
        if self.hip_posterior:
            model_output = self._call_model(model, x, idx)
            x0, sample = kernels.PosteriorStepFn.apply(x, model_output, noise, self.step_coefs[idx])
            return {'sample': sample, 'pred_xstart': x0}
        out = self.p_mean_variance(model, x, t)
        sample = out['mean']
        if idx != 0:
            sample = sample + torch.exp(0.5 * out['log_variance']) * noise
        return {'sample': sample, 'pred_xstart': out['pred_xstart']}


@register_sampler(name='ddim')
class DDIM(SpacedDiffusion):
    #: the reference's p_sample takes eta per call and never passes it (:481); kept as the default
    eta = 0.0

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        # the default-eta records of every step, built once here (numpy fp32 scalar arithmetic, ~30 us each): built lazily
        # inside the loop they cost ten times that on the host once a process group's threads are alive (measured with a
        # one-rank RCCL group: 234 us of host time per step instead of 29)
        for t in range(self.num_timesteps):
            self.sample_coefs(t)

    def sample_coefs(self, idx, eta=None):
        eta = self.eta if eta is None else eta
        key = (idx, float(eta))
        cache = self.__dict__.setdefault('_ddim_coefs', {})
        if key not in cache:
            cache[key] = kernels.make_ddim_coefs(self.sqrt_recip_alphas_cumprod[idx],
                                                 self.sqrt_recipm1_alphas_cumprod[idx], self.alphas_cumprod[idx],
                                                 self.alphas_cumprod_prev[idx], eta, idx != 0)
        return cache[key]

    def p_sample(self, model, x, t, eta=None, noise=None):
        """reference :479-509.  The noise is drawn whether or not it is used, as there (:493)."""
        idx = int(t)
        if noise is None:
            noise = self._randn(x)
        if self.hip_posterior:
            model_output = self._call_model(model, x, idx)
            x0, sample = kernels.PosteriorStepFn.apply(x, model_output, noise, self.sample_coefs(idx, eta))
            return {'sample': sample, 'pred_xstart': x0}
        # other mean / variance parameterisations: device tensor arithmetic in the reference's op order
        eta = self.eta if eta is None else eta
        out = self.p_mean_variance(model, x, t)
        a = float(np.float32(self.sqrt_recip_alphas_cumprod[idx]))
        b = float(np.float32(self.sqrt_recipm1_alphas_cumprod[idx]))
        eps = (a * x - out['pred_xstart']) / b
        abar = float(np.float32(self.alphas_cumprod[idx]))
        abar_prev = float(np.float32(self.alphas_cumprod_prev[idx]))
        sigma = eta * math.sqrt((1 - abar_prev) / (1 - abar)) * math.sqrt(1 - abar / abar_prev)
        sample = out['pred_xstart'] * math.sqrt(abar_prev) + math.sqrt(1 - abar_prev - sigma ** 2) * eps
        if idx != 0:
            sample = sample + sigma * noise
        return {'sample': sample, 'pred_xstart': out['pred_xstart']}


@register_sampler(name='search_ddpm')
class SearchDDPM(DDPM):
    """Best-of-N per step: every particle is replaced by the one whose proposal is closest to y
    (reference :592-641).  Scoring, argmin and the winner's replication are HIP; the winner's index
    never leaves the device."""

    #: optional hook for multi-GPU runs: callable(costs_local, particles_local, n_out=None) -> n_out copies of the
    #: global winner (default: as many as particles_local)
    global_select = None
    #: After a select every particle is a copy of the winner (reference :633), so from the second step on the loop keeps
    #: ONE state particle: one model evaluation per step instead of N, S1 reads the state once for all N proposals, the
    #: winner is copied out once (dpsx_search_step_one_f32).  Results are those of the replicated form (the per-particle
    #: noise draws, costs and winner indices are the same); False keeps N copies as the reference does.
    single_state = True

    def search_step(self, model, img, idx, measurement, handle, noise=None):
        with torch.no_grad():
            model_out = self._call_model(model, img, idx)
        if noise is None:
            noise = self._randn(img)
        # S1 -> scoring launch -> costs + select -> the winner's replication: one library call, nothing leaves the device
        local = self.global_select is None
        x_next, sample, costs, best, _ = handle.search_step(img, model_out, noise, measurement, self.step_coefs[idx],
                                                            replicate=local)
        self.last_best = best
        if not local:
            return self.global_select(costs, sample), costs
        return x_next, costs

    def search_step_one(self, model, state, n, idx, measurement, handle, noise=None):
        """One step from the single state particle `state` [1,C,H,W] -> (winner [1,C,H,W], costs [n])."""
        with torch.no_grad():
            model_out = self._call_model(model, state, idx)
        if noise is None:
            noise = self._randn(state, shape=(n,) + tuple(state.shape[1:]))
        local = self.global_select is None
        winner, sample, costs, best, _ = handle.search_step_one(state, model_out, noise, measurement,
                                                                self.step_coefs[idx], want_winner=local)
        self.last_best = best
        if not local:
            winner = self.global_select(costs, sample, n_out=1)
        return winner, costs

    def p_sample_loop(self, model, x_start, measurement, measurement_cond_fn, record, save_root, operator,
                      potential_type='min', resample_every_steps=10, rs_temp=0.1, **kwargs):
        img = x_start.detach()
        kernels.require_cuda(img, "x_start")
        if not self.hip_posterior:
            raise NotImplementedError("search_ddpm runs the epsilon / learned_range / clip configuration")
        mask = kwargs.get('mask', None)
        handle = operator.hip_handle_for(mask) if operator.name == 'inpainting' else operator.hip_handle(img)
        self.best_paths, self.best_costs = [], []
        n = img.shape[0]
        state = None                      # the single state particle, once a select has made all particles equal
        for idx in range(self.num_timesteps - 1, -1, -1):
            if state is None:
                img, costs = self.search_step(model, img, idx, measurement, handle)
                if self.single_state and n > 1:
                    state = img[:1]
            else:
                state, costs = self.search_step_one(model, state, n, idx, measurement, handle)
            if kwargs.get('trace', False):
                self.best_costs.append(costs)
        if state is not None:             # the reference returns N copies of the final winner
            return state.repeat(n, 1, 1, 1)
        return img.clone()

    @torch.no_grad()
    def resample_update(self, candidates, denoised_candidates, operator, measurement, resample=True, rs_temp=0.01,
                        prev_costs=None, potential_type='min', steps_done=1, **kwargs):
        """reference :515-587 (defined there, never called by its loops): multinomial resampling on the accumulated
        costs, then the cost update.  The draw is torch.multinomial ([N] weights, RNG parity); the particle gathers
        and the cost update -- ||y - A(x)||_1^2 / CHW per particle combined with the previous costs by
        `potential_type` -- are HIP (dpsx_gather_f32, dpsx_resample_cost_f32: one launch).
        kwargs: `mask` for inpainting (the reference calls operator.forward without it and raises there)."""
        if potential_type not in ('mean', 'min', 'diff', 'curr'):
            raise NotImplementedError
        n = denoised_candidates.shape[0]
        if resample and prev_costs is not None:
            pot = torch.exp(-rs_temp * prev_costs / steps_done) if potential_type == 'mean' \
                else torch.exp(-rs_temp * prev_costs)
            if pot.max() != pot.min():                                                     # :545
                ids = torch.multinomial(pot.cpu(), n, replacement=True).to(pot.device) if self.rng_parity \
                    else torch.multinomial(pot, n, replacement=True)
                self.last_resample_ids = ids
                candidates = kernels.gather(candidates, ids, validate=False)      # ids drawn above over [0, n)
                denoised_candidates = kernels.gather(denoised_candidates, ids, validate=False)
                prev_costs = kernels.gather(prev_costs.reshape(n, 1), ids, validate=False).reshape(n)
        mask = kwargs.get('mask', None)
        handle = operator.hip_handle_for(mask) if operator.name == 'inpainting' \
            else operator.hip_handle(denoised_candidates)
        self.last_curr_costs, net = handle.resample_cost(denoised_candidates, measurement, prev_costs, potential_type)
        return candidates, net


@register_sampler(name='ttc_ddim')
class TTC_DDIM(DDIM):
    """DDIM + multinomial particle resampling every 10 steps (reference :644-707).

    Multi-GPU (SURVEY.md 8e iii): with `global_resample` set (the driver does when WORLD_SIZE > 1) the weights of
    all ranks' particles are all-gathered, every rank draws the same ids from `resample_generator` (a host
    generator seeded identically on all ranks) and fetches its slots of the resampled set -- the same particle
    set one process holding all particles would produce from that generator."""

    global_resample = False
    resample_generator = None
    #: how the drawn particles travel between ranks: "selected" (one all-to-all of the drawn ones, each once per
    #: destination), "all" (all-gather of every state, no host read of device ids) or "auto" (distributed.resample_particles)
    resample_fetch = "auto"

    def p_sample_loop(self, model, x_start, measurement, measurement_cond_fn, record, save_root, **kwargs):
        img = x_start.detach()
        kernels.require_cuda(img, "x_start")
        resample_every_steps, resample_scale = 10, 100
        distance = None
        self.last_resample_ids = None
        # 'ps'-type methods run the three fused launches with the DDIM variant of S1; the rest (e.g. 'mcg', the
        # method whose two return values fit the reference loop's unpacking at :672) go through the per-op path
        plan = self._fusion_plan(measurement_cond_fn, img)
        for idx in range(self.num_timesteps - 1, -1, -1):
            if plan is not None:
                noise = self._randn(img)
                if self.rng_parity:
                    self._randn(measurement, self.parity_measurement_stride)      # q_sample's draw (:668)
                # this loop passes no beta_scale / t to the conditioning method (:678-682): same on both routes
                img, distance = self.dps_step(model, img, idx, measurement, plan[0], plan[1], plan[2], noise=noise,
                                              loop_kw={})
            else:
                img = img.detach().requires_grad_()
                out = self.p_sample(x=img, t=torch.tensor([idx], device=img.device), model=model)
                noisy_measurement = self.q_sample(measurement, t=idx)
                ret = measurement_cond_fn(x_t=out['sample'], measurement=measurement,
                                          noisy_measurement=noisy_measurement, x_prev=img,
                                          x_0_hat=out['pred_xstart'])
                img, distance = ret[0].detach(), ret[1].detach()
            if idx % resample_every_steps == 0:
                img, distance = self._resample(img, distance, resample_scale)
        return img.clone(), distance.clone()

    def _resample(self, img, distance, resample_scale):
        """the resampling block (:685-698): w = exp(-d / 100), multinomial with replacement, skipped when all
        weights are equal.  The draw is torch.multinomial (RNG parity), the particle gather is HIP."""
        n = len(distance)
        if self.global_resample:
            from . import distributed as dd
            img, distance, ids = dd.global_resample(img, distance, resample_scale, self.resample_generator,
                                                     fetch=self.resample_fetch)
            self.last_resample_ids = ids if ids is not None else self.last_resample_ids
            return img, distance
        if n <= 1:
            return img, distance
        weights = torch.exp(-distance / resample_scale)
        if self.rng_parity:                      # replay of the reference's host RNG stream (tests)
            if weights.max() == weights.min():
                return img, distance
            ids = torch.multinomial(weights.cpu(), n, replacement=True).to(img.device)
        else:
            # no host read: when all weights are equal the draw is replaced by the identity on the device
            # ([N]-sized glue; an all-zero weight vector must not reach multinomial)
            flat = weights.max() == weights.min()
            drawn = torch.multinomial(torch.where(flat, torch.ones_like(weights), weights), n, replacement=True)
            ids = torch.where(flat, torch.arange(n, device=img.device), drawn)
        self.last_resample_ids = ids
        return kernels.gather(img, ids, validate=False), kernels.gather(distance.reshape(n, 1), ids, validate=False).reshape(n)
