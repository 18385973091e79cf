"""Conditioning methods -- the reference's registry API
(guided_diffusion/condition_methods.py) over the HIP kernels.

    get_conditioning_method(name, operator, noiser, **params) -> cm     (condition_methods.py:18)
    cm.conditioning(x_prev=, x_t=, x_0_hat=, measurement=, **kw)
        'ps' / 'ps_anneal' / 'mcg' / 'ps+'  -> (x_t_updated, norm, ...)  (:85-106, 206-232)
        'ps_semantic'                       -> (norm_grad, measurement_err, semantic_err)   (:190-195)

Called on tensors that came out of this package's p_sample (PosteriorStepFn) and
operator.forward (OperatorFn), `torch.autograd.grad` walks:  UNet  <-  [HIP S1 VJP]
<-  [HIP A^T]  <-  [HIP norm VJP].  The samplers in gaussian_diffusion.py bypass
this per-op chain with the three fused launches when `fused_spec()` allows it.
"""
import functools
import math
from abc import ABC, abstractmethod

import torch

from .kernels import ResidualNormFn

__CONDITIONING_METHOD__ = {}


def register_conditioning_method(name: str):
    def wrapper(cls):
        if __CONDITIONING_METHOD__.get(name, None):
            raise NameError(f"Name {name} is already registered!")
        __CONDITIONING_METHOD__[name] = cls
        return cls
    return wrapper


def get_conditioning_method(name: str, operator, noiser, **kwargs):
    if __CONDITIONING_METHOD__.get(name, None) is None:
        raise NameError(f"Name {name} is not defined!")
    return __CONDITIONING_METHOD__[name](operator=operator, noiser=noiser, **kwargs)


class ConditioningMethod(ABC):
    #: True when conditioning() returns the gradient (the loop subtracts it), False when it
    #: returns the already-updated x_t
    returns_gradient = False

    def __init__(self, operator, noiser, **kwargs):
        self.operator = operator
        self.noiser = noiser
        self.l1 = kwargs.get('l1', 0.0)

    def project(self, data, noisy_measurement, **kwargs):
        return self.operator.project(data=data, measurement=noisy_measurement, **kwargs)

    def measurement_norm(self, x_0_hat, measurement, **kwargs):
        """||y - A(x0_hat)||_2 per particle (condition_methods.py:36-39), differentiable through HIP VJPs."""
        Ax = self.operator.forward(x_0_hat, **kwargs)
        return ResidualNormFn.apply(Ax, measurement)

    def grad_and_value(self, x_prev, x_0_hat, measurement, **kwargs):
        if self.noiser.__name__ == 'gaussian':
            norm = self.measurement_norm(x_0_hat, measurement, **kwargs)
            norm_power = norm ** 2 if kwargs.get('norm_exp', 1) == 2 else norm      # :41-47
            norm_grad = torch.autograd.grad(outputs=norm_power.sum(), inputs=x_prev)[0]
        elif self.noiser.__name__ == 'poisson':
            # not on the hot path of any shipped config (:50-55); same formula, A through the HIP operator
            Ax = self.operator.forward(x_0_hat, **kwargs)
            difference = measurement - Ax
            norm = (torch.linalg.norm(difference) / measurement.abs()).mean()
            norm_grad = torch.autograd.grad(outputs=norm, inputs=x_prev)[0]
        else:
            raise NotImplementedError
        return norm_grad, norm

    def fused_spec(self, **kwargs):
        """None, or dict(scale, power) when one step of this method is exactly
        x_{t-1} = sample - grad_{x_prev}[ scale * ||y - A(x0_hat)||^power ]  with a Gaussian noiser."""
        return None

    @abstractmethod
    def conditioning(self, x_t, measurement, noisy_measurement=None, **kwargs):
        pass


@register_conditioning_method(name='vanilla')
class Identity(ConditioningMethod):
    def conditioning(self, x_t, *args, **kwargs):
        return x_t


@register_conditioning_method(name='projection')
class Projection(ConditioningMethod):
    def conditioning(self, x_t, noisy_measurement, **kwargs):
        return self.project(data=x_t, noisy_measurement=noisy_measurement)


@register_conditioning_method(name='mcg')
class ManifoldConstraintGradient(ConditioningMethod):
    def __init__(self, operator, noiser, **kwargs):
        super().__init__(operator, noiser)
        self.scale = kwargs.get('scale', 1.0)

    def conditioning(self, x_prev, x_t, x_0_hat, measurement, noisy_measurement, **kwargs):
        norm_grad, norm = self.grad_and_value(x_prev=x_prev, x_0_hat=x_0_hat, measurement=measurement, **kwargs)
        x_t = x_t - norm_grad * self.scale
        x_t = self.project(data=x_t, noisy_measurement=noisy_measurement, **kwargs)
        return x_t, norm


def _drop_loop_kwargs(kwargs):
    """kwargs the loop passes that are not operator.forward arguments"""
    return {k: v for k, v in kwargs.items() if k == 'mask'}


@register_conditioning_method(name='ps')
class PosteriorSampling(ConditioningMethod):
    def __init__(self, operator, noiser, **kwargs):
        super().__init__(operator, noiser)
        self.scale = kwargs.get('scale', 0.3)
        self.operator_name = operator.name

    def fused_spec(self, **kwargs):
        if self.noiser.__name__ != 'gaussian':
            return None
        return {"scale": float(self.scale), "power": 2 if kwargs.get('norm_exp', 1) == 2 else 1}

    def conditioning(self, x_prev, x_t, x_0_hat, measurement, **kwargs):
        norm_exp = kwargs.get('norm_exp', 1)
        norm_grad, norm = self.grad_and_value(x_prev=x_prev, x_0_hat=x_0_hat, measurement=measurement,
                                              norm_exp=norm_exp, **_drop_loop_kwargs(kwargs))
        x_t = x_t - norm_grad * self.scale          # out of place: x_t is an autograd output here
        net_scaling = self.scale / 2 / norm         # condition_methods.py:105
        return x_t, norm.detach(), net_scaling.detach()


@register_conditioning_method(name='ps_semantic')
class PosteriorSamplingSemanticGuid(ConditioningMethod):
    returns_gradient = True

    def __init__(self, operator, noiser, **kwargs):
        super().__init__(operator, noiser)
        self.operator_name = operator.name
        self.scale = kwargs.get('scale', 0.3)
        self.sem_guid_scale = kwargs.get('sem_guid_scale', 0.5)
        self.anneal_factor = kwargs.get('anneal_factor', 1.0)
        self.norm_exp = kwargs.get('norm_exp', 1)
        self.guid_images = kwargs.get('guid_images', None)
        # pluggable face-embedding network: embedder(x0_hat[N,3,H,W]) -> [N, D]; the reference hard-wires
        # facenet_pytorch's InceptionResnetV1('vggface2') on cuda:0 (condition_methods.py:126-141)
        self.embedder = kwargs.get('embedder', None)
        self.guid_image_emb = kwargs.get('guid_image_emb', None)
        if self.guid_images is None or self.sem_guid_scale == 0:
            self.n_guid_images = 1
        else:
            self.n_guid_images = len(self.guid_images)
        if self.sem_guid_scale != 0 and self.embedder is None:
            from facenet_pytorch import MTCNN, InceptionResnetV1  # same dependency as the reference
            device = getattr(operator, 'device', 'cuda:0')
            mtcnn = MTCNN(image_size=256, margin=10, min_face_size=20, device=device)
            self.embedder = InceptionResnetV1(pretrained='vggface2', device=device).eval()
            with torch.no_grad():
                cropped = torch.stack(mtcnn(self.guid_images)).to(device)
                self.guid_image_emb = self.embedder(cropped).unsqueeze(0)

    def semantic_scale(self, t):
        """condition_methods.py:155"""
        return self.sem_guid_scale * (1 + (self.anneal_factor - 1) / (1 + math.exp(-10 * (0.3 - t))))

    def fused_spec(self, **kwargs):
        if self.noiser.__name__ != 'gaussian':
            return None
        # the measurement term is first-power whatever norm_exp says (:177-184)
        spec = {"scale": float(self.scale), "power": 1}
        if self.sem_guid_scale != 0:
            # the semantic term depends on x0_hat only: its cotangent on x0_hat joins coef * A^T r in the fused
            # backward launch (dpsx_step_bwd_extra_f32), the embedder and its VJP stay torch
            spec["semantic"] = functools.partial(self.semantic_cotangent, t=kwargs.get('t', 1))
        return spec

    def semantic_cotangent(self, x_0_hat, t=1):
        """-> (d [sem_guid_scale_t * semantic_loss.sum()] / d x0_hat, sem_guid_norm[N])   (:150-175)"""
        x = x_0_hat.detach().requires_grad_()
        with torch.enable_grad():
            emb = self.embedder(x).unsqueeze(1)
            sem_diff = (emb - self.guid_image_emb).reshape(emb.shape[0], -1)
            sem_guid_norm = torch.norm(sem_diff, dim=-1) / self.n_guid_images
            semantic_loss = sem_guid_norm ** 2 if self.norm_exp == 2 else sem_guid_norm
            (g,) = torch.autograd.grad((self.semantic_scale(t) * semantic_loss).sum(), x)
        return g, sem_guid_norm.detach()

    def measurement_semantic_guidance(self, x_prev, x_0_hat, measurement, **kwargs):
        if self.sem_guid_scale == 0:
            sem_guid_norm = torch.tensor(0.0).to(x_0_hat.device)
            sem_guid_scale_t = 0
        else:
            sem_guid_scale_t = self.semantic_scale(kwargs.get('t', 1))
            emb = self.embedder(x_0_hat).unsqueeze(1)
            sem_diff = (emb - self.guid_image_emb).reshape(emb.shape[0], -1)
            sem_guid_norm = torch.norm(sem_diff, dim=-1) / self.n_guid_images
        semantic_loss = sem_guid_norm ** 2 if self.norm_exp == 2 else sem_guid_norm
        if self.noiser.__name__ != 'gaussian':
            raise NotImplementedError
        measurement_guid_norm = self.measurement_norm(x_0_hat, measurement, **_drop_loop_kwargs(kwargs))
        net_loss = self.scale * measurement_guid_norm + sem_guid_scale_t * semantic_loss
        norm_grad = torch.autograd.grad(outputs=net_loss.sum(), inputs=x_prev)[0]
        return norm_grad, measurement_guid_norm.detach(), sem_guid_norm.detach()

    def conditioning(self, x_prev, x_t, x_0_hat, measurement, **kwargs):
        return self.measurement_semantic_guidance(x_prev=x_prev, x_0_hat=x_0_hat, measurement=measurement, **kwargs)


@register_conditioning_method(name='ps_anneal')
class PosterorSamplingAnnealing(ConditioningMethod):
    def __init__(self, operator, noiser, **kwargs):
        super().__init__(operator, noiser)
        self.noise_sigma = max(noiser.sigma, 0.05)
        self.scale = kwargs.get('scale', 0.3)
        self.operator_name = operator.name

    def net_scaling(self, **kwargs):
        beta_scale = kwargs.get('beta_scale', self.scale)
        anneal = kwargs.get('anneal', 1.0)
        return beta_scale / (anneal * self.noise_sigma ** 2)            # condition_methods.py:209

    def fused_spec(self, **kwargs):
        if self.noiser.__name__ != 'gaussian':
            return None
        return {"scale": float(self.net_scaling(**kwargs)), "power": 2}

    def conditioning(self, x_prev, x_t, x_0_hat, measurement, **kwargs):
        net_scaling = torch.tensor(self.net_scaling(**kwargs)).to(x_t.device)
        grad, norm = self.grad_and_value(x_prev=x_prev, x_0_hat=x_0_hat, measurement=measurement, norm_exp=2,
                                         **_drop_loop_kwargs(kwargs))
        x_t = x_t - net_scaling * grad
        return x_t, norm.detach(), net_scaling


@register_conditioning_method(name='ps+')
class PosteriorSamplingPlus(ConditioningMethod):
    def __init__(self, operator, noiser, **kwargs):
        super().__init__(operator, noiser)
        self.num_sampling = kwargs.get('num_sampling', 5)
        self.scale = kwargs.get('scale', 1.0)

    def conditioning(self, x_prev, x_t, x_0_hat, measurement, **kwargs):
        norm = 0
        for _ in range(self.num_sampling):
            x_0_hat_noise = x_0_hat + 0.05 * torch.rand_like(x_0_hat)
            norm = norm + torch.linalg.norm(measurement - self.operator.forward(x_0_hat_noise)) / self.num_sampling
        norm_grad = torch.autograd.grad(outputs=norm, inputs=x_prev)[0]
        x_t = x_t - norm_grad * self.scale
        return x_t, norm
