"""Mean / variance processor registries (reference: guided_diffusion/posterior_mean_variance.py).

The shipped configuration -- `epsilon` + `learned_range` + clip_denoised
(configs/diffusion_config.yaml) -- never goes through these classes: DDPM.p_sample
hands it to the fused HIP kernel (kernels.PosteriorStepFn).  The other registered
names are kept so that create_sampler() accepts every configuration the reference
accepts; they are plain device-side tensor arithmetic (off the hot path).
"""
from abc import ABC, abstractmethod

import numpy as np
import torch

__MODEL_MEAN_PROCESSOR__ = {}
__MODEL_VAR_PROCESSOR__ = {}


def register_mean_processor(name: str):
    def wrapper(cls):
        if __MODEL_MEAN_PROCESSOR__.get(name, None):
            raise NameError(f"Name {name} is already registerd.")
        __MODEL_MEAN_PROCESSOR__[name] = cls
        return cls
    return wrapper


def get_mean_processor(name: str, **kwargs):
    if __MODEL_MEAN_PROCESSOR__.get(name, None) is None:
        raise NameError(f"Name {name} is not defined.")
    return __MODEL_MEAN_PROCESSOR__[name](**kwargs)


def register_var_processor(name: str):
    def wrapper(cls):
        if __MODEL_VAR_PROCESSOR__.get(name, None):
            raise NameError(f"Name {name} is already registerd.")
        __MODEL_VAR_PROCESSOR__[name] = cls
        return cls
    return wrapper


def get_var_processor(name: str, **kwargs):
    if __MODEL_VAR_PROCESSOR__.get(name, None) is None:
        raise NameError(f"Name {name} is not defined.")
    return __MODEL_VAR_PROCESSOR__[name](**kwargs)


def extract_and_expand(array, time, target):
    """f64 table entry -> fp32 scalar tensor broadcast to `target` (reference :248-252)."""
    value = torch.as_tensor(np.asarray(array)[int(time)], dtype=torch.float32, device=target.device)
    return value.expand_as(target)


def _posterior_tables(betas):
    alphas = 1.0 - betas
    abar = np.cumprod(alphas, axis=0)
    abar_prev = np.append(1.0, abar[:-1])
    coef1 = betas * np.sqrt(abar_prev) / (1.0 - abar)
    coef2 = (1.0 - abar_prev) * np.sqrt(alphas) / (1.0 - abar)
    var = betas * (1.0 - abar_prev) / (1.0 - abar)
    return abar, coef1, coef2, var


class MeanProcessor(ABC):
    @abstractmethod
    def __init__(self, betas, dynamic_threshold, clip_denoised):
        self.dynamic_threshold = dynamic_threshold
        self.clip_denoised = clip_denoised
        self.abar, self.coef1, self.coef2, _ = _posterior_tables(betas)

    @abstractmethod
    def get_mean_and_xstart(self, x, t, model_output):
        pass

    def process_xstart(self, x):
        if self.dynamic_threshold:       # util/img_utils.py:247-249
            x = torch.clip(x * torch.quantile(x.abs(), 0.95), -1.0, 1.0)
        if self.clip_denoised:
            x = x.clamp(-1, 1)
        return x

    def q_posterior_mean(self, x_start, x_t, t):
        return extract_and_expand(self.coef1, t, x_start) * x_start + extract_and_expand(self.coef2, t, x_t) * x_t


@register_mean_processor(name='previous_x')
class PreviousXMeanProcessor(MeanProcessor):
    def __init__(self, betas, dynamic_threshold, clip_denoised):
        super().__init__(betas, dynamic_threshold, clip_denoised)

    def get_mean_and_xstart(self, x, t, model_output):
        c1 = extract_and_expand(1.0 / self.coef1, t, x)
        c2 = extract_and_expand(self.coef2 / self.coef1, t, x)
        return model_output, self.process_xstart(c1 * model_output - c2 * x)


@register_mean_processor(name='start_x')
class StartXMeanProcessor(MeanProcessor):
    def __init__(self, betas, dynamic_threshold, clip_denoised):
        super().__init__(betas, dynamic_threshold, clip_denoised)

    def get_mean_and_xstart(self, x, t, model_output):
        pred_xstart = self.process_xstart(model_output)
        return self.q_posterior_mean(pred_xstart, x, t), pred_xstart


@register_mean_processor(name='epsilon')
class EpsilonXMeanProcessor(MeanProcessor):
    def __init__(self, betas, dynamic_threshold, clip_denoised):
        super().__init__(betas, dynamic_threshold, clip_denoised)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.abar)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.abar - 1)

    def predict_xstart(self, x_t, t, eps):
        return extract_and_expand(self.sqrt_recip_alphas_cumprod, t, x_t) * x_t - \
            extract_and_expand(self.sqrt_recipm1_alphas_cumprod, t, eps) * eps

    def get_mean_and_xstart(self, x, t, model_output):
        pred_xstart = self.process_xstart(self.predict_xstart(x, t, model_output))
        return self.q_posterior_mean(pred_xstart, x, t), pred_xstart


class VarianceProcessor(ABC):
    @abstractmethod
    def __init__(self, betas):
        pass

    @abstractmethod
    def get_variance(self, x, t):
        pass


@register_var_processor(name='fixed_small')
class FixedSmallVarianceProcessor(VarianceProcessor):
    def __init__(self, betas):
        self.posterior_variance = _posterior_tables(betas)[3]

    def get_variance(self, x, t):
        v = self.posterior_variance
        return extract_and_expand(v, t, x), extract_and_expand(np.log(v), t, x)


@register_var_processor(name='fixed_large')
class FixedLargeVarianceProcessor(VarianceProcessor):
    def __init__(self, betas):
        self.betas = betas
        self.posterior_variance = _posterior_tables(betas)[3]

    def get_variance(self, x, t):
        v = np.append(self.posterior_variance[1], self.betas[1:])
        return extract_and_expand(v, t, x), extract_and_expand(np.log(v), t, x)


@register_var_processor(name='learned')
class LearnedVarianceProcessor(VarianceProcessor):
    def __init__(self, betas):
        pass

    def get_variance(self, x, t):
        return torch.exp(x), x


@register_var_processor(name='learned_range')
class LearnedRangeVarianceProcessor(VarianceProcessor):
    def __init__(self, betas):
        self.betas = betas
        var = _posterior_tables(betas)[3]
        self.posterior_log_variance_clipped = np.log(np.append(var[1], var[1:]))

    def get_variance(self, x, t):
        min_log = extract_and_expand(self.posterior_log_variance_clipped, t, x)
        max_log = extract_and_expand(np.log(self.betas), t, x)
        frac = (x + 1.0) / 2.0
        log_var = frac * max_log + (1 - frac) * min_log
        return torch.exp(log_var), log_var
