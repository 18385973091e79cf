"""Measurement operators A and noise models -- the reference's registry API
(guided_diffusion/measurements.py) over the HIP kernels.

    get_operator(name, device=..., **yaml_kwargs) -> op     (measurements.py:29)
    op.forward(data, **kwargs) -> Tensor                    (:84, 108, 142, 158, 186)
    get_noise(name, **kwargs) -> callable noiser            (:235)

Same names, constructor arguments and error behaviour as the reference; forward()
returns a tensor that carries the operator's exact HIP adjoint as its autograd
VJP, so `torch.autograd.grad(norm, x_prev)` in a conditioning method works
unchanged.  Operators run on an MI355X only (no CPU fallback).
"""
from abc import ABC, abstractmethod
from functools import partial

import numpy as np
import torch
from torch.nn import functional as F

from . import host_tables
from .kernels import OperatorFn, OpHandle

__OPERATOR__ = {}


def register_operator(name: str):
    def wrapper(cls):
        if __OPERATOR__.get(name, None):
            raise NameError(f"Name {name} is already registered!")
        __OPERATOR__[name] = cls
        return cls
    return wrapper


def get_operator(name: str, **kwargs):
    if __OPERATOR__.get(name, None) is None:
        raise NameError(f"Name {name} is not defined.")
    return __OPERATOR__[name](**kwargs)


class _HipOperator:
    """Shared plumbing: lazily built dpsx_op handle, autograd-aware apply."""
    _handle = None

    def _build_handle(self, data):
        raise NotImplementedError

    def hip_handle(self, data=None):
        if self._handle is None:
            self._handle = self._build_handle(data)
        return self._handle

    def new_hip_handle(self, data=None, **kwargs):
        """a fresh, uncached handle (own constant tables, workspace, tail counters): one per HIP stream that drives this
        operator concurrently (kernels.ParticleGroups)"""
        return self._build_handle(data)

    def _apply(self, data):
        return OperatorFn.apply(data, self.hip_handle(data))


class LinearOperator(ABC, _HipOperator):
    def __init__(self):
        self.name = 'linear'

    @abstractmethod
    def forward(self, data, **kwargs):
        pass

    @abstractmethod
    def transpose(self, data, **kwargs):
        pass

    def ortho_project(self, data, **kwargs):
        # (I - A^T A) x          measurements.py:48-50
        return data - self.transpose(self.forward(data, **kwargs), **kwargs)

    def project(self, data, measurement, **kwargs):
        # (I - A^T A) y - A x     measurements.py:52-54
        return self.ortho_project(measurement, **kwargs) - self.forward(data, **kwargs)


@register_operator(name='noise')
class DenoiseOperator(LinearOperator):
    def __init__(self, device):
        self.name = 'noise'
        self.device = device

    def _build_handle(self, data):
        return OpHandle.identity(self.device)

    def forward(self, data, **kwargs):
        return data

    def transpose(self, data, **kwargs):
        return data

    def ortho_project(self, data, **kwargs):
        return data

    def project(self, data, **kwargs):
        return data


@register_operator(name='super_resolution')
class SuperResolutionOperator(LinearOperator):
    def __init__(self, in_shape, scale_factor, device):
        self.name = 'super_resolution'
        self.device = device
        self.in_shape = tuple(int(v) for v in in_shape)
        self.scale_factor = scale_factor
        self.up_sample = partial(F.interpolate, scale_factor=scale_factor)   # measurements.py:80
        h, w = self.in_shape[-2:]
        # Resizer(in_shape, 1/scale_factor): the same scale on H and W (util/resizer.py:76-102)
        self.w_h, self.i_h = host_tables.resizer_axis(h, 1.0 / scale_factor)
        self.w_w, self.i_w = host_tables.resizer_axis(w, 1.0 / scale_factor)

    def _build_handle(self, data):
        h, w = self.in_shape[-2:]
        return OpHandle.resize(h, w, self.w_h, self.i_h, self.w_w, self.i_w, self.device)

    def forward(self, data, **kwargs):
        if tuple(data.shape[-2:]) != self.in_shape[-2:]:
            raise ValueError(f"super_resolution was built for {self.in_shape[-2:]}, got {tuple(data.shape[-2:])}")
        return self._apply(data)

    def transpose(self, data, **kwargs):
        # nearest-neighbour upsample, NOT the adjoint (measurements.py:87-88); only projection/mcg use it
        return self.up_sample(data)

    def project(self, data, measurement, **kwargs):
        return data - self.transpose(self.forward(data)) + self.transpose(measurement)


class _BlurOperator(LinearOperator):
    """ReflectionPad2d(k//2) + depthwise conv with one k x k kernel (util/img_utils.py:268-308)."""

    def _set_weights(self, k2d):
        self._weights = np.ascontiguousarray(np.asarray(k2d, dtype=np.float32))
        self._handle = None

    def _build_handle(self, data):
        return OpHandle.blur(self._weights, self.device)

    def forward(self, data, **kwargs):
        return self._apply(data)

    def transpose(self, data, **kwargs):
        return data                                                    # measurements.py:112-113, 145-146


@register_operator(name='motion_blur')
class MotionBlurOperator(_BlurOperator):
    def __init__(self, kernel_size, intensity, device):
        self.name = 'motion_blur'
        self.device = device
        self.kernel_size = kernel_size
        try:   # the reference's generator, when installed (measurements.py:8, 104)
            from motionblur.motionblur import Kernel
            # the reference draws twice from numpy's global RNG: once inside Blurkernel
            # (util/img_utils.py:295), once here; keep the stream position identical
            Kernel(size=(kernel_size, kernel_size), intensity=intensity)
            km = Kernel(size=(kernel_size, kernel_size), intensity=intensity).kernelMatrix
        except ImportError:
            host_tables.random_motion_kernel(kernel_size, intensity)
            km = host_tables.random_motion_kernel(kernel_size, intensity)
        self.kernel_matrix = np.asarray(km, dtype=np.float64)
        self._set_weights(self.kernel_matrix)

    def get_kernel(self):
        k = torch.from_numpy(self.kernel_matrix).type(torch.float32).to(self.device)
        return k.view(1, 1, self.kernel_size, self.kernel_size)

    def set_kernel(self, kernel):
        """:kernel: np.array ksize x ksize; stored transposed as the reference does (measurements.py:125)"""
        self._set_weights(np.asarray(kernel, dtype=np.float32).T)


@register_operator(name='gaussian_blur')
class GaussialBlurOperator(_BlurOperator):
    def __init__(self, kernel_size, intensity, device):
        self.name = 'gaussian_blur'
        self.device = device
        self.kernel_size = kernel_size
        self.kernel = torch.from_numpy(host_tables.gaussian_blur_kernel(kernel_size, intensity))  # float64
        self._set_weights(self.kernel.numpy())

    def get_kernel(self):
        return self.kernel.view(1, 1, self.kernel_size, self.kernel_size)


@register_operator(name='inpainting')
class InpaintingOperator(LinearOperator):
    '''This operator get pre-defined mask and return masked image.'''

    def __init__(self, device):
        self.device = device
        self.name = 'inpainting'
        self._mask_key = None

    def hip_handle_for(self, mask):
        key = (mask.data_ptr(), tuple(mask.shape), mask._version)
        if self._mask_key != key:
            self._handle = OpHandle.mask(mask, self.device)
            self._mask_key = key
        return self._handle

    def new_hip_handle(self, data=None, **kwargs):
        if kwargs.get('mask', None) is None:
            raise ValueError("Require mask")
        return OpHandle.mask(kwargs['mask'], self.device)

    def forward(self, data, **kwargs):
        mask = kwargs.get('mask', None)
        if mask is None:
            raise ValueError("Require mask")                           # measurements.py:159-162
        return OperatorFn.apply(data, self.hip_handle_for(mask))

    def transpose(self, data, **kwargs):
        return data

    def ortho_project(self, data, **kwargs):
        return data - self.forward(data, **kwargs)


class NonLinearOperator(ABC, _HipOperator):
    @abstractmethod
    def forward(self, data, **kwargs):
        pass

    def project(self, data, measurement, **kwargs):
        return data + measurement - self.forward(data)


@register_operator(name='phase_retrieval')
class PhaseRetrievalOperator(NonLinearOperator):
    def __init__(self, oversample, device):
        self.pad = int((oversample / 8.0) * 256)                       # measurements.py:182
        self.device = device
        self.name = 'phase_retrieval'
        self._key = None

    def hip_handle(self, data=None):
        n, c, h, w = data.shape
        if h != w:
            raise ValueError("phase_retrieval expects square images")
        if self._handle is None or self._key != h:
            self._handle = OpHandle.phase(h, self.pad, max(n * c, 1), self.device)
            self._key = h
        return self._handle

    def new_hip_handle(self, data=None, **kwargs):
        n, c, h, w = data.shape
        if h != w:
            raise ValueError("phase_retrieval expects square images")
        return OpHandle.phase(h, self.pad, max(n * c, 1), self.device)

    def forward(self, data, **kwargs):
        return self._apply(data)


@register_operator(name='nonlinear_blur')
class NonlinearBlurOperator(NonLinearOperator):
    def __init__(self, opt_yml_path, device):
        # needs the un-vendored `bkse` KernelWizard network and its weights (measurements.py:198-211)
        raise NotImplementedError("nonlinear_blur needs the external bkse model; out of scope of the HIP hot path")

    def forward(self, data, **kwargs):
        raise NotImplementedError


# =============
# Noise classes
# =============

__NOISE__ = {}


def register_noise(name: str):
    def wrapper(cls):
        if __NOISE__.get(name, None):
            raise NameError(f"Name {name} is already defined!")
        __NOISE__[name] = cls
        return cls
    return wrapper


def get_noise(name: str, **kwargs):
    if __NOISE__.get(name, None) is None:
        raise NameError(f"Name {name} is not defined.")
    noiser = __NOISE__[name](**kwargs)
    noiser.__name__ = name
    return noiser


class Noise(ABC):
    def __call__(self, data):
        return self.forward(data)

    @abstractmethod
    def forward(self, data):
        pass


@register_noise(name='clean')
class Clean(Noise):
    def forward(self, data):
        return data


@register_noise(name='gaussian')
class GaussianNoise(Noise):
    def __init__(self, sigma):
        self.sigma = sigma

    def forward(self, data):
        # once per image, outside the per-step hot path (sample_condition_batched_ttc.py:165)
        return data + torch.randn_like(data, device=data.device) * self.sigma


@register_noise(name='poisson')
class PoissonNoise(Noise):
    def __init__(self, rate):
        self.rate = rate

    def forward(self, data):
        # host numpy RNG exactly as the reference (measurements.py:277-285); once per image
        x = ((data + 1.0) / 2.0).clamp(0, 1)
        device = x.device
        x = x.detach().cpu()
        x = torch.from_numpy(np.random.poisson(x * 255.0 * self.rate) / 255.0 / self.rate)
        return (x * 2.0 - 1.0).clamp(-1, 1).to(device)
