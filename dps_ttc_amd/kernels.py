"""torch-tensor front end of the C ABI (include/dpsx.h) + the autograd glue.

Every function here enqueues hand-written HIP kernels on torch's current
stream; tensors are only the owners of device memory.  Nothing in this module
computes with torch ops.
"""
import ctypes
from ctypes import byref, c_int64, c_void_p

import numpy as np
import os

import torch

from . import _lib
from ._lib import Coefs, check, f32c, lib, ptr, require_cuda, stream_of


def make_coefs(a, b, c1, c2, min_log, max_log, add_noise):
    return Coefs(float(a), float(b), float(c1), float(c2), float(min_log), float(max_log), int(bool(add_noise)))


def make_ddim_coefs(a, b, abar, abar_prev, eta, add_noise):
    """struct dpsx_coefs of one DDIM step (reference gaussian_diffusion.py:481-509).  The reference evaluates
    sigma and the two square roots as fp32 tensor arithmetic on .float()-cast table entries, one rounding per
    op; numpy float32 scalars reproduce that bit for bit."""
    one, ab, abp = np.float32(1.0), np.float32(abar), np.float32(abar_prev)
    sigma = np.float32(eta) * np.sqrt((one - abp) / (one - ab)) * np.sqrt(one - ab / abp)
    return Coefs(float(np.float32(a)), float(np.float32(b)), float(np.sqrt(abp)),
                 float(np.sqrt(one - abp - sigma ** 2)), float(sigma), 0.0, 2 | int(bool(add_noise)))


# ------------------------------------------------------------------ S1
def posterior_fwd(x_t, model_out, noise, coefs, want_inside=False, want_x0=True):
    """p_mean_variance + DDPM.p_sample (gaussian_diffusion.py:308-330, 466-476) -> (x0_hat, sample[, inside]).
    want_x0=False skips the x0_hat store (search_ddpm only consumes the sample): x0_hat is then None."""
    x_t, model_out = f32c(x_t, "x_t"), f32c(model_out, "model_out")
    noise = None if noise is None else f32c(noise, "noise")
    n, chw = x_t.shape[0], x_t[0].numel() if x_t.shape[0] else 0
    if model_out.shape[0] != n or (n and model_out[0].numel() != 2 * chw):
        raise ValueError(f"model_out {tuple(model_out.shape)} does not hold 2x the channels of x {tuple(x_t.shape)}")
    x0, sample = (torch.empty_like(x_t) if want_x0 else None), torch.empty_like(x_t)
    inside = torch.empty(x_t.shape, dtype=torch.uint8, device=x_t.device) if want_inside else None
    check(lib().dpsx_posterior_fwd_f32(ptr(x_t), ptr(model_out), ptr(noise), ptr(x0), ptr(sample), ptr(inside),
                                       n, chw, byref(coefs), stream_of(x_t)), "dpsx_posterior_fwd_f32")
    return (x0, sample, inside) if want_inside else (x0, sample)


def posterior_bwd(g_x0, g_sample, x_t, model_out, noise, coefs):
    x_t, model_out = f32c(x_t), f32c(model_out)
    g_x0 = None if g_x0 is None else f32c(g_x0)
    g_sample = None if g_sample is None else f32c(g_sample)
    noise = None if noise is None else f32c(noise)
    n, chw = x_t.shape[0], x_t[0].numel() if x_t.shape[0] else 0
    g_x, g_mo = torch.empty_like(x_t), torch.empty_like(model_out)
    check(lib().dpsx_posterior_bwd_f32(ptr(g_x0), ptr(g_sample), ptr(x_t), ptr(model_out), ptr(noise), ptr(g_x),
                                       ptr(g_mo), n, chw, byref(coefs), stream_of(x_t)), "dpsx_posterior_bwd_f32")
    return g_x, g_mo


class PosteriorStepFn(torch.autograd.Function):
    """(x_t, model_out) -> (x0_hat, sample) with the HIP VJP, so torch.autograd only sees UNet -> [HIP tail]."""

    @staticmethod
    def forward(ctx, x_t, model_out, noise, coefs):
        x0, sample = posterior_fwd(x_t, model_out, noise, coefs)
        ctx.save_for_backward(x_t, model_out, noise if noise is not None else x_t.new_empty(0))
        ctx.coefs = coefs
        return x0, sample

    @staticmethod
    def backward(ctx, g_x0, g_sample):
        x_t, model_out, noise = ctx.saved_tensors
        g_x, g_mo = posterior_bwd(g_x0, g_sample, x_t, model_out, noise if noise.numel() else None, ctx.coefs)
        return g_x, g_mo, None, None


# ------------------------------------------------------------------ operator handles
class OpHandle:
    """Owns one dpsx_op (constant tables on the device) and its scratch workspace."""

    def __init__(self, handle, device, keep=()):
        self._h = handle
        self.device = device
        self._keep = keep          # tensors the op borrows (mask)
        self._ws = None
        self.kind = lib().dpsx_op_kind(self._h)

    @classmethod
    def blur(cls, kernel2d, device, force_taps=False):
        k = np.ascontiguousarray(np.asarray(kernel2d, dtype=np.float32))
        if k.ndim != 2 or k.shape[0] != k.shape[1]:
            raise ValueError("blur kernel must be square")
        _cuda_device(device)
        h = c_void_p()
        check(lib().dpsx_op_create_blur(k.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), k.shape[0],
                                        _lib.BLUR_FORCE_TAPS if force_taps else _lib.BLUR_AUTO, byref(h)),
              "dpsx_op_create_blur")
        return cls(h, device)

    @classmethod
    def resize(cls, in_h, in_w, w_h, i_h, w_w, i_w, device):
        w_h = np.ascontiguousarray(w_h, dtype=np.float32)
        w_w = np.ascontiguousarray(w_w, dtype=np.float32)
        i_h = np.ascontiguousarray(i_h, dtype=np.int64)
        i_w = np.ascontiguousarray(i_w, dtype=np.int64)
        _cuda_device(device)
        h = c_void_p()
        fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64)
        check(lib().dpsx_op_create_resize(in_h, in_w, w_h.ctypes.data_as(fp), i_h.ctypes.data_as(ip),
                                          w_h.shape[0], w_h.shape[1], w_w.ctypes.data_as(fp),
                                          i_w.ctypes.data_as(ip), w_w.shape[0], w_w.shape[1], byref(h)),
              "dpsx_op_create_resize")
        return cls(h, device)

    @classmethod
    def mask(cls, mask, device):
        m = f32c(mask.to(device), "mask")
        hh, ww = m.shape[-2:]
        if m.numel() != hh * ww:
            raise ValueError("mask must be [1,1,H,W]")
        h = c_void_p()
        check(lib().dpsx_op_create_mask(ptr(m), hh, ww, byref(h)), "dpsx_op_create_mask")
        return cls(h, device, keep=(m,))

    @classmethod
    def identity(cls, device):
        _cuda_device(device)
        h = c_void_p()
        check(lib().dpsx_op_create_identity(byref(h)), "dpsx_op_create_identity")
        return cls(h, device)

    @classmethod
    def phase(cls, side, pad, max_planes, device):
        _cuda_device(device)
        h = c_void_p()
        check(lib().dpsx_op_create_phase(side, pad, max_planes, byref(h)), "dpsx_op_create_phase")
        obj = cls(h, device)
        # the hand-written spectral step (256 + 2 x 64 = 384 points) consumes x0_hat inside its first pass
        obj.spectral = (side, pad) == (256, 64) and "DPSX_PHASE_LIBRARY_FFT" not in os.environ
        return obj

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        try:
            if h and _lib._lib is not None:
                _lib._lib.dpsx_op_destroy(h)
        except (AttributeError, TypeError):     # interpreter shutdown: module globals are already gone
            pass

    # -- geometry / scratch
    def out_hw(self, h, w):
        oh, ow = c_int64(), c_int64()
        check(lib().dpsx_op_out_shape(self._h, h, w, byref(oh), byref(ow)), "dpsx_op_out_shape")
        return oh.value, ow.value

    def workspace(self, n, c, h, w, device):
        need = lib().dpsx_op_workspace_bytes(self._h, n, c, h, w)
        if need < 0:
            check(int(need), "dpsx_op_workspace_bytes")
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(max(int(need), 256), dtype=torch.uint8, device=device)
        return self._ws

    # -- A and A^T
    def forward(self, x):
        x = _nchw(f32c(x, "operator input"))
        n, c, h, w = x.shape
        oh, ow = self.out_hw(h, w)
        y = torch.empty((n, c, oh, ow), dtype=torch.float32, device=x.device)
        ws = self.workspace(n, c, h, w, x.device)
        check(lib().dpsx_op_forward_f32(self._h, ptr(x), ptr(y), n, c, h, w, ptr(ws), ws.numel(), stream_of(x)),
              "dpsx_op_forward_f32")
        return y

    def adjoint(self, u, x=None, in_hw=None):
        u = _nchw(f32c(u, "cotangent"))
        n, c = u.shape[:2]
        h, w = in_hw if in_hw is not None else x.shape[-2:]
        x = None if x is None else f32c(x)
        g = torch.empty((n, c, h, w), dtype=torch.float32, device=u.device)
        ws = self.workspace(n, c, h, w, u.device)
        check(lib().dpsx_op_adjoint_f32(self._h, ptr(u), ptr(x), ptr(g), n, c, h, w, ptr(ws), ws.numel(),
                                        stream_of(u)), "dpsx_op_adjoint_f32")
        return g

    def score(self, x, y):
        """costs[p] = ||y - A(x_p)||_2 (gaussian_diffusion.py:626-630) without materialising A x."""
        x, y = _nchw(f32c(x)), f32c(y)
        n, c, h, w = x.shape
        costs = torch.empty(n, dtype=torch.float32, device=x.device)
        ws = self.workspace(n, c, h, w, x.device)
        check(lib().dpsx_score_f32(self._h, ptr(x), ptr(y), y.shape[0], ptr(costs), n, c, h, w, ptr(ws),
                                   ws.numel(), stream_of(x)), "dpsx_score_f32")
        return costs

    def score_argmin(self, x, y):
        """-> (costs [N], best int64 scalar, costs[best] [1]): scoring launch + one small launch that finishes the
        per-particle norms and the torch.argmin-order select (gaussian_diffusion.py:626-632); all on the device."""
        x, y = _nchw(f32c(x)), f32c(y)
        n, c, h, w = x.shape
        if n == 0:
            raise ValueError("best-of-N over an empty particle set")
        costs = torch.empty(n, dtype=torch.float32, device=x.device)
        best = torch.empty((), dtype=torch.int64, device=x.device)
        val = torch.empty(1, dtype=torch.float32, device=x.device)
        ws = self.workspace(n, c, h, w, x.device)
        check(lib().dpsx_score_argmin_f32(self._h, ptr(x), ptr(y), y.shape[0], ptr(costs), ptr(best), ptr(val),
                                          n, c, h, w, ptr(ws), ws.numel(), stream_of(x)), "dpsx_score_argmin_f32")
        return costs, best, val

    def search_step(self, x_t, model_out, noise, y, coefs, replicate=True):
        """One search_ddpm step (gaussian_diffusion.py:618-633): S1, costs of the proposals, select and -- with
        replicate=True -- the winner copied over all particles.  -> (x_next or None, sample, costs, best, costs[best]);
        one library call, nothing leaves the device."""
        x_t, model_out, y = _nchw(f32c(x_t, "x_t")), f32c(model_out, "model_out"), f32c(y, "measurement")
        noise = None if noise is None else f32c(noise, "noise")
        n, c, h, w = x_t.shape
        if n == 0:
            raise ValueError("best-of-N over an empty particle set")
        if model_out.shape[0] != n or model_out[0].numel() != 2 * c * h * w:
            raise ValueError(f"model_out {tuple(model_out.shape)} does not hold 2x the channels of x {tuple(x_t.shape)}")
        sample = torch.empty_like(x_t)
        x_next = torch.empty_like(x_t) if replicate else None
        costs = torch.empty(n, dtype=torch.float32, device=x_t.device)
        best = torch.empty((), dtype=torch.int64, device=x_t.device)
        val = torch.empty(1, dtype=torch.float32, device=x_t.device)
        ws = self.workspace(n, c, h, w, x_t.device)
        check(lib().dpsx_search_step_f32(self._h, ptr(x_t), ptr(model_out), ptr(noise), ptr(y), y.shape[0], ptr(sample),
                                         ptr(costs), ptr(best), ptr(val), ptr(x_next), n, c, h, w, byref(coefs), ptr(ws),
                                         ws.numel(), stream_of(x_t)), "dpsx_search_step_f32")
        return x_next, sample, costs, best, val

    def search_step_one(self, x_one, model_out_one, noise, y, coefs, want_winner=True):
        """The same step from ONE state particle (after a select all particles are copies of the winner): x_one
        [1,C,H,W], model_out_one [1,2C,H,W], noise [N,C,H,W] -> (winner [1,C,H,W] or None, sample [N,...], costs,
        best, costs[best]).  Bit-identical to search_step on N copies of the state; one model evaluation per step."""
        x_one, model_out_one, y = _nchw(f32c(x_one, "x_t")), f32c(model_out_one, "model_out"), f32c(y, "measurement")
        noise = _nchw(f32c(noise, "noise"))
        n, c, h, w = noise.shape
        if n == 0:
            raise ValueError("best-of-N over an empty particle set")
        if x_one.shape != (1, c, h, w) or model_out_one.shape[0] != 1 or model_out_one[0].numel() != 2 * c * h * w:
            raise ValueError(f"one state particle expected: x {tuple(x_one.shape)}, model_out {tuple(model_out_one.shape)}, "
                             f"noise {tuple(noise.shape)}")
        sample = torch.empty_like(noise)
        winner = torch.empty_like(x_one) if want_winner else None
        costs = torch.empty(n, dtype=torch.float32, device=noise.device)
        best = torch.empty((), dtype=torch.int64, device=noise.device)
        val = torch.empty(1, dtype=torch.float32, device=noise.device)
        ws = self.workspace(n, c, h, w, noise.device)
        check(lib().dpsx_search_step_one_f32(self._h, ptr(x_one), ptr(model_out_one), ptr(noise), ptr(y), y.shape[0],
                                             ptr(sample), ptr(costs), ptr(best), ptr(val), ptr(winner), n, c, h, w,
                                             byref(coefs), ptr(ws), ws.numel(), stream_of(noise)),
              "dpsx_search_step_one_f32")
        return winner, sample, costs, best, val

    def resample_cost(self, x, y, prev_costs=None, potential_type='min'):
        """SearchDDPM.resample_update's cost update (gaussian_diffusion.py:556-585) in one launch:
        curr[p] = ||y - A(x_p)||_1^2 / (C H W), net = combine(curr, prev_costs) -> (curr, net)."""
        if potential_type not in _lib.POTENTIALS:
            raise NotImplementedError(potential_type)
        x, y = _nchw(f32c(x)), f32c(y)
        n, c, h, w = x.shape
        prev = None if prev_costs is None else f32c(prev_costs.reshape(-1), "prev_costs")
        if prev is not None and prev.numel() != n:
            raise ValueError("prev_costs must hold one cost per particle")
        curr = torch.empty(n, dtype=torch.float32, device=x.device)
        net = torch.empty(n, dtype=torch.float32, device=x.device)
        ws = self.workspace(n, c, h, w, x.device)
        check(lib().dpsx_resample_cost_f32(self._h, ptr(x), ptr(y), y.shape[0], ptr(prev),
                                           _lib.POTENTIALS[potential_type], ptr(curr), ptr(net), n, c, h, w,
                                           ptr(ws), ws.numel(), stream_of(x)), "dpsx_resample_cost_f32")
        return curr, net


def _cuda_device(device):
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError(f"dps_ttc_amd operators need an MI355X device, got {dev} (no CPU fallback by design)")
    torch.cuda.set_device(dev)
    torch.cuda.current_stream(dev)   # make sure the HIP context exists before libdpsx allocates


def _nchw(t):
    if t.dim() != 4:
        raise ValueError(f"expected an [N,C,H,W] tensor, got {tuple(t.shape)}")
    return t


class OperatorFn(torch.autograd.Function):
    """operator.forward with its exact HIP adjoint as the VJP."""

    @staticmethod
    def forward(ctx, x, handle):
        ctx.handle = handle
        ctx.in_hw = tuple(x.shape[-2:])
        if handle.kind == _lib.KIND_PHASE:
            ctx.save_for_backward(x)
        return handle.forward(x)

    @staticmethod
    def backward(ctx, u):
        x = ctx.saved_tensors[0] if ctx.handle.kind == _lib.KIND_PHASE else None
        return ctx.handle.adjoint(u, x=x, in_hw=ctx.in_hw), None


# ------------------------------------------------------------------ residual norm
def residual_norm(y, ax, want_residual=True):
    y, ax = f32c(y, "measurement"), f32c(ax, "A x")
    n = ax.shape[0]
    m = ax[0].numel() if n else 0
    y_n = y.shape[0]
    if y_n not in (1, n) or (n and y[0].numel() != m):
        raise ValueError(f"measurement {tuple(y.shape)} does not broadcast against {tuple(ax.shape)}")
    r = torch.empty_like(ax) if want_residual else None
    norm = torch.empty(n, dtype=torch.float32, device=ax.device)
    ws = torch.empty(max(n * 256, 1), dtype=torch.float32, device=ax.device)
    check(lib().dpsx_residual_norm_f32(ptr(y), y_n, ptr(ax), ptr(r), ptr(norm), n, m, ptr(ws), ws.numel() * 4,
                                       stream_of(ax)), "dpsx_residual_norm_f32")
    return r, norm


def norm_bwd(r, norm, g_norm, power=1):
    r, norm, g_norm = f32c(r), f32c(norm), f32c(g_norm)
    n = r.shape[0]
    g = torch.empty_like(r)
    check(lib().dpsx_norm_bwd_f32(ptr(r), ptr(norm), ptr(g_norm), power, ptr(g), n, r[0].numel() if n else 0,
                                  stream_of(r)), "dpsx_norm_bwd_f32")
    return g


class ResidualNormFn(torch.autograd.Function):
    """norm[p] = ||y - ax_p||_2 per particle  (condition_methods.py:37-39); VJP w.r.t. ax only."""

    @staticmethod
    def forward(ctx, ax, y):
        r, norm = residual_norm(y, ax)
        ctx.save_for_backward(r, norm)
        return norm

    @staticmethod
    def backward(ctx, g_norm):
        r, norm = ctx.saved_tensors
        return norm_bwd(r, norm, g_norm, 1), None


# ------------------------------------------------------------------ update / select
def update(sample, g_a, g_b=None):
    sample, g_a = f32c(sample), f32c(g_a)
    g_b = None if g_b is None else f32c(g_b)
    out = torch.empty_like(sample)
    check(lib().dpsx_update_f32(ptr(sample), ptr(g_a), ptr(g_b), ptr(out), sample.numel(), stream_of(sample)),
          "dpsx_update_f32")
    return out


def argmin(v, want_value=False):
    """torch.argmin semantics on the device, result stays on the device (int64 scalar tensor).
    want_value: also return v[argmin] as a [1] fp32 device tensor (no host index, no sync)."""
    v = f32c(v.reshape(-1), "scores")
    if v.numel() == 0:
        raise ValueError("argmin of an empty score vector")
    out = torch.empty((), dtype=torch.int64, device=v.device)
    val = torch.empty(1, dtype=torch.float32, device=v.device) if want_value else None
    check(lib().dpsx_argmin_f32(ptr(v), v.numel(), ptr(out), ptr(val), stream_of(v)), "dpsx_argmin_f32")
    return (out, val) if want_value else out


def gather(src, ids, validate=True):
    """src[ids] for an [N, ...] fp32 tensor and int64 ids (gaussian_diffusion.py:697).  validate=True (the default for
    ids that arrive through the public API) adds torch's IndexError check -- two host reads; the package's own loops pass
    validate=False for ids they drew themselves (torch.multinomial over [0, N)): no host sync, and the kernel never reads
    out of bounds anyway (a bad id yields a NaN particle)."""
    src = f32c(src)
    ids = ids.to(device=src.device, dtype=torch.int64).contiguous()
    if validate and ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= src.shape[0]):
        raise IndexError("gather index out of range")
    dst = torch.empty((ids.numel(),) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
    chw = src[0].numel() if src.shape[0] else 0
    check(lib().dpsx_gather_f32(ptr(src), ptr(ids), ptr(dst), ids.numel(), src.shape[0], chw, stream_of(src)),
          "dpsx_gather_f32")
    return dst


def replicate(src, idx_dev, n_out=None):
    """img[best.repeat(n)] (gaussian_diffusion.py:633): idx stays on the device, no host sync."""
    src = f32c(src)
    n_out = src.shape[0] if n_out is None else n_out
    idx_dev = idx_dev.to(device=src.device, dtype=torch.int64).reshape(1).contiguous()
    dst = torch.empty((n_out,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
    chw = src[0].numel() if src.shape[0] else 0
    check(lib().dpsx_replicate_f32(ptr(src), ptr(idx_dev), ptr(dst), n_out, src.shape[0], chw, stream_of(src)),
          "dpsx_replicate_f32")
    return dst


def pack_champion(particles, costs=None, best=None, best_val=None, out=None):
    """This rank's record for the champion exchange (distributed._exchange_champions): [C*H*W floats of particles[best] |
    cost, (float)best, 0, 0] in ONE launch.  best=None: the torch.argmin-order select over `costs` runs inside the launch."""
    particles = f32c(particles)
    n, chw = particles.shape[0], particles[0].numel()
    costs = None if costs is None else f32c(costs.reshape(-1), "costs")
    if costs is not None and costs.numel() != n:
        raise ValueError("one cost per particle")
    best = None if best is None else best.to(device=particles.device, dtype=torch.int64).reshape(1).contiguous()
    best_val = None if best_val is None else f32c(best_val.reshape(1), "best_val")
    if out is None:
        out = torch.empty(chw + 4, dtype=torch.float32, device=particles.device)
    check(lib().dpsx_pack_champion_f32(ptr(particles), ptr(costs), ptr(best), ptr(best_val), ptr(out), n, chw,
                                       stream_of(particles)), "dpsx_pack_champion_f32")
    return out


def select_champion(table, shape, n_out=1, want_index=False):
    """table [world, C*H*W + 4] as gathered -> n_out copies of the winning rank's champion [n_out, *shape]
    (first minimum over table[:, chw], lowest rank wins ties); want_index: also (winner rank, its local index) as device
    int64 scalars.  ONE launch, nothing read on the host."""
    table = f32c(table)
    world, chw = table.shape[0], table.shape[1] - 4
    dst = torch.empty((int(n_out),) + tuple(shape), dtype=torch.float32, device=table.device)
    if dst[0].numel() != chw:
        raise ValueError("table rows do not hold a particle of this shape plus its header")
    wr = torch.empty((), dtype=torch.int64, device=table.device) if want_index else None
    wl = torch.empty((), dtype=torch.int64, device=table.device) if want_index else None
    check(lib().dpsx_select_champion_f32(ptr(table), world, chw, ptr(dst), int(n_out), ptr(wr), ptr(wl), stream_of(table)),
          "dpsx_select_champion_f32")
    return (dst, wr, wl) if want_index else dst


# ------------------------------------------------------------------ fused DPS step
class StepBuffers:
    """Persistent per-(N, C, H, W) device buffers of the fused step (resident in HBM across steps).
    parent / offset: the buffers are the particle slice [offset, offset + n) of another StepBuffers (ParticleGroups: the
    groups' sample / x_next / norm / g_model_out are contiguous slices of one full-batch set, so the UNet and the select see
    [N, ...] tensors without a copy); the op-defined residual scratch is always the group's own."""

    def __init__(self, handle, n, c, h, w, device, parent=None, offset=0):
        self.shape = (n, c, h, w)
        f = dict(dtype=torch.float32, device=device)
        if parent is None:
            self.x0_hat = torch.empty((n, c, h, w), **f)
            self.sample = torch.empty((n, c, h, w), **f)
            self.inside = torch.empty((n, c, h, w), dtype=torch.uint8, device=device)
            self.norm = torch.empty(n, **f)
            # variance half of the UNet-output cotangent is identically zero for the DPS loss: zeroed once
            self.g_model_out = torch.zeros((n, 2 * c, h, w), **f)
            self.x_next = [torch.empty((n, c, h, w), **f), torch.empty((n, c, h, w), **f)]
        else:
            if tuple(parent.shape[1:]) != (c, h, w) or offset < 0 or offset + n > parent.shape[0]:
                raise ValueError("particle slice outside the parent buffers")
            sl = slice(offset, offset + n)
            self.x0_hat, self.sample, self.inside = parent.x0_hat[sl], parent.sample[sl], parent.inside[sl]
            self.norm, self.g_model_out = parent.norm[sl], parent.g_model_out[sl]
            self.x_next = [parent.x_next[0][sl], parent.x_next[1][sl]]
        rb = 0
        if handle is not None:
            rb = lib().dpsx_step_resid_bytes(handle._h, n, c, h, w)
            if rb < 0:
                check(int(rb), "dpsx_step_resid_bytes")
        self.resid = torch.empty(max(int(rb), 256), dtype=torch.uint8, device=device)
        self.flip = 0


def _stream_arg(stream, t):
    """stream: None (torch's current stream on t's device -- a 4.5 us lookup per launch) or a torch.cuda.Stream, whose
    raw handle a caller that drives several particle groups passes explicitly instead of entering a stream context
    (another 8 us per group and step: three groups cost the host 96 us per step that way, 35 this way).
    torch's caching allocator does not know about work enqueued this way: every tensor handed to a launch on a side stream
    must outlive that stream's work or be marked with `t.record_stream(stream)` (ParticleGroups does both for what it
    owns / is handed per step)."""
    if stream is None:
        return stream_of(t)
    if stream.device != t.device:
        raise ValueError(f"stream on {stream.device} given for tensors on {t.device}")
    return ctypes.c_void_p(stream.cuda_stream)


def step_fwd(handle, buf, x_t, model_out, noise, y, coefs, finalize_norm=False, want_x0=True, stream=None):
    """K1.  finalize_norm=False (the loop's setting): buf.norm is filled by the following step_bwd, whose prologue
    finalises the per-tile partial sums this launch leaves in the workspace.  finalize_norm=True: the launch finishes
    buf.norm itself (each particle's last block re-sums the partials in the same fixed order -- same bits) and
    step_bwd reads one float per particle; measured at N = 64: K1 +4.8 us, K2 -3.3 us, so the loop does not use it.
    want_x0=False (blur and resize operators): x0_hat is consumed inside the launch and not written to buf.x0_hat --
    the `ps` step reads it nowhere afterwards (the backward half works from the clamp gate); other operators ignore
    the flag."""
    n, c, h, w = buf.shape
    ws = handle.workspace(n, c, h, w, x_t.device)
    buf.norm_ready = bool(finalize_norm)
    optional = handle.kind in (_lib.KIND_SEP, _lib.KIND_TAPS, _lib.KIND_RESIZE) or \
        (handle.kind == _lib.KIND_PHASE and (h, w) == (256, 256) and getattr(handle, "spectral", False))
    x0_out = None if optional and not want_x0 else buf.x0_hat
    check(lib().dpsx_step_fwd_f32(handle._h, ptr(x_t), ptr(model_out), ptr(noise), ptr(y), y.shape[0],
                                  ptr(x0_out), ptr(buf.sample), ptr(buf.inside), ptr(buf.resid),
                                  ptr(buf.norm) if finalize_norm else None,
                                  n, c, h, w, byref(coefs), ptr(ws), ws.numel(), _stream_arg(stream, x_t)),
          "dpsx_step_fwd_f32")


def step_bwd(handle, buf, y, scale, power, coefs, g_x0_extra=None, stream=None):
    """K2.  g_x0_extra: optional [N, C, H, W] cotangent on x0_hat of a further loss term (the semantic-guidance
    term's VJP through the embedder), added to coef * A^T r before the clamp gate."""
    n, c, h, w = buf.shape
    ws = handle.workspace(n, c, h, w, buf.x0_hat.device)
    ready = getattr(buf, "norm_ready", True)
    if g_x0_extra is not None:
        g_x0_extra = f32c(g_x0_extra, "g_x0_extra")
        if tuple(g_x0_extra.shape) != (n, c, h, w):
            raise ValueError("g_x0_extra must have the particle batch's shape")
    check(lib().dpsx_step_bwd_extra_f32(handle._h, ptr(buf.resid), ptr(buf.norm) if ready else None, ptr(buf.norm),
                                        ptr(buf.inside), ptr(buf.x0_hat),
                                        ptr(y), y.shape[0], float(scale), int(power), ptr(g_x0_extra),
                                        ptr(buf.g_model_out),
                                        n, c, h, w, byref(coefs), ptr(ws), ws.numel(), _stream_arg(stream, buf.x0_hat)),
          "dpsx_step_bwd_extra_f32")
    buf.norm_ready = True


def step_update(buf, g_unet, coefs, stream=None):
    n, c, h, w = buf.shape
    out = buf.x_next[buf.flip]
    buf.flip ^= 1
    check(lib().dpsx_step_update_f32(ptr(buf.sample), ptr(buf.g_model_out), ptr(g_unet), ptr(out), n, c * h * w,
                                     byref(coefs), _stream_arg(stream, buf.sample)), "dpsx_step_update_f32")
    return out


# ------------------------------------------------------------------ particle groups on streams
class ParticleGroups:
    """The N particles of a fused DPS loop as `groups` independent sub-batches, each with its own operator handle,
    residual scratch and HIP stream (an operator handle, its workspace and its tail counters serve ONE stream at a time).

    The three launches of a step depend on each other, the particles do not (no collective, no cross-particle term in
    reference gaussian_diffusion.py:207-257), and the counters say every tile kernel's load / compute / store phases add
    up inside one chain: side by side, the bandwidth-bound launch of one group fills the
    arithmetic-bound phase of another.  Per-particle results do not depend on the grouping, bit for bit.
    Used by `GaussianDiffusion.p_sample_loop` (sampler.particle_groups), the driver (--particle_groups) and bench.py.

    All groups' sample / gate / norm / g_model_out / x_next are contiguous particle slices of ONE full-batch StepBuffers
    (`self.full`): `self.full.norm` is the [N] distance vector and `self.x_next()` the [N, C, H, W] state, no copy."""

    def __init__(self, operator, n, c, h, w, device, groups, mask=None, like=None, record_streams=True):
        """record_streams=False: the caller keeps every tensor it hands to the launches alive until the groups are joined
        (bench.py's device-resident rings), so the per-launch `record_stream` calls (about 1 us of host time each) are
        skipped."""
        device = torch.device(device)
        self.record_streams = bool(record_streams)
        groups = max(1, min(int(groups), max(n, 1)))
        self.n, self.shape = n, (n, c, h, w)
        self.sizes = [n // groups + (1 if j < n % groups else 0) for j in range(groups)]   # may differ by one particle
        self.starts = [sum(self.sizes[:j]) for j in range(groups)]
        self.slices = [slice(s, s + m) for s, m in zip(self.starts, self.sizes)]
        self.full = StepBuffers(None, n, c, h, w, device)
        probe = like if like is not None else self.full.sample
        self.handles = [operator.new_hip_handle(probe, mask=mask) for _ in range(groups)]
        self.bufs = [StepBuffers(hd, m, c, h, w, device, parent=self.full, offset=s)
                     for hd, m, s in zip(self.handles, self.sizes, self.starts)]
        self.streams = [torch.cuda.Stream(device=device) for _ in range(groups)]
        self.device = device

    def __len__(self):
        return len(self.bufs)

    # -- ordering against the caller's stream
    def fork(self, stream=None):
        """every group's stream waits for the work enqueued so far on `stream` (default: torch's current stream)"""
        cur = torch.cuda.current_stream(self.device) if stream is None else stream
        for s in self.streams:
            s.wait_stream(cur)

    def join(self, stream=None):
        """`stream` (default: torch's current stream) waits for every group's work"""
        cur = torch.cuda.current_stream(self.device) if stream is None else stream
        for s in self.streams:
            cur.wait_stream(s)

    def _slice(self, j, t):
        """the group's particles of a full-batch tensor (a contiguous view); a tensor that already has the group's
        size is taken as it is.  Tensors the caller allocates per step on its own stream are marked as in use by the
        group's stream (the caching allocator would otherwise hand their memory out again while the launch reads it)."""
        if t is None:
            return None
        if t.shape[0] == self.n and self.sizes[j] != self.n:
            v = t[self.slices[j]]
        elif t.shape[0] == self.sizes[j]:
            v = t
        else:
            raise ValueError(f"tensor of {t.shape[0]} particles handed to a group of {self.sizes[j]} (batch of {self.n})")
        if self.record_streams:
            v.record_stream(self.streams[j])
        return v

    # -- the three launches of group j, on its stream
    def step_fwd(self, j, x_t, model_out, noise, y, coefs, want_x0=True):
        step_fwd(self.handles[j], self.bufs[j], self._slice(j, x_t), self._slice(j, model_out), self._slice(j, noise),
                 y, coefs, want_x0=want_x0, stream=self.streams[j])

    def step_bwd(self, j, y, scale, power, coefs, g_x0_extra=None):
        step_bwd(self.handles[j], self.bufs[j], y, scale, power, coefs, g_x0_extra=self._slice(j, g_x0_extra),
                 stream=self.streams[j])

    def step_update(self, j, g_unet, coefs):
        return step_update(self.bufs[j], self._slice(j, g_unet), coefs, stream=self.streams[j])

    def x_next(self):
        """[N, C, H, W]: the state the groups' last step_update calls wrote (all groups flip together)"""
        flips = {b.flip for b in self.bufs}
        if len(flips) != 1:
            raise RuntimeError("the groups are not at the same step")
        return self.full.x_next[flips.pop() ^ 1]
