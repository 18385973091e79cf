"""ORACLE, second form -- the reference's per-step arithmetic as STOCK torch CPU ops.  TEST INFRASTRUCTURE.

SURVEY.md 8d asks for two CPU baselines beside the GPU figure: the C restatement (dps_oracle.c, `kind: "port"`) and
"stock torch-CPU ops equivalent to the reference path".  This module is the second: the ATen ops the reference itself
launches per step, in its order, with autograd doing the backward half --

    S1      p_mean_variance + DDPM.p_sample        gaussian_diffusion.py:308-330, 466-476; posterior_mean_variance.py:96-129, 211-242
    A       ReflectionPad2d + depthwise Conv2d     util/img_utils.py:275-283        (gaussian_blur, motion_blur)
            x[fov] * w, sum(0) per axis            util/resizer.py:59-72            (super_resolution)
            x * mask                               measurements.py:160              (inpainting)
            pad, ifftshift, fft2(ortho), fftshift, abs   measurements.py:186-189, util/fastmri_utils.py:67-89
    norm    linalg.norm(y - A x0_hat) per particle, autograd.grad    condition_methods.py:179-185
    update  sample - grad                          gaussian_diffusion.py:255

The UNet is outside the path here as everywhere: model_out is a leaf and its VJP back to x_t is the synthetic g_unet,
so autograd runs norm -> A -> clamp -> (x_t, eps).  Only tests/ and bench.py's cpu_baseline leg import this.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _gather_axis(x, w, idx, dim):
    """Resizer.forward for one axis (resizer.py:59-72): transpose, x[fov] * w, sum over the taps"""
    xt = x.transpose(dim, 0)
    out = (xt[idx] * w.reshape(w.shape + (1,) * (xt.dim() - 1))).sum(0)
    return out.transpose(dim, 0)


class TorchOperator:
    def __init__(self, name, kernel=None, tables=None, mask=None, pad=0):
        self.name = name
        if kernel is not None:
            k = torch.as_tensor(np.asarray(kernel, dtype=np.float32))
            self.weight = k.reshape(1, 1, *k.shape).repeat(3, 1, 1, 1)
            self.pad = k.shape[0] // 2
        if tables is not None:      # oracle.tables.resize_tables(): axes in Resizer's order (torch dims 2 = H, 3 = W)
            self.axes = [(d, torch.as_tensor(np.asarray(tables["w_h" if d == 2 else "w_w"], np.float32)),
                          torch.as_tensor(np.asarray(tables["i_h" if d == 2 else "i_w"], np.int64))) for d in tables["order"]]
        if mask is not None:
            self.mask = torch.as_tensor(np.asarray(mask, dtype=np.float32))
        self.fpad = pad

    def forward(self, x):
        if self.name in ("gaussian_blur", "motion_blur"):
            return F.conv2d(F.pad(x, (self.pad,) * 4, mode="reflect"), self.weight, groups=3)
        if self.name == "super_resolution":
            for d, w, idx in self.axes:
                x = _gather_axis(x, w, idx, d)
            return x
        if self.name == "inpainting":
            return x * self.mask
        if self.name == "phase_retrieval":
            p = F.pad(x, (self.fpad,) * 4)
            z = torch.fft.fftshift(torch.fft.fftn(torch.fft.ifftshift(p.to(torch.complex64), dim=(-2, -1)), dim=(-2, -1),
                                                  norm="ortho"), dim=(-2, -1))
            return z.abs()
        raise NameError(self.name)


def dps_step(op, x_t, model_out, noise, y, coefs, scale, g_unet):
    """one `ps` step as the reference's ATen ops -> dict(x0_hat, sample, norm, x_next), numpy fp32"""
    x = torch.as_tensor(x_t).clone().requires_grad_()
    mo = torch.as_tensor(model_out)
    c = mo.shape[1] // 2
    eps = mo[:, :c].clone().requires_grad_()
    v = mo[:, c:]
    a, b, c1, c2 = (float(np.float32(coefs[k])) for k in ("a", "b", "c1", "c2"))
    x0 = (a * x - b * eps).clamp(-1, 1)
    mean = c1 * x0 + c2 * x
    frac = (v + 1) / 2
    logvar = frac * float(np.float32(coefs["max_log"])) + (1 - frac) * float(np.float32(coefs["min_log"]))
    sample = mean + torch.exp(0.5 * logvar) * torch.as_tensor(noise) if coefs["add_noise"] else mean
    ax = op.forward(x0)
    diff = torch.as_tensor(y) - ax
    norm = torch.linalg.norm(diff.reshape(diff.shape[0], -1), dim=-1)
    g_x, g_eps = torch.autograd.grad((scale * norm).sum(), (x, eps))
    # grad w.r.t. x_t = direct term + the UNet's VJP of g_eps (synthetic here)
    x_next = sample.detach() - (g_x + torch.as_tensor(g_unet))
    return {"x0_hat": x0.detach().numpy(), "sample": sample.detach().numpy(), "norm": norm.detach().numpy(),
            "x_next": x_next.numpy(), "g_eps": g_eps.numpy()}
