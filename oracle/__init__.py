"""ORACLE -- CPU restatement of the reference DPS hot path.  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (dps_ttc_amd) never imports, links or executes it.

numpy-in / numpy-out wrappers over oracle/_build/libdps_oracle.so (plain C,
see dps_oracle.c) plus the float64 host tables in oracle/tables.py.  Pinned
against the imported reference by tests/golden/make_golden.py.
"""
import ctypes
import os
import subprocess

import numpy as np

from . import tables  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdps_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_SO) or \
            os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "dps_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_argmin.restype = ctypes.c_int64
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _p(a, typ=_f32p):
    return None if a is None else a.ctypes.data_as(typ)


def _c(v):
    return ctypes.c_float(float(v))


def _l(v):
    return ctypes.c_int64(int(v))


# ---------------------------------------------------------------- S1
def posterior_fwd(x, model_out, noise, coefs):
    """-> dict(x0_hat, mean, logvar, sample, inside).  x [N,C,H,W], model_out [N,2C,H,W]."""
    x, model_out = _f(x), _f(model_out)
    noise = None if noise is None else _f(noise)
    n, chw = x.shape[0], int(np.prod(x.shape[1:]))
    assert model_out.shape[0] == n and int(np.prod(model_out.shape[1:])) == 2 * chw
    out = {k: np.empty_like(x) for k in ("x0_hat", "mean", "logvar", "sample")}
    out["inside"] = np.empty(x.shape, dtype=np.uint8)
    add = int(coefs["add_noise"])
    assert noise is not None or not add
    lib().orc_posterior_fwd(_p(x), _p(model_out), _p(noise), _p(out["x0_hat"]), _p(out["mean"]),
                            _p(out["logvar"]), _p(out["sample"]), _p(out["inside"], _u8p),
                            _l(n), _l(chw), _c(coefs["a"]), _c(coefs["b"]), _c(coefs["c1"]),
                            _c(coefs["c2"]), _c(coefs["min_log"]), _c(coefs["max_log"]),
                            ctypes.c_int(add))
    return out


def posterior_bwd(g_x0, g_sample, x, model_out, noise, coefs):
    """-> (g_x [N,C,H,W], g_model_out [N,2C,H,W])."""
    x, model_out = _f(x), _f(model_out)
    g_x0 = None if g_x0 is None else _f(g_x0)
    g_sample = None if g_sample is None else _f(g_sample)
    noise = None if noise is None else _f(noise)
    n, chw = x.shape[0], int(np.prod(x.shape[1:]))
    g_x, g_mo = np.empty_like(x), np.empty_like(model_out)
    lib().orc_posterior_bwd(_p(g_x0), _p(g_sample), _p(x), _p(model_out), _p(noise), _p(g_x),
                            _p(g_mo), _l(n), _l(chw), _c(coefs["a"]), _c(coefs["b"]),
                            _c(coefs["c1"]), _c(coefs["c2"]), _c(coefs["min_log"]),
                            _c(coefs["max_log"]), ctypes.c_int(int(coefs["add_noise"])))
    return g_x, g_mo


# ---------------------------------------------------------------- operators
def blur_fwd(x, kernel, skip_zero_taps=True):
    x, kernel = _f(x), _f(kernel)
    ks = kernel.shape[-1]
    assert kernel.shape == (ks, ks)
    h, w = x.shape[-2:]
    y = np.empty_like(x)
    lib().orc_blur_fwd(_p(x), _p(kernel), _p(y), _l(x.size // (h * w)), _l(h), _l(w), _l(ks),
                       ctypes.c_int(int(skip_zero_taps)))
    return y


def blur_adj(u, kernel):
    u, kernel = _f(u), _f(kernel)
    ks = kernel.shape[-1]
    h, w = u.shape[-2:]
    g = np.empty_like(u)
    lib().orc_blur_adj(_p(u), _p(kernel), _p(g), _l(u.size // (h * w)), _l(h), _l(w), _l(ks))
    return g


def _resize_axis(x, axis, wt, idx, adjoint, in_hw=None):
    wt = _f(wt)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    taps, n_out = wt.shape
    if not adjoint:
        h, w = x.shape[-2:]
        oshape = list(x.shape)
        oshape[-2 + axis] = n_out
        y = np.empty(oshape, dtype=np.float32)
        lib().orc_resize_axis_fwd(_p(x), _p(y), _l(x.size // (h * w)), _l(h), _l(w),
                                  ctypes.c_int(axis), _l(taps), _l(n_out), _p(wt), _p(idx, _i64p))
        return y
    h, w = in_hw
    gshape = list(x.shape[:-2]) + [h, w]
    g = np.empty(gshape, dtype=np.float32)
    lib().orc_resize_axis_adj(_p(x), _p(g), _l(g.size // (h * w)), _l(h), _l(w),
                              ctypes.c_int(axis), _l(taps), _l(n_out), _p(wt), _p(idx, _i64p))
    return g


def resize_fwd(x, tabs):
    """Resizer.forward (util/resizer.py:55-74), axes in tabs['order'] (torch dims 2=H, 3=W)."""
    x = _f(x)
    for d in tabs["order"]:
        key = "h" if d == 2 else "w"
        x = _resize_axis(x, d - 2, tabs["w_" + key], tabs["i_" + key], False)
    return x


def resize_adj(u, tabs, in_hw):
    """Exact adjoint (what autograd gives), axes undone in reverse order."""
    u = _f(u)
    h, w = in_hw
    for d in reversed(tabs["order"]):
        key = "h" if d == 2 else "w"
        cur_h = h if d == 2 else u.shape[-2]
        cur_w = w if d == 3 else u.shape[-1]
        u = _resize_axis(u, d - 2, tabs["w_" + key], tabs["i_" + key], True, (cur_h, cur_w))
    return u


def mask_mul(x, mask):
    """InpaintingOperator.forward and its adjoint (measurements.py:158-162)."""
    x, mask = _f(x), _f(mask)
    h, w = x.shape[-2:]
    assert mask.size == h * w
    y = np.empty_like(x)
    lib().orc_mask_mul(_p(x), _p(mask), _p(y), _l(x.size // (h * w)), _l(h * w))
    return y


def phase_fwd(x, pad, want_spectrum=False):
    x = _f(x)
    h, w = x.shape[-2:]
    assert h == w
    s = h + 2 * pad
    shape = list(x.shape[:-2]) + [s, s]
    amp = np.empty(shape, dtype=np.float32)
    re = np.empty(shape, dtype=np.float32)
    im = np.empty(shape, dtype=np.float32)
    lib().orc_phase_fwd(_p(x), _p(amp), _p(re), _p(im), _l(x.size // (h * w)), _l(h), _l(pad))
    return (amp, re, im) if want_spectrum else amp


def phase_adj(u, spec_re, spec_im, h, pad):
    u, spec_re, spec_im = _f(u), _f(spec_re), _f(spec_im)
    s = h + 2 * pad
    g = np.empty(list(u.shape[:-2]) + [h, h], dtype=np.float32)
    lib().orc_phase_adj(_p(u), _p(spec_re), _p(spec_im), _p(g), _l(u.size // (s * s)), _l(h), _l(pad))
    return g


# ---------------------------------------------------------------- norm / update / select
def residual_norm(y, ax):
    y, ax = _f(y), _f(ax)
    n = ax.shape[0]
    m = ax.size // n
    y_n = y.shape[0]
    assert y.size // y_n == m and y_n in (1, n)
    r = np.empty_like(ax)
    norm = np.empty(n, dtype=np.float32)
    lib().orc_residual_norm(_p(y), _l(y_n), _p(ax), _p(r), _p(norm), _l(n), _l(m))
    return r, norm


def norm_bwd(r, norm, g_norm, power=1):
    r, norm, g_norm = _f(r), _f(norm), _f(g_norm)
    n = r.shape[0]
    g = np.empty_like(r)
    lib().orc_norm_bwd(_p(r), _p(norm), _p(g_norm), ctypes.c_int(power), _p(g), _l(n), _l(r.size // n))
    return g


def update(sample, grad):
    sample, grad = _f(sample), _f(grad)
    out = np.empty_like(sample)
    lib().orc_update(_p(sample), _p(grad), _p(out), _l(sample.size))
    return out


def argmin(v):
    v = _f(v)
    return int(lib().orc_argmin(_p(v), _l(v.size)))


def gather(src, ids):
    src = _f(src)
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    dst = np.empty((ids.size,) + src.shape[1:], dtype=np.float32)
    lib().orc_gather(_p(src), _p(ids, _i64p), _p(dst), _l(ids.size), _l(src.size // src.shape[0]))
    return dst


def resample_cost(op, x, y, prev_costs=None, potential_type="min"):
    """Cost update of SearchDDPM.resample_update (gaussian_diffusion.py:556-585):
    curr[p] = ||y - A(x_p)||_1^2 / (C H W) (:557-563); net by potential type (:565-585).  -> (curr, net) float32."""
    x = _f(x)
    n = x.shape[0]
    ax = op.forward(x)
    yy = np.broadcast_to(_f(y), ax.shape) if _f(y).shape[0] == 1 else _f(y)
    l1 = np.abs(yy.astype(np.float64) - ax.astype(np.float64)).reshape(n, -1).sum(axis=1)
    curr = (l1 ** 2 / float(np.prod(x.shape[1:]))).astype(np.float32)
    if potential_type not in ("mean", "min", "diff", "curr"):
        raise NotImplementedError(potential_type)
    if prev_costs is None or potential_type == "curr":
        return curr, curr.copy()
    prev = _f(prev_costs)
    if potential_type == "mean":
        net = curr + prev
    elif potential_type == "min":
        net = np.where(np.isnan(curr) | np.isnan(prev), np.float32(np.nan), np.minimum(curr, prev))
    else:
        net = curr - prev
    return curr, net.astype(np.float32)


def resample_update(op, candidates, denoised, y, prev_costs=None, potential_type="min", ids=None):
    """SearchDDPM.resample_update (gaussian_diffusion.py:515-587) for a GIVEN multinomial draw `ids` (the draw itself
    is torch.multinomial on exp(-rs_temp * prev / steps), :539-546; None = no resampling) -> (candidates, net)."""
    candidates, denoised = _f(candidates), _f(denoised)
    if ids is not None and prev_costs is not None:
        candidates, denoised, prev_costs = gather(candidates, ids), gather(denoised, ids), _f(prev_costs)[np.asarray(ids)]
    return candidates, resample_cost(op, denoised, y, prev_costs, potential_type)[1]


# ---------------------------------------------------------------- operator objects
class Operator:
    """forward / adjoint pair for one measurement operator (linear unless noted)."""

    def __init__(self, name, **kw):
        self.name = name
        self.kw = kw
        self._spec = None

    def forward(self, x):
        k = self.kw
        if self.name in ("gaussian_blur", "motion_blur"):
            return blur_fwd(x, k["kernel"], k.get("skip_zero_taps", True))
        if self.name == "super_resolution":
            return resize_fwd(x, k["tables"])
        if self.name == "inpainting":
            return mask_mul(x, k["mask"])
        if self.name == "noise":
            return _f(x).copy()
        if self.name == "phase_retrieval":
            amp, re, im = phase_fwd(x, k["pad"], True)
            self._spec = (re, im)
            return amp
        raise NameError(self.name)

    def adjoint(self, u, in_hw):
        """VJP at the point of the last forward() call (linear ops ignore the point)."""
        k = self.kw
        if self.name in ("gaussian_blur", "motion_blur"):
            return blur_adj(u, k["kernel"])
        if self.name == "super_resolution":
            return resize_adj(u, k["tables"], in_hw)
        if self.name == "inpainting":
            return mask_mul(u, k["mask"])
        if self.name == "noise":
            return _f(u).copy()
        if self.name == "phase_retrieval":
            return phase_adj(u, self._spec[0], self._spec[1], in_hw[0], k["pad"])
        raise NameError(self.name)


def make_operator(name, **cfg):
    """Mirror of get_operator(name, **yaml) (measurements.py:29) for the oracle."""
    if name == "gaussian_blur":
        k = tables.gaussian_kernel2d(cfg["kernel_size"], cfg["intensity"]).astype(np.float32)
        return Operator(name, kernel=k)
    if name == "motion_blur":
        return Operator(name, kernel=np.asarray(cfg["kernel"], dtype=np.float32))
    if name == "super_resolution":
        shp = cfg["in_shape"]
        return Operator(name, tables=tables.resize_tables(shp[-2], shp[-1], cfg["scale_factor"]))
    if name == "inpainting":
        return Operator(name, mask=np.asarray(cfg["mask"], dtype=np.float32))
    if name == "phase_retrieval":
        return Operator(name, pad=int((cfg["oversample"] / 8.0) * 256))
    if name == "noise":
        return Operator(name)
    raise NameError(f"Name {name} is not defined.")


# ---------------------------------------------------------------- one DPS step
def dps_step(op, x_prev, model_out, noise, y, coefs, scale=1.0, power=1, g_unet_fn=None, g_x0_extra=None):
    """One step of the base loop with 'ps_semantic' (sem_guid_scale=0)  /  'ps':
    gaussian_diffusion.py:207-257 + condition_methods.py:145-187.

    g_unet_fn(g_model_out) -> J_model^T g  (the UNet VJP; zeros if None).
    g_x0_extra: optional cotangent on x0_hat of a loss term beyond the measurement norm.
    Returns dict(x0_hat, sample, norm, grad, x_next, g_model_out, g_direct).
    """
    f = posterior_fwd(x_prev, model_out, noise, coefs)
    ax = op.forward(f["x0_hat"])
    r, norm = residual_norm(y, ax)
    g_ax = norm_bwd(r, norm, np.full(norm.shape, scale, dtype=np.float32), power)
    g_x0 = op.adjoint(g_ax.reshape(ax.shape), x_prev.shape[-2:])
    if g_x0_extra is not None:      # cotangent of a further loss term on x0_hat (semantic guidance, :155-187)
        g_x0 = (_f(g_x0) + _f(g_x0_extra)).astype(np.float32)
    g_direct, g_mo = posterior_bwd(g_x0, None, x_prev, model_out, noise, coefs)
    grad = g_direct if g_unet_fn is None else g_direct + _f(g_unet_fn(g_mo))
    x_next = update(f["sample"], grad)
    return {"x0_hat": f["x0_hat"], "sample": f["sample"], "norm": norm, "grad": grad,
            "x_next": x_next, "g_model_out": g_mo, "g_direct": g_direct, "inside": f["inside"]}
