"""ORACLE (test infrastructure only) -- host-side constant tables, numpy float64.

CPU restatement of the tables the reference builds once before the DPS loop.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product path (dps_ttc_amd) never does.

Each function cites the reference file:line (relative to /root/reference) it
restates.  Pinned against the reference itself by tests/golden/make_golden.py
(fixtures: tests/golden/tables.npz).
"""
import math

import numpy as np


# --------------------------------------------------------------------------
# Noise schedule  (guided_diffusion/gaussian_diffusion.py:718-735, 70-109)
# --------------------------------------------------------------------------
def named_betas(name, steps):
    """gaussian_diffusion.py:718-743 (linear) and :746-763 (cosine)."""
    if name == "linear":
        k = 1000.0 / steps
        return np.linspace(k * 1e-4, k * 2e-2, steps, dtype=np.float64)
    if name == "cosine":
        f = lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.array([min(1.0 - f((i + 1) / steps) / f(i / steps), 0.999)
                         for i in range(steps)], dtype=np.float64)
    raise NotImplementedError(name)


def kept_timesteps(steps, spec):
    """gaussian_diffusion.py:338-392 -- which base timesteps a respacing keeps."""
    if isinstance(spec, str):
        if spec.startswith("ddim"):
            want = int(spec[4:])
            for stride in range(1, steps):
                if len(range(0, steps, stride)) == want:
                    return sorted(range(0, steps, stride))
            raise ValueError("no integer stride gives %d steps" % want)
        spec = [int(s) for s in spec.split(",")]
    elif isinstance(spec, int):
        spec = [spec]
    base, extra = divmod(steps, len(spec))
    kept, origin = [], 0
    for i, count in enumerate(spec):
        size = base + (1 if i < extra else 0)
        if size < count:
            raise ValueError("cannot take %d steps from a section of %d" % (count, size))
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            kept.append(origin + round(pos))
            pos += stride
        origin += size
    return sorted(set(kept))


def schedule(steps=1000, name="linear", respacing=""):
    """All per-timestep tables of the (spaced) diffusion, float64.

    gaussian_diffusion.py:34-56 (create_sampler), :403-418 (SpacedDiffusion
    re-derives betas from alpha-bar ratios), :70-109 (the tables),
    posterior_mean_variance.py:98-107, 213-228 (same tables again).
    """
    base = named_betas(name, steps)
    keep = kept_timesteps(steps, respacing if respacing else [steps])
    abar_base = np.cumprod(1.0 - base)
    betas, last = [], 1.0
    for i in keep:
        betas.append(1.0 - abar_base[i] / last)
        last = abar_base[i]
    betas = np.array(betas, dtype=np.float64)
    alphas = 1.0 - betas
    abar = np.cumprod(alphas)
    abar_prev = np.append(1.0, abar[:-1])
    post_var = betas * (1.0 - abar_prev) / (1.0 - abar)
    return {
        "timestep_map": np.array(keep, dtype=np.int64),
        "original_steps": steps,
        "betas": betas,
        "alphas_cumprod": abar,
        "alphas_cumprod_prev": abar_prev,
        "sqrt_alphas_cumprod": np.sqrt(abar),
        "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - abar),
        "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / abar),
        "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / abar - 1.0),
        "posterior_mean_coef1": betas * np.sqrt(abar_prev) / (1.0 - abar),
        "posterior_mean_coef2": (1.0 - abar_prev) * np.sqrt(alphas) / (1.0 - abar),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": np.log(np.append(post_var[1], post_var[1:])),
        "log_betas": np.log(betas),
    }


def step_coefs(sched, t):
    """The six fp32 scalars one DDPM step uses at (spaced) index t.

    extract_and_expand casts the f64 table entry with .float()
    (gaussian_diffusion.py:769-773, posterior_mean_variance.py:248-252).
    """
    f = lambda key: np.float32(sched[key][t])
    return {
        "a": f("sqrt_recip_alphas_cumprod"),        # posterior_mean_variance.py:121
        "b": f("sqrt_recipm1_alphas_cumprod"),      # :122
        "c1": f("posterior_mean_coef1"),            # :116
        "c2": f("posterior_mean_coef2"),            # :117
        "min_log": f("posterior_log_variance_clipped"),  # :235
        "max_log": f("log_betas"),                  # :236
        "add_noise": int(t != 0),                   # gaussian_diffusion.py:473
    }


def ddim_step_coefs(sched, t, eta=0.0):
    """The scalars of one DDIM step (gaussian_diffusion.py:481-509) in the same record: c1 = sqrt(abar_prev),
    c2 = sqrt(1 - abar_prev - sigma^2), min_log = sigma, add_noise = 2 | (t != 0).  The reference evaluates
    them as fp32 tensor arithmetic on .float()-cast table entries, one rounding per op -- restated here
    with numpy float32 scalars."""
    f = lambda key: np.float32(sched[key][t])
    one = np.float32(1.0)
    ab, abp = f("alphas_cumprod"), f("alphas_cumprod_prev")
    sigma = np.float32(eta) * np.sqrt((one - abp) / (one - ab)) * np.sqrt(one - ab / abp)
    return {
        "a": f("sqrt_recip_alphas_cumprod"),
        "b": f("sqrt_recipm1_alphas_cumprod"),
        "c1": np.sqrt(abp),
        "c2": np.sqrt(one - abp - sigma ** 2),
        "min_log": np.float32(sigma),
        "max_log": np.float32(0.0),
        "add_noise": 2 | int(t != 0),
    }


def model_timestep(sched, t, rescale=True):
    """gaussian_diffusion.py:455-463 -- what the wrapped UNet receives as t."""
    v = float(sched["timestep_map"][t])
    return np.float32(v * (1000.0 / sched["original_steps"])) if rescale else v


# --------------------------------------------------------------------------
# Gaussian blur kernel  (util/img_utils.py:286-293 via scipy.ndimage)
# --------------------------------------------------------------------------
def gaussian_taps(sigma, truncate=4.0):
    """scipy.ndimage.gaussian_filter1d's kernel: radius=int(truncate*sigma+0.5),
    exp(-x^2/(2 sigma^2)) normalised to sum 1 (scipy 1.9.1 .. 1.15 agree)."""
    r = int(truncate * float(sigma) + 0.5)
    x = np.arange(-r, r + 1, dtype=np.float64)
    g = np.exp(-0.5 / (sigma * sigma) * x * x)
    return g / g.sum()


def gaussian_kernel2d(size, sigma):
    """img_utils.py:286-293: gaussian_filter of a centred delta on a size x size
    grid, scipy default mode='reflect' (d c b a | a b c d | d c b a).  The
    filter runs axis by axis, so the result is the outer product of two 1-D
    responses; each response is the tap vector folded at the grid edges."""
    g = gaussian_taps(sigma)
    r = (len(g) - 1) // 2
    c = size // 2
    line = np.zeros(size, dtype=np.float64)
    period = 2 * size
    for j, wgt in enumerate(g):           # delta at c spreads to c + (j - r)
        p = (c + j - r) % period
        if p >= size:
            p = period - 1 - p            # half-sample symmetric reflection
        line[p] += wgt
    return np.outer(line, line)


# --------------------------------------------------------------------------
# Antialiased cubic resize tables  (util/resizer.py:104-178)
# --------------------------------------------------------------------------
def _cubic(x):
    """resizer.py:173-178 (Keys cubic, a=-0.5)."""
    ax = np.abs(x)
    near = (1.5 * ax ** 3 - 2.5 * ax ** 2 + 1.0) * (ax <= 1)
    far = (-0.5 * ax ** 3 + 2.5 * ax ** 2 - 4.0 * ax + 2.0) * ((ax > 1) & (ax <= 2))
    return near + far


def resize_axis_table(n_in, scale):
    """One axis of Resizer(in_shape, scale) with the default cubic kernel and
    antialiasing (resizer.py:104-167).  Returns (weights[K, n_out] f32,
    index[K, n_out] int64) exactly as Resizer stores them (:42, :50)."""
    n_out = int(np.ceil(n_in * scale))                           # :100
    aa = scale < 1                                               # :26
    support = 4.0 / scale if aa else 4.0                         # :113
    out_pos = np.arange(1, n_out + 1) - (n_out - n_in * scale) / 2.0   # :116-121
    centre = out_pos / scale + 0.5 * (1.0 - 1.0 / scale)         # :132
    left = np.floor(centre - support / 2.0)                      # :135
    span = int(np.ceil(support) + 2)                             # :139
    idx = np.int16(left[:, None] + np.arange(span) - 1)          # :144-145
    dist = centre[:, None] - idx - 1.0                           # :150
    wts = (scale * _cubic(scale * dist)) if aa else _cubic(dist)  # :112
    tot = wts.sum(axis=1)
    tot[tot == 0] = 1.0
    wts = wts / tot[:, None]                                     # :153-155
    mirror = np.concatenate([np.arange(n_in), np.arange(n_in - 1, -1, -1)])
    idx = mirror[np.mod(idx, 2 * n_in)]                          # :158-159
    live = np.nonzero(np.any(wts, axis=0))[0]                    # :162-164
    wts, idx = wts[:, live], idx[:, live]
    return wts.T.astype(np.float32), idx.T.astype(np.int64)


def resize_tables(h, w, factor):
    """SuperResolutionOperator(in_shape=[1,3,h,w], scale_factor=factor)
    (measurements.py:78-82): scale 1/factor on H and W; Resizer orders the axes
    with np.argsort of the per-axis scales [1, 1, s, s] (:29-30), which for
    these four values yields W (dim 3) before H (dim 2)."""
    s = 1.0 / factor
    order = [int(d) for d in np.argsort(np.array([1, 1, s, s])) if d >= 2]
    wh, ih = resize_axis_table(h, s)
    ww, iw = resize_axis_table(w, s)
    return {"order": order, "w_h": wh, "i_h": ih, "w_w": ww, "i_w": iw}
