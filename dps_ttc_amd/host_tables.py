"""Host-side (numpy, float64) constants the measurement operators are built from.

These run once per operator, before the loop; the per-step work is all HIP.
Reference (paths under /root/reference): util/img_utils.py:286-293 (Gaussian
kernel through scipy.ndimage.gaussian_filter), util/resizer.py:104-178 (cubic
antialiased contributions), util/img_utils.py:184-235 (mask_generator).
"""
import math

import numpy as np


# ------------------------------------------------------------------ Gaussian blur kernel
def gaussian_blur_kernel(kernel_size, std, truncate=4.0):
    """Response of scipy.ndimage.gaussian_filter(sigma=std) to a centred delta on a
    kernel_size x kernel_size grid (util/img_utils.py:286-293), float64.

    scipy filters axis by axis with taps exp(-x^2 / 2 sigma^2) / sum over
    |x| <= int(truncate * sigma + 0.5) and mirrors (d c b a | a b c d) at the grid
    edge, so the 2-D response is the outer product of one folded 1-D response.
    """
    radius = int(truncate * float(std) + 0.5)
    taps = [math.exp(-0.5 * (i / std) ** 2) for i in range(-radius, radius + 1)]
    total = math.fsum(taps)
    centre = kernel_size // 2
    line = np.zeros(kernel_size, dtype=np.float64)
    for i, t in enumerate(taps):
        pos = centre + i - radius
        pos %= 2 * kernel_size                       # symmetric (half-sample) extension has period 2n
        if pos >= kernel_size:
            pos = 2 * kernel_size - 1 - pos
        line[pos] += t / total
    return np.outer(line, line)


# ------------------------------------------------------------------ resizer tables
def _keys_cubic(d):
    d = abs(d)
    if d <= 1.0:
        return (1.5 * d - 2.5) * d * d + 1.0
    if d <= 2.0:
        return ((-0.5 * d + 2.5) * d - 4.0) * d + 2.0
    return 0.0


def resizer_axis(in_len, scale):
    """(weights [K, out] f32, field_of_view [K, out] int64) of Resizer for one axis
    (util/resizer.py:104-167), cubic kernel, antialiasing when scale < 1."""
    out_len = int(math.ceil(in_len * scale))
    stretch = scale if scale < 1 else 1.0            # antialiasing widens the kernel by 1/scale
    width = 4.0 / stretch
    span = int(math.ceil(width)) + 2
    rows_w, rows_i = [], []
    for o in range(1, out_len + 1):
        centre = (o - (out_len - in_len * scale) / 2.0) / scale + 0.5 * (1.0 - 1.0 / scale)
        first = math.floor(centre - width / 2.0)
        pos = [first + t - 1 for t in range(span)]
        wts = [stretch * _keys_cubic(stretch * (centre - p - 1)) for p in pos]
        rows_w.append(wts)
        rows_i.append(pos)
    w = np.array(rows_w, dtype=np.float64)
    tot = w.sum(axis=1, keepdims=True)
    tot[tot == 0] = 1.0
    w /= tot
    idx = np.array(rows_i, dtype=np.int64) % (2 * in_len)     # reflection via the mirrored index line
    idx = np.where(idx >= in_len, 2 * in_len - 1 - idx, idx)
    keep = np.any(w != 0.0, axis=0)                           # drop taps that are zero for every output
    return w[:, keep].T.astype(np.float32), idx[:, keep].T.astype(np.int64)


# ------------------------------------------------------------------ motion kernel (fallback generator)
def random_motion_kernel(size, intensity, rng=None):
    """A normalised random-walk blur path.  The reference draws its kernels from the
    third-party `motionblur` package (measurements.py:104, not installed and unpinned):
    kernel GENERATION is therefore parity-unpinned; any k x k kernel can be injected with
    MotionBlurOperator.set_kernel().  Uses numpy's global RNG like the reference so that
    np.random.seed(kernel_idx) (sample_condition_batched_ttc.py:76) selects the kernel."""
    rng = np.random if rng is None else rng
    k = np.zeros((size, size), dtype=np.float64)
    steps = max(8, int(size * 2))
    pos = np.array([size / 2.0, size / 2.0])
    ang = rng.uniform(0, 2 * math.pi)
    for _ in range(steps):
        ang += rng.normal(0.0, 0.2 + 1.5 * intensity)
        pos = np.clip(pos + 0.5 * np.array([math.sin(ang), math.cos(ang)]), 1.0, size - 2.001)
        i, j = int(pos[0]), int(pos[1])
        fi, fj = pos[0] - i, pos[1] - j
        k[i, j] += (1 - fi) * (1 - fj)
        k[i + 1, j] += fi * (1 - fj)
        k[i, j + 1] += (1 - fi) * fj
        k[i + 1, j + 1] += fi * fj
    return k / k.sum()
