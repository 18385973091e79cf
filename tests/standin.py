"""Deterministic stand-ins shared by tests/golden/make_golden.py and the tests.

* StandInModel: differentiable element-wise replacement for the UNet
  (`model(x, t) -> [N, 2C, H, W]`).  Uses only + - * / abs and a table lookup,
  all IEEE-exact in fp32, so CPU and GPU evaluate it bit-identically and the
  free-running loop fixtures are not polluted by libm differences.
* synthetic_motion_kernel: a sparse, asymmetric, normalised k x k blur path
  (the reference's third-party `motionblur` generator is not installed; the
  kernel is an *input* of the hot path -- SURVEY.md 8c).
"""
import numpy as np
import torch


class StandInModel(torch.nn.Module):
    def __init__(self, steps=1000):
        super().__init__()
        betas = np.linspace(1e-4, 2e-2, steps, dtype=np.float64) * (1000.0 / steps)
        abar = np.cumprod(1.0 - betas)
        self.register_buffer("sa", torch.tensor(np.sqrt(abar), dtype=torch.float32))
        self.register_buffer("s1", torch.tensor(np.sqrt(1.0 - abar), dtype=torch.float32))
        self.steps = steps

    def forward(self, x, t):
        idx = (t.float() * (self.steps / 1000.0)).round().long().clamp(0, self.steps - 1)
        sa = self.sa[idx].view(-1, 1, 1, 1)
        s1 = self.s1[idx].view(-1, 1, 1, 1)
        squash = x / (1.0 + x.abs())
        x0 = 1.6 * squash
        eps = (x - sa * x0) / s1
        v = 0.8 * squash * (1.0 - 0.5 * (t.float().view(-1, 1, 1, 1) / 1000.0))
        return torch.cat([eps, v], dim=1)


class ToyEmbedder(torch.nn.Module):
    """Stand-in for the reference's face-embedding network (facenet_pytorch InceptionResnetV1, not available offline):
    average-pool to 8 x 8, a fixed random linear map to 16 features, tanh.  Differentiable, deterministic (seeded
    weights), batch-independent -- enough to drive the semantic-guidance term's plumbing end to end."""

    def __init__(self, dim=16, seed=5):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.register_buffer("w", torch.randn(3 * 8 * 8, dim, generator=g) / 8.0)

    def forward(self, x):
        f = torch.nn.functional.adaptive_avg_pool2d(x.float(), 8).reshape(x.shape[0], -1)
        return torch.tanh(f @ self.w)


def toy_embedder(device):
    """factory for the driver's --embedder flag (module:callable)"""
    return ToyEmbedder().to(device).eval()


def synthetic_motion_kernel(size=61, seed=0, steps=120):
    """Random-walk path rasterised with bilinear splats, normalised to sum 1."""
    rng = np.random.RandomState(seed)
    k = np.zeros((size, size), dtype=np.float64)
    pos = np.array([size / 2.0, size / 2.0])
    vel = rng.randn(2) * 0.6
    for _ in range(steps):
        vel = 0.9 * vel + 0.35 * rng.randn(2)
        pos = np.clip(pos + vel, 1.0, size - 2.001)
        i, j = int(pos[0]), int(pos[1])
        fi, fj = pos[0] - i, pos[1] - j
        k[i, j] += (1 - fi) * (1 - fj)
        k[i + 1, j] += fi * (1 - fj)
        k[i, j + 1] += (1 - fi) * fj
        k[i + 1, j + 1] += fi * fj
    return (k / k.sum()).astype(np.float32)


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.linalg.norm(b.ravel())
    num = np.linalg.norm((a - b).ravel())
    return num / den if den > 0 else num
