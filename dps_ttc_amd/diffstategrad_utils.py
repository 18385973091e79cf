"""DiffStateGrad: project the guidance gradient onto the low-rank subspace of the current state
(reference guided_diffusion/diffstategrad_utils.py:4-78, loop hook gaussian_diffusion.py:240-251).

Off by default (`project=False`).  It is dense linear algebra on one 3 x H x W image every `period`
steps -- library SVD / GEMM on device tensors (rocSOLVER / rocBLAS through PyTorch-ROCm), not a
hand-written kernel.  The reference's batch-0 semantics are kept: the subspace comes from z_t[0], only
norm_grad[0] is projected and the [1, C, H, W] result then broadcasts over every particle in the
update (:255).
"""
import numpy as np
import torch


def compute_rank_for_explained_variance(singular_values, explained_variance_cutoff):
    """reference :4-22, literally: each list entry is cumulated FLATTENED (numpy.cumsum without an axis), so
    for the [C, W] array the loop hook passes, the channels' spectra are concatenated before the cutoff is
    searched; the index (+1) is divided by 3.  Host numpy in the singular values' own dtype, as there."""
    total_rank = 0
    for channel_singular_values in singular_values:
        squared = channel_singular_values ** 2
        cumulative = np.cumsum(squared) / np.sum(squared)
        total_rank += np.searchsorted(cumulative, explained_variance_cutoff) + 1
    return int(total_rank / 3)


def compute_svd_and_adaptive_rank(z_t, var_cutoff):
    """reference :24-44 -- SVD of the C channel matrices of z_t[0] (rocSOLVER through torch.linalg.svd); the
    rank needs the singular values on the host, as in the reference (one sync per projected step)."""
    U, s, Vh = torch.linalg.svd(z_t[0].float(), full_matrices=False)
    adaptive_rank = compute_rank_for_explained_variance([s.detach().cpu().numpy()], var_cutoff)
    return U, s, Vh, adaptive_rank


def apply_diffstategrad(norm_grad, iteration_count, period, U=None, s=None, Vh=None, adaptive_rank=None):
    """reference :46-78"""
    if period != 0 and iteration_count % period == 0:
        if any(p is None for p in (U, s, Vh, adaptive_rank)):
            raise ValueError("SVD components and adaptive_rank must be provided when iteration_count % period == 0")
        A = U[:, :, :adaptive_rank]
        B = Vh[:, :adaptive_rank, :]
        low_rank_grad = torch.matmul(A.permute(0, 2, 1), norm_grad[0]) @ B.permute(0, 2, 1)
        projected = torch.matmul(A, low_rank_grad) @ B
        return projected.float().unsqueeze(0)
    return norm_grad
