"""Reconstruction metrics the driver logs.  PSNR as torchmetrics' PeakSignalNoiseRatio() computes it with its
defaults (compute_metrics.py:77-90): data range = max(target) - min(target).  LPIPS / FID need downloaded networks
and stay out of scope."""
import torch


def compute_psnr(real_images, fake_images):
    real, fake = real_images.float(), fake_images.float()
    mse = torch.mean((fake - real) ** 2)
    data_range = real.max() - real.min()
    return 10.0 * torch.log10(data_range ** 2 / mse)


def compute_psnr_manual(real_images, fake_images):
    mse = torch.mean((real_images - fake_images) ** 2)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))
