"""Host-side image helpers of the driver (reference: util/img_utils.py:45-63, 164-235).  numpy / once per image."""
import numpy as np
import torch


def normalize_np(img):
    """arbitrary range -> [0, 1] (min-max)"""
    img = img - np.min(img)
    mx = np.max(img)
    return img / mx if mx > 0 else img


def clear_color(x):
    if torch.is_complex(x):
        x = torch.abs(x)
    a = x.detach().float().cpu().squeeze().numpy()
    if x.shape[1] == 3:
        return normalize_np(np.transpose(a, (1, 2, 0)))
    if x.shape[1] == 1:
        return normalize_np(a)
    raise NotImplementedError


def random_sq_bbox(img, mask_shape, image_size=256, margin=(16, 16)):
    B, C, H, W = img.shape
    h, w = mask_shape
    t = np.random.randint(margin[0], image_size - margin[0] - h)
    l = np.random.randint(margin[1], image_size - margin[1] - w)
    mask = torch.ones([B, C, H, W], device=img.device)
    mask[..., t:t + h, l:l + w] = 0
    return mask, t, t + h, l, l + w


class mask_generator:
    """numpy's global RNG, as the reference (the driver seeds it with --kernel_idx)"""

    def __init__(self, mask_type, mask_len_range=None, mask_prob_range=None, image_size=256, margin=(16, 16)):
        assert mask_type in ['box', 'random', 'both', 'extreme']
        self.mask_type, self.mask_len_range, self.mask_prob_range = mask_type, mask_len_range, mask_prob_range
        self.image_size, self.margin = image_size, margin

    def _retrieve_box(self, img):
        lo, hi = int(self.mask_len_range[0]), int(self.mask_len_range[1])
        mh, mw = np.random.randint(lo, hi), np.random.randint(lo, hi)
        return random_sq_bbox(img, (mh, mw), self.image_size, self.margin)[0]

    def _retrieve_random(self, img):
        total = self.image_size ** 2
        prob = np.random.uniform(*self.mask_prob_range)
        vec = torch.ones([1, total])
        vec[:, np.random.choice(total, int(total * prob), replace=False)] = 0
        plane = vec.view(1, self.image_size, self.image_size).repeat(3, 1, 1)
        mask = torch.ones_like(img)
        mask[:, ...] = plane.to(img.device)
        return mask

    def __call__(self, img):
        if self.mask_type == 'random':
            return self._retrieve_random(img)
        if self.mask_type == 'box':
            return self._retrieve_box(img)
        if self.mask_type == 'extreme':
            return 1. - self._retrieve_box(img)
        raise NotImplementedError(self.mask_type)
