"""Dataset registry of the driver (reference: data/dataloader.py): `ffhq` = sorted recursive *.png glob -> RGB ->
transform.  No torchvision: the ToTensor + Normalize((.5,.5,.5),(.5,.5,.5)) pair of the reference's driver
(sample_condition_batched_ttc.py:125-126) is `to_minus1_1`.  Once per image, off the hot path."""
from glob import glob

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

__DATASET__ = {}


def register_dataset(name: str):
    def wrapper(cls):
        if __DATASET__.get(name, None):
            raise NameError(f"Name {name} is already registered!")
        __DATASET__[name] = cls
        return cls
    return wrapper


def get_dataset(name: str, root: str, **kwargs):
    if __DATASET__.get(name, None) is None:
        raise NameError(f"Dataset {name} is not defined.")
    return __DATASET__[name](root=root, **kwargs)


def get_dataloader(dataset, batch_size: int, num_workers: int, train: bool):
    return DataLoader(dataset, batch_size, shuffle=train, num_workers=num_workers, drop_last=train)


def to_minus1_1(img):
    """PIL RGB image -> float32 [3, H, W] in [-1, 1]"""
    a = np.asarray(img, dtype=np.float32) / 255.0
    return torch.from_numpy(a).permute(2, 0, 1).contiguous() * 2.0 - 1.0


@register_dataset(name='ffhq')
class FFHQDataset(Dataset):
    def __init__(self, root: str, transforms=None):
        self.root, self.transforms = root, transforms
        self.fpaths = sorted(glob(root + '/**/*.png', recursive=True))
        assert len(self.fpaths) > 0, "File list is empty. Check the root."

    def __len__(self):
        return len(self.fpaths)

    def __getitem__(self, index: int):
        from PIL import Image
        img = Image.open(self.fpaths[index]).convert('RGB')
        return self.transforms(img) if self.transforms is not None else to_minus1_1(img)
