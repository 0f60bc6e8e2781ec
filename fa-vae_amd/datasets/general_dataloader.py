"""Input pipeline of the FA-VAE training script -- drop-in for the reference's `datasets/general_dataloader.py`
(`GeneralDataset` :20-70, `load_data` :73-86; imported by favae_scripts/train_favae.py:20, called at :286).

Same constructor / `load_data(args)` signature and the same per-sample result: a (3, R, R) fp32 tensor in [-1, 1] =
Normalize(0.5, 0.5)(ToTensor(CenterCrop(R)(Resize((R, R))(RGB image)))).  torchvision is not a dependency here: on a PIL image
`T.Resize((R, R))` is `Image.resize((R, R), BILINEAR)`, `CenterCrop(R)` of an R x R image is the identity, and ToTensor /
Normalize are `u/255`, `(t - 0.5)/0.5` in fp32 -- restated with the same operations (parity of that tail is bit-exact and
tested; the torchvision call chain itself cannot be imported in this image: "unpinned" in DESIGN.md).

MI355X-first addition (opt-in, `device_normalize=True` / `args.device_normalize`): the host workers stop at the resized uint8
HWC image, the batch crosses PCIe as bytes (196 KB instead of 786 KB per 256x256 image) and `to_device_batch()` finishes
ToTensor + Normalize in one HIP kernel (`favae_u8_to_float_nhwc`) that writes the channels-last layout the convs read.
"""
import pickle as pk

import numpy as np
import torch
from PIL import Image, ImageFile

from .statistic import *  # noqa: F401,F403  (mean, std, clip_mean, clip_std: reference datasets/general_dataloader.py:15)

ImageFile.LOAD_TRUNCATED_IMAGES = True            # reference :17

_NORM_MEAN = (0.5, 0.5, 0.5)                      # T.Normalize((0.5,)*3, (0.5,)*3), reference :37
_NORM_STD = (0.5, 0.5, 0.5)


def resize_rgb_u8(img, resolution):
    """PIL RGB image -> (R, R, 3) uint8 array: T.Resize((R, R)) [PIL bilinear] + T.CenterCrop(R) [identity] (reference :34-35)."""
    if img.size != (resolution, resolution):
        img = img.resize((resolution, resolution), Image.BILINEAR)
    return np.asarray(img, dtype=np.uint8)


def normalize_u8(arr_u8, mean=_NORM_MEAN, std=_NORM_STD):
    """(H, W, 3) uint8 -> (3, H, W) fp32: T.ToTensor() then T.Normalize(mean, std) (reference :36-37), same fp32 operations."""
    t = torch.from_numpy(np.array(arr_u8, dtype=np.uint8)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    m = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)
    return t.sub_(m).div_(s)


class GeneralDataset(torch.utils.data.Dataset):
    """Characterizes a dataset for PyTorch (reference :20-70): `train_file` / `test_file` = pickled list of image paths."""

    def __init__(self, resolution, train=True, val=False, train_file=None, test_file=None, device_normalize=False):
        if train:
            with open(train_file, "rb") as input_file:
                self.names_dict = pk.load(input_file)
        if val:
            with open(test_file, "rb") as input_file:
                self.names_dict = pk.load(input_file)
        self.resolution = resolution
        self.device_normalize = device_normalize
        self.transform = lambda img: normalize_u8(resize_rgb_u8(img, resolution))

    def __len__(self):
        return len(self.names_dict)

    def load_image(self, name):
        try:
            image = Image.open(name)
            if not image.mode == "RGB":
                image = image.convert("RGB")
            return image
        except Exception:           # unreadable file: the caller moves on to the next index (reference :57-58, 66-67)
            return None

    def __getitem__(self, index):
        name = self.names_dict[index]
        img = self.load_image(name)
        if img is None:
            return self.__getitem__(index + 1)
        if self.device_normalize:
            return torch.from_numpy(resize_rgb_u8(img, self.resolution).copy())      # (R, R, 3) uint8
        return self.transform(img)


def to_device_batch(batch, device, non_blocking=True):
    """One batch of the loader -> the (B, 3, R, R) fp32 tensor `train()` feeds the model (train_favae.py:71 `x = x.to(device)`).
    Float batches are moved as they are; uint8 HWC batches (device_normalize=True) are moved as bytes and normalised by the HIP
    kernel on the current stream."""
    if batch.dtype == torch.uint8:
        from favae_hip import ops as K
        return K.u8_to_float(batch.to(device, non_blocking=non_blocking).contiguous(), _NORM_MEAN, _NORM_STD)
    return batch.to(device, non_blocking=non_blocking)


def load_data(args):
    """(train_loader, test_loader) for `args.train_file` / `args.test_file` (either may be None), reference :73-86."""
    dn = bool(getattr(args, "device_normalize", False))
    pin = dn and torch.cuda.is_available()

    def make(list_file, is_train):
        if list_file is None:
            return None
        ds = GeneralDataset(resolution=args.resolution, train=is_train, val=not is_train,
                            train_file=list_file if is_train else None, test_file=None if is_train else list_file,
                            device_normalize=dn)
        dl = torch.utils.data.DataLoader(ds, batch_size=args.batch_size, shuffle=is_train, num_workers=args.num_workers,
                                         pin_memory=pin)
        print("\nLoaded the {} set length {}, dataloader length {}".format("train" if is_train else "test", len(ds), len(dl)))
        return dl

    return make(args.train_file, True), make(args.test_file, False)
