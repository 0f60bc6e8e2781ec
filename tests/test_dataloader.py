"""Input pipeline: the drop-in `datasets.general_dataloader` against a numpy/PIL restatement of the reference's
transform chain (datasets/general_dataloader.py:33-38) on synthetic image files, and the device-side ToTensor+Normalize kernel
bit-exact against the host path.  torchvision is absent from this image, so the chain is restated, not imported ("unpinned")."""
import argparse
import os
import pickle

import numpy as np
import pytest
import torch
from PIL import Image


def _make_files(tmp, n=5, seed=0):
    rng = np.random.default_rng(seed)
    names = []
    for i in range(n):
        h, w = int(rng.integers(40, 90)), int(rng.integers(40, 90))
        mode = ["RGB", "L", "RGBA", "RGB", "P"][i % 5]
        arr = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        img = Image.fromarray(arr, "RGB").convert(mode)
        p = os.path.join(tmp, "img_%d.png" % i)
        img.save(p)
        names.append(p)
    lst = os.path.join(tmp, "names.pkl")
    with open(lst, "wb") as f:
        pickle.dump(names, f)
    return names, lst


def _expected(name, res):
    """ToTensor(Resize((R,R))(img)) -> Normalize(0.5, 0.5): float64-free restatement in numpy fp32, same operation order."""
    img = Image.open(name).convert("RGB").resize((res, res), Image.BILINEAR)
    a = np.asarray(img, dtype=np.uint8).astype(np.float32) / np.float32(255)
    a = (a - np.float32(0.5)) / np.float32(0.5)
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))


def test_dataset_matches_restated_transform_chain(tmp_path):
    from datasets.general_dataloader import GeneralDataset
    names, lst = _make_files(str(tmp_path))
    ds = GeneralDataset(resolution=32, train=True, val=False, train_file=lst)
    assert len(ds) == len(names)
    for i, n in enumerate(names):
        x = ds[i]
        assert x.dtype == torch.float32 and tuple(x.shape) == (3, 32, 32)
        assert torch.equal(x, _expected(n, 32)), i
        assert float(x.min()) >= -1.0 and float(x.max()) <= 1.0


def test_unreadable_file_falls_through_to_next_index(tmp_path):
    from datasets.general_dataloader import GeneralDataset
    names, lst = _make_files(str(tmp_path), n=3)
    bad = os.path.join(str(tmp_path), "broken.png")
    open(bad, "wb").write(b"not an image")
    with open(lst, "wb") as f:
        pickle.dump([bad] + names, f)
    ds = GeneralDataset(resolution=16, train=False, val=True, test_file=lst)
    assert torch.equal(ds[0], ds[1])                      # reference :66-67: `return self.__getitem__(index+1)`


def test_load_data_surface(tmp_path, capsys):
    from datasets.general_dataloader import load_data
    names, lst = _make_files(str(tmp_path), n=5)
    args = argparse.Namespace(train_file=lst, test_file=lst, resolution=24, batch_size=2, num_workers=0)
    tr, te = load_data(args)
    assert len(tr) == 3 and len(te) == 3
    xb = next(iter(te))
    assert tuple(xb.shape) == (2, 3, 24, 24) and xb.dtype == torch.float32
    assert torch.equal(xb[0], _expected(names[0], 24))
    assert "Loaded the train set length 5, dataloader length 3" in capsys.readouterr().out
    args = argparse.Namespace(train_file=None, test_file=lst, resolution=24, batch_size=2, num_workers=0, device_normalize=True)
    tr, te = load_data(args)
    assert tr is None
    xb = next(iter(te))
    assert tuple(xb.shape) == (2, 24, 24, 3) and xb.dtype == torch.uint8


def test_statistic_constants():
    from datasets import statistic as S
    assert list(S.mean) == [0.5, 0.5, 0.5] and list(S.std) == [0.5, 0.5, 0.5]
    assert len(S.clip_mean) == 3 and len(S.clip_std) == 3


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 32, 32, 3), (1, 5, 7, 3), (3, 256, 256, 3), (2, 9, 11, 1), (1, 6, 6, 4)])
def test_device_normalize_bit_exact(shape):
    from datasets.general_dataloader import normalize_u8
    from favae_hip import ops as K
    g = torch.Generator().manual_seed(5)
    u = torch.randint(0, 256, shape, dtype=torch.uint8, generator=g)
    C = shape[-1]
    mean, std = (0.5, 0.25, 0.125, 0.75)[:C], (0.5, 0.3, 0.7, 0.9)[:C]
    y = K.u8_to_float(u.cuda(), mean, std)
    assert tuple(y.shape) == (shape[0], C, shape[1], shape[2])
    assert y.is_contiguous(memory_format=torch.channels_last) or C == 1
    ref = torch.stack([normalize_u8(u[i].numpy(), mean, std) for i in range(shape[0])])
    assert torch.equal(y.cpu(), ref)


@pytest.mark.gpu
def test_device_batch_feeds_the_model_like_the_host_batch(tmp_path):
    """uint8 batch + HIP normalise == float batch of the reference pipeline, bit for bit, as model input."""
    from datasets.general_dataloader import GeneralDataset, to_device_batch
    names, lst = _make_files(str(tmp_path), n=4)
    a = GeneralDataset(resolution=64, train=True, train_file=lst)
    b = GeneralDataset(resolution=64, train=True, train_file=lst, device_normalize=True)
    xa = torch.stack([a[i] for i in range(4)])
    xb = torch.stack([b[i] for i in range(4)])
    ya, yb = to_device_batch(xa, "cuda:0"), to_device_batch(xb, "cuda:0")
    assert yb.dtype == torch.float32 and tuple(yb.shape) == (4, 3, 64, 64)
    assert torch.equal(ya.cpu(), yb.cpu())


def test_u8_to_float_has_no_cpu_path():
    from favae_hip import ops as K
    with pytest.raises(RuntimeError):
        K.u8_to_float(torch.zeros(1, 4, 4, 3, dtype=torch.uint8))
