"""Harness helpers the reference training script imports (reference utils.py:108-119)."""
import os

import torch


def save_model(state, path):
    """Atomic checkpoint write: *.tmp then rename, keeping the previous file as *.old until success."""
    tmp, old = path + ".tmp", path + ".old"
    torch.save(state, tmp)
    if os.path.exists(path):
        os.replace(path, old)
    os.replace(tmp, path)
    if os.path.exists(old):
        os.remove(old)
