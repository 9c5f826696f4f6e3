"""Seeded, portable synthetic weights and inputs (numpy PCG64: the same tensors on every box).

Used by bench.py, the CLI's ``--synthetic`` mode and tools/: SURVEY 8(d) prescribes synthetic 256x256 uint8 tiles
(src/histopathology_gan.py:106-109 turns them into floats in [-1, 1]), N(0,1) RNA rows with 16 distinct profiles per
batch (src/histopathology_gan.py:148-151: StandardScaler output; tiles of a slide share one profile) and seeded
kaiming-scale weights.  oracle/ref_cpu.py keeps its own copy of these generators (the oracle is self-contained test
infrastructure); tests/test_synth_cpu.py checks that the two agree bit for bit.
"""
from __future__ import annotations

import math
import zlib
from typing import Sequence

import numpy as np
import torch
import torch.nn as nn


def _name_seed(seed: int, name: str):
    return [int(seed) & 0x7FFFFFFF, zlib.crc32(name.encode("utf-8"))]


def seeded_tensor(name: str, shape: Sequence[int], seed: int) -> torch.Tensor:
    """One tensor of the seeded weight generator, seeded per tensor NAME so that the value does not depend on the
    module traversal order: running_var 1 + |N(0, .1)|, running_mean N(0, .1), BN gamma 1 + N(0, .1), other 1-D
    N(0, .05), weights N(0, 2 / fan_in) with fan_in = prod(shape[1:])."""
    rng = np.random.default_rng(_name_seed(seed, name))
    shape = tuple(int(s) for s in shape)
    if name.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.int64)
    if name.endswith("running_var"):
        v = 1.0 + np.abs(rng.normal(0.0, 0.1, size=shape))
    elif name.endswith("running_mean"):
        v = rng.normal(0.0, 0.1, size=shape)
    elif len(shape) == 1 and name.endswith("weight"):
        v = 1.0 + rng.normal(0.0, 0.1, size=shape)
    elif len(shape) <= 1:
        v = rng.normal(0.0, 0.05, size=shape)
    else:
        fan_in = int(np.prod(shape[1:]))
        v = rng.standard_normal(size=shape, dtype=np.float32) * np.float32(math.sqrt(2.0 / fan_in))
    return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))


def seeded_fill_(module: nn.Module, seed: int) -> nn.Module:
    """Fill every parameter and buffer of ``module`` in place with ``seeded_tensor``."""
    with torch.no_grad():
        for name, t in list(module.named_parameters()) + list(module.named_buffers()):
            t.copy_(seeded_tensor(name, t.shape, seed))
    return module


def synthetic_tiles_u8(n: int, size: int, seed: int, channels: int = 3) -> torch.Tensor:
    """uint8 CHW tiles, uniform in [0, 255]: what the reference's dataset hands to its transform."""
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.integers(0, 256, size=(n, channels, size, size), dtype=np.uint8))


def synthetic_images(n: int, size: int, seed: int, channels: int = 3) -> torch.Tensor:
    """uint8 uniform tiles -> /255 -> (x - 0.5) / 0.5 on the host (src/histopathology_gan.py:106-109)."""
    return (synthetic_tiles_u8(n, size, seed, channels).float() / 255.0 - 0.5) / 0.5


def synthetic_rna(n: int, features: int, seed: int, distinct: int = 16) -> torch.Tensor:
    rng = np.random.default_rng(seed)
    rows = rng.normal(0.0, 1.0, size=(min(distinct, n), features)).astype(np.float32)
    idx = np.arange(n) % rows.shape[0]
    return torch.from_numpy(rows[idx])


def synthetic_uniform(n: int, dims: int, seed: int, lo=-0.3, hi=0.3) -> torch.Tensor:
    """The uniform draw of a train_op (src/wgan_loss.py:100), seeded: same values as the oracle's generator."""
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.uniform(lo, hi, size=(n, dims)).astype(np.float32))
