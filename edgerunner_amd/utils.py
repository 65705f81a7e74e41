"""Host-side helpers of the decode path that the reference keeps in core/utils.py."""
from __future__ import annotations

import random
from typing import Sequence

import numpy as np

# upper bounds of the face-count buckets 1..4 (bucket 0 = unconditional, 5 = above the last bound)
_FACE_BOUNDS = (1000, 2000, 4000, 8000)


def quantize_num_faces(n):
    """Face-count bucket fed to ``embed_num_face`` (reference: core/utils.py:89-116).

    int -> int; sequence / numpy / torch tensor -> same container type of buckets:
    n<=0 -> 0, (0,1000] -> 1, (1000,2000] -> 2, (2000,4000] -> 3, (4000,8000] -> 4, >8000 -> 5.
    """
    if isinstance(n, (int, np.integer)):
        if n <= 0:
            return 0
        return int(np.searchsorted(_FACE_BOUNDS, n, side="left")) + 1
    try:
        import torch
        if isinstance(n, torch.Tensor):
            bounds = torch.tensor(_FACE_BOUNDS, device=n.device, dtype=n.dtype)
            b = torch.searchsorted(bounds, n.contiguous(), right=False) + 1
            return torch.where(n <= 0, torch.zeros_like(b), b).to(n.dtype)
    except ImportError:  # pragma: no cover
        pass
    arr = np.asarray(n)
    b = np.searchsorted(_FACE_BOUNDS, arr, side="left") + 1
    return np.where(arr <= 0, 0, b).astype(arr.dtype)


def seed_everything(seed: int) -> None:
    """What ``kiui.seed_everything`` does for the reference's infer.py:38."""
    import torch
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def trim_tokens(tokens: np.ndarray) -> np.ndarray:
    """Post-processing of infer.py:113-116: cut at the first EOS (2) and shift by -3,
    giving the ``*_tokens.npy`` on-disk format (meto token ids)."""
    tokens = np.asarray(tokens)
    eos = np.nonzero(tokens == 2)[0]
    if len(eos) > 0:
        tokens = tokens[: eos[0]]
    return tokens - 3
