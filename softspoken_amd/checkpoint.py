"""Checkpoint -> weights blob for ss_create.

The reference reads `torch.load(path, weights_only=True)['model_state_dict']` and `['epoch']`
(root/code/frontend/NNDetector.py:42-53).  Here the same file is read with the same safe loader and its
tensors are copied verbatim (no arithmetic) into the "SSWBLOB1" container the C ABI takes;
BatchNorm folding and MFMA packing happen inside the library (csrc/engine.hip).
"""
from __future__ import annotations

import struct

import numpy as np

MAGIC = b"SSWBLOB1"
_ENTRY = struct.Struct("<96sII4qQQ")     # name, dtype, ndim, shape[4], offset, nbytes  (152 bytes)


def pack_state_dict(sd) -> bytes:
    """sd: mapping name -> numpy array or torch tensor.  float32 tensors and int64 counters only."""
    items = []
    for k, v in sd.items():
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        a = np.asarray(v)
        if a.dtype == np.int64:
            dt = 1
        else:
            a = a.astype(np.float32, copy=False)
            dt = 0
        if a.ndim > 4:
            raise ValueError(f"{k}: rank {a.ndim} tensor not supported")
        items.append((k, dt, np.ascontiguousarray(a)))
    table = 16 + len(items) * _ENTRY.size
    off = (table + 15) & ~15
    head = bytearray(MAGIC + struct.pack("<II", len(items), 0))
    blobs = []
    for k, dt, a in items:
        name = k.encode()
        if len(name) > 95:
            raise ValueError(f"tensor name too long: {k}")
        shape = list(a.shape) + [0] * (4 - a.ndim)
        head += _ENTRY.pack(name, dt, a.ndim, *shape, off, a.nbytes)
        blobs.append((off, a.tobytes()))
        off = (off + a.nbytes + 15) & ~15
    out = bytearray(off)
    out[:len(head)] = head
    for o, b in blobs:
        out[o:o + len(b)] = b
    return bytes(out)


def load_checkpoint_file(path, map_location="cpu"):
    """-> (state_dict, epoch).  Safe loader only, as the reference (weights_only=True)."""
    import torch
    ck = torch.load(path, map_location=map_location, weights_only=True)
    return ck["model_state_dict"], int(ck.get("epoch", -1))
