"""Host mirrors of integer batch tensors (input_ids / labels / attention_mask).

The <image>/text splice is integer bookkeeping done on the host (splice_plan.py): it needs the ids, labels and mask as numpy arrays.  The
collator built them on the host a moment before the Trainer moved the batch to the device; pulling them back (`.cpu()`) costs a
device -> host copy and, worse, a synchronisation that pins the host to the device once per step (no launch run-ahead across steps;
round 4 measured ~3.7 ms per step).  So whoever moves a batch to the device registers the host originals here, keyed by the DEVICE tensor
object (weakly: the entry dies with the tensor) and by its version counter (an in-place write invalidates it), and the model asks `host_array`
first.  CPU tensors are their own mirror.  Nothing registered: the old `.cpu()` path, still correct.

Reference: the batch contract of DataCollatorForSupervisedDataset (train.py:1258-1284) consumed at metamorph_arch.py:177-425."""
from __future__ import annotations

import weakref

import numpy as np
import torch

_MIRRORS: dict = {}                                          # id(device tensor) -> (weakref to it, its version, host array); the weakref's
                                                             # callback drops the entry (a WeakKeyDictionary would compare tensors with ==)
_WARNED = False
STATS = {"mirror": 0, "cpu": 0, "sync": 0}                   # how host_array was served (tests / bench assert on `sync`)


def attach(device_tensor: torch.Tensor, host) -> torch.Tensor:
    """Remember `host` (a CPU tensor or numpy array with the same shape and values) as the host copy of `device_tensor`."""
    # a COPY: the caller may reuse or edit its host buffer (collator / pinned staging buffers, in-place label edits) after the move, and the
    # plan must be built from the values the device tensor holds (these are [B, L] integer arrays: a few tens of KB)
    arr = np.array(host.detach().numpy() if isinstance(host, torch.Tensor) else host, copy=True)
    if tuple(arr.shape) != tuple(device_tensor.shape):
        raise ValueError(f"host copy {arr.shape} does not match the device tensor {tuple(device_tensor.shape)}")
    key = id(device_tensor)
    _MIRRORS[key] = (weakref.ref(device_tensor, lambda _r, key=key: _MIRRORS.pop(key, None)), device_tensor._version, arr)
    return device_tensor


def to_device(t: torch.Tensor, device, **kw) -> torch.Tensor:
    """t.to(device) that keeps the host original as the mirror of the result (integer / bool batch tensors)."""
    out = t.to(device, **kw)
    if t.device.type == "cpu" and out.device.type != "cpu" and out.dtype == t.dtype:
        attach(out, t)
    return out


def host_array(t: torch.Tensor | None) -> np.ndarray | None:
    """The values of `t` as a numpy array WITHOUT a device synchronisation when a mirror is known."""
    if t is None:
        return None
    if t.device.type == "cpu":
        STATS["cpu"] += 1
        return t.detach().numpy()
    hit = _MIRRORS.get(id(t))
    if hit is not None and hit[0]() is t and hit[1] == t._version:
        STATS["mirror"] += 1
        return hit[2]
    STATS["sync"] += 1
    global _WARNED
    if not _WARNED:
        _WARNED = True
        import warnings
        warnings.warn("metamorph_amd: a batch tensor reached the model without a host mirror -- its values are copied back from the device "
                      "(one synchronisation per step, ~4 ms at 8B).  Move batches with metamorph_amd.hostmirror.to_device(t, device), or "
                      "use metamorph_amd.trainer.MetaMorphTrainer, which registers the collator's host tensors itself.  (Shown once.)",
                      RuntimeWarning, stacklevel=3)
    return t.detach().cpu().numpy()
