"""TEST INFRASTRUCTURE ONLY -- see package docstring.  ``ensure_tensor`` is imported by the
reference's autotune module (/root/reference/src/pytorch_mppi/autotune.py:8, used :154,183)."""
import torch


def ensure_tensor(device, dtype, *args):
    out = tuple(a.to(device=device, dtype=dtype) if torch.is_tensor(a)
                else torch.tensor(a, device=device, dtype=dtype) for a in args)
    return out if len(out) > 1 else out[0]
