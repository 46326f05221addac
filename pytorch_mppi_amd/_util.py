"""Small pieces shared by the controller's modules: the reference's sampler hook type and the ctypes plumbing
helpers."""
import torch

from . import _native as N

_DT = {torch.float32: N.F32, torch.float64: N.F64}


class SpecificActionSampler:
    """Same hook as mppi.py:16-32."""

    def __init__(self):
        self.start_idx = 0
        self.end_idx = 0
        self.slice = slice(0, 0)

    def sample_trajectories(self, state, info):
        raise NotImplementedError

    def specific_dynamics(self, next_state, state, action, t):
        return next_state

    def register_sample_start_end(self, start_idx, end_idx):
        self.start_idx = start_idx
        self.end_idx = end_idx
        self.slice = slice(start_idx, end_idx)


def _ptr(t):
    # a plain int is what a ctypes c_void_p field wants; no wrapper object per pointer per command
    return None if t is None else t.data_ptr()
