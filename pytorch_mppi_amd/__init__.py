"""MI355X-native MPPI rollout-and-update engine behind the `pytorch_mppi` API surface."""
from .mppi import MPPI, SMPPI, KMPPI, MPPI_Batched, run_mppi, SpecificActionSampler, TimeKernel, RBFKernel
from . import models, jit

__all__ = ["MPPI", "SMPPI", "KMPPI", "MPPI_Batched", "run_mppi", "SpecificActionSampler", "TimeKernel", "RBFKernel",
        "models", "jit"]
