"""TEST INFRASTRUCTURE ONLY -- loader for the LIVE reference implementation.

Imports ``pytorch_mppi`` from the read-only checkout at ``/root/reference/src`` (present in
the build container only; it does NOT exist on the GPU box) together with the
``arm_pytorch_utilities`` stand-in next to this file, and installs the z-injection seam:
the reference's only RNG call on the hot path is ``torch.randn(*shape, nu, ...)`` at
/root/reference/src/pytorch_mppi/mppi.py:203, so replacing the module-level name ``torch``
inside ``pytorch_mppi.mppi`` with a proxy whose ``randn`` pops pre-drawn tensors lets the
reference and the engine consume IDENTICAL standard-normal draws without touching reference
files.  Queue order = generator consumption order:
``randn(T,nu)`` at construction when ``U_init is None`` (mppi.py:144-145) and at ``reset()``
(:290); ``randn(K,T,nu)`` per MPPI command (:378); ``randn(K,S,nu)`` per KMPPI command (:660).

Used by ``oracle/gen_golden.py`` (fixture generation) and by the container-only tests that pin
``oracle/mppi_oracle.py`` against the real thing.  Never imported by the product package.
"""
import importlib
import os
import sys

import torch

REFERENCE_SRC = "/root/reference/src"
_HERE = os.path.dirname(os.path.abspath(__file__))


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_SRC, "pytorch_mppi", "mppi.py"))


class _TorchProxy:
    """Forwards everything to real torch except ``randn``, which serves queued tensors first."""

    def __init__(self):
        self.queue = []
        self.log = []   # shapes requested, for bookkeeping assertions

    def randn(self, *shape, **kw):
        self.log.append(tuple(shape))
        if self.queue:
            z = self.queue.pop(0)
            assert tuple(z.shape) == tuple(shape), (tuple(z.shape), tuple(shape))
            return z.to(device=kw.get("device", None), dtype=kw.get("dtype", z.dtype)).clone()
        return torch.randn(*shape, **kw)

    def __getattr__(self, name):
        return getattr(torch, name)


_loaded = None


def load_reference():
    """Returns (module pytorch_mppi.mppi, proxy).  ``proxy.queue.append(z)`` injects noise."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError("live reference not available (no /root/reference on this machine)")
    for p in (_HERE, REFERENCE_SRC):
        if p not in sys.path:
            sys.path.insert(0, p)
    mod = importlib.import_module("pytorch_mppi.mppi")
    proxy = _TorchProxy()
    mod.torch = proxy
    _loaded = (mod, proxy)
    return _loaded
