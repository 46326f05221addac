"""TEST INFRASTRUCTURE ONLY -- minimal stand-in for the un-vendored dependency
``arm-pytorch-utilities>=0.4`` (/root/reference/pyproject.toml:57) so that the live
reference can be imported as an oracle inside the build container.

Only ``handle_batch_input`` is needed by the reference hot path's module import
(/root/reference/src/pytorch_mppi/mppi.py:7,221,225); it is OFF the rollout path (the
T-loop calls ``_dynamics_fn`` directly, mppi.py:314,318) and is reached only from
``get_rollouts`` (mppi.py:443).  Behaviour is pinned by the reference's own
/root/reference/tests/test_batch_wrapper.py:19-47 (re-run in tests/test_oracle_reference.py).

Nothing under ``pytorch_mppi_amd/`` imports this package.
"""
import functools

import torch


def handle_batch_input(n):
    """Decorator: make a function written for n-dimensional tensors accept extra leading
    batch dimensions (flattened into one, restored on return) or fewer (unsqueezed)."""

    def decorator(fn):
        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            lead = None     # leading batch shape to restore
            missing = 0     # singleton dims that had to be prepended
            for a in args:
                if torch.is_tensor(a):
                    if a.dim() > n:
                        lead = a.shape[:a.dim() - (n - 1)]
                        break
                    if 0 < a.dim() < n:
                        missing = max(missing, n - a.dim())
            if lead is not None:
                new_args = [a.reshape(-1, *a.shape[a.dim() - (n - 1):])
                            if torch.is_tensor(a) and a.dim() > 2 and a.dim() >= n else a
                            for a in args]
                ret = fn(*new_args, **kwargs)

                def restore(r):
                    if not torch.is_tensor(r):
                        return r
                    if r.dim() == n:
                        return r.reshape(*lead, *r.shape[-(n - 1):])
                    return r.reshape(*lead)

                if isinstance(ret, tuple):
                    return tuple(restore(r) for r in ret)
                return restore(ret)
            if missing:
                new_args = [a.reshape(*([1] * missing), *a.shape)
                            if torch.is_tensor(a) and a.dim() > 0 else a for a in args]
                ret = fn(*new_args, **kwargs)

                def squeeze(r):
                    if not torch.is_tensor(r):
                        return r
                    for _ in range(missing):
                        if r.dim() > 0 and r.shape[0] == 1:
                            r = r.squeeze(0)
                    return r

                if isinstance(ret, tuple):
                    return tuple(squeeze(r) for r in ret)
                return squeeze(ret)
            return fn(*args, **kwargs)

        return wrapper

    return decorator
