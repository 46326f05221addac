"""From a plain torch callable to a device functor.

The reference's plugin API is "any Python callable on batched tensors" (/root/reference/src/pytorch_mppi/mppi.py:63-64,
:147-154, :314, :318, :325).  `jit.compile_model` fuses user dynamics when the formulas are given once more as C++
snippets; this module WRITES those snippets: it runs the callable once on symbolic per-sample tensors (batch of one,
every element an expression node), records the arithmetic, and prints it as the `step` / `cost` / `terminal` bodies
`jit.compile_model` compiles around csrc/rollout.hpp.

Symbolic tensors are NOT torch.Tensor subclasses: `SymT` implements the tensor methods users call on states and actions
(indexing, views, arithmetic, reductions, elementwise maths), `__torch_function__` for the torch.* / torch.nn.functional
entry points (torch.cat, torch.clamp, F.linear through an nn.Module, ...) and `__array_ufunc__` for numpy ufuncs applied
to tensors (the reference's own pendulum uses np.sin / np.clip on tensors: tests/pendulum.py:45-46).  Shapes are
concrete (numpy arrays of node ids with a batch axis of size 1), so every view / broadcast / concatenation is numpy's.

Scope: elementwise maths, + - * / ** %, clamp / where / min / max, small constant matrices (captured tensors, nn.Linear
weights), cat / stack / slicing / views, sum / mean / prod over non-batch axes.  Anything else -- data-dependent control
flow, integer indexing by values, in-place writes into the inputs, ops not listed -- raises `TraceUnsupported`, and
the caller keeps the generic path.  The traced functor is VERIFIED before use: compiled for the host (g++) and compared
with the callable on random batches (1e-9 relative, fp64).

The tracer lives in four modules (round 6; this one re-exports what callers use): `trace_graph.py` (the expression graph, the
exceptions, run-time parameter places), `trace_sym.py` (the symbolic tensors), `trace_emit.py` (tracing the callables, code
generation, the MLP matcher), `trace_host.py` (the host check)."""
from .trace_graph import DENSE_MIN, Graph, PathParam, StaleTrace, TraceUnsupported, param_tensor            # noqa: F401
from .trace_sym import SymS, SymT                                                                            # noqa: F401
from .trace_emit import (emit, gather_params, generate, layer_members, match_mlp_residual, same_functor,     # noqa: F401
                         same_param_sources, trace_callables)
from .trace_host import evaluate_on_host, source_key, verify_on_host                                         # noqa: F401
