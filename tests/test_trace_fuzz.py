"""Differential fuzzing of pytorch_mppi_amd/trace.py: random compositions of the traceable operations (depth 4, three state
outputs and a cost) are translated and compared with the torch callable on the host.  A translation may be REFUSED for
numerical reasons (an ill-conditioned composition such as fmod(exp(x), 0.9) far from the origin) -- it must never fail to
compile or raise anything but TraceUnsupported, and most programs must go through.  (The first run of this harness found a
protective copy declared twice when two outputs are the same input component.)"""
import math
import random

import pytest
import torch

from pytorch_mppi_amd import trace

UN = [torch.sin, torch.cos, torch.tanh, torch.exp, lambda x: torch.log(1 + x * x), lambda x: torch.sqrt(1 + x * x), torch.abs,
      torch.neg, torch.sigmoid, torch.relu, lambda x: x * x, lambda x: x ** 3, lambda x: x ** 0.5 if False else torch.sqrt(x.abs() + 0.1),
      lambda x: x.clamp(-0.7, 1.3), lambda x: x.clamp(min=-0.2), lambda x: x.clamp(max=0.4), torch.floor, torch.sign, torch.atan,
      lambda x: torch.where(x > 0.1, x, -2 * x), lambda x: x / 4.0, lambda x: x / 3.0, lambda x: 0.0 - x, lambda x: x - 0.0, lambda x: x * 1.0,
      lambda x: x * -1.0, lambda x: 1.0 / (2.0 + x * x), lambda x: x % 0.7, lambda x: torch.fmod(x, 0.9), lambda x: x + 0.0, lambda x: 0 * x + 1.5,
      lambda x: x ** 2.0, lambda x: x ** 1, lambda x: x ** 0, lambda x: (x.abs() + 0.1) ** 1.7, lambda x: 2.0 ** x.clamp(-3, 3)]
BIN = [torch.add, torch.sub, torch.mul, lambda a, b: a / (1.5 + b * b), torch.maximum, torch.minimum, lambda a, b: torch.atan2(a, 1.0 + b * b),
       lambda a, b: torch.where(a < b, a, b * 0.5), lambda a, b: a - b, lambda a, b: b - a, lambda a, b: a * b + a, lambda a, b: (a > b).double() * 1.0 + (a <= b).double() * 2.0]

def rand_expr(rng, depth, leaves):
    if depth == 0 or rng.random() < 0.15:
        k = rng.randrange(len(leaves) + 2)
        if k >= len(leaves):
            c = rng.choice([0.0, 1.0, -1.0, 0.5, 2.0, 0.25, -3.0, 1e-3, math.pi])
            return lambda s, a, c=c: s[:, 0] * 0.0 + c
        return leaves[k]
    if rng.random() < 0.5:
        f = rng.choice(UN); e = rand_expr(rng, depth - 1, leaves)
        return lambda s, a, f=f, e=e: f(e(s, a))
    f = rng.choice(BIN); e1 = rand_expr(rng, depth - 1, leaves); e2 = rand_expr(rng, depth - 1, leaves)
    return lambda s, a, f=f, e1=e1, e2=e2: f(e1(s, a), e2(s, a))

def program(seed, nx=3, nu=2, depth=4):
    rng = random.Random(seed)
    leaves = [lambda s, a, i=i: s[:, i] for i in range(nx)] + [lambda s, a, i=i: a[:, i] for i in range(nu)]
    outs = [rand_expr(rng, depth, leaves) for _ in range(nx)]
    ce = rand_expr(rng, depth, leaves)
    f = lambda s, a: torch.stack([o(s, a) for o in outs], 1)
    q = lambda s, a: ce(s, a)
    return f, q



def test_random_programs_translate_or_are_refused_for_numerical_reasons():
    ok = refused = 0
    for seed in range(48):
        f, q = program(seed)
        try:
            code = trace.generate(f, q, 3, 2)
            trace.verify_on_host(code, f, q, 3, 2)
            ok += 1
        except trace.TraceUnsupported as e:
            assert "differs from the callable" in str(e) or "which results are finite" in str(e), (seed, str(e)[:300])
            refused += 1
    assert ok >= 40, (ok, refused)
