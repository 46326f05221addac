"""Parity-margin ledger (VERDICT r02 'weak' 1a): every fp32 / fp64 parity comparison of the -m gpu tests records
how much of its tolerance it used -- err/scale of the engine against the fp64 reference, floor/scale of the reference's
own fp32 run against it, the relative tolerance asserted -- and conftest.py writes the ledger to
gpurun_out/parity_margins.json at session end (copied to profiles/r03_parity_margins.json for the record)."""
import json
import os

_LEDGER = []
ASSERT = True    # False: `check` records without asserting (seed sweeps that judge the DISTRIBUTION: median_ratio_over_seeds)
TAG = None       # tools/margin_distributions.py: (seed, kernel) of the run an entry belongs to


def seed_offset():
    """MPPI_MARGIN_SEED (default 0): added to the seeds of the full-size parity scenarios (initial state, nominal sequence,
    generator key) so that tools/margin_distributions.py can re-run them on other draws -- VERDICT r05 next #2: a margin
    measured on one seed is a sample of one"""
    return int(os.environ.get("MPPI_MARGIN_SEED", "0"))


def record(test, quantity, err_over_scale, floor_over_scale=None, rtol=None, note=None, scale=None):
    _LEDGER.append({"test": test, "quantity": quantity, "err_over_scale": float(err_over_scale),
                    "floor_over_scale": None if floor_over_scale is None else float(floor_over_scale),
                    "rtol": rtol, "note": note, "scale": None if scale is None else float(scale),
                    **({"tag": TAG} if TAG is not None else {})})


BUDGET = 1.5      # of the reference's own fp32-vs-fp64 error, for entries above rtol (tests/test_zz_margin_budget.py)


def over_budget(ledger=None, budget=BUDGET):
    """SURVEY 7.3 lets an entry above rtol pass at up to 2 x the reference's own fp32 error; the suite keeps
    its entries below `budget` x that floor, so that a change drifting towards 2 x is noticed while there is still room.
    The floor of an `action` entry is the larger of its own and that of the sequence it is the first row of (`U`, same test and
    step): the returned action IS U[0] (mppi.py:271-275), its error is bounded by U's, and the reference's fp32 error on those nu
    numbers alone is a sample of one.  Round 6 put seed distributions behind that (profiles/r06_margin_distributions.txt, 32 draws
    per scenario, tools/margin_distributions.py): err / OWN floor of an `action` entry has median ~1.0-1.3 like every other
    quantity but p95 3-15 and maxima up to 30 -- with the EXACT-fp32 kernel as much as with the split-operand one -- while `U` of
    the same runs stays at p95 ~1.7-2.2: the own floor of nu numbers is not a yardstick, U's is.  The same table is why this is a
    budget on the suite's FIXED seeds and no more: for the ill-conditioned scenarios (peaked softmax, SMPPI's 1/dt, the pendulum)
    the ratio is a random variable of median ~1 whose p95 is ~2-3 for ANY fp32 implementation, the reference's own included; what
    is asserted about it across seeds is its median (`median_ratio_over_seeds`, tests/test_gpu_fullsize_parity.py).
    Returns [(ratio, entry, floor used)] of the entries over budget."""
    ledger = [e for e in _LEDGER if not e.get("tag")] if ledger is None else ledger     # (seed sweeps are judged as distributions)
    by_key = {(e["test"], e["quantity"]): e for e in ledger}
    bad = []
    for e in ledger:
        f, rt = e.get("floor_over_scale"), e.get("rtol")
        if not f or rt is None or e["err_over_scale"] <= rt:
            continue
        floor = f
        if "action" in e["quantity"] and e.get("scale"):
            sib = by_key.get((e["test"], e["quantity"].replace("action_sequence", "\0").replace("action", "U").replace("\0", "action_sequence")))
            if sib is not None and sib.get("scale") and sib.get("floor_over_scale") and "action_sequence" not in e["quantity"]:
                floor = max(floor, sib["floor_over_scale"] * sib["scale"] / e["scale"])
        ratio = e["err_over_scale"] / floor
        if ratio > budget:
            bad.append((ratio, e, floor))
    return sorted(bad, key=lambda t: -t[0])


def check(test, quantity, got, ref64, ref32=None, rtol=1e-5, scale_floor=0.0, note=None, floor_factor=2.0):
    """SURVEY 7.3 criterion on one quantity: err(engine vs ref64) <= max(rtol * scale, 2 * err(ref32 vs ref64)), scale =
    max |ref64| (optionally floored), recorded in the ledger whether it passes or not.  Returns (err/scale, floor/scale)."""
    import numpy as np
    g = np.asarray(got, dtype=np.float64)
    r = np.asarray(ref64, dtype=np.float64)
    scale = max(float(np.abs(r).max()), scale_floor) or 1.0
    err = float(np.abs(g - r).max())
    floor = float(np.abs(np.asarray(ref32, dtype=np.float64) - r).max()) if ref32 is not None else 0.0
    record(test, quantity, err / scale, floor / scale if ref32 is not None else None, rtol, note, scale)
    assert not ASSERT or err <= max(rtol * scale, floor_factor * floor), (test, quantity, "err/scale", err / scale, "ref32 floor/scale", floor / scale, "rtol", rtol,
                                                            "floor factor", floor_factor)
    return err / scale, floor / scale


def median_ratio_over_seeds(entries):
    """{quantity: (median of err / own floor, worst err / scale, seeds)} of ledger entries that differ only in the seed"""
    import numpy as np
    by = {}
    for e in entries:
        if e.get("floor_over_scale"):
            by.setdefault(e["quantity"], []).append((e["err_over_scale"] / e["floor_over_scale"], e["err_over_scale"]))
    return {q: (float(np.median([r for r, _ in v])), float(max(x for _, x in v)), len(v)) for q, v in by.items()}


def dump(path=None):
    if not _LEDGER:
        return None
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = path or os.path.join(root, "gpurun_out", "parity_margins.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    worst = {}
    for e in _LEDGER:
        key = e["test"].split("[")[0] + " :: " + e["quantity"]
        w = worst.get(key)
        if w is None or e["err_over_scale"] > w["err_over_scale"]:
            worst[key] = e
    json.dump({"criterion": "err(engine vs fp64 reference) <= max(rtol * scale, 2 * err(reference fp32 vs fp64)); scale = max |fp64 reference|",
               "entries": len(_LEDGER), "worst_per_test_and_quantity": worst, "all": _LEDGER}, open(path, "w"), indent=1)
    return path
