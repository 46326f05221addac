"""Runs LAST in the -m gpu suite (file order): the parity-margin ledger the other tests filled (tests/margins.py) against its
budget -- no entry above rtol may use more than 1.5 x the reference's own fp32-vs-fp64 error (SURVEY 7.3 allows 2 x: the budget
is there so that a change drifting towards the limit is seen while there is still room)."""
import pytest

import margins

pytestmark = pytest.mark.gpu


def test_no_parity_entry_uses_more_than_its_budget():
    if not margins._LEDGER:
        pytest.skip("no parity comparison ran in this session")
    bad = margins.over_budget()
    assert not bad, "\n".join(f"x{r:.2f} of the floor ({fl:.3g} of scale): {e['test']} :: {e['quantity']} err {e['err_over_scale']:.3g}"
                              for r, e, fl in bad[:20])


def test_budget_rule_on_a_synthetic_ledger():
    L = [dict(test="t", quantity="step 1 action", err_over_scale=3.0e-5, floor_over_scale=1.75e-5, rtol=1e-5, scale=1.0),
         dict(test="t", quantity="step 1 U", err_over_scale=5.0e-5, floor_over_scale=4.0e-5, rtol=1e-5, scale=1.2),
         dict(test="t", quantity="cost_total", err_over_scale=3.2e-5, floor_over_scale=2.0e-5, rtol=1e-5, scale=10.0),
         dict(test="t", quantity="omega", err_over_scale=0.9e-5, floor_over_scale=1.0e-6, rtol=1e-5, scale=1.0),
         dict(test="u", quantity="action", err_over_scale=3.0e-5, floor_over_scale=1.75e-5, rtol=1e-5, scale=1.0)]
    bad = margins.over_budget(L)
    assert [(e["test"], e["quantity"]) for _, e, _ in bad] == [("u", "action"), ("t", "cost_total")]      # 1.71 x (no sibling), 1.6 x
    assert abs(bad[0][0] - 3.0 / 1.75) < 1e-9
