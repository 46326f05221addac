import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# plain torch callables stay on the generic (callback) path in the suite unless a test asks for the tracer (auto_jit=True):
# the generic path has its own tests, and a traced model costs a hipcc run where no prebuilt object travels with the snapshot
os.environ.setdefault("MPPI_AUTO_JIT", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs the live reference checkout at /root/reference")


def pytest_sessionfinish(session, exitstatus):
    # parity-margin ledger of the -m gpu tests (tests/margins.py) -> gpurun_out/parity_margins.json
    try:
        import margins
        margins.dump()
    except Exception:
        pass
