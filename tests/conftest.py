import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


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
