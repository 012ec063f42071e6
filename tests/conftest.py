import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle.oracle import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle import oracle

    if not oracle.have_ref() and not os.path.exists("/root/reference/lib_rspt/signal_packer.h"):
        pytest.skip("oracle/_ref not built and /root/reference absent")
    return oracle.Ref()


@pytest.fixture(scope="session")
def golden():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def packer_cases():
    import cases

    return {c["name"]: c for c in cases.packer_cases()}
