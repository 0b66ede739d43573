import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle_py import ORACLE_SO, Oracle
    if not os.path.exists(ORACLE_SO):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "librt_oracle.so"])
    return Oracle()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
