import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import microbecensus_amd  # noqa: E402
microbecensus_amd.configure_process_env()     # the test process is ours (child processes inherit it): see the function's docstring


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def repo():
    return REPO


@pytest.fixture(scope="session")
def oracle_bin(repo):
    """Path of the oracle CLI (oracle/rs_port), built on demand with gcc."""
    exe = os.path.join(repo, "oracle", "rs_port")
    src = os.path.join(repo, "oracle", "rapsearch_port.c")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", os.path.join(repo, "oracle"), "rs_port", "librapsearch_port.so"])
    return exe


@pytest.fixture(scope="session")
def ref_dir(repo):
    """oracle/_ref (bundled RAPsearch2 binary + rebuilt marker DB); skip when it was never built."""
    d = os.path.join(repo, "oracle", "_ref")
    if not os.path.isfile(os.path.join(d, "rapdb_2.15")):
        pytest.skip("oracle/_ref not built (needs /root/reference once: python oracle/build_ref.py)")
    return d
