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


ORACLE_PIECES = ("oracle/_ref/rapdb_2.15", "oracle/_ref/rapdb_2.15.info", "oracle/_ref/rapsearch_Linux_2.15", "oracle/rs_port", "oracle/librapsearch_port.so")


@pytest.hookimpl(tryfirst=True)
def pytest_runtest_setup(item):
    """Under -m gpu the checker must be there: a GPU test that cannot find the oracle (oracle/_ref, built once where /root/reference
    exists and carried to the GPU box; oracle/rs_port and librapsearch_port.so, built by __graft_entry__.build()) FAILS - it never
    skips, and it never passes without having compared anything."""
    if item.get_closest_marker("gpu") is None:
        return
    missing = [p for p in ORACLE_PIECES if not os.path.exists(os.path.join(REPO, p))]
    if missing:
        pytest.fail("the oracle is not built - %s missing: run __graft_entry__.build() where /root/reference exists; GPU parity tests do not run without their checker" % ", ".join(missing), pytrace=False)


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
