import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def small_case(tmp_path_factory):
    from bart_amd import synth
    d = tmp_path_factory.mktemp("case_small")
    return synth.make_case(str(d), nlayers=100, nwave=777)


@pytest.fixture(scope="session")
def demo_case(tmp_path_factory):
    """Demo shape: one molecule (CH4), 2501 samples (2-4 um at 1 cm-1)."""
    from bart_amd import synth
    d = tmp_path_factory.mktemp("case_demo")
    return synth.make_case(str(d), nlayers=100, nwave=2501, wnlow=2500.0,
                           opmol=("CH4",), seed=7)
