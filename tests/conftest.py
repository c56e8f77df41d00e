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
    """The CPU oracle (test infrastructure, oracle/)."""
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def p128_keys(oracle):
    """(product keyset on the device, oracle keyset) regenerated from one seed -- the
    keys are 30 MB + 83 MB, so fixtures hold seeds and digests, never keys."""
    from peba1_amd import api
    seed = 0x5EBA2
    pp = api.ParameterSet(128)
    ks = api.SecretKeySet(pp, seed, device=True)
    oks = oracle.KeySet(oracle.params("P128"), seed)
    yield pp, ks, oks
    ks.close()
