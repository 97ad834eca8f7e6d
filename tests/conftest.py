"""pytest configuration: registers the ``gpu`` marker and puts the package dir
(``mvxnet-makise_amd/``, which mirrors the reference's import root) and the
oracle on sys.path."""
import os
import sys

import pytest

os.environ.setdefault('MVX_ALLOW_MUTATION', '1')      # the 1 % mutation hooks of the gradient tests (modules/_hip.py mutate)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'mvxnet-makise_amd')
for p in (PKG, os.path.join(REPO, 'oracle'), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'))
    return load
