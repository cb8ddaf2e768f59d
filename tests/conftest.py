import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = 'single-speaker-tts_amd'


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def pkg(sub=None):
    return importlib.import_module(PKG + ('.' + sub if sub else ''))


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


@pytest.fixture(scope='session')
def hparams():
    return pkg('tacotron.params').ModelParams()


@pytest.fixture(scope='session')
def weights(hparams):
    """Seeded synthetic weights of the reference architecture, float32."""
    return pkg('tacotron.weights').synthetic_weights(0, hparams)


@pytest.fixture(scope='session')
def weights64(weights):
    return {k: v.astype(np.float64) for k, v in weights.items()}


@pytest.fixture(scope='session')
def engine(hparams, weights):
    eng = pkg().Engine(hparams)
    eng.load_weights(weights)
    yield eng
    eng.close()
