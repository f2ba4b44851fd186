import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import conette_amd  # noqa: E402,F401  (registers the package alias)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def synth_weights_np():
    from conette_amd import synth
    return synth.synth_state_dict()


@pytest.fixture(scope="session")
def synth_weights(synth_weights_np):
    from oracle import cpu_ref
    return cpu_ref.to_torch(synth_weights_np)


@pytest.fixture(scope="session")
def synth_cfg():
    from conette_amd import synth
    return synth.synth_config_dict()
