import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def engine():
    """One engine for the whole GPU session (the product path: libsoundkit_amd.so on cuda:0)."""
    import soundkit_amd
    eng = soundkit_amd.Engine(0, 8192)
    # route the audio_bytes / audio_pipeline / decoder mirrors through the same engine
    import soundkit_amd.engine as E
    E._default = eng
    yield eng
    E._default = None
    eng.close()
