import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libsdvref.so (the real reference, build container only)")


@pytest.fixture(scope="session")
def oracle_lib():
    import libs
    libs.build_oracle()
    return libs.load_oracle()


@pytest.fixture(scope="session")
def emu_lib():
    import ctypes as C
    import engine_api
    from sdvpcmdecoder_amd import build as b
    return engine_api.bind(C.CDLL(b.build_emu()))
