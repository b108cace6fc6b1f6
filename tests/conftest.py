import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.dirname(__file__)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """Make sure libt2v_hip.so exists (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as ge
    ge.build()
    from sd_webui_text2video_amd import _lib
    return _lib.load()
