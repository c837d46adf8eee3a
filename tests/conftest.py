"""pytest configuration: registers the `gpu` marker and loads the product package.

The product package directory is named `h.264_amd` (not an importable identifier), so it is loaded
by path once and published as `h264_amd`.
"""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference and oracle/_ref (container only)")


def load_pkg():
    if "h264_amd" in sys.modules:
        return sys.modules["h264_amd"]
    path = os.path.join(ROOT, "h.264_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("h264_amd", path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["h264_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()
