"""pytest configuration: registers the ``gpu`` marker and common fixtures."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR


def pytest_collection_modifyitems(config, items):
    """GPU tests are *selected* with -m gpu; if someone runs them without a GPU, fail loudly
    rather than silently skipping (the product has no CPU fallback)."""
    return
