import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package rrrmc.jl_amd (the directory name has a dot, so it is loaded by path)."""
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O
