import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A kernel that never ends must fail its test, not hold the GPU box until the runner's own limit: every GPU test gets a
    15-minute ceiling (the slowest, the full-size configs, take about a minute) when pytest-timeout is importable."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for it in items:
        if it.get_closest_marker("gpu") and not it.get_closest_marker("timeout"):
            it.add_marker(pytest.mark.timeout(900))


def pkg(name):
    """The package directory is `msu-latentafis_amd` (hyphen): import its modules through importlib."""
    return importlib.import_module("msu-latentafis_amd." + name)


@pytest.fixture(scope="session")
def templates_mod():
    return pkg("host.templates")


@pytest.fixture(scope="session")
def synth_mod():
    return pkg("host.synth")


@pytest.fixture(scope="session")
def codebook_bytes():
    with open(os.path.join(ROOT, "tests", "golden", "codebook_EmbeddingSize_96_stride_16_subdim_6.dat"), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle
    return Oracle()
