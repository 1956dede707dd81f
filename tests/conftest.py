import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "3d-lidar-multi-object-tracking_amd")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_pkg():
    """import the package (its directory name is not a valid identifier) as `mot_amd`"""
    if "mot_amd" in sys.modules:
        return sys.modules["mot_amd"]
    spec = importlib.util.spec_from_file_location("mot_amd", os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    m = importlib.util.module_from_spec(spec)
    sys.modules["mot_amd"] = m
    spec.loader.exec_module(m)
    return m


def load_sub(name):
    full = "mot_amd." + name
    if full in sys.modules:
        return sys.modules[full]
    load_pkg()
    spec = importlib.util.spec_from_file_location(full, os.path.join(PKG_DIR, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules[full] = m
    spec.loader.exec_module(m)
    return m


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mot():
    return load_pkg()


@pytest.fixture(scope="session")
def synth():
    return load_sub("synth")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build_oracle()
    return oracle_lib


@pytest.fixture(scope="session")
def hip_lib(mot):
    """the real HIP extension; fails loudly (no fallback) when it has not been built"""
    build = load_sub("build")
    if not os.path.exists(build.LIB):
        build.build()
    return mot.load_library()
