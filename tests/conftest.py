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
    # the workload generators (synthetic clouds / rendered sequences) are bench + test infrastructure: tools/synth, not the product package
    where = os.path.join(ROOT, "tools", "synth") if name in ("synth", "synth_dev") else PKG_DIR
    spec = importlib.util.spec_from_file_location(full, os.path.join(where, name + ".py"))
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
def _oracle_module():
    import oracle_lib
    oracle_lib.build_oracle()
    return oracle_lib


ORACLE_LOG = []   # (test id, {(function, which oracle answered): calls})


@pytest.fixture
def oracle(request, _oracle_module):
    """CPU tests: the C restatement (tests/oracle_lib.py; pinned to the reference build by tests/test_oracle_vs_ref.py).
    `-m gpu` tests: the reference's OWN sources first (oracle_lib.RefFirst: oracle/_ref/*.so is on the GPU box), the restatement only
    where the reference cannot answer; which one answered is printed per test at the end of the run (MOT_ORACLE=restatement: off)."""
    if request.node.get_closest_marker("gpu") is None or os.environ.get("MOT_ORACLE") == "restatement" or _oracle_module.ref() is None:
        yield _oracle_module
        return
    o = _oracle_module.RefFirst(_oracle_module)
    yield o
    ORACLE_LOG.append((request.node.nodeid, dict(o.used)))


def pytest_terminal_summary(terminalreporter):
    if not ORACLE_LOG:
        return
    tr = terminalreporter
    tr.section("oracle used per GPU test (reference build = the reference's own sources, oracle/_ref)")
    for nodeid, used in ORACLE_LOG:
        by = {}
        for (fn, who), k in sorted(used.items()):
            by.setdefault(who.split(" (")[0], []).append(f"{fn} x{k}")
        tr.write_line(f"{nodeid}: " + ("; ".join(f"{who}: {', '.join(v)}" for who, v in sorted(by.items())) or "no oracle call"))
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "oracle_used_per_test.txt"), "w") as f:
            for nodeid, used in ORACLE_LOG:
                f.write(nodeid + "\n" + "".join(f"    {fn:<16}{who}  x{k}\n" for (fn, who), k in sorted(used.items())))
    except OSError:
        pass


@pytest.fixture(scope="session")
def hip_lib(mot):
    """the real HIP extension; fails loudly (no fallback) when it has not been built"""
    build = load_sub("build")
    if not os.path.exists(build.LIB):
        build.build()
    return mot.load_library()
