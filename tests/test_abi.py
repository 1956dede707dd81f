"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/mot.h declares, refuses to create a
context without a GPU (no fallback), and its parameter presets equal the oracle's independently restated table."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "mot.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mot_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(mot, hip_lib):
    names = declared_functions()
    assert len(names) >= 18
    missing = [n for n in names if not hasattr(hip_lib, n)]
    assert not missing, missing
    assert set(names) == set(mot.EXPORTS), set(names) ^ set(mot.EXPORTS)
    assert hip_lib.mot_abi_version() == mot.ABI_VERSION == 6


def test_presets_match_the_oracle_table(mot, hip_lib, oracle):
    for preset in (0, 1):
        a, b = mot.params(preset), oracle.params(preset)
        for name, _ in mot.MotParams._fields_:
            assert getattr(a, name) == getattr(b, name), (preset, name)
    assert C.sizeof(mot.MotParams) == C.sizeof(oracle.MotParams)
    assert C.sizeof(mot.MotTrack) == 144


def test_no_cpu_fallback(mot, hip_lib):
    """without a GPU the product refuses to run instead of computing on the CPU"""
    try:
        import hiprt
        n = C.c_int(0)
        has_gpu = hiprt.hip().hipGetDeviceCount(C.byref(n)) == 0 and n.value > 0
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(mot.MotError) as e:
        mot.Context()
    assert e.value.code == mot.MOT_E_HIP


def test_missing_library_is_an_error(mot):
    with pytest.raises(ImportError):
        mot.load_library("/nonexistent/libmot_hip.so")
