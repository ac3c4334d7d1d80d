"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads without a GPU and exports every
function include/sdm.h declares; struct layouts seen by ctypes match the header."""
import ctypes as C
import os
import re

import pytest

from semantic_dsp_map_amd import binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header="sdm.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sdm_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    if not os.path.exists(binding.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = C.CDLL(binding.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "declared in include/sdm.h but not exported: %s" % missing


def test_object_layer_header_is_exported_and_bound():
    """include/sdm_objects.h (host-side object layer): every function exported, every one reachable from Python."""
    from semantic_dsp_map_amd import objects
    lib = C.CDLL(binding.LIB_PATH)
    names = declared_functions("sdm_objects.h")
    assert len(names) == 10
    assert not [n for n in names if not hasattr(lib, n)]
    L = objects._lib()
    for n in names:
        assert getattr(L, n).argtypes is not None, n
    assert C.sizeof(objects.ObjectsConfig) == 96 and C.sizeof(objects.Observation) == 32
    assert C.sizeof(objects.ObjectMove) == 68 and C.sizeof(objects.ObjectInfo) == 200


def test_binding_covers_the_header():
    L = binding.load_library()
    for n in declared_functions():
        assert getattr(L, n) is not None


def test_struct_layouts():
    assert C.sizeof(binding.Config) == 80
    assert C.sizeof(binding.Params) == 52
    assert C.sizeof(binding.RingState) == 60
    assert binding.LABELED_POINT.itemsize == 20 and binding.VOXEL_RESULT.itemsize == 8
    assert binding.OBJECT_MOVE.itemsize == 68 and binding.POINT.itemsize == 16


def test_no_cpu_path():
    """Without a GPU the library must refuse to create a map instead of falling back."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    from tests import kat_cases as kc
    with pytest.raises(binding.SdmError) as e:
        binding.SdmMap(kc.K0, kc.PARAMS)
    assert "SDM_ERR_NO_DEVICE" in str(e.value) or "SDM_ERR_HIP" in str(e.value)


def test_product_package_does_not_touch_the_oracle():
    """oracle/ is test infrastructure: nothing in the product package may import, include, link or run it."""
    pkg = os.path.join(ROOT, "semantic_dsp_map_amd")
    pat = re.compile(r"(^\s*(from|import)\s+oracle\b)|(#include\s*[\"<][^\">]*oracle)|(oracle[/\\])|(libsdm_oracle)|(cpu_ref)", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), "%s references the oracle" % f


def test_host_placement_calls_are_safe_without_a_gpu():
    """sdm_bind_host_thread / sdm_host_numa_node_early (sdm.h "host placement"): no GPU and no KFD topology here - they must
    say so and leave the caller's CPU affinity alone; SDM_NUMA_BIND=0 switches the binding off altogether."""
    import ctypes
    import os
    import subprocess
    import sys
    code = ("import ctypes, os; L = ctypes.CDLL(%r); "
            "L.sdm_host_numa_node_early.restype = ctypes.c_int32; L.sdm_bind_host_thread.restype = ctypes.c_int32; "
            "a = os.sched_getaffinity(0); e = L.sdm_host_numa_node_early(0); b = L.sdm_bind_host_thread(0); "
            "print(e, b, a == os.sched_getaffinity(0))")
    from semantic_dsp_map_amd import binding
    for env_extra in ({}, {"SDM_NUMA_BIND": "0"}):
        out = subprocess.run([sys.executable, "-c", code % binding.LIB_PATH], capture_output=True, text=True, timeout=120,
                             env=dict(os.environ, **env_extra)).stdout.split()
        has_kfd = os.path.isdir("/sys/class/kfd/kfd/topology/nodes")
        if not has_kfd:
            assert out[0] == "-2", out          # cannot tell without the runtime
            assert out[1] == "-1", out          # ... and the runtime finds no device: nothing done
            assert out[2] == "True", out        # affinity untouched
        if env_extra:
            assert out[1] == "-1" and out[2] == "True", out
