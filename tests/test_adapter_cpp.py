"""The C++ host side of the drop-in boundary: include/semantic_dsp_map.h (the reference's class API) compiled
against stand-in Eigen/OpenCV/PCL headers and linked with libsdm_hip.so."""
import os
import subprocess

import pytest

from semantic_dsp_map_amd import binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "adapter_smoke")


def build_exe():
    csrc = os.path.dirname(binding.LIB_PATH)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "tests", "mock_includes"), "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "adapter_smoke.cpp"), "-o", EXE, "-L", csrc, "-lsdm_hip",
           "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)


def test_adapter_header_compiles_and_links():
    build_exe()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "adapter constructed" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("defs,setting,boost", [
    ([], 3, 1),                                   # nothing defined: what the reference ships (settings.h:22-29)
    (["-DSETTING=0"], 0, 0), (["-DSETTING=1"], 1, 0), (["-DSETTING=2"], 2, 0), (["-DSETTING=3"], 3, 1),
    (["-DSETTING=3", "-DBOOST_MODE=0"], 3, 0), (["-DSETTING=2", "-DBOOST_MODE=1"], 2, 1),
])
def test_adapter_honours_the_setting_macros(defs, setting, boost, tmp_path):
    """`#define SETTING n` / BOOST_MODE select the adapter's preset like they select the reference's constants."""
    csrc = os.path.dirname(binding.LIB_PATH)
    exe = str(tmp_path / "adapter_setting")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall"] + defs +
                          ["-I", os.path.join(ROOT, "tests", "mock_includes"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "adapter_setting.cpp"), "-o", exe, "-L", csrc, "-lsdm_hip",
                           "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "preset ok" in out.stdout, out.stdout + out.stderr
    assert "setting %d boost %d:" % (setting, boost) in out.stdout, out.stdout


def build_setting_exe(defs, exe):
    csrc = os.path.dirname(binding.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall"] + defs +
                          ["-I", os.path.join(ROOT, "tests", "mock_includes"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "adapter_setting.cpp"), "-o", exe, "-L", csrc, "-lsdm_hip",
                           "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib"])


@pytest.mark.gpu
@pytest.mark.parametrize("defs", [[], ["-DSETTING=0"], ["-DSETTING=1"], ["-DSETTING=2"], ["-DSETTING=3", "-DBOOST_MODE=0"]])
def test_adapter_runs_the_shipped_grids(defs, tmp_path):
    """The default-constructed class at every grid the reference ships (nothing defined = SETTING 3 + BOOST_MODE, its ZED2
    configuration with 1280 x 720 inputs reduced on the device) takes a wall scene through update()."""
    exe = str(tmp_path / "adapter_setting")
    build_setting_exe(defs, exe)
    out = subprocess.run([exe, "run"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "occupied voxels after 4 frames" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_adapter_runs_a_wall_scene():
    build_exe()
    out = subprocess.run([EXE, "run"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
