"""Native consumers of the boundary: a plain-C translation unit against include/*.h (the view a cgo / JNI
binding has) and a C++ host against csrc/caffe_api.hpp written as INTEGRATION.md section B shows."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mms_answer_selection_amd")


def test_headers_compile_as_c99(tmp_path):
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    obj = tmp_path / "abi_is_c.o"
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           "-c", os.path.join(ROOT, "tests", "native", "abi_is_c.c"), "-o", str(obj)])
    assert obj.stat().st_size > 0


@pytest.mark.gpu
def test_cpp_host_against_the_layer_mirror(tmp_path, hiplib):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "layer_host"
    subprocess.check_call([hipcc, "-O1", "-std=c++17", "-x", "hip", "--offload-arch=gfx950",
                           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(PKG, "csrc"),
                           os.path.join(ROOT, "tests", "native", "layer_host.cpp"), "-o", str(exe),
                           "-L", PKG, "-lmms_caffe", "-lmms_hip", "-Wl,-rpath," + PKG])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "layer_host ok" in out.stdout


def test_path_a_replacement_units_compile(tmp_path):
    """INTEGRATION.md path A: the three translation units a maintainer drops into the reference's tree in place of
    sim_cross_layer.cu / sim_matrix_layer.cu / pair_rank_loss_layer.cu compile (hipcc, gfx950, -Wall -Werror) against
    the reference's class declarations (restated on the Layer / Blob mirror), and the document shows exactly them."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    pa = os.path.join(ROOT, "tests", "native", "path_a")
    for unit in ("sim_cross_layer_gpu.cpp", "sim_matrix_layer_gpu.cpp", "pair_rank_loss_layer_gpu.cpp"):
        src = os.path.join(pa, unit)
        obj = tmp_path / (unit + ".o")
        subprocess.check_call([hipcc, "-O1", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", "-Wall", "-Werror",
                               "-I", os.path.join(ROOT, "include"), "-I", os.path.join(PKG, "csrc"), "-I", pa,
                               "-c", src, "-o", str(obj)])
        assert obj.stat().st_size > 0
        assert open(src).read() in doc, unit + " is not shown verbatim in INTEGRATION.md"
