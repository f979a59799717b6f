"""The drop-in boundary is a C ABI: these tests consume include/codetr_hip.h from plain C, without Python or torch in
the calling program -- what a cgo / JNI / TensorRT-plugin style caller would link against.
CPU: the header is valid C11 and C++17 and a C program linked against libcodetr_hip.so runs its host-only entry
points.  GPU: a C program in the shape of the reference plugin's enqueue (deformable_attention_plugin.cpp:285-355)
runs an fp32 MSDA forward on raw hipMalloc'ed pointers and checks it against the C oracle."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "co-detr-tensorrt_amd", "codetr")
SRC = os.path.join(ROOT, "tests", "cabi_c")
ROCM = "/opt/rocm"


def _run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, **kw)
    assert r.returncode == 0, f"{' '.join(cmd)}\n{r.stdout}\n{r.stderr}"
    return r.stdout


def test_header_is_valid_c_and_cxx(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    tu = tmp_path / "tu.c"
    tu.write_text('#include "codetr_hip.h"\nint main(void) { return CODETR_HIP_ABI_VERSION > 0 ? 0 : 1; }\n')
    _run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", f"-I{INC}", str(tu)])
    _run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-x", "c++", "-fsyntax-only", f"-I{INC}", str(tu)])


def test_c_program_links_and_runs_host_only_entry_points(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    exe = tmp_path / "host_only"
    _run(["gcc", "-std=c11", "-Wall", "-O1", f"-I{INC}", os.path.join(SRC, "host_only.c"), "-o", str(exe),
          f"-L{LIBDIR}", "-lcodetr_hip", f"-Wl,-rpath,{LIBDIR}", f"-Wl,-rpath,{ROCM}/lib"])
    out = _run([str(exe)])
    assert out.startswith("abi ")


@pytest.mark.gpu
def test_plugin_style_c_caller_matches_oracle(tmp_path):
    from oracle import msda_oracle

    oracle_so = msda_oracle.build()
    exe = tmp_path / "plugin_style_caller"
    _run(["gcc", "-std=c11", "-O1", f"-I{INC}", f"-I{ROCM}/include", os.path.join(SRC, "plugin_style_caller.c"), "-o",
          str(exe), f"-L{LIBDIR}", "-lcodetr_hip", oracle_so, f"-L{ROCM}/lib", "-lamdhip64", "-lm",
          f"-Wl,-rpath,{LIBDIR}", f"-Wl,-rpath,{ROCM}/lib", f"-Wl,-rpath,{os.path.dirname(oracle_so)}"])
    out = _run([str(exe)], timeout=120)
    assert "max |gpu - oracle|" in out
