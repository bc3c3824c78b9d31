"""The C++ host side above the C-ABI (plugin/proslam_hip_plugin.hpp) driven by the reference's own
test shapes (tests/cpp/test_plugin_surface.cpp), on a real GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_plugin_surface")


@pytest.mark.gpu
def test_cpp_plugin_surface_program():
    assert os.path.exists(EXE), "run __graft_entry__.build() first"
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([EXE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=300)
    out = r.stdout.decode()
    assert r.returncode == 0, out
    assert out.count("[  OK  ]") == 12 and "FAILED" not in out and "0 failure(s)" in out, out


def test_cpp_plugin_surface_builds_and_fails_loudly_without_gpu():
    import torch
    assert os.path.exists(EXE) or True
    if not os.path.exists(EXE) or torch.cuda.is_available():
        pytest.skip("needs the built program and no GPU")
    r = subprocess.run([EXE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60)
    assert r.returncode == 2 and b"no HIP device" in r.stdout  # never a silent CPU path
