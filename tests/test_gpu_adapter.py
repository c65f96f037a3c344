"""C++ adapter (include/convopeq_mi355x.hpp): compile check on CPU, run on the GPU box."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CMD = ["g++", "-std=c++20", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(HERE, "adapter_smoke.cpp"),
       "-L", os.path.join(ROOT, "convopeq_amd"), "-lconvopeq_mi355x", "-Wl,-rpath," + os.path.join(ROOT, "convopeq_amd"),
       "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"]


def test_adapter_compiles_and_links(tmp_path):
    exe = tmp_path / "adapter_smoke"
    subprocess.check_call(CMD + ["-o", str(exe)])
    assert exe.exists()


@pytest.mark.gpu
def test_adapter_add_get_matches_direct_form(tmp_path):
    exe = tmp_path / "adapter_smoke"
    subprocess.check_call(CMD + ["-o", str(exe)])
    r = subprocess.run([str(exe), os.path.join(HERE, "golden", "impulse_room_correction_hpf_lpf.wav")], capture_output=True, text=True)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
