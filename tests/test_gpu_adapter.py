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


@pytest.mark.gpu
def test_library_loaded_before_torch_shares_one_hip_runtime():
    """Import order must not matter: PyTorch-ROCm bundles its own libamdhip64; loaded second beside the system copy this
    library links, it used to leave torch.cuda with "No HIP GPUs are available" (convopeq_amd/_capi.py maps torch's copy
    first when a torch installation exists)."""
    import sys
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import convopeq_amd as amd; "
            "e = amd.BatchedEngine(1, max_ir_len=512, max_blocks_per_call=1); import torch; "
            "assert torch.cuda.is_available(); t = torch.ones(8, dtype=torch.float64).cuda(); torch.cuda.synchronize(); "
            "assert float(t.sum()) == 8.0; e.close(); print('one runtime')")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "one runtime" in r.stdout, r.stdout + r.stderr
