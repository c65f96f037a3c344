"""AddressSanitizer + UndefinedBehaviorSanitizer over the host-side code that ships in the library (layer plan, h_eff,
spectral gains, SVF / biquad design, time-parallel tables) and over the oracle's C restatement, on the CPU build
(GPU sanitizers are not available on the pool).  The driver also cross-checks plan and coefficient fields between the
two implementations for every argument combination it sweeps."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_design_and_oracle_under_asan_ubsan(tmp_path):
    exe = tmp_path / "host_sanitize"
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
    csrc = os.path.join(ROOT, "convopeq_amd", "csrc")
    obj = tmp_path / "oracle.o"
    subprocess.run(["gcc", "-O1", "-g", "-std=gnu11", "-mavx2", "-mfma", "-ffp-contract=off", *san, "-c",
                    os.path.join(ROOT, "oracle", "cpq_oracle.c"), "-o", str(obj)], check=True)
    subprocess.run(["g++", "-O1", "-g", "-std=c++20", "-ffp-contract=off", *san,
                    "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-I" + os.path.join(ROOT, "oracle"),
                    os.path.join(HERE, "sanitize", "host_sanitize.cpp"), os.path.join(csrc, "host_design.cpp"), str(obj),
                    "-lm", "-o", str(exe)], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "0 failed checks" in r.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_ir_ingest_under_asan_ubsan(tmp_path):
    """the WAV reader against truncated / corrupted / random files and the conditioning + analysis code on degenerate
    buffers, with AddressSanitizer, LeakSanitizer and UBSan."""
    exe = tmp_path / "ir_ingest_sanitize"
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
    csrc = os.path.join(ROOT, "convopeq_amd", "csrc")
    subprocess.run(["g++", "-O1", "-g", "-std=c++20", "-ffp-contract=off", *san, "-I" + os.path.join(ROOT, "include"),
                    os.path.join(HERE, "sanitize", "ir_ingest_sanitize.cpp"), os.path.join(csrc, "ir_ingest.cpp"),
                    "-lm", "-o", str(exe)], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert " 0 failed checks" in r.stdout
