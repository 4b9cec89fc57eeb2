"""AddressSanitizer + UndefinedBehaviorSanitizer over the HIP-free host code: the NUTS state machine
of the native sampler (abd_nuts.hpp) and the plain-C oracle.  CPU only -- GPU sanitizers are not available on the pool."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    exe = tmp_path / "sanitize_main"
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
    obj = tmp_path / "abd_oracle.o"
    subprocess.check_call(["gcc", "-std=c11", "-fopenmp", "-Wall", "-Wextra", *san, "-c", os.path.join(ROOT, "oracle", "abd_oracle.c"),
                           "-o", str(obj)])
    subprocess.check_call(["g++", "-std=c++17", "-fopenmp", "-Wall", "-Wextra", *san, "-I", os.path.join(ROOT, "abdpymc_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "sanitize_main.cpp"),
                           os.path.join(ROOT, "tests", "native", "nuts_harness.cpp"), str(obj), "-lm", "-o", str(exe)])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="3")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "sanitize ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
