"""CPU tier: the product's plain-C host code under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md section 5:
the reference's host code is sanitizer-clean C; so must this be). `make -C qnnpack_amd/csrc asan` compiles the host
translation units with -fsanitize=address,undefined against tests/hip_stub.c (a malloc/memcpy stand-in for the HIP
seam; launches validate their argument block and touch every buffer end to end) and tests/host_asan_test.c, which
walks create -> setup -> run -> re-setup -> delete over every operator type and packing / table path. Leaks count."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_is_clean_under_asan_and_ubsan():
    csrc = os.path.join(ROOT, "qnnpack_amd", "csrc")
    build = subprocess.run(["make", "-C", csrc, "asan"], capture_output=True, text=True)
    assert build.returncode == 0, build.stdout + build.stderr
    exe = os.path.join(csrc, "build", "asan", "host_asan_test")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    env.pop("LD_PRELOAD", None)
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0 and "host-sanitizers-ok" in run.stdout, run.stdout[-2000:] + run.stderr[-6000:]
