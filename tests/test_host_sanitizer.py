"""csrc/host_plumbing.h (the host side of SURVEY.md section 8f rows N3 / N4) under AddressSanitizer and
UndefinedBehaviorSanitizer on the CPU: GPU sanitizers are not available on the pool, the host code
needs no device.  The driver (tests/native/host_plumbing_sanitized.cpp) walks every function over
sizes around its internal boundaries; any report aborts the run.  No GPU."""

import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_plumbing_under_asan_and_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "host_plumbing_sanitized")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", "-Wall", "-Wextra", "-Werror",
           "-I", os.path.join(ROOT, "sgfhe.jl_amd", "csrc"),
           os.path.join(ROOT, "tests", "native", "host_plumbing_sanitized.cpp"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and "sanitize" in b.stderr and "cannot find" in b.stderr:
        pytest.skip("the sanitizer runtimes are not installed: " + b.stderr[-300:])
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    digest = r.stdout.strip()
    assert len(digest) == 16 and int(digest, 16) != 0
    # the same binary without the sanitizers computes the same digest (no dependence on poisoned or
    # uninitialised memory)
    exe2 = str(tmp_path / "host_plumbing_plain")
    subprocess.run([gxx, "-std=c++17", "-O2", "-I", os.path.join(ROOT, "sgfhe.jl_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "host_plumbing_sanitized.cpp"), "-o", exe2],
                   check=True, timeout=300)
    r2 = subprocess.run([exe2], capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0 and r2.stdout.strip() == digest


def test_oracle_c_under_asan_and_ubsan(tmp_path):
    """The checker itself (oracle/sgfhe_oracle.c) under the sanitizers: key generation, bootstraps in
    both of its forms, the truth table and one packing at Params(64)."""
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    src = [os.path.join(ROOT, "tests", "native", "oracle_sanitized.c"), os.path.join(ROOT, "oracle", "sgfhe_oracle.c")]
    exe = str(tmp_path / "oracle_sanitized")
    b = subprocess.run([gcc, "-O1", "-g", "-fopenmp", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        "-fno-omit-frame-pointer", "-I", os.path.join(ROOT, "oracle")] + src + ["-o", exe, "-lm"],
                       capture_output=True, text=True, timeout=600)
    if b.returncode != 0 and "sanitize" in b.stderr and "cannot find" in b.stderr:
        pytest.skip("the sanitizer runtimes are not installed")
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1",
               OMP_NUM_THREADS="1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    exe2 = str(tmp_path / "oracle_plain")
    subprocess.run([gcc, "-O2", "-fopenmp", "-I", os.path.join(ROOT, "oracle")] + src + ["-o", exe2, "-lm"],
                   check=True, timeout=600)
    r2 = subprocess.run([exe2], capture_output=True, text=True, timeout=900, env=env)
    assert r2.returncode == 0 and r2.stdout.strip() == r.stdout.strip()


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_coalescer_under_thread_and_address_sanitizers(tmp_path, san):
    """csrc/coalescer.h -- the queueing half of the gathering of independent callers on one key (round 5,
    sgfhe_set_coalesce) -- under ThreadSanitizer and under ASan / UBSan on the CPU, with a stand-in for the combined
    call (tests/native/coalescer_tsan.cpp): twelve threads of mixed requests; every request gets its own rows, rounds
    never overlap or mix modes, errors reach exactly their round, the statistics add up, a lone caller is not
    delayed, and the knobs change under the callers' feet as sgfhe_set_coalesce changes them.  The requests live on their callers' stacks while another thread serves them: exactly what these tools
    are for."""
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "coalescer_san")
    b = subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-pthread", "-fsanitize=" + san, "-fno-sanitize-recover=all",
                        "-fno-omit-frame-pointer", "-Wall", "-Wextra", "-Werror",
                        "-I", os.path.join(ROOT, "sgfhe.jl_amd", "csrc"),
                        os.path.join(ROOT, "tests", "native", "coalescer_tsan.cpp"), "-o", exe],
                       capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and "sanitize" in b.stderr and "cannot find" in b.stderr:
        pytest.skip("the sanitizer runtimes are not installed: " + b.stderr[-300:])
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1",
               ASAN_OPTIONS="detect_leaks=1:abort_on_error=1:detect_stack_use_after_return=1", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    tag, n_calls, n_reqs, n_alone = r.stdout.split()
    # something was gathered, and some requests were above the req_max another thread kept changing (wants())
    assert tag == "ok" and int(n_reqs) + int(n_alone) == 12 * 60 + 200 and int(n_alone) > 0 and int(n_calls) < int(n_reqs)
