"""The boundary is a C ABI, not a Python extension: the header compiles as C99 and as C++, and a plain C program
(tests/c_abi/standalone.c, gcc, no HIP headers) drives the library on the GPU and reproduces the oracle."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _compile(cmd):
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, " ".join(cmd) + "\n" + res.stdout + res.stderr


@pytest.mark.parametrize("compiler,std", [("gcc", "-std=c99"), ("g++", "-std=c++11")])
def test_header_is_plain_c_and_cxx(tmp_path, compiler, std):
    src = tmp_path / ("use_header" + (".c" if compiler == "gcc" else ".cpp"))
    src.write_text('#include "tetris_piclim.h"\nint probe(void) { return (int)sizeof(tpl_status) + TPL_OBS_DIM; }\n')
    _compile([compiler, std, "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-c", str(src),
              "-o", str(tmp_path / "use_header.o")])


def test_standalone_program_links_against_the_library(tmp_path):
    """Link check on the CPU: every symbol the C program uses resolves against the built .so."""
    import tetris_piclim as T
    lib = T._lib.build_library()
    _compile(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
              os.path.join(ROOT, "tests", "c_abi", "standalone.c"), "-o", str(tmp_path / "standalone"),
              lib, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"])


@pytest.mark.gpu
def test_standalone_c_program_reproduces_the_oracle(tmp_path):
    import tetris_piclim as T
    from oracle import oracle
    lib = T._lib.build_library()
    exe = str(tmp_path / "standalone")
    _compile(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
              os.path.join(ROOT, "tests", "c_abi", "standalone.c"), "-o", exe,
              lib, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"])
    n, L, M, pool, steps, seed = 4096, 5, 20, 512, 60, 7
    res = subprocess.run([exe, str(n), str(steps)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    out = dict(line.split(" ", 1) for line in res.stdout.strip().splitlines())
    assert "gfx950" in out["version"]
    # the same run on the CPU oracle
    rows, pieces = oracle.synth_boards(seed, 0, pool, L), oracle.synth_pieces(seed, 0, pool, M)
    cpu = oracle.Env(n, L, M, 0, seed)
    cpu.set_pool(rows, pieces)
    cpu.set_options(auto_reset=True, assign_mode=0, per_line=1.0, win=5.0, lose=-1.0)
    cpu.reset()
    reward_sum, done_count = 0.0, 0
    for t in range(steps):
        r, d = cpu.step(oracle.synth_actions(seed, 0, n, t))
        reward_sum += float(r.astype(np.float64).sum())
        done_count += int(d.sum())
    s = cpu.stats()
    assert [int(x) for x in out["stats"].split()] == [s["episodes"], s["lines"], s["wins"], s["topouts"]]
    got = out["reward_sum"].split()
    assert float(got[0]) == round(reward_sum, 1) and int(got[2]) == done_count
    # the supply side through the C ABI alone: both generators on the device equal their host forms; the forward generator over
    # the reference's own seeds 0..99 at L = 5, M = 20 finds the 22 winnable games of tests/golden/forward_L5_M20.npz
    assert out["carve"] == "device==host ok"
    assert out["forward"] == "device==host ok winnable 22"


def test_host_generators_under_address_and_ub_sanitizers(tmp_path):
    """The host code of the library (carving generator, CPython random stream, forward generator + solver, tpl_carve)
    compiled host-only with -fsanitize=address,undefined and driven by tests/c_abi/sanitize_host.cpp: every
    configuration it generates is also carved back from a full stack with its own solution.  (Device code cannot
    run under a sanitizer on this pool; the host side can, on the CPU.)"""
    import tetris_piclim as T
    csrc = T._lib._CSRC
    flags = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I", os.path.join(ROOT, "include")]
    objs = []
    for unit in ("carve_generator.hip", "forward_generator.hip"):
        obj = str(tmp_path / (unit + ".o"))
        _compile(["hipcc", "--cuda-host-only"] + flags + ["-c", os.path.join(csrc, unit), "-o", obj])
        objs.append(obj)
    exe = str(tmp_path / "sanitize_host")
    _compile(["/opt/rocm/lib/llvm/bin/clang++"] + flags + [os.path.join(ROOT, "tests", "c_abi", "sanitize_host.cpp")] + objs +
             ["-o", exe, "-lpthread"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0 and "sanitizer run ok" in res.stdout, res.stdout + res.stderr
