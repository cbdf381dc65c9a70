"""Which native code a process of the PRODUCT loads: a fresh interpreter that imports only `tetris_piclim` and steps
boards on the GPU must have <repo>/lib/libtetris_piclim.so mapped and nothing of the oracle (the checker lives in
tests/, smoke() and bench.py's cpu_baseline leg only)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

SCRIPT = r"""
import sys
sys.path.insert(0, {root!r})
import torch
import tetris_piclim as T
env = T.BatchedTetris(5, 20, 4096, device="cuda:0", seed=3, auto_reset=True)
rows, pieces = env.synthetic_configs(512)
env.load_configs(rows, pieces)
env.reset()
for t in range(40):
    env.step(env.synthetic_actions(t), observe=False)
stats = env.stats()
assert stats["episodes"] > 0
maps = open("/proc/self/maps").read()
mods = sorted(m for m in sys.modules if "oracle" in m)
print("LIB", T.LIB_PATH)
print("MAPPED", int(T.LIB_PATH in maps))
print("ORACLE_SO", int("libtetris_oracle" in maps))
print("ORACLE_MODULES", mods)
"""


@pytest.mark.gpu
def test_product_process_maps_the_hip_library_and_never_the_oracle():
    res = subprocess.run([sys.executable, "-c", SCRIPT.format(root=ROOT)], capture_output=True, text=True, timeout=600,
                         cwd=ROOT)
    assert res.returncode == 0, res.stdout + res.stderr
    out = dict(line.split(" ", 1) for line in res.stdout.strip().splitlines() if " " in line)
    lib = out["LIB"]
    assert lib == os.path.join(ROOT, "lib", "libtetris_piclim.so")
    assert out["MAPPED"] == "1" and out["ORACLE_SO"] == "0" and out["ORACLE_MODULES"] == "[]"


def test_library_path_is_short_and_in_tree():
    import tetris_piclim as T
    assert T.LIB_PATH == os.path.join(ROOT, "lib", "libtetris_piclim.so")
    assert os.path.realpath(T.LIB_PATH).startswith(os.path.realpath(ROOT))
