"""The quick-start block of README.md is executed as written."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.gpu
def test_readme_quick_start_runs():
    text = open(os.path.join(ROOT, "README.md")).read()
    block = re.search(r"```python\n(.*?)```", text, re.S).group(1)
    scope = {}
    exec(compile(block, "README.md", "exec"), scope)
    env = scope["env"]
    assert tuple(scope["obs"].shape) == (env.num_envs, 217) and scope["reward"].shape[0] == env.num_envs
    assert tuple(scope["board"].shape) == (env.num_envs, 20, 10)
    env.terminate()


@pytest.mark.gpu
def test_integration_stub_runs_as_documented():
    """The ctypes stub of INTEGRATION.md section 2, executed as written (only the library path is resolved), drives the
    library with torch-owned buffers and agrees with BatchedTetris."""
    import numpy as np
    import torch
    import tetris_piclim as T
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(import ctypes as C.*?)```", text, re.S).group(1)
    block = block.replace('C.CDLL("libtetris_piclim.so")', f'C.CDLL("{T._lib.build_library()}")')
    scope = {}
    exec(compile(block, "INTEGRATION.md", "exec"), scope)
    L, M, n = 5, 20, 3000
    ref = T.BatchedTetris(L, M, n, seed=3, assign="sequential")
    rows, pieces = ref.synthetic_configs(n)
    ref.load_configs(rows, pieces)
    ref.reset()
    stub = scope["TetrisBatch"](L, M, n, device=0, seed=3)
    scope["_check"](scope["_lib"].tpl_set_options(stub._h, 0, 1, 1.0, 0.0, 0.0))     # sequential assignment, as `ref`
    stub.load(rows.data_ptr(), pieces.data_ptr(), n)
    stub.reset()
    dev = ref.device
    reward = torch.empty(n, dtype=torch.float32, device=dev)
    done = torch.empty(n, dtype=torch.uint8, device=dev)
    for t in range(12):
        a = ref.synthetic_actions(t).to(torch.int64)
        rot, loc = a // 10, a % 10
        r_ref, d_ref, _ = ref.move(rot, loc)
        stub.move(rot.data_ptr(), loc.data_ptr(), reward.data_ptr(), done.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(reward, r_ref) and torch.equal(done.view(torch.bool), d_ref)
    obs = torch.empty((n, 217), dtype=torch.float32, device=dev)
    for t in range(12, 16):                                          # step(action) -> (obs, reward, done) in one launch
        a = ref.synthetic_actions(t)
        o_ref, r_ref, d_ref, _ = ref.step(a)
        stub.step(a.data_ptr(), reward.data_ptr(), done.data_ptr(), obs.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(reward, r_ref) and torch.equal(done.view(torch.bool), d_ref) and torch.equal(obs, o_ref)
    out = {k: torch.empty(n, dtype=torch.uint8, device=dev) for k in ("cur", "nxt", "lines", "moves", "state")}
    rows_out = torch.empty((n, 20), dtype=torch.int16, device=dev)
    stub.get_state(rows_out.data_ptr(), *(out[k].data_ptr() for k in ("cur", "nxt", "lines", "moves", "state")))
    torch.cuda.synchronize()
    want = ref.packed_state()
    assert torch.equal(rows_out, want["rows"]) and all(torch.equal(out[k], want[k]) for k in out)
    stub.terminate()
    ref.terminate()
