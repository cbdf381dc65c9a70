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
