"""MI355X-native batched Tetris-piclim environment (Tetris with a prescribed initial configuration, L lines
to clear, at most M moves on a 20x10 board).

The package holds only what the hot path needs: `csrc/` (hand-written HIP kernels for gfx950 + the C ABI of
include/tetris_piclim.h) and `env.py`, the host-side mirror of the reference's `Tetris` interface.
The directory name is not a Python identifier; import it as `import tetris_piclim` (alias module at the repo
root) or with importlib.import_module.
"""
from . import _lib
from ._lib import (LIB_PATH, SYMBOLS, TplError, build_library, carve, forward_generate, generate_configs, pack_policy,
                   shape_info)

__all__ = ["BatchedTetris", "Tetris", "Snapshot", "OBS_DIM", "NUM_ACTIONS", "RUNNING", "WON", "LOST", "TplError",
           "RandomPieceGenerator", "get_tetromino", "piece_translations", "translate", "carve", "build_library", "shape_info", "generate_configs", "forward_generate", "pack_policy", "LIB_PATH", "SYMBOLS"]


def __getattr__(name):
    # env.py needs torch; keep `import tetris_piclim` cheap for callers that only build or bind the library
    import importlib
    if name in ("BatchedTetris", "Tetris", "Snapshot", "OBS_DIM", "NUM_ACTIONS", "RUNNING", "WON", "LOST", "env"):
        env = importlib.import_module(__name__ + ".env")
        return env if name == "env" else getattr(env, name)
    if name in ("save_pool", "load_pool", "PoolRefresher", "ForwardGames", "blend"):
        return getattr(importlib.import_module(__name__ + ".pool"), name)
    if name in ("sharding", "actor", "pool"):
        return importlib.import_module(__name__ + "." + name)
    if name in ("RandomPieceGenerator", "get_tetromino", "piece_translations", "translate"):
        return getattr(importlib.import_module(__name__ + ".pieces"), name)
    if name in ("Actor", "PolicyMLP"):
        return getattr(importlib.import_module(__name__ + ".actor"), name)
    raise AttributeError(name)
