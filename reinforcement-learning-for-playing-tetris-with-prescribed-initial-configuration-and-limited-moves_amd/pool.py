"""On-disk format of a pool of prescribed configurations (the reference keeps them only in a process queue,
game/tetris.py:195,473-488 of the upstream repo).

One .npz file: L, M, rows uint16 [n, 20] (bit x = column x), pieces uint8 [n, M+1] (ids I0 L1 J2 T3 S4 Z5 O6) and,
optionally, the carved solution uint8 [n, M, 2] (rotations, location) with its length int32 [n] -- the triple
(board, pieces, solution) of the reference's debug mode (game/tetris.py:155-156, 259-260)."""
from __future__ import annotations

import numpy as np


def save_pool(path: str, L: int, M: int, rows, pieces, solution=None, solution_len=None) -> None:
    rows = np.ascontiguousarray(rows, dtype=np.uint16)
    pieces = np.ascontiguousarray(pieces, dtype=np.uint8)
    if rows.ndim != 2 or rows.shape[1] != 20 or pieces.shape != (rows.shape[0], M + 1):
        raise ValueError(f"rows must be [n, 20] and pieces [n, {M + 1}]")
    if pieces.max(initial=0) > 6 or rows.max(initial=0) > 0x3FF:
        raise ValueError("piece ids must be 0..6 and rows must use 10 columns")
    extra = {}
    if solution is not None:
        extra = dict(solution=np.ascontiguousarray(solution, dtype=np.uint8),
                     solution_len=np.ascontiguousarray(solution_len, dtype=np.int32))
    np.savez_compressed(path, L=np.int32(L), M=np.int32(M), rows=rows, pieces=pieces, **extra)


def load_pool(path: str) -> dict:
    f = np.load(path, allow_pickle=False)
    out = dict(L=int(f["L"]), M=int(f["M"]), rows=f["rows"], pieces=f["pieces"])
    if "solution" in f.files:
        out["solution"], out["solution_len"] = f["solution"], f["solution_len"]
    if out["pieces"].shape != (out["rows"].shape[0], out["M"] + 1):
        raise ValueError("corrupt pool file: pieces do not match rows / M")
    return out
