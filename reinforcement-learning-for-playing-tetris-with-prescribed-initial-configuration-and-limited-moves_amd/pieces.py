"""The reference's piece supply, `RandomPieceGenerator` (game/tetris.py:64-108 of the upstream repo), with the same
interface: a bag of the seven piece ids that refills itself when it runs empty.

Host logic only (it feeds the configuration generators, not the step).  It draws from Python's `random` in the same
order as the reference does -- `randint` per single piece, `shuffle` per bag of a sequence -- so under the same
`random.seed` it produces the same pieces (`tests/golden/pieces.npz`).  The native generators (`generate_configs`,
`forward_generate`) carry their own copies of this logic and of CPython's random stream.
"""
from __future__ import annotations

import random
from typing import List, Tuple

# letter -> piece id of Tetris.move (game/tetris.py:8-16)
piece_translations = {"I": 0, "L": 1, "J": 2, "T": 3, "S": 4, "Z": 5, "O": 6}


def get_tetromino(piece: int, rotations: int):
    """The reference's get_tetromino (game/tetris.py:60-61): (mask, reverse_topography) of `piece` after `rotations`
    quarter turns -- a bool array [h, w] and a tuple of w ints -- decoded from the table the device kernels use."""
    import numpy as np
    from ._lib import shape_info
    h, w, masks, topo = shape_info(piece, rotations)
    mask = np.array([[(m >> x) & 1 for x in range(w)] for m in masks], dtype=bool)
    return mask, tuple(int(t) for t in topo)


def translate(games, rng=None) -> List[Tuple["np.ndarray", List[int]]]:
    """The reference's `translate(batch)` (game/tetris.py:19-20): the winnable games of the forward generator + solver as the
    `(board, pieces)` pairs its reset queue holds -- the board as a 20x10 bool array, the pieces as ONE random piece id
    (`random.randint(0, 6)`) followed by the game's sequence, M + 1 ids.  `games` is what `forward_generate()` returns (a dict
    of arrays over all seeds; only the winnable ones are translated, as `generate_batch` keeps only those).  The batched form
    of the same thing, on the device, is `pool.ForwardGames.translate`."""
    import numpy as np
    rng = rng if rng is not None else random
    out = []
    for k in np.flatnonzero(np.asarray(games["winnable"])):
        rows = np.asarray(games["rows"][k]).astype(np.uint16)
        board = ((rows[:, None] >> np.arange(10)) & 1).astype(bool)
        out.append((board, [rng.randint(0, 6)] + [int(x) for x in games["sequence"][k]]))
    return out


class RandomPieceGenerator:
    def __init__(self, rng=None) -> None:
        self.pieces: List[int] = []
        self._rng = rng if rng is not None else random

    def generate_pieces(self) -> None:
        self.pieces = list(range(7))

    def _refill(self) -> bool:
        """Opens a fresh bag if the current one is used up; tells whether it did."""
        if self.pieces:
            return False
        self.generate_pieces()
        return True

    def get_random_piece(self) -> Tuple[Tuple[int, int], bool]:
        """((piece id, its index in the bag), a fresh bag was opened).  The piece stays in the bag until
        delete_index(index) is called."""
        regenerated = self._refill()
        index = self._rng.randint(0, len(self.pieces) - 1)
        return (self.pieces[index], index), regenerated

    def delete_index(self, index: int) -> None:
        del self.pieces[index]

    def get_random_sequence(self, length: int) -> List[int]:
        """`length` piece ids made of shuffled whole bags, the last one cut short; leaves the bag empty."""
        sequence: List[int] = []
        while len(sequence) < length:
            self._refill()
            self._rng.shuffle(self.pieces)
            sequence.extend(self.pieces[: length - len(sequence)])
            self.pieces = []
        return sequence

    def reset(self) -> None:
        self.pieces.clear()

    def __len__(self) -> int:
        return len(self.pieces)
