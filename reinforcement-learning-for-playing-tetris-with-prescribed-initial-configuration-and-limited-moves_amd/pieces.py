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
