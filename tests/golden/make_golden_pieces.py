#!/usr/bin/env python3
"""Fixture for the piece supply: what the REFERENCE's RandomPieceGenerator (game/tetris.py:64-108) produces under
random.seed(k), k = 0..31 -- a 41-piece sequence, and 30 single draws each followed by delete_index.

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_pieces.py
"""
import os
import random
import sys

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(os.environ.get("TPL_REFERENCE", "/root/reference"), "game"))
HERE = os.path.dirname(os.path.abspath(__file__))
os.chdir("/tmp")
import tetris as ref  # noqa: E402

seeds = np.arange(32)
sequences = np.zeros((len(seeds), 41), np.uint8)
draws = np.zeros((len(seeds), 30, 3), np.uint8)          # piece, index, regenerated
for k in seeds:
    random.seed(int(k))
    sequences[k] = ref.RandomPieceGenerator().get_random_sequence(41)
    random.seed(int(k))
    gen = ref.RandomPieceGenerator()
    for i in range(30):
        (piece, index), regenerated = gen.get_random_piece()
        draws[k, i] = (piece, index, int(regenerated))
        gen.delete_index(index)
np.savez_compressed(os.path.join(HERE, "pieces.npz"), seeds=seeds, sequences=sequences, draws=draws)
print("pieces.npz:", sequences.shape, draws.shape)
