"""RandomPieceGenerator: the reference's own two tests (game/main.py:6-29) as it wrote them, and equality with the
reference's output under the same random.seed (tests/golden/pieces.npz, made by make_golden_pieces.py)."""
import os
import random
import sys
import unittest

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tetris_piclim as tetris  # noqa: E402


class TestRandomPieceGeneration(unittest.TestCase):
    def test_regenerate(self):
        random_piece_generator = tetris.RandomPieceGenerator()

        for i in range(16):
            (random_piece, random_piece_index), regenerated = random_piece_generator.get_random_piece()
            self.assertEqual(regenerated, i % 7 == 0, 'Regeneration notification failed')

            self.assertEqual(len(random_piece_generator), 7 - (i % 7), 'Regeneration failed')

            random_piece_generator.delete_index(random_piece_index)

            self.assertEqual(len(random_piece_generator), 7 - (i % 7) - 1, 'Deletion failed')

    def test_sequence(self):
        random_piece_generator = tetris.RandomPieceGenerator()
        sequence_length = 16
        sequence = random_piece_generator.get_random_sequence(sequence_length)

        self.assertEqual(len(sequence), sequence_length, 'Sequence of wrong length')

        for i in range(0, len(sequence), 7):
            permutation_group = sequence[i:i+7]
            self.assertEqual(len(permutation_group), len(set(permutation_group)), 'Groups contain duplicates')


def test_same_pieces_as_the_reference_under_the_same_seed():
    f = np.load(os.path.join(ROOT, "tests", "golden", "pieces.npz"))
    for k in f["seeds"]:
        random.seed(int(k))
        assert tetris.RandomPieceGenerator().get_random_sequence(41) == f["sequences"][k].tolist()
        random.seed(int(k))
        gen = tetris.RandomPieceGenerator()
        for i in range(30):
            (piece, index), regenerated = gen.get_random_piece()
            assert (piece, index, int(regenerated)) == tuple(int(x) for x in f["draws"][k, i])
            gen.delete_index(index)
