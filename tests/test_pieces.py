"""RandomPieceGenerator: the two properties the reference tests (game/main.py:6-29), stated here in this repo's own
words, and equality with the reference's output under the same random.seed (tests/golden/pieces.npz, made by
make_golden_pieces.py)."""
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tetris_piclim as tetris  # noqa: E402


def test_bag_refills_every_seventh_draw_and_shrinks_by_one_per_deletion():
    """Property of the reference's bag (game/tetris.py:64-108; its test_regenerate checks the same thing): a draw reports
    a refill exactly when the bag was empty, i.e. on draws 0, 7, 14, ...; drawing does not remove, delete_index does."""
    bag = tetris.RandomPieceGenerator()
    for draw in range(23):
        left_in_bag = 7 - draw % 7
        (piece, index), refilled = bag.get_random_piece()
        assert bool(refilled) == (left_in_bag == 7), draw
        assert 0 <= piece <= 6 and 0 <= index < left_in_bag
        assert len(bag) == left_in_bag                      # the draw itself leaves the bag as it is
        bag.delete_index(index)
        assert len(bag) == left_in_bag - 1


@pytest.mark.parametrize("length", [1, 7, 16, 41])
def test_sequence_is_made_of_whole_bags(length):
    """get_random_sequence(n) (game/tetris.py:95-102): n ids, every aligned group of seven a permutation of 0..6 and
    the ragged tail free of repeats."""
    seq = tetris.RandomPieceGenerator().get_random_sequence(length)
    assert len(seq) == length and all(0 <= p <= 6 for p in seq)
    groups = [seq[i:i + 7] for i in range(0, length, 7)]
    assert all(sorted(g) == list(range(7)) for g in groups if len(g) == 7)
    assert all(len(set(g)) == len(g) for g in groups)


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


def test_translate_hands_over_the_winnable_forward_games_with_one_random_piece_in_front():
    """translate() (game/tetris.py:19-20) on the forward generator's games for the reference's own seeds 0..99 at L = 5, M = 20
    (tests/golden/forward_L5_M20.npz: 22 of them winnable): boards as 20x10 bool arrays, M + 1 piece ids of which pieces[1:] is
    the game's sequence in Tetris.move's ids and pieces[0] what random.randint(0, 6) gives next."""
    f = np.load(os.path.join(ROOT, "tests", "golden", "forward_L5_M20.npz"))
    games = tetris.forward_generate(5, 20, f["seeds"])
    random.seed(5)
    batch = tetris.translate(games)
    random.seed(5)
    leads = [random.randint(0, 6) for _ in batch]
    keep = np.flatnonzero(f["winnable"])
    assert len(batch) == len(keep) == 22
    letter_to_id = np.array([0, 2, 1, 6, 4, 3, 5])             # I J L O S T Z -> piece_translations
    for (board, pieces), k, lead in zip(batch, keep, leads):
        assert board.shape == (20, 10) and board.dtype == bool
        assert np.array_equal((board * (1 << np.arange(10))).sum(1), f["rows"][k])
        assert pieces == [lead] + letter_to_id[f["sequence"][k]].tolist() and len(pieces) == 21


def test_forward_games_draw_their_extra_piece_on_the_device_without_a_host_copy():
    """pool.ForwardGames.translate puts one random piece in front of every forward game (game/tetris.py:19-20).  Round-5 advisor
    finding: it drew that piece with numpy and copied it up from pageable memory (a host wait at every pool swap).  The draw is
    now tensor arithmetic on the games' device -- a splitmix64 hash of (seed, batch, game) in wrapping int64 -- so here, on the
    CPU: a function of its three keys alone, pieces 0..6 about evenly, different batches unrelated."""
    import torch
    import tetris_piclim as T
    games = T.pool.ForwardGames.__new__(T.pool.ForwardGames)
    games._torch, games.count, games.device, games._index = torch, 70000, torch.device("cpu"), None
    games.rows = torch.zeros((games.count, 20), dtype=torch.int16)
    games.sequence = torch.full((games.count, 20), 9, dtype=torch.uint8)
    a = games._draw(3, 5).numpy().ravel()
    assert a.dtype == np.uint8 and a.min() == 0 and a.max() == 6
    assert np.abs(np.bincount(a, minlength=7) / games.count - 1 / 7).max() < 0.006          # ~4 sigma at 70,000 draws
    assert np.array_equal(games._draw(3, 5).numpy().ravel(), a)
    for other in (games._draw(3, 6), games._draw(4, 5)):
        assert abs((other.numpy().ravel() == a).mean() - 1 / 7) < 0.01
    assert np.array_equal(games._draw(3, 5).numpy().ravel()[:1000], games.__class__._draw(games, 3, 5)[:1000].numpy().ravel())
    rows, pieces = games.translate(3, 5)
    assert pieces.shape == (games.count, 21) and np.array_equal(pieces[:, 0].numpy(), a) and (pieces[:, 1:] == 9).all()
    _, tail = games.translate(3, 5, lead=False)
    assert np.array_equal(tail[:, -1].numpy(), a) and (tail[:, :-1] == 9).all()
