"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/tetris_piclim.h declares; the device shape table decodes to the reference's table; the product never
touches the oracle; failures are loud.  No compute call is made here (no GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden

import tetris_piclim as T

PKG = os.path.dirname(T.__file__)


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "tetris_piclim.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tpl_[a-z_0-9]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    path = T.build_library()
    assert os.path.exists(path) and path.startswith(os.path.join(ROOT, "lib"))   # in-tree (short path), not site-packages
    lib = ctypes.CDLL(path)
    declared = _declared_symbols()
    assert declared, "header parse found nothing"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/tetris_piclim.h but not exported"
    assert sorted(T.SYMBOLS) == declared                              # the binding covers the whole header


def test_code_object_targets_gfx950_only():
    blob = open(T.build_library(), "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_"):
        assert other not in blob


def test_device_shape_table_matches_reference_table():
    f = load_golden("shapes.npz")
    for p in range(7):
        for r in range(12):
            h, w, masks, topo = T.shape_info(p, r)
            rr = r % int(f["nrot"][p])
            assert (h, w) == (f["h"][p, rr], f["w"][p, rr])
            assert masks == f["mask"][p, rr, :h].tolist() and topo == f["revtopo"][p, rr, :w].tolist()
    with pytest.raises(T.TplError):
        T.shape_info(7, 0)
    # the reference's module-level accessor, same table
    mask, topo = T.get_tetromino(1, 5)                       # L after 5 quarter turns = after 1
    assert mask.dtype == bool and mask.shape == (3, 2) and mask.tolist() == [[True, True], [False, True], [False, True]]
    assert topo == (0, 2) and T.piece_translations["T"] == 3


def test_sizes_and_argument_errors_without_gpu():
    lib = T._lib.lib()
    n, M = 1 << 20, 40
    ws = lib.tpl_workspace_bytes(n, M)
    assert n * 32 + n // 4 <= ws < n * 32 + n // 4 + (1 << 16)        # 32 B of resident state per board + an 8-B step clock per 32
    assert lib.tpl_pool_bytes(1000, 49) == 1000 * 128 and lib.tpl_pool_bytes(1000, 50) == 1000 * 192
    assert lib.tpl_pool_bytes(1000, M) == 1000 * (64 + 64)            # a 64-B record + a 64-B side record per configuration at M=40
    assert lib.tpl_pool_bytes(1000, 254) == 1000 * (256 + 64)          # record strides are powers of two
    assert lib.tpl_workspace_bytes(0, M) == 0
    h = ctypes.c_void_p()
    assert lib.tpl_create(ctypes.byref(h), 0, 10, 40, 0, 0, 0, None, 0) < 0        # bad num_envs
    assert b"num_envs" in lib.tpl_last_error()
    assert lib.tpl_create(ctypes.byref(h), 16, 0, 40, 0, 0, 0, None, 0) < 0         # bad L
    assert lib.tpl_create(ctypes.byref(h), 16, 10, 255, 0, 0, 0, None, 0) < 0       # bad M
    assert lib.tpl_step(None, None, 0, None, None, None) < 0 and b"null" in lib.tpl_last_error()
    assert lib.tpl_destroy(None) == 0


def test_round_4_entry_points_refuse_bad_arguments_without_a_gpu():
    """The entry points added in round 4 -- compact trajectory, float32 / split megakernels, split policy -- check their
    arguments before they touch the device: callable (and refusing) in a process that has no GPU."""
    lib = T._lib.lib()
    assert lib.tpl_rollout_trajectory(None, None, 0, 1, None, None, None) < 0 and b"null" in lib.tpl_last_error()
    assert lib.tpl_rollout_random_trajectory(None, 0, 0, 1, None, None, None, None) < 0
    assert lib.tpl_decode_trajectory(None, None, 1, None, None, None) < 0
    assert lib.tpl_actor_rollout_f32(None, None, 1, 0.0, 0, 0, None, None, None, None, None, None) < 0
    assert lib.tpl_actor_rollout_split(None, None, 1, 0.0, 0, 0, None, None, None, None, None, None) < 0
    assert lib.tpl_policy_act_split(None, None, None, None, None) < 0
    assert lib.tpl_policy_pack_split(*([None] * 11)) < 0 and b"null" in lib.tpl_last_error()
    # three bf16 planes of the bf16 image's fragments + the float32 biases
    assert lib.tpl_policy_image_bytes_split() == 3 * (lib.tpl_policy_image_bytes() - (4 * 128 + 16) * 4) + (4 * 128 + 16) * 4
    # the generators: a cut-off beyond 2^28 (the device counts trips in 32 bits) and a negative one are refused
    import numpy as np
    rows, pieces = np.zeros((2, 20), np.uint16), np.zeros((2, 41), np.uint8)
    for cutoff in (-1, (1 << 28) + 1):
        rc = lib.tpl_generate_configs(10, 40, 0, 0, 2, 1, cutoff, rows.ctypes.data, pieces.ctypes.data, None, None)
        assert rc < 0 and b"cutoff" in lib.tpl_last_error()


def test_product_never_touches_the_oracle_or_a_cpu_fallback():
    for base, _, files in os.walk(PKG):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(base, fn)).read()
                assert "oracle" not in text.lower(), f"{fn} mentions the oracle"
    src = open(os.path.join(PKG, "env.py")).read()
    assert "no CPU path" in src


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises((RuntimeError, ValueError)):
        T.BatchedTetris(5, 20, 16)
    with pytest.raises(ValueError):
        T.BatchedTetris(5, 20, 16, device="cpu")


def test_snapshot_object_carries_its_tags_without_a_gpu():
    """BatchedTetris.snapshot() returns a Snapshot: the copy of the state together with the pool generation, assignment
    mode and swap-guard value it was taken under; clone() and to() keep them (attributes hung on a tensor were lost)."""
    import torch
    snap = T.Snapshot(torch.arange(12, dtype=torch.uint8), 3, "hash", 17)
    for other in (snap.clone(), snap.to(torch.device("cpu")), snap.clone().to("cpu")):
        assert isinstance(other, T.Snapshot)
        assert (other.pool_generation, other.assign, other.hold) == (3, "hash", 17)
        assert torch.equal(other.data, snap.data)
    assert snap.clone().data.data_ptr() != snap.data.data_ptr()


def test_generator_work_memory_is_sized_by_lanes_not_by_configurations():
    """The device generator is persistent: its work memory is one slice per LANE (at most 4096 waves of 64), not per
    configuration -- 2^20 configurations need what 262,144 need."""
    lib = T._lib.lib()
    small, big, huge = (lib.tpl_generate_configs_device_work_bytes(40, c) for c in (1000, 1 << 18, 1 << 20))
    assert 0 < small < big == huge
    assert lib.tpl_generate_configs_device_work_bytes(0, 10) == 0 and lib.tpl_generate_configs_device_work_bytes(40, 0) == 0


def test_bench_names_the_host_cpu():
    import bench_side as bench
    model = bench.cpu_model()
    assert model is None or (isinstance(model, str) and model.strip())
