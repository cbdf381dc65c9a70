"""Build and bind the C-ABI shared library (include/tetris_piclim.h) with ctypes.

The library is compiled in-tree for gfx950 with hipcc (<repo>/lib/libtetris_piclim.so -- a short path, so that the
line /proc/<pid>/maps shows for it is not truncated by tools that record which native code a process loaded) and
loaded with ctypes; there is no fallback: if it cannot be built or loaded the package raises.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess

# Kernel arguments in device memory (the ROCm 7 default on this hardware): with host-resident arguments every launch
# fetches them over PCIe, which adds 1.6 us to the 15.4-us step at 2^20 boards (measured on MI355X).  Read by the HIP
# runtime when it initialises, so it has to be in the environment before the first HIP call of the process.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")
_ROOT = os.path.dirname(_PKG)
# development switches: TPL_DIAG_CLOCK=1 builds a diagnostic library whose policy kernel stamps its clocks
# (tools/policy_clock.py); TPL_EXTRA_DEFINE=NAME builds libtetris_piclim_NAME.so with -DNAME for A/B runs on one box
_DIAG = os.environ.get("TPL_DIAG_CLOCK") == "1"
_EXTRA = os.environ.get("TPL_EXTRA_DEFINE", "")
_LIBDIR = os.path.join(_ROOT, "lib")
LIB_PATH = os.path.join(_LIBDIR, "libtetris_piclim_diag.so" if _DIAG else
                        f"libtetris_piclim_{_EXTRA}.so" if _EXTRA else "libtetris_piclim.so")
_UNITS = [os.path.join(_CSRC, f) for f in ("tetris_piclim.hip", "carve_generator.hip", "carve_device.hip",
                                           "forward_generator.hip", "forward_device.hip", "policy_mlp.hip", "policy_f32.hip", "policy_split.hip", "observe.hip")]
_SOURCES = _UNITS + [os.path.join(_CSRC, "tpl_device.h"), os.path.join(_CSRC, "tpl_internal.h"),
                     os.path.join(_CSRC, "tpl_step.h"), os.path.join(_CSRC, "tpl_observe.h"), os.path.join(_CSRC, "tpl_policy.h"), os.path.join(_CSRC, "py_random.h"),
                     os.path.join(_ROOT, "include", "tetris_piclim.h")]

# entry points declared in include/tetris_piclim.h (tests check that the .so exports every one of them)
SYMBOLS = [
    "tpl_last_error", "tpl_version", "tpl_workspace_bytes", "tpl_pool_bytes", "tpl_create", "tpl_destroy",
    "tpl_set_options", "tpl_load_configs", "tpl_reset", "tpl_move", "tpl_step", "tpl_step_observe", "tpl_get_state",
    "tpl_expand_obs", "tpl_expand_states", "tpl_get_board", "tpl_carve", "tpl_get_stats", "tpl_shape_info", "tpl_state_ptrs", "tpl_synth_configs",
    "tpl_synth_actions", "tpl_set_tuning", "tpl_rollout", "tpl_rollout_random", "tpl_rollout_trajectory", "tpl_rollout_random_trajectory", "tpl_decode_trajectory", "tpl_decode_actions", "tpl_generate_configs", "tpl_generate_configs_pyseed", "tpl_forward_generate",
    "tpl_forward_generate_device_work_bytes", "tpl_forward_generate_device",
    "tpl_generate_configs_device_work_bytes", "tpl_generate_configs_device", "tpl_generate_configs_device_waves",
    "tpl_policy_image_bytes", "tpl_policy_pack", "tpl_policy_act",
    "tpl_policy_image_bytes_f32", "tpl_policy_pack_f32", "tpl_policy_act_f32",
    "tpl_policy_image_bytes_split", "tpl_policy_pack_split", "tpl_policy_act_split",
    "tpl_explore_actions", "tpl_actor_rollout", "tpl_actor_rollout_f32", "tpl_actor_rollout_split", "tpl_pool_info", "tpl_pool_set_hold", "tpl_note_steps", "tpl_clock_ptr", "tpl_stream_create", "tpl_stream_destroy",
]

TPL_U8, TPL_I32, TPL_I64 = 0, 1, 2
TPL_F32, TPL_BF16 = 0, 1
TPL_ASSIGN_HASH, TPL_ASSIGN_SEQUENTIAL = 0, 1


class TplError(RuntimeError):
    """A C-ABI call returned a negative status; carries tpl_last_error()."""


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (no CPU fallback exists)")


def _source_digest() -> str:
    import hashlib
    h = hashlib.sha256()
    for path in _SOURCES:
        with open(path, "rb") as f:
            h.update(os.path.relpath(path, _ROOT).encode() + b"\0" + f.read())   # relative: the tree may be copied
    return h.hexdigest()


def build_library(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> <repo>/lib/libtetris_piclim.so.

    Staleness is decided by a digest of the sources kept next to the library (file times do not survive a copy of
    the tree), and concurrent callers -- the ranks of one torchrun job -- serialise on a lock file, so exactly one
    of them compiles and the others load the finished library."""
    stamp = LIB_PATH + ".sha256"
    digest = _source_digest()

    def fresh():
        return os.path.exists(LIB_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == digest

    if not force and fresh():
        return LIB_PATH
    import fcntl
    os.makedirs(_LIBDIR, exist_ok=True)
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or not fresh():
                tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
                cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", tmp] + _UNITS
                if _DIAG:
                    cmd.insert(1, "-DTPL_DIAG_CLOCK")
                if _EXTRA:
                    cmd.insert(1, "-D" + _EXTRA)
                res = subprocess.run(cmd, capture_output=True, text=True)
                if res.returncode != 0:
                    raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
                os.replace(tmp, LIB_PATH)
                with open(stamp + ".tmp", "w") as f:
                    f.write(digest)
                os.replace(stamp + ".tmp", stamp)
                if verbose:
                    print("built", LIB_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64/libhsa-runtime64; it must be the first HIP runtime in the process, so that
    # this library binds to the SAME runtime (shared streams, pointers, device context).  Loading the system ROCm
    # runtime first and torch's second leaves the process with two runtimes and no visible device.
    import torch  # noqa: F401
    L = C.CDLL(build_library())
    vp, i32, i64, u64, f32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_size_t
    L.tpl_last_error.restype = C.c_char_p
    L.tpl_last_error.argtypes = []
    L.tpl_version.restype = C.c_char_p
    L.tpl_version.argtypes = []
    L.tpl_workspace_bytes.restype = sz
    L.tpl_workspace_bytes.argtypes = [i64, i32]
    L.tpl_pool_bytes.restype = sz
    L.tpl_pool_bytes.argtypes = [i64, i32]
    L.tpl_create.argtypes = [C.POINTER(vp), i64, i32, i32, i32, i64, u64, vp, sz]
    L.tpl_destroy.argtypes = [vp]
    L.tpl_set_options.argtypes = [vp, i32, i32, f32, f32, f32]
    L.tpl_load_configs.argtypes = [vp, vp, vp, i64, vp, sz, vp]
    L.tpl_reset.argtypes = [vp, vp, vp]
    L.tpl_move.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp]
    L.tpl_step.argtypes = [vp, vp, i32, vp, vp, vp]
    L.tpl_step_observe.argtypes = [vp, vp, i32, vp, vp, vp, i32, vp]
    L.tpl_rollout.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp, vp]
    L.tpl_rollout_random.argtypes = [vp, u64, C.c_uint32, i32, vp, vp, vp, vp, vp, vp]
    L.tpl_rollout_trajectory.argtypes = [vp, vp, i64, i32, vp, vp, vp]
    L.tpl_rollout_random_trajectory.argtypes = [vp, u64, C.c_uint32, i32, vp, vp, vp, vp]
    L.tpl_decode_trajectory.argtypes = [vp, vp, i32, vp, vp, vp]
    L.tpl_get_state.argtypes = [vp] * 9
    L.tpl_expand_obs.argtypes = [vp, vp, i32, vp]
    L.tpl_expand_states.argtypes = [vp, vp, vp, i64, vp, i32, vp]
    L.tpl_decode_actions.argtypes = [vp, vp, i32, vp, vp]
    L.tpl_policy_image_bytes.restype = sz
    L.tpl_policy_image_bytes.argtypes = []
    L.tpl_policy_pack.argtypes = [vp] * 11
    L.tpl_policy_act.argtypes = [vp, vp, vp, vp, vp]
    L.tpl_policy_image_bytes_split.restype = sz
    L.tpl_policy_image_bytes_split.argtypes = []
    L.tpl_policy_pack_split.argtypes = [vp] * 11
    L.tpl_policy_act_split.argtypes = [vp, vp, vp, vp, vp]
    L.tpl_policy_image_bytes_f32.restype = sz
    L.tpl_policy_image_bytes_f32.argtypes = []
    L.tpl_policy_pack_f32.argtypes = [vp] * 11
    L.tpl_policy_act_f32.argtypes = [vp, vp, vp, vp, vp]
    L.tpl_explore_actions.argtypes = [vp, vp, f32, u64, C.c_uint32, vp]
    L.tpl_actor_rollout.argtypes = [vp, vp, i32, f32, u64, C.c_uint32, vp, vp, vp, vp, vp, vp]
    L.tpl_actor_rollout_f32.argtypes = [vp, vp, i32, f32, u64, C.c_uint32, vp, vp, vp, vp, vp, vp]
    L.tpl_actor_rollout_split.argtypes = [vp, vp, i32, f32, u64, C.c_uint32, vp, vp, vp, vp, vp, vp]
    L.tpl_get_stats.argtypes = [vp, vp, vp]
    L.tpl_shape_info.argtypes = [i32, i32, C.POINTER(i32), C.POINTER(i32), vp, vp]
    L.tpl_state_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.tpl_clock_ptr.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
    L.tpl_pool_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    L.tpl_pool_set_hold.argtypes = [vp, i64]
    L.tpl_note_steps.argtypes = [vp, i64]
    L.tpl_stream_create.argtypes = [i32, i32, i32, C.POINTER(vp)]
    L.tpl_stream_destroy.argtypes = [i32, vp]
    L.tpl_get_board.argtypes = [vp, vp, vp]
    L.tpl_carve.argtypes = [vp, i32, i32, i32, i32, C.POINTER(i32)]
    L.tpl_set_tuning.argtypes = [vp, i32, i32]
    L.tpl_generate_configs.argtypes = [i32, i32, u64, i64, i64, i32, i64, vp, vp, vp, vp]
    L.tpl_generate_configs_pyseed.argtypes = [i32, i32, vp, i64, i32, i64, vp, vp, vp, vp]
    L.tpl_generate_configs_device_work_bytes.restype = sz
    L.tpl_generate_configs_device_work_bytes.argtypes = [i32, i64]
    L.tpl_generate_configs_device.argtypes = [i32, i32, u64, i64, i64, i64, vp, vp, vp, vp, vp, vp, sz, vp]
    L.tpl_generate_configs_device_waves.argtypes = [i32, i32, u64, i64, i64, i64, i32, vp, vp, vp, vp, vp, vp, sz, vp]
    L.tpl_forward_generate.argtypes = [i32, i32, i32, i32, vp, i64, i32, vp, vp, vp, vp, vp, vp, vp]
    L.tpl_forward_generate_device_work_bytes.restype = sz
    L.tpl_forward_generate_device_work_bytes.argtypes = [i32, i64]
    L.tpl_forward_generate_device.argtypes = [i32, i32, i32, i32, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.tpl_synth_configs.argtypes = [vp, u64, i64, i64, vp, vp, vp]
    L.tpl_synth_actions.argtypes = [vp, u64, i64, i64, u64, vp, vp]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int:
            fn.restype = i32
    _lib = L
    return L


def check(status: int) -> None:
    if status != 0:
        raise TplError(f"tetris_piclim status {status}: {lib().tpl_last_error().decode()}")


def cpu_budget() -> int:
    """Host threads worth starting: the affinity mask, capped by the cgroup CPU quota when there is one (a
    container that sees 256 CPUs but may use 16 of them runs slower with 256 threads than with 16); TPL_CPU_BUDGET=<n>
    overrides both."""
    override = os.environ.get("TPL_CPU_BUDGET", "")
    if override.isdigit() and int(override) >= 1:
        return int(override)
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def carve(rows, piece: int, rotations: int, location: int, allow_partial: bool):
    """Tetris.carve (game/tetris.py:286-352) on one board given as uint16 rows[20] (host); returns (ok, rows_after)."""
    import numpy as np
    r = np.ascontiguousarray(rows, dtype=np.uint16).copy()
    ok = C.c_int32(0)
    check(lib().tpl_carve(r.ctypes.data_as(C.c_void_p), int(piece), int(rotations), int(location), int(bool(allow_partial)),
                          C.byref(ok)))
    return bool(ok.value), r


def generate_configs(L: int, M: int, count: int = 0, seed: int = 0, first: int = 0, threads: int = 0, *,
                     cutoff: int = 0, with_solutions: bool = False, python_seeds=None, max_iters: int = 0):
    """Carved (solvable) prescribed configurations, produced on the host cores (game/tetris.py:226-352).

    Returns (rows uint16 [count, 20], pieces uint8 [count, M+1]) and, with_solutions, also
    (solution uint8 [count, M, 2], solution_len int32 [count]).  Configuration i is the first attempt of (seed, first + i)
    that ends within the restart rule's iteration cut-off (`cutoff` overrides its base; 0 = by L).  Everything after `threads`
    is keyword only: through round 3 the seventh positional argument was `max_iters` (one search, 0 = unbounded), since round 4
    it is the restart rule's `cutoff` -- a positional caller would silently get the other meaning.  With
    python_seeds=[s0, s1, ...] configuration i is exactly what the reference builds after random.seed(s_i) (CPython's
    random stream is reproduced; one search each, `max_iters` > 0 bounds it)."""
    import numpy as np
    threads = threads or cpu_budget()
    if python_seeds is not None:
        seeds = np.ascontiguousarray(python_seeds, dtype=np.uint64)
        count = len(seeds)
    rows = np.empty((count, 20), np.uint16)
    pieces = np.empty((count, M + 1), np.uint8)
    sol = np.zeros((count, M, 2), np.uint8) if with_solutions else None
    sol_len = np.zeros(count, np.int32) if with_solutions else None
    ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    if python_seeds is not None:
        check(lib().tpl_generate_configs_pyseed(L, M, ptr(seeds), count, threads, max_iters, ptr(rows), ptr(pieces), ptr(sol), ptr(sol_len)))
    else:
        check(lib().tpl_generate_configs(L, M, seed, first, count, threads, cutoff, ptr(rows), ptr(pieces), ptr(sol), ptr(sol_len)))
    return (rows, pieces, sol, sol_len) if with_solutions else (rows, pieces)


def forward_generate(L: int, M: int, seeds, initial_height_max: int = 4, max_attempts: int = 1000, threads: int = 0):
    """The reference's forward generator + solver (game/tetris_algo_main/) for the given integer seeds.

    Returns a dict of numpy arrays over ALL seeds: rows [n, 20], sequence [n, M] (piece ids of Tetris.move),
    winnable [n] bool, failed_attempts [n], solution [n, M, 2] (rotations, location), solver_stack [n, M, 3],
    solution_len [n].  Game i equals TetrisGameGenerator(seed=seeds[i], ...) and its TetrisSolver verdict."""
    import numpy as np
    threads = threads or cpu_budget()
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    n = len(seeds)
    out = dict(rows=np.zeros((n, 20), np.uint16), sequence=np.zeros((n, M), np.uint8), winnable=np.zeros(n, np.uint8),
               failed_attempts=np.zeros(n, np.int32), solution=np.zeros((n, M, 2), np.uint8),
               solver_stack=np.zeros((n, M, 3), np.uint8), solution_len=np.zeros(n, np.int32))
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    check(lib().tpl_forward_generate(L, M, initial_height_max, max_attempts, ptr(seeds), n, threads, ptr(out["rows"]),
                                     ptr(out["sequence"]), ptr(out["winnable"]), ptr(out["failed_attempts"]),
                                     ptr(out["solution"]), ptr(out["solver_stack"]), ptr(out["solution_len"])))
    out["winnable"] = out["winnable"].astype(bool)
    return out


def pack_policy(params, f32=False):
    """Five (weight, bias) pairs of Model(217, 14) (float32 numpy, torch layout) -> packed image (numpy uint8): for the
    bf16 kernel (weights rounded to bf16), f32=True for the float32 kernel (weights as they are), or f32="split" for the
    kernel that reaches float32 accuracy on the bf16 matrix pipe (every weight as three bf16 pieces)."""
    import numpy as np
    flat = []
    for w, b in params:
        flat += [np.ascontiguousarray(w, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)]
    shapes = [a.shape for a in flat]
    want = [(128, 217), (128,), (128, 128), (128,), (128, 128), (128,), (128, 128), (128,), (14, 128), (14,)]
    if shapes != want:
        raise ValueError(f"policy parameters must have shapes {want}, got {shapes}")
    if f32 == "split":
        size, pack = lib().tpl_policy_image_bytes_split(), lib().tpl_policy_pack_split
    elif f32:
        size, pack = lib().tpl_policy_image_bytes_f32(), lib().tpl_policy_pack_f32
    else:
        size, pack = lib().tpl_policy_image_bytes(), lib().tpl_policy_pack
    image = np.empty(size, np.uint8)
    check(pack(*[a.ctypes.data_as(C.c_void_p) for a in flat], image.ctypes.data_as(C.c_void_p)))
    return image


def shape_info(piece: int, rotations: int):
    """get_tetromino(piece, rotations) as the device table encodes it (host-side decode, no GPU needed)."""
    h, w = C.c_int32(), C.c_int32()
    masks = (C.c_uint8 * 4)()
    topo = (C.c_uint8 * 4)()
    check(lib().tpl_shape_info(piece, rotations, C.byref(h), C.byref(w), masks, topo))
    return h.value, w.value, list(masks)[: h.value], list(topo)[: w.value]
