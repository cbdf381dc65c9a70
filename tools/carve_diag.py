#!/usr/bin/env python3
"""Where a launch of the device carving generator spends its time (diagnostic build: TPL_EXTRA_DEFINE=TPL_CARVE_DIAG):
the moment the first lane finds the queue empty against the launch's start and end (100-MHz wall clock), the search
iterations executed in total (against the configurations' own: the oracle says what a batch needs), the attempts begun by
lanes that had run out of work and the attempts dropped because a lower one had finished.
    TPL_EXTRA_DEFINE=TPL_CARVE_DIAG python tools/carve_diag.py [--count 1048576] [--L 10] [--M 40] [--waves 0] [--cutoff 0]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--count", type=int, default=1 << 20)
    ap.add_argument("--L", type=int, default=10)
    ap.add_argument("--M", type=int, default=40)
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--cutoff", type=int, default=0)
    ap.add_argument("--launches", type=int, default=2)
    args = ap.parse_args()
    assert os.environ.get("TPL_EXTRA_DEFINE") == "TPL_CARVE_DIAG", "run with TPL_EXTRA_DEFINE=TPL_CARVE_DIAG"
    import torch
    import tetris_piclim as T
    lib = T._lib.lib()
    d = torch.device("cuda", 0)
    n, M = args.count, args.M
    rows = torch.empty((n, 20), dtype=torch.int16, device=d)
    pieces = torch.empty((n, M + 1), dtype=torch.uint8, device=d)
    status = torch.empty(n, dtype=torch.int32, device=d)
    nbytes = lib.tpl_generate_configs_device_work_bytes(M, n)
    work = torch.empty(nbytes, dtype=torch.uint8, device=d)
    stride = (256 + 512 + (M // 7 + 3) * 44 + 63) // 64 * 64
    slices = min(((n + 63) // 64 + 3) // 4 * 4, 4096) * 64
    for k in range(args.launches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        T._lib.check(lib.tpl_generate_configs_device_waves(args.L, M, 7, k * n, n, args.cutoff, args.waves, C.c_void_p(rows.data_ptr()),
                                                           C.c_void_p(pieces.data_ptr()), None, None, C.c_void_p(status.data_ptr()),
                                                           C.c_void_p(work.data_ptr()), nbytes, None))
        e1.record()
        torch.cuda.synchronize()
        ctl = work[stride * slices: stride * slices + 64].view(torch.int64).cpu().numpy().astype("uint64")
        inv = lambda v: int((~v) & 0xFFFFFFFFFFFFFFFF)
        t_dry, t_end, iters, helped, dropped, t0 = inv(ctl[2]), int(ctl[3]), int(ctl[4]), int(ctl[5]), int(ctl[6]), inv(ctl[7])
        print(f"L={args.L} M={M} count={n} waves={args.waves or 'auto'} cutoff={args.cutoff or 'by L'}: {e0.elapsed_time(e1):.2f} ms by events; "
              f"queue dry {(t_dry - t0) / 100e3:.2f} ms after the first wave started, last wave out at {(t_end - t0) / 100e3:.2f} ms; "
              f"{iters / n:.0f} iterations per configuration, {helped} attempts by lanes out of work, {dropped} dropped; capped {int(status.sum())}",
              flush=True)
        waves = args.waves or (n + 63) // 64
        waves = (max(1, min(waves, (n + 63) // 64, 4096)) + 3) // 4 * 4
        import numpy as np
        wd = work[stride * slices + 64: stride * slices + 64 + waves * 32].view(torch.int64).cpu().numpy().reshape(waves, 4)
        if os.environ.get("TPL_CARVE_DIAG_DUMP"):
            np.save(os.environ["TPL_CARVE_DIAG_DUMP"], np.concatenate([wd, np.full((waves, 1), t0, dtype=np.int64)], axis=1))
        tail_ms = (wd[:, 1] - wd[:, 0]) / 100e3
        q = lambda a: " ".join(f"{v:.2f}" for v in np.percentile(a, [0, 10, 50, 90, 99, 100]))
        print(f"    per wave (min p10 p50 p90 p99 max): out of the loop {q((wd[:, 1] - t0) / 100e3)} ms after the start; {q(tail_ms)} ms after "
              f"its first lane found the queue dry; trips after that {q(wd[:, 3])}; us per trip after that {q(tail_ms * 1e3 / np.maximum(wd[:, 3], 1))}; "
              f"us per trip before {q((wd[:, 0] - t0) / 100.0 / np.maximum((32 if args.L >= 8 else 8) * wd[:, 2] - wd[:, 3], 1))}", flush=True)


if __name__ == "__main__":
    main()
