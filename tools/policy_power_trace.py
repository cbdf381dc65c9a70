#!/usr/bin/env python3
"""An INDEPENDENT look at the clock and power the chip holds under the bf16 policy kernel: while the main thread launches
tpl_policy_act back to back for a few seconds, a sampler thread reads what the driver publishes in sysfs for every card
it can see -- hwmon freq1_input (current shader clock), power1_average / power1_input, power1_cap, and the
pp_dpm_sclk level marked current -- every few milliseconds.  Printed: the idle readings before the loop, the
distribution of the readings during it, and the kernel's duration in the same window.  (The in-kernel clock of
tools/policy_clock.py -- s_memtime over s_memrealtime in a diagnostic build -- is the other witness.)
    python tools/policy_power_trace.py [--seconds 3]"""
import argparse
import glob
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.getcwd())


def read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def cards():
    out = []
    for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
        if hw:
            out.append((dev, hw[0]))
    return out


def sample(dev, hw):
    s = {}
    v = read(os.path.join(hw, "freq1_input"))
    if v and v.isdigit():
        s["sclk_mhz"] = int(v) / 1e6
    for name in ("power1_average", "power1_input"):
        v = read(os.path.join(hw, name))
        if v and v.isdigit():
            s["power_w"] = int(v) / 1e6
            break
    v = read(os.path.join(hw, "power1_cap"))
    if v and v.isdigit():
        s["power_cap_w"] = int(v) / 1e6
    v = read(os.path.join(dev, "pp_dpm_sclk"))
    if v:
        cur = [l for l in v.splitlines() if l.rstrip().endswith("*")]
        if cur:
            s["dpm_sclk"] = cur[0].split(":")[1].replace("*", "").strip()
    v = read(os.path.join(dev, "gpu_busy_percent"))
    if v and v.isdigit():
        s["busy"] = int(v)
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--boards", type=int, default=262144)
    args = ap.parse_args()
    import torch
    import tetris_piclim as T
    cs = cards()
    n = args.boards
    env = T.BatchedTetris(10, 40, n, auto_reset=True)
    rows, pieces = env.synthetic_configs(n)
    env.load_configs(rows, pieces)
    env.reset()
    torch.manual_seed(0)
    image = T.actor.policy_image(T.PolicyMLP(), env.device)
    act = torch.empty(n, dtype=torch.uint8, device=env.device)
    for _ in range(5):
        env.policy_act(image, out=act)
    torch.cuda.synchronize()
    time.sleep(1.0)
    idle = [sample(*c) for c in cs]
    stop = threading.Event()
    trace = [[] for _ in cs]

    def sampler():
        while not stop.is_set():
            t = time.perf_counter()
            for k, c in enumerate(cs):
                s = sample(*c)
                s["t"] = t
                trace[k].append(s)
            time.sleep(0.005)

    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    launches = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < args.seconds:
        for _ in range(200):
            env.policy_act(image, out=act)
        launches += 200
        torch.cuda.synchronize()                      # (keeps the host's queue short: the loop ends when the clock says so)
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    us = e0.elapsed_time(e1) * 1e3 / launches
    tf = 2.0 * (217 * 128 + 3 * 128 * 128 + 128 * 14) * n / (us * 1e-6) / 1e12
    print(json.dumps({"kernel": "tpl::p16::policy_kernel", "boards": n, "launches": launches, "us_per_launch_incl_sync_gaps": us,
                      "tflops": tf, "frac_of_2500": tf / 2500}))

    def dist(vals):
        vals = sorted(vals)
        if not vals:
            return None
        q = lambda f: vals[min(len(vals) - 1, int(f * len(vals)))]
        return {"n": len(vals), "min": vals[0], "p10": q(0.1), "median": q(0.5), "p90": q(0.9), "max": vals[-1]}

    for k, (dev, hw) in enumerate(cs):
        tr = [s for s in trace[k] if s["t"] - t0 > 0.3]             # past the ramp
        row = {"card": dev, "idle": idle[k],
               "under_load": {key: dist([s[key] for s in tr if key in s]) for key in ("sclk_mhz", "power_w", "busy")},
               "dpm_sclk_levels_seen": sorted({s.get("dpm_sclk") for s in tr if s.get("dpm_sclk")})}
        loaded = row["under_load"]["power_w"]
        if loaded and idle[k].get("power_w") is not None and loaded["median"] < idle[k]["power_w"] + 50:
            row["note"] = "power did not move: not the card this process runs on"
        print(json.dumps(row))
    env.terminate()


if __name__ == "__main__":
    main()
