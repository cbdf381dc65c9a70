#!/usr/bin/env python3
"""Replay of the carving kernel's schedule (csrc/carve_device.hip) from measured search lengths, for a GROUP of `lanes`
lanes that share their idle lanes (64 = what the kernel does: helping stays inside a wave; 256 = a block of four waves
pooled): how long after the launch's start does the group's last configuration get its answer, and how much of the
lanes' time was useful.  Time is counted in search trips; the restart rule is the product's (tpl_device.h): attempt a of a
configuration may use c trips for a < 12, 2c for 12..17, 4c for 18..23; the answer is the lowest attempt that ends within
its cut-off, known once every lower one has failed.

    tools/carve_pool_sim.py [per_lane=1] [cutoff=3328] [groups=400] [schedule, e.g. 4x5,4x10,4x16,6x32,6x64]

`per_lane` = configurations per lane of the launch (1: the 262,144-configuration batch on 4096 waves; 4: 2^20).  The lengths
are those of 20,000 configurations at L = 10, M = 40 without the restart rule (profiles/r04_carve/search_lengths_L10_M40.npz,
what iteration_histogram.json summarises)."""
import heapq
import os
import sys

import numpy as np

its = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_carve",
                           "search_lengths_L10_M40.npz"))["iterations"].astype(np.int64)
per_lane = int(sys.argv[1]) if len(sys.argv) > 1 else 1
C = int(sys.argv[2]) if len(sys.argv) > 2 else 3328
groups = int(sys.argv[3]) if len(sys.argv) > 3 else 400
A, BURST = 24, 32


# the cut-off schedule: the product's, or (4th argument) one to try, as comma-separated "count x multiplier of C in sixteenths"
SCHEDULE = None
if len(sys.argv) > 4:
    SCHEDULE = []
    for part in sys.argv[4].split(","):
        cnt, mult = part.split("x")
        SCHEDULE += [max(1, C * int(mult) // 16)] * int(cnt)
    assert len(SCHEDULE) == A, len(SCHEDULE)


def limit(a):
    if SCHEDULE is not None:
        return SCHEDULE[a]
    return C << (0 if a < 12 else (a - 12) // 6 + 1)


def run(lanes, helpers, rng, order="fewest", reserve=0):
    total = lanes * per_lane
    taken = 0
    cfg = []                    # per configuration: dict(att={a: [start, end, ok, lane, dropped]}, ticket, done, home)
    job = [None] * lanes
    ev = []
    idle = set()
    busy_trips = 0

    def start(lane, c, a, now):
        x = int(its[rng.integers(len(its))])
        ok = x <= limit(a)
        dur = x if ok else limit(a)
        dur = (dur + BURST - 1) // BURST * BURST          # an attempt's end is seen at the end of its burst
        cfg[c]["att"][a] = [now, now + dur, ok, lane, False]
        job[lane] = (c, a)
        heapq.heappush(ev, (now + dur, lane, c, a))

    def take(lane, now):
        nonlocal taken
        if taken >= total:
            return False
        taken += 1
        cfg.append({"att": {}, "ticket": 0, "done": None, "home": lane})
        start(lane, len(cfg) - 1, 0, now)
        return True

    def finished(c, now):
        return any(v[2] and v[1] <= now and not v[4] for v in cfg[c]["att"].values())

    def resolve(c, now):
        cf = cfg[c]
        fin = [a for a, v in cf["att"].items() if v[2] and v[1] <= now and not v[4]]
        if fin:
            w = min(fin)
            if all(b in cf["att"] and cf["att"][b][1] <= now and not cf["att"][b][2] for b in range(w)):
                cf["done"] = now

    def help_(now):
        while idle:
            cands = []
            for c, cf in enumerate(cfg):
                if cf["done"] is not None or finished(c, now) or cf["ticket"] + 1 >= A:
                    continue
                flying = sum(1 for v in cf["att"].values() if v[1] > now and not v[4])
                if flying >= helpers:
                    continue
                cands.append((flying if order == "fewest" else 0, c))
            if not cands:
                return
            cands.sort()
            level = cands[0][0]
            for f, c in cands:
                if f != level or not idle:
                    break
                lane = idle.pop()
                cfg[c]["ticket"] += 1
                start(lane, c, cfg[c]["ticket"], now)

    for lane in range(lanes - reserve):
        take(lane, 0)
    for lane in range(lanes - reserve, lanes):
        idle.add(lane)
    help_(0)
    end = 0
    while ev:
        now, lane, c, a = heapq.heappop(ev)
        v = cfg[c]["att"][a]
        if v[4] or job[lane] != (c, a):
            continue
        busy_trips += v[1] - v[0]
        job[lane] = None
        cf = cfg[c]
        if cf["done"] is None:
            resolve(c, now)
        if v[2]:
            for b, u in cf["att"].items():
                if b > a and u[1] > now and not u[4]:
                    u[4] = True
                    busy_trips += now - u[0]
                    job[u[3]] = None
                    idle.add(u[3])
        if cf["done"] is None and not v[2] and cf["home"] == lane and not finished(c, now) and cf["ticket"] + 1 < A:
            cf["ticket"] += 1
            start(lane, c, cf["ticket"], now)
        elif not take(lane, now):
            idle.add(lane)
        if taken >= total:
            help_(now)
        end = now
    end = max(cf["done"] for cf in cfg)
    useful = sum(min(v[1] for v in cf["att"].values() if v[2] and v[1] <= cf["done"]) - 0 for cf in cfg)
    return end, busy_trips / (end * lanes)


rng = np.random.default_rng(7)
print(f"{per_lane} configuration(s) per lane, cut-off {C}, {groups} groups each; a launch of 4096 waves ends with its slowest group")
CASES = ((64, 12, "fewest", 0),) if SCHEDULE is not None else ((64, 12, "fewest", 0), (64, 4, "fewest", 0), (64, 24, "fewest", 0), (128, 12, "fewest", 0),
                                       (256, 12, "fewest", 0), (256, 24, "fewest", 0), (1024, 12, "fewest", 0))
for lanes, helpers, order, reserve in CASES:
    n = max(8, groups * 64 // lanes)
    r = [run(lanes, helpers, rng, order, reserve) for _ in range(n)]
    ends = np.array([x[0] for x in r])
    # the launch holds 262144 lanes: its end = the maximum over 262144 / lanes groups; estimate by the matching quantile
    ngroups = 262144 // lanes
    q = 1.0 - 1.0 / ngroups
    print(f"lanes {lanes:5d} helpers {helpers:2d}: group end median {np.median(ends):7.0f}  p90 {np.percentile(ends, 90):7.0f}  max of {n} {ends.max():7.0f}"
          f"  (launch ~ quantile {q:.5f}: {np.quantile(ends, min(q, 1.0)):7.0f})  lanes busy {np.mean([x[1] for x in r]):.2f}")
