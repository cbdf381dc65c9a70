#!/usr/bin/env python3
"""profiles/valu_issue.json -- what bench.py prices its `valu-issue` rooflines with -- from the per-form summaries of
tools/profile_valu.sh and the static instruction mix of tools/valu_mix.py:

    python tools/update_valu_issue.py gpurun_out/prof_<tag> profiles/r05_valu

For every form: vector instructions per launch (SQ_INSTS_VALU) and per UNIT (a board-step, or a configuration), the launch
duration (kernel trace), the shader clock the chip held (GRBM_GUI_ACTIVE is summed over the 8 XCDs), lanes active per vector
instruction, and the issue cost of the kernel's instruction mix.  The bound: a SIMD issues one wave64 vector instruction per
`cycles_per_valu_instruction` cycles at best, the chip has 1,024 SIMDs -- so a launch cannot be shorter than
    valu_per_launch x cycles_per_valu / (1024 x clock),   and   frac = that / its duration."""
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from valu_mix import mix  # noqa: E402

SIMDS, PEAK_CLOCK_GHZ = 1024, 2.4
FORMS = {   # form -> (kernel as the trace names it, mangled name for the mix, units per launch, what a unit is, bench key)
    "rollout_f32_u8_50": ("rollout_kernel<true, false, false>", "rollout_kernelILb1ELb0ELb0", (1 << 20) * 50, "board-step", "fused_rollout"),
    "rollout_compact_50": ("rollout_kernel<true, false, true>", "rollout_kernelILb1ELb0ELb1", (1 << 20) * 50, "board-step", "fused_rollout.compact_trajectory"),
    "rollout_random_100": ("rollout_kernel<true, true, false>", "rollout_kernelILb1ELb1ELb0", (1 << 20) * 100, "board-step", "fused_rollout.device_random_policy"),
    "rollout_shard_131072": ("rollout_kernel<true, false, false>", "rollout_kernelILb1ELb0ELb0", 131072 * 50, "board-step", "shard_run.tpl_rollout"),
    "carve_1048576": ("carve_kernel", "carve_kernel", 1 << 20, "configuration", "config_supply.carve_device"),
    "carve_262144": ("carve_kernel", "carve_kernel", 1 << 18, "configuration", "config_supply.carve_device.batch_of_262144"),
}


def main():
    src, dst = sys.argv[1].rstrip("/"), sys.argv[2].rstrip("/")
    os.makedirs(dst, exist_ok=True)
    out = {"formula": "frac = valu_per_launch x cycles_per_valu_instruction / (1024 SIMDs x clock) / duration; peak = 1024 x 2.4 GHz / "
                      "cycles_per_valu_instruction wave-instructions/s; cycles_per_valu_instruction = the kernel's static instruction mix "
                      "priced with the measured issue costs of profiles/r04_final/valu_rates_gfx950.log (tools/valu_mix.py)",
           "forms": {}}
    mixes = {}
    for form, (kernel, mangled, units, unit, key) in FORMS.items():
        path = os.path.join(src, form, "summary.json")
        if not os.path.exists(path):
            continue
        s = json.load(open(path))
        shutil.copy(path, os.path.join(dst, f"summary_{form}.json"))
        kern = max((k for k in s["kernels"] if kernel in k["name"]), key=lambda k: k["grid"], default=None)
        ckey = next((k for k in s["counters"] if kernel in k and (kern is None or k.endswith(f"grid={kern['grid']}"))), None)
        if kern is None or ckey is None:
            continue
        c = s["counters"][ckey]
        if mangled not in mixes:
            mixes[mangled] = mix(mangled, "lib/libtetris_piclim.so")
        m = mixes[mangled]
        dur_ns = kern["median_ns"]
        clock = c["GRBM_GUI_ACTIVE"] / 8.0 / dur_ns                       # GHz
        valu = c["SQ_INSTS_VALU"]
        bound_ns_at_held = valu * m["cycles_per_valu_instruction"] / (SIMDS * clock)
        bound_ns_at_peak = valu * m["cycles_per_valu_instruction"] / (SIMDS * PEAK_CLOCK_GHZ)
        out["forms"][form] = {
            "bench_key": key, "kernel": kernel, "grid": kern["grid"], "unit": unit, "units_per_launch": units,
            "valu_per_launch": valu, "valu_per_unit": valu / units, "valu_per_wave_per_unit_of_64": valu / units * 64,
            "duration_ns_median": dur_ns, "launches_traced": kern["calls"], "clock_GHz_held": clock,
            "cycles_per_valu_instruction": m["cycles_per_valu_instruction"], "static_valu_instructions": m["static_valu_instructions"],
            "lanes_active_per_valu_instruction": c.get("lanes_active_per_valu_instruction"),
            "frac_at_clock_held": bound_ns_at_held / dur_ns, "frac": bound_ns_at_peak / dur_ns,
            "stamp": s["stamp"], "source": os.path.join(dst, f"summary_{form}.json")}
    json.dump(out, open(os.path.join(os.path.dirname(dst), "valu_issue.json"), "w"), indent=1)
    print(json.dumps({f: {k: v[k] for k in ("valu_per_unit", "duration_ns_median", "clock_GHz_held", "frac", "frac_at_clock_held",
                                             "lanes_active_per_valu_instruction")} for f, v in out["forms"].items()}, indent=1))


if __name__ == "__main__":
    main()
