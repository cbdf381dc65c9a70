#!/usr/bin/env python3
"""What a vector instruction of a kernel costs to ISSUE, on average: the static mix of the kernel's VALU instructions (from
the disassembly of the built library) weighted by the measured issue cost of each opcode on gfx950
(profiles/r04_final/valu_rates_gfx950.log, from tools/valu_rates.hip: most opcodes take four cycles per wave64 instruction, a
handful of plain two-operand ones -- add, sub, and / or / xor, right shifts, moves, f32 add / mul -- about 2.5).

    python tools/valu_mix.py rollout_kernel carve_kernel ...        (needs the built library, no GPU)

Prints one JSON object per kernel-name substring: static VALU count, cycles per instruction of the mix, the counts by class.
The figure prices the `valu-issue` rooflines of bench.py (profiles/valu_issue.json, written by tools/update_valu_issue.py).
MFMA instructions are not vector-ALU issue in this sense and are left out."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RATES = os.path.join(ROOT, "profiles", "r04_final", "valu_rates_gfx950.log")
FULL = 1.58          # relative to v_add_u32: what nearly every opcode measures (1.5-1.65)


def measured_rates():
    """opcode -> cost relative to v_add_u32, and the cycles v_add_u32 itself takes (from the log's first line)."""
    rel, base_cycles = {}, None
    for line in open(RATES):
        m = re.match(r"v_add_u32: ([\d.]+) ms for (\d+) x (\d+) instructions on eight waves per SIMD = ([\d.]+) GHz if", line)
        if m:
            # "x GHz if one wave64 instruction takes four cycles" at the 2.27 GHz the chip ran -> cycles per instruction
            base_cycles = 4.0 * 2.27 / float(m.group(4))
            continue
        m = re.match(r"(v_\w+)\s.*=\s+([\d.]+) x v_add_u32", line)
        if m and "(" not in line.split("=")[0].replace(m.group(1), ""):
            rel.setdefault(m.group(1), float(m.group(2)))
    return rel, base_cycles


def disassemble(pattern, lib):
    out = subprocess.run([os.path.join(ROOT, "tools", "dump_isa.sh"), pattern, lib], capture_output=True, text=True, cwd=ROOT).stdout
    return [l.split()[0] for l in out.splitlines() if re.match(r"^\s+[vs]_|^\s+(ds|global|buffer|flat)_", l)]


def cost_of(op, rel):
    base = re.sub(r"_(e32|e64|dpp|e64_dpp)$", "", op)
    if base.endswith("_sdwa"):
        return rel.get(base, rel.get("v_add_u32_sdwa", FULL))
    if base in rel:
        return rel[base]
    if base in ("v_subrev_u32", "v_subrev_co_u32"):
        return rel.get("v_sub_u32", 1.0)
    if base in ("v_not_b32", "v_sub_f32", "v_subrev_f32", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_accvgpr_mov_b32"):
        return rel.get("v_mov_b32", 0.85)
    return FULL


def mix(pattern, lib):
    rel, base_cycles = measured_rates()
    ops = disassemble(pattern, lib)
    valu = [o for o in ops if o.startswith("v_") and not o.startswith("v_mfma") and not o.startswith("v_smfmac")]
    if not valu:
        return {"kernel": pattern, "error": "no such kernel in the library"}
    by = collections.Counter(re.sub(r"_(e32|e64)$", "", o) for o in valu)
    cost = sum(cost_of(o, rel) for o in valu) / len(valu)
    cheap = sum(1 for o in valu if cost_of(o, rel) <= 1.05)
    return {"kernel": pattern, "static_valu_instructions": len(valu), "static_mfma_instructions": sum(o.startswith("v_mfma") for o in ops),
            "static_salu_instructions": sum(o.startswith("s_") for o in ops),
            "relative_cost_of_the_mix": cost, "v_add_u32_cycles": base_cycles, "cycles_per_valu_instruction": cost * base_cycles,
            "instructions_at_about_2_5_cycles": cheap, "instructions_at_about_4_cycles": len(valu) - cheap,
            "most_frequent": by.most_common(12), "rates_from": os.path.relpath(RATES, ROOT)}


if __name__ == "__main__":
    lib = os.environ.get("TPL_LIB", "lib/libtetris_piclim.so")
    for pat in sys.argv[1:] or ["rollout_kernel", "carve_kernel"]:
        print(json.dumps(mix(pat, lib)))
