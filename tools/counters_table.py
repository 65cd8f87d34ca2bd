#!/usr/bin/env python
"""One line per kernel from the raw output of `tools/gpu_run.sh counters|mixedev` (tools/pmc_any.py lines of two
--pmc passes): the derived figures of profiles/r03_sweep_counters.txt.

    python tools/counters_table.py gpurun_out/<tag>_<name>_counters_raw.txt [substring ...]

us = mean duration; per wave: vector instructions, vector-memory reads, LDS instructions; VALU busy = 4 x SQ_ACTIVE_INST_VALU /
(1024 SIMDs x GRBM_GUI_ACTIVE / 8); lanes = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU; occ = waves per SIMD =
4 x SQ_WAVE_CYCLES / (1024 x GRBM_GUI_ACTIVE / 8); wait = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (issue stalls);
parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES (s_waitcnt / barrier).  SQ_* cycle counters are quad-cycles (MI355X_MICROARCH.md)."""
import re
import sys


def main(path, subs):
    rows = {}
    for line in open(path):
        m = re.match(r"(\S.*?)\s+(SQ_\w+|GRBM_\w+)\s+n=\s*(\d+)\s+mean=\s*([\d.]+)\s+dur_us=\s*([\d.]+)", line)
        if not m:
            continue
        k = m.group(1).replace("scint::", "")
        if subs and not any(s in k for s in subs):
            continue
        r = rows.setdefault(k, {})
        r[m.group(2)] = float(m.group(4))
        r["n"], r["us"] = int(m.group(3)), float(m.group(5))
    print(f"{'kernel':52s} {'n':>5s} {'us':>8s} {'waves':>8s} {'VALU/w':>7s} {'VMEMrd/w':>8s} {'LDS/w':>7s} {'VALUbusy':>8s} {'lanes':>6s} {'occ':>5s} {'wait':>5s} {'parked':>6s}")
    for k, r in sorted(rows.items()):
        w = r.get("SQ_WAVES", 0) or 1
        simd_cycles = 1024 * r.get("GRBM_GUI_ACTIVE", 0) / 8 or 1
        act = r.get("SQ_ACTIVE_INST_VALU", 0)
        wc = r.get("SQ_WAVE_CYCLES", 0) or 1
        print(f"{k[:52]:52s} {r['n']:5d} {r['us']:8.1f} {w:8.0f} {r.get('SQ_INSTS_VALU', 0) / w:7.0f} {r.get('SQ_INSTS_VMEM_RD', 0) / w:8.1f} "
              f"{r.get('SQ_INSTS_LDS', 0) / w:7.1f} {4 * act / simd_cycles:8.2f} {r.get('SQ_THREAD_CYCLES_VALU', 0) / (act or 1):6.1f} "
              f"{4 * wc / simd_cycles:5.2f} {r.get('SQ_WAIT_INST_ANY', 0) / wc:5.2f} {r.get('SQ_WAIT_ANY', 0) / wc:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
