#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected
separately with --kernel-trace only) of ANY command:

    python tools/pmc_kernels.py <fetch.db> <write.db> <out.json> "<command that was profiled>"

Same corrections as tools/pmc_summary.py (gfx950: FETCH_SIZE x2 for wide streaming reads,
WRITE_SIZE as reported); no bench line needed.
"""
import json
import sys

from pmc_summary import per_kernel


def main(fetch_db, write_db, out_path, command):
    f, w = per_kernel(fetch_db), per_kernel(write_db)
    out = {"command": command, "fetch_correction": 2.0, "write_correction": 1.0, "kernels": {}}
    for k in sorted(set(f) | set(w)):
        nf, fb = f.get(k, (0, 0.0))
        nw, wb = w.get(k, (0, 0.0))
        out["kernels"][k] = {"launches": nf or nw, "fetch_MB_per_launch": round(2.0 * fb / max(nf, 1) / 1e6, 3),
                             "write_MB_per_launch": round(wb / max(nw, 1) / 1e6, 3)}
    json.dump(out, open(out_path, "w"), indent=1)
    for k, v in out["kernels"].items():
        if k.startswith("scint::"):
            print(f"{v['launches']:5d}  fetch {v['fetch_MB_per_launch']:9.1f} MB  write {v['write_MB_per_launch']:9.1f} MB  {k[:110]}")


if __name__ == "__main__":
    main(*sys.argv[1:5])
