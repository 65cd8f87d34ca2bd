#!/usr/bin/env python
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected separately, with
--kernel-trace only) of `bench.py` into profiles/<tag>_pmc_summary.json.

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly half of the
bytes of a wide coalesced streaming read -> doubled here.  Calibration inside the same run: the
in-place FFT column pass reads 4096*4096*16 B = 262144 KiB and FETCH_SIZE shows ~131.2-131.8 MiB
(x2 = 262.4-263.6 MiB); its WRITE_SIZE is 262144.0 KiB, i.e. WRITE_SIZE needs no correction.
"""
import json
import sqlite3
import sys


def per_kernel(db):
    c = sqlite3.connect(db)
    out = {}
    for name, n, tot in c.execute("select kernel_name, count(*), sum(value) from counters_collection "
                                  "group by kernel_name"):
        out[name.split("(")[0].replace("void ", "")] = (n, tot * 1024.0)
    return out


def main(fetch_db, write_db, fetch_log, out_path, command=None):
    line = [l for l in open(fetch_log) if l.startswith('{"metric"')][-1]
    bench = json.loads(line)
    r = bench["roofline"]
    alg_per_launch = r["algorithmic_bytes_per_step"] * bench["steps"] / r["launches"]
    f, w = per_kernel(fetch_db), per_kernel(write_db)
    summary = {"command": command or "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 1 "
                                     "--warmup 0 --no-cpu-baseline --modeler-steps 0",
               "fetch_correction": 2.0, "write_correction": 1.0, "kernels": {},
               # the kernel sources of the profiled library (bench.py: library_fingerprint); bench.py quotes this summary's ratio
               # only while the sources it runs on hash to the same value
               "csrc_sha256": (bench.get("library") or {}).get("csrc_sha256")}
    for k in sorted(set(f) | set(w)):
        nf, fb = f.get(k, (0, 0.0))
        nw, wb = w.get(k, (0, 0.0))
        summary["kernels"][k] = {"launches": nf or nw,
                                 "fetch_bytes_per_launch": 2.0 * fb / max(nf, 1),
                                 "write_bytes_per_launch": wb / max(nw, 1)}
    mv_name = max((k for k in summary["kernels"] if "matvec_kernel" in k and k.startswith("scint::pk")),
                  key=lambda k: summary["kernels"][k]["fetch_bytes_per_launch"] * summary["kernels"][k]["launches"])
    summary["dominant_kernel"] = mv_name
    mv = summary["kernels"][mv_name]
    mv["algorithmic_bytes_per_launch"] = alg_per_launch
    mv["hbm_bytes_per_launch"] = mv["fetch_bytes_per_launch"] + mv["write_bytes_per_launch"]
    mv["traffic_over_algorithmic"] = mv["hbm_bytes_per_launch"] / alg_per_launch
    ga = summary["kernels"].get("scint::thth_gather_packed_kernel")
    if ga and "gather" in bench:
        n_min, n_max = bench["config"]["N_min"], bench["config"]["N_max"]
        ga["hbm_bytes_per_launch"] = ga["fetch_bytes_per_launch"] + ga["write_bytes_per_launch"]
        ga["note"] = (f"reads come mostly from L2 / Infinity Cache (the touched CS region is a fraction of the plane); "
                      f"algorithmic 16 N^2 per eta with N = {n_min}..{n_max}")
    with open(out_path, "w") as fh:
        json.dump(summary, fh, indent=1)
    print(json.dumps(mv, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:6])
