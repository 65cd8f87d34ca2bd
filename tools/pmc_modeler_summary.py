#!/usr/bin/env python
"""HBM traffic of ONE step of the modeler / chi^2 objective against its algorithmic bytes:

    python tools/pmc_modeler_summary.py <fetch.db> <write.db> <bench line of the FETCH pass> <out.json> "<command>"

The two databases are separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (with --kernel-trace only) of
`bench.py --objective chisq --steps 1 --warmup 0 --no-cpu-baseline`; corrections as in tools/pmc_summary.py (gfx950:
FETCH_SIZE x2 for wide streaming reads, WRITE_SIZE as reported).  The bench line's `modeler.roofline` carries the
algorithmic bytes of the same run (bench.py, modeler_objects).  bench.py quotes `traffic_over_algorithmic` of the newest
profiles/*_pmc_modeler_summary.json in `modeler.roofline.traffic`."""
import json
import sys

from pmc_summary import per_kernel


def main(fetch_db, write_db, bench_log, out_path, command):
    bench = json.loads([l for l in open(bench_log) if l.startswith('{"metric"')][-1])
    m = bench["modeler"]
    neta = round(m["value"] * m["ms_per_step"] / 1e3)
    alg = m["roofline"]["algorithmic_bytes_per_eta"] * neta * bench["steps"]
    f, w = per_kernel(fetch_db), per_kernel(write_db)
    out = {"command": command, "fetch_correction": 2.0, "write_correction": 1.0, "steps": bench["steps"], "curvatures_per_step": neta,
           "algorithmic_bytes": alg, "algorithmic_bytes_per_eta_by_part": m["roofline"]["algorithmic_bytes_per_eta_by_part"],
           "lanczos_steps_mean": m["lanczos_steps_mean"], "kernels": {},
           "csrc_sha256": (bench.get("library") or {}).get("csrc_sha256")}
    total = 0.0
    for k in sorted(set(f) | set(w)):
        nf, fb = f.get(k, (0, 0.0))
        nw, wb = w.get(k, (0, 0.0))
        if not k.startswith("scint::"):
            continue
        out["kernels"][k] = {"launches": nf or nw, "fetch_GB": round(2.0 * fb / 1e9, 4), "write_GB": round(wb / 1e9, 4)}
        total += 2.0 * fb + wb
    out["hbm_bytes"] = total
    out["traffic_over_algorithmic"] = total / alg
    with open(out_path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: out[k] for k in ("algorithmic_bytes", "hbm_bytes", "traffic_over_algorithmic")}))
    for k, v in sorted(out["kernels"].items(), key=lambda kv: -(kv[1]["fetch_GB"] + kv[1]["write_GB"]))[:8]:
        print(f"{v['launches']:6d}  fetch {v['fetch_GB']:9.2f} GB  write {v['write_GB']:9.2f} GB  {k[:100]}")


if __name__ == "__main__":
    main(*sys.argv[1:6])
