#!/usr/bin/env python
"""What bounds calc_sspec's three kernels: HBM bytes (rocprofv3 PMC, FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate
passes) and vector-instruction issue (SQ_INSTS_VALU, SQ_WAVES of a third pass) of `python tools/time_fft.py <size> sspec`:

    python tools/sspec_roofline.py <fetch.db> <write.db> <counters.db> <size> <out.json>

(tools/gpu_run.sh sspecroof collects the three passes.)  Per kernel: launches, bytes fetched / written per launch, wavefronts
and instructions per wavefront, and the two floors a launch cannot beat on an MI355X:
  hbm_floor_us     its measured HBM bytes at the 6.3 TB/s the guide calls achievable (and at the 8 TB/s peak),
  issue_floor_us   VALU instructions x wavefronts x 4 cycles / (1024 SIMDs x 2.4 GHz): every vector instruction of a 64-wide
                   wavefront occupies its 16-lane SIMD for four cycles, whatever its type.
bench.py's `sspec.<size>.roofline` combines them with the kernel times it measures live (hipEvent brackets inside
scint_sspec) -- as long as the kernel sources still hash to `csrc_sha256`."""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import per_kernel  # noqa: E402

SIMDS, CLOCK_HZ, ACHIEVABLE, PEAK = 1024, 2.4e9, 6.3e12, 8.0e12


def counters(db):
    c = sqlite3.connect(db)
    acc = {}
    for k, cn, _, v, d in c.execute("select kernel_name, counter_name, dispatch_id, sum(value), max(duration) from counters_collection "
                                    "group by kernel_name, counter_name, dispatch_id"):
        k = k.split("(")[0].replace("void ", "")
        a = acc.setdefault(k, {}).setdefault(cn, [0, 0.0, 0.0])
        a[0] += 1; a[1] += v; a[2] += d
    return {k: {cn: (n, v / n, d / n / 1e3) for cn, (n, v, d) in cs.items()} for k, cs in acc.items()}


def main(fetch_db, write_db, ctr_db, size, out_path):
    from bench import library_fingerprint
    f, w, c = per_kernel(fetch_db), per_kernel(write_db), counters(ctr_db)
    out = {"command": f"rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR "
                      f"--kernel-trace -- python tools/time_fft.py {size} sspec (three passes)",
           "size": int(size), "fetch_correction": 2.0, "write_correction": 1.0, "csrc_sha256": library_fingerprint()["csrc_sha256"],
           "constants": {"simds": SIMDS, "clock_hz": CLOCK_HZ, "hbm_achievable_Bps": ACHIEVABLE, "hbm_peak_Bps": PEAK}, "kernels": {}}
    for k in sorted(set(f) | set(w)):
        if "sspec_" not in k:
            continue
        nf, fb = f.get(k, (0, 0.0))
        nw, wb = w.get(k, (0, 0.0))
        ck = c.get(k, {})
        waves = ck.get("SQ_WAVES", (0, 0.0, 0.0))[1]
        valu = ck.get("SQ_INSTS_VALU", (0, 0.0, 0.0))[1]
        hbm = 2.0 * fb / max(nf, 1) + wb / max(nw, 1)
        name = "prep" if "prep_kernel" in k else ("prep_means" if "prep_means" in k else ("cols" if "cols" in k else "rows"))
        out["kernels"][name] = {
            "kernel": k, "launches": nf or nw, "fetch_bytes_per_launch": 2.0 * fb / max(nf, 1), "write_bytes_per_launch": wb / max(nw, 1),
            "hbm_bytes_per_launch": hbm, "hbm_floor_us": 1e6 * hbm / ACHIEVABLE, "hbm_floor_at_peak_us": 1e6 * hbm / PEAK,
            "wavefronts": waves, "valu_insts_per_wavefront": valu / max(waves, 1),
            "salu_insts_per_wavefront": ck.get("SQ_INSTS_SALU", (0, 0.0, 0.0))[1] / max(waves, 1),
            "lds_insts_per_wavefront": ck.get("SQ_INSTS_LDS", (0, 0.0, 0.0))[1] / max(waves, 1),
            "vmem_rd_insts_per_wavefront": ck.get("SQ_INSTS_VMEM_RD", (0, 0.0, 0.0))[1] / max(waves, 1),
            "vmem_wr_insts_per_wavefront": ck.get("SQ_INSTS_VMEM_WR", (0, 0.0, 0.0))[1] / max(waves, 1),
            "issue_floor_us": 1e6 * valu * 4.0 / (SIMDS * CLOCK_HZ),
            "duration_under_counters_us": ck.get("SQ_WAVES", (0, 0.0, 0.0))[2]}
    n = int(size)
    alg = 8.0 * n * n + 8.0 * n * (2 * n)
    tot = sum(v["hbm_bytes_per_launch"] for v in out["kernels"].values())
    out["algorithmic_bytes"] = alg
    out["hbm_bytes"] = tot
    out["traffic_over_algorithmic"] = tot / alg
    with open(out_path, "w") as fh:
        json.dump(out, fh, indent=1)
    for k, v in out["kernels"].items():
        print(f"{k:10s} hbm {v['hbm_bytes_per_launch'] / 1e6:8.1f} MB  floor {v['hbm_floor_us']:6.1f} us | {v['valu_insts_per_wavefront']:7.0f} VALU x "
              f"{v['wavefronts']:8.0f} waves  floor {v['issue_floor_us']:6.1f} us | {v['duration_under_counters_us']:6.1f} us under the counters")
    print("traffic / algorithmic", round(tot / alg, 3))


if __name__ == "__main__":
    main(*sys.argv[1:6])
