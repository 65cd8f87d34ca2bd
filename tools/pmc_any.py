#!/usr/bin/env python
"""Per-kernel averages of whatever counters a `rocprofv3 --pmc ... --kernel-trace` pass collected:

    python tools/pmc_any.py <results.db> [substring of the kernel name ...]  > summary.txt

One line per (kernel, counter): launches, mean value per launch (summed over the counter's instances),
mean duration.  SQ_* cycle counters (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*) count quad-cycles
(MI355X_MICROARCH.md).
"""
import sqlite3
import sys


def short(name):
    return name.split("(")[0].replace("void ", "")


def main(db, subs):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, dispatch_id, sum(value), max(duration) from counters_collection "
                     "group by kernel_name, counter_name, dispatch_id")
    acc = {}
    for k, cn, _, v, d in rows:
        k = short(k)
        if subs and not any(s in k for s in subs):
            continue
        a = acc.setdefault((k, cn), [0, 0.0, 0.0])
        a[0] += 1; a[1] += v; a[2] += d
    for (k, cn), (n, v, d) in sorted(acc.items()):
        print(f"{k[:90]:90s} {cn:28s} n={n:5d}  mean={v / n:16.1f}  dur_us={d / n / 1e3:10.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
