#!/usr/bin/env python
"""Where the time of a sweep goes when the dominant kernel is NOT running: from a rocprofv3 rocpd
database, the intervals without a launch of `--kernel` in flight, classified by what else runs
(other kernels by name, or nothing), plus the longest such intervals with their neighbours.

    python tools/timeline.py <results.db> [--kernel pk2_matvec] [--top 12]
"""
import argparse
import sqlite3
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("void ", "").replace("scint::", "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--kernel", default="pk2_matvec")
    ap.add_argument("--top", type=int, default=12)
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    t0 = min(r[1] for r in rows)
    ev = []
    for i, (n, s, e, *_r) in enumerate(rows):
        ev.append((s, 1, i))
        ev.append((e, -1, i))
    ev.sort()
    running = set()
    last = None
    by_state = defaultdict(int)
    gaps = []
    for t, d, i in ev:
        if last is not None and t > last:
            names = sorted({short(rows[j][0]) for j in running})
            if not any(a.kernel in n for n in names):
                key = "+".join(names) if names else "(idle)"
                by_state[key] += t - last
                gaps.append((t - last, last - t0, key))
        if d > 0:
            running.add(i)
        else:
            running.discard(i)
        last = t
    window = max(r[2] for r in rows) - t0
    tot = sum(by_state.values())
    print(f"window {window/1e6:.1f} ms; without {a.kernel}: {tot/1e6:.1f} ms ({100*tot/window:.1f} %)")
    for k, v in sorted(by_state.items(), key=lambda kv: -kv[1])[:a.top]:
        print(f"  {v/1e6:8.2f} ms  {k}")
    print("longest intervals without it (ms, at ms, state):")
    for g in sorted(gaps, reverse=True)[:a.top]:
        print(f"  {g[0]/1e6:7.3f}  @{g[1]/1e6:9.2f}  {g[2]}")


if __name__ == "__main__":
    main()
