#!/usr/bin/env python
"""Turn a rocprofv3 rocpd database (*_results.db) into the per-kernel summary we commit
under profiles/:

    python tools/rocpd_summary.py <results.db> [out.csv] [out.json]

CSV: name, calls, total / average / min / max duration, registers, LDS -- what
`rocprofv3 --stats` prints -- plus two columns a multi-stream program needs:

  union_ms       length of the UNION of the kernel's launch intervals.  The eta sweep may drive
                 two streams, so launches of one kernel can overlap in time; the plain sum of
                 their durations then exceeds the wall time, and bytes / union is the
                 kernel's throughput (bytes / sum is a per-launch rate under sharing).
  self_overlap   fraction of that union during which >= 2 launches of the SAME kernel ran.

JSON (optional): per kernel the co-residency histogram -- time spent with 1, 2, ... launches of
the kernel itself running, and time shared with every other kernel -- and the trace window.
"""
import json
import sqlite3
import sys
from collections import defaultdict


def union_and_depth(spans):
    """spans: list of (start, end).  Returns (union length, {depth: time at that depth})."""
    events = []
    for s, e in spans:
        events.append((s, 1))
        events.append((e, -1))
    events.sort()
    depth, last, hist = 0, None, defaultdict(int)
    for t, d in events:
        if depth > 0 and last is not None:
            hist[depth] += t - last
        depth += d
        last = t
    return sum(hist.values()), dict(hist)


def overlap_with(spans_a, spans_b):
    """Time during which at least one launch of a AND at least one launch of b are running."""
    ev = [(s, 0, 1) for s, _ in spans_a] + [(e, 0, -1) for _, e in spans_a] + \
         [(s, 1, 1) for s, _ in spans_b] + [(e, 1, -1) for _, e in spans_b]
    ev.sort()
    depth, last, tot = [0, 0], None, 0
    for t, which, d in ev:
        if depth[0] > 0 and depth[1] > 0 and last is not None:
            tot += t - last
        depth[which] += d
        last = t
    return tot


def short(name):
    return name.split("(")[0].replace("void ", "")


def main(db, out_csv=None, out_json=None):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    spans = defaultdict(list)
    for name, s, e in c.execute("select name, start, end from kernels"):
        spans[name].append((s, e))
    total = sum(r[2] for r in rows) or 1
    lines = ["kernel,calls,total_ms,avg_us,min_us,max_us,pct,vgpr,agpr,sgpr,lds_bytes,scratch_bytes,"
             "union_ms,self_overlap"]
    detail = {}
    everything = [sp for v in spans.values() for sp in v]
    window = (min(s for s, _ in everything), max(e for _, e in everything)) if everything else (0, 0)
    all_union, _ = union_and_depth(everything)
    for r in rows:
        u, hist = union_and_depth(spans[r[0]])
        multi = sum(t for d, t in hist.items() if d >= 2)
        lines.append(f"\"{r[0]}\",{r[1]},{r[2]/1e6:.3f},{r[3]/1e3:.2f},{r[4]/1e3:.2f},{r[5]/1e3:.2f},"
                     f"{100*r[2]/total:.2f},{r[6]},{r[7]},{r[8]},{r[9]},{r[10]},{u/1e6:.3f},"
                     f"{(multi / u if u else 0):.3f}")
        detail[short(r[0])] = {"calls": r[1], "sum_ms": r[2] / 1e6, "union_ms": u / 1e6,
                               "ms_at_depth": {str(d): t / 1e6 for d, t in sorted(hist.items())}}
    if out_json:
        top = [r[0] for r in rows[:6]]
        for a in top:
            detail[short(a)]["ms_shared_with"] = {short(b): overlap_with(spans[a], spans[b]) / 1e6
                                                  for b in top if b != a}
        with open(out_json, "w") as fh:
            json.dump({"trace_window_ms": (window[1] - window[0]) / 1e6, "any_kernel_busy_ms": all_union / 1e6,
                       "kernels": detail}, fh, indent=1)
    text = "\n".join(lines) + "\n"
    if out_csv:
        with open(out_csv, "w") as fh:
            fh.write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, sys.argv[3] if len(sys.argv) > 3 else None)
