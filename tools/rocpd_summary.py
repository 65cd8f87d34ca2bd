#!/usr/bin/env python
"""Turn a rocprofv3 rocpd database (*_results.db) into the per-kernel summary we commit
under profiles/ (name, calls, total / average / min / max duration, registers, LDS)."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["kernel,calls,total_ms,avg_us,min_us,max_us,pct,vgpr,agpr,sgpr,lds_bytes,scratch_bytes"]
    for r in rows:
        lines.append(f"\"{r[0]}\",{r[1]},{r[2]/1e6:.3f},{r[3]/1e3:.2f},{r[4]/1e3:.2f},{r[5]/1e3:.2f},"
                     f"{100*r[2]/total:.2f},{r[6]},{r[7]},{r[8]},{r[9]},{r[10]}")
    text = "\n".join(lines) + "\n"
    if out:
        with open(out, "w") as fh:
            fh.write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
