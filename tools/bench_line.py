"""One-line summary of a bench.py JSON line file (used by tools/gpu_run.sh):  python tools/bench_line.py <file>."""
import json
import sys


def main(path):
    try:
        d = json.loads([l for l in open(path) if l.startswith("{")][-1])
    except Exception as exc:   # noqa: BLE001
        print(path, "no bench line:", exc)
        return
    c, r = d["config"], d["roofline"]
    extra = ""
    if "modeler" in d and d["modeler"]:
        extra += f"  modeler {d['modeler'].get('value', 0):.0f}"
    if "sspec" in d and d["sspec"]:
        extra += "  sspec " + " ".join(f"{k}:{v['ms']:.3f}ms" for k, v in d["sspec"].items())
    print(f"{path.split('/')[-1]}: {d['value']:.1f} eta/s  {d['ms_per_step']:.1f} ms/step  passes {c.get('lanczos_steps_mean', 0):.2f}  "
          f"failed {c.get('failed_etas')}  matvec {r['achieved']:.0f} GB/s (frac {r['frac']:.3f}, share {r.get('share_of_step_time', 0):.3f}){extra}")


if __name__ == "__main__":
    main(sys.argv[1])
