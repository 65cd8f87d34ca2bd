"""TEST INFRASTRUCTURE ONLY: run bench.py's main() against the host interpreter (tests/emu) at a tiny size -- a check of
bench.py's own Python paths (argument handling, the legs of the JSON line, the accounting) where no GPU exists, not a
measurement: every rate it prints is the interpreter's.  Usage: python tests/emu/bench_probe.py <bench.py arguments>."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)


def main():
    import torch
    from _pytest.monkeypatch import MonkeyPatch
    import emulated
    patch = MonkeyPatch()
    emulated.install(patch)
    patch.setattr(torch.cuda, "set_device", lambda d: None)
    patch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    patch.setattr(torch.cuda, "device_count", lambda: 1)
    real_empty = torch.empty

    def empty(*a, **k):                      # bench.py's few device="cuda" buffers live in host memory here
        if str(k.get("device", "")).startswith("cuda"):
            k["device"] = "cpu"
        return real_empty(*a, **k)
    patch.setattr(torch, "empty", empty)
    sys.argv = [os.path.join(REPO, "bench.py")] + sys.argv[1:]
    try:
        runpy.run_path(sys.argv[0], run_name="__main__")
    finally:
        patch.undo()


if __name__ == "__main__":
    main()
