#!/usr/bin/env python
"""Print selected fields of the last JSON line on stdin: jpick.py key.subkey ..."""
import json
import sys
line = [l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
out = []
for path in sys.argv[1:]:
    v = d
    for k in path.split("."):
        v = v[k]
    out.append(f"{path}={v:.5g}" if isinstance(v, float) else f"{path}={v}")
print("  ".join(out))
