#!/usr/bin/env python
"""Print selected fields of the last JSON line of a file (or of stdin with `-`):

    jpick.py FILE key.subkey ...      |      ... | jpick.py - key.subkey ...

The file form never touches stdin, so it cannot block a non-interactive shell."""
import json
import sys

if len(sys.argv) < 2:
    sys.exit(__doc__)
src = sys.argv[1]
text = sys.stdin.read() if src == "-" else open(src).read()
lines = [ln for ln in text.splitlines() if ln.startswith("{")]
if not lines:
    sys.exit(f"jpick: no JSON line in {src}")
d = json.loads(lines[-1])
out = []
for path in sys.argv[2:]:
    v = d
    for k in path.split("."):
        v = v[k]
    out.append(f"{path}={v:.5g}" if isinstance(v, float) else f"{path}={v}")
print("  ".join(out))
