#!/usr/bin/env python
"""Table behind ten_log10 of scintools_amd/csrc/sspec.hip: for i = 0 .. 127, c = 1 + (i + 0.5) / 128,
(cinv, klc) with cinv = 1/c rounded to double and klc = -(10 / ln 10) ln(cinv) for THAT rounded cinv, computed with
60 decimal digits and rounded once -- so that 10 log10(m) = klc + (10 / ln 10) log1p(m cinv - 1) holds to the last bit
of the table whatever the rounding of cinv was.  Prints the C initialiser (hex floats)."""
from decimal import Decimal, getcontext

getcontext().prec = 60
K = Decimal(10) / Decimal(10).ln()
rows = []
for i in range(128):
    c = 1.0 + (i + 0.5) / 128.0            # exact in binary
    cinv = 1.0 / c                          # one rounding; the table absorbs it
    klc = float(-K * Decimal(cinv).ln())
    rows.append(f"    {{{cinv.hex()}, {klc.hex()}}},")
print("\n".join(rows))
