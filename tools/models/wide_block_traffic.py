"""Model of the HBM bytes one matrix pass of the block Lanczos sweep moves beside the matrix itself.

    python tools/models/wide_block_traffic.py [N]

Per pass and curvature: the packed upper-triangle tiles (64 KiB each) once, plus what the recurrence
needs around them -- column partials (written by the mat-vec, read by the reduce), row partials, the
X_J / X_I blocks a workgroup loads (or rebuilds from W_{j-1} and Q_{j-1}: two reads), and the vector
passes of the reduce / qbuild kernels.  Nothing is assumed to hit a cache (the measured 1.10 of the
two-vector kernel, profiles/r02_pmc_summary.json, agrees with its 1.13 here).  The last column is the
cost of a sweep relative to today's default: passes(W) x (1 + extra).
"""
import sys

PASSES = {1: 40.0, 2: 31.9, 4: 24.6, 8: 19.9}      # tools/models/block_lanczos_passes.py at N = 4095


def model(nb, W, strip, band=1, col_parts=1, rebuild=False, q_family=False):
    tile = 64 * 64 * 16
    ntiles = nb * (nb + 1) // 2
    vec = 64 * W * 16                               # one 64-row block of a W-column vector
    wgs = rows = colp = xcopies = 0
    for I0 in range(0, nb, band):
        nrow = min(band, nb - I0)
        for J0 in range(I0, nb, strip):
            J1 = min(nb, J0 + strip)
            wgs += 1
            rows += nrow                            # row partials of this workgroup
            xcopies += (J1 - J0) + nrow             # X_J blocks + X_I blocks
            colp += sum(1 for J in range(J0, J1) if J > I0)
    extra = 2 * colp * vec * col_parts + 2 * rows * vec + xcopies * vec * (2 if rebuild else 1)
    extra += (6 if q_family else 4) * nb * vec      # reduce (+ qbuild): reads and writes of the N x W vectors
    return ntiles * tile, extra, wgs


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4095
    nb = -(-n // 64)
    rows = [
        ("W=2 default (strips of 16, rebuild)", 2, dict(strip=16, rebuild=True)),
        ("W=2 wide-block family, bands of 4 x chunks of 16", 2, dict(strip=16, band=4, q_family=True)),
        ("W=4 vector FMAs, quarter strips of 8", 4, dict(strip=8, col_parts=4, rebuild=True)),
        ("W=4 matrix cores, strips of 8 (rebuild)", 4, dict(strip=8, rebuild=True)),
        ("W=4 wide-block family, strips of 8", 4, dict(strip=8, q_family=True)),
        ("W=4 wide-block family, bands of 4 x chunks of 8", 4, dict(strip=8, band=4, q_family=True)),
        ("W=8 strips of 4", 8, dict(strip=4, q_family=True)),
        ("W=8 strips of 8 (one workgroup per CU)", 8, dict(strip=8, q_family=True)),
        ("W=8 bands of 4 x chunks of 4", 8, dict(strip=4, band=4, q_family=True)),
    ]
    base = None
    print(f"N = {n}: {nb} block rows, {nb * (nb + 1) // 2} tiles, {nb * (nb + 1) // 2 * 65536 / 1e6:.0f} MB of matrix per pass")
    for name, W, kw in rows:
        mat, extra, wgs = model(nb, W, **kw)
        cost = PASSES[W] * (1 + extra / mat)
        base = base or cost
        print(f"{name:50s} extra {extra / 1e6:6.1f} MB = {extra / mat:5.2f} x   workgroups {wgs:5d}   "
              f"passes {PASSES[W]:5.1f}   sweep cost vs default {cost / base:5.2f}")


if __name__ == "__main__":
    main()
