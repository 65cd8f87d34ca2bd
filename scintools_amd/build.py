"""Build libscint_hip.so (gfx950) in-tree with hipcc.

    python -m scintools_amd.build [--force]

hipcc cross-compiles without a GPU.  thth.hip, arcnorm.hip and mosaic.hip are built with
-ffp-contract=off (bin decisions, np.interp arithmetic and the mosaic's sums must round like NumPy);
the other units keep the default contraction.  Objects land in scintools_amd/csrc/_obj, the shared
library next to the package so that it travels with a repository snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libscint_hip.so")
ARCH = "gfx950"

UNITS = {
    "capi.hip": [],
    "fft.hip": [],
    "sspec.hip": [],
    "thth.hip": ["-ffp-contract=off"],
    "eigen.hip": [],
    "eigen_packed.hip": [],
    "arcnorm.hip": ["-ffp-contract=off"],
    "mosaic.hip": ["-ffp-contract=off"],
}
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result"]
# Experiment builds (tools/build_variant.sh: -DSCINT_ROWS64=4 ...).  The flag string is recorded beside the objects and every unit
# is rebuilt when it changes -- a variable left exported can therefore never leave stale experiment constants in the product
# library -- and it is printed whenever it is not empty.
VARIANT_FLAGS = os.environ.get("SCINT_VARIANT_FLAGS", "").split()
STAMP = os.path.join(OBJ, "variant_flags.txt")


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libscint_hip.so")
    return exe


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    flags = " ".join(VARIANT_FLAGS)
    try:
        with open(STAMP) as fh:
            built_with = fh.read().strip()
    except OSError:
        built_with = ""
    if flags != built_with:
        force = True
    if flags:
        print(f"scintools_amd.build: SCINT_VARIANT_FLAGS = {flags!r} (an EXPERIMENT build, not the product's constants)", flush=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "scint_hip.h"))
    objs, cmds = [], []
    for src, extra in UNITS.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + headers + [__file__]):
            cmds.append([hipcc] + COMMON + VARIANT_FLAGS + extra + ["-c", s, "-o", o])
    if cmds:  # independent translation units: compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        with ThreadPoolExecutor(max_workers=min(len(cmds), os.cpu_count() or 1)) as pool:
            list(pool.map(run, cmds))
    with open(STAMP, "w") as fh:
        fh.write(flags + "\n")
    if force or _newer(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
