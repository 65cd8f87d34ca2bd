"""Build libscint_hip.so (gfx950) in-tree with hipcc.

    python -m scintools_amd.build [--force]

hipcc cross-compiles without a GPU.  thth.hip and arcnorm.hip are built with
-ffp-contract=off (bin decisions and np.interp arithmetic must round like NumPy);
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
}
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result"]
COMMON += os.environ.get("SCINT_VARIANT_FLAGS", "").split()    # tools/build_variant.sh only (experiment builds, forced)


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
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "scint_hip.h"))
    objs, cmds = [], []
    for src, extra in UNITS.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + headers + [__file__]):
            cmds.append([hipcc] + COMMON + extra + ["-c", s, "-o", o])
    if cmds:  # independent translation units: compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        with ThreadPoolExecutor(max_workers=min(len(cmds), os.cpu_count() or 1)) as pool:
            list(pool.map(run, cmds))
    if force or _newer(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
