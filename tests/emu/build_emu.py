"""TEST INFRASTRUCTURE ONLY: build the kernel sources for the x86 host on top of the workgroup
interpreter in this directory (tests/emu/_build/libscint_emu_test.so).

    python tests/emu/build_emu.py [--force]

The product library (scintools_amd/libscint_hip.so) is built by scintools_amd/build.py with hipcc
for gfx950 and is the only thing scintools_amd ever loads; this host build is loaded by
tests/test_emu_cpu.py alone, through its own ctypes handle, to exercise the kernels' control flow
and arithmetic where no GPU exists.  The sources are compiled unchanged except for two textual
substitutions: `extern __shared__` (dynamic LDS) becomes a plain `extern` of an array the
interpreter owns, and the one line of inline assembly (the LDS-only barrier of common.hpp) becomes
`__syncthreads()`.
"""
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(REPO, "scintools_amd", "csrc")
# SCINT_EMU_SANITIZE=address: an AddressSanitizer build (global and LDS arrays get red zones, so an
# out-of-bounds access of a kernel is reported instead of silently reading a neighbour).  Run as
#   LD_PRELOAD=$(python tests/emu/build_emu.py --asan-runtime) ASAN_OPTIONS=detect_leaks=0 \
#   SCINT_EMU_SANITIZE=address python -m pytest tests/test_emu_cpu.py
SANITIZE = os.environ.get("SCINT_EMU_SANITIZE", "")
# SCINT_EMU_DEFINES="-DSCINT_ROWS32=8 ...": extra definitions for the kernel sources (the build constants of csrc/packed.hpp), in a build
# directory of their own -- to run the interpreter tests on a variant before it is sent to a GPU (tools/build_variant.sh takes the same -D)
DEFINES = os.environ.get("SCINT_EMU_DEFINES", "").split()
OUT = os.path.join(HERE, "_build" + ("_" + SANITIZE if SANITIZE else "") +
                   ("_" + "".join(c if c.isalnum() else "_" for c in "".join(DEFINES)) if DEFINES else ""))
LIB = os.path.join(OUT, "libscint_emu_test.so")

# same per-unit floating-point contraction as the product build (scintools_amd/build.py)
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


def _clang():
    for exe in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++") or ""):
        if exe and os.path.exists(exe):
            return exe
    raise RuntimeError("clang++ not found (the kernel sources use clang vector / address-space extensions)")


def _stage_sources():
    """Copy csrc into the build directory with the dynamic-LDS declarations rewritten."""
    src_out = os.path.join(OUT, "scintools_amd", "csrc")
    os.makedirs(src_out, exist_ok=True)
    changed = False
    for name in sorted(os.listdir(CSRC)):
        if not name.endswith((".hip", ".hpp")):
            continue
        with open(os.path.join(CSRC, name)) as fh:
            text = fh.read()
        text = re.sub(r"\bextern\s+__shared__", "extern", text)
        # the LDS-only barrier (inline assembly, common.hpp) is a plain barrier on the interpreter
        text = re.sub(r'asm volatile\("s_waitcnt lgkmcnt\(0\)\\n\\ts_barrier" ::: "memory"\);', "__syncthreads();", text)
        # optimisation fences on a register (no instruction): nothing to interpret
        text = re.sub(r'asm volatile\("" : "\+v"\((\w+)\)\);', "", text)
        # the in-wave LDS hand-off: every lane of the wave must have stored before any lane loads
        text = re.sub(r'asm volatile\("s_waitcnt lgkmcnt\(0\)" ::: "memory"\);', "(void)__shfl(0, 0, 64);", text)
        dst = os.path.join(src_out, name.replace(".hip", ".hip.cpp"))
        old = None
        if os.path.exists(dst):
            with open(dst) as fh:
                old = fh.read()
        if old != text:
            with open(dst, "w") as fh:
                fh.write(text)
            changed = True
    inc_out = os.path.join(OUT, "include")
    os.makedirs(inc_out, exist_ok=True)
    shutil.copy2(os.path.join(REPO, "include", "scint_hip.h"), os.path.join(inc_out, "scint_hip.h"))
    return src_out, changed


def build(force=False, verbose=False):
    cxx = _clang()
    os.makedirs(OUT, exist_ok=True)
    src_out, _ = _stage_sources()
    common = [cxx, "-x", "c++", "-std=c++17", "-O1", "-g", "-fPIC", "-fno-omit-frame-pointer",
              "-I", os.path.join(HERE, "include"), "-Wno-unused-result", "-Wno-unknown-attributes",
              "-Wno-ignored-attributes", "-Wno-unknown-pragmas", "-Wno-pass-failed"] + DEFINES
    link_extra = []
    if SANITIZE:
        common += [f"-fsanitize={SANITIZE}", "-shared-libsan"]
        link_extra = [f"-fsanitize={SANITIZE}", "-shared-libsan"]
    deps = [os.path.join(HERE, "include", "hip", "hip_runtime.h"), __file__]
    deps += [os.path.join(src_out, f) for f in os.listdir(src_out) if f.endswith(".hpp")]
    jobs, objs = [], []
    units = [(os.path.join(src_out, u.replace(".hip", ".hip.cpp")), extra) for u, extra in UNITS.items()]
    units.append((os.path.join(HERE, "emu_runtime.cpp"), []))
    units.append((os.path.join(HERE, "emu_selftest.cpp"), []))
    for src, extra in units:
        obj = os.path.join(OUT, os.path.basename(src) + ".o")
        objs.append(obj)
        stale = force or not os.path.exists(obj) or any(
            os.path.getmtime(d) > os.path.getmtime(obj) for d in [src] + deps)
        if stale:
            jobs.append(common + extra + ["-c", src, "-o", obj])
    if jobs:
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
            list(pool.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        cmd = [cxx, "-shared", "-fPIC", "-o", LIB] + link_extra + objs + ["-lm", "-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


def asan_runtime():
    out = subprocess.run([_clang(), "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    return out.stdout.strip()


if __name__ == "__main__":
    if "--asan-runtime" in sys.argv:
        print(asan_runtime())
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
