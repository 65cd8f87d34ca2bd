"""Static instruction statistics of the gfx950 kernels of one translation unit.

    python tools/isa_stats.py scintools_amd/csrc/fft.hip [substring ...]

Compiles the unit to assembly (device only, the flags of scintools_amd/build.py) and prints, per
kernel whose demangled name contains every given substring: instruction count, branches, global
loads/stores, LDS ops, 64-bit integer ops and fp64 divisions' helpers.  Used to compare variants of
a kernel without a GPU (an issue-bound kernel's time follows its instruction count).
"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scintools_amd import build as b


def main():
    src = sys.argv[1]
    subs = sys.argv[2:]
    out = "/tmp/isa/" + os.path.basename(src).replace(".hip", ".s")
    os.makedirs("/tmp/isa", exist_ok=True)
    extra = b.UNITS.get(os.path.basename(src), [])
    if not os.path.exists(out) or os.path.getmtime(out) < max(
            os.path.getmtime(os.path.join(b.CSRC, f)) for f in os.listdir(b.CSRC) if f.endswith((".hip", ".hpp"))):
        subprocess.run([b._hipcc()] + b.COMMON + extra + ["-S", "--cuda-device-only", src, "-o", out], check=True)
    names, cur, stats = {}, None, {}
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            stats[cur] = dict(instr=0, branch=0, gload=0, gstore=0, lds=0, i64=0, mul32=0, div=0, waitcnt=0)
            continue
        if cur is None:
            continue
        s = line.strip()
        if s.startswith("s_endpgm"):
            cur = None
            continue
        if not s or s.startswith((";", ".")) or s.endswith(":"):
            continue
        op = s.split()[0]
        st = stats[cur]
        st["instr"] += 1
        if op.startswith(("s_cbranch", "s_branch")): st["branch"] += 1
        elif op.startswith(("global_load", "flat_load", "buffer_load")): st["gload"] += 1
        elif op.startswith(("global_store", "flat_store", "buffer_store")): st["gstore"] += 1
        elif op.startswith("ds_"): st["lds"] += 1
        elif op in ("v_lshl_add_u64", "v_mad_u64_u32", "v_mad_i64_i32", "v_add_co_u32", "v_addc_co_u32",
                    "v_lshlrev_b64", "v_ashrrev_i64", "v_lshrrev_b64"): st["i64"] += 1
        elif op in ("v_mul_lo_u32", "v_mul_hi_u32"): st["mul32"] += 1
        elif op in ("v_div_scale_f64", "v_rcp_f64", "v_div_fmas_f64", "v_div_fixup_f64"): st["div"] += 1
        elif op == "s_waitcnt": st["waitcnt"] += 1
    dem = subprocess.run(["c++filt"], input="\n".join(stats), capture_output=True, text=True).stdout.split("\n")
    for mangled, d in zip(stats, dem):
        if all(x in d for x in subs) and stats[mangled]["instr"]:
            short = re.sub(r"\(.*", "", d.replace("scint::", ""))
            print(f"{stats[mangled]['instr']:6d} " + " ".join(f"{k}={v}" for k, v in stats[mangled].items() if k != "instr") + "  " + short[:150])


if __name__ == "__main__":
    main()
