"""Static instruction-class histogram of the gfx950 kernels of an assembly file (hipcc --cuda-device-only -S):
  python scripts/isa_histogram.py <file.s> [kernel-name regex]
Per kernel: instruction counts by class (FP64 FMA / MUL / ADD / other FP64, transcendental, integer + moves, compares /
selects, DPP + cross-lane, LDS, global / scratch memory, scalar ALU, scalar memory, waits, branches) and the register /
LDS / scratch figures of the kernel descriptor.  Static counts: a loop body counts once."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_fma_f64") or op.startswith("v_fmac_f64"):
        return "fp64 fma"
    if op.startswith("v_mul_f64"):
        return "fp64 mul"
    if op.startswith("v_add_f64"):
        return "fp64 add"
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64", "v_trig", "v_frexp", "v_ldexp")):
        return "fp64 transcendental / division steps"
    if op.startswith("v_mfma"):
        return "mfma"
    if "_f64" in op:
        return "fp64 other (min / max / cmp / cvt / fract ...)"
    if "dpp" in op or op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "ds_bpermute", "ds_permute", "ds_swizzle", "v_permlane")):
        return "cross-lane (dpp / readlane / permute)"
    if op.startswith(("v_cmp", "v_cndmask")):
        return "compare / select"
    if op.startswith("v_accvgpr"):
        return "agpr moves"
    if op.startswith("v_"):
        return "valu integer / 32-bit / moves"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "global memory"
    if op.startswith("scratch_"):
        return "scratch (spills)"
    if op.startswith(("s_load", "s_buffer_load", "s_store")):
        return "scalar memory"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep")):
        return "waits / nops"
    if op.startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm", "s_setpc", "s_swappc")):
        return "branches / barriers"
    if op.startswith("s_"):
        return "scalar alu"
    return "other"


def main():
    text = open(sys.argv[1]).read()
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if pat and not pat.search(name):
            continue
        if ".amdhsa_kernel " + name not in text:
            continue
        hist = collections.Counter()
        for line in body.split("\n"):
            t = line.strip()
            if not t or t.startswith((";", ".")) or t.endswith(":"):
                continue
            hist[classify(t.split()[0])] += 1
        tail = text[text.index(".amdhsa_kernel " + name):][:6000]
        def stat(key):
            r = re.search(r"; %s:? =? ?(\d+)" % key, tail)
            return r.group(1) if r else "?"
        total = sum(hist.values())
        print(f"{name}\n  {total} instructions; VGPRs {stat('NumVgprs')} + AGPRs {stat('NumAgprs')}, scratch {stat('ScratchSize')} B/lane, "
              f"LDS {stat('LDSByteSize')} B, occupancy {stat('Occupancy')} waves/SIMD")
        for k, v in hist.most_common():
            print(f"    {k:48s} {v:6d}  {100.0 * v / total:5.1f} %")


if __name__ == "__main__":
    main()
