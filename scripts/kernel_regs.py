"""Prints VGPR / AGPR / scratch / LDS / occupancy of the kernels of a -save-temps gfx950 assembly file whose names match a
pattern:  python scripts/kernel_regs.py pytheiasfm_amd/csrc/_obj/ransac-hip-amdgcn-amd-amdhsa-gfx950.s 'k_g?dls'"""
import re
import sys


def main():
    text = open(sys.argv[1]).read()
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", text, re.S):
        name, body = m.group(1), m.group(2)
        if pat and not pat.search(name):
            continue
        tail = text[m.end():m.end() + 4000]
        def stat(key):
            r = re.search(r"; %s:? =? ?(\d+)" % key, tail)
            return r.group(1) if r else "?"
        short = re.sub(r"^_ZN?\d*", "", name)[:48]
        print(f"{short:50s} vgpr {stat('NumVgprs'):>4s} agpr {stat('NumAgprs'):>3s} scratch {stat('ScratchSize'):>5s} "
              f"lds {stat('LDSByteSize'):>6s} occupancy {stat('Occupancy'):>2s} code {stat('codeLenInByte'):>7s}")


if __name__ == "__main__":
    main()
