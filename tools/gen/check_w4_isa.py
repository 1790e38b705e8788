"""Checks the compiled gemm_nt_w4_kernel instances: outside the hand-written asm statements the compiler must not touch the
accumulator AGPRs at all, and must not touch the fragment VGPRs v128..v255 in any block of the K loop (between two bodies).
Usage: python tools/gen/check_w4_isa.py gemm.s   (gemm.s = hipcc -S --cuda-device-only of grove_amd/csrc/gemm.hip)"""
import re
import sys


def main():
    text = open(sys.argv[1]).read().split("\n")
    bad_a, hi_v = [], {}
    in_fn, in_asm, label = False, False, None
    fn = None
    has_body = {}
    for ln in text:
        m = re.match(r"^(_ZN\S*gemm_nt_w4_kernel\S*):", ln)
        if m:
            in_fn, fn = True, m.group(1)
            continue
        if in_fn and ln.startswith("\t.amdhsa_kernel") or (in_fn and ln.startswith(".Lfunc_end")):
            in_fn = False
        if not in_fn:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            label = (fn, m.group(1))
        if "#ASMSTART" in ln:
            in_asm = True
            continue
        if "#ASMEND" in ln:
            in_asm = False
            continue
        code = ln.split(";")[0]
        if in_asm:
            if "v_mfma" in code:
                has_body[label] = True
            continue
        if re.search(r"\ba\[?\d+", code) and "v_accvgpr" not in code:
            bad_a.append((label, code.strip()))
        elif "v_accvgpr" in code:
            bad_a.append((label, code.strip()))
        for r in re.findall(r"\bv\[?(\d+)(?::(\d+))?\]?", code):
            top = int(r[1]) if r[1] else int(r[0])
            if top >= 128:
                hi_v.setdefault(label, []).append(code.strip())
    print("compiler-made AGPR uses:", len(bad_a))
    for b in bad_a[:10]:
        print("   ", b)
    blocks_bad = [l for l in hi_v if has_body.get(l)]
    print("blocks that hold a K-tile body AND use v128+ outside it:", len(blocks_bad))
    for l in blocks_bad[:10]:
        print("   ", l, hi_v[l][:3])
    print("other blocks using v128+ (epilogues, prologue):", len(hi_v) - len(blocks_bad))
    return 1 if bad_a or blocks_bad else 0


if __name__ == "__main__":
    sys.exit(main())
