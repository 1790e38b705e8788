"""Offline ISA check (no GPU): compiler-inserted `s_waitcnt vmcnt(N)` inside the MFMA loops of every kernel of a .s file.
A wait hipcc places at a LOOP HEADER for a hazard that lives on a rare path — an epilogue's loads / stores still pending on the back
edge into the K loop — runs in EVERY iteration, and with LDS-DMA pieces in the same counter it is `vmcnt(0)`: the staging queue
drained once per K tile. Round 5 found one in gemm_tn_pp_kernel (3 % of the kernel) and one in the 256-row GELU instance of
gemm_nt_pp_kernel (7 %). Waits written by the source (inside ;;#ASMSTART ... ;;#ASMEND) are not reported; by default only loops that
issue LDS-DMA are (register-staged kernels need their waits; --all lists those too).
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only grove_amd/csrc/X.hip -o /tmp/X.s && python tools/isa_loop_waits.py /tmp/X.s"""
import re
import sys


def scan(path, dma_only=True):
    """-> list of dicts {kernel, loop: (first, last line), mfmas, dma, waits: [(line, N, next instruction)]}"""
    src = open(path).read().split("\n")
    starts = [i for i, l in enumerate(src) if re.match(r"^_Z\S+:\s*;\s*@", l)]
    out = []
    for s in starts:
        name = src[s].split(":")[0]
        try:
            e = next(i for i in range(s, len(src)) if "s_endpgm" in src[i])
        except StopIteration:
            continue
        body = src[s:e]
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        loops = []
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        seen = set()
        for a, b in sorted(loops, key=lambda x: x[1] - x[0]):  # innermost first; enclosing loops are not reported again
            seg = body[a:b + 1]
            nm = sum("v_mfma" in l for l in seg)
            if nm < 8 or any(a <= x and y <= b for x, y in seen):
                continue
            seen.add((a, b))
            dma = sum("global_load_lds" in l or ("buffer_load" in l and " lds" in l) for l in seg)
            if dma_only and dma == 0:
                continue
            in_asm, found = False, []
            for i, l in enumerate(seg):
                if "#ASMSTART" in l:
                    in_asm = True
                elif "#ASMEND" in l:
                    in_asm = False
                elif not in_asm:
                    m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)
                    if m:
                        nxt = next((x.strip() for x in seg[i + 1:] if x.strip() and not x.strip().startswith((";", "."))), "")
                        found.append((a + i, int(m.group(1)), nxt[:60]))
            out.append({"kernel": name, "loop": (a, b), "mfmas": nm, "dma": dma, "waits": found})
    return out


def short(name):
    return re.sub(r"^_ZN\d+_GLOBAL__N_\d+", "", name)[:80]


if __name__ == "__main__":
    tot = 0
    for r in scan(sys.argv[1], dma_only="--all" not in sys.argv):
        if r["waits"]:
            print(f"{short(r['kernel'])}: loop lines {r['loop'][0]}-{r['loop'][1]} ({r['mfmas']} MFMAs, {r['dma']} LDS-DMA): {len(r['waits'])} compiler vmcnt wait(s)")
            for ln, n, nxt in r["waits"][:6]:
                print(f"      line {ln}: vmcnt({n}) before `{nxt}`")
            tot += len(r["waits"])
    print("total compiler-inserted vmcnt waits inside MFMA loops" + ("" if "--all" in sys.argv else " that issue LDS-DMA") + ":", tot)
