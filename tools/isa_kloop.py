"""Offline ISA check of the persistent GEMM instances (no GPU needed): per gemm_nt_pp_kernel instance, total s_waitcnt vmcnt(0), scratch
ops, 128-bit LDS ops and — per K-loop body (runs of MFMAs) — the vmcnt(0) waits INSIDE it (the baseline has the four explicit tail waits;
anything more is hipcc guarding LDS reads / reloading spilled staging pointers, i.e. the staging queue drained once per tile).
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -Iinclude grove_amd/csrc/gemm.hip -o /tmp/gemm.s && python tools/isa_kloop.py /tmp/gemm.s"""
import re,sys
lines=open(sys.argv[1]).read().split('\n')
starts=[(i,l.split(':')[0]) for i,l in enumerate(lines) if re.match(r'^_Z.*gemm_nt_pp_kernel.*:\s*;',l)]
starts.append((len(lines),'end'))
for (a,name),(b,_) in zip(starts,starts[1:]):
    body=lines[a:b]
    try: e=next(i for i,l in enumerate(body) if 's_endpgm' in l)
    except StopIteration: e=len(body)
    meta=' '.join(l.strip() for l in lines[a+e:b] if re.search(r'\.sgpr_count|\.vgpr_count|scratch|NumVgprs|ScratchSize|Occupancy|LDSByteSize',l))[:0]
    body=body[:e]
    mf=[i for i,l in enumerate(body) if 'v_mfma' in l]
    runs=[]; s=mf[0]; p=mf[0]
    for i in mf[1:]:
        if i-p>120: runs.append((s,p)); s=i
        p=i
    runs.append((s,p))
    det=[]
    for (s,p) in runs:
        n=sum(1 for l in body[s:p+1] if re.search(r's_waitcnt.*vmcnt\(0\)',l))
        nm=sum(1 for l in body[s:p+1] if 'v_mfma' in l)
        det.append((nm,n))
    n0=sum(1 for l in body if re.search(r's_waitcnt.*vmcnt\(0\)',l))
    nsc=sum(1 for l in body if 'scratch_' in l)
    nds=sum(1 for l in body if 'ds_write_b128' in l or 'ds_read_b128' in l)
    m=re.search(r'gemm_nt_pp_kernelILi(\d+)ELb(\d)ELi(n?\d+)ELb(\d)',name)
    print(f"BM {m.group(1)} g {m.group(2)} act {m.group(3):>3s} fp8 {m.group(4)} lines {e:6d} vmcnt0 {n0:3d} scratch {nsc:3d} ds128 {nds:4d} K-bodies {det}")
