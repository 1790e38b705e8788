cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $c -d gpurun_out/pmc_mix/p$i -o p -- python3 tools/dev/pmc_gemm_mix.py > gpurun_out/pmc_mix_$i.log 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_mix/p$i gemm_nt_pp > gpurun_out/pmc_mix_$i.json
done
python3 - <<'PY'
import json
out={}
for i in (1,2,3):
    d=json.load(open(f'gpurun_out/pmc_mix_{i}.json'))
    for k,v in d.items(): out.setdefault(k,{}).update(v)
json.dump(out,open('gpurun_out/pmc_gemm_mix.json','w'),indent=1)
for k,v in out.items():
    mf=v.get('SQ_VALU_MFMA_BUSY_CYCLES',1)
    print(k[:48], 'wait_any',v.get('SQ_WAIT_ANY_share_of_wave_cycles'),'active',v.get('SQ_ACTIVE_INST_ANY_share_of_wave_cycles'),'| per 1000 MFMA-busy cycles: VALU %.1f SALU %.1f LDS %.1f VMEM_RD %.2f VMEM_WR %.2f'%tuple(1000*v.get(c,0)/mf for c in ('SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_INSTS_VMEM_RD','SQ_INSTS_VMEM_WR')),'| L2 hit %.3f'%(v.get('TCC_HIT_sum',0)/max(v.get('TCC_REQ_sum',1),1)))
PY
