#!/bin/bash
# PMC passes over the C3 scoring bench (one counter group per run, as the guide prescribes): tools/pmc_collect.sh OUTDIR
# then tools/pmc_summary.py OUTDIR > profiles/<round>_pmc_k_oplist_c3.json
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/$1; mkdir -p $O
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp -d $O/p$i --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-search > $O/p$i.log 2>&1 || echo "pass $i ($grp) failed" | tee -a $O/errors.txt
  echo "pass $i done: $grp" >> $O/progress.txt
done
