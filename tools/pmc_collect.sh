#!/bin/bash
# PMC passes over the scoring bench (one counter group per rocprofv3 run, as MI355X_MICROARCH.md prescribes):
#   tools/pmc_collect.sh OUTDIR [workload] [extra bench flag, e.g. --stored-only]
# then tools/pmc_summary.py OUTDIR > profiles/<round>_pmc_k_oplist_<leg>_<workload>.json (stamped with the build's commit)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/$1; W=${2:-c3}; X=${3:---no-stored}; mkdir -p $O
export BENCH_NO_C4=1 BENCH_CLOCK_WARMUP_S=0.2
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  if [ -n "$PMC_PASSES" ] && [ $i -gt $PMC_PASSES ]; then break; fi
  timeout -k 10 400 rocprofv3 --pmc $grp -d $O/p$i --output-format csv -- python3 $R/bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline --no-search $X > $O/p$i.log 2>&1 || echo "pass $i ($grp) failed" | tee -a $O/errors.txt
  echo "pass $i done: $grp" >> $O/progress.txt
done
