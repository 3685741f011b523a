#!/bin/bash
# dynamic instruction mix of the chained scoring kernel (one counter group per rocprofv3 run): tools/pmc_instmix.sh OUTDIR
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/$1; mkdir -p $O
rocprofv3 --list-avail > $O/avail.txt 2>&1
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_FLAT_LDS_ONLY" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY" "SQ_WAVES SQ_IFETCH SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp -d $O/p$i --output-format csv -- python3 $R/tools/score_loop.py 12 > $O/p$i.log 2>&1 || echo "pass $i ($grp) failed" | tee -a $O/errors.txt
  echo "pass $i done: $grp" >> $O/progress.txt
done
cd $R && python tools/pmc_summary.py $O "k_oplist<11>" "instruction mix, C3 chained scoring pass" > $O/summary.json; find $O -name "*.csv" -size +1M -delete; rm -rf $O/p*/*/*agent_info.csv
cat $O/summary.json | head -60; cat $O/errors.txt 2>/dev/null
