#!/bin/bash
# L2 hit-rate / wait-state counters of one igemm shape (GPU box).  usage: pmc_l2.sh B H C N k tag
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_l2_$6
mkdir -p $O
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE --kernel-trace -d $O/p1 -o p1 --output-format csv -- python $R/tools/pmc_one.py $1 $2 $3 $4 $5 > $O/p1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace -d $O/p2 -o p2 --output-format csv -- python $R/tools/pmc_one.py $1 $2 $3 $4 $5 > $O/p2.log 2>&1
cd $R && python tools/pmc_summary.py "$O/p1/**/*counter_collection.csv" > $O/p1.txt 2>&1; python tools/pmc_summary.py "$O/p2/**/*counter_collection.csv" > $O/p2.txt 2>&1
cat $O/p1.txt $O/p2.txt | grep -v "^$" | head -40
