#!/bin/bash
# Which hardware counter tells the fast state of pass 2 from the slow one?  (DESIGN.md 3.3 "Placement")
R=$PWD; O=$R/gpurun_out/r03/pmc2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_SERIALIZATION_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/set$i -- python3 $R/scratch/placement5.py > $O/set$i.log 2>$O/set$i.err
  echo "$set" > $O/set$i.counters
done
cd $R
python3 scratch/placement_pmc_analyse.py
