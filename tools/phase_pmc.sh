#!/bin/bash
# What do the L2 / fabric counters say about the clock-phased stores?  rocprofv3 --kernel-trace --pmc (two counters per run, no other
# tracing) over a short headline run with phase_ticks = 8000 and 0; per kernel the median over its dispatches.  Run on the GPU box from
# the repository root; output: gpurun_out/r04b_phase_pmc.log
R=$PWD
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $O/r04b_phase_pmc.log
for pair in "TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL" "TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ" "TCC_EA0_WRREQ_LEVEL TCC_EA0_WRREQ" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL" "TCC_READ_REQ_LATENCY TCC_READ_REQ" "TCC_TAG_STALL TCC_BUSY"; do
	for t in 8000 0; do
		rm -rf $O/prof_pmc
		rocprofv3 --kernel-trace --pmc $pair --output-format csv -d $O/prof_pmc -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-caller --no-live-pmc --no-profile --no-reference-form --value-runs 1 --sustain-seconds 0 --opt phase_ticks=$t > /dev/null 2> $O/phase_pmc.err
		python3 - "$pair" $t $O/prof_pmc >> $O/r04b_phase_pmc.log <<'PY'
import csv, glob, sys, collections, re
pair, t, d = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(list)
for f in glob.glob(d + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_qdot|k_sadd|k_rows_dot_all|k_fisher_y|k_fisher_t)", r["Kernel_Name"])
        if m:
            agg[(m.group(1), r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    v.sort()
    print("phase_ticks=%-5s %-16s %-36s median %.4g  (%d dispatches)" % (t, k, c, v[len(v) // 2], len(v)))
PY
	done
done
cat $O/r04b_phase_pmc.log
