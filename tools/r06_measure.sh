#!/bin/bash
# Round 6: one measurement session per single-GPU BASELINE configuration with the FINAL binary (the commit is written into every
# summary): kernel stats (rocprofv3 --kernel-trace --stats), HBM traffic (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes),
# and the configuration's own JSON line with its `roofline` object.  C3 = bench.py (the headline), C2 / C4 = tools/bench_configs.py.
# Run on the GPU box from the repository root; afterwards, here: python profiles/summarise.py r06_c2 r06/c2_   (etc.)
set -o pipefail
R=$PWD; O=$R/gpurun_out/r06; mkdir -p $O
echo "${COMMIT:-unknown}" > $O/commit.txt
which=${@:-c3 c2 c4}
cd /tmp && export TMPDIR=/tmp
for c in $which; do
	if [ $c = c3 ]; then
		CMD="python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-caller --no-live-pmc --value-runs 1"; SHORT="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-caller --no-live-pmc --no-profile --value-runs 1 --sustain-seconds 0 --no-reference-form"
	else
		CMD="python3 $R/tools/bench_configs.py $c"; SHORT="$CMD"
	fi
	rm -rf $O/${c}_prof_stats $O/${c}_prof_fetch $O/${c}_prof_write
	C4_QUICK=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${c}_prof_fetch -- $SHORT > /dev/null 2> $O/${c}_fetch.err || { echo "$c: FETCH pass failed"; tail -3 $O/${c}_fetch.err; }
	C4_QUICK=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${c}_prof_write -- $SHORT > /dev/null 2> $O/${c}_write.err || { echo "$c: WRITE pass failed"; tail -3 $O/${c}_write.err; }
	C4_QUICK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${c}_prof_stats -- $CMD > $O/${c}_prof_stats.json 2> $O/${c}_stats.err || { echo "$c: stats pass failed"; tail -3 $O/${c}_stats.err; }
	echo "== $c under rocprofv3 --stats:"; tail -c 400 $O/${c}_prof_stats.json; echo
	# the configuration's own line, outside the profiler (the number that is quoted), with the traffic the two passes above counted
	(cd $R && python3 profiles/summarise.py r06_${c} r06/${c}_ > $O/${c}_summarise.txt 2>&1)
	(cd $R && PMC_JSON=$R/profiles/r06_${c}_pmc_traffic.json C4_QUICK=1 $CMD > $O/${c}_line.json 2> $O/${c}_line.err) || { echo "$c: plain run failed"; tail -3 $O/${c}_line.err; }
	echo "== $c plain:"; cut -c1-600 $O/${c}_line.json
done
ls $O
