#!/bin/bash
set -o pipefail
O=gpurun_out/soak; mkdir -p $O
AMD_LOG_LEVEL=4 AMD_LOG_MASK=2147483647 tools/pin_probe paths 2> /tmp/paths.log > /dev/null; wc -l /tmp/paths.log
grep -v "hipGetLastError\|hipSetDevice\|hipGetDevice" /tmp/paths.log | cut -c1-240 | head -1500 > $O/probe_paths_raw.txt
profiles/src/tune12 > gpurun_out/r05_tune_fused_update.log 2>&1; tail -16 gpurun_out/r05_tune_fused_update.log
bash tools/r05_ladder.sh 1000000 > gpurun_out/r05_ladder_device.log 2>&1; cat gpurun_out/r05_ladder_device.log
for spec in "8388608 h2d sync" "8388608 h2d async" "8388608 d2h async" "25165824 h2d async" "25165824 d2h async" "167772160 h2d sync" "167772160 d2h sync" "167772160 h2d async" "167772160 d2h async" "272629760 d2h async"; do
	tools/pin_probe stale $spec 3 > $O/stale.txt 2>&1; rc=$?
	echo "== stale $spec: rc $rc"; tail -3 $O/stale.txt; cat $O/stale.txt >> $O/probe_stale.txt
	if [ $rc -ne 0 ]; then echo "stopping after the first failure"; break; fi
done
