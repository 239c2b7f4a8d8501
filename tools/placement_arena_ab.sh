#!/bin/bash
# tools/placement_arena_ab.sh -- VERDICT r03 #5: does it matter where the library's OWN buffers land?  Ten fresh processes of the
# headline workload each way -- scratch pools from hipMalloc as they come (arena 0) against one arena reserved at the library's
# first call, before the caller makes its arrays (STOCHQN_HIP_ARENA_MB=256) -- interleaved, one JSON line per process.
out=${1:-gpurun_out/r04_placement_arena.jsonl}
: > "$out"
VARIANTS=${VARIANTS:-"0 256"}        # MB; a trailing u = reserved but not used (STOCHQN_HIP_ARENA_UNUSED=1)
for i in 1 2 3 4 5 6 7 8 9 10; do
	for v in $VARIANTS; do
		mb=${v%u}; unused=""; [ "$v" != "$mb" ] && unused=1
		env STOCHQN_HIP_ARENA_MB=$mb ${unused:+STOCHQN_HIP_ARENA_UNUSED=1} python bench.py --steps 100 --no-cpu-baseline --no-host-caller --no-live-pmc --no-reference-form --no-profile \
			--sustain-seconds 0 --value-runs 1 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print(json.dumps({'arena_mb': $mb, 'variant': '$v', 'process': $i, 'value': d['value'], 'ms_per_step': d['ms_per_step']}))" >> "$out" || exit 1
	done
done
python - "$out" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
for var in sorted(set(r["variant"] for r in rows), key=lambda s: (int(s.rstrip("u")), s)):
    v = sorted(r["value"] for r in rows if r["variant"] == var)
    print("arena %5s MB: n=%d min %.2f median %.2f mean %.2f max %.2f spread %.1f %%" % (var, len(v), v[0], v[len(v) // 2], sum(v) / len(v), v[-1], 100 * (v[-1] / v[0] - 1)))
PY
