#!/usr/bin/env python3
"""Re-quote `roofline.traffic` of committed bench.py lines from the committed PMC files with bench.py's CURRENT lookup.

Round 5's lookup took the lexicographically last profiles/r*_pmc_traffic*.json -- the adaQN run (C4) -- for every line whose
traffic was not counted live, so SQN lines carried the first `k_sadd` / `k_qdot` entry of that file (5.6 / 7.0 / 20.0 GB against
17.6 / 22.0 GB algorithmic: VERDICT r05 weak #3).  Nothing about the MEASUREMENT changes here: timings, rates and fractions stay
as they were taken; only the looked-up traffic figure, its source and the ratio are replaced, and the record says so.

    python tools/requote_traffic.py profiles/r05_c3_line.json profiles/r05_c5_shard_1gpu.json ...
"""
import importlib.util
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
WORDS = {"sdot": lambda m: m + 1, "qdot": lambda m: m + 2, "sadd": lambda m: m + 2, "bwd": lambda m: 4, "fwd": lambda m: 4}


def requote(roof, m, config):
    kernel = roof["kernel"].split()[0]
    if kernel not in WORDS or "counted in this run" in (roof.get("traffic_source") or ""):
        return False
    alg = roof["alg_bytes_per_launch"]
    n = alg // (WORDS[kernel](m) * 8)
    tr, src = bench.pmc_traffic(kernel, n, m, alg, config=config)
    if tr == roof.get("traffic"):
        return False
    roof["traffic_as_committed_in_round_5"] = {"traffic": roof.get("traffic"), "traffic_source": roof.get("traffic_source")}
    roof["traffic"], roof["traffic_source"] = tr, src
    roof["traffic_over_algorithmic"] = round(tr / alg, 4) if tr else None
    roof["traffic_requoted"] = ("round 6, tools/requote_traffic.py: the lookup of round 5 served this line from the adaQN run's file; "
                                "re-quoted from the committed counters of this kernel instantiation, the measurement itself is unchanged")
    return True


def fix(doc):
    changed = False
    cfg = doc.get("config")
    if not isinstance(cfg, dict) or "workload" not in cfg:
        return False
    m = int(re.search(r"m=(\d+)", cfg["workload"]).group(1))
    name = cfg.get("name", "c3")
    name = name if name in bench.CONFIG_PROFILE else "c3"
    for roof in (doc.get("roofline"), (doc.get("reference_form") or {}).get("roofline")):
        if isinstance(roof, dict) and "alg_bytes_per_launch" in roof:
            changed |= requote(roof, m, name)
    return changed


for path in sys.argv[1:]:
    text = open(path).read()
    if path.endswith(".jsonl"):
        docs = [json.loads(l) for l in text.splitlines() if l.strip()]
        n = sum(fix(d) for d in docs)
        if n:
            open(path, "w").write("".join(json.dumps(d) + "\n" for d in docs))
    else:
        d = json.loads(text)
        n = int(fix(d))
        if n:
            open(path, "w").write(json.dumps(d) + "\n")
    print("%s: %d line(s) re-quoted" % (path, n))
