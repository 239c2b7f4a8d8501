"""Turns the raw rocprofv3 output under gpurun_out/ into the small summaries committed here.

    python profiles/summarise.py r01
    python profiles/summarise.py r05_c2 r05/c2_        (round 5: one set of passes per BASELINE configuration; inputs then
                                                        gpurun_out/r05/c2_prof_stats, ..._prof_fetch, ..._prof_write, ..._prof_stats.json)

Inputs (written on the GPU box by the commands quoted in profiles/README.md):
    gpurun_out/prof_stats/*/*_kernel_stats.csv       rocprofv3 --kernel-trace --stats
    gpurun_out/prof_fetch/*/*_counter_collection.csv rocprofv3 --kernel-trace --pmc FETCH_SIZE
    gpurun_out/prof_write/*/*_counter_collection.csv rocprofv3 --kernel-trace --pmc WRITE_SIZE
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts wide coalesced streaming reads at
half their size (MI355X_MICROARCH.md, HBM section), so reads are doubled.  The torch `mul` kernel
in the same run (2 reads + 1 write of n doubles, 16 B per lane) is the calibration line.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
pre = sys.argv[2] if len(sys.argv) > 2 else ""               # prefix of the input directories under gpurun_out/
out = os.path.join(ROOT, "profiles")


def short(name):
    name = name.replace("sqn::(anonymous namespace)::", "")
    m = re.search(r"k_sweep<(\d), (\d), (\w+)(<[^>]*>)?", name)
    if m:
        return "k_sweep<W=%s,NP=%s,%s%s>" % (m.group(1), m.group(2), m.group(3), m.group(4) or "")
    m = re.search(r"(k_rows_dot_all|k_rows_dot|k_combine|k_qdot|k_sadd|k_fisher_t_split|k_fisher_t|k_fisher_y|k_pair_y_diff|k_c2_step)<([^>]*)>", name)
    if m:
        return "%s<%s>" % (m.group(1), m.group(2))
    return re.sub(r"\(.*", "", name)[:80]


def newest(pattern):
    """gpurun merges every call's output into the same directories: take the most recent run's file"""
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]


stats = newest(os.path.join(ROOT, "gpurun_out", pre + "prof_stats", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(out, tag + "_rocprofv3_kernel_stats.csv"))

pmc = {}
for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    files = newest(os.path.join(ROOT, "gpurun_out", pre + "prof_" + kind, "*", "*_counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == ctr:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v.sort()
        pmc.setdefault(k, {})[ctr] = {"dispatches": len(v), "median_KiB": v[len(v) // 2], "max_KiB": v[-1]}

summary = {}
for k, d in pmc.items():
    rd = d.get("FETCH_SIZE", {}).get("median_KiB", 0.0) * 1024 * 2      # gfx950 half-count correction
    wr = d.get("WRITE_SIZE", {}).get("median_KiB", 0.0) * 1024
    summary[k] = {"read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr, "raw": d}
json.dump(summary, open(os.path.join(out, tag + "_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
if os.path.exists(os.path.join(ROOT, "gpurun_out", pre + "prof_stats.json")):
    shutil.copy(os.path.join(ROOT, "gpurun_out", pre + "prof_stats.json"), os.path.join(out, tag + "_bench_under_rocprofv3.json"))
for k in sorted(summary, key=lambda k: -summary[k]["hbm_bytes_per_launch"])[:12]:
    s = summary[k]
    print("%-60s read %.3e  write %.3e  total %.3e" % (k, s["read_bytes_per_launch"], s["write_bytes_per_launch"], s["hbm_bytes_per_launch"]))
