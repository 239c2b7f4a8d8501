import csv, glob, os, collections, json
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r03", os.environ.get("PMC_DIR", "pmc2"))
for d in sorted(glob.glob(os.path.join(root, "set*"))):
    if not os.path.isdir(d):
        continue
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not cc or not kt:
        print(d, "no output"); continue
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(cc[0])):
        name = r["Kernel_Name"]
        for short in ("k_qdot", "k_sadd", "k_rows_dot_all"):
            if short in name:
                per[(short, r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    for short in ("k_qdot", "k_sadd", "k_rows_dot_all"):
        rows = [(dur.get(k[1]), v) for k, v in per.items() if k[0] == short and dur.get(k[1])]
        if not rows:
            continue
        rows.sort(key=lambda t: t[0])
        lo, hi = rows[:len(rows) // 3], rows[-(len(rows) // 3):]
        def mean(rs, c): return sum(v[c] for _, v in rs) / len(rs)
        names = sorted(rows[0][1])
        print(os.path.basename(d), short, "fast third %.3f ms, slow third %.3f ms" % (sum(t for t, _ in lo) / len(lo), sum(t for t, _ in hi) / len(hi)),
              {c: ("%.4g -> %.4g (x%.3f)" % (mean(lo, c), mean(hi, c), mean(hi, c) / max(mean(lo, c), 1e-30))) for c in names})
