#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into HBM bytes per
launch for each kernel, with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE tallies 128-B requests at 64 B: x2;
both counters are in KiB).  Usage: hbm_traffic_summary.py <fetch counter_collection.csv> <write counter_collection.csv> [out.json]"""
import csv, collections, json, re, sys

def load(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = re.sub(r"\(.*", "", name).strip()
        a = acc[name]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
rows = []
for k in fetch:
    n, f = fetch[k]
    nw, w = write.get(k, [0, 0.0])
    if n == 0 or nw != n:
        continue
    rb, wb = 2.0 * f * 1024.0 / n, w * 1024.0 / n
    rows.append({"kernel": k, "launches": n, "hbm_read_bytes_per_launch": rb, "hbm_write_bytes_per_launch": wb,
                 "hbm_bytes_per_launch": rb + wb, "total_gb": (rb + wb) * n / 1e9})
rows.sort(key=lambda r: -r["total_gb"])
for r in rows[:16]:
    print(f'{r["total_gb"]:9.2f} GB  n={r["launches"]:5d}  rd/launch={r["hbm_read_bytes_per_launch"]/1e6:9.2f} MB  wr/launch={r["hbm_write_bytes_per_launch"]/1e6:9.2f} MB  {r["kernel"][:70]}')
if len(sys.argv) > 3:
    json.dump({"note": "HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, separate --pmc passes; see profiles/README.md",
               "kernels": rows[:40]}, open(sys.argv[3], "w"), indent=1)
