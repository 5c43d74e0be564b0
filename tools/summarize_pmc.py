#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: mean FETCH_SIZE / WRITE_SIZE per dispatch of each kernel.
FETCH_SIZE/WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts a wide coalesced
stream at half its bytes (MI355X_MICROARCH.md, HBM section) - both raw and x2 are printed."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
res = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(out, "pmc_%s_%s" % (counter, tag), "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        res.setdefault(k, {})[counter] = dict(mean_kib=sum(v) / len(v), n=len(v))
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", {}).get("mean_kib", 0))[:12]:
    f = v.get("FETCH_SIZE", {}).get("mean_kib", 0.0)
    w = v.get("WRITE_SIZE", {}).get("mean_kib", 0.0)
    print("%-60s fetch %.1f MiB (x2 corrected %.1f MiB)  write %.1f MiB  n=%d" % (k[:60], f / 1024, 2 * f / 1024, w / 1024,
          v.get("FETCH_SIZE", v.get("WRITE_SIZE"))["n"]))
json.dump(res, open(os.path.join(out, "pmc_summary_%s.json" % tag), "w"), indent=1)
