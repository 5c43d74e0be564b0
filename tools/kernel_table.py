#!/usr/bin/env python3
"""Join rocprofv3 kernel stats with the algorithmic bytes printed by profile_all_kernels.py."""
import csv
import json
import sys

stats, meta = sys.argv[1], json.load(open(sys.argv[2]))
B = meta["algorithmic_bytes"]
rows = []
for r in csv.DictReader(open(stats)):
    name = r["Name"].replace("(anonymous namespace)::", "")
    key = next((k for k in sorted(B, key=len, reverse=True) if k in name), None)
    if "k_step_fused_pipe" in name:      # one pipeline, three cores: price each against its own bytes
        key = "k_sarl_step" if "SarlCore" in name else "k_gain" if "GainCore" in name else "k_step_fused"
    if "k_step_fused_lat<" in name:      # single-step instantiations are priced as the fused step, the T-step ones by their own bytes
        args = name.split("k_step_fused_lat<", 1)[1].split(">", 1)[0].split(",")
        key = "k_step_fused_lat" if len(args) > 3 and args[3].strip() == "true" else "k_step_fused"
    if key is None or "at::native" in name:
        continue
    avg_us = float(r["AverageNs"]) / 1e3
    gbs = B[key] / (avg_us * 1e-6) / 1e9
    rows.append((key, name.split("(")[0].replace("void risvec::", "").replace("risvec::", ""), int(r["Calls"]), avg_us, B[key] / 1e6, gbs))
print("| kernel | calls | avg us | algorithmic MB | GB/s | % of 8 TB/s |\n|---|---|---|---|---|---|")
for key, name, calls, us, mb, gbs in sorted(rows, key=lambda x: -x[3]):
    print("| `%s` | %d | %.1f | %.1f | %.0f | %.1f |" % (name, calls, us, mb, gbs, gbs / 80.0))
