#!/usr/bin/env python3
"""Print the head of a rocprofv3 kernel_stats.csv compactly.  usage: top_kernels.py <stats.csv> [n]"""
import csv
import sys

n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in list(csv.DictReader(open(sys.argv[1])))[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("risvec::", "")
    print("%-72s calls %6s  avg %9.1f us  %6s%%" % (name[:72], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
