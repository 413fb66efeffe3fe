#!/usr/bin/env python3
"""Development helper: print a rocprofv3 *kernel_stats.csv (name cut to 60 chars, calls, average us, total ms)."""
import csv
import glob
import sys
paths = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for row in csv.DictReader(open(paths[0])):
    print("%-60s %6s calls  avg %10.1f us  total %9.3f ms" % (row["Name"].replace("(anonymous namespace)::", "")[:60], row["Calls"], float(row["AverageNs"]) / 1e3,
                                                               float(row["TotalDurationNs"]) / 1e6))
