#!/usr/bin/env python3
"""gather24_report.py OUT_DIR STDOUT_LOG -- FETCH_SIZE of tools/ubench/gather24.hip's dispatches (second round) against their known counts"""
import csv, glob, os, sys
d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and any(k in r["Kernel_Name"] for k in ("gather24", "rows24", "stream16")):
            rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"].split("(")[0], float(r["Counter_Value"]),
                         (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
rows.sort()
assert len(rows) == 10, len(rows)
nfetch = 256 * 32 * 256 * 64.0
big = float(3 << 30)
label = ["stream16 3072 MiB", "gather24 3072 MiB", "rows24 3072 MiB", "gather24 96 MiB", "rows24 96 MiB"]
print("FETCH_SIZE is reported in KiB (rocprofv3 derived counter); second round of dispatches; times under the profiler")
for (t, name, v, ms), lab in zip(rows[5:], label):
    b = v * 1024
    if name.endswith("stream16"):
        print("%-20s FETCH_SIZE %.4e B for %.4e B read: ratio %.3f  (%.2f ms)" % (lab, b, big, b / big, ms))
    else:
        print("%-20s FETCH_SIZE %.4e B for %.4e records: %.1f B per 24-byte record  [80 = 64-byte requests counted in full, "
              "72 = 128-byte requests counted as 64]  (%.2f ms, %.2f G records/s)" % (lab, b, nfetch, b / nfetch, ms, nfetch / ms / 1e6))
if len(sys.argv) > 2:
    print("--- the program's own output (HIP events, same run) ---")
    sys.stdout.write(open(sys.argv[2]).read())
