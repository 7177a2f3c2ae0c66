"""Summarise rocprofv3 csv outputs of tools/profile_gpu.sh: per-kernel averages."""
import csv, glob, collections, sys, os
d = sys.argv[1]
for tag in sorted(os.listdir(d)):
    fs = glob.glob(os.path.join(d, tag, "*", "*_counter_collection.csv"))
    for f in fs:
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "sweep_kernel" in r["Kernel_Name"] or "prepass" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"], r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("Scratch_Size"))].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(tag, k, "n=%d" % len(v), "max=%.5g" % max(v), "sum=%.5g" % sum(v))
    fs = glob.glob(os.path.join(d, tag, "*", "*_kernel_stats.csv"))
    for f in fs:
        print(open(f).read())
