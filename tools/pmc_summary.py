"""Per-launch means of every counter tools/pmc_collect.sh gathered, for the kernels whose name contains argv[2] (default k_oplist)."""
import csv, glob, json, sys
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else "k_oplist"
acc = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc.setdefault((r["Kernel_Name"].split("(")[0], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
out = {}
for (k, c), v in sorted(acc.items()):
    out.setdefault(k, {})[c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
print(json.dumps(out, indent=1))
