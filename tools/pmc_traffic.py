"""profiles/<round>_pmc_traffic_<workload>.json (round = 5th argument, default r02) from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in separate runs) of
`python3 bench.py --workload W --steps 5 --warmup 1 --no-cpu-baseline --no-search`; corrections as
MI355X_MICROARCH.md prescribes (values are KB; gfx950 FETCH_SIZE counts wide coalesced reads at half -> x2)."""
import csv, glob, json, sys
wl, fetch_dir, write_dir, algo = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
rnd = sys.argv[5] if len(sys.argv) > 5 else "r03"
def per_launch(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_oplist<11>" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(vals) / len(vals), len(vals), f
fk, nf, ff = per_launch(fetch_dir, "FETCH_SIZE")
wk, nw, wf = per_launch(write_dir, "WRITE_SIZE")
import os
try:
    commit = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pepr_amd", "BUILD_COMMIT")).read().strip()
except OSError:
    commit = None
out = {"workload": wl, "kernel": "pml::k_oplist<11>", "commit": commit,
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --workload %s --steps 5 --warmup 1 --no-cpu-baseline --no-search" % wl,
       "FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk, "launches_averaged": [nf, nw],
       "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact for 16-B/lane stores",
       "traffic_bytes_per_launch": 2 * fk * 1024 + wk * 1024, "algorithmic_bytes_per_launch": algo}
json.dump(out, open("profiles/%s_pmc_traffic_%s.json" % (rnd, wl), "w"), indent=1)
print(json.dumps(out, indent=1))
