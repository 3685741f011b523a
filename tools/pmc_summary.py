"""Per-launch means of every counter tools/pmc_collect.sh gathered, for the kernels whose name contains argv[2] (default
k_oplist<11>), with the derived figures bench.py reads and the commit of the build that was measured:
python tools/pmc_summary.py OUTDIR [kernel substring] [description] > profiles/<round>_pmc_k_oplist_<leg>_<workload>.json"""
import csv, glob, json, os, sys
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else "k_oplist<11>"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
try:
    commit = open(os.path.join(root, "pepr_amd", "BUILD_COMMIT")).read().strip()
except OSError:
    commit = None
acc = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
n = {k: len(v) for k, v in acc.items()}
der = {}
if "GRBM_GUI_ACTIVE" in c:
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0                    # summed over the 8 XCDs
    der["kernel_cycles"] = cyc
    if "SQ_INSTS_VALU_MFMA_MOPS_F64" in c:
        mf = c["SQ_INSTS_VALU_MFMA_MOPS_F64"]; der["mfma_instructions"] = mf
        der["algorithmic_GFLOP_executed"] = mf * 512 / 1e9       # v_mfma_f64_4x4x4_4b: 4 blocks x 4x4x4 x 2 flop per instruction
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            der["busy_cycles_per_mfma"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / mf
            der["mfma_pipe_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)      # 1024 SIMDs
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # values are KB; gfx950: FETCH_SIZE counts wide coalesced reads at half (MI355X_MICROARCH.md, HBM section) -> x2
    der["hbm_traffic_GB"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / 1e9
if "SQ_WAVE_CYCLES" in c:
    wc = c["SQ_WAVE_CYCLES"]
    if "SQ_WAIT_ANY" in c: der["wave_cycles_waitcnt_frac"] = c["SQ_WAIT_ANY"] / wc
    if "SQ_WAIT_INST_ANY" in c: der["wave_cycles_issue_stall_frac"] = c["SQ_WAIT_INST_ANY"] / wc
    if "SQ_ACTIVE_INST_ANY" in c: der["wave_cycles_issuing_frac"] = c["SQ_ACTIVE_INST_ANY"] / wc
if "SQ_LDS_BANK_CONFLICT" in c: der["lds_bank_conflicts"] = c["SQ_LDS_BANK_CONFLICT"]
print(json.dumps({"what": sys.argv[3] if len(sys.argv) > 3 else "", "kernel": pat, "commit": commit,
                  "source": "tools/pmc_collect.sh (separate rocprofv3 --pmc passes over python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-search), tools/pmc_summary.py",
                  "counters": c, "launches_averaged": n, "derived": der}, indent=1))
