#!/usr/bin/env python
"""PEPR's tree-building step as one command on the GPU engine.

Reads a directory of per-gene protein alignments (FASTA, one file per single-copy family, titles =
taxon names, as PhylogenomicPipeline2 holds them after muscle+Gblocks) and writes what
buildConcatenatedTreeWithGeneWiseJackKnifeSupport writes (PhylogenomicPipeline2.java:994-1126):
    <run>.nwk   full ML tree of the concatenation with integer jackknife supports as node labels
    <run>.sup   the support trees, one Newick per line
usage: pepr_tree_step.py -run_name X -alignment_dir DIR [-support_reps 100] [-seed 1] [-device 0]
(flag style and names follow .../util/HandyConstants.java: run_name, support_reps)
"""
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pepr_amd import engine


def read_fasta(path):
    names, rows = [], []
    for line in open(path):
        line = line.strip()
        if not line:
            continue
        if line.startswith(">"):
            names.append(line[1:].split()[0]); rows.append([])
        else:
            rows[-1].append(line)
    return names, ["".join(r) for r in rows]


def main(argv):
    args = {}
    i = 0
    while i < len(argv):                       # CommandLineProperties style: -flag value
        if argv[i].startswith("-"):
            key = argv[i][1:]; vals = []
            i += 1
            while i < len(argv) and not argv[i].startswith("-"):
                vals.append(argv[i]); i += 1
            args[key] = vals[0] if vals else "true"
        else:
            i += 1
    if "run_name" not in args or "alignment_dir" not in args:
        raise SystemExit(__doc__)
    files = sorted(glob.glob(os.path.join(args["alignment_dir"], "*")))
    genes = [read_fasta(f) for f in files if os.path.isfile(f)]
    genes = [g for g in genes if len(g[0]) >= 3]
    reps = int(args.get("support_reps", 100))
    ctx = engine.Context(int(args.get("device", 0)))
    t0 = time.time()
    r = ctx.jackknife(genes, reps=reps, seed=int(args.get("seed", 1)), spr_radius_full=5)
    dt = time.time() - t0
    run = args["run_name"]
    open(run + ".nwk", "w").write(r["newick"] + "\n")
    open(run + ".sup", "w").write("\n".join(r["support_trees"]) + "\n")
    print("%s: %d genes, %d columns, lnL %.3f, alpha %.4f, %d support trees, %.2f s -> %s.nwk %s.sup" % (
        run, len(genes), r["nsites"], r["lnl"], r["alpha"], reps, dt, run, run))


if __name__ == "__main__":
    main(sys.argv[1:])
