"""first divergence between the value sequences (PML_DET_LOG) of the same gene in different batches: python tools/dbg_detlog_diff.py log"""
import collections, sys
seq = collections.defaultdict(list)
for line in open(sys.argv[1]):
    p = line.split(None, 2)
    if len(p) == 3: seq[(p[1], p[0])].append(p[2].strip())
by_gene = collections.defaultdict(list)
for (g, b), s in seq.items(): by_gene[g].append((b, s))
for g, runs in by_gene.items():
    ref_b, ref = runs[0]
    for b, s in runs[1:]:
        if s == ref: continue
        k = next((i for i, (x, y) in enumerate(zip(ref, s)) if x != y), min(len(ref), len(s)))
        print("gene %s: batch %s vs %s diverge at entry %d of %d/%d" % (g, ref_b, b, k, len(ref), len(s)))
        for i in range(max(0, k - 2), min(k + 3, max(len(ref), len(s)))):
            print("   %4d  %-70s | %s" % (i, ref[i] if i < len(ref) else "-", s[i] if i < len(s) else "-"))
        break
print("genes", len(by_gene), "sequences", len(seq))
