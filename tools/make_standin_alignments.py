"""Builds the stand-in gene alignments for BASELINE configs C1/C2 (bundled Erysipelotrichales /
Aquificales examples) -> tests/golden/standin_<set>.json.

The stock pipeline's alignments cannot be regenerated here (blastall, muscle, Gblocks and Java are
unavailable: SURVEY.md section 8c "Bundled-data parity"), so, as SURVEY proposes, single-copy
families are picked by PATRIC product annotation: a product name that occurs exactly once in
EVERY genome of the set.  muscle/Gblocks do not run here, so each family is aligned by a small
centre-star aligner in this script (global Needleman-Wunsch of every member against the member
of median length, linear gap cost, +2/-1 scores; "once a gap, always a gap" merge) and trimmed
to its gap-free columns (the Gblocks role); families whose lengths differ by more than 12 % or
whose trimmed alignment has mean pairwise identity < 0.45 are dropped.  These are STAND-INS for
the reference's muscle+Gblocks alignments, not reproductions of them.
Run in the build container (needs /root/reference):  python tools/make_standin_alignments.py
"""
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = "/root/reference/examples"


def read_faa(path):
    out, name, seq = [], None, []
    for line in open(path):
        line = line.rstrip()
        if line.startswith(">"):
            if name is not None:
                out.append((name, "".join(seq)))
            name, seq = line[1:], []
        else:
            seq.append(line)
    if name is not None:
        out.append((name, "".join(seq)))
    return out


def product(title):
    m = re.match(r"fid\|[^|]*\|locus\|[^|]*\|\s*(.*?)\s*\[[^\]]*\]\s*$", title)
    p = m.group(1) if m else title
    return re.sub(r"\s+", " ", p).strip()


def taxon(path):
    # FastaUtilities (reference FastaUtilities.java:24-115): taxon = genome name; forbidden chars -> '_'
    return re.sub(r"[,():\s]", "_", os.path.basename(path).replace(".PATRIC.faa", ""))


def nw_align(a, b, match=2.0, mismatch=-1.0, gap=2.0):
    """Global alignment (linear gaps), row-vectorised; returns the two gapped strings."""
    import numpy as np
    A = np.frombuffer(a.encode(), dtype=np.uint8); B = np.frombuffer(b.encode(), dtype=np.uint8)
    n, m = len(A), len(B)
    H = np.zeros((n + 1, m + 1)); idx = np.arange(m + 1) * gap
    H[0] = -idx
    for i in range(1, n + 1):
        sub = np.where(B == A[i - 1], match, mismatch)
        V = np.empty(m + 1)
        V[0] = H[i - 1, 0] - gap
        V[1:] = np.maximum(H[i - 1, :-1] + sub, H[i - 1, 1:] - gap)
        H[i] = np.maximum.accumulate(V + idx) - idx        # horizontal gaps: max_k (V[k] - gap (j-k))
    i, j, ra, rb = n, m, [], []
    while i > 0 or j > 0:
        if i > 0 and j > 0 and abs(H[i, j] - (H[i - 1, j - 1] + (match if A[i - 1] == B[j - 1] else mismatch))) < 1e-9:
            ra.append(a[i - 1]); rb.append(b[j - 1]); i -= 1; j -= 1
        elif i > 0 and abs(H[i, j] - (H[i - 1, j] - gap)) < 1e-9:
            ra.append(a[i - 1]); rb.append("-"); i -= 1
        else:
            ra.append("-"); rb.append(b[j - 1]); j -= 1
    return "".join(reversed(ra)), "".join(reversed(rb))


def star_align(seqs):
    """Centre-star MSA; returns rows of equal length (with '-')."""
    order = sorted(range(len(seqs)), key=lambda k: len(seqs[k]))
    c = order[len(order) // 2]
    center = seqs[c]
    # gaps inserted into the centre by each pairwise alignment: gaps_before[pos] = max count
    pair = {}
    gaps = [0] * (len(center) + 1)
    for k, sq in enumerate(seqs):
        if k == c:
            continue
        ca, sa = nw_align(center, sq)
        pair[k] = (ca, sa)
        pos = 0; run = 0
        for ch in ca:
            if ch == "-":
                run += 1
            else:
                gaps[pos] = max(gaps[pos], run); run = 0; pos += 1
        gaps[pos] = max(gaps[pos], run)
    rows = [None] * len(seqs)
    def expand(ca, sa):
        out = []; pos = 0; run = 0; buf = []
        for x, y in zip(ca, sa):
            if x == "-":
                buf.append(y); run += 1
            else:
                out.append("".join(buf) + "-" * (gaps[pos] - run)); buf = []; run = 0
                out.append(y); pos += 1
        out.append("".join(buf) + "-" * (gaps[pos] - run))
        return "".join(out)
    rows[c] = expand(center, center)
    for k, (ca, sa) in pair.items():
        rows[k] = expand(ca, sa)
    assert len(set(map(len, rows))) == 1
    return rows


def build(setname, max_families=60):
    files = sorted(glob.glob(os.path.join(EX, setname, "*.faa"))) + sorted(glob.glob(os.path.join(EX, setname, "outgroup", "*.faa")))
    genomes = {taxon(f): read_faa(f) for f in files}
    per = {}
    for t, prots in genomes.items():
        d = defaultdict(list)
        for title, seq in prots:
            d[product(title)].append(seq)
        per[t] = d
    taxa = sorted(genomes)
    fams = []
    allprod = sorted(set().union(*[set(per[t]) for t in taxa]))
    for prod in allprod:
        if "hypothetical" in prod.lower() or not prod:
            continue
        # members = genomes with exactly one copy; genomes with 0 or >1 copies are absent from the
        # family (the concatenation pads them with '?', MSAConcatenator.java:164-170)
        members = [t for t in taxa if len(per[t].get(prod, [])) == 1]
        if len(members) < max(4, int(0.8 * len(taxa) + 0.999)):
            continue
        seqs = [per[t][prod][0].rstrip("*") for t in members]
        lens = list(map(len, seqs))
        if min(lens) < 60 or max(lens) > 1.12 * min(lens) or max(lens) > 700:
            continue
        rows = star_align(seqs)
        keep = [j for j in range(len(rows[0])) if all(r[j] != "-" for r in rows)]
        seqs = ["".join(r[j] for j in keep) for r in rows]
        if len(keep) < 50:
            continue
        L = len(seqs[0]); ident = []
        for i in range(len(seqs)):
            for j in range(i + 1, len(seqs)):
                ident.append(sum(a == b for a, b in zip(seqs[i], seqs[j])) / L)
        if sum(ident) / len(ident) < 0.45:
            continue
        fams.append({"product": prod, "names": members, "rows": seqs})
    fams = fams[:max_families]
    out = {"set": setname, "taxa": taxa, "outgroup": [taxon(f) for f in files if "/outgroup/" in f],
           "note": "stand-in alignments (equal-length single-copy families by PATRIC product annotation), see tools/make_standin_alignments.py",
           "genes": fams}
    path = os.path.join(ROOT, "tests", "golden", "standin_%s.json" % setname)
    json.dump(out, open(path, "w"))
    print(setname, "taxa", len(taxa), "families", len(fams), "columns", sum(len(f["rows"][0]) for f in fams), "->", path, os.path.getsize(path) // 1024, "KB")
    for f in fams[:8]:
        print("   ", len(f["rows"][0]), f["product"])


if __name__ == "__main__":
    for s in ("Aquificales", "Erysipelotrichales"):
        build(s)
