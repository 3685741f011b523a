"""Test harness: PEPR's progressive-refinement loop (PhylogeneticTreeRefiner.refine, PhylogeneticTreeRefiner.java:81-275)
driven around the engine.  The loop itself stays in Java in the target system (BASELINE north star); this Python restatement
exists so that BASELINE configs[1] ("Aquificales, 1 refinement round") can be driven end to end in a test:

    tree with supports -> root on the outgroup -> pml_refine_next picks the clade -> the tree-building step (pml_jackknife)
    re-run on that clade's genomes plus <= 2 genomes of the sister clade as outgroup (:160-196) -> subtree rooted on them
    -> grafted in place of the clade (:246) -> repeat until no node qualifies.

Host-side tree surgery only; every likelihood comes from the engine.  In PEPR the recursive run redoes homology search and
alignment for the clade's genomes; here the stand-in gene families are restricted to the clade's taxa instead."""
import re

import util


class Node:
    def __init__(self, name="", length=0.0, support=None):
        self.name, self.length, self.support, self.kids = name, length, support, []

    def leaves(self):
        return [self.name] if not self.kids else [x for k in self.kids for x in k.leaves()]


def parse(nw):
    """Newick -> Node tree; an inner node's integer label is the support of the branch above it."""
    def conv(t):
        kids, name, length = t
        n = Node(name if not kids else "", length, int(name) if kids and re.fullmatch(r"\d+", name or "") else None)
        n.kids = [conv(k) for k in kids]
        return n
    return conv(util.parse_newick(nw))


def fmt(n, top=True):
    s = ("(" + ",".join(fmt(k, False) for k in n.kids) + ")" + (str(n.support) if n.support is not None and not top else "")) if n.kids else n.name
    return s + (";" if top else ":%.8f" % n.length)


def root_on(nw, outgroup):
    """AdvancedTree.setOutGroup: the tree re-rooted on the branch that separates the outgroup taxa from the rest (the
    branch whose far side holds all outgroup taxa and the fewest others); edge supports stay with their edges."""
    t = parse(nw)
    # undirected graph: node ids, edges {(a,b): (length, support)}
    nodes, adj, edge = [], {}, {}

    def walk(n, parent):
        i = len(nodes); nodes.append(n); adj[i] = []
        if parent is not None:
            adj[i].append(parent); adj[parent].append(i)
            edge[frozenset((i, parent))] = (n.length, n.support)
        for k in n.kids:
            walk(k, i)
    walk(t, None)
    if len(adj[0]) == 2:                      # rooted input: merge the two root edges
        a, b = adj[0]
        la, sa = edge.pop(frozenset((0, a))); lb, sb = edge.pop(frozenset((0, b)))
        adj[a].remove(0); adj[b].remove(0); adj[a].append(b); adj[b].append(a); adj[0] = []
        edge[frozenset((a, b))] = (la + lb, sa if sa is not None else sb)
    og = set(outgroup)

    def side(a, b):                           # leaves reached from a without crossing to b
        out, st = [], [(a, b)]
        while st:
            x, f = st.pop()
            if not nodes[x].kids and nodes[x].name:
                out.append(nodes[x].name)
            st += [(y, x) for y in adj[x] if y != f]
        return out
    best = None
    for e in edge:
        a, b = tuple(e)
        for x, y in ((a, b), (b, a)):
            s = side(x, y)
            if og <= set(s) and (best is None or len(s) < best[0]):
                best = (len(s), x, y)
    _, x, y = best

    def build(v, f, length, support):
        n = Node(nodes[v].name if not nodes[v].kids else "", length, support if nodes[v].kids else None)
        for w in adj[v]:
            if w != f:
                l, s = edge[frozenset((v, w))]
                n.kids.append(build(w, v, l, s))
        return n
    l, s = edge[frozenset((x, y))]
    root = Node()
    root.kids = [build(y, x, 0.5 * l, s), build(x, y, 0.5 * l, s)]      # ingroup first, outgroup last
    return root


def find_clade(n, members):
    if set(n.leaves()) == set(members):
        return n
    for k in n.kids:
        r = find_clade(k, members)
        if r is not None:
            return r
    return None


def parent_of(root, node):
    for k in root.kids:
        if k is node:
            return root
        r = parent_of(k, node)
        if r is not None:
            return r
    return None


def refine(ctx, genes, outgroup, reps=100, cutoff=100, seed=1, max_rounds=5, log=print):
    """Returns (final rooted Newick with supports, list of refined clades).  genes: [(names, rows)] gene families."""
    from pepr_amd import engine
    first = ctx.jackknife(genes, reps=reps, seed=seed, spr_radius_full=5)
    tree = root_on(first["newick"], outgroup)
    done, rounds = [], []
    for rnd in range(1, max_rounds + 1):
        ingroup, _ = engine.refine_next(fmt(tree), cutoff, done)
        if not ingroup:
            break
        node = find_clade(tree, ingroup)
        par = parent_of(tree, node)
        pool = sorted(set(par.leaves()) - set(ingroup)) if par is not None else []     # :160-178 sister clade = outgroup pool
        sub_out = pool[:min(len(pool), 2)]                                              # outgroup_count = min(pool, 2), :218
        keep = set(ingroup) | set(sub_out)
        sub_genes = []
        for names, rows in genes:
            idx = [i for i, t in enumerate(names) if t in keep]
            if len(idx) >= 4 and sum(names[i] in ingroup for i in idx) >= 3:
                sub_genes.append(([names[i] for i in idx], [rows[i] for i in idx]))
        r = ctx.jackknife(sub_genes, reps=reps, seed=seed + rnd, spr_radius_full=5)
        sub = root_on(r["newick"], sub_out) if sub_out else parse(r["newick"])
        new = find_clade(sub, ingroup)
        assert new is not None, "refined subtree does not keep the ingroup together"
        new.length, new.support = node.length, node.support           # the branch above the clade belongs to the outer tree
        if par is None:
            tree = new
        else:
            par.kids[[k is node for k in par.kids].index(True)] = new
        done.append(sorted(ingroup)); rounds.append({"ingroup": sorted(ingroup), "outgroup": sub_out, "genes": len(sub_genes), "subtree": fmt(sub)})
        log("refinement round %d: clade of %d taxa, outgroup %s, %d gene families" % (rnd, len(ingroup), sub_out, len(sub_genes)))
    return fmt(tree), rounds, first
