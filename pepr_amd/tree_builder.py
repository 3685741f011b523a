"""Host-side mirror of the reference's tree-building interface for the hot path.

Same names, argument meaning and error behaviour as the Java classes, so that the parity tests
read like the reference's own call sites; every `run()` goes through the C ABI
(include/peprml.h) to the HIP engine instead of spawning a process:

  PhylogeneticTreeBuilder  <- .../pepr/tree/pipeline/PhylogeneticTreeBuilder.java:97-129,168-196,215-338
  RAxMLRunner              <- .../pepr/tree/RAxMLRunner.java:64-152,162-213,320-336
  FastTreeRunner           <- .../pepr/tree/FastTreeRunner.java:38-135,142-199,235

Behaviour kept from the reference: a failed build leaves the result `None` (FastTreeRunner.java:
125-131 logs and continues; callers see a null tree string); the ML matrix is a RAxML model
string (default PROTGAMMAWAG, PhylogenomicPipeline2.java:248-250); threads/processes are accepted
and ignored (the GPU engine needs no -T).  Not mirrored (out of scope): parsimony
bootstrap (-Y), nucleotide (-gtr -nt).
"""
import logging

from . import engine

ML, FAST_TREE, PARSIMONY, PARSIMONY_BL, NEIGHBOR_JOINING = "ml", "FastTree", "parsimony", "parsimony_bl", "nj"
log = logging.getLogger("pepr_amd")
_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = engine.Context(0)
    return _default_ctx


class SequenceAlignment:
    """Minimal stand-in for .../pepr/alignment/SequenceAlignment.java: taxa + aligned char rows."""

    def __init__(self, taxa, rows):
        self.taxa = list(taxa)
        self.rows = list(rows)

    def getTaxa(self):
        return self.taxa

    def getLength(self):
        return len(self.rows[0]) if self.rows else 0

    def as_gene(self):
        return (self.taxa, self.rows)


def _model_from_matrix(matrix):
    """RAxML -m strings PEPR passes (RAxMLRunner.java:115-132; the 23 names of -matrix_eval,
    PhylogenomicPipeline2.java:260-284).  TWO models are built -- PROTGAMMAWAG (WAG exchangeabilities, RAxML's
    3-decimal frequencies, Gamma4) and PROTGAMMAWAGF (the same with frequencies counted from the alignment) -- and a
    likelihood is never reported under another model's name: PROTCATWAG, PROTGAMMAIWAG and the other matrices (whose
    tables the reference does not hold) raise."""
    m = (matrix or "PROTGAMMAWAG").upper()
    if m == "PROTGAMMAWAGF":
        return {"ncat": 4, "pi_mode": engine.PI_EMPIRICAL}
    if m != "PROTGAMMAWAG":
        raise ValueError("the GPU engine implements PROTGAMMAWAG and PROTGAMMAWAGF only; %r is not built (it would be a "
                         "different likelihood function, not a variant spelling)" % matrix)
    return {"ncat": 4, "pi_mode": engine.PI_RAXML_3DP}


class RAxMLRunner:
    def __init__(self, threads=1, ctx=None):
        self.threads = threads
        self.ctx = ctx
        self.alignment = None
        self.matrix = "PROTGAMMAWAG"
        self.bootstrapReps = 0
        self.perSiteLL = False
        self.perSiteLLTrees = None
        self.bestTree = None
        self.perSiteLLs = None
        self.lnl = None
        self.alpha = None
        self.spr_radius = 5           # RAxML "best rearrangement setting 5" 
        self.algorithm = 0            # 1 = parsimony only, 2 = parsimony + ML lengths (RAxMLRunner.java:28-29)
        self.parsimonyTree = None
        self.parsimonyWithBLTree = None
        self.bestTreeWithSupports = None
        self.seed = 12345

    def setAlignment(self, a):
        self.alignment = a

    def getAlignment(self):
        return self.alignment

    def setMatrix(self, m):
        self.matrix = m

    def setBootstrapReps(self, reps):
        self.bootstrapReps = reps

    def setUseTaxonNames(self, b):
        pass

    def setParsimonyOnly(self, b):          # RAxMLRunner.java:542-548
        if b:
            self.algorithm = 1

    def setParsimonyWithBL(self, b):        # RAxMLRunner.java:550-556
        if b:
            self.algorithm = 2

    def getParsimonyTree(self):             # RAxML_parsimonyTree.<run>, RAxMLRunner.java:338-359
        return self.parsimonyTree

    def getParsimonyWithBLTree(self):       # RAxML_result.<run>BL, RAxMLRunner.java:361-383
        return self.parsimonyWithBLTree

    def setPerSiteLogLikelihoods(self, b):
        self.perSiteLL = b

    def setPerSiteLLTrees(self, trees):
        self.perSiteLLTrees = list(trees)

    def run(self):
        """-f d (ML search) or, with setPerSiteLogLikelihoods(true), -f g on the given trees."""
        mdl = _model_from_matrix(self.matrix)          # an unbuilt model name is refused, loudly, before any device work
        ctx = self.ctx or default_context()
        try:
            gene = self.alignment.as_gene()
            if self.perSiteLL:
                self.perSiteLLs = []
                for nw in self.perSiteLLTrees:
                    o = ctx.optimize([gene], [nw], **mdl)[0]        # RAxML -f g optimises model + lengths first
                    r = ctx.score([gene], [o["newick"]], alpha=o["alpha"], site_lnl=True, **mdl)[0]
                    self.perSiteLLs.append(r["site_lnl"])
                return
            if self.algorithm in (1, 2):
                if self.bootstrapReps:
                    raise ValueError("parsimony bootstrap (-Y -N) is not on the GPU path")
                p = ctx.parsimony([gene], seed=self.seed)[0]                  # -f d -y
                self.parsimonyTree = p["newick"]
                if self.algorithm == 2:                                        # -f e -t RAxML_parsimonyTree.<run>
                    o = ctx.optimize([gene], [p["newick"]], **mdl)[0]
                    self.parsimonyWithBLTree, self.lnl, self.alpha = o["newick"], o["lnl"], o["alpha"]
                return
            if self.bootstrapReps:                                                # -f a -x <odd seed> -N reps
                r = ctx.bootstrap(gene, reps=self.bootstrapReps, seed=self.seed | 1, spr_radius=self.spr_radius, **mdl)
                self.bestTreeWithSupports, self.lnl, self.alpha = r["newick"], r["lnl"], r["alpha"]
                self.bestTree = None                                              # -f a writes no RAxML_result
                return
            r = ctx.search([gene], None, spr_radius=self.spr_radius, seed=self.seed, **mdl)[0]      # parsimony start, as -f d
            self.bestTree, self.lnl, self.alpha = r["newick"], r["lnl"], r["alpha"]
        except Exception as e:          # reference: rc logged, result stays null
            log.error("RAxMLRunner failed: %s", e)
            self.bestTree = self.parsimonyTree = self.parsimonyWithBLTree = None

    def getBestTree(self):
        return self.bestTree

    def getBestTreeWithSupports(self):      # RAxML_bipartitions.<run>, RAxMLRunner.java:302-318
        return self.bestTreeWithSupports

    def getPerSiteLLs(self):
        return self.perSiteLLs


class FastTreeRunner:
    def __init__(self, ctx=None):
        self.ctx = ctx
        self.alignment = None
        self.result = None
        self.useRaxmlBranchLengths = False
        self.bootstrapReps = 0
        self.lnl = None
        self.constraints = None

    def setAlignment(self, a):
        self.alignment = a

    def getAlignment(self):
        return self.alignment

    def setConstraints(self, fasta_text):
        """FASTA text of 0/1/- rows, one column per constrained split (FastTreeRunner.java:220-222)."""
        self.constraints = fasta_text

    def setConstraintTree(self, treeString):
        """One column per node of the tree: 1 = leaf below the node (FastTreeRunner.java:224-229,243-273)."""
        if treeString is not None:
            names, rows = engine.constraints_from_tree(treeString)
            self.setConstraints("".join(">%s\n%s\n" % (n, r) for n, r in zip(names, rows)))

    def _constraint_matrix(self):
        if self.constraints is None:
            return None
        names, rows = [], []
        for line in self.constraints.splitlines():
            line = line.strip()
            if line.startswith(">"):
                names.append(line[1:]); rows.append("")
            elif line and names:
                rows[-1] += line
        return names, rows

    def setUseRaxmlBranchLengths(self, b):
        self.useRaxmlBranchLengths = b

    def getUseRaxmlBranchLengths(self):
        return self.useRaxmlBranchLengths

    def setBootstrapReps(self, reps):
        self.bootstrapReps = reps

    def setRunName(self, n):
        self.runName = n

    def setThreadCount(self, n):
        pass

    def setNucleoide(self, nuc):
        if nuc:
            raise ValueError("nucleotide models are not on the GPU path")

    def run(self):
        """`FastTree_WAG -gamma [-nosupport]` (FastTreeRunner.java:67-86): NJ start + NNI hill climbing under WAG+Gamma, then
        FastTree's Gamma20 step -- the tree PEPR reads from stdout carries the lengths multiplied by the fitted rescale."""
        import re
        ctx = self.ctx or default_context()
        try:
            # FastTree_WAG carries the full-precision WAG frequencies, RAxML the 3-decimal ones
            gene = self.alignment.as_gene()
            r = ctx.search([gene], None, nni=True, spr_radius=0, pi_mode=engine.PI_WAG_FULL,
                           constraints=self._constraint_matrix())[0]
            g20 = ctx.gamma20([gene], [r["newick"]], pi_mode=engine.PI_WAG_FULL)[0]
            self.lnl, self.gamma20LogLk, self.gamma20Alpha, self.rescale = r["lnl"], g20["lnl"], g20["alpha"], g20["rescale"]
            self.result = g20["newick"]
            if self.bootstrapReps > 0 and len(self.alignment.getTaxa()) > 3:   # FastTreeRunner.java:67-70: no -nosupport -> SH-like supports
                s = ctx.sh_support([gene], [r["newick"]], alpha=r["alpha"], pi_mode=engine.PI_WAG_FULL)[0]
                self.result = re.sub(r":([0-9.eE+-]+)", lambda m: ":%.10f" % (float(m.group(1)) * g20["rescale"]), s["newick"])
        except Exception as e:
            log.error("FastTreeRunner failed: %s", e)
            self.result = None

    def getResult(self):
        return self.result


class PhylogeneticTreeBuilder:
    def __init__(self, ctx=None):
        self.ctx = ctx
        self.alignment = None
        self.treeBuildingMethod = ML
        self.mlMatrix = "PROTGAMMAWAG"
        self.treeString = None
        self.processes = 1
        self.bootstrapReps = 0
        self.runName = None
        self._useRaxmlBL = False
        self.constraintTree = None

    def setAlignment(self, a):
        self.alignment = a

    def getAlignment(self):
        return self.alignment

    def setTreeBuildingMethod(self, m):
        self.treeBuildingMethod = m

    def setMLMatrix(self, m):
        self.mlMatrix = m

    def setProcesses(self, n):
        self.processes = n

    def setBootstrapReps(self, reps):
        self.bootstrapReps = reps

    def getBootstrapReps(self):
        return self.bootstrapReps

    def setRunName(self, n):
        self.runName = n

    def getRunName(self):
        return self.runName

    def useRaxmlBranchLengths(self, b):
        self._useRaxmlBL = b

    def setConstraintTree(self, t):
        self.constraintTree = t

    def setNucleotide(self, nuc):
        if nuc:
            raise ValueError("nucleotide models are not on the GPU path")

    def setTreeString(self, s):
        self.treeString = s

    def getTreeString(self):
        return self.treeString

    def run(self):
        if self.treeBuildingMethod == ML:
            r = RAxMLRunner(self.processes, self.ctx)
            r.setBootstrapReps(self.bootstrapReps); r.setAlignment(self.alignment); r.setMatrix(self.mlMatrix)
            r.run()
            self.setTreeString(r.getBestTree() if self.bootstrapReps == 0 else r.getBestTreeWithSupports())
        elif self.treeBuildingMethod in (PARSIMONY, PARSIMONY_BL):        # PhylogeneticTreeBuilder.java:136-161
            r = RAxMLRunner(self.processes, self.ctx)
            if self.treeBuildingMethod == PARSIMONY:
                r.setParsimonyOnly(True)
            else:
                r.setParsimonyWithBL(True)
            r.setBootstrapReps(self.bootstrapReps); r.setAlignment(self.alignment)
            r.run()
            self.setTreeString(r.getParsimonyTree() if self.treeBuildingMethod == PARSIMONY else r.getParsimonyWithBLTree())
        elif self.treeBuildingMethod == FAST_TREE:
            f = FastTreeRunner(self.ctx)
            f.setAlignment(self.alignment); f.setRunName(self.runName); f.setBootstrapReps(self.bootstrapReps)
            f.setUseRaxmlBranchLengths(self._useRaxmlBL)
            if self.constraintTree is not None:             # PhylogeneticTreeBuilder.java:190-192
                f.setConstraintTree(self.constraintTree)
            f.run()
            self.setTreeString(f.getResult())
        else:
            raise ValueError("tree building method %r is outside the GPU path (ml, FastTree, parsimony, parsimony_bl)" % self.treeBuildingMethod)
