package edu.vt.vbi.ci.pepr.tree;

/**
 * JNI face of libpeprml.so (include/peprml.h) for PEPR.  Drop this class into
 * src/edu/vt/vbi/ci/pepr/tree/ and replace the exec blocks of FastTreeRunner.run()
 * (FastTreeRunner.java:44-124) and RAxMLRunner.run() (RAxMLRunner.java:98-151) with calls to it
 * (see INTEGRATION.md).  Every method returns null on failure, which is what the runners already
 * produce when the external tool fails (FastTreeRunner.java:125-131).
 *
 * Not compiled in this repository's build image (no JDK / jni.h there): `make -C bindings/jni`
 * builds libpeprml_jni.so when JAVA_HOME is set.
 */
public final class NativeTreeEngine {
    static { System.loadLibrary("peprml_jni"); }

    private NativeTreeEngine() {}

    /** NJ start (or startNewick) + NNI (+ lazy SPR if sprRadius &gt; 0) under WAG+G4; Newick or null. */
    public static native String search(String[] taxa, char[][] rows, String startNewick, int nni, int sprRadius);

    /** raxmlHPC -f e: branch lengths + alpha on a fixed topology; Newick (20-digit lengths) or null. */
    public static native String optimize(String[] taxa, char[][] rows, String newick);

    /** raxmlHPC -f g: per-site log likelihoods of a tree (lengths + alpha optimised first); null on failure. */
    public static native double[] siteLnL(String[] taxa, char[][] rows, String newick);

    /**
     * PhylogenomicPipeline2.buildConcatenatedTreeWithGeneWiseJackKnifeSupport in one call:
     * result[0] = full tree with integer support labels, result[1..reps] = support trees.
     */
    public static native String[] jackknife(String[][] geneTaxa, char[][][] geneRows, int reps, long seed);

    /** raxmlHPC -f d -y: randomised stepwise-addition parsimony tree (topology only); seed 0 = input order. */
    public static native String parsimony(String[] taxa, char[][] rows, int seed);

    /** raxmlHPC -f a -x seed -N reps: best ML tree with percent bootstrap supports as inner labels. */
    public static native String bootstrap(String[] taxa, char[][] rows, int reps, long seed);

    /** FastTree_WAG -gamma without -nosupport: SH-like local supports (0-1 labels) on the given tree. */
    public static native String shSupport(String[] taxa, char[][] rows, String newick, double alpha);
}
