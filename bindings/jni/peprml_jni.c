/*
 * peprml_jni.c -- JNI glue between edu.vt.vbi.ci.pepr.tree.NativeTreeEngine and the C ABI
 * (include/peprml.h).  Pure marshalling: char[][] / String[] -> pml_alignment, results -> Java
 * strings / arrays; failures map to null (the runners' existing behaviour).
 * Build (needs a JDK): make -C bindings/jni   (not built in this repository's image: no jni.h)
 */
#include <jni.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/peprml.h"

static pml_ctx *g_ctx;     /* one engine context per JVM; concurrent single-gene calls are coalesced into device batches by the library */

static int ensure_ctx(void) {
    if (g_ctx) return 0;
    pml_config cfg = {0, 0, 0};
    return pml_create(&cfg, &g_ctx);
}

typedef struct { int n, len; const char **names; char **rows; jstring *jnames; } aln_buf;

static int aln_from_java(JNIEnv *env, jobjectArray taxa, jobjectArray rows, aln_buf *a) {
    a->n = (*env)->GetArrayLength(env, taxa);
    a->len = 0;
    a->names = (const char **)calloc(a->n, sizeof *a->names);
    a->rows = (char **)calloc(a->n, sizeof *a->rows);
    a->jnames = (jstring *)calloc(a->n, sizeof *a->jnames);
    if (!a->names || !a->rows || !a->jnames) return -1;
    for (int i = 0; i < a->n; ++i) {
        a->jnames[i] = (jstring)(*env)->GetObjectArrayElement(env, taxa, i);
        a->names[i] = (*env)->GetStringUTFChars(env, a->jnames[i], NULL);
        jcharArray r = (jcharArray)(*env)->GetObjectArrayElement(env, rows, i);
        const int L = (*env)->GetArrayLength(env, r);
        if (i == 0) a->len = L; else if (L != a->len) return -1;
        jchar *c = (*env)->GetCharArrayElements(env, r, NULL);
        a->rows[i] = (char *)malloc((size_t)L + 1);
        if (!a->rows[i]) return -1;
        for (int s = 0; s < L; ++s) a->rows[i][s] = (char)c[s];
        a->rows[i][L] = 0;
        (*env)->ReleaseCharArrayElements(env, r, c, JNI_ABORT);
    }
    return 0;
}
static void aln_release(JNIEnv *env, aln_buf *a) {
    for (int i = 0; i < a->n; ++i) {
        if (a->names && a->names[i]) (*env)->ReleaseStringUTFChars(env, a->jnames[i], a->names[i]);
        if (a->rows) free(a->rows[i]);
    }
    free(a->names); free(a->rows); free(a->jnames);
}

JNIEXPORT jstring JNICALL Java_edu_vt_vbi_ci_pepr_tree_NativeTreeEngine_search(
        JNIEnv *env, jclass cls, jobjectArray taxa, jobjectArray rows, jstring start, jint nni, jint spr) {
    (void)cls;
    if (ensure_ctx()) return NULL;
    aln_buf a; memset(&a, 0, sizeof a);
    jstring out = NULL;
    if (aln_from_java(env, taxa, rows, &a) == 0) {
        pml_alignment aln = {a.n, a.len, a.names, (const char *const *)a.rows};
        pml_model model = {4, 1.0, PML_PI_RAXML_3DP};
        pml_search_opts opts = {1, nni, spr, 1e-3, 0};
        const char *st = start ? (*env)->GetStringUTFChars(env, start, NULL) : NULL;
        pml_result res;
        if (pml_search(g_ctx, &aln, st, &model, &opts, &res) == PML_OK) out = (*env)->NewStringUTF(env, res.newick);
        pml_result_free(&res);
        if (st) (*env)->ReleaseStringUTFChars(env, start, st);
    }
    aln_release(env, &a);
    return out;
}

JNIEXPORT jstring JNICALL Java_edu_vt_vbi_ci_pepr_tree_NativeTreeEngine_optimize(
        JNIEnv *env, jclass cls, jobjectArray taxa, jobjectArray rows, jstring newick) {
    (void)cls;
    if (ensure_ctx() || !newick) return NULL;
    aln_buf a; memset(&a, 0, sizeof a);
    jstring out = NULL;
    if (aln_from_java(env, taxa, rows, &a) == 0) {
        pml_alignment aln = {a.n, a.len, a.names, (const char *const *)a.rows};
        pml_model model = {4, 1.0, PML_PI_RAXML_3DP};
        pml_search_opts opts = {1, 0, 0, 1e-4, 0};
        const char *nw = (*env)->GetStringUTFChars(env, newick, NULL);
        pml_result res;
        if (pml_optimize(g_ctx, &aln, nw, &model, &opts, &res) == PML_OK) out = (*env)->NewStringUTF(env, res.newick);
        pml_result_free(&res);
        (*env)->ReleaseStringUTFChars(env, newick, nw);
    }
    aln_release(env, &a);
    return out;
}

JNIEXPORT jdoubleArray JNICALL Java_edu_vt_vbi_ci_pepr_tree_NativeTreeEngine_siteLnL(
        JNIEnv *env, jclass cls, jobjectArray taxa, jobjectArray rows, jstring newick) {
    (void)cls;
    if (ensure_ctx() || !newick) return NULL;
    aln_buf a; memset(&a, 0, sizeof a);
    jdoubleArray out = NULL;
    if (aln_from_java(env, taxa, rows, &a) == 0) {
        pml_alignment aln = {a.n, a.len, a.names, (const char *const *)a.rows};
        pml_model model = {4, 1.0, PML_PI_RAXML_3DP};
        pml_search_opts opts = {1, 0, 0, 1e-4, 0};
        const char *nw = (*env)->GetStringUTFChars(env, newick, NULL);
        pml_result o, r;
        if (pml_optimize(g_ctx, &aln, nw, &model, &opts, &o) == PML_OK) {      /* -f g optimises first */
            pml_model m2 = {4, o.alpha, PML_PI_RAXML_3DP};
            if (pml_score(g_ctx, &aln, o.newick, &m2, PML_WANT_SITE_LNL, &r) == PML_OK) {
                out = (*env)->NewDoubleArray(env, r.nsites);
                if (out) (*env)->SetDoubleArrayRegion(env, out, 0, r.nsites, r.site_lnl);
            }
            pml_result_free(&r);
        }
        pml_result_free(&o);
        (*env)->ReleaseStringUTFChars(env, newick, nw);
    }
    aln_release(env, &a);
    return out;
}

JNIEXPORT jobjectArray JNICALL Java_edu_vt_vbi_ci_pepr_tree_NativeTreeEngine_jackknife(
        JNIEnv *env, jclass cls, jobjectArray geneTaxa, jobjectArray geneRows, jint reps, jlong seed) {
    (void)cls;
    if (ensure_ctx()) return NULL;
    const int ng = (*env)->GetArrayLength(env, geneTaxa);
    aln_buf *bufs = (aln_buf *)calloc(ng, sizeof *bufs);
    pml_alignment *alns = (pml_alignment *)calloc(ng, sizeof *alns);
    jobjectArray out = NULL;
    int ok = bufs && alns;
    for (int g = 0; g < ng && ok; ++g) {
        jobjectArray t = (jobjectArray)(*env)->GetObjectArrayElement(env, geneTaxa, g);
        jobjectArray r = (jobjectArray)(*env)->GetObjectArrayElement(env, geneRows, g);
        ok = aln_from_java(env, t, r, &bufs[g]) == 0;
        alns[g].ntax = bufs[g].n; alns[g].nsites = bufs[g].len; alns[g].names = bufs[g].names;
        alns[g].rows = (const char *const *)bufs[g].rows;
    }
    if (ok) {
        pml_model model = {4, 1.0, PML_PI_RAXML_3DP};
        pml_jackknife_opts jo = {reps, 0, (unsigned long long)seed, 5, 1e-3};
        pml_result res; char *sup = NULL;
        if (pml_jackknife(g_ctx, ng, alns, &model, &jo, &res, &sup) == PML_OK) {
            jclass str = (*env)->FindClass(env, "java/lang/String");
            out = (*env)->NewObjectArray(env, reps + 1, str, NULL);
            (*env)->SetObjectArrayElement(env, out, 0, (*env)->NewStringUTF(env, res.newick));
            char *p = sup;
            for (int i = 1; i <= reps && p && *p; ++i) {
                char *nl = strchr(p, '\n'); if (nl) *nl = 0;
                (*env)->SetObjectArrayElement(env, out, i, (*env)->NewStringUTF(env, p));
                p = nl ? nl + 1 : NULL;
            }
        }
        pml_result_free(&res); pml_free(sup);
    }
    for (int g = 0; g < ng && bufs; ++g) aln_release(env, &bufs[g]);
    free(bufs); free(alns);
    return out;
}


/* raxmlHPC -f d -y : parsimony start tree (topology only), RAxMLRunner.java:241-251 */
JNIEXPORT jstring JNICALL Java_edu_vt_vbi_ci_pepr_tree_NativeTreeEngine_parsimony(
        JNIEnv *env, jclass cls, jobjectArray taxa, jobjectArray rows, jint seed) {
    (void)cls;
    if (ensure_ctx()) return NULL;
    aln_buf a; memset(&a, 0, sizeof a);
    jstring out = NULL;
    if (aln_from_java(env, taxa, rows, &a) == 0) {
        pml_alignment aln = {a.n, a.len, a.names, (const char *const *)a.rows};
        pml_parsimony_opts po = {(unsigned)seed, 20};
        pml_result res;
        if (pml_parsimony(g_ctx, &aln, &po, &res, NULL) == PML_OK) out = (*env)->NewStringUTF(env, res.newick);
        pml_result_free(&res);
    }
    aln_release(env, &a);
    return out;
}

/* raxmlHPC -f a -x seed -N reps : best tree with percent supports (RAxML_bipartitions.<run>), RAxMLRunner.java:115-132,302-318 */
JNIEXPORT jstring JNICALL Java_edu_vt_vbi_ci_pepr_tree_NativeTreeEngine_bootstrap(
        JNIEnv *env, jclass cls, jobjectArray taxa, jobjectArray rows, jint reps, jlong seed) {
    (void)cls;
    if (ensure_ctx()) return NULL;
    aln_buf a; memset(&a, 0, sizeof a);
    jstring out = NULL;
    if (aln_from_java(env, taxa, rows, &a) == 0) {
        pml_alignment aln = {a.n, a.len, a.names, (const char *const *)a.rows};
        pml_model model = {4, 1.0, PML_PI_RAXML_3DP};
        pml_result res;
        if (pml_bootstrap(g_ctx, &aln, &model, reps, (unsigned long long)seed, 5, 1e-3, &res, NULL) == PML_OK) out = (*env)->NewStringUTF(env, res.newick);
        pml_result_free(&res);
    }
    aln_release(env, &a);
    return out;
}

/* FastTree_WAG -gamma without -nosupport : SH-like local supports (0-1) on a given tree, FastTreeRunner.java:67-70 */
JNIEXPORT jstring JNICALL Java_edu_vt_vbi_ci_pepr_tree_NativeTreeEngine_shSupport(
        JNIEnv *env, jclass cls, jobjectArray taxa, jobjectArray rows, jstring newick, jdouble alpha) {
    (void)cls;
    if (ensure_ctx() || !newick) return NULL;
    aln_buf a; memset(&a, 0, sizeof a);
    jstring out = NULL;
    if (aln_from_java(env, taxa, rows, &a) == 0) {
        pml_alignment aln = {a.n, a.len, a.names, (const char *const *)a.rows};
        pml_model model = {4, alpha, PML_PI_WAG_FULL};
        const char *nw = (*env)->GetStringUTFChars(env, newick, NULL);
        pml_result res;
        if (pml_sh_support(g_ctx, &aln, nw, &model, 1000, 314159ULL, &res) == PML_OK) out = (*env)->NewStringUTF(env, res.newick);
        pml_result_free(&res);
        (*env)->ReleaseStringUTFChars(env, newick, nw);
    }
    aln_release(env, &a);
    return out;
}
