/*
 * jni_double.c -- drives bindings/jni/peprml_jni.c without a JVM (test infrastructure; see jni.h beside it).
 * Implements the JNIEnv entries the glue uses over plain C objects, builds the String[] / char[][] arguments a
 * Java caller would pass (FastTreeRunner / RAxMLRunner hand SequenceAlignment.getTaxa() and getAlignedSequenceChars(),
 * SequenceAlignment.java:61), calls the Java_edu_vt_vbi_ci_pepr_tree_NativeTreeEngine_* entry points and hands the
 * results back as C data.  It also counts Get / Release pairs: the glue must release what it pins.
 */
#include <jni.h>
#include <stdlib.h>
#include <string.h>

enum { K_STRING = 1, K_CHARS, K_OBJECTS, K_DOUBLES, K_CLASS };
struct jd_object { int kind; int n; char *utf; jchar *chars; struct jd_object **elems; double *doubles; };
static long g_pins, g_unpins;

static jobject new_obj(int kind, int n) { jobject o = (jobject)calloc(1, sizeof *o); o->kind = kind; o->n = n; return o; }
static jclass d_FindClass(JNIEnv *e, const char *name) { (void)e; jobject o = new_obj(K_CLASS, 0); o->utf = strdup(name); return o; }
static jsize d_GetArrayLength(JNIEnv *e, jarray a) { (void)e; return a->n; }
static jchar *d_GetCharArrayElements(JNIEnv *e, jcharArray a, jboolean *copy) { (void)e; if (copy) *copy = 0; ++g_pins; return a->chars; }
static jobject d_GetObjectArrayElement(JNIEnv *e, jobjectArray a, jsize i) { (void)e; return a->elems[i]; }
static const char *d_GetStringUTFChars(JNIEnv *e, jstring s, jboolean *copy) { (void)e; if (copy) *copy = 0; ++g_pins; return s->utf; }
static jdoubleArray d_NewDoubleArray(JNIEnv *e, jsize n) { (void)e; jobject o = new_obj(K_DOUBLES, n); o->doubles = (double *)calloc(n > 0 ? n : 1, sizeof(double)); return o; }
static jobjectArray d_NewObjectArray(JNIEnv *e, jsize n, jclass c, jobject init) { (void)e; (void)c; jobject o = new_obj(K_OBJECTS, n); o->elems = (jobject *)calloc(n > 0 ? n : 1, sizeof(jobject)); for (int i = 0; i < n; ++i) o->elems[i] = init; return o; }
static jstring d_NewStringUTF(JNIEnv *e, const char *s) { (void)e; if (!s) return NULL; jobject o = new_obj(K_STRING, (int)strlen(s)); o->utf = strdup(s); return o; }
static void d_ReleaseCharArrayElements(JNIEnv *e, jcharArray a, jchar *p, jint mode) { (void)e; (void)a; (void)p; (void)mode; ++g_unpins; }
static void d_ReleaseStringUTFChars(JNIEnv *e, jstring s, const char *p) { (void)e; (void)s; (void)p; ++g_unpins; }
static void d_SetDoubleArrayRegion(JNIEnv *e, jdoubleArray a, jsize off, jsize n, const jdouble *src) { (void)e; memcpy(a->doubles + off, src, (size_t)n * sizeof(double)); }
static void d_SetObjectArrayElement(JNIEnv *e, jobjectArray a, jsize i, jobject v) { (void)e; a->elems[i] = v; }

static const struct JNINativeInterface_ g_table = {
    d_FindClass, d_GetArrayLength, d_GetCharArrayElements, d_GetObjectArrayElement, d_GetStringUTFChars, d_NewDoubleArray,
    d_NewObjectArray, d_NewStringUTF, d_ReleaseCharArrayElements, d_ReleaseStringUTFChars, d_SetDoubleArrayRegion, d_SetObjectArrayElement};
static JNIEnv g_env = &g_table;

/* the entry points of the glue */
jstring Java_edu_vt_vbi_ci_pepr_tree_NativeTreeEngine_search(JNIEnv *, jclass, jobjectArray, jobjectArray, jstring, jint, jint);
jstring Java_edu_vt_vbi_ci_pepr_tree_NativeTreeEngine_optimize(JNIEnv *, jclass, jobjectArray, jobjectArray, jstring);
jdoubleArray Java_edu_vt_vbi_ci_pepr_tree_NativeTreeEngine_siteLnL(JNIEnv *, jclass, jobjectArray, jobjectArray, jstring);
jobjectArray Java_edu_vt_vbi_ci_pepr_tree_NativeTreeEngine_jackknife(JNIEnv *, jclass, jobjectArray, jobjectArray, jint, jlong);
jstring Java_edu_vt_vbi_ci_pepr_tree_NativeTreeEngine_parsimony(JNIEnv *, jclass, jobjectArray, jobjectArray, jint);

static jobjectArray java_strings(int n, const char *const *s) {
    jobjectArray a = d_NewObjectArray(&g_env, n, NULL, NULL);
    for (int i = 0; i < n; ++i) a->elems[i] = d_NewStringUTF(&g_env, s[i]);
    return a;
}
static jobjectArray java_char_rows(int n, const char *const *rows) {         /* char[][]: UTF-16 units, as the JVM holds them */
    jobjectArray a = d_NewObjectArray(&g_env, n, NULL, NULL);
    for (int i = 0; i < n; ++i) {
        const int L = (int)strlen(rows[i]);
        jobject r = new_obj(K_CHARS, L);
        r->chars = (jchar *)calloc(L > 0 ? L : 1, sizeof(jchar));
        for (int k = 0; k < L; ++k) r->chars[k] = (jchar)(unsigned char)rows[i][k];
        a->elems[i] = r;
    }
    return a;
}
static char *dup_or_null(jstring s) { return s ? strdup(s->utf) : NULL; }

/* ---- C entry points for tests/test_jni_double.py (ctypes) ---- */
__attribute__((visibility("default"))) char *jd_search(int n, const char *const *names, const char *const *rows, const char *start, int nni, int spr) {
    return dup_or_null(Java_edu_vt_vbi_ci_pepr_tree_NativeTreeEngine_search(&g_env, NULL, java_strings(n, names), java_char_rows(n, rows),
                                                                              start ? d_NewStringUTF(&g_env, start) : NULL, nni, spr));
}
__attribute__((visibility("default"))) char *jd_optimize(int n, const char *const *names, const char *const *rows, const char *newick) {
    return dup_or_null(Java_edu_vt_vbi_ci_pepr_tree_NativeTreeEngine_optimize(&g_env, NULL, java_strings(n, names), java_char_rows(n, rows), d_NewStringUTF(&g_env, newick)));
}
__attribute__((visibility("default"))) int jd_site_lnl(int n, const char *const *names, const char *const *rows, const char *newick, double *out, int cap) {
    jdoubleArray a = Java_edu_vt_vbi_ci_pepr_tree_NativeTreeEngine_siteLnL(&g_env, NULL, java_strings(n, names), java_char_rows(n, rows), d_NewStringUTF(&g_env, newick));
    if (!a) return -1;
    for (int i = 0; i < a->n && i < cap; ++i) out[i] = a->doubles[i];
    return a->n;
}
__attribute__((visibility("default"))) char *jd_parsimony(int n, const char *const *names, const char *const *rows, int seed) {
    return dup_or_null(Java_edu_vt_vbi_ci_pepr_tree_NativeTreeEngine_parsimony(&g_env, NULL, java_strings(n, names), java_char_rows(n, rows), seed));
}
/* genes flattened: gene g has ntax[g] taxa, its names / rows follow each other in the flat arrays; out receives reps + 1 strings */
__attribute__((visibility("default"))) int jd_jackknife(int ng, const int *ntax, const char *const *names, const char *const *rows, int reps, long long seed, char **out) {
    jobjectArray T = d_NewObjectArray(&g_env, ng, NULL, NULL), R = d_NewObjectArray(&g_env, ng, NULL, NULL);
    int off = 0;
    for (int g = 0; g < ng; ++g) { T->elems[g] = java_strings(ntax[g], names + off); R->elems[g] = java_char_rows(ntax[g], rows + off); off += ntax[g]; }
    jobjectArray a = Java_edu_vt_vbi_ci_pepr_tree_NativeTreeEngine_jackknife(&g_env, NULL, T, R, reps, seed);
    if (!a) return -1;
    for (int i = 0; i < a->n; ++i) out[i] = dup_or_null(a->elems[i]);
    return a->n;
}
__attribute__((visibility("default"))) void jd_pin_counts(long *pins, long *unpins) { *pins = g_pins; *unpins = g_unpins; }
__attribute__((visibility("default"))) void jd_free(char *p) { free(p); }
