/* JNI thunks over libskeres_amd's C ABI (include/skeres_amd.h), one per native method of
 * bindings/java/com/google/ceres/SkeresNative.java — what replaces the SWIG-generated ceres_wrap.cc of the reference
 * (ceres.i:1-224 -> build.sh:13-43).  Handles are jlong (pointers); native double buffers are jlong addresses, exactly as
 * SWIG's SWIGTYPE_p_double carries them (ceres.i:95-107).  Status codes other than SK_OK become Java exceptions:
 * SK_ERR_INVALID_ARGUMENT -> IllegalArgumentException (the Scala `require`s, CORE/CostFunctor.scala:32-33), anything else ->
 * RuntimeException, both with sk_last_error() as the message.
 *
 * The director path (ceres.i:48: CostFunction::Evaluate surfaced to the JVM): jvm_evaluate below is the sk_evaluate_fn
 * handed to sk_cost_function_new_callback; it calls  boolean evaluate(long parameters, long residuals, long jacobians)  on
 * the Java object (a global reference kept by the native side until skCostFunctionFree).
 *
 * NOT COMPILED IN THIS REPOSITORY'S IMAGE (no JDK: no jni.h).  tests/test_bindings_cpu.py compiles this file for SYNTAX
 * ONLY against tests/jni_stub/jni.h, a minimal declaration of the JNI types and JNIEnv members used here; that pins
 * nothing about behaviour.  Build where a JDK exists:
 *   cc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude bindings/jni/skeres_amd_jni.c \
 *      -Lskeres_amd -lskeres_amd -o libskeres_amd_jni.so
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "skeres_amd.h"

#define SK_JNI(ret, name) JNIEXPORT ret JNICALL Java_com_google_ceres_SkeresNative_##name
#define PTR(T, h) ((T*)(intptr_t)(h))
#define HANDLE(p) ((jlong)(intptr_t)(p))

static JavaVM* g_vm = NULL;

JNIEXPORT jint JNICALL JNI_OnLoad(JavaVM* vm, void* reserved) {
  (void)reserved;
  g_vm = vm;
  return JNI_VERSION_1_6;
}

static void throw_status(JNIEnv* env, int status) {
  const char* cls = status == SK_ERR_INVALID_ARGUMENT ? "java/lang/IllegalArgumentException" : "java/lang/RuntimeException";
  jclass c = (*env)->FindClass(env, cls);
  if (c) (*env)->ThrowNew(env, c, sk_last_error());
}
static jint check(JNIEnv* env, int status) {
  if (status != SK_OK) throw_status(env, status);
  return (jint)status;
}
/* A factory that returns NULL: an argument was wrong (an unknown functor id, an index out of range, a nesting too deep) —
 * IllegalArgumentException with sk_last_error().  (Round 4, found by executing this file against the mock JNIEnv of
 * tests/jni_stub: it used to consult sk_last_status(), which belongs to sk_solver_create alone, and so threw whatever an
 * EARLIER failure of the process had left there.) */
static jlong check_handle(JNIEnv* env, const void* p) {
  if (!p) throw_status(env, SK_ERR_INVALID_ARGUMENT);
  return HANDLE(p);
}
/* ... and a factory that records a typed status of its own for a NULL result (sk_solver_create, sk_allreduce_rccl_init: both set
 * sk_last_status on every call, so nothing stale can be read): that status decides the exception class — a communicator that
 * cannot come up is SK_ERR_COMM / SK_ERR_NO_DEVICE, not an illegal argument. */
static jlong check_handle_status(JNIEnv* env, const void* p) {
  if (!p) { const int st = sk_last_status(); throw_status(env, st != SK_OK ? st : SK_ERR_INVALID_ARGUMENT); }
  return HANDLE(p);
}

/* ---- library ---- */
SK_JNI(jstring, skVersion)(JNIEnv* env, jclass c) { (void)c; return (*env)->NewStringUTF(env, sk_version()); }
SK_JNI(jstring, skLastError)(JNIEnv* env, jclass c) { (void)c; return (*env)->NewStringUTF(env, sk_last_error()); }
SK_JNI(jint, skDeviceCount)(JNIEnv* env, jclass c) { (void)env; (void)c; return sk_device_count(); }
SK_JNI(void, skInitLogging)(JNIEnv* env, jclass c, jstring name) {  /* ceres.initGoogleLogging, ceres.i:131-135 */
  (void)c;
  const char* s = (*env)->GetStringUTFChars(env, name, NULL);
  sk_init_logging(s);
  (*env)->ReleaseStringUTFChars(env, name, s);
}

/* ---- DoubleArray / DoubleArraySlice / DoubleMatrix / StdVectorDoublePointer (ceres.i:79-125) ---- */
SK_JNI(jlong, skArrayNew)(JNIEnv* env, jclass c, jint n) { (void)c; return check_handle(env, sk_array_new(n)); }
SK_JNI(void, skArrayFree)(JNIEnv* env, jclass c, jlong a) { (void)env; (void)c; sk_array_free(PTR(double, a)); }
SK_JNI(jdouble, skArrayGetitem)(JNIEnv* env, jclass c, jlong a, jint i) { (void)env; (void)c; return sk_array_getitem(PTR(double, a), i); }
SK_JNI(void, skArraySetitem)(JNIEnv* env, jclass c, jlong a, jint i, jdouble v) { (void)env; (void)c; sk_array_setitem(PTR(double, a), i, v); }
SK_JNI(jlong, skArraySlice)(JNIEnv* env, jclass c, jlong a, jint start) { (void)env; (void)c; return HANDLE(sk_array_slice(PTR(double, a), start)); }
/* bulk copies: one crossing instead of one per element (CORE/RichDoubleArray.scala:36-39, 65-69) */
SK_JNI(void, skArrayCopyIn)(JNIEnv* env, jclass c, jlong dst, jdoubleArray src, jint n) {
  (void)c;
  (*env)->GetDoubleArrayRegion(env, src, 0, n, PTR(double, dst));
}
SK_JNI(void, skArrayCopyOut)(JNIEnv* env, jclass c, jlong src, jdoubleArray dst, jint n) {
  (void)c;
  (*env)->SetDoubleArrayRegion(env, dst, 0, n, PTR(double, src));
}
SK_JNI(jboolean, skMatrixIsNull)(JNIEnv* env, jclass c, jlong m) { (void)env; (void)c; return sk_matrix_is_null(PTR(double*, m)) ? JNI_TRUE : JNI_FALSE; }
SK_JNI(jlong, skMatrixRow)(JNIEnv* env, jclass c, jlong m, jint i) { (void)env; (void)c; return HANDLE(sk_matrix_row(PTR(double*, m), i)); }
SK_JNI(jlong, skPtrvecNew)(JNIEnv* env, jclass c) { (void)c; return check_handle(env, sk_ptrvec_new()); }
SK_JNI(void, skPtrvecFree)(JNIEnv* env, jclass c, jlong v) { (void)env; (void)c; sk_ptrvec_free(PTR(sk_ptrvec, v)); }
SK_JNI(void, skPtrvecAdd)(JNIEnv* env, jclass c, jlong v, jlong p) { (void)env; (void)c; sk_ptrvec_add(PTR(sk_ptrvec, v), PTR(double, p)); }
SK_JNI(jint, skPtrvecSize)(JNIEnv* env, jclass c, jlong v) { (void)env; (void)c; return sk_ptrvec_size(PTR(sk_ptrvec, v)); }
SK_JNI(jlong, skPtrvecGet)(JNIEnv* env, jclass c, jlong v, jint i) { (void)env; (void)c; return HANDLE(sk_ptrvec_get(PTR(sk_ptrvec, v), i)); }
SK_JNI(jlong, skPtrvecToPointerPointer)(JNIEnv* env, jclass c, jlong v) { (void)env; (void)c; return HANDLE(sk_ptrvec_to_pointer_pointer(PTR(sk_ptrvec, v))); }
SK_JNI(void, skPtrvecSet)(JNIEnv* env, jclass c, jlong v, jint i, jlong p) { (void)env; (void)c; sk_ptrvec_set(PTR(sk_ptrvec, v), i, PTR(double, p)); }

/* ---- PredefinedLossFunctions (ceres.i:159-184) ---- */
SK_JNI(jlong, skLossTrivial)(JNIEnv* env, jclass c) { (void)c; return check_handle(env, sk_loss_trivial()); }
SK_JNI(jlong, skLossHuber)(JNIEnv* env, jclass c, jdouble a) { (void)c; return check_handle(env, sk_loss_huber(a)); }
SK_JNI(jlong, skLossSoftLOne)(JNIEnv* env, jclass c, jdouble a) { (void)c; return check_handle(env, sk_loss_soft_l_one(a)); }
SK_JNI(jlong, skLossCauchy)(JNIEnv* env, jclass c, jdouble a) { (void)c; return check_handle(env, sk_loss_cauchy(a)); }
SK_JNI(jlong, skLossTukey)(JNIEnv* env, jclass c, jdouble a) { (void)c; return check_handle(env, sk_loss_tukey(a)); }
SK_JNI(jlong, skLossTolerant)(JNIEnv* env, jclass c, jdouble a, jdouble b) { (void)c; return check_handle(env, sk_loss_tolerant(a, b)); }
SK_JNI(jlong, skLossComposed)(JNIEnv* env, jclass c, jlong f, jlong g) { (void)c; return check_handle(env, sk_loss_composed(PTR(sk_loss_function, f), PTR(sk_loss_function, g))); }
SK_JNI(jlong, skLossScaled)(JNIEnv* env, jclass c, jlong rho, jdouble a) { (void)c; return check_handle(env, sk_loss_scaled(PTR(sk_loss_function, rho), a)); }
SK_JNI(void, skLossFree)(JNIEnv* env, jclass c, jlong l) { (void)env; (void)c; sk_loss_free(PTR(sk_loss_function, l)); }

/* ---- PredefinedLocalParameterizations (ceres.i:186-210) ---- */
SK_JNI(jlong, skLocalParameterizationIdentity)(JNIEnv* env, jclass c, jint size) { (void)c; return check_handle(env, sk_local_parameterization_identity(size)); }
SK_JNI(jlong, skLocalParameterizationSubset)(JNIEnv* env, jclass c, jint size, jintArray constant) {
  (void)c;
  const jsize n = (*env)->GetArrayLength(env, constant);
  jint* p = (*env)->GetIntArrayElements(env, constant, NULL);
  sk_local_parameterization* lp = sk_local_parameterization_subset(size, (const int*)p, (int)n);
  (*env)->ReleaseIntArrayElements(env, constant, p, JNI_ABORT);
  return check_handle(env, lp);
}
SK_JNI(jlong, skLocalParameterizationQuaternion)(JNIEnv* env, jclass c) { (void)c; return check_handle(env, sk_local_parameterization_quaternion()); }
SK_JNI(jlong, skLocalParameterizationHomogeneousVector)(JNIEnv* env, jclass c, jint size) { (void)c; return check_handle(env, sk_local_parameterization_homogeneous_vector(size)); }
SK_JNI(void, skLocalParameterizationFree)(JNIEnv* env, jclass c, jlong p) { (void)env; (void)c; sk_local_parameterization_free(PTR(sk_local_parameterization, p)); }

/* ---- CostFunction ---- */
/* a functor with a device body (CORE/CostFunctor.scala:44 with deviceFunctorId defined) */
SK_JNI(jlong, skCostFunctionNewAutodiff)(JNIEnv* env, jclass c, jint functor_id, jdoubleArray consts) {
  (void)c;
  const jsize n = consts ? (*env)->GetArrayLength(env, consts) : 0;
  jdouble* p = n ? (*env)->GetDoubleArrayElements(env, consts, NULL) : NULL;
  sk_cost_function* cf = sk_cost_function_new_autodiff(functor_id, p, (int)n);
  if (p) (*env)->ReleaseDoubleArrayElements(env, consts, p, JNI_ABORT);
  return check_handle(env, cf);
}

/* the director: one of these per JVM cost function (skDirectorNew), released with it (skDirectorFree) */
typedef struct { jobject self; jmethodID evaluate; } jvm_director;

static int jvm_evaluate(void* user, double const* const* parameters, double* residuals, double** jacobians) {
  jvm_director* d = (jvm_director*)user;
  JNIEnv* env = NULL;
  int attached = 0;
  if ((*g_vm)->GetEnv(g_vm, (void**)&env, JNI_VERSION_1_6) != JNI_OK) {  /* upcalls arrive on the thread that called sk_solve: normally attached */
    if ((*g_vm)->AttachCurrentThread(g_vm, (void**)&env, NULL) != JNI_OK) return 0;
    attached = 1;
  }
  /* an earlier evaluate() of this solve threw: the solver may still ask for other blocks (or retry the step) before it gives
   * up, and calling into the JVM with an exception pending is undefined behaviour (an abort under -Xcheck:jni).  Every
   * further evaluation fails without an up-call; the exception surfaces, untouched, when sk_solve returns (skSolve).
   * (On a thread this function had to attach, DetachCurrentThread would drop a pending exception: such a thread has no Java
   * frame to throw into — the failure is then reported by the solve's status alone.) */
  int result = 0;
  if (!(*env)->ExceptionCheck(env)) {
    const jboolean ok = (*env)->CallBooleanMethod(env, d->self, d->evaluate, HANDLE(parameters), HANDLE(residuals), HANDLE(jacobians));
    result = ok == JNI_TRUE;
    if ((*env)->ExceptionCheck(env)) result = 0;  /* evaluate threw: the block cannot be evaluated */
  }
  if (attached) (*g_vm)->DetachCurrentThread(g_vm);
  return result;
}

/* self: an object with  boolean evaluateNative(long parameters, long residuals, long jacobians)  (SizedCostFunction.scala).
 * The director holds a global reference to it: the JVM object cannot be collected while native code may still call it
 * (what CORE/Problem.scala:29-32 pins by hand in the reference). */
SK_JNI(jlong, skDirectorNew)(JNIEnv* env, jclass c, jobject self) {
  (void)c;
  jmethodID m = (*env)->GetMethodID(env, (*env)->GetObjectClass(env, self), "evaluateNative", "(JJJ)Z");
  if (!m) return 0;  /* NoSuchMethodError is pending */
  jvm_director* d = (jvm_director*)malloc(sizeof(jvm_director));
  if (!d) { throw_status(env, SK_ERR_HIP); return 0; }
  d->self = (*env)->NewGlobalRef(env, self);
  d->evaluate = m;
  return HANDLE(d);
}
SK_JNI(void, skDirectorFree)(JNIEnv* env, jclass c, jlong director) {
  (void)c;
  jvm_director* d = PTR(jvm_director, director);
  if (!d) return;
  (*env)->DeleteGlobalRef(env, d->self);
  free(d);
}
SK_JNI(jlong, skCostFunctionNewCallback)(JNIEnv* env, jclass c, jlong director, jint num_residuals, jintArray block_sizes) {
  (void)c;
  const jsize nb = (*env)->GetArrayLength(env, block_sizes);
  jint* bs = (*env)->GetIntArrayElements(env, block_sizes, NULL);
  sk_cost_function* cf = sk_cost_function_new_callback(jvm_evaluate, PTR(jvm_director, director), num_residuals, (const int*)bs, (int)nb);
  (*env)->ReleaseIntArrayElements(env, block_sizes, bs, JNI_ABORT);
  return check_handle(env, cf);
}

/* a recorded functor body (Recording.scala -> sk_cost_function_new_tape) */
SK_JNI(jlong, skCostFunctionNewTape)(JNIEnv* env, jclass c, jint num_residuals, jintArray block_sizes, jintArray instructions, jdoubleArray tape_constants,
                                     jint num_registers, jintArray output_operands, jdoubleArray captured) {
  (void)c;
  const jsize nb = (*env)->GetArrayLength(env, block_sizes), ni = (*env)->GetArrayLength(env, instructions);
  const jsize nc = tape_constants ? (*env)->GetArrayLength(env, tape_constants) : 0, ncap = captured ? (*env)->GetArrayLength(env, captured) : 0;
  jint* bs = (*env)->GetIntArrayElements(env, block_sizes, NULL);
  jint* ins = (*env)->GetIntArrayElements(env, instructions, NULL);
  jint* out = (*env)->GetIntArrayElements(env, output_operands, NULL);
  jdouble* tc = nc ? (*env)->GetDoubleArrayElements(env, tape_constants, NULL) : NULL;
  jdouble* cap = ncap ? (*env)->GetDoubleArrayElements(env, captured, NULL) : NULL;
  sk_cost_function* cf = sk_cost_function_new_tape(num_residuals, (const int*)bs, (int)nb, (const int*)ins, (int)(ni / 5), tc, (int)nc, num_registers,
                                                   (const int*)out, cap, (int)ncap);
  if (cap) (*env)->ReleaseDoubleArrayElements(env, captured, cap, JNI_ABORT);
  if (tc) (*env)->ReleaseDoubleArrayElements(env, tape_constants, tc, JNI_ABORT);
  (*env)->ReleaseIntArrayElements(env, output_operands, out, JNI_ABORT);
  (*env)->ReleaseIntArrayElements(env, instructions, ins, JNI_ABORT);
  (*env)->ReleaseIntArrayElements(env, block_sizes, bs, JNI_ABORT);
  return check_handle(env, cf);
}
SK_JNI(void, skCostFunctionFree)(JNIEnv* env, jclass c, jlong cf) { (void)env; (void)c; sk_cost_function_free(PTR(sk_cost_function, cf)); }
SK_JNI(jint, skCostFunctionNumResiduals)(JNIEnv* env, jclass c, jlong cf) { (void)env; (void)c; return sk_cost_function_num_residuals(PTR(sk_cost_function, cf)); }
SK_JNI(jint, skCostFunctionNumParameterBlocks)(JNIEnv* env, jclass c, jlong cf) { (void)env; (void)c; return sk_cost_function_num_parameter_blocks(PTR(sk_cost_function, cf)); }
SK_JNI(jint, skCostFunctionParameterBlockSize)(JNIEnv* env, jclass c, jlong cf, jint i) { (void)env; (void)c; return sk_cost_function_parameter_block_size(PTR(sk_cost_function, cf), i); }
/* CostFunction.evaluate for a cost function with a device body (TEST/AutodiffCostFuntionSpec.scala:39-44 calls it directly) */
SK_JNI(jboolean, skCostFunctionEvaluate)(JNIEnv* env, jclass c, jlong cf, jlong parameters, jlong residuals, jlong jacobians) {
  (void)c;
  const int rc = sk_cost_function_evaluate(PTR(sk_cost_function, cf), (double const* const*)PTR(double*, parameters), PTR(double, residuals), PTR(double*, jacobians));
  if (rc == SK_ERR_EVALUATION_FAILED) return JNI_FALSE;  /* the functor returned an empty array: CORE/AutodiffCostFunction.scala:88-89 */
  check(env, rc);
  return rc == SK_OK ? JNI_TRUE : JNI_FALSE;
}

/* ---- Problem (CORE/Problem.scala:16-32) ---- */
SK_JNI(jlong, skProblemNew)(JNIEnv* env, jclass c) { (void)c; return check_handle(env, sk_problem_new()); }
SK_JNI(void, skProblemFree)(JNIEnv* env, jclass c, jlong p) { (void)env; (void)c; sk_problem_free(PTR(sk_problem, p)); }
/* CeresProblem.addResidualBlock(CostFunction, LossFunction, StdVectorDoublePointer): the pointers come as a sk_ptrvec */
SK_JNI(jlong, skProblemAddResidualBlock)(JNIEnv* env, jclass c, jlong p, jlong cost, jlong loss, jlong ptrvec) {
  (void)c;
  sk_ptrvec* v = PTR(sk_ptrvec, ptrvec);
  const int n = sk_ptrvec_size(v);
  double* blocks[16];
  double** b = n <= 16 ? blocks : (double**)malloc(sizeof(double*) * (size_t)n);
  if (!b) { throw_status(env, SK_ERR_HIP); return 0; }
  for (int i = 0; i < n; ++i) b[i] = sk_ptrvec_get(v, i);
  sk_residual_block_id id = 0;
  const int rc = sk_problem_add_residual_block(PTR(sk_problem, p), PTR(sk_cost_function, cost), PTR(sk_loss_function, loss), b, n, &id);
  if (b != blocks) free(b);
  check(env, rc);
  return (jlong)id;
}
/* the setup loop of EX/SimpleBundleAdjuster.scala:139-145 in ONE crossing: n blocks of one device functor; `offsets` are
 * n x num_blocks element offsets into the one contiguous native array `base` (BalProblem's layout, :18-34) */
SK_JNI(jint, skProblemAddResidualBlocks)(JNIEnv* env, jclass c, jlong p, jint functor_id, jint n, jdoubleArray consts, jlong loss, jlong base, jlongArray offsets) {
  (void)c;
  const jsize no = (*env)->GetArrayLength(env, offsets);
  jlong* off = (*env)->GetLongArrayElements(env, offsets, NULL);
  jdouble* cs = (*env)->GetDoubleArrayElements(env, consts, NULL);
  double** blocks = (double**)malloc(sizeof(double*) * (size_t)(no ? no : 1));
  int rc = SK_ERR_HIP;
  if (blocks) {
    for (jsize i = 0; i < no; ++i) blocks[i] = PTR(double, base) + off[i];
    rc = sk_problem_add_residual_blocks(PTR(sk_problem, p), functor_id, n, cs, PTR(sk_loss_function, loss), blocks);
    free(blocks);
  }
  (*env)->ReleaseDoubleArrayElements(env, consts, cs, JNI_ABORT);
  (*env)->ReleaseLongArrayElements(env, offsets, off, JNI_ABORT);
  return check(env, rc);
}
SK_JNI(jint, skProblemAddResidualBlocksTape)(JNIEnv* env, jclass c, jlong p, jlong cost, jint n, jdoubleArray captured, jlong loss, jlong base, jlongArray offsets) {
  (void)c;
  const jsize no = (*env)->GetArrayLength(env, offsets);
  jlong* off = (*env)->GetLongArrayElements(env, offsets, NULL);
  jdouble* cs = captured ? (*env)->GetDoubleArrayElements(env, captured, NULL) : NULL;
  double** blocks = (double**)malloc(sizeof(double*) * (size_t)(no ? no : 1));
  int rc = SK_ERR_HIP;
  if (blocks) {
    for (jsize i = 0; i < no; ++i) blocks[i] = PTR(double, base) + off[i];
    rc = sk_problem_add_residual_blocks_tape(PTR(sk_problem, p), PTR(sk_cost_function, cost), n, cs, PTR(sk_loss_function, loss), blocks);
    free(blocks);
  }
  if (cs) (*env)->ReleaseDoubleArrayElements(env, captured, cs, JNI_ABORT);
  (*env)->ReleaseLongArrayElements(env, offsets, off, JNI_ABORT);
  return check(env, rc);
}
SK_JNI(jint, skProblemAddParameterBlock)(JNIEnv* env, jclass c, jlong p, jlong values, jint size, jlong parameterization) {
  (void)c; return check(env, sk_problem_add_parameter_block(PTR(sk_problem, p), PTR(double, values), size, PTR(sk_local_parameterization, parameterization)));
}
SK_JNI(jint, skProblemSetParameterization)(JNIEnv* env, jclass c, jlong p, jlong values, jlong parameterization) {
  (void)c; return check(env, sk_problem_set_parameterization(PTR(sk_problem, p), PTR(double, values), PTR(sk_local_parameterization, parameterization)));
}
SK_JNI(jint, skProblemSetParameterBlockConstant)(JNIEnv* env, jclass c, jlong p, jlong values) { (void)c; return check(env, sk_problem_set_parameter_block_constant(PTR(sk_problem, p), PTR(double, values))); }
SK_JNI(jint, skProblemSetParameterBlockVariable)(JNIEnv* env, jclass c, jlong p, jlong values) { (void)c; return check(env, sk_problem_set_parameter_block_variable(PTR(sk_problem, p), PTR(double, values))); }
SK_JNI(jint, skProblemNumResidualBlocks)(JNIEnv* env, jclass c, jlong p) { (void)env; (void)c; return sk_problem_num_residual_blocks(PTR(sk_problem, p)); }
SK_JNI(jint, skProblemNumParameterBlocks)(JNIEnv* env, jclass c, jlong p) { (void)env; (void)c; return sk_problem_num_parameter_blocks(PTR(sk_problem, p)); }
SK_JNI(jint, skProblemNumParameters)(JNIEnv* env, jclass c, jlong p) { (void)env; (void)c; return sk_problem_num_parameters(PTR(sk_problem, p)); }
SK_JNI(jint, skProblemNumResiduals)(JNIEnv* env, jclass c, jlong p) { (void)env; (void)c; return sk_problem_num_residuals(PTR(sk_problem, p)); }

/* ---- Solver.Options (the setters EX/SimpleBundleAdjuster.scala:147-149, EX/CurveFitting.scala:119-122 use, and the rest) ---- */
SK_JNI(jlong, skOptionsNew)(JNIEnv* env, jclass c) { (void)c; return check_handle(env, sk_options_new()); }
SK_JNI(void, skOptionsFree)(JNIEnv* env, jclass c, jlong o) { (void)env; (void)c; sk_options_free(PTR(sk_options, o)); }
#define SK_OPT_INT(jname, cname) SK_JNI(jint, jname)(JNIEnv* env, jclass c, jlong o, jint v) { (void)c; return check(env, cname(PTR(sk_options, o), v)); }
#define SK_OPT_DBL(jname, cname) SK_JNI(jint, jname)(JNIEnv* env, jclass c, jlong o, jdouble v) { (void)c; return check(env, cname(PTR(sk_options, o), v)); }
SK_OPT_INT(skOptionsSetLinearSolverType, sk_options_set_linear_solver_type)
SK_OPT_INT(skOptionsSetMinimizerType, sk_options_set_minimizer_type)
SK_OPT_INT(skOptionsSetMaxNumIterations, sk_options_set_max_num_iterations)
SK_OPT_INT(skOptionsSetMinimizerProgressToStdout, sk_options_set_minimizer_progress_to_stdout)
SK_OPT_DBL(skOptionsSetFunctionTolerance, sk_options_set_function_tolerance)
SK_OPT_DBL(skOptionsSetGradientTolerance, sk_options_set_gradient_tolerance)
SK_OPT_DBL(skOptionsSetParameterTolerance, sk_options_set_parameter_tolerance)
SK_OPT_DBL(skOptionsSetInitialTrustRegionRadius, sk_options_set_initial_trust_region_radius)
SK_OPT_DBL(skOptionsSetMaxTrustRegionRadius, sk_options_set_max_trust_region_radius)
SK_OPT_DBL(skOptionsSetMinTrustRegionRadius, sk_options_set_min_trust_region_radius)
SK_OPT_DBL(skOptionsSetMinRelativeDecrease, sk_options_set_min_relative_decrease)
SK_OPT_DBL(skOptionsSetMinLmDiagonal, sk_options_set_min_lm_diagonal)
SK_OPT_DBL(skOptionsSetMaxLmDiagonal, sk_options_set_max_lm_diagonal)
SK_OPT_INT(skOptionsSetJacobiScaling, sk_options_set_jacobi_scaling)
SK_OPT_INT(skOptionsSetMaxNumConsecutiveInvalidSteps, sk_options_set_max_num_consecutive_invalid_steps)
SK_OPT_INT(skOptionsSetDevice, sk_options_set_device)
SK_OPT_INT(skOptionsSetCholeskyEnvelope, sk_options_set_cholesky_envelope)
SK_OPT_INT(skOptionsSetCholeskyDissection, sk_options_set_cholesky_dissection)
SK_OPT_INT(skOptionsSetCholeskyBorder, sk_options_set_cholesky_border)
SK_OPT_INT(skOptionsSetResidentKernels, sk_options_set_resident_kernels)
SK_OPT_INT(skOptionsSetGraphReplay, sk_options_set_graph_replay)
SK_OPT_INT(skOptionsSetMaxSegments, sk_options_set_max_segments)
SK_OPT_INT(skOptionsSetDistributionMode, sk_options_set_distribution_mode)
SK_JNI(jint, skOptionsSetRetainedPoints)(JNIEnv* env, jclass c, jlong o, jint mode, jint max_points) { (void)c; return check(env, sk_options_set_retained_points(PTR(sk_options, o), mode, max_points)); }
SK_JNI(jint, skOptionsSetCholeskyTuning)(JNIEnv* env, jclass c, jlong o, jint group, jint lookahead) { (void)c; return check(env, sk_options_set_cholesky_tuning(PTR(sk_options, o), group, lookahead)); }
/* multi-GPU from the JVM: the library's own RCCL hook (no collective to write on the JVM side) */
SK_JNI(jbyteArray, skRcclUniqueId)(JNIEnv* env, jclass c) {
  (void)c;
  char id[128];
  if (check(env, sk_rccl_unique_id(id)) != SK_OK) return NULL;
  jbyteArray out = (*env)->NewByteArray(env, 128);
  if (out) (*env)->SetByteArrayRegion(env, out, 0, 128, (const jbyte*)id);
  return out;
}
SK_JNI(jlong, skAllreduceRcclInit)(JNIEnv* env, jclass c, jint rank, jint world, jbyteArray id) {
  (void)c;
  char buf[128];
  (*env)->GetByteArrayRegion(env, id, 0, 128, (jbyte*)buf);
  return check_handle_status(env, sk_allreduce_rccl_init(rank, world, buf));
}
SK_JNI(void, skAllreduceRcclFree)(JNIEnv* env, jclass c, jlong h) { (void)env; (void)c; sk_allreduce_rccl_free(PTR(sk_rccl, h)); }
SK_JNI(jint, skOptionsSetDistributedRccl)(JNIEnv* env, jclass c, jlong o, jint rank, jint world, jlong rccl) {
  (void)c; return check(env, sk_options_set_distributed(PTR(sk_options, o), rank, world, sk_allreduce_rccl_fn(), PTR(sk_rccl, rccl)));
}

/* ---- Solver.Summary, ceres.solve ---- */
SK_JNI(jlong, skSummaryNew)(JNIEnv* env, jclass c) { (void)c; return check_handle(env, sk_summary_new()); }
SK_JNI(void, skSummaryFree)(JNIEnv* env, jclass c, jlong s) { (void)env; (void)c; sk_summary_free(PTR(sk_summary, s)); }
SK_JNI(jdouble, skSummaryInitialCost)(JNIEnv* env, jclass c, jlong s) { (void)env; (void)c; return sk_summary_initial_cost(PTR(sk_summary, s)); }
SK_JNI(jdouble, skSummaryFinalCost)(JNIEnv* env, jclass c, jlong s) { (void)env; (void)c; return sk_summary_final_cost(PTR(sk_summary, s)); }
SK_JNI(jint, skSummaryNumIterations)(JNIEnv* env, jclass c, jlong s) { (void)env; (void)c; return sk_summary_num_iterations(PTR(sk_summary, s)); }
SK_JNI(jint, skSummaryNumSuccessfulSteps)(JNIEnv* env, jclass c, jlong s) { (void)env; (void)c; return sk_summary_num_successful_steps(PTR(sk_summary, s)); }
SK_JNI(jint, skSummaryNumUnsuccessfulSteps)(JNIEnv* env, jclass c, jlong s) { (void)env; (void)c; return sk_summary_num_unsuccessful_steps(PTR(sk_summary, s)); }
SK_JNI(jint, skSummaryTerminationType)(JNIEnv* env, jclass c, jlong s) { (void)env; (void)c; return sk_summary_termination_type(PTR(sk_summary, s)); }
SK_JNI(jint, skSummaryLinearSolverTypeUsed)(JNIEnv* env, jclass c, jlong s) { (void)env; (void)c; return sk_summary_linear_solver_type_used(PTR(sk_summary, s)); }
SK_JNI(jint, skSummaryLinearSolverTypeGiven)(JNIEnv* env, jclass c, jlong s) { (void)env; (void)c; return sk_summary_linear_solver_type_given(PTR(sk_summary, s)); }
SK_JNI(jstring, skSummaryMessage)(JNIEnv* env, jclass c, jlong s) { (void)c; return (*env)->NewStringUTF(env, sk_summary_message(PTR(sk_summary, s))); }
SK_JNI(jstring, skSummaryBriefReport)(JNIEnv* env, jclass c, jlong s) { (void)c; return (*env)->NewStringUTF(env, sk_summary_brief_report(PTR(sk_summary, s))); }
SK_JNI(jstring, skSummaryFullReport)(JNIEnv* env, jclass c, jlong s) { (void)c; return (*env)->NewStringUTF(env, sk_summary_full_report(PTR(sk_summary, s))); }
SK_JNI(jdouble, skSummaryIterationField)(JNIEnv* env, jclass c, jlong s, jint it, jint field) { (void)env; (void)c; return sk_summary_iteration_field(PTR(sk_summary, s), it, field); }
/* ceres.solve(options, problem, summary): blocks until done; director upcalls arrive on this thread */
SK_JNI(jint, skSolve)(JNIEnv* env, jclass c, jlong options, jlong problem, jlong summary) {
  (void)c;
  const int rc = sk_solve(PTR(sk_options, options), PTR(sk_problem, problem), PTR(sk_summary, summary));
  if ((*env)->ExceptionCheck(env)) return (jint)rc;  /* an evaluate() upcall threw: let that exception propagate */
  return check(env, rc);
}
/* device twin of CORE/Rotation.scala for cross-checks from the JVM */
SK_JNI(jint, skRotationApply)(JNIEnv* env, jclass c, jint op, jboolean row_major, jint jet_dim, jdoubleArray in, jint n, jdoubleArray out) {
  (void)c;
  jdouble* pi = (*env)->GetDoubleArrayElements(env, in, NULL);
  jdouble* po = (*env)->GetDoubleArrayElements(env, out, NULL);
  const int rc = sk_rotation_apply(op, row_major == JNI_TRUE, jet_dim, pi, n, po);
  (*env)->ReleaseDoubleArrayElements(env, out, po, 0);
  (*env)->ReleaseDoubleArrayElements(env, in, pi, JNI_ABORT);
  return check(env, rc);
}
