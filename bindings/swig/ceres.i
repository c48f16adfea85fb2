// SWIG interface over include/skeres_amd.h — the generator route to the same Java surface as the hand-written
// bindings/jni/skeres_amd_jni.c + bindings/java/com/google/ceres/SkeresNative.java (use one or the other).  Replaces the
// reference's ceres.i (which wraps the Ceres C++ headers, ceres.i:137-152): no directors are needed (ceres.i:47-50) —
// cost functions are opaque handles, and the one upcall (CostFunction::Evaluate, ceres.i:48) goes through the
// sk_evaluate_fn trampoline of skeres_amd_jni.c, which this module links as well.
//
//   swig -java -package com.google.ceres -outdir $JAVA_OUT -o skeres_amd_wrap.c -Iinclude bindings/swig/ceres.i
//
// Not run in this repository's image (no swig, no JDK).
%module ceres
%{
#include "skeres_amd.h"
%}

// native double arrays exactly as the reference exposes them (ceres.i:95-96): new DoubleArray(n), getitem, setitem, cast, frompointer
%include "carrays.i"
%array_class(double, DoubleArray);

// lowerCamelCase like the reference (ceres.i:89-92): sk_options_set_max_num_iterations -> skOptionsSetMaxNumIterations
%rename("%(lowercamelcase)s", %$isfunction) "";
%rename("%(lowercamelcase)s", %$isvariable) "";

// the slice helper keeps its reference name and shape (ceres.i:99-107)
%inline %{
struct DoubleArraySlice { static double* get(double* buffer, int start) { return sk_array_slice(buffer, start); } };
struct DoubleMatrix {
  static bool isNull(double** matrix) { return sk_matrix_is_null(matrix) != 0; }
  static double* row(double** matrix, int i) { return sk_matrix_row(matrix, i); }
  static double** toPointerPointer(sk_ptrvec* v) { return sk_ptrvec_to_pointer_pointer(v); }
};
// ceres.i:131-135
void initGoogleLogging(const char* name) { sk_init_logging(name); }
%}

// factory-made losses and parameterizations belong to the JVM proxy (ceres.i:160-167, 187-191)
%newobject sk_loss_trivial; %newobject sk_loss_huber; %newobject sk_loss_soft_l_one; %newobject sk_loss_cauchy; %newobject sk_loss_tukey;
%newobject sk_loss_tolerant; %newobject sk_loss_composed; %newobject sk_loss_scaled;
%newobject sk_local_parameterization_identity; %newobject sk_local_parameterization_subset;
%newobject sk_local_parameterization_quaternion; %newobject sk_local_parameterization_homogeneous_vector;

// int / double arrays of the bulk entry points as Java arrays
%include "arrays_java.i"
%apply int[] { const int* block_sizes, const int* instructions, const int* output_operands, const int* constant_parameters, const int* cuts, const int* last };
%apply double[] { const double* consts, const double* tape_constants, const double* captured };

// function-pointer arguments are bound by hand (the director trampoline, the RCCL hook): not through SWIG
%ignore sk_cost_function_new_callback;
%ignore sk_options_set_distributed;
%ignore sk_allreduce_rccl_fn;

%include "skeres_amd.h"

// load the native libraries as the reference's module class does (ceres.i:213-223)
%pragma(java) jniclasscode=%{
  static {
    try {
      System.loadLibrary("skeres_amd");
      System.loadLibrary("skeres_amd_jni");
    } catch (UnsatisfiedLinkError e) {
      System.err.println("Native code library failed to load. \n" + e);
      System.exit(1);
    }
  }
%}
