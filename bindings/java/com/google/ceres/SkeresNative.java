package com.google.ceres;

/**
 * Native methods of libskeres_amd_jni (bindings/jni/skeres_amd_jni.c), one per entry point of the C ABI the Scala layer
 * uses (include/skeres_amd.h).  Replaces the SWIG-generated com.google.ceres.ceresJNI of the reference (ceres.i, build.sh:13-24).
 * Handles and native double buffers are longs.  Loads the native libraries exactly as the reference's module class does
 * (ceres.i:213-223): a failure prints the reason and exits.
 *
 * Not compiled in this repository's image (no JDK).
 */
public final class SkeresNative {
  static {
    try {
      System.loadLibrary("skeres_amd");      // the solver: HIP kernels + C ABI
      System.loadLibrary("skeres_amd_jni");  // these thunks
    } catch (UnsatisfiedLinkError e) {
      System.err.println("Native code library failed to load. \n" + e);
      System.exit(1);
    }
  }
  private SkeresNative() {}

  // library
  public static native String skVersion();
  public static native String skLastError();
  public static native int skDeviceCount();
  public static native void skInitLogging(String name);
  // DoubleArray / DoubleArraySlice / DoubleMatrix / StdVectorDoublePointer
  public static native long skArrayNew(int n);
  public static native void skArrayFree(long a);
  public static native double skArrayGetitem(long a, int i);
  public static native void skArraySetitem(long a, int i, double v);
  public static native long skArraySlice(long a, int start);
  public static native void skArrayCopyIn(long dst, double[] src, int n);
  public static native void skArrayCopyOut(long src, double[] dst, int n);
  public static native boolean skMatrixIsNull(long m);
  public static native long skMatrixRow(long m, int i);
  public static native long skPtrvecNew();
  public static native void skPtrvecFree(long v);
  public static native void skPtrvecAdd(long v, long p);
  public static native int skPtrvecSize(long v);
  public static native long skPtrvecGet(long v, int i);
  public static native void skPtrvecSet(long v, int i, long p);
  public static native long skPtrvecToPointerPointer(long v);
  // PredefinedLossFunctions
  public static native long skLossTrivial();
  public static native long skLossHuber(double a);
  public static native long skLossSoftLOne(double a);
  public static native long skLossCauchy(double a);
  public static native long skLossTukey(double a);
  public static native long skLossTolerant(double a, double b);
  public static native long skLossComposed(long f, long g);
  public static native long skLossScaled(long rho, double a);
  public static native void skLossFree(long loss);
  // PredefinedLocalParameterizations
  public static native long skLocalParameterizationIdentity(int size);
  public static native long skLocalParameterizationSubset(int size, int[] constantParameters);
  public static native long skLocalParameterizationQuaternion();
  public static native long skLocalParameterizationHomogeneousVector(int size);
  public static native void skLocalParameterizationFree(long p);
  // CostFunction
  public static native long skCostFunctionNewAutodiff(int functorId, double[] consts);
  public static native long skDirectorNew(Object self);   // self.evaluateNative(long, long, long): boolean
  public static native void skDirectorFree(long director);
  public static native long skCostFunctionNewCallback(long director, int numResiduals, int[] blockSizes);
  public static native long skCostFunctionNewTape(int numResiduals, int[] blockSizes, int[] instructions, double[] tapeConstants,
                                                  int numRegisters, int[] outputOperands, double[] captured);
  public static native void skCostFunctionFree(long cf);
  public static native int skCostFunctionNumResiduals(long cf);
  public static native int skCostFunctionNumParameterBlocks(long cf);
  public static native int skCostFunctionParameterBlockSize(long cf, int i);
  public static native boolean skCostFunctionEvaluate(long cf, long parameters, long residuals, long jacobians);
  // Problem
  public static native long skProblemNew();
  public static native void skProblemFree(long p);
  public static native long skProblemAddResidualBlock(long p, long cost, long loss, long ptrvec);
  public static native int skProblemAddResidualBlocks(long p, int functorId, int n, double[] consts, long loss, long base, long[] offsets);
  public static native int skProblemAddResidualBlocksTape(long p, long cost, int n, double[] captured, long loss, long base, long[] offsets);
  public static native int skProblemAddParameterBlock(long p, long values, int size, long parameterization);
  public static native int skProblemSetParameterization(long p, long values, long parameterization);
  public static native int skProblemSetParameterBlockConstant(long p, long values);
  public static native int skProblemSetParameterBlockVariable(long p, long values);
  public static native int skProblemNumResidualBlocks(long p);
  public static native int skProblemNumParameterBlocks(long p);
  public static native int skProblemNumParameters(long p);
  public static native int skProblemNumResiduals(long p);
  // Solver.Options
  public static native long skOptionsNew();
  public static native void skOptionsFree(long o);
  public static native int skOptionsSetLinearSolverType(long o, int v);
  public static native int skOptionsSetMinimizerType(long o, int v);
  public static native int skOptionsSetMaxNumIterations(long o, int v);
  public static native int skOptionsSetMinimizerProgressToStdout(long o, int v);
  public static native int skOptionsSetFunctionTolerance(long o, double v);
  public static native int skOptionsSetGradientTolerance(long o, double v);
  public static native int skOptionsSetParameterTolerance(long o, double v);
  public static native int skOptionsSetInitialTrustRegionRadius(long o, double v);
  public static native int skOptionsSetMaxTrustRegionRadius(long o, double v);
  public static native int skOptionsSetMinTrustRegionRadius(long o, double v);
  public static native int skOptionsSetMinRelativeDecrease(long o, double v);
  public static native int skOptionsSetMinLmDiagonal(long o, double v);
  public static native int skOptionsSetMaxLmDiagonal(long o, double v);
  public static native int skOptionsSetJacobiScaling(long o, int v);
  public static native int skOptionsSetMaxNumConsecutiveInvalidSteps(long o, int v);
  public static native int skOptionsSetDevice(long o, int v);
  public static native int skOptionsSetCholeskyEnvelope(long o, int v);
  public static native int skOptionsSetCholeskyDissection(long o, int v);
  public static native int skOptionsSetCholeskyBorder(long o, int v);
  public static native int skOptionsSetRetainedPoints(long o, int mode, int maxPoints);
  public static native int skOptionsSetResidentKernels(long o, int v);
  public static native int skOptionsSetGraphReplay(long o, int v);
  public static native int skOptionsSetMaxSegments(long o, int v);
  public static native int skOptionsSetDistributionMode(long o, int v);
  public static native int skOptionsSetCholeskyTuning(long o, int group, int lookahead);
  public static native byte[] skRcclUniqueId();
  public static native long skAllreduceRcclInit(int rank, int world, byte[] id);
  public static native void skAllreduceRcclFree(long h);
  public static native int skOptionsSetDistributedRccl(long o, int rank, int world, long rccl);
  // Solver.Summary, ceres.solve
  public static native long skSummaryNew();
  public static native void skSummaryFree(long s);
  public static native double skSummaryInitialCost(long s);
  public static native double skSummaryFinalCost(long s);
  public static native int skSummaryNumIterations(long s);
  public static native int skSummaryLinearSolverTypeUsed(long s);
  public static native int skSummaryLinearSolverTypeGiven(long s);
  public static native int skSummaryNumSuccessfulSteps(long s);
  public static native int skSummaryNumUnsuccessfulSteps(long s);
  public static native int skSummaryTerminationType(long s);
  public static native String skSummaryMessage(long s);
  public static native String skSummaryBriefReport(long s);
  public static native String skSummaryFullReport(long s);
  public static native double skSummaryIterationField(long s, int iteration, int field);
  public static native int skSolve(long options, long problem, long summary);
  public static native int skRotationApply(int op, boolean rowMajor, int jetDim, double[] in, int n, double[] out);
}
