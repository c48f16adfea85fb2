/*
 * skeres_amd.h — C ABI of libskeres_amd.so, the MI355X (gfx950) replacement for
 * the SWIG/JNI module `ceres` + native libceres that fgcallari/skeres binds.
 *
 * Every entry point states the reference interface it replaces.  Paths are
 * relative to the reference repository:
 *   ceres.i
 *   CORE = core/src/main/scala/org/somelightprojections/skeres
 *   EX   = examples/src/main/scala/org/somelightprojections/skeres/examples
 *
 * Conventions:
 *   - plain pointers and sizes only; no C++ types, no exceptions cross the ABI;
 *   - functions returning `int` return an sk_status (0 == SK_OK) unless the
 *     comment says "boolean"; sk_last_error() holds the message of the last
 *     failure on the calling thread;
 *   - parameter memory is CALLER-OWNED host memory, identified by address, and
 *     is updated in place when sk_solve returns (README.md:51-55);
 *   - cost / loss objects are caller-owned; a Problem never frees them
 *     (CORE/Problem.scala:7-13);
 *   - one caller thread per problem / solver handle; sk_solve blocks;
 *   - there is NO CPU fallback: every compute entry point needs a gfx950 device
 *     and fails with SK_ERR_NO_DEVICE without one.
 */
#ifndef SKERES_AMD_H
#define SKERES_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sk_status {
  SK_OK = 0,
  SK_ERR_INVALID_ARGUMENT = 1, /* JVM side: require(...) -> IllegalArgumentException */
  SK_ERR_NO_DEVICE = 2,
  SK_ERR_HIP = 3,
  SK_ERR_UNSUPPORTED = 4,
  SK_ERR_EVALUATION_FAILED = 5,
  SK_ERR_COMM = 6
} sk_status;

/* ceres::LinearSolverType (ceres/types.h, %include'd at ceres.i:137);
 * used at EX/SimpleBundleAdjuster.scala:148, EX/CurveFitting.scala:121 */
typedef enum sk_linear_solver_type {
  SK_DENSE_NORMAL_CHOLESKY = 0,
  SK_DENSE_QR = 1,
  SK_SPARSE_NORMAL_CHOLESKY = 2, /* not implemented: SK_ERR_UNSUPPORTED */
  SK_DENSE_SCHUR = 3,
  SK_SPARSE_SCHUR = 4,    /* not implemented */
  SK_ITERATIVE_SCHUR = 5, /* not implemented */
  SK_CGNR = 6             /* not implemented */
} sk_linear_solver_type;

/* ceres::MinimizerType; EX/Powell.scala:78 */
typedef enum sk_minimizer_type { SK_LINE_SEARCH = 0, SK_TRUST_REGION = 1 } sk_minimizer_type;

/* ceres::TerminationType, reported through Solver::Summary */
typedef enum sk_termination_type {
  SK_CONVERGENCE = 0,
  SK_NO_CONVERGENCE = 1,
  SK_FAILURE = 2,
  SK_USER_SUCCESS = 3,
  SK_USER_FAILURE = 4
} sk_termination_type;

/* ceres::Ownership; CORE/Problem.scala:10-13 forces DO_NOT_TAKE_OWNERSHIP */
typedef enum sk_ownership { SK_DO_NOT_TAKE_OWNERSHIP = 0, SK_TAKE_OWNERSHIP = 1 } sk_ownership;

/* Device functor registry.  In the reference the functor
 * body is a JVM closure reached through the SWIG director upcall
 * (ceres.i:48); a GPU cannot call it, so each functor on the hot path is a
 * device template instantiated for double and Jet<N>, addressed by id, with
 * its captured doubles passed as `consts`. */
typedef enum sk_functor_id {
  SK_FUNCTOR_HOST_CALLBACK = 0,           /* sk_cost_function_new_callback */
  SK_FUNCTOR_SNAVELY_REPROJECTION = 1,    /* EX/SimpleBundleAdjuster.scala:79-119; consts (observedX, observedY) */
  SK_FUNCTOR_EXPONENTIAL_RESIDUAL = 2,    /* EX/CurveFitting.scala:92-98; consts (x, y) */
  SK_FUNCTOR_POWELL_F1 = 3,               /* EX/Powell.scala:14-21 */
  SK_FUNCTOR_POWELL_F2 = 4,               /* EX/Powell.scala:24-31 */
  SK_FUNCTOR_POWELL_F3 = 5,               /* EX/Powell.scala:34-42 */
  SK_FUNCTOR_POWELL_F4 = 6,               /* EX/Powell.scala:45-53 */
  SK_FUNCTOR_BINARY_SCALAR_COST = 7,      /* core/src/test/.../AutodiffCostFuntionSpec.scala:14-26; consts (a) */
  SK_FUNCTOR_BINARY_VECTOR3_COST = 8,     /* AutodiffCostFuntionSpec.scala:55-69; consts (a) */
  SK_FUNCTOR_TEN_PARAMETER_COST = 9,      /* AutodiffCostFuntionSpec.scala:111-119 */
  /* BASELINE.json config 5 (synthetic dense problem; no reference counterpart): one residual
   * r = tanh(a . x) - y over ONE parameter block of any size n, the row a regenerated on the
   * device from a counter-based generator; consts (seed, row index, y).  Added with
   * sk_problem_add_dense_rows. */
  SK_FUNCTOR_SYNTH_TANH_ROW = 10,
  SK_FUNCTOR_HELLO_WORLD = 11,            /* EX/HelloWorld.scala:11-14: r = 10 - x */
  SK_FUNCTOR_QUATERNION_ROTATION = 12,    /* r = R(q) p - t, q = (w, x, y, z) normalised; consts (p[3], t[3]); no reference
                                           * counterpart: a 4-parameter block for the local parameterizations */
  SK_FUNCTOR_TAPE = 13                    /* a recorded functor body: sk_cost_function_new_tape */
} sk_functor_id;

/* A recorded functor body ("tape").  CORE/CostFunctor.scala:40-51 lets the user write ANY residual as
 *   def apply[T: Field: Trig: NRoot: Order: ClassTag](x: Array[T]*): Array[T]
 * and the reference calls it back per residual block on the JVM (ceres.i:48).  The functors of the reference's own
 * examples have device bodies above; every other functor gets to the device by being run ONCE with a recording T
 * (the Scala side: a T whose Field / Trig / NRoot instances append an instruction and return a fresh register;
 * skeres_amd/tape.py is that recording T in Python) — the list of instructions is the tape, evaluated on the GPU per
 * residual block by forward-mode autodiff exactly as CORE/AutodiffCostFunction.scala:96-130 seeds and reads its Jets.
 * An instruction is five 32-bit integers: opcode, destination register, operands a, b, c.
 * An operand is kind << 28 | index: */
typedef enum sk_tape_operand_kind {
  SK_TAPE_REGISTER = 0,   /* a register written by an earlier instruction */
  SK_TAPE_PARAMETER = 1,  /* x(i)(j), numbered through the blocks: index = N(0) + ... + N(i-1) + j */
  SK_TAPE_CAPTURED = 2,   /* the index-th double the closure captures (observedX, observedY, ...), per residual block */
  SK_TAPE_CONSTANT = 3    /* the index-th literal of the body (tape_constants) */
} sk_tape_operand_kind;
typedef enum sk_tape_opcode {
  SK_TAPE_MOV = 0, SK_TAPE_ADD, SK_TAPE_SUB, SK_TAPE_MUL, SK_TAPE_DIV, SK_TAPE_NEG, SK_TAPE_SQRT, SK_TAPE_EXP, SK_TAPE_LOG,
  SK_TAPE_SIN, SK_TAPE_COS, SK_TAPE_TAN, SK_TAPE_ASIN, SK_TAPE_ACOS, SK_TAPE_ATAN, SK_TAPE_ATAN2 /* atan2(a, b) */, SK_TAPE_ABS,
  SK_TAPE_LT,     /* 1 if real(a) <  real(b) else 0: spire's Order[Jet] compares real parts (CORE/Rotation.scala:458) */
  SK_TAPE_LE,     /* 1 if real(a) <= real(b) else 0 */
  SK_TAPE_SELECT  /* real(a) != 0 ? b : c — a data-dependent `if` of the body, BOTH of whose arms are on the tape */
} sk_tape_opcode;

typedef struct sk_ptrvec sk_ptrvec;
typedef struct sk_loss_function sk_loss_function;
typedef struct sk_cost_function sk_cost_function;
typedef struct sk_local_parameterization sk_local_parameterization;
typedef struct sk_problem sk_problem;
typedef struct sk_options sk_options;
typedef struct sk_summary sk_summary;
typedef struct sk_solver sk_solver;
typedef int sk_residual_block_id; /* CORE/package.scala:13 ResidualBlockId */

/* ---- library ------------------------------------------------------------ */
const char* sk_version(void);
const char* sk_last_error(void);
/* ceres.i:131-135 initGoogleLogging(name): kept for call-site compatibility
 * (EX/SimpleBundleAdjuster.scala:128); records the program name only. */
void sk_init_logging(const char* name);
/* Number of visible gfx950 devices (0 when none); does not fail. */
int sk_device_count(void);

/* ---- DoubleArray: ceres.i:95-96 %array_class(double, DoubleArray) -------- */
double* sk_array_new(int n);                         /* new DoubleArray(n) */
void sk_array_free(double* a);                       /* DoubleArray.delete() */
double sk_array_getitem(const double* a, int i);     /* getitem; CORE/RichDoubleArray.scala:20 */
void sk_array_setitem(double* a, int i, double v);   /* setitem; CORE/RichDoubleArray.scala:27 */
/* DoubleArraySlice.get(buffer, start) == &buffer[start]; ceres.i:99-107,
 * CORE/RichDoubleArray.scala:52.  Non-owning interior pointer. */
double* sk_array_slice(double* buffer, int start);
/* Bulk forms of CORE/RichDoubleArray.scala:36-39 (copyFrom) and :65-69
 * (toArray), so a JVM caller crosses JNI once instead of once per element. */
void sk_array_copy_in(double* dst, const double* src, int n);
void sk_array_copy_out(const double* src, double* dst, int n);

/* ---- DoubleMatrix: ceres.i:113-125 -------------------------------------- */
int sk_matrix_is_null(double* const* matrix);        /* boolean */
double* sk_matrix_row(double* const* matrix, int i);

/* ---- StdVectorDoublePointer: ceres.i:82 (std::vector<double*>) ----------- */
sk_ptrvec* sk_ptrvec_new(void);
void sk_ptrvec_free(sk_ptrvec* v);
void sk_ptrvec_add(sk_ptrvec* v, double* p);         /* CORE/Problem.scala:24-25 */
int sk_ptrvec_size(const sk_ptrvec* v);
double* sk_ptrvec_get(const sk_ptrvec* v, int i);
void sk_ptrvec_set(sk_ptrvec* v, int i, double* p);  /* CORE/RichDoubleMatrix.scala:74-75 */
double** sk_ptrvec_to_pointer_pointer(sk_ptrvec* v); /* DoubleMatrix.toPointerPointer; NULL when empty */

/* ---- Rotation (CORE/Rotation.scala:63-522) ----------------------------------------------
 * The conversion / rotation functions device functors call (csrc/rotation.hpp), evaluated ON THE DEVICE for
 * `n` inputs at once.  jet_dim 0: `in` is [n][in_len] doubles, `out` [n][out_len].  jet_dim K in 1..4: every
 * value is a Jet, stored as (real, K infinitesimal parts): in [n][in_len][1+K], out [n][out_len][1+K] — what
 * the functions compute under automatic differentiation (RotationSpec.scala:459-562).  Quaternions are
 * (w, x, y, z); 3x3 matrices are 9 values, column-major unless row_major != 0 (the reference's
 * ColumnMajorMatrixAdapter3x3 / RowMajorMatrixAdapter3x3); Euler angles are (pitch, roll, yaw) in degrees.
 * Input layout of the two-argument functions: first argument, then second (q then pt; z then w; x then y). */
typedef enum {
  SK_ROT_ANGLE_AXIS_TO_QUATERNION = 0,         /* :72   3 -> 4 */
  SK_ROT_QUATERNION_TO_ANGLE_AXIS = 1,         /* :104  4 -> 3 */
  SK_ROT_ROTATION_MATRIX_TO_QUATERNION = 2,    /* :162  9 -> 4 */
  SK_ROT_ROTATION_MATRIX_TO_ANGLE_AXIS = 3,    /* :203  9 -> 3 */
  SK_ROT_ANGLE_AXIS_TO_ROTATION_MATRIX = 4,    /* :211  3 -> 9 */
  SK_ROT_EULER_ANGLES_TO_ROTATION_MATRIX = 5,  /* :269  3 -> 9 */
  SK_ROT_QUATERNION_TO_SCALED_ROTATION = 6,    /* :326  4 -> 9 */
  SK_ROT_QUATERNION_TO_ROTATION = 7,           /* :364  4 -> 9; the zero quaternion is SK_ERR_EVALUATION_FAILED (`require`) */
  SK_ROT_UNIT_QUATERNION_ROTATE_POINT = 8,     /* :393  4+3 -> 3 */
  SK_ROT_QUATERNION_ROTATE_POINT = 9,          /* :422  4+3 -> 3 */
  SK_ROT_QUATERNION_PRODUCT = 10,              /* :435  4+4 -> 4 */
  SK_ROT_CROSS_PRODUCT = 11,                   /* :441  3+3 -> 3 (the mathematical product; the reference's first component is a typo) */
  SK_ROT_DOT_PRODUCT = 12,                     /* :445  3+3 -> 1 */
  SK_ROT_ANGLE_AXIS_ROTATE_POINT = 13          /* :449  3+3 -> 3 */
} sk_rotation_op;
int sk_rotation_apply(int op, int row_major, int jet_dim, const double* in, int n, double* out);

/* ---- LossFunction: PredefinedLossFunctions, ceres.i:168-184 -------------- */
sk_loss_function* sk_loss_trivial(void);             /* trivialLoss(): rho(s) = s; caller frees (%newobject, ceres.i:160) */
sk_loss_function* sk_loss_huber(double a);           /* huberLoss(a), ceres.i:171 */
sk_loss_function* sk_loss_soft_l_one(double a);      /* softLOneLoss(a), ceres.i:172 */
sk_loss_function* sk_loss_cauchy(double a);          /* cauchyLoss(a), ceres.i:173 */
sk_loss_function* sk_loss_tukey(double a);           /* tukeyLoss(a), ceres.i:174 */
sk_loss_function* sk_loss_tolerant(double a, double b); /* tolerantLoss(a, b), ceres.i:175 */
/* composedLoss(f, g): rho(s) = f(g(s)); scaledLoss(rho, a): a * rho(s), rho may be NULL (= a * s), ceres.i:176-182.
 * The reference builds both with DO_NOT_TAKE_OWNERSHIP; here the arguments are COPIED, so they may be freed
 * at once.  Nesting deeper than 4 is SK_ERR_UNSUPPORTED (NULL + sk_last_error). */
sk_loss_function* sk_loss_composed(const sk_loss_function* f, const sk_loss_function* g);
sk_loss_function* sk_loss_scaled(const sk_loss_function* rho, double a);
/* ceres::LossFunction::Evaluate(sq_norm, rho[3]) for n squared norms: rho + 3*i receives rho(s_i), rho'(s_i), rho''(s_i).
 * Evaluated ON THE DEVICE by the code the solvers run (known-answer tests). */
int sk_loss_evaluate(const sk_loss_function* loss, const double* sq_norm, int n, double* rho);
void sk_loss_free(sk_loss_function* loss);

/* ---- LocalParameterization: PredefinedLocalParameterizations, ceres.i:186-210 -------------
 * identity(size), subset(size, constant_parameters), quaternion(), homogeneousVector(size) — the Ceres 1.x classes
 * the reference creates through SWIG for ceres::Problem::AddParameterBlock / SetParameterization, which its Problem
 * inherits (CORE/Problem.scala:16).  Global size at most 16.  Creation returns NULL (sk_last_error) on the argument
 * errors Ceres checks: subset indices out of range or duplicated, a homogeneous vector of size < 2.  The objects
 * are copied into a problem when set; free them any time after.
 * Plus(x, delta) and ComputeJacobian(x) (global x local, row-major) are evaluated ON THE DEVICE — the code the solver
 * runs — for n points at once: x [n][global], delta [n][local] (one unused value per point when local is 0). */
sk_local_parameterization* sk_local_parameterization_identity(int size);
sk_local_parameterization* sk_local_parameterization_subset(int size, const int* constant_parameters, int num_constant);
sk_local_parameterization* sk_local_parameterization_quaternion(void);          /* (w, x, y, z) */
sk_local_parameterization* sk_local_parameterization_homogeneous_vector(int size);
void sk_local_parameterization_free(sk_local_parameterization* p);
int sk_local_parameterization_global_size(const sk_local_parameterization* p);  /* LocalParameterization::GlobalSize */
int sk_local_parameterization_local_size(const sk_local_parameterization* p);   /* LocalParameterization::LocalSize */
int sk_local_parameterization_plus(const sk_local_parameterization* p, const double* x, const double* delta, int n, double* x_plus);
int sk_local_parameterization_compute_jacobian(const sk_local_parameterization* p, const double* x, int n, double* jacobian);

/* ---- CostFunction -------------------------------------------------------- */
/* AutoDiffCostFunctor.toAutoDiffCostFunction (CORE/CostFunctor.scala:44) for a
 * functor with a device body.  Validates like CostFunctor / SizedCostFunction
 * (CORE/CostFunctor.scala:31-34, CORE/SizedCostFunction.scala:7-13).  Returns
 * NULL on an unknown id or wrong `num_consts`. */
sk_cost_function* sk_cost_function_new_autodiff(int functor_id, const double* consts, int num_consts);

/* Host-callback cost function: the reference's director path
 * ceres::CostFunction::Evaluate(double const* const* parameters,
 *   double* residuals, double** jacobians) -> bool   (ceres.i:48; overridden
 * at CORE/AutodiffCostFunction.scala:74-78).  `jacobians` may be NULL and each
 * jacobians[i] may be NULL; block i is row-major num_residuals x block_sizes[i].
 * Returns boolean (0 == evaluation failed). */
typedef int (*sk_evaluate_fn)(void* user, double const* const* parameters, double* residuals,
                              double** jacobians);
sk_cost_function* sk_cost_function_new_callback(sk_evaluate_fn fn, void* user, int num_residuals,
                                                const int* block_sizes, int num_blocks);
/* AutoDiffCostFunctor.toAutoDiffCostFunction (CORE/CostFunctor.scala:44) for a functor WITHOUT a device body: its
 * recorded body (see sk_tape_opcode).  `instructions`: 5 * num_instructions integers; `output_operands`: one operand per
 * residual; `captured`: the doubles this closure captures (num_captured of them; sk_problem_add_residual_blocks_tape
 * passes them per block instead).  Validates sizes as CostFunctor / SizedCostFunction do (CORE/CostFunctor.scala:31-34)
 * and every instruction (known opcode, operands in range, registers written before they are read).  NULL on error. */
sk_cost_function* sk_cost_function_new_tape(int num_residuals, const int* block_sizes, int num_blocks,
                                            const int* instructions, int num_instructions,
                                            const double* tape_constants, int num_tape_constants,
                                            int num_registers, const int* output_operands,
                                            const double* captured, int num_captured);
void sk_cost_function_free(sk_cost_function* cf);
int sk_cost_function_num_residuals(const sk_cost_function* cf);             /* CostFunction.numResiduals() */
int sk_cost_function_num_parameter_blocks(const sk_cost_function* cf);      /* parameterBlockSizes().size() */
int sk_cost_function_parameter_block_size(const sk_cost_function* cf, int i);
/* CostFunction.evaluate (CORE/AutodiffCostFunction.scala:74-134): evaluates ONE
 * residual block.  Device functors run on the GPU (one lane).  Boolean result;
 * a negative value is an error (see sk_last_error). */
int sk_cost_function_evaluate(const sk_cost_function* cf, double const* const* parameters,
                              double* residuals, double** jacobians);

/* ---- Problem: CeresProblem + CORE/Problem.scala --------------------------- */
sk_problem* sk_problem_new(void);                    /* new Problem (Options: never owns cost/loss) */
void sk_problem_free(sk_problem* p);
/* CeresProblem.addResidualBlock(CostFunction, LossFunction, StdVectorDoublePointer)
 * (CORE/Problem.scala:20-27; the only overload left by ceres.i:53-70).
 * Parameter blocks are identified BY POINTER VALUE; first sighting registers a
 * block of the size the cost function declares; a later sighting with another
 * size is SK_ERR_INVALID_ARGUMENT.  `loss` may be NULL (== trivial). */
int sk_problem_add_residual_block(sk_problem* p, const sk_cost_function* cost,
                                  const sk_loss_function* loss, double* const* parameter_blocks,
                                  int num_parameter_blocks, sk_residual_block_id* id_out);
/* Bulk form of the setup loop EX/SimpleBundleAdjuster.scala:139-145: adds
 * `n` residual blocks of one device functor in one call.  consts is
 * n x num_consts row-major; parameter_blocks is n x num_blocks row-major. */
int sk_problem_add_residual_blocks(sk_problem* p, int functor_id, int n, const double* consts,
                                   const sk_loss_function* loss, double* const* parameter_blocks);
/* The same for a recorded functor (sk_cost_function_new_tape): n residual blocks of the body of `cost`, block b with
 * the captured doubles captured[b * num_captured ..] (NULL when the body captures none). */
int sk_problem_add_residual_blocks_tape(sk_problem* p, const sk_cost_function* cost, int n, const double* captured,
                                        const sk_loss_function* loss, double* const* parameter_blocks);
/* Bulk add of `num_rows` residual blocks of a dense-row functor (SK_FUNCTOR_SYNTH_TANH_ROW) that all
 * depend on the single parameter block x[0..n).  consts is num_rows x 3 row-major.  `loss` (NULL == trivial) applies to
 * every row of the call; the dense-rows solver takes ONE loss for all rows of a problem (SK_ERR_UNSUPPORTED otherwise). */
int sk_problem_add_dense_rows(sk_problem* p, int functor_id, int num_rows, const double* consts,
                              const sk_loss_function* loss, double* x, int n);
/* ceres::Problem::AddParameterBlock(values, size[, local_parameterization]), SetParameterization,
 * SetParameterBlockConstant / SetParameterBlockVariable (inherited by CORE/Problem.scala:16 from the SWIG-wrapped
 * ceres::Problem).  `parameterization` may be NULL (none / remove).  The minimiser then works in the tangent
 * space: Jacobian columns J * dPlus/ddelta, steps applied through Plus; a constant block takes no step.
 * Implemented for DENSE_QR / DENSE_NORMAL_CHOLESKY over residual blocks (every type), and under DENSE_SCHUR for what
 * bundle adjustment uses: constant camera / point blocks, identity, and subset (e.g. fixed intrinsics on the 9-block) —
 * a coordinate that is held constant has a zero Jacobian column and takes no step.  sk_solve reports
 * SK_ERR_UNSUPPORTED for quaternion / homogeneous-vector blocks under DENSE_SCHUR and for dense-row problems. */
int sk_problem_add_parameter_block(sk_problem* p, double* values, int size, const sk_local_parameterization* parameterization);
int sk_problem_set_parameterization(sk_problem* p, double* values, const sk_local_parameterization* parameterization);
int sk_problem_set_parameter_block_constant(sk_problem* p, double* values);
int sk_problem_set_parameter_block_variable(sk_problem* p, double* values);
int sk_problem_num_residual_blocks(const sk_problem* p);   /* Problem::NumResidualBlocks */
int sk_problem_num_parameter_blocks(const sk_problem* p);  /* Problem::NumParameterBlocks */
int sk_problem_num_parameters(const sk_problem* p);        /* Problem::NumParameters */
int sk_problem_num_residuals(const sk_problem* p);         /* Problem::NumResiduals */

/* ---- Solver.Options (setters are ceres.i:89-92 lowerCamelCase renames) ---- */
sk_options* sk_options_new(void);                    /* Ceres 1.x defaults, SURVEY.md §8a row a13 */
void sk_options_free(sk_options* o);
int sk_options_set_linear_solver_type(sk_options* o, int type);       /* setLinearSolverType */
int sk_options_set_minimizer_type(sk_options* o, int type);           /* setMinimizerType; TRUST_REGION only */
int sk_options_set_max_num_iterations(sk_options* o, int n);          /* setMaxNumIterations */
int sk_options_set_minimizer_progress_to_stdout(sk_options* o, int on); /* setMinimizerProgressToStdout */
int sk_options_set_function_tolerance(sk_options* o, double v);
int sk_options_set_gradient_tolerance(sk_options* o, double v);
int sk_options_set_parameter_tolerance(sk_options* o, double v);
int sk_options_set_initial_trust_region_radius(sk_options* o, double v);
int sk_options_set_max_trust_region_radius(sk_options* o, double v);
int sk_options_set_min_trust_region_radius(sk_options* o, double v);
int sk_options_set_min_relative_decrease(sk_options* o, double v);
int sk_options_set_min_lm_diagonal(sk_options* o, double v);
int sk_options_set_max_lm_diagonal(sk_options* o, double v);
int sk_options_set_jacobi_scaling(sk_options* o, int on);
int sk_options_set_max_num_consecutive_invalid_steps(sk_options* o, int n);
/* MI355X-side knobs (no reference counterpart) */
int sk_options_set_device(sk_options* o, int hip_device);             /* default: current device */
int sk_options_set_stream(sk_options* o, void* hip_stream);           /* default: a private stream */
/* Tuning of the dense Cholesky: `group` = depth of the trailing SYRK in 128-column blocks (K = 128*group).
 * <= 0 (default) leaves the plan to the library: groups of 3 for a full factorisation; with a sparse block
 * envelope, groups of 2 where the trailing SYRK is the long pole and single columns under a resident panel chain
 * (one workgroup that factors the diagonal blocks, per-column launches that hand it its operands through device
 * counters) where the serial chain is — DESIGN.md section 4.  The first solver of a process then spends about
 * 0.1 s choosing hardware queues for that chain.  An explicit `group` is used for every column, launch by launch
 * (then the envelope factorisation is bit-identical to the full one at the same `group`).
 * `lookahead` != 0 factors the next block-column group on its own stream next to the trailing SYRK of the
 * current one. */
int sk_options_set_cholesky_tuning(sk_options* o, int group, int lookahead);
/* DENSE_SCHUR: the reduced camera system of a bundle-adjustment problem is block-banded (cameras that share no
 * point give a zero block, and the Cholesky factor keeps the block envelope).  on != 0 (default): the dense
 * factorisation touches only the 128-blocks inside the envelope; the blocks it leaves out are exact zeros in the
 * full computation too, so at equal SYRK depth (sk_options_set_cholesky_tuning) the result is bit-identical to
 * on == 0, which factors every block. */
int sk_options_set_cholesky_envelope(sk_options* o, int on);
/* DENSE_SCHUR: two-way dissection of the camera sequence.  The block-banded reduced system of a camera sequence is a
 * serial chain of one 128-block column after the other; the cameras are split into a head, a separator and a tail (no
 * point seen from both head and tail), the head is eliminated front to back and the tail back to front, side by side
 * on two sets of queues, and the separator's system — plus both Schur complements — is factored last: the same
 * arithmetic in another elimination order (results agree with the undissected factorisation to rounding, not bit for
 * bit), with a chain about half as long.  ON: whenever a separator exists, with the library's own plan (no explicit
 * `group`), the envelope on and one process; OFF: never; AUTO (default): as OFF on one device — measured on MI355X the
 * two chains disturb each other on one chip more than the shorter chain gains — the model's
 * prediction is reported by sk_solver_stat. */
enum { SK_DISSECTION_AUTO = 0, SK_DISSECTION_ON = 1, SK_DISSECTION_OFF = 2 };
int sk_options_set_cholesky_dissection(sk_options* o, int mode);
/* DENSE_SCHUR: loop closures.  A camera sequence that revisits a place couples two distant windows of the band, and in
 * the band's own order every block column between the two windows joins the envelope.  The solver may instead number
 * the revisiting cameras BEHIND the band, as a border whose block rows are active from the first block column that
 * reaches them (and whose own block columns are factored last): the band keeps its width.  The result of the solve
 * does not depend on the order of the cameras inside the reduced system (Ceres' DENSE_SCHUR,
 * examples/.../SimpleBundleAdjuster.scala:147-152, orders it itself); another elimination order agrees to rounding,
 * not bit for bit.  AUTO (default): when the model of the factorisation's serial chain predicts a gain of 10 %;
 * ON: whenever the camera lists of the points show visits (jumps in the camera sequence) — tests, small problems;
 * OFF: never.  sk_solver_stat: "border_cameras", "border_gap", "border_model_us", "border_model_us_plain". */
/* Resident kernels.  By default the factorisation of the reduced system runs its chain-bound block columns under a RESIDENT
 * panel chain (one workgroup that factors the diagonal blocks as their updates land, column launches enqueued ahead of time
 * that wait for it) and the back-substitution is one resident launch — kernels that wait for other kernels of the same
 * process.  on == 0: this solver uses none — the same factorisation plan and the same arithmetic, one launch per step —
 * for processes in which kernels are serialised (hardware-counter collection under rocprofv3 --pmc) or in which the caller
 * does not want kernels that spin.  (A wait that gives up after its 1 s time-out has the same effect for the device from
 * then on, and says so on stderr.) */
int sk_options_set_resident_kernels(sk_options* o, int on);
/* Launch-bound problems (a reduced system of at most eight 128-blocks: BAL problem-49) replay their iteration as a hipGraph;
 * on == 0: every launch is enqueued directly. */
int sk_options_set_graph_replay(sk_options* o, int on);
/* Several ranks, SK_DISTRIBUTION_SEGMENTED / AUTO: cut the camera sequence into at most n segments (0, the default: at
 * most one per rank; n >= 2 otherwise). */
int sk_options_set_max_segments(sk_options* o, int n);
enum { SK_BORDER_AUTO = 0, SK_BORDER_ON = 1, SK_BORDER_OFF = 2 };
int sk_options_set_cholesky_border(sk_options* o, int mode);
/* DENSE_SCHUR: RETAINED POINTS.  The Schur complement of a point seen by k cameras is a dense k x k square of camera blocks; a
 * landmark that stays in view for hundreds of frames sets the height of the reduced system's block envelope over every block
 * column it spans (the Ladybug-1723-shaped problem: 5 of 156 502 points account for 80 % of the factorisation's flops).  Such
 * points are not eliminated: their three coordinates stay in the reduced system as three more rows behind the cameras — a
 * border in the sense of sk_options_set_cholesky_border — and the Schur complement is formed from the other points alone.  The
 * linear system, and therefore the LM step, is the same (Ceres' DENSE_SCHUR eliminates every point,
 * examples/.../SimpleBundleAdjuster.scala:147-152; which unknowns are eliminated first changes rounding only).  AUTO (default):
 * the widest tracks, as many as the model of the factorisation's serial chain says pay, when it predicts 10 % less than
 * eliminating everything; ON: the best count whatever the model says (tests, small problems); OFF: every point is eliminated.
 * max_points: at most this many — with ON: exactly this many, as far as there are tracks wider than a block — (a multiple of three is
 * used; 0: the library's limit, 1536).  Candidates: the widest tracks by span and by number of observations, and the tracks of
 * loop closures (a jump in the point's camera list) at their exact number.  A point with two residual blocks on one camera is never
 * retained.  Several ranks: the SEGMENTED distribution takes them with any number of segments — cut in two, the pseudo-cameras are
 * members of the one separator; cut in more, a border of the root and of every segment's front (AUTO tries two segments first and more
 * where its model says so; sk_options_set_max_segments caps the number) — and a retained point's observations are split over the
 * ranks by camera.  sk_solver_stat: "retained_points", "retained_model_us", "retained_model_us_without",
 * "model_us_two_segments_with_members". */
enum { SK_RETAINED_AUTO = 0, SK_RETAINED_ON = 1, SK_RETAINED_OFF = 2 };
int sk_options_set_retained_points(sk_options* o, int mode, int max_points);
/* Multi-GPU: this process is rank `rank` of `world` ranks,
 * one per GPU.  Points (e-blocks) are partitioned over ranks; the
 * normal-equation terms are summed with `allreduce` once per linear solve.
 * The hook must sum `count` doubles in device memory in place across all
 * ranks, ordered after prior work on `hip_stream`, and return 0.  It may return as soon as the collective is ENQUEUED on
 * `hip_stream` (the solver goes on enqueueing behind it and synchronises only where it reads a result: no host
 * synchronisation follows the hook since round 5); a hook that stages through the host synchronises the stream itself. */
typedef int (*sk_allreduce_fn)(void* user, double* device_buffer, size_t count, void* hip_stream);
int sk_options_set_distributed(sk_options* o, int rank, int world, sk_allreduce_fn allreduce,
                               void* user);
/* A ready-made `allreduce` over RCCL for callers that have no collective of their own (C, C++, the JVM: INTEGRATION.md
 * section 3) — the native counterpart of skeres_amd/dist.py's torch.distributed hook.  librccl.so is opened at run time
 * (dlopen): the library loads without it, and these calls then fail with SK_ERR_COMM / NULL + sk_last_error.
 *   rank 0:      sk_rccl_unique_id(id)  — 128 bytes (ncclUniqueId), passed to the other ranks by the caller's own means;
 *   every rank:  h = sk_allreduce_rccl_init(rank, world, id)  — ncclCommInitRank on the CURRENT HIP device — or
 *                h = sk_allreduce_rccl_create(comm) around an ncclComm_t the caller already has (not destroyed by _free);
 *                sk_options_set_distributed(o, rank, world, sk_allreduce_rccl_fn(), h);
 *   afterwards:  sk_allreduce_rccl_free(h).
 * The hook is an in-place ncclAllReduce(ncclDouble, ncclSum) on the stream the solver hands it. */
typedef struct sk_rccl sk_rccl;
int sk_rccl_unique_id(void* id128);
sk_rccl* sk_allreduce_rccl_init(int rank, int world, const void* id128);
sk_rccl* sk_allreduce_rccl_create(void* nccl_comm);
void sk_allreduce_rccl_free(sk_rccl* h);
long sk_allreduce_rccl_calls(const sk_rccl* h);
sk_allreduce_fn sk_allreduce_rccl_fn(void);
/* SEGMENTED: the camera sequence is dissected (sk_options_set_cholesky_dissection): rank 0's device eliminates the head and
 * its points, rank 1's the tail (further ranks replicate rank r mod 2 and add zeros to the sums); what is all-reduced per
 * iteration is the separator's system with both Schur complements — a few MB instead of the reduced system — and the two
 * serial chains run on two chips.  Chosen by AUTO when the model of the two chains predicts a gain; asked for explicitly it
 * is SK_ERR_UNSUPPORTED for a problem without a separator or with an explicit Cholesky grouping.
 * What a world > 1 does with its ranks.  SHARDED: points partitioned, one all-reduce of the reduced system
 * (its lower block triangle, packed) per linear solve.  REPLICATED: every rank solves the whole problem, no
 * collective.  AUTO (default): the solver times the all-reduce on the real buffer at set-up, estimates the
 * per-iteration work sharding would remove, and shards only when that pays (same decision on every rank). */
enum { SK_DISTRIBUTION_AUTO = 0, SK_DISTRIBUTION_SHARDED = 1, SK_DISTRIBUTION_REPLICATED = 2, SK_DISTRIBUTION_SEGMENTED = 3 };
int sk_options_set_distribution_mode(sk_options* o, int mode);
/* The caller may hand the solver the buffer the big all-reduce runs on (so a
 * torch.distributed / RCCL communicator can register it).  bytes must be >=
 * sk_reduce_buffer_bytes(problem, options): the whole lower block triangle, an upper bound — the solver packs and
 * reduces only the blocks inside the envelope (sk_solver_stat "allreduce_bytes"), at the start of the buffer. */
int sk_options_set_reduce_buffer(sk_options* o, void* device_ptr, size_t bytes);
size_t sk_reduce_buffer_bytes(const sk_options* o, const sk_problem* p);

/* ---- Solver.Summary ------------------------------------------------------- */
sk_summary* sk_summary_new(void);
void sk_summary_free(sk_summary* s);
double sk_summary_initial_cost(const sk_summary* s);
double sk_summary_final_cost(const sk_summary* s);
int sk_summary_num_iterations(const sk_summary* s);          /* iterations incl. iteration 0, as Ceres counts */
int sk_summary_num_successful_steps(const sk_summary* s);
int sk_summary_num_unsuccessful_steps(const sk_summary* s);
int sk_summary_termination_type(const sk_summary* s);
const char* sk_summary_message(const sk_summary* s);
const char* sk_summary_brief_report(const sk_summary* s);    /* Summary.briefReport(); EX/CurveFitting.scala:131 */
const char* sk_summary_full_report(const sk_summary* s);     /* Summary.fullReport(); EX/SimpleBundleAdjuster.scala:154 */
/* Per-iteration log (what minimizer_progress_to_stdout prints). field:
 * 0 cost, 1 cost_change, 2 gradient_max_norm, 3 step_norm, 4 relative_decrease,
 * 5 trust_region_radius, 6 step_is_valid, 7 step_is_successful */
int sk_summary_num_logged_iterations(const sk_summary* s);
double sk_summary_iteration_field(const sk_summary* s, int iteration, int field);
/* Device time per phase, seconds, summed over the solve (HIP events on the
 * solver's stream). phase: 0 jacobian_eval, 1 schur_assemble (or J^T J),
 * 2 cholesky (or QR), 3 back_substitute, 4 cost_eval, 5 allreduce, 6 total */
double sk_summary_phase_seconds(const sk_summary* s, int phase);
/* Summary.linear_solver_type_used / _given (ceres::Solver::Summary; SWIG exposes both through /root/reference ceres.i:186-210).
   They differ when DENSE_SCHUR was asked for on a problem without the 2-residual / 9- and 3-parameter block structure: the
   alternate, DENSE_QR, is used, as Ceres does for a Schur-type solver with nothing to eliminate. */
int sk_summary_linear_solver_type_used(const sk_summary* s);
int sk_summary_linear_solver_type_given(const sk_summary* s);

/* ---- solve ----------------------------------------------------------------- */
/* ceres.solve(options, problem, summary) — EX/SimpleBundleAdjuster.scala:152,
 * EX/CurveFitting.scala:127.  Blocking; parameters updated in place. */
int sk_solve(const sk_options* options, sk_problem* problem, sk_summary* summary);

/* Stepping form of the same loop, for benchmarks and JVM IterationCallback-style
 * drivers: create (uploads + iteration 0), step (ONE trust-region iteration),
 * finish (writes parameters back + fills the summary). */
sk_solver* sk_solver_create(const sk_options* options, sk_problem* problem);
/* The sk_status that goes with the last NULL returned by sk_solver_create — or by sk_allreduce_rccl_init — on the calling thread
 * (SK_OK after a success; both set it on every call): NULL alone cannot say whether the device is missing, the configuration
 * unsupported, the communicator down or an argument wrong. */
int sk_last_status(void);
void sk_solver_free(sk_solver* s);
/* Returns SK_OK and sets *done (boolean) when a termination test fired. */
int sk_solver_step(sk_solver* s, int* done);
int sk_solver_finish(sk_solver* s, sk_summary* summary);
/* Device seconds of kernel `name` accumulated since create, and its launch
 * count (HIP events around the launches when profiling is on).  on: 0 off,
 * 1 every named launch (diagnostic: the event packets slow the small kernels
 * of the panel chain), 2 only the dominant kernel "gemm_syrk". */
int sk_solver_set_kernel_timing(sk_solver* s, int on);
double sk_solver_kernel_seconds(const sk_solver* s, const char* name, int* launches);
/* Algorithmic flop count of the dense Cholesky's trailing updates per linear
 * solve (what roofline.achieved is computed from). */
double sk_solver_syrk_flops_per_solve(const sk_solver* s);
/* Bytes of the C tiles those launches read and write per linear solve (each 128 x 128 tile once in, once out per
 * launch): the algorithmic memory traffic of the trailing updates, next to which bench.py reports the measured one.
 * 0 for solvers that do not report it. */
double sk_solver_syrk_c_bytes_per_solve(const sk_solver* s);
/* SK_DISTRIBUTION_SHARDED or _REPLICATED as decided at sk_solver_create; with AUTO, *allreduce_seconds is the
 * measured all-reduce of the reduced system and *saved_seconds the estimated per-iteration work sharding removes
 * (either pointer may be NULL). */
int sk_solver_distribution(const sk_solver* s, double* allreduce_seconds, double* saved_seconds);
/* Named figures of the solver's plan, for benchmarks and reports (no reference counterpart: Ceres prints the like in
 * Summary::FullReport).  Returns SK_OK and sets *value, or SK_ERR_INVALID_ARGUMENT for a name this solver does not
 * report.  DENSE_SCHUR:
 *   "envelope_fill"         fraction of the lower-triangular 128-blocks of the reduced camera system inside the block
 *                           envelope that is factored (1 with sk_options_set_cholesky_envelope(o, 0))
 *   "camera_order"          0 first appearance, 1 memory order of the camera blocks, 2 reverse Cuthill-McKee
 *   "cholesky_flops_full"   n^3 / 3, n = 9 * cameras: SURVEY.md section 8(d)'s figure for phase C
 *   "cholesky_flops_plan"   flops of the factorisation as planned (potrf + TRSM + updates of the blocks inside the envelope)
 *   "cholesky_columns_resident"  block columns factored under the resident panel chain
 *   "allreduce_bytes"       bytes of the reduced system that travel in the per-iteration all-reduce of the sharded mode (the
 *                           lower block triangle inside the envelope), "allreduce_bytes_full_triangle" beside it
 *   "dissected"             1 when the camera sequence is dissected (sk_options_set_cholesky_dissection), with
 *   "dissection_head_cameras" / "dissection_separator_cameras" / "dissection_tail_cameras" and the model's
 *   "dissection_model_us_plain" / "dissection_model_us" (microseconds of factorisation it predicted either way)
 *   "chain_steps"           serial steps of the factorisation (block columns; two leaf fronts in lock-step count once)
 * every solver:
 *   "phase_seconds_<i>"     seconds accumulated so far in phase i (0 Jacobians, 1 Schur assembly, 2 Cholesky, 3 back-substitution,
 *                           4 candidate cost, 5 all-reduce): sk_summary_phase_seconds, readable between steps
 * dense rows (DENSE_NORMAL_CHOLESKY over one parameter block):
 *   "jtj_flops_algorithmic" m n (n + 1): SURVEY.md section 8(d)'s figure for J^T J (sk_solver_syrk_flops_per_solve counts
 *                           the padded 128 x 128 tiles the launch computes) */
int sk_solver_stat(const sk_solver* s, const char* name, double* value);

/* ---- multi-GPU sharding (host logic, no device needed) -----------------------
 * How sk_solve splits a bundle-adjustment-shaped problem over `world` ranks
 * (SURVEY.md §8e): cameras are the distinct blocks in parameter slot 0, points
 * those in slot 1, both numbered in first-appearance order.  Rank r owns the
 * points [cuts[r], cuts[r+1]) and every observation of them; the runs have
 * (nearly) equal sum of squared track lengths.  `cuts` has world+1 entries.
 * point_of_block[i] (optional, one entry per residual block) receives the point
 * number of residual block i.  Returns SK_ERR_UNSUPPORTED when the problem is
 * not bundle-adjustment shaped. */
int sk_problem_point_partition(const sk_problem* p, int world, int* cuts, int* num_cameras,
                               int* num_points, int* point_of_block);
/* The plan of the SEGMENTED distribution for this problem, from host data alone (no device needed; what every rank derives
 * for itself at set-up): the camera sequence cut into *num_segments segments (at most max_segments; 1 = not cut) with a
 * separator between neighbours.  camera_part_of_block[b] (may be NULL): the segment of residual block b's camera, or -k for a
 * camera of separator k (1 <= k < segments); point_owner_of_block[b] (may be NULL): the rank that owns block b's point.
 * forced != 0: as SK_DISTRIBUTION_SEGMENTED cuts (as many segments as the sequence allows); 0: as AUTO (the chain model's
 * choice).  No point is seen from two segments; a point's blocks all have one owner. */
int sk_problem_segment_plan(const sk_problem* p, int max_segments, int forced, int* num_segments, int* camera_part_of_block,
                            int* point_owner_of_block);
/* The order of the cameras inside the reduced system, and the border of loop-closure cameras (sk_options_set_cholesky_border),
 * as one process derives them at set-up — from host data alone, no device needed.  *num_border_cameras: cameras ordered into
 * the trailing border (0: none); camera_position_of_block[b] (may be NULL): position of residual block b's camera in the
 * reduced system (border cameras: the last *num_border_cameras positions); gap: the jump in a point's camera list that
 * separated visits; model_us / model_us_plain: the chain model's microseconds per factorisation with the order chosen /
 * with the best unbordered order; envelope_fill: 128-blocks inside the envelope over the lower block triangle. */
int sk_problem_border_plan(const sk_problem* p, int mode, int* num_border_cameras, int* camera_position_of_block, int* gap,
                           double* model_us, double* model_us_plain, double* envelope_fill);
/* The retained points (sk_options_set_retained_points) as one process chooses them at set-up — host data alone.  *num_retained:
 * how many (0: none); retained_of_block[b] (may be NULL): 1 when residual block b's point is retained; model_us /
 * model_us_without: the chain model's microseconds per factorisation with them / with every point eliminated (the loop-closure
 * border of `border_mode` in both). */
int sk_problem_retained_plan(const sk_problem* p, int mode, int max_points, int border_mode, int* num_retained, int* retained_of_block,
                             double* model_us, double* model_us_without);

/* ---- inputs of BASELINE.json config 5 (utility) ----------------------------------
 * y_out[i] = tanh(a_i . x_star) for the generated rows a_i of SK_FUNCTOR_SYNTH_TANH_ROW
 * (rows 0 .. m-1 of `seed`, n parameters), computed on the GPU: the planted targets of the
 * synthetic dense problem at sizes where a host loop over m * n generated coefficients is too slow. */
int sk_synth_dense_targets(double seed, int m, int n, const double* x_star, double* y_out);

/* ---- dense SPD solve (utility; the factorisation sk_solve uses) --------------
 * Solves A x = b on the GPU for a symmetric positive definite A (n x n,
 * row-major HOST memory, only the lower triangle is read) with the blocked
 * fp64-MFMA Cholesky.  Optional outputs: L (n x n row-major, lower, upper part
 * zero).  `group` is the SYRK depth in 128-column blocks (0 = default).
 * Returns SK_ERR_EVALUATION_FAILED when A is not positive definite. */
int sk_cholesky_solve(int n, const double* A, const double* b, double* x, double* L, int group);
/* The same with what sk_solve adds for a block-banded reduced camera system.  `last` (optional): the block envelope,
 * one entry per 128-block column of the padded matrix (ceil((n + 1) / 128) of them): last[c] >= c is the last block
 * row in which block column c of the FACTOR can be non-zero (non-decreasing; the final block row, which carries the
 * right-hand side, is always active) — blocks outside are neither read nor written, so A must be zero there.
 * automatic_plan != 0: the grouping is the library's (DENSE_SCHUR's default: resident panel chain where the serial
 * chain decides, launch-by-launch groups where the trailing SYRK does) and `group` <= 0 means 1; otherwise the
 * explicit `group`, launch by launch.  Known-answer tests of the plans sk_solve runs at full size. */
int sk_cholesky_solve_ex(int n, const double* A, const double* b, double* x, double* L, int group, const int* last,
                         int automatic_plan);
/* ... and with a BORDER (sk_options_set_cholesky_border): rows from border_begin_row on are the border — they may couple
 * with any column before them — and the rows before it a block-banded matrix.  The bordered block envelope is derived
 * from A's own non-zero 128-blocks: per block column the run of the band, and the border's block rows from the first
 * block column that reaches them (a border row, once reached, stays active, and so does every border row behind it). */
int sk_cholesky_solve_bordered(int n, const double* A, const double* b, double* x, double* L, int group,
                               int border_begin_row, int automatic_plan);

/* The same system solved by two-way dissection (what DENSE_SCHUR does with the reduced camera system of a camera
 * sequence when that pays — DESIGN.md section 4): rows [0, head) are eliminated front to back, rows [tail_begin, n) back
 * to front, side by side, each leaving its Schur complement on the separator [head, tail_begin), which is factored
 * last.  A(tail, head) must be zero (SK_ERR_INVALID_ARGUMENT otherwise).  The block envelopes of the three fronts are
 * derived from the non-zeros of A.  Known-answer tests of the dissected factorisation against a plain one. */
int sk_cholesky_solve_dissected(int n, const double* A, const double* b, double* x, int head, int tail_begin, int group,
                                int automatic_plan);

/* ... and by multi-way dissection: num_segments >= 2 runs of rows separated by num_segments - 1 separators
 * (cuts[2 (k - 1)], cuts[2 (k - 1) + 1]) = [begin, end) of separator k, ascending; rows behind a separator must not couple
 * with rows before it).  Every segment leaves its Schur complement on the separators next to it — a segment between two
 * separators as a partial factorisation whose border rows of the LEFT separator are active in every column (the spike) —
 * and the separators' block-tridiagonal system is factored last.  This is the arithmetic of the segmented distribution
 * over several GPUs (SK_DISTRIBUTION_SEGMENTED) run on one device: a known-answer test of it. */
int sk_cholesky_solve_segments(int n, const double* A, const double* b, double* x, int num_segments, const int* cuts, int group,
                               int automatic_plan);

#ifdef __cplusplus
}
#endif
#endif /* SKERES_AMD_H */
