"""Mirror of examples/.../SimpleBundleAdjuster.scala:126-155.

    python -m skeres_amd.examples.simple_bundle_adjuster <data_file.txt> [--recorded]

Reads a BAL text file (SimpleBundleAdjuster.scala:37-76), adds one residual block per observation
with a shared trivial loss, solves with DENSE_SCHUR and prints the full report.
--recorded: the functor is the generic body of examples/traced_functors.py, recorded once and interpreted on the
device, instead of the body registered in the device functor registry (what a user's own functor goes through)."""
import sys

import skeres_amd as sk
from skeres_amd import bal


def main(argv):
    sk.ceres.initGoogleLogging("SimpleBundleAdjuster")
    if len(argv) < 2:
        print("Usage: SimpleBundleAdjuster <data_file.txt>")
        return 1
    sys.stdout.write("Loading BalProblem from %s ..." % argv[1])
    bal_problem = bal.BalProblem.from_file(argv[1])
    print(" done")
    parameters = sk.RichDoubleArray.fromArray(bal_problem.parameters)
    cameras = parameters                                   # mutableCameras
    points = parameters.slice(9 * bal_problem.num_cameras)  # mutablePoints
    problem = sk.Problem()
    loss_function = sk.PredefinedLossFunctions.trivialLoss()
    # Create residuals for each observation in the bundle adjustment problem. The
    # parameters for cameras and points are added automatically.
    recorded = None
    if "--recorded" in argv[2:]:
        from .traced_functors import TracedSnavelyReprojectionError
        recorded = TracedSnavelyReprojectionError(0.0, 0.0)
    for i in range(bal_problem.num_observations):
        functor = recorded.withCaptured(*bal_problem.observations[i]) if recorded else sk.SnavelyReprojectionError(*bal_problem.observations[i])
        obs_cost = functor.toAutoDiffCostFunction()
        obs_camera = cameras.slice(9 * int(bal_problem.camera_index[i]))
        obs_point = points.slice(3 * int(bal_problem.point_index[i]))
        problem.addResidualBlock(obs_cost, loss_function, obs_camera, obs_point)
    options = sk.Solver.Options()
    options.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
    options.setMinimizerProgressToStdout(True)
    summary = sk.Solver.Summary()
    sk.ceres.solve(options, problem, summary)
    print(summary.fullReport())
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
