"""Line-by-line mirrors of the reference's example programs
(examples/src/main/scala/org/somelightprojections/skeres/examples/) on top of skeres_amd."""
