#!/bin/bash
# Developer tool: ms per LM iteration of the Ladybug-1723 solve under the knobs of the dissected factorisation.
run() { echo "== $*"; env "$@" timeout -k 10 120 python tools/explore_c3.py --timing 0 --iters 8 --perturb 1e-2 1e-1 1e-1 2>&1 | grep -E "iterations:|device ms|dissect|timed out|plan:" | tail -5 ; }
run SK_DISSECT_AT=0
run SK_DISSECT_TIMING=1
run SK_DISSECT_TIMING=1 SK_DISSECT_THREAD=0
run SK_DISSECT_TIMING=1 SK_DISSECT_AT=850
run SK_DISSECT_TIMING=1 SK_DISSECT_AT=950
run SK_DISSECT_TIMING=1 SK_DISSECT_AT=1000
run SK_DISSECT_TIMING=1 SK_DISSECT_AT=850 SK_DISSECT_B_STREAMS=plain
