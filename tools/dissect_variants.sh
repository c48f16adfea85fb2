#!/bin/bash
# Developer tool: ms per LM iteration of the Ladybug-1723 solve under the knobs of the dissected factorisation.
run() { echo "== $*"; env "$@" timeout -k 10 120 python tools/explore_c3.py --timing 0 --iters 6 --perturb 1e-2 1e-1 1e-1 2>&1 | grep -E "iterations:|dissected fact|timed out" | tail -3 ; }
export SK_DISSECT_TIMING=1 SK_DISSECT_AT=850
run SK_X=0
run SK_DISSECT_B_SINGLE=1
run SK_DISSECT_B_SINGLE=1 SK_DISSECT_NO_FORK=1
run SK_DISSECT_NO_FORK=1
run SK_CHAIN_NO_SERVER_JOIN=1
run SK_CHAIN_NO_SERVER_JOIN=1 SK_DISSECT_B_SINGLE=1 SK_DISSECT_NO_FORK=1
run SK_DISSECT_B_SINGLE=1 SK_DISSECT_NO_FORK=1 SK_DISSECT_B_STREAMS=plain
run SK_DISSECT_AT=0 SK_CHAIN_NO_SERVER_JOIN=1
