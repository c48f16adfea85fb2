#!/bin/bash
# Developer tool: ms per LM iteration of the Ladybug-1723 solve under the knobs of the dissected factorisation.
run() { echo "== $*"; env "$@" timeout -k 10 120 python tools/explore_c3.py --timing 0 --iters 8 --perturb 1e-2 1e-1 1e-1 2>&1 | grep -E "iterations:|dissected fact|timed out|plan:|secondary|combination" | tail -5 ; }
export SK_DEBUG_QUEUES=1 SK_DISSECT_TIMING=1 SK_DISSECT_AT=850
run SK_X=0
run SK_DISSECT_B_PANEL=0
run SK_DISSECT_B_PANEL=1
run SK_DISSECT_B_PANEL=2
run SK_DISSECT_B_PANEL=3
run SK_DISSECT_B_PANEL=0 SK_DISSECT_B_BULK=0
run SK_DISSECT_B_PANEL=1 SK_DISSECT_B_BULK=0
