# developer tool: the CU reservation of the bulk (SYRK) queues against the iteration time
for cfg in "4 2" "6 2" "6 4" "8 4" "8 6"; do
  set -- $cfg
  for w in venice-1778-993923 ladybug-1723-156502; do
    echo "== RESERVED_PER_XCD=$1 RESERVED_EARLY=$2 $w"
    SK_LA_RESERVED_PER_XCD=$1 SK_LA_RESERVED_EARLY=$2 timeout -k 10 200 python tools/explore_c3.py --workload $w --iters 10 --timing 0 2>&1 | grep "iterations:\|device ms"
  done
done
