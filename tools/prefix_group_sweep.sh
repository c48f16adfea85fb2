cd "$(dirname "$0")/.."
show() { python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ms_per_step %.3f  cholesky %.3f' % (d['ms_per_step'], d['phases_ms_per_step']['cholesky']))
except Exception as e:
    print('   failed:', e)"; }
for wl in ladybug-1723-156502 venice-1778-993923; do
B="bench.py --workload $wl --steps 12 --warmup 3 --cpu-iters 0 --no-c5 --no-alone"
for g in 2 3 4; do
for mt in 24 16 32; do
echo "== $wl: prefix group $g, chain max trailing $mt"; SK_CHAIN_PREFIX_GROUP=$g SK_CHAIN_MAX_TRAILING=$mt python3 $B 2>/dev/null | show
done
done
done
