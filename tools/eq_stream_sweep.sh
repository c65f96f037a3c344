for s in 1 16 64 128 200 255 256 300 384 512 1024; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --eq-only --ir-len 4096 --streams $s 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('streams $s: EQ', d['kernels_ms_per_step']['k_svf_cascade_tp'], 'ms  per 256 streams', round(d['kernels_ms_per_step']['k_svf_cascade_tp']*256/$s,3))"
done
