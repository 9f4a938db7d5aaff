#!/bin/bash
# bench_matrix.sh "<opts A>" "<opts B>" ... : headline / terminating / c3 / pixel tiles for each option set
for o in "$@"; do for w in "" "--regime terminating" "--workload c3" "--workload c4"; do
  timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline $o $w 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d.get('kernels',{})
print('%-24s %-22s %8.3f M/s %8.3f ms  hash_bwd %.3f ms' % ('$o', '$w', d['value']/1e6, d['ms_per_step'], k.get('hash_bwd',{}).get('avg_ms',0)))" || exit 1
done; done
