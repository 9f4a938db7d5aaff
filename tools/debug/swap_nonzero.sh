#!/bin/bash
# swap_nonzero.sh "<bench args>" name...: kernels.hash_bwd / hash_bwd_nonzero of the default line per library
args=$1; shift
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 $args 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernels']; print('%-10s %8.3f M/s  hash_bwd %.3f  nonzero %.3f' % ('$1', d['value']/1e6, k['hash_bwd']['avg_ms'], k['hash_bwd_nonzero']['avg_ms']))"; }
cp f2-nerf_amd/lib/libf2nerf_hip.so /tmp/lib_orig.so
run built || exit 1
for n in "$@"; do cp tools/ab/libf2nerf_hip_$n.so f2-nerf_amd/lib/libf2nerf_hip.so; run $n || exit 1; done
cp /tmp/lib_orig.so f2-nerf_amd/lib/libf2nerf_hip.so
