#!/bin/bash
# swap_bench.sh "<bench args>" name1 name2 ...: bench with the built library, then with tools/ab/libf2nerf_hip_NAME.so
# swapped in (on the GPU box's copy of the tree only)
args=$1; shift
run() { timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline $args 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-10s %8.3f M/s %8.3f ms ' % ('$1', d['value']/1e6, d['ms_per_step']), {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()})"; }
cp f2-nerf_amd/lib/libf2nerf_hip.so /tmp/lib_orig.so
run built || exit 1
for n in "$@"; do cp tools/ab/libf2nerf_hip_$n.so f2-nerf_amd/lib/libf2nerf_hip.so; run $n || exit 1; done
cp /tmp/lib_orig.so f2-nerf_amd/lib/libf2nerf_hip.so
