#!/bin/bash
# swap_matrix.sh NAME: headline / terminating / c3 / c4 with the built library and with tools/ab/libf2nerf_hip_NAME.so
for w in "--steps 5 --warmup 2" "--regime terminating --steps 5 --warmup 2" "--workload c3 --steps 3 --warmup 1" "--workload c4 --steps 50 --warmup 10 --graph-iters 0"; do
  echo "== $w"; bash tools/debug/swap_bench.sh "$w" "$@" || exit 1
done
