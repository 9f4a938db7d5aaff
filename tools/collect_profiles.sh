#!/bin/bash
# Collects what profiles/ holds for one round on a GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh c2|c4c5|misc|lines OUTDIR
# bench JSON lines, rocprofv3 kernel-trace stats and the two PMC passes (FETCH_SIZE, WRITE_SIZE in
# separate runs, kernel-trace only next to them), each step under its own timeout.
set -eo pipefail
what=$1
out=$(realpath -m "$2")
mkdir -p "$out"
root=$(pwd)
export TMPDIR=/tmp
prof() {  # prof NAME KIND(stats|pmc COUNTER) -- bench args
  local name=$1 kind=$2; shift 2
  local counter=""
  if [ "$kind" = pmc ]; then counter=$1; shift; fi
  shift  # the "--"
  local dir=$out/$name
  rm -rf "$dir"
  if [ "$kind" = stats ]; then
    (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$dir" -o p -- python3 "$root/bench.py" "$@" > "$out/$name.log" 2>&1)
    cp "$(find "$dir" -name '*kernel_stats.csv' | head -1)" "$out/$name.kernel_stats.csv"
    grep '^{"metric"' "$out/$name.log" | tail -1 > "$out/$name.bench.json"
    rm -rf "$dir"
  else
    (cd /tmp && timeout -k 10 400 rocprofv3 --pmc "$counter" --kernel-trace --output-format csv -d "$dir" -o p -- python3 "$root/bench.py" "$@" > "$out/$name.log" 2>&1)
  fi
}
bench() {  # bench NAME args
  local name=$1; shift
  timeout -k 10 500 python3 bench.py "$@" > "$out/$name.log" 2>&1
  grep '^{"metric"' "$out/$name.log" | tail -1 > "$out/$name.json"
  python3 -c "import json,sys; d=json.load(open('$out/$name.json')); print('$name', d['value'], d['unit'], d['ms_per_step'])"
}
case $what in
  c2)
    bench bench_c2 --steps 20 --warmup 5
    prof c2_stats stats -- --steps 6 --warmup 2 --no-cpu-baseline --no-extras
    prof pmc_fetch_c2 pmc FETCH_SIZE -- --steps 2 --warmup 1 --no-cpu-baseline --no-extras
    prof pmc_write_c2 pmc WRITE_SIZE -- --steps 2 --warmup 1 --no-cpu-baseline --no-extras
    ;;
  c4c5)
    bench bench_c4 --workload c4 --steps 50 --warmup 10 --no-cpu-baseline --train-iters 50
    prof c4_stats stats -- --workload c4 --steps 50 --warmup 10 --no-cpu-baseline --graph-iters 0
    bench bench_c5 --workload c5 --steps 3 --warmup 1
    prof c5_stats stats -- --workload c5 --steps 2 --warmup 1 --no-cpu-baseline
    prof pmc_fetch_c5 pmc FETCH_SIZE -- --workload c5 --steps 1 --warmup 1 --no-cpu-baseline
    prof pmc_write_c5 pmc WRITE_SIZE -- --workload c5 --steps 1 --warmup 1 --no-cpu-baseline
    ;;
  lines)  # the bench lines again, once the PMC summaries of this build are in profiles/
    bench bench_c2 --steps 20 --warmup 5
    bench bench_c5 --workload c5 --steps 3 --warmup 1
    ;;
  misc)
    bench bench_c2_term --steps 10 --warmup 3 --no-cpu-baseline --regime terminating
    bench bench_c1_shape --levels 4 --samples 64 --steps 10 --warmup 3 --no-cpu-baseline
    bench bench_c1_chunk8192 --levels 4 --samples 64 --chunk 8192 --steps 5 --warmup 2 --no-cpu-baseline
    bench bench_c3 --workload c3 --steps 5 --warmup 2 --no-cpu-baseline
    bench bench_c2_tiles --pixel-tiles 8 --steps 10 --warmup 3 --no-cpu-baseline --no-extras
    bench bench_inference --steps 3 --warmup 1 --no-cpu-baseline --render-images 10
    bench bench_2rank_gloo_shared --gpus 2 --backend gloo --share-gpu --steps 3 --warmup 1 --no-cpu-baseline
    ;;
esac
ls "$out"
