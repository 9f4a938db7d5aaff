#!/bin/bash
# usage: run_variants.sh GRADSCALE name1 name2 ...   (from repo root on the GPU box)
gs=$1; shift
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/r3/var_$v
  mkdir -p $out
  ( cd $GRAFT_REPO_ROOT/tools && timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $out -o p --output-format csv -- python3 ab_hash_bwd.py --config ${CFG:-c2} --points ${PTS:-train} --grad-scale $gs --other $v --only $v --reps ${REPS:-10} > $out/log.txt 2>&1 ) || exit 1
  python3 - $out $v <<'P'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'hash_bwd' in r['Name']:
        print("%-10s %-40s calls %4s avg %9.1f us"%(sys.argv[2], r['Name'].split('(')[1][-40:] if False else r['Name'][28:68], r['Calls'], float(r['AverageNs'])/1e3))
P
done
