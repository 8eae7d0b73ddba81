#!/bin/bash
# dev only (GPU box): round-robin the library variants of ffd_variants.sh through scripts/micro/ffn_opt_bench.py:  ffd_variants_run.sh "tag1 tag2" [reps] [opt] [vals]
tags=$1; reps=${2:-2}; opt=${3:-ffn_skew}; vals=${4:-1}
root=${GRAFT_REPO_ROOT:-/root/repo}
cp $root/probpose_code_amd/libprobpose_mi355x.so /tmp/lib_orig.so
for i in $(seq $reps); do for t in $tags; do
  cp $root/scripts/micro/build/lib_$t.so $root/probpose_code_amd/libprobpose_mi355x.so
  python $root/scripts/micro/ffn_opt_bench.py $opt "$vals" $t 2>/dev/null
done; done
cp /tmp/lib_orig.so $root/probpose_code_amd/libprobpose_mi355x.so
