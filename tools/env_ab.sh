#!/bin/bash
# bench.py's replayed train and search step under several values of ONE environment switch, on one box:
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/env_ab.sh <tag> SENAS_WGRAD_LAG 0 1 2 3'
set -e -o pipefail
tag=$1; var=$2; shift 2
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  (cd $root && env $var=$v python3 bench.py --steps 60 --lp-steps 0 --search-steps 60 --no-cpu-baseline > $out/bench_$v.log 2> $out/bench_$v.err) || { echo "$var=$v: bench failed"; tail -5 $out/bench_$v.err; continue; }
  tail -1 $out/bench_$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d.get('search_step',{}); print('$var=$v', 'train ms/step', d.get('ms_per_step'), 'search ms/step', s.get('ms_per_step'), 'gate', d.get('parity_gate',{}).get('passed'))" | tee -a $out/summary.txt
done
echo done
