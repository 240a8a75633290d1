#!/bin/bash
# round 3: bf16 convolution probe (graph-timed), then the search step's HBM counters (separate --pmc passes, eager launches)
set -e -o pipefail
root=$(pwd)
out=$root/gpurun_out/r3b
mkdir -p $out
timeout -k 10 400 python tools/bf_conv_check.py --time > $out/bf_check.log 2>&1 || { tail -30 $out/bf_check.log; exit 1; }
cat $out/bf_check.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/spF -o p -- python3 $root/tools/search_profile.py 2 --eager > $out/spF.log 2>&1 || { tail -5 $out/spF.log; exit 1; }
timeout -k 10 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/spW -o p -- python3 $root/tools/search_profile.py 2 --eager > $out/spW.log 2>&1 || { tail -5 $out/spW.log; exit 1; }
cd $root
python3 tools/pmc_traffic.py $out/spF $out/spW $out/pmc_traffic_search.json > $out/pmc_traffic_search.txt
rm -rf $out/spF $out/spW
echo done
