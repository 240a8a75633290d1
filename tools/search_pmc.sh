#!/bin/bash
# HBM counters of the search step (separate --pmc passes over the eagerly launched step), then the default bench so that
# search_step.roofline.traffic is read from the fresh aggregate:
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/search_pmc.sh r3'
set -e -o pipefail
tag=${1:-rX}
root=$(pwd)
out=$root/gpurun_out/${tag}_search_pmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/spF -o p -- python3 $root/tools/search_profile.py 2 --eager > $out/spF.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/spW -o p -- python3 $root/tools/search_profile.py 2 --eager > $out/spW.log 2>&1
cd $root
python3 tools/pmc_traffic.py $out/spF $out/spW $out/pmc_traffic_search.json --all --steps 4 > $out/pmc_traffic_search.txt
rm -rf $out/spF $out/spW
cp $out/pmc_traffic_search.json profiles/${tag}_pmc_traffic_search.json
python3 bench.py > $out/bench.log 2> $out/bench.err
tail -1 $out/bench.log > $out/bench.json
head -30 $out/pmc_traffic_search.txt
echo done
