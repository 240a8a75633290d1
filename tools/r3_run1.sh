#!/bin/bash
# round 3, GPU call: the whole GPU suite (parity margins recorded), the bf16 convolution probe, then the search step's HBM
# counters (separate --pmc passes)
set -e -o pipefail
root=$(pwd)
out=$root/gpurun_out/r3a
mkdir -p $out
timeout -k 10 400 python tools/bf_conv_check.py --time > $out/bf_check.log 2>&1 || { tail -30 $out/bf_check.log; exit 1; }
cat $out/bf_check.log
if ! python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; then tail -60 $out/pytest.log; exit 1; fi
tail -3 $out/pytest.log
cp gpurun_out/parity_margins.json $out/ || true
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/spF -o p -- python3 $root/tools/search_profile.py 3 > $out/spF.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/spW -o p -- python3 $root/tools/search_profile.py 3 > $out/spW.log 2>&1
cd $root
python3 tools/pmc_traffic.py $out/spF $out/spW $out/pmc_traffic_search.json > $out/pmc_traffic_search.txt
rm -rf $out/spF $out/spW
echo done
