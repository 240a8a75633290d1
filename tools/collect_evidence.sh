#!/bin/bash
# The round's evidence set, collected in one gpurun call (see profiles/README.md for what each file is):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/collect_evidence.sh r2'
# Writes under gpurun_out/<tag>/; copy the summaries into profiles/ by hand (tools/README.md).
set -e -o pipefail
tag=${1:-rX}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
python3 bench.py > $out/bench.log 2> $out/bench.err
tail -1 $out/bench.log > $out/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o st -- python3 $root/bench.py --steps 20 --warmup 5 --search-steps 0 --no-cpu-baseline > $out/stats.log 2>&1
SENAS_TRACE_MARKER=1 rocprofv3 --kernel-trace --output-format csv -d $out/train -o tr -- python3 $root/bench.py --steps 20 --search-steps 0 --no-cpu-baseline > $out/train.log 2>&1
SENAS_TRACE_MARKER=1 rocprofv3 --kernel-trace --output-format csv -d $out/search -o tr -- python3 $root/tools/search_profile.py 10 > $out/search.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmcM -o p -- python3 $root/bench.py --steps 3 --warmup 1 --search-steps 0 --no-cpu-baseline > $out/pmcM.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmcF -o p -- python3 $root/bench.py --steps 3 --warmup 1 --search-steps 0 --no-cpu-baseline > $out/pmcF.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmcW -o p -- python3 $root/bench.py --steps 3 --warmup 1 --search-steps 0 --no-cpu-baseline > $out/pmcW.log 2>&1
cd $root
python3 tools/trace_by_grid.py $out/train --steady 20 > $out/train_steady.txt
python3 tools/trace_by_grid.py $out/search --steady 10 > $out/search_steady.txt
python3 tools/pmc_mfma.py $out/pmcM $out/pmc_mfma.json > $out/pmc_mfma.txt
python3 tools/pmc_traffic.py $out/pmcF $out/pmcW $out/pmc_traffic.json > $out/pmc_traffic.txt
cp $out/stats/*kernel_stats.csv $out/kernel_stats.csv
# keep what travels back small: the raw traces are only needed for the aggregates above
rm -rf $out/train $out/search $out/stats $out/pmcM $out/pmcF $out/pmcW
echo done
