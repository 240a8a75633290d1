#!/bin/bash
# The round's evidence set, collected in one gpurun call (see profiles/README.md for what each file is):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/collect_evidence.sh r4'
# Writes under gpurun_out/<tag>/; copy the summaries into profiles/ by hand (tools/README.md).
set -e -o pipefail
tag=${1:-rX}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --search-steps 0 --lp-steps 0 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o st -- $B --steps 20 --warmup 5 > $out/stats.log 2>&1
SENAS_TRACE_MARKER=1 rocprofv3 --kernel-trace --output-format csv -d $out/train -o tr -- $B --steps 20 > $out/train.log 2>&1
SENAS_TRACE_MARKER=1 rocprofv3 --kernel-trace --output-format csv -d $out/trainx3 -o tr -- $B --steps 20 --profile-math bf16x3 > $out/trainx3.log 2>&1
SENAS_TRACE_MARKER=1 rocprofv3 --kernel-trace --output-format csv -d $out/search -o tr -- python3 $root/tools/search_profile.py 10 > $out/search.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmcM -o p -- $B --steps 3 --warmup 1 > $out/pmcM.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmcMx3 -o p -- $B --steps 3 --warmup 1 --profile-math bf16x3 > $out/pmcMx3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmcF -o p -- $B --steps 3 --warmup 1 > $out/pmcF.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmcW -o p -- $B --steps 3 --warmup 1 > $out/pmcW.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmcFx3 -o p -- $B --steps 3 --warmup 1 --profile-math bf16x3 > $out/pmcFx3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmcWx3 -o p -- $B --steps 3 --warmup 1 --profile-math bf16x3 > $out/pmcWx3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/spF -o p -- python3 $root/tools/search_profile.py 2 --eager > $out/spF.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/spW -o p -- python3 $root/tools/search_profile.py 2 --eager > $out/spW.log 2>&1
cd $root
python3 tools/trace_by_grid.py $out/train --steady 20 > $out/train_steady.txt
python3 tools/trace_by_grid.py $out/trainx3 --steady 20 > $out/train_bf16x3_steady.txt
python3 tools/trace_by_grid.py $out/search --steady 10 > $out/search_steady.txt
python3 tools/aten_in_step.py $out/search 10 > $out/search_foreign_kernels.txt
python3 tools/pmc_mfma.py $out/pmcM $out/pmc_mfma.json > $out/pmc_mfma.txt
python3 tools/pmc_mfma.py $out/pmcMx3 $out/pmc_mfma_bf16x3.json > $out/pmc_mfma_bf16x3.txt
python3 tools/pmc_traffic.py $out/pmcF $out/pmcW $out/pmc_traffic.json > $out/pmc_traffic.txt
python3 tools/pmc_traffic.py $out/pmcFx3 $out/pmcWx3 $out/pmc_traffic_bf16x3.json > $out/pmc_traffic_bf16x3.txt
# (search_profile.py --eager: 2 warm-up + 2 timed steps, all launched eagerly: 4 steps in the counters)
python3 tools/pmc_traffic.py $out/spF $out/spW $out/pmc_traffic_search.json --all --steps 4 > $out/pmc_traffic_search.txt
cp $out/stats/*kernel_stats.csv $out/kernel_stats.csv
# what really overlaps under replay (a tracing profiler serialises the hardware queues): device time stamps per cell
{ echo "# tools/lane_timeline.py search --serial (every cell on one stream, the runtime's graph replay)"; python3 tools/lane_timeline.py search --serial --steady 2>/dev/null | grep -v amdgpu;
  echo; echo "# tools/lane_timeline.py search (columns of up cells on lanes, lane scheduler)"; python3 tools/lane_timeline.py search --steady --segments 2>/dev/null | grep -v amdgpu; } > $out/search_by_level.txt
{ echo "# tools/lane_timeline.py train --serial"; python3 tools/lane_timeline.py train --serial --steady 2>/dev/null | grep -v amdgpu;
  echo; echo "# tools/lane_timeline.py train (lanes + lane scheduler)"; python3 tools/lane_timeline.py train --steady --segments 2>/dev/null | grep -v amdgpu; } > $out/train_by_level.txt
# kernel launches and kernel time per cell of the search step (serial schedule, cut at the stamp kernels), four cells launch by launch
rocprofv3 --kernel-trace -d $out/cellstrace -o cells --output-format csv -- python3 tools/lane_timeline.py search --serial --order $out/order.json > $out/cells_timeline.log 2>&1
{ echo "# tools/cell_kernels.py over rocprofv3 --kernel-trace -- python3 tools/lane_timeline.py search --serial --order order.json: launches and kernel time per cell"
  echo "# (serial schedule: one stream; pass 1 = architecture pass, pass 2 = weight pass, whose weight-gradient launches sit in line here and on their own lane under lanes)"
  python3 tools/cell_kernels.py $out/cellstrace $out/order.json --ordered down4 up13 up40 head; } > $out/search_cells.txt
rm -rf $out/cellstrace
# the bench line last: its `traffic` fields are read from the counter aggregates of THIS run
cp $out/pmc_traffic.json profiles/${tag}_pmc_traffic.json
cp $out/pmc_traffic_search.json profiles/${tag}_pmc_traffic_search.json
cp $out/search_steady.txt profiles/${tag}_search_steady.txt          # (search_step.roofline.by_family reads this table)
python3 bench.py > $out/bench.log 2> $out/bench.err
tail -1 $out/bench.log > $out/bench.json
# keep what travels back small: the raw traces are only needed for the aggregates above
rm -rf $out/train $out/trainx3 $out/search $out/stats $out/pmcM $out/pmcMx3 $out/pmcF $out/pmcW $out/pmcFx3 $out/pmcWx3 $out/spF $out/spW
echo done
