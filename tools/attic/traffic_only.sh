#!/bin/bash
# The two PMC passes behind profiles/r05_pmc_traffic.json alone (run through gpurun from the repo root): re-run after any change to the conv sources.
set -x
R=$PWD
O=$R/gpurun_out/r05
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-latency --no-other-configs --pipeline 0 --micro 1 > /dev/null 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o write --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-latency --no-other-configs --pipeline 0 --micro 1 > /dev/null 2> $O/write.err
python $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
