#!/bin/bash
# Collect the round's profile artefacts on the GPU box (run through gpurun from the repo root); outputs under gpurun_out/r06/.
# Counters are collected in their own passes (--kernel-trace only), as the guide prescribes.
set -x
R=$PWD
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/ks -o ks --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency --no-other-configs --pipeline 0 --micro 1 > $O/ks_bench.json 2> $O/ks.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-latency --no-other-configs --pipeline 0 --micro 1 > /dev/null 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o write --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-latency --no-other-configs --pipeline 0 --micro 1 > /dev/null 2> $O/write.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU -d $O/pmc_sq1 -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-latency --no-other-configs --pipeline 0 --micro 1 > /dev/null 2> $O/sq1.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d $O/pmc_sq2 -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-latency --no-other-configs --pipeline 0 --micro 1 > /dev/null 2> $O/sq2.err
python $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json $O/pmc_sq1
cp $O/pmc_traffic.json $R/profiles/r06_pmc_traffic.json      # bench.py reads traffic / mfma_busy from here (keyed by the source hash)
# the anchor study on the final kernels: H again (seconds), the oracle chunks the build container had not finished, the report
python $R/tools/anchor_study.py run --variants HGD --redo H --threads 32 --dir $O/anchor_study > $O/anchor_study_gpu.log 2>&1
timeout -k 5 ${ORACLE_SECONDS:-700} python $R/tools/anchor_study.py run --variants A --threads 32 --dir $O/anchor_study > $O/anchor_study_cpu.log 2>&1
python $R/tools/anchor_study.py report --dir $O/anchor_study --out $O/anchor_mismatch.json 2>&1 | grep -v "Warn\|allow_tf32" > $O/anchor_study_report.txt
cp $O/anchor_mismatch.json $R/profiles/r06_anchor_mismatch.json
python $R/bench.py --steps 20 --warmup 5 > $O/final_bench.json 2> $O/final_bench.err
python $R/tools/pmc_sum.py $O/pmc_sq1 conv3x3 > $O/conv_pmc.txt; python $R/tools/pmc_sum.py $O/pmc_sq2 conv3x3 >> $O/conv_pmc.txt
for c in 3 4 5a 5b; do python $R/bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline 2>> $O/named_configs.err; done > $O/named_configs.jsonl
python $R/tools/profile_layers.py > $O/final_layers.txt 2>&1
python $R/tools/operating_points.py > $O/operating_points.txt 2>&1
python $R/tools/profile_layers.py --batch 1 > $O/final_layers_n1.txt 2>&1
python $R/tools/kmeans_latency.py > $O/kmeans_latency.txt 2>&1
(cd $R && tools/build/mfma_4x4x1_probe) > $O/mfma_4x4x1_probe.txt 2>&1
{ echo "== attention_kernel (DISCO_ATTN_MFMA=0)"; DISCO_ATTN_MFMA=0 python $R/tools/attn_ab.py 2>&1 | grep tokens; echo "== attention_mfma_kernel (default from 1 024 tokens)"; python $R/tools/attn_ab.py 2>&1 | grep tokens; } > $O/attn_final.txt 2>&1
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/final_kernel_stats.csv
# keep the merged output small: drop the raw traces
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
ls -la $O
