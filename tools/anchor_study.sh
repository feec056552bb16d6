#!/bin/bash
# The anchor study on a GPU box: eight CPU workers (the oracle, variant A, on every image; the oneDNN-off control B on the first 256 images
# of every set) next to one GPU worker (the HIP path H, the oracle's code on torch-ROCm in fp32 G and fp64 D), every one under its own
# time limit, results written chunk by chunk into gpurun_out/anchor_study/; then the report.   bash tools/anchor_study.sh [seconds per worker]
cd "$(dirname "$0")/.."
LIM=${1:-1100}
SETS=${SETS:-256x256:2048,512x512:512,768x512:512}
D=gpurun_out/anchor_study
mkdir -p $D
pids=()
for k in 0 1 2 3 4 5 6 7; do
  timeout -k 5 $LIM python tools/anchor_study.py run --sets $SETS --variants AB --limit B:16 --part $k/8 --threads 28 --dir $D > $D/cpu_$k.log 2>&1 &
  pids+=($!)
done
timeout -k 5 $LIM python tools/anchor_study.py run --sets $SETS --variants HGD --threads 16 --dir $D > $D/gpu.log 2>&1 &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
grep -h "^\[" $D/gpu.log | tail -3; for k in 0 7; do grep -h "^\[" $D/cpu_$k.log | tail -1; done
grep -il "error\|Traceback" $D/*.log
python tools/anchor_study.py report --sets $SETS --dir $D --out $D/anchor_mismatch.json 2>&1 | grep -v "Warn\|allow_tf32"
ls $D/*.npy | wc -l
