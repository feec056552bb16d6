#!/bin/bash
# What could ONE fused launch of SpixelNet's tail (conv0_1 -> pred_mask0) save at best?  Measured with ablation builds of the f16x3 kernel
# (tools/build_ablations.sh 4 16; results wrong by design, timing only), N = 64, tools/bench_conv.py:
#   conv0_1 without its output stores  +  pred_mask0 without its pixel DMA  =  a lower bound of the fused kernel (same MFMAs, same fragment
#   reads, same epilogue math; before the 1.31x / 1.12x of the recomputed halo rows and the 30-of-32-column tiles).
mkdir -p gpurun_out/r05
for rep in 1 2; do
  echo "== as built (pass $rep)"; python tools/bench_conv.py --only "16" --iters 30 2>&1 | grep "@256"
  echo "== no output stores (MX_ABL 16) (pass $rep)"; DISCO_HIP_LIB=tools/build/libdisco_abl16.so python tools/bench_conv.py --only "16" --iters 30 2>&1 | grep "@256"
  echo "== no pixel DMA after a workgroup's first chunk (MX_ABL 4) (pass $rep)"; DISCO_HIP_LIB=tools/build/libdisco_abl4.so python tools/bench_conv.py --only "16" --iters 30 2>&1 | grep "@256"
done
