#!/bin/bash
# A/B of the stride-2 tile's four-images-per-weight-chunk mode (DISCO_CONV_NJ=0/1) on one box: output hashes, per-layer times, headline
mkdir -p gpurun_out/r05
out=gpurun_out/r05/nj_ab.txt
: > $out
for v in 0 1; do
  echo "== DISCO_CONV_NJ=$v: small batches" >> $out
  DISCO_CONV_NJ=$v timeout 300 python tools/small_batch_latency.py 2>&1 | grep -v amdgpu.ids >> $out
  echo "== DISCO_CONV_NJ=$v: layers at batch 64" >> $out
  DISCO_CONV_NJ=$v timeout 300 python tools/profile_layers.py --batch 64 2>&1 | grep -E "stage totals|conv launches|conv2_3.0|conv3_3.0|conv4_3.0|down1.conv.0|down2.conv.0|conv[1234]a" >> $out
done
for rep in 1 2; do for v in 0 1; do
  echo "== DISCO_CONV_NJ=$v: bench (pass $rep)" >> $out
  DISCO_CONV_NJ=$v timeout 600 python bench.py --no-latency --no-other-configs 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('result_checksum'))" >> $out
done; done
