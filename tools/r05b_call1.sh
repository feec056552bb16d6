set -x
mkdir -p gpurun_out/r05b
O=gpurun_out/r05b
timeout 120 tools/build/mfma_4x4x1_probe > $O/mfma_4x4x1_probe.txt 2>&1; echo "probe rc $?" >> $O/mfma_4x4x1_probe.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention or encoder" > $O/attn_tests.log 2>&1; tail -5 $O/attn_tests.log
for v in 0 1024; do echo "== DISCO_ATTN_MFMA=$v"; DISCO_ATTN_MFMA=$v timeout 600 python tools/operating_points.py --min-tokens 1024 2>&1 | grep " x "; done > $O/attn_mfma_ab.txt 2>&1
for v in 0 1024; do echo "== DISCO_ATTN_MFMA=$v"; DISCO_ATTN_MFMA=$v timeout 600 python tools/operating_points.py --min-tokens 1024 2>&1 | grep " x "; done >> $O/attn_mfma_ab.txt 2>&1
timeout 900 bash tools/epilogue_bound.sh > $O/epilogue_bound.txt 2>&1
timeout 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
