set -x
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r05/gpu_tests.log 2>&1; tail -3 gpurun_out/r05/gpu_tests.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r05/smoke.log 2>&1; tail -2 gpurun_out/r05/smoke.log
timeout 2400 bash tools/collect_profiles.sh > gpurun_out/r05/collect.log 2>&1
tail -c 1500 gpurun_out/r05/final_bench.json
cat gpurun_out/r05/pmc_traffic.json | head -c 600
