#!/bin/bash
# round 5: the driver's three steps (suite, smoke, bench) + the profile passes behind profiles/round5_* on one box
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 ) > gpurun_out/r5h_tests.log 2>&1
tail -22 gpurun_out/r5h_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5h_smoke.log 2>&1; tail -2 gpurun_out/r5h_smoke.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r5h_bench.log 2> gpurun_out/r5h_bench.err
cp bench_last.json gpurun_out/r5h_bench_last.json
tail -3 gpurun_out/r5h_bench.err
bash tools/round5_profiles.sh > gpurun_out/r5h_profiles.log 2>&1
tail -12 gpurun_out/r5h_profiles.log
timeout 900 python tools/suite.py > gpurun_out/r5h_suite.log 2>&1; tail -3 gpurun_out/r5h_suite.log
