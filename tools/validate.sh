#!/bin/bash
# Development tool (GPU box): the driver's three steps - suite, smoke, bench - then the profile passes behind profiles/round6_*
# and the SURVEY 8(d) sweep, on one box.  usage: gpurun -- 'bash tools/validate.sh [tag] [steps...]'
#   steps (default: all): tests smoke bench profiles suite
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r6}; shift
STEPS=${*:-tests smoke bench profiles suite}
for s in $STEPS; do
  case $s in
    tests)    ( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 ) > gpurun_out/${TAG}_tests.log 2>&1; tail -22 gpurun_out/${TAG}_tests.log ;;
    smoke)    python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/${TAG}_smoke.log ;;
    bench)    ( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/${TAG}_bench.log 2> gpurun_out/${TAG}_bench.err
              cp bench_last.json gpurun_out/${TAG}_bench_last.json; tail -3 gpurun_out/${TAG}_bench.err; cut -c1-1500 gpurun_out/${TAG}_bench.log ;;
    profiles) bash tools/round6_profiles.sh > gpurun_out/${TAG}_profiles.log 2>&1; tail -12 gpurun_out/${TAG}_profiles.log ;;
    suite)    timeout 1200 python tools/suite.py > gpurun_out/${TAG}_suite.log 2>&1; tail -3 gpurun_out/${TAG}_suite.log ;;
  esac
done
