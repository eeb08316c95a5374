#!/bin/bash
# Development tool: build a profiling variant of the library (per-segment s_memtime sums inside the tiled
# 256x128 grouped GEMM) into tools/ab/libhpc_amd_prof.so.  Run on the GPU box as
#   cp tools/ab/libhpc_amd_prof.so hpc-ops_amd/hpc/libhpc_amd.so && python tools/prof_tiled256.py
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Iinclude -Ihpc-ops_amd/csrc -DHPC_TILED256_PROFILE \
  -c hpc-ops_amd/csrc/group_gemm_tiled256.hip -o /tmp/group_gemm_tiled256_prof.o
objs=$(ls hpc-ops_amd/build/*.o | grep -v group_gemm_tiled256.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libhpc_amd_prof.so $objs /tmp/group_gemm_tiled256_prof.o -lpthread
echo built tools/ab/libhpc_amd_prof.so
