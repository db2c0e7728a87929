#!/bin/bash
# Host-side AddressSanitizer builds of the library and of the C++ test drivers (device code is not instrumented), into
# lio-slam_amd/asan/ (SAN=undefined builds with UBSan instead).  Run the drivers' tests against them on the GPU box with
#   NDT_TEST_EXE_DIR=asan ASAN_OPTIONS=protect_shadow_gap=0:detect_leaks=0 python -m pytest tests/test_gpu_facade.py tests/test_gpu_odometry.py tests/test_gpu_lio_pipeline.py -m gpu
set -e
cd "$(dirname "$0")/.."
mkdir -p lio-slam_amd/asan
( cd lio-slam_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 \
    -Xarch_host -fsanitize=${SAN:-address} -shared -o ../asan/libndt_mi355.so ndt_api.hip ndt_map.hip ndt_residual.hip ndt_pcl.hip ndt_group.hip -ldl )
for t in facade_test odometry_test lio_pipeline_test; do
  /opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=${SAN:-address} -Iinclude tests/cpp/$t.cpp -o lio-slam_amd/asan/$t -Llio-slam_amd/asan -lndt_mi355 \
    -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib
done
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=${SAN:-address} -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include tests/cpp/abi_walk.cpp -o lio-slam_amd/asan/abi_walk \
  -Llio-slam_amd/asan -lndt_mi355 -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib
