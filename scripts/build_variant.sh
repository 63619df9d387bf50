#!/bin/bash
# build_variant.sh NAME -DFOO=1 ...  ->  sycl-ray-tracer_amd/csrc/build/librt_var_NAME.so (A/B kernel tuning; select with RT_MI355X_LIB)
set -e
cd "$(dirname "$0")/../sycl-ray-tracer_amd/csrc"
name=$1; shift
mkdir -p build
make -s build/scene_build.o build/lbvh_gpu.o
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt \
    -fno-gpu-flush-denormals-to-zero -mllvm -amdgpu-sched-strategy=max-ilp -Wno-unused-result -fno-slp-vectorize "$@" -c rt_mi355x.hip -o build/var_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/librt_var_$name.so build/var_$name.o build/scene_build.o build/lbvh_gpu.o
echo built build/librt_var_$name.so
