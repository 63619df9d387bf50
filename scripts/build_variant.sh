#!/bin/bash
# build_variant.sh NAME -DFOO=1 ...  ->  sycl-ray-tracer_amd/csrc/build/librt_var_NAME.so (A/B kernel tuning; select with RT_MI355X_LIB).
# HIPX="<flags>": extra flags for the HIP translation units only (e.g. -mllvm options of the AMDGPU backend).
# Every source of the library is compiled with the extra flags (the builder and the GPU LBVH read some of the same constants).
set -e
cd "$(dirname "$0")/../sycl-ray-tracer_amd/csrc"
name=$1; shift
mkdir -p build
COMMON="-std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-parameter"
HIPF="$COMMON $HIPX --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -mllvm -amdgpu-sched-strategy=${SCHED:-max-ilp} -Wno-unused-result -fno-slp-vectorize"
/opt/rocm/bin/hipcc $COMMON -x c++ "$@" -c scene_build.cpp -o build/var_${name}_scene.o &
for u in lbvh_gpu rt_abi rt_comm rt_probes; do /opt/rocm/bin/hipcc $HIPF "$@" -c $u.hip -o build/var_${name}_$u.o & done
/opt/rocm/bin/hipcc $HIPF "$@" -c rt_frame.hip -o build/var_${name}_rt_frame.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o build/librt_var_$name.so build/var_${name}_*.o
echo built build/librt_var_$name.so
