#!/bin/bash
# scripts/isa.sh OUT.s [extra hipcc flags]: device ISA of the product library's kernels with the Makefile's flags (hipcc -S --cuda-device-only):
# the render kernels (rt_frame.hip) followed by the probe kernels (rt_probes.hip);
# scripts/isa.sh -k KERNEL_MANGLED_PREFIX IN.s: one kernel's listing out of it
if [ "$1" = "-k" ]; then
  L=$(grep -n "^$2.*:" "$3" | head -1 | cut -d: -f1)
  awk -v s=$L 'NR>=s{print} NR>s && /^\.Lfunc_end/{exit}' "$3"
  exit 0
fi
out=$1; shift
cd "$(dirname "$0")/../sycl-ray-tracer_amd/csrc"
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-parameter --offload-arch=gfx950 \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -mllvm -amdgpu-sched-strategy=max-ilp -Wno-unused-result \
  -fno-slp-vectorize "$@" -S --cuda-device-only rt_frame.hip -o "$out" 2>/dev/null
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-parameter --offload-arch=gfx950 \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -mllvm -amdgpu-sched-strategy=max-ilp -Wno-unused-result \
  -fno-slp-vectorize "$@" -S --cuda-device-only rt_probes.hip -o "$out.probes" 2>/dev/null && cat "$out.probes" >> "$out" && rm -f "$out.probes"
