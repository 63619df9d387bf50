// rt_knobs.h — tuning knobs and test hooks that read the environment exist in DEVELOPER builds only.
//
// The product library (librt_mi355x.so) reads three environment variables, all documented in include/rt_mi355x.h: GPU_MAX_HW_QUEUES (the
// host's HIP setting, read once), RT_PROFILE_KERNELS and RT_KERNEL_STATS (diagnostics). Everything else a sweep script or a test may want
// to turn — the BVH builder's parameters, the wavefront schedule's initial value, occupancy and claim sizes, an injected allocation
// failure — goes through dev_knob(), which is getenv() in `make dev` (-DRT_DEVELOPER_KNOBS -> librt_mi355x_dev.so) and a constant
// nullptr otherwise: the names do not even reach the product binary. The public control is the API (rt_renderer_set_schedule).
#pragma once
#include <cstdlib>

namespace rt {

inline const char* dev_knob(const char* name) {
#ifdef RT_DEVELOPER_KNOBS
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

} // namespace rt
