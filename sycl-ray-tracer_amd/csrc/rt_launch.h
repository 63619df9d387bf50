// rt_launch.h — what the host side and the kernels share about a launch: the plain structs passed as kernel arguments (ray queues, the
// dynamic queue, the megakernel's frame constants and pixel slices) and the block / occupancy constants the grids are sized with. No
// __global__ function here: every translation unit of the library may include it (the kernels themselves: rt_kernels.h, rt_probe_kernels.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "rt_types.h"

namespace rt {


// SoA ray queue: 34 bytes per ray in four coalesced planes (the reference's Buffers pads the same
// state to 44 bytes: src/render_wavefront.hpp:10-38).
//   org_id : (org.x, org.y, org.z, bits(pixel id))           16 B
//   a      : halves (dir.x, dir.y, dir.z, att.r)              8 B
//   b      : halves (att.g, att.b, rad.r, rad.g)              8 B
//   c      : half   (rad.b)                                   2 B
struct QueueDev {
    float4* org_id;
    uint2* a;
    uint2* b;
    uint16_t* c;
};


// threads per workgroup and waves per SIMD of the persistent traversal kernels (k_megakernel, k_wf_finish): the sweeps are in EXPERIMENTS.md
constexpr uint32_t kMegaBlock = 512;   // threads per workgroup: 8 independent waves share one LDS copy of the top of the BVH
constexpr uint32_t kMegaWaves = 6;     // waves per SIMD the kernels are compiled for (80 VGPRs): 5 -> 4.83, 6 -> 5.05, 8 (spills) -> 3.67 Grays/s

// Pixel slices (round 5). A pixel's samples are one sequential chain — its RNG word runs on from sample to sample
// (src/render_megakernel.cpp:146-153) — and a lane that takes a pixel is busy for ~520 rays, ~20 ms of the 1080p atrium frame: when the
// cursor reaches the frame's last pixel every lane is somewhere inside such a chain, and the frame's last fifth is a drain in which the
// machine empties lane by lane (profiles/r04_wave_stats.txt: queue empty after 105.7 of 129.9 ms). The chain has to stay sequential, but
// not on ONE lane: it is cut into a few SLICES of decreasing length (e.g. 45 + 15 + 4 samples), the cursor hands out all first slices,
// then all second ones, and so on, and the lane that finishes a slice stores the pixel's state — colour sum and RNG word, 32 bytes —
// for whichever lane takes the next slice. The frame then drains over the length of the LAST slice. Per pixel the operations and their
// order are untouched: same RNG sequence, same additions in the same order, bit-identical frame.
//   state : per pixel of the tile four 8-byte granules {value, tag} — sum r, g, b and the RNG word — written as two write-through (sc1)
//           16-byte stores and read with two sc1 loads: "the data is the flag" (cdna_hip_programming.md §6, Guideline 16, R2). A reader
//           takes the state only if ALL FOUR tags say "written for the slice of this frame that starts with sample s" (tag0 + s; tag0
//           grows by more than spp from frame to frame, the buffer is never cleared): no flag, no fence, no ordering between the stores.
//   order : slot q of the cursor = slice q / n_slots of pixel slot q % n_slots, so slice j of a pixel is handed out a whole sweep of the
//           cursor after slice j - 1 was; the host sizes the slices so that a sweep takes longer than the slice before it
//           (mega_slices). A lane whose slice is not ready yet (never seen at full size; small tiles) looks again in its wave's next
//           shading round. No lane ever waits for a LATER slot: nothing can deadlock.
//   bound : all the same, a wave that holds ONLY waiting lanes counts its looks (one every ~3 us) and gives up after `wait_cap` of them — seconds —
//           with an error word the host turns into RT_ERR_HIP: a logic error must become an error code, not a wave that never leaves.
//   cuts  : a lane knows its slice by the sample it starts with (taken from bound[] when the slot is claimed) and finds the slice's end
//           in a 64-bit mask: bit k set = a slice ends before sample k << shift (shift = 0 up to 64 spp).
constexpr uint32_t kMaxSlices = 8;
struct SliceDev {
    void* state;             // four 8-byte granules per pixel of the tile (above)
    uint32_t tag0;
    uint32_t shift;
    unsigned long long cuts;
    uint32_t bound[kMaxSlices]; // slice j renders samples bound[j - 1] (0 for j = 0) ... bound[j] - 1; bound[n_slices - 1 ...] = spp
    uint32_t n_slices;          // 1: a lane keeps its pixel for all its samples
    uint32_t wait_cap;          // looks of a wave with nothing but waiting lanes before it gives up (above)
};
constexpr uint32_t kSliceWaitCap = 1u << 22; // x ~3.4 us of sleep + the look itself: > 15 s
// The frame's constants that only a shading round or a refill reads — the camera's twelve floats, the slices — live in DEVICE memory and are
// loaded where they are used (scalar loads, hot in the scalar cache): as kernel arguments hipcc kept them in ~35 scalar registers through the
// traversal loop, in a kernel that has none to spare (with the slices on top: 38 scalar registers spilled into vector lanes, 46 vector
// registers into scratch, 122 -> 145 ms). frame_now() hides the pointer's origin from the compiler, or it would hoist the loads out of the
// wave's loop again.
struct MegaFrame {
    CameraDev cam;
    SliceDev sl;
};

// EXTEND (k_wf_extend)
constexpr uint32_t kExtendChunk = 128;  // queue slots a wave claims per cursor atomic
constexpr uint32_t kExtendBlock = 1024; // 16 independent waves per workgroup
constexpr uint32_t kExtendWaves = 8;    // waves per SIMD (64 VGPRs): EXTEND is occupancy-sensitive (8 -> 6: -19 %)

// the streaming schedule's dynamic queue (k_wf_finish<.., REQ>: rt_kernels.h)
struct FinishRequeue {
    unsigned long long* entries; // P entries, all ones before the launch
    uint32_t* ctl;               // one 64-byte line each, zero before the launch: [0] tail, [32] head
    uint32_t* error;             // set when a wave gave up waiting (kRqIdleCap polls with nothing to trace: seconds) — never observed;
                                 // it turns a logic error into RT_ERR_HIP instead of a hung GPU
    uint32_t capacity;           // >= P (host-sized)
};

constexpr uint32_t kOutBlock = 256u;   // SHOOT: slots of the next queue a workgroup reserves per device atomic

} // namespace rt
