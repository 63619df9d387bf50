/*
 * rt_mi355x.h — C ABI of librt_mi355x.so, the MI355X (gfx950) path-tracing hot path.
 *
 * This is the drop-in boundary for the per-pixel ray-trace loop of felipeagc/sycl-ray-tracer.
 * Every entry point names the reference interface it replaces (file:line relative to the
 * reference tree). No C++ types, no torch types, no exceptions cross this boundary: plain
 * pointers and sizes only. Every function returns RT_OK (0) or a negative rt_status;
 * rt_last_error() returns a thread-local message for the last failure.
 *
 * Threading: a renderer is used from one host thread at a time (the reference is
 * single-threaded: src/main.cpp:57-70). Different renderers may be used from different threads.
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 8

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,     /* bad argument (null pointer, out-of-range index, zero size) */
    RT_ERR_NO_DEVICE = -2,   /* no usable gfx950 device / HIP runtime not functional      */
    RT_ERR_HIP = -3,         /* a HIP call failed; message carries hipGetErrorString      */
    RT_ERR_OOM = -4,         /* host or device allocation failed                          */
    RT_ERR_UNSUPPORTED = -5  /* valid request this build does not implement               */
} rt_status;

/* ---- texture array constants: src/image_manager.hpp:12-14 --------------------------------- */
#define RT_TEX_SIZE 512u      /* IMAGE_SIZE = {512,512}            */
#define RT_TEX_CHANNELS 4u    /* IMAGE_CHANNELS                     */
#define RT_TEX_MAX_LAYERS 128u /* MAX_IMAGES                        */

/* ---- Camera: the by-value POD the kernels consume, == raytracer::Camera (src/camera.hpp:65-72)
 * center / pixel00_loc / pixel_delta_u / pixel_delta_v / img_size. */
typedef struct rt_camera {
    float center[3];
    float pixel00[3];
    float delta_u[3];
    float delta_v[3];
    int32_t width;
    int32_t height;
} rt_camera;

/* Host-side camera constructor == Camera::Camera(img_size, cam_center, cam_dir, focal_length)
 * (src/camera.hpp:74-106). Pure host arithmetic; needs no GPU. */
int rt_camera_init(rt_camera* out, int32_t width, int32_t height, const float center[3],
                   const float dir[3], float focal_length);

/* ---- Materials: == raytracer::Material tagged union (src/material.hpp:56-61,163-238) ------- */
enum { RT_MAT_NONE = 0, RT_MAT_DIFFUSE = 1, RT_MAT_METALLIC = 2, RT_MAT_DIELECTRIC = 3 };
/* == TextureType (src/material.hpp:13-16) */
enum { RT_TEX_COLOR = 0, RT_TEX_IMAGE = 1 };

typedef struct rt_material {
    uint32_t type;      /* RT_MAT_*                                                        */
    uint32_t tex_kind;  /* RT_TEX_*: albedo is `color` or layer `tex_layer` of the array   */
    float color[3];     /* Texture::color  (baseColorFactor.rgb)                           */
    uint32_t tex_layer; /* ImageRef::index (src/image_manager.hpp:26-28)                   */
    float emissive[3];  /* MaterialDiffuse/Metallic::emissive; ignored for dielectric      */
    float roughness;    /* MaterialMetallic::roughness                                     */
    float ior;          /* MaterialDielectric::ior                                         */
} rt_material;          /* 44 bytes */

/* ---- Instance: one Embree instance geometry + its GeometryData (src/scene.hpp:17-24,
 * filled at src/scene.cpp:487-505). `transform` is the 4x4 column-major global matrix handed
 * to rtcSetGeometryTransform; `normal_mat` is GeometryData::obj_to_world =
 * transpose(inverse(mat3(global))) as a column-major 3x3. The index of an instance in the
 * array is Embree's instID[0] (attach order, src/scene.cpp:101-106). */
typedef struct rt_instance {
    float transform[16];
    float normal_mat[9];
    uint32_t material; /* index into rt_scene_desc::materials */
} rt_instance;         /* 104 bytes */

/* ---- Scene description: flat, caller-owned host arrays, copied during rt_scene_create.
 * Replaces what trace_ray reaches through RTCScene + GeometryData (src/trace_ray.hpp:18-45):
 *   positions/normals/uvs : object-space vertex attributes (GeometryData::vertex/normal/uv_buffer)
 *   indices               : 3 vertex indices per triangle (GeometryData::index_buffer), already
 *                           offset into the shared vertex arrays
 *   tri_instance          : instance (instID) of each triangle; triangles of one instance are
 *                           contiguous and in primID order
 *   textures              : n_layers x 512 x 512 x RGBA8, row 0 first (image_manager.hpp:76-100)
 *   sky                   : Scene::sky_color (src/scene.hpp:76) */
typedef struct rt_scene_desc {
    uint32_t n_vertices;
    const float* positions;       /* 3 * n_vertices */
    const float* normals;         /* 3 * n_vertices */
    const float* uvs;             /* 2 * n_vertices */
    uint32_t n_triangles;
    const uint32_t* indices;      /* 3 * n_triangles */
    const uint32_t* tri_instance; /* n_triangles */
    uint32_t n_instances;
    const rt_instance* instances;
    uint32_t n_materials;
    const rt_material* materials;
    uint32_t n_layers;
    const uint8_t* textures;      /* n_layers * 512*512*4, may be NULL iff n_layers == 0 */
    float sky[3];
} rt_scene_desc;

typedef struct rt_scene rt_scene;       /* opaque: device-resident flattened triangles + BVH */
typedef struct rt_renderer rt_renderer; /* opaque: ray queues, RNG states, accumulators      */

/* BVH builder selection for rt_scene_create (the image is independent of the choice). */
/*   RT_BVH_SAH      binned surface-area heuristic, built on the host (default: fastest traversal)
 *   RT_BVH_LBVH     Morton-order LBVH, built on the host
 *   RT_BVH_LBVH_GPU the same LBVH family built entirely on the device (Morton codes, radix sort, Karras tree,
 *                   refit, BVH4 collapse + quantisation): fastest build, needs device >= 0 */
enum { RT_BVH_DEFAULT = 0, RT_BVH_LBVH = 1, RT_BVH_SAH = 2, RT_BVH_LBVH_GPU = 3 };

/* Builds the world-space triangle set and the BVH, uploads everything to HIP device `device`.
 * Replaces Scene's Embree side: rtcNewScene/rtcAttachGeometry/rtcCommitScene
 * (src/scene.cpp:101-107,406-439,487-507). device < 0 builds a host-only scene (no HIP call):
 * usable with rt_scene_info / rt_scene_check_bvh only. */
int rt_scene_create(const rt_scene_desc* desc, int device, int bvh_kind, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);

typedef struct rt_scene_info_t {
    uint32_t n_triangles;
    uint32_t n_nodes;
    uint32_t max_depth;     /* deepest leaf */
    uint32_t max_leaf_tris;
    float bounds_lo[3];
    float bounds_hi[3];
    double sah_cost;        /* surface-area-heuristic cost of the tree (diagnostic) */
    uint64_t device_bytes;  /* bytes resident in HBM for this scene */
    uint32_t n_leaf_records;     /* triangle records in the leaves: n_triangles, plus one per extra leaf a pre-split triangle sits in */
    uint32_t n_split_triangles;  /* triangles the SAH builder's pre-splitting pass cut into several references (large triangles whose
                                    boxes enclose much empty space; a result cannot change: closest t, ties to the lowest index) */
} rt_scene_info_t;
int rt_scene_info(const rt_scene* scene, rt_scene_info_t* out);

/* Host-side structural check of the built BVH: every triangle in exactly one leaf, every child
 * box inside its parent's, every triangle inside its leaf box. Returns RT_OK or RT_ERR_INVALID
 * (message names the first violation). Needs no GPU. */
int rt_scene_check_bvh(const rt_scene* scene);
/* Diagnostic, host only (works on a scene built with device < 0): closest-hit walks of the tree for n rays as the traversal kernels make them
 * (children nearest first, culled by the closest hit so far), counting node visits and triangle tests, with a choice of the child boxes
 * tested: mode 0 the decoded 8-bit quantised boxes (what the kernels test), 1 the exact padded bounds of each child's subtree (what fp32
 * boxes would hold), 2 the exact bounds re-quantised with two more bits per plane. t / tri (may be NULL) receive the closest hits. The
 * difference between the modes is what the quantisation costs in visits on a given scene and ray set (profiles/r05_quantisation.txt). */
int rt_scene_count_visits(const rt_scene* scene, uint32_t n, const float* org, const float* dir, int mode,
                          uint64_t* node_visits, uint64_t* tri_tests, float* t, uint32_t* tri);

/* Closest-hit query for a batch of rays: the replacement for rtcIntersect1 at
 * src/trace_ray.hpp:18-27 (tnear = 1e-4, tfar = +inf, no culling, no masks).
 * org/dir: 3*n floats (host). Outputs (host, n each): t (+inf on miss), u, v and
 * tri = global triangle index in rt_scene_desc order (0xFFFFFFFF on miss).
 * Range of the contract: the result equals the brute-force closest hit over all triangles (same fp32 Moller-Trumbore) for ray
 * origins at most 100 scene scales outside the scene's bounds on any axis (scene scale = max(largest extent of the bounds, largest
 * |coordinate| of the bounds): what the padding of the BVH boxes, 2e-5 x scale, is derived from) — every ray the renderers
 * generate from a camera inside that range. Farther out the fp32 error of the ray itself (~1e-7 x |origin|) exceeds the padding and
 * a hit that only exists by that error could be culled, so the range is ENFORCED: rt_intersect_batch returns RT_ERR_INVALID (naming
 * the first offending ray) if any origin lies outside it or is not finite, and rt_render_frame* refuse such a camera centre.
 * The reference's rtcIntersect1 (src/trace_ray.hpp:18-27) states no range; this is a documented narrowing, not a silent one. */
int rt_intersect_batch(rt_scene* scene, uint32_t n, const float* org, const float* dir, float* t,
                       float* u, float* v, uint32_t* tri);

/* ---- Renderers: == IRenderer implementations (src/render.hpp:11-18) ------------------------ */
enum {
    RT_RENDERER_MEGAKERNEL = 0, /* MegakernelRenderer (src/render_megakernel.hpp:13-19) */
    RT_RENDERER_WAVEFRONT = 1   /* WavefrontRenderer  (src/render_wavefront.hpp:55-61)  */
};
/* Per-pixel xorshift seed:
 *   RT_SEED_WAVEFRONT : x + y*W                        (src/render_wavefront.cpp:69-73)
 *   RT_SEED_MEGAKERNEL: x*H8 + y, H8 = 8*ceil(H/8)     (src/render_megakernel.cpp:90-93,144-146)
 *   RT_SEED_DEFAULT   : the renderer kind's own rule */
enum { RT_SEED_DEFAULT = 0, RT_SEED_WAVEFRONT = 1, RT_SEED_MEGAKERNEL = 2 };

/* == the renderer constructors (App&, img_size, image&, max_depth, sample_count). The output
 * image is passed to rt_render_frame instead of being captured by reference. */
int rt_renderer_create(int kind, rt_scene* scene, int32_t width, int32_t height,
                       uint32_t max_depth, uint32_t sample_count, uint32_t seed_mode,
                       rt_renderer** out);
void rt_renderer_destroy(rt_renderer* r);

/* Multi-GPU tile split (no reference counterpart: the reference is single-device). The frame is
 * cut into horizontal strips of `strip_rows` rows; strip k belongs to rank k % world. A renderer
 * renders only its own strips, packed top to bottom into a compact buffer of
 * rt_renderer_local_rows() rows. RNG seeds use GLOBAL pixel coordinates, so the union of the
 * tiles is bit-identical to a single-GPU frame. Default: rank 0 of world 1. */
int rt_renderer_set_tile(rt_renderer* r, uint32_t rank, uint32_t world, uint32_t strip_rows);
int32_t rt_renderer_local_rows(const rt_renderer* r);
/* Global row index of local row `local_row` (for de-interleaving a gathered frame). */
int32_t rt_renderer_global_row(const rt_renderer* r, int32_t local_row);

/* Per-launch hipEvent timing of the traversal / shading kernels, each on the stream it is launched on: rt_stats.hot_kernel_ms (the
 * dominant kernel of the schedule that ran: k_megakernel, k_wf_finish, k_wf_extend or the fused per-bounce kernel) and
 * rt_stats.kernel_ms[RT_K_*] per family. The megakernel is always timed; off by default for the wavefront renderer (two event records per
 * launch — three per EXTEND + SHADE pair —, which the per-bounce schedules with their thousands of launches per frame feel); also enabled
 * by the environment variable RT_PROFILE_KERNELS=1 at renderer creation. No reference counterpart (the reference's print_elapsed helper is
 * commented out: src/render_wavefront.cpp:129-137). */
int rt_renderer_set_profiling(rt_renderer* r, int enable);

/* Russian roulette, an EXTENSION: the reference only lists it as a to-do (PLAN.md:23-27) and never implements
 * it, so it is off by default (start_bounce = 0) and changes the image and the ray count when turned on.
 * A path that continues after its bounce b, start_bounce <= b < max_depth, survives with probability
 * p = clamp(max component of its stored attenuation, 0.05, 1) (one extra RNG draw) and carries on with
 * attenuation / p; otherwise it ends with no contribution. Same rule in both renderers and in the oracle. */
int rt_renderer_set_russian_roulette(rt_renderer* r, uint32_t start_bounce);

/* ---- Schedule of the wavefront renderer --------------------------------------------------------------------------------
 * The reference's WavefrontRenderer::render_frame (src/render_wavefront.cpp:396-431) has ONE schedule: per sample, one
 * generate_camera_rays launch and one shoot_rays launch per bounce, survivors compacted between bounces (:282-311). This
 * library renders the same frame (bit for bit) under several schedules; which one runs is chosen here — never silently — and
 * reported back in rt_stats, per kernel family, so that a test can assert that the kernels it means to test were launched.
 *   finish_depth        bounces rendered launch by launch: k_wf_extend + k_wf_shade with __ballot / mbcnt compaction of the
 *                       survivors into the next queue, the reference's shape. The rest of every path is followed by
 *                       k_wf_finish in one launch. 0 = everything in k_wf_finish (default); RT_SCHED_ALL_BOUNCES (or any
 *                       value >= max_depth) = the reference's per-bounce schedule.
 *   samples_per_launch  samples of a pixel one k_wf_finish launch renders. 0 = all of them: ONE launch per frame (default).
 *   stream_lanes        interleaved sub-tiles rendered on HIP streams of their own. 0 = automatic (1 for the one-launch
 *                       schedule, 3 for a launch or launch pair per bounce all the way down, 2 otherwise). HIP serves streams from
 *                       GPU_MAX_HW_QUEUES hardware queues (default 4). The library never changes the environment: the HOST sets the
 *                       variable (8) before its first HIP call; the library reads it once, when the process's first renderer is
 *                       created, and resolves an automatic lane count down to lanes + 2 <= queues (rt_stats.hw_queues / .stream_lanes).
 *                       Renderers of one device share their lane streams: drive them from one host thread.
 *   requeue             with samples_per_launch > 1: 1 = a pixel between two samples goes through a device-wide dynamic queue
 *                       (breadth first), 0 = it stays in its lane (depth first), -1 = automatic.
 *   reorder, matsort    SURVEY 8(f) row f-3, per-bounce schedule only: k_wf_shade bins a block's survivors by direction octant
 *                       and 4x4x4 origin cell / partitions a block's rays by material kind before shading.
 *   cost_order          one-launch schedule: sample 0 in a launch of its own, the other samples with the most expensive 8x8
 *                       blocks first. -1 = automatic (>= 32 spp, 1..4 pixel generations), 0 = off, 1 = whenever the tile
 *                       consists of whole 8x8 blocks (a tile that does not falls back to queue order).
 *   hip_graph           1 = the frame's launches are captured once and replayed as a hipGraph.
 *   fused_bounce        1 = the bounces rendered launch by launch (finish_depth) use ONE kernel per bounce that intersects, shades and
 *                       compacts the survivors into the next queue — shoot_rays as the reference has it (src/render_wavefront.cpp:222-312)
 *                       — instead of the EXTEND + SHADE pair with its hit-record round trip. reorder / matsort act in k_wf_shade and
 *                       are ignored then.
 *   pixel_slices        MEGAKERNEL renderers (the only field they use). A pixel's samples are a sequential chain (one RNG word), but not
 *                       bound to one lane: the chain is cut into slices of decreasing length, all first slices are rendered, then all
 *                       second ones ..., the pixel's colour sum and RNG word travelling through memory in between, so that the frame
 *                       drains over its last, short slices instead of over whole pixels (bit-identical frame). -1 = automatic (as many
 *                       as the tile's size calls for; one for a tile of at most ~1.25 pixels per resident lane), 0 or 1 = off,
 *                       2 .. 8 = that many. rt_stats.pixel_slices reports what ran.
 * Environment: the library reads GPU_MAX_HW_QUEUES (above), RT_PROFILE_KERNELS=1 (rt_renderer_set_profiling at creation) and
 * RT_KERNEL_STATS=1 (the instrumented kernel instantiations; their report goes to stderr) — and nothing else. Sweep knobs and test
 * hooks (RT_WF_*, RT_MEGA_*, RT_BVH_*, RT_INJECT_ALLOC_FAILURE) exist in the developer build only: `make -C csrc dev` ->
 * librt_mi355x_dev.so (csrc/rt_knobs.h). */
#define RT_SCHED_ALL_BOUNCES 0xFFFFFFFFu
typedef struct rt_schedule {
    uint32_t finish_depth;
    uint32_t samples_per_launch;
    uint32_t stream_lanes;
    int32_t requeue;
    uint32_t reorder;
    uint32_t matsort;
    int32_t cost_order;
    uint32_t hip_graph;
    uint32_t fused_bounce;
    int32_t pixel_slices; /* (ABI 8) megakernel */
} rt_schedule;
/* Megakernel renderers accept the call and use pixel_slices only. No frame may be in flight; the tile's queues are re-allocated. */
int rt_renderer_set_schedule(rt_renderer* r, const rt_schedule* s);
int rt_renderer_get_schedule(const rt_renderer* r, rt_schedule* out);

/* Kernel families, for rt_stats::launches_by_kernel. A k_wf_shade launch counts in RT_K_WF_SHADE and, when the flags are on,
 * also in RT_K_WF_SHADE_REORDER / RT_K_WF_SHADE_MATSORT; a k_wf_finish launch with the dynamic queue counts in RT_K_WF_FINISH
 * and RT_K_WF_FINISH_REQUEUE. */
enum {
    RT_K_MEGAKERNEL = 0,
    RT_K_WF_INIT = 1,
    RT_K_WF_GENERATE = 2,
    RT_K_WF_EXTEND = 3,
    RT_K_WF_SHADE = 4,
    RT_K_WF_SHADE_REORDER = 5,
    RT_K_WF_SHADE_MATSORT = 6,
    RT_K_WF_FINISH = 7,
    RT_K_WF_FINISH_REQUEUE = 8,
    RT_K_WF_TILE_ORDER = 9, /* k_wf_tile_cost + k_wf_order_tiles (cost ordering) */
    RT_K_WF_RESOLVE = 10,
    RT_K_FILL_BLACK = 11,
    RT_K_WF_SHOOT = 12, /* k_wf_finish limited to one bounce: intersect + shade + compact, one launch per bounce (fused_bounce) */
    RT_K_COUNT = 16
};

typedef struct rt_stats {
    uint64_t rays;       /* trace_ray calls (src/render_megakernel.cpp:32, render_wavefront.cpp:407) */
    double seconds;      /* host wall clock, first launch -> last kernel complete                  */
    double device_ms;    /* same region measured with hipEvents on the render stream               */
    double hot_kernel_ms;/* summed duration of the dominant kernel's launches (hipEvents), or 0    */
    uint32_t hot_kernel_launches;
    uint32_t launches;   /* kernel launches issued for the frame                                   */
    /* what actually ran (ABI 5): launches of the frame per kernel family, and the schedule as it was resolved for this tile */
    uint32_t launches_by_kernel[RT_K_COUNT];
    uint32_t stream_lanes;       /* HIP streams the frame's launch chains ran on (1 for the megakernel)            */
    uint32_t samples_per_launch; /* samples of a pixel per k_wf_finish launch as resolved (spp = one launch per frame) */
    uint32_t finish_depth;       /* bounces rendered as EXTEND + SHADE launch pairs, min(schedule, max_depth)      */
    uint32_t cost_ordered;       /* 1 = the cost-ordered second launch ran                                          */
    /* with profiling on (rt_renderer_set_profiling): summed hipEvent duration of the launches of the traversal / shading kernel
     * families (RT_K_MEGAKERNEL, RT_K_WF_EXTEND, RT_K_WF_SHADE, RT_K_WF_SHOOT, RT_K_WF_FINISH), each on the stream it was launched on; else 0 */
    double kernel_ms[RT_K_COUNT];
    /* (ABI 7) hardware queues the library assumed HIP serves the process's streams from: GPU_MAX_HW_QUEUES as the HOST had set it when the
     * renderer was created, 4 (HIP's default) when unset. The library never changes the environment; an automatic stream-lane count is resolved
     * down to lanes + 2 <= hw_queues (the frame's stream and one stream of the host framework beside the lanes'), see stream_lanes above. */
    uint32_t hw_queues;
    uint32_t pixel_slices; /* (ABI 8) megakernel: slices a pixel's samples were rendered in (1 = every pixel on one lane) */
} rt_stats;

/* == IRenderer::render_frame(camera, scene) (src/render_megakernel.cpp:75-187,
 * src/render_wavefront.cpp:396-431). Renders this renderer's tile and copies it to host:
 *   rgba_f32: local_rows*W*4 floats, the pre-quantisation framebuffer (sqrt(mean rgb), alpha 1)
 *   rgba_u8 : local_rows*W*4 bytes, what the reference's RGBA-unorm8 image holds
 * Either may be NULL. Unlike the reference it prints nothing and writes no file: the caller
 * (host adapter / CLI) prints the three stat lines and writes out.png. */
int rt_render_frame(rt_renderer* r, const rt_camera* cam, float* rgba_f32, uint8_t* rgba_u8,
                    rt_stats* stats);

/* Same, but the outputs are DEVICE pointers on the renderer's device (e.g. a tensor's data_ptr)
 * and `stream` is a hipStream_t (NULL = the renderer's own stream). Returns after the frame has
 * completed on the device (stats need the ray counter). */
int rt_render_frame_device(rt_renderer* r, const rt_camera* cam, void* d_rgba_f32,
                           void* d_rgba_u8, void* stream, rt_stats* stats);

/* The same frame in two calls: _begin enqueues it on `stream` (NULL = the renderer's own) and returns at once, _end waits
 * for it and fills `stats`. One frame per renderer may be in flight; frames of DIFFERENT renderers overlap on the device,
 * which is how a caller hides the end-of-frame drain (the last pixels' sequential samples) behind the next frame.
 * No reference counterpart (the reference blocks after every kernel: src/render_wavefront.cpp:396-431). */
int rt_render_frame_begin(rt_renderer* r, const rt_camera* cam, void* d_rgba_f32, void* d_rgba_u8, void* stream);
int rt_render_frame_end(rt_renderer* r, rt_stats* stats);

/* ---- Multi-GPU frame gather over xGMI (no reference counterpart: the reference renders on ONE device and hands its
 * single image to stbi_write_png, src/main.cpp:57-70, src/util.hpp:8-33). SURVEY §8(e): the frame is split into interleaved
 * strips (rt_renderer_set_tile), every GPU renders its strips into its own compact device buffer, and ONE collective brings
 * them to the root GPU: ncclGather of the per-GPU strip buffers (rccl.h: ncclCommInitAll / ncclGroupStart / ncclGather /
 * ncclGroupEnd) followed by a de-interleave kernel on the root. The frame touches host memory only when the caller asks
 * for it (the PNG). librccl.so is loaded on first use (dlopen): single-GPU callers never load it.
 *
 * One process drives all devices (the C++ adapter's `--devices A,B,..`); the one-process-per-GPU form of the same gather is
 * torch.distributed (backend nccl = RCCL) in rtamd/dist.py. */
typedef struct rt_comm rt_comm;
/* Communicator over `n` devices, rank i on devices[i]; rank 0 is the root. If the same device appears more than once
 * (rehearsal of an n-GPU split on fewer GPUs) RCCL cannot be used (one rank per device): the gather then moves the strips
 * with device-to-device copies and runs the same de-interleave kernel; rt_comm_uses_rccl() tells which. */
int rt_comm_create(int n, const int* devices, rt_comm** out);
void rt_comm_destroy(rt_comm* c);
int rt_comm_uses_rccl(const rt_comm* c);
/* The renderer's own device buffers of its tile (local_rows x W x 4 floats / bytes): render into them without any host copy
 * with rt_render_frame_begin(r, cam, rt_renderer_tile_f32(r), rt_renderer_tile_u8(r), NULL) + rt_render_frame_end. */
void* rt_renderer_tile_f32(rt_renderer* r);
void* rt_renderer_tile_u8(rt_renderer* r);
/* Gathers the last frame of renderers[0..n-1] — renderer i must be tile (i, n, strip_rows) of one W x H frame on devices[i],
 * rendered into its own tile buffers — into the full frame on the root device and, if the pointers are not NULL, copies it
 * to the host (rgba_f32: H*W*4 floats, rgba_u8: H*W*4 bytes; the f32 / u8 planes are only gathered when asked for here or
 * through want_device_*). The device-resident frame stays valid until the next gather: rt_comm_frame_f32 / _u8. */
int rt_frame_gather(rt_comm* c, rt_renderer* const* renderers, float* rgba_f32, uint8_t* rgba_u8, int want_device_f32,
                    int want_device_u8);
/* The same in two calls. _begin only ENQUEUES the collective and the de-interleave (event waits between the renderers' streams
 * and the root's; no host wait): a renderer's next frame may be begun right after it and overlaps with the gather. rt_comm_wait
 * blocks until the gathered frame is complete on the root device and copies it to the host where a pointer is given.
 * If a collective fails half-way the communicator is marked unusable (every later call returns RT_ERR_HIP): destroy it and
 * create a new one. */
int rt_frame_gather_begin(rt_comm* c, rt_renderer* const* renderers, int want_f32, int want_u8);
int rt_comm_wait(rt_comm* c, float* rgba_f32, uint8_t* rgba_u8);
int rt_comm_size(const rt_comm* c); /* ranks of the communicator (with RCCL: what ncclCommInitAll was given) */
const void* rt_comm_frame_f32(const rt_comm* c);
const void* rt_comm_frame_u8(const rt_comm* c);

/* ---- Device unit probes (parity tests of the building blocks; tiny launches) --------------- */
/* XorShift32State::operator() (src/xorshift.hpp:11-20) run on the device: n draws from `seed`. */
int rt_probe_xorshift(int device, uint32_t seed, uint32_t n, float* out, uint32_t* state_out);
/* float -> half -> float round trip as RayData stores dir/att/rad (src/camera.hpp:18-43). */
int rt_probe_half_roundtrip(int device, uint32_t n, const float* in, float* out,
                            uint16_t* bits_out);
/* Material::scatter (src/material.hpp:211-224) on the device for n independent inputs:
 * dir/normal: 3n, uv: 2n, seed: n. Outputs ok (n, 0/1), out_dir 3n, out_att 3n, seed_out n. */
int rt_probe_scatter(rt_scene* scene, uint32_t material, uint32_t n, const float* dir,
                     const float* normal, const float* uv, const uint32_t* seed, uint8_t* ok,
                     float* out_dir, float* out_att, uint32_t* seed_out);

/* The kernels' short forms of RN(1 / x) and RN(1 / RN(sqrt(x))) (rt_device.h: rcp_rn, inv_sqrt2; glm::normalize as src/trace_ray.hpp and
 * src/material.hpp use it, the 1 / det of the triangle test) against the IEEE expressions, on ALL 2^32 float bit patterns:
 * mismatches[0], mismatches[1] = inputs on which they differ. The arithmetic contract requires both to be 0. ~1 s. */
int rt_probe_rounding(int device, uint64_t* mismatches);

const char* rt_last_error(void);
int rt_abi_version(void);
/* Number of HIP devices visible, or a negative rt_status. */
int rt_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355X_H */
