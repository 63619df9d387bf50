// rt_abi.hip — the C entry points of include/rt_mi355x.h for scenes, renderers and frames: argument checks and object life time. What a frame
// launches is rt_frame.hip; the gather rt_comm.hip; the probes rt_probes.hip.
#include "rt_internal.h"

namespace rtlib {
thread_local std::string g_err;
}

extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }
int rt_abi_version(void) { return RT_ABI_VERSION; }

int rt_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(RT_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return n;
}

// Camera::Camera (src/camera.hpp:74-106). Host arithmetic, same operation order as the reference's
// constructor: normalize, two cross products, viewport (aspect, 1), pixel00, per-pixel deltas.
int rt_camera_init(rt_camera* out, int32_t width, int32_t height, const float center[3], const float dir_in[3],
                   float focal_length) {
    if (!out || !center || !dir_in || width <= 0 || height <= 0) return fail(RT_ERR_INVALID, "bad camera arguments");
    auto norm = [](const float v[3], float o[3]) {
        float inv = 1.0f / std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
        o[0] = v[0] * inv, o[1] = v[1] * inv, o[2] = v[2] * inv;
    };
    auto cross = [](const float a[3], const float b[3], float o[3]) {
        o[0] = a[1] * b[2] - a[2] * b[1], o[1] = a[2] * b[0] - a[0] * b[2], o[2] = a[0] * b[1] - a[1] * b[0];
    };
    float dir[3], right[3], up[3], tmp[3];
    const float world_up[3] = {0.0f, 1.0f, 0.0f};
    norm(dir_in, dir);
    cross(dir, world_up, tmp), norm(tmp, right);
    cross(right, dir, tmp), norm(tmp, up);
    const float vp0 = 1.0f * ((float)width / (float)height), vp1 = 1.0f;
    const float du_div = (float)width / (vp0 * 2.0f), dv_div = (float)height / (vp1 * 2.0f);
    for (int a = 0; a < 3; ++a) {
        const float viewport_u = (-right[a]) * vp0, viewport_v = up[a] * vp1;
        out->center[a] = center[a];
        out->pixel00[a] = ((center[a] + viewport_u) + viewport_v) + dir[a] * focal_length;
        out->delta_u[a] = right[a] / du_div;
        out->delta_v[a] = (-up[a]) / dv_div;
    }
    out->width = width, out->height = height;
    return RT_OK;
}

int rt_scene_create(const rt_scene_desc* desc, int device, int bvh_kind, rt_scene** out) {
    if (!out) return fail(RT_ERR_INVALID, "null output pointer");
    *out = nullptr;
    if (bvh_kind != RT_BVH_DEFAULT && bvh_kind != RT_BVH_LBVH && bvh_kind != RT_BVH_SAH && bvh_kind != RT_BVH_LBVH_GPU)
        return fail(RT_ERR_INVALID, "unknown bvh_kind");
    if (bvh_kind == RT_BVH_LBVH_GPU) { // the build itself runs on the device
        if (device < 0) return fail(RT_ERR_NO_DEVICE, "RT_BVH_LBVH_GPU needs a device (device >= 0)");
        int rc0 = device_ok(device);
        if (rc0 != RT_OK) return rc0;
    }
    rt_scene* s = new (std::nothrow) rt_scene();
    if (!s) return fail(RT_ERR_OOM, "host allocation failed");
    std::string err;
    int rc = RT_OK;
    try {
        rc = build_host_scene(desc, bvh_kind, s->hs, err);
    } catch (const std::bad_alloc&) {
        delete s;
        return fail(RT_ERR_OOM, "host allocation failed while building the scene");
    } catch (const std::exception& e) { // nothing may cross the C ABI as an exception
        delete s;
        return fail(RT_ERR_INVALID, std::string("scene build failed: ") + e.what());
    }
    if (rc != RT_OK) {
        delete s;
        return fail(rc, err);
    }
    s->device = device;
    if (device >= 0) {
        rc = device_ok(device);
        if (rc == RT_OK) {
            if (s->hs.nodes.size() > (size_t)(0x7FFFFFFF / 64)) { // (a tree of 33 M nodes: ~130 M triangles)
                rt_scene_destroy(s);
                return fail(RT_ERR_INVALID, "BVH too large for 32-bit node offsets");
            }
            std::vector<BvhNode> dn(s->hs.nodes);
            for (BvhNode& n : dn)
                for (int k = 0; k < 4; ++k)
                    if (n.child[k] >= 0) n.child[k] *= 64;
            rc = upload(dn, &s->dev.nodes, s->device_bytes);
        }
        if (rc == RT_OK) { // the device's triangle records: the host records' ten live dwords at a kTriBytes stride
            std::vector<uint8_t> packed(s->hs.tris.size() * (size_t)kTriBytes + 48, 0); // + 48: the whole-leaf step reads 80 bytes at every leaf, also at a last leaf of one record
            for (size_t i = 0; i < s->hs.tris.size(); ++i) std::memcpy(packed.data() + i * kTriBytes, &s->hs.tris[i], std::min<size_t>(kTriBytes, sizeof(TriRec)));
            rc = upload(packed, &s->dev.tris, s->device_bytes);
        }
        if (rc == RT_OK) rc = upload(s->hs.shade, &s->dev.shade, s->device_bytes);
        if (rc == RT_OK) rc = upload(s->hs.inst, &s->dev.inst, s->device_bytes);
        if (rc == RT_OK) rc = upload(s->hs.mats, &s->dev.mats, s->device_bytes);
        if (rc == RT_OK) rc = upload(s->hs.tex, &s->dev.tex, s->device_bytes);
        if (rc != RT_OK) {
            rt_scene_destroy(s);
            return rc;
        }
        s->dev.n_nodes = (uint32_t)s->hs.nodes.size();
        s->dev.n_tris = (uint32_t)(s->hs.wverts.size() / 9);
        std::memcpy(s->dev.sky, s->hs.sky, 12);
        s->dev.packed_mat = s->hs.packed_mat ? 1u : 0u;
        // what the shading kernels stage in LDS (rt_device.h: ShadeTables): the head of the distinct-matrix table and of the material table
        const bool stage = s->hs.packed_mat && s->hs.n_layers <= 65536u; // (the staged material keeps its layer in 16 bits)
        s->dev.lds_nm = stage ? (uint32_t)std::min<size_t>(s->hs.inst.size(), kLdsNm) : 0u;
        s->dev.lds_mats = stage ? (uint32_t)std::min<size_t>(s->hs.mats.size(), kLdsMats) : 0u;
        for (int a = 0; a < 3; ++a) {
            const float ext = s->hs.bounds_hi[a] - s->hs.bounds_lo[a];
            s->dev.cell_lo[a] = s->hs.bounds_lo[a];
            s->dev.cell_scale[a] = ext > 0.0f && std::isfinite(ext) ? 4.0f / ext : 0.0f;
        }
    }
    *out = s;
    return RT_OK;
}

void rt_scene_destroy(rt_scene* s) {
    if (!s) return;
    if (s->device >= 0 && hipSetDevice(s->device) == hipSuccess) {
        (void)hipFree((void*)s->dev.nodes), (void)hipFree((void*)s->dev.tris), (void)hipFree((void*)s->dev.shade);
        (void)hipFree((void*)s->dev.inst), (void)hipFree((void*)s->dev.mats), (void)hipFree((void*)s->dev.tex);
    }
    delete s;
}

int rt_scene_info(const rt_scene* s, rt_scene_info_t* out) {
    if (!s || !out) return fail(RT_ERR_INVALID, "null argument");
    out->n_triangles = (uint32_t)(s->hs.wverts.size() / 9);
    out->n_nodes = (uint32_t)s->hs.nodes.size();
    out->max_depth = s->hs.max_depth;
    out->max_leaf_tris = s->hs.max_leaf_tris;
    std::memcpy(out->bounds_lo, s->hs.bounds_lo, 12), std::memcpy(out->bounds_hi, s->hs.bounds_hi, 12);
    out->sah_cost = s->hs.sah_cost;
    out->device_bytes = s->device_bytes;
    out->n_leaf_records = (uint32_t)s->hs.tris.size();
    out->n_split_triangles = s->hs.n_split_triangles;
    return RT_OK;
}

int rt_scene_check_bvh(const rt_scene* s) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    std::string err;
    int rc = check_bvh(s->hs, err);
    return rc == RT_OK ? RT_OK : fail(rc, err);
}

int rt_scene_count_visits(const rt_scene* s, uint32_t n, const float* org, const float* dir, int mode, uint64_t* node_visits, uint64_t* tri_tests,
                          float* t, uint32_t* tri) {
    if (!s || (n && (!org || !dir))) return fail(RT_ERR_INVALID, "null argument");
    std::string err;
    const int rc = no_throw([&] { return count_visits(s->hs, n, org, dir, mode, node_visits, tri_tests, t, tri, err); });
    return rc == RT_OK ? RT_OK : fail(rc, err.empty() ? g_err : err);
}

int rt_renderer_create(int kind, rt_scene* scene, int32_t width, int32_t height, uint32_t max_depth,
                       uint32_t sample_count, uint32_t seed_mode, rt_renderer** out) {
    if (!out) return fail(RT_ERR_INVALID, "null output pointer");
    *out = nullptr;
    if (!scene) return fail(RT_ERR_INVALID, "null scene");
    if (kind != RT_RENDERER_MEGAKERNEL && kind != RT_RENDERER_WAVEFRONT) return fail(RT_ERR_INVALID, "unknown renderer kind");
    if (width <= 0 || height <= 0 || (int64_t)width * height > (int64_t)1 << 30) return fail(RT_ERR_INVALID, "bad image size");
    if (sample_count == 0) return fail(RT_ERR_INVALID, "sample_count must be >= 1");
    if (seed_mode > RT_SEED_MEGAKERNEL) return fail(RT_ERR_INVALID, "unknown seed mode");
    if ((uint64_t)sample_count * ((uint64_t)max_depth + 1) > (1ull << 26)) return fail(RT_ERR_INVALID, "sample_count * max_depth too large");
    if (scene->device < 0) return fail(RT_ERR_NO_DEVICE, "scene was built host-only (device < 0)");
    int rc = device_ok(scene->device);
    if (rc != RT_OK) return rc;
    rt_renderer* r = new (std::nothrow) rt_renderer();
    if (!r) return fail(RT_ERR_OOM, "host allocation failed");
    r->kind = kind, r->scene = scene, r->width = width, r->height = height;
    r->max_depth = max_depth, r->spp = sample_count;
    r->seed_mode = seed_mode != RT_SEED_DEFAULT ? seed_mode
                   : (kind == RT_RENDERER_MEGAKERNEL ? RT_SEED_MEGAKERNEL : RT_SEED_WAVEFRONT);
    r->hw_queues = hw_queues_from_env();
    const char* prof = std::getenv("RT_PROFILE_KERNELS");
    r->profile_kernels = prof && prof[0] == '1';
    if (const char* e = dev_knob("RT_MEGA_LDS_PAD")) r->mega_lds_pad = (uint32_t)std::max(0, std::min(100 * 1024, std::atoi(e)));
    if (const char* e = dev_knob("RT_MEGA_OCC")) r->mega_occ = (uint32_t)std::max(1, std::min((int)kMegaWaves, std::atoi(e)));
    auto bail = [&](int code) {
        rt_renderer_destroy(r);
        return code;
    };
    { // the device's lane-stream pool is set up by the FIRST renderer of either kind (see lane_stream_of: early, and in one burst)
        hipStream_t s0 = nullptr;
        bool owned = false;
        if (lane_stream_of(scene->device, 0, &s0, &owned) != hipSuccess) return bail(fail(RT_ERR_HIP, "cannot create the stream-lane pool"));
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, scene->device) == hipSuccess && prop.multiProcessorCount > 0) r->n_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(RT_ERR_HIP, "hipStreamCreate failed"));
    if (hipEventCreate(&r->ev_begin) != hipSuccess || hipEventCreate(&r->ev_end) != hipSuccess ||
        hipEventCreateWithFlags(&r->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&r->ev_tile_read, hipEventDisableTiming) != hipSuccess)
        return bail(fail(RT_ERR_HIP, "hipEventCreate failed"));
    // (the megakernel's camera and slices; the wavefront renderer's one-launch schedule uses the slices part)
    if (hipMalloc((void**)&r->d_frame, sizeof(MegaFrame)) != hipSuccess || hipHostMalloc((void**)&r->h_frame, sizeof(MegaFrame)) != hipSuccess)
        return bail(fail(RT_ERR_OOM, "frame constant buffer allocation failed"));
    if (kind == RT_RENDERER_WAVEFRONT) {
        if (hipMalloc((void**)&r->d_cam, sizeof(CameraDev)) != hipSuccess || hipHostMalloc((void**)&r->h_cam, sizeof(CameraDev)) != hipSuccess)
            return bail(fail(RT_ERR_OOM, "camera buffer allocation failed"));
        // Environment shim for sweep scripts, DEVELOPER builds only (rt_knobs.h; the API is rt_renderer_set_schedule): the variables fill the initial schedule.
        // RT_WF_STREAMS / RT_WF_REQUEUE alone have always meant "not the one-launch schedule": they imply a launch per sample.
        rt_schedule& sc = r->sched;
        if (const char* e = dev_knob("RT_WF_GRAPH")) sc.hip_graph = e[0] == '1';
        if (const char* e = dev_knob("RT_WF_FINISH_DEPTH")) sc.finish_depth = (uint32_t)std::max(0, std::atoi(e));
        if (const char* e = dev_knob("RT_WF_SAMPLES_PER_LAUNCH")) sc.samples_per_launch = (uint32_t)std::max(0, std::atoi(e));
        if (const char* e = dev_knob("RT_WF_REQUEUE")) sc.requeue = e[0] != '0' ? 1 : 0;
        if (const char* e = dev_knob("RT_WF_STREAMS")) sc.stream_lanes = (uint32_t)std::max(1, std::min(8, std::atoi(e)));
        if ((dev_knob("RT_WF_REQUEUE") || dev_knob("RT_WF_STREAMS")) && sc.samples_per_launch == 0) sc.samples_per_launch = 1;
        if (const char* e = dev_knob("RT_WF_LPT")) sc.cost_order = e[0] == '0' ? 0 : (e[0] == '2' ? 1 : -1); // 2: forced
        if (const char* e = dev_knob("RT_WF_EXTEND_OCC")) r->wf_extend_occ = (uint32_t)std::max(1, std::min((int)kExtendWaves, std::atoi(e)));
        if (const char* e = dev_knob("RT_WF_FINISH_OCC")) r->wf_finish_occ = (uint32_t)std::max(1, std::min((int)kMegaWaves, std::atoi(e)));
        if (const char* e = dev_knob("RT_WF_SHOOT_CHUNK")) r->wf_shoot_chunk = (uint32_t)std::max(16, std::min(1024, std::atoi(e)));
        if (const char* e = dev_knob("RT_WF_SHOOT_TAIL")) r->wf_shoot_tail = e[0] == '1';
        if (const char* e = dev_knob("RT_WF_SHOOT_STATIC_PCT")) r->wf_shoot_static_pct = (uint32_t)std::max(0, std::min(100, std::atoi(e)));
        if (const char* e = dev_knob("RT_WF_REORDER")) sc.reorder = e[0] == '1';
        if (const char* e = dev_knob("RT_WF_MATSORT")) sc.matsort = e[0] == '1';
        if (const char* e = dev_knob("RT_WF_FUSED_BOUNCE")) sc.fused_bounce = e[0] == '1';
    }
    rc = no_throw([&] { return alloc_tile_buffers(r); });
    if (rc != RT_OK) return bail(rc);
    *out = r;
    return RT_OK;
}

void rt_renderer_destroy(rt_renderer* r) {
    if (!r) return;
    if (r->scene && hipSetDevice(r->scene->device) == hipSuccess) {
        drain_streams(r, r->pending_stream);
        free_tile_buffers(r);
        if (r->d_cam) (void)hipFree(r->d_cam);
        if (r->h_cam) (void)hipHostFree(r->h_cam);
        if (r->d_frame) (void)hipFree(r->d_frame);
        if (r->h_frame) (void)hipHostFree(r->h_frame);
        for (hipEvent_t e : r->ev_pool) (void)hipEventDestroy(e);
        if (r->ev_begin) (void)hipEventDestroy(r->ev_begin);
        if (r->ev_end) (void)hipEventDestroy(r->ev_end);
        if (r->ev_fork) (void)hipEventDestroy(r->ev_fork);
        if (r->ev_tile_read) (void)hipEventDestroy(r->ev_tile_read);
        if (r->stream) (void)hipStreamDestroy(r->stream);
    }
    delete r;
}

int rt_renderer_set_tile(rt_renderer* r, uint32_t rank, uint32_t world, uint32_t strip_rows) {
    if (!r) return fail(RT_ERR_INVALID, "null renderer");
    if (world == 0 || rank >= world || strip_rows == 0) return fail(RT_ERR_INVALID, "bad tile split");
    if (r->frame_pending) return fail(RT_ERR_INVALID, "a frame is in flight (rt_render_frame_end first)");
    HIPCHK(hipSetDevice(r->scene->device));
    drain_streams(r, r->pending_stream); // the queues are about to be freed: nothing may be running on any of the renderer's streams
    const TileDev old = r->tile;
    r->tile.rank = rank, r->tile.world = world, r->tile.strip_rows = strip_rows;
    const int rc = no_throw([&] { return alloc_tile_buffers(r); });
    if (rc != RT_OK) r->tile.rank = old.rank, r->tile.world = old.world, r->tile.strip_rows = old.strip_rows;
    return rc;
}

int32_t rt_renderer_local_rows(const rt_renderer* r) { return r ? r->tile.local_rows : 0; }

int32_t rt_renderer_global_row(const rt_renderer* r, int32_t local_row) {
    if (!r || local_row < 0 || local_row >= r->tile.local_rows) return -1;
    const uint32_t strip = (uint32_t)local_row / r->tile.strip_rows, within = (uint32_t)local_row % r->tile.strip_rows;
    return (int32_t)((strip * r->tile.world + r->tile.rank) * r->tile.strip_rows + within);
}

int rt_renderer_set_profiling(rt_renderer* r, int enable) {
    if (!r) return fail(RT_ERR_INVALID, "null renderer");
    r->profile_kernels = enable != 0;
    return RT_OK;
}

int rt_renderer_set_russian_roulette(rt_renderer* r, uint32_t start_bounce) {
    if (!r) return fail(RT_ERR_INVALID, "null renderer");
    if (r->rr_start != start_bounce) drop_graph(r); // the bounce flags are baked into the captured launches
    r->rr_start = start_bounce;
    return RT_OK;
}

int rt_renderer_get_schedule(const rt_renderer* r, rt_schedule* out) {
    if (!r || !out) return fail(RT_ERR_INVALID, "null argument");
    *out = r->sched;
    return RT_OK;
}

int rt_renderer_set_schedule(rt_renderer* r, const rt_schedule* s) {
    if (!r || !s) return fail(RT_ERR_INVALID, "null argument");
    if (s->stream_lanes > 8) return fail(RT_ERR_INVALID, "at most 8 stream lanes");
    if (s->requeue < -1 || s->requeue > 1 || s->cost_order < -1 || s->cost_order > 1 || s->reorder > 1 || s->matsort > 1 || s->hip_graph > 1 || s->fused_bounce > 1)
        return fail(RT_ERR_INVALID, "schedule field out of range");
    if (r->frame_pending) return fail(RT_ERR_INVALID, "a frame is in flight (rt_render_frame_end first)");
    if (s->pixel_slices < -1 || s->pixel_slices > (int32_t)kMaxSlices) return fail(RT_ERR_INVALID, "pixel_slices: -1 (automatic), 0 or 1 (off), 2 .. 8");
    if (r->kind != RT_RENDERER_WAVEFRONT && (s->pixel_slices > 1 || s->pixel_slices < 0) == (r->sched.pixel_slices > 1 || r->sched.pixel_slices < 0)) { // the megakernel is one launch: only its pixel slices are a choice (their state buffer exists or not)
        r->sched = *s;
        return RT_OK;
    }
    HIPCHK(hipSetDevice(r->scene->device));
    drain_streams(r, r->pending_stream); // the queues are re-allocated (stream lanes, second queue, hit records, dynamic queue)
    const rt_schedule old = r->sched;
    r->sched = *s;
    const int rc = no_throw([&] { return alloc_tile_buffers(r); });
    if (rc != RT_OK) r->sched = old; // the buffers are gone (frames are refused), the schedule on record is the last one that worked
    return rc;
}

int rt_render_frame(rt_renderer* r, const rt_camera* cam, float* rgba_f32, uint8_t* rgba_u8, rt_stats* stats) {
    if (!r) return fail(RT_ERR_INVALID, "null renderer");
    int rc = no_throw([&] { return render_impl(r, cam, rgba_f32 ? r->d_f32 : nullptr, rgba_u8 ? r->d_u8 : nullptr, r->stream, stats); });
    if (rc != RT_OK) return rc;
    if (rgba_f32 && r->n_local) HIPCHK(hipMemcpy(rgba_f32, r->d_f32, (size_t)r->n_local * 16, hipMemcpyDeviceToHost));
    if (rgba_u8 && r->n_local) HIPCHK(hipMemcpy(rgba_u8, r->d_u8, (size_t)r->n_local * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_render_frame_device(rt_renderer* r, const rt_camera* cam, void* d_rgba_f32, void* d_rgba_u8, void* stream,
                           rt_stats* stats) {
    if (!r) return fail(RT_ERR_INVALID, "null renderer");
    return no_throw([&] { return render_impl(r, cam, (float*)d_rgba_f32, (uint8_t*)d_rgba_u8, stream ? (hipStream_t)stream : r->stream, stats); });
}

int rt_render_frame_begin(rt_renderer* r, const rt_camera* cam, void* d_rgba_f32, void* d_rgba_u8, void* stream) {
    if (!r) return fail(RT_ERR_INVALID, "null renderer");
    return no_throw([&] { return frame_begin(r, cam, (float*)d_rgba_f32, (uint8_t*)d_rgba_u8, stream ? (hipStream_t)stream : r->stream); });
}

int rt_render_frame_end(rt_renderer* r, rt_stats* stats) {
    if (!r) return fail(RT_ERR_INVALID, "null renderer");
    return no_throw([&] { return frame_end(r, stats); });
}

} // extern "C"
