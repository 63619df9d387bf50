// rt_comm.hip — the multi-GPU frame gather of include/rt_mi355x.h: ncclGather of the renderers' strip buffers to the root device and
// the de-interleave kernel there; librccl.so is loaded on first use.
#include <rccl/rccl.h> // types and prototypes only: librccl.so is dlopen()ed by rt_comm_create
#include <dlfcn.h>

#include "rt_internal.h"

// ---- multi-GPU frame gather: ncclGather of the strip buffers + de-interleave on the root -------------------
namespace {

struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err; // why the library is unusable: dlerror() read ONCE, right after the failing call (a second dlerror() returns NULL)
    bool ok() const { return lib && CommInitAll && CommDestroy && GroupStart && GroupEnd && Gather && GetErrorString; }
};

RcclApi& rccl() { // loaded once, on the first rt_comm_create that needs it
    static RcclApi api = [] {
        RcclApi a;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (a.lib) break;
            const char* e = dlerror();
            a.err = e ? e : "dlopen failed";
        }
        if (a.lib) {
            a.err.clear();
            auto sym = [&a](const char* name) {
                void* f = dlsym(a.lib, name);
                if (!f && a.err.empty()) {
                    const char* e = dlerror();
                    a.err = e ? e : (std::string("symbol missing: ") + name);
                }
                return f;
            };
            a.CommInitAll = (decltype(a.CommInitAll))sym("ncclCommInitAll");
            a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
            a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
            a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
            a.Gather = (decltype(a.Gather))sym("ncclGather");
            a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
        }
        return a;
    }();
    return api;
}

#define NCCLCHK(expr)                                                                                     \
    do {                                                                                                  \
        ncclResult_t r_ = (expr);                                                                         \
        if (r_ != ncclSuccess) return fail(RT_ERR_HIP, std::string(#expr) + ": " + rccl().GetErrorString(r_)); \
    } while (0)

// gathered[rank][row][x] (every rank padded to rows_max rows) -> full[global row][x]; one thread per pixel of the frame
__global__ void __launch_bounds__(256) k_deinterleave(uint32_t width, uint32_t height, uint32_t world, uint32_t strip_rows, uint32_t rows_max,
                                                       const float4* __restrict__ g_f32, const uint32_t* __restrict__ g_u8,
                                                       float4* __restrict__ full_f32, uint32_t* __restrict__ full_u8) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= width * height) return;
    const uint32_t x = i % width, y = i / width;
    const uint32_t strip = y / strip_rows, rank = strip % world;
    const uint32_t local_row = (strip / world) * strip_rows + y % strip_rows; // the inverse of tile_global_row
    const size_t src = ((size_t)rank * rows_max + local_row) * width + x;
    if (g_f32) full_f32[i] = g_f32[src];
    if (g_u8) full_u8[i] = g_u8[src];
}

} // namespace

struct rt_comm {
    std::vector<int> devices;
    std::vector<ncclComm_t> comms; // empty: local mode (a device appears more than once)
    bool use_rccl = false;
    // root-device buffers, (re)allocated when the frame geometry changes
    int32_t width = 0, height = 0;
    uint32_t rows_max = 0;
    float4* d_gather_f32 = nullptr;
    uint32_t* d_gather_u8 = nullptr;
    float4* d_full_f32 = nullptr;
    uint32_t* d_full_u8 = nullptr;
    hipStream_t root_stream = nullptr;
    // ordering without host waits: ev_tile[i] (on device i) marks "tile i's strips have left / arrived" on renderer i's stream,
    // ev_frame (root) marks "the de-interleave has read the gather buffer": the next gather's receive waits for it
    std::vector<hipEvent_t> ev_tile;
    hipEvent_t ev_frame = nullptr;
    bool frame_recorded = false;
    bool gathered_f32 = false, gathered_u8 = false; // planes of the last rt_frame_gather_begin
    bool broken = false; // a collective failed half-way: the communicators are unusable (rt_comm_destroy + rt_comm_create)
};

namespace {
void comm_free_buffers(rt_comm* c) {
    auto fr = [](auto*& p) {
        if (p) (void)hipFree((void*)p);
        p = nullptr;
    };
    fr(c->d_gather_f32), fr(c->d_gather_u8), fr(c->d_full_f32), fr(c->d_full_u8);
}
} // namespace

extern "C" {

int rt_comm_create(int n, const int* devices, rt_comm** out) {
    if (!out) return fail(RT_ERR_INVALID, "null output pointer");
    *out = nullptr;
    if (n <= 0 || n > 64 || !devices) return fail(RT_ERR_INVALID, "bad device list");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device available");
    bool distinct = true;
    for (int i = 0; i < n; ++i) {
        if (devices[i] < 0 || devices[i] >= n_dev) return fail(RT_ERR_INVALID, "device index out of range");
        for (int j = 0; j < i; ++j) distinct = distinct && devices[j] != devices[i];
    }
    return no_throw([&] {
        rt_comm* c = new rt_comm();
        c->devices.assign(devices, devices + n);
        auto bail = [&](int code) {
            rt_comm_destroy(c);
            return code;
        };
        if (hipSetDevice(devices[0]) != hipSuccess || hipStreamCreateWithFlags(&c->root_stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_frame, hipEventDisableTiming) != hipSuccess)
            return bail(fail(RT_ERR_HIP, "cannot create a stream on the root device"));
        c->ev_tile.assign((size_t)n, nullptr);
        for (int i = 0; i < n; ++i)
            if (hipSetDevice(devices[i]) != hipSuccess || hipEventCreateWithFlags(&c->ev_tile[(size_t)i], hipEventDisableTiming) != hipSuccess)
                return bail(fail(RT_ERR_HIP, "cannot create an event on device " + std::to_string(devices[i])));
        (void)hipSetDevice(devices[0]);
        if (distinct) { // one rank per device: RCCL over xGMI
            if (!rccl().ok()) return bail(fail(RT_ERR_UNSUPPORTED, "librccl.so could not be loaded: " + (rccl().err.empty() ? std::string("symbols missing") : rccl().err)));
            c->comms.assign((size_t)n, nullptr);
            const ncclResult_t rc = rccl().CommInitAll(c->comms.data(), n, devices);
            if (rc != ncclSuccess) {
                c->comms.clear();
                return bail(fail(RT_ERR_HIP, std::string("ncclCommInitAll: ") + rccl().GetErrorString(rc)));
            }
            c->use_rccl = true;
        }
        *out = c;
        return (int)RT_OK;
    });
}

void rt_comm_destroy(rt_comm* c) {
    if (!c) return;
    for (size_t i = 0; i < c->comms.size(); ++i)
        if (c->comms[i] && hipSetDevice(c->devices[i]) == hipSuccess) (void)rccl().CommDestroy(c->comms[i]);
    for (size_t i = 0; i < c->ev_tile.size(); ++i)
        if (c->ev_tile[i] && hipSetDevice(c->devices[i]) == hipSuccess) (void)hipEventDestroy(c->ev_tile[i]);
    if (!c->devices.empty() && hipSetDevice(c->devices[0]) == hipSuccess) {
        if (c->root_stream) (void)hipStreamSynchronize(c->root_stream);
        comm_free_buffers(c);
        if (c->ev_frame) (void)hipEventDestroy(c->ev_frame);
        if (c->root_stream) (void)hipStreamDestroy(c->root_stream);
    }
    delete c;
}

int rt_comm_uses_rccl(const rt_comm* c) { return c && c->use_rccl ? 1 : 0; }
const void* rt_comm_frame_f32(const rt_comm* c) { return c ? c->d_full_f32 : nullptr; }
const void* rt_comm_frame_u8(const rt_comm* c) { return c ? c->d_full_u8 : nullptr; }
void* rt_renderer_tile_f32(rt_renderer* r) { return r ? r->d_f32 : nullptr; }
void* rt_renderer_tile_u8(rt_renderer* r) { return r ? r->d_u8 : nullptr; }

// Enqueues the gather of the renderers' last frames and returns at once: nothing in here waits on the host. Order on the device:
//   renderer i's stream : [frame i rendered] -> (root only: wait ev_frame = the previous de-interleave has read the gather buffer)
//                         -> ncclGather send (root: + receive) -> record ev_tile[i]
//   root stream         : wait ev_tile[0] (RCCL: the receive is complete when the root's part of the collective is; device copies:
//                         wait every ev_tile[i], then the copies) -> k_deinterleave -> record ev_frame
// A renderer's next frame is enqueued on its own stream behind its send, so the caller may begin the next frame right after this
// call and collect this one later with rt_comm_wait: the gather overlaps with the next frame's traversal.
int rt_frame_gather_begin(rt_comm* c, rt_renderer* const* rs, int want_f32, int want_u8) {
    if (!c || !rs) return fail(RT_ERR_INVALID, "null communicator or renderer list");
    if (c->broken) return fail(RT_ERR_HIP, "a collective of this communicator failed: destroy it and create a new one");
    const uint32_t n = (uint32_t)c->devices.size();
    const bool f32 = want_f32 != 0, u8 = want_u8 != 0;
    if (!f32 && !u8) return fail(RT_ERR_INVALID, "nothing to gather: ask for the f32 and / or the u8 plane");
    for (uint32_t i = 0; i < n; ++i) {
        const rt_renderer* r = rs[i];
        if (!r) return fail(RT_ERR_INVALID, "null renderer");
        if (r->frame_pending) return fail(RT_ERR_INVALID, "a renderer still has a frame in flight (rt_render_frame_end first)");
        if (r->unusable) return fail(RT_ERR_INVALID, "renderer " + std::to_string(i) + " has no buffers: its last rt_renderer_set_tile / rt_renderer_set_schedule failed (call either again)");
        if (r->scene->device != c->devices[i]) return fail(RT_ERR_INVALID, "renderer " + std::to_string(i) + " is not on the communicator's device " + std::to_string(c->devices[i]));
        if (r->width != rs[0]->width || r->height != rs[0]->height) return fail(RT_ERR_INVALID, "renderers differ in frame size");
        if (r->tile.world != n || r->tile.rank != i || r->tile.strip_rows != rs[0]->tile.strip_rows)
            return fail(RT_ERR_INVALID, "renderer " + std::to_string(i) + " is not tile (" + std::to_string(i) + ", " + std::to_string(n) + ") of the frame");
    }
    return no_throw([&]() -> int {
        const int32_t W = rs[0]->width, H = rs[0]->height;
        const uint32_t strip = rs[0]->tile.strip_rows, rows_max = (uint32_t)rs[0]->tile.local_rows; // rank 0 owns the most rows
        const size_t tile_px = (size_t)rows_max * (size_t)W, frame_px = (size_t)W * (size_t)H;
        HIPCHK(hipSetDevice(c->devices[0]));
        if (c->width != W || c->height != H || c->rows_max != rows_max) {
            HIPCHK(hipStreamSynchronize(c->root_stream));
            comm_free_buffers(c);
            c->width = W, c->height = H, c->rows_max = rows_max;
            c->frame_recorded = false;
        }
        if (f32 && !c->d_full_f32) {
            HIPCHK(hipMalloc((void**)&c->d_gather_f32, std::max<size_t>(1, tile_px * n) * 16));
            HIPCHK(hipMalloc((void**)&c->d_full_f32, std::max<size_t>(1, frame_px) * 16));
        }
        if (u8 && !c->d_full_u8) {
            HIPCHK(hipMalloc((void**)&c->d_gather_u8, std::max<size_t>(1, tile_px * n) * 4));
            HIPCHK(hipMalloc((void**)&c->d_full_u8, std::max<size_t>(1, frame_px) * 4));
        }
        c->gathered_f32 = f32, c->gathered_u8 = u8;
        if (frame_px == 0) return RT_OK;
        if (c->use_rccl) {
            // the previous frame's de-interleave must have read the gather buffer before the root receives into it again
            if (c->frame_recorded) HIPCHK(hipStreamWaitEvent(rs[0]->stream, c->ev_frame, 0));
            // ONE grouped collective per plane: every rank sends rows_max * W pixels from its own stream (ordered behind its
            // render), the root receives n blocks. Single-process multi-device RCCL calls must be grouped (rccl.h:213). An error
            // inside the group is remembered and the group is CLOSED all the same: returning between GroupStart and GroupEnd would
            // leave this thread's group open and every later collective queued instead of issued.
            ncclResult_t first = rccl().GroupStart();
            if (first != ncclSuccess) return fail(RT_ERR_HIP, std::string("ncclGroupStart: ") + rccl().GetErrorString(first));
            std::string what;
            for (uint32_t i = 0; i < n && first == ncclSuccess; ++i) {
                if (hipSetDevice(c->devices[i]) != hipSuccess) {
                    first = ncclUnhandledCudaError, what = "hipSetDevice";
                    break;
                }
                if (f32) first = rccl().Gather(rs[i]->d_f32, i == 0 ? (void*)c->d_gather_f32 : nullptr, tile_px * 4, ncclFloat, 0, c->comms[i], rs[i]->stream), what = "ncclGather (f32)";
                if (u8 && first == ncclSuccess) first = rccl().Gather(rs[i]->d_u8, i == 0 ? (void*)c->d_gather_u8 : nullptr, tile_px * 4, ncclUint8, 0, c->comms[i], rs[i]->stream), what = "ncclGather (u8)";
            }
            const ncclResult_t ge = rccl().GroupEnd();
            (void)hipSetDevice(c->devices[0]);
            if (first != ncclSuccess || ge != ncclSuccess) {
                c->broken = true; // part of the collective may be queued on some ranks' streams and never complete
                return fail(RT_ERR_HIP, (first != ncclSuccess ? what : std::string("ncclGroupEnd")) + ": " + rccl().GetErrorString(first != ncclSuccess ? first : ge) +
                                            " (the communicator is unusable now: rt_comm_destroy + rt_comm_create)");
            }
            HIPCHK(hipEventRecord(c->ev_tile[0], rs[0]->stream)); // the root's receive
            HIPCHK(hipStreamWaitEvent(c->root_stream, c->ev_tile[0], 0));
        } else {
            // the same device more than once (rehearsal): RCCL allows one rank per device, so the blocks move by device copies
            for (uint32_t i = 0; i < n; ++i) {
                HIPCHK(hipSetDevice(c->devices[i]));
                HIPCHK(hipEventRecord(c->ev_tile[i], rs[i]->stream));
            }
            HIPCHK(hipSetDevice(c->devices[0]));
            for (uint32_t i = 0; i < n; ++i) {
                HIPCHK(hipStreamWaitEvent(c->root_stream, c->ev_tile[i], 0));
                if (f32) HIPCHK(hipMemcpyAsync(c->d_gather_f32 + tile_px * i, rs[i]->d_f32, tile_px * 16, hipMemcpyDeviceToDevice, c->root_stream));
                if (u8) HIPCHK(hipMemcpyAsync(c->d_gather_u8 + tile_px * i, rs[i]->d_u8, tile_px * 4, hipMemcpyDeviceToDevice, c->root_stream));
            }
        }
        hipLaunchKernelGGL(k_deinterleave, dim3((uint32_t)((frame_px + 255) / 256)), dim3(256), 0, c->root_stream, (uint32_t)W, (uint32_t)H, n, strip,
                           rows_max, f32 ? c->d_gather_f32 : nullptr, u8 ? c->d_gather_u8 : nullptr, c->d_full_f32, c->d_full_u8);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c->ev_frame, c->root_stream));
        c->frame_recorded = true;
        // A renderer's next frame must not overwrite its tile before it has been read: RCCL's send sits on the renderer's own stream already;
        // the device copies read from the root stream, so the own stream is made to wait for them. Either way an event recorded on the own
        // stream now marks "tile read", and rt_render_frame_begin makes a CALLER's stream wait for it too.
        for (uint32_t i = 0; i < n; ++i) {
            HIPCHK(hipSetDevice(c->devices[i]));
            if (!c->use_rccl) HIPCHK(hipStreamWaitEvent(rs[i]->stream, c->ev_frame, 0));
            HIPCHK(hipEventRecord(rs[i]->ev_tile_read, rs[i]->stream));
            rs[i]->tile_read_recorded = true;
        }
        HIPCHK(hipSetDevice(c->devices[0]));
        return RT_OK;
    });
}

// Waits for the gather enqueued by rt_frame_gather_begin; copies the frame to the host where a pointer is given.
int rt_comm_wait(rt_comm* c, float* rgba_f32, uint8_t* rgba_u8) {
    if (!c) return fail(RT_ERR_INVALID, "null communicator");
    if ((rgba_f32 && !c->gathered_f32) || (rgba_u8 && !c->gathered_u8)) return fail(RT_ERR_INVALID, "that plane was not gathered");
    const size_t frame_px = (size_t)c->width * (size_t)c->height;
    HIPCHK(hipSetDevice(c->devices[0]));
    if (rgba_f32 && frame_px) HIPCHK(hipMemcpyAsync(rgba_f32, c->d_full_f32, frame_px * 16, hipMemcpyDeviceToHost, c->root_stream));
    if (rgba_u8 && frame_px) HIPCHK(hipMemcpyAsync(rgba_u8, c->d_full_u8, frame_px * 4, hipMemcpyDeviceToHost, c->root_stream));
    HIPCHK(hipStreamSynchronize(c->root_stream));
    return RT_OK;
}

int rt_frame_gather(rt_comm* c, rt_renderer* const* rs, float* rgba_f32, uint8_t* rgba_u8, int want_device_f32, int want_device_u8) {
    const int rc = rt_frame_gather_begin(c, rs, rgba_f32 || want_device_f32, rgba_u8 || want_device_u8);
    return rc != RT_OK ? rc : rt_comm_wait(c, rgba_f32, rgba_u8);
}

int rt_comm_size(const rt_comm* c) { return c ? (int)c->devices.size() : 0; }

} // extern "C"
