// rt_probes.hip — rt_intersect_batch and the rt_probe_* entry points (unit probes of the device functions, used by the tests), with their kernels.
#include "rt_internal.h"
#include "rt_probe_kernels.h"

extern "C" {

int rt_intersect_batch(rt_scene* s, uint32_t n, const float* org, const float* dir, float* t, float* u, float* v,
                       uint32_t* tri) {
    if (!s || (n && (!org || !dir || !t || !u || !v || !tri))) return fail(RT_ERR_INVALID, "null argument");
    if (s->device < 0) return fail(RT_ERR_NO_DEVICE, "scene was built host-only (device < 0)");
    if (n == 0) return RT_OK;
    for (uint32_t i = 0; i < n; ++i)
        if (!origin_in_contract_range(s->hs, org + 3 * (size_t)i))
            return fail(RT_ERR_INVALID, "ray " + std::to_string(i) + ": origin more than 100 scene scales outside the scene's bounds (outside the range of the closest-hit contract)");
    HIPCHK(hipSetDevice(s->device));
    DevBuf b_org, b_dir, b_out;
    HIPCHK(b_org.alloc((size_t)n * 12));
    HIPCHK(b_dir.alloc((size_t)n * 12));
    HIPCHK(b_out.alloc((size_t)n * 16));
    float *d_org = b_org.as<float>(), *d_dir = b_dir.as<float>(), *d_out = b_out.as<float>();
    HIPCHK(hipMemcpy(d_org, org, (size_t)n * 12, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_dir, dir, (size_t)n * 12, hipMemcpyHostToDevice));
    float* d_t = d_out;
    float* d_u = d_out + n;
    float* d_v = d_out + 2 * (size_t)n;
    uint32_t* d_tri = (uint32_t*)(d_out + 3 * (size_t)n);
    hipLaunchKernelGGL(k_intersect_batch, dim3((n + 255u) / 256u), dim3(256), 0, 0, s->dev, n, d_org, d_dir, d_t, d_u, d_v, d_tri);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(t, d_t, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(u, d_u, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(v, d_v, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tri, d_tri, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

} // extern "C"

extern "C" {

int rt_probe_xorshift(int device, uint32_t seed, uint32_t n, float* out, uint32_t* state_out) {
    if (!out || !state_out || n == 0) return fail(RT_ERR_INVALID, "bad arguments");
    int rc = device_ok(device);
    if (rc != RT_OK) return rc;
    DevBuf b_out, b_st;
    HIPCHK(b_out.alloc((size_t)n * 4));
    HIPCHK(b_st.alloc(4));
    hipLaunchKernelGGL(k_probe_xorshift, dim3(1), dim3(64), 0, 0, seed, n, b_out.as<float>(), b_st.as<uint32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, b_out.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(state_out, b_st.p, 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_probe_half_roundtrip(int device, uint32_t n, const float* in, float* out, uint16_t* bits_out) {
    if (!in || !out || !bits_out || n == 0) return fail(RT_ERR_INVALID, "bad arguments");
    int rc = device_ok(device);
    if (rc != RT_OK) return rc;
    DevBuf b_in, b_out, b_bits;
    HIPCHK(b_in.alloc((size_t)n * 4));
    HIPCHK(b_out.alloc((size_t)n * 4));
    HIPCHK(b_bits.alloc((size_t)n * 2));
    HIPCHK(hipMemcpy(b_in.p, in, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe_half, dim3((n + 255u) / 256u), dim3(256), 0, 0, n, b_in.as<float>(), b_out.as<float>(), b_bits.as<uint16_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, b_out.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(bits_out, b_bits.p, (size_t)n * 2, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_probe_rounding(int device, uint64_t* mismatches) {
    if (!mismatches) return fail(RT_ERR_INVALID, "bad arguments");
    int rc = device_ok(device);
    if (rc != RT_OK) return rc;
    DevBuf b;
    HIPCHK(b.alloc(16));
    HIPCHK(hipMemset(b.p, 0, 16));
    hipLaunchKernelGGL(k_probe_rounding, dim3(4096), dim3(256), 0, 0, b.as<unsigned long long>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(mismatches, b.p, 16, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_probe_scatter(rt_scene* s, uint32_t material, uint32_t n, const float* dir, const float* normal, const float* uv,
                     const uint32_t* seed, uint8_t* ok, float* out_dir, float* out_att, uint32_t* seed_out) {
    if (!s || !dir || !normal || !uv || !seed || !ok || !out_dir || !out_att || !seed_out || n == 0)
        return fail(RT_ERR_INVALID, "bad arguments");
    if (s->device < 0) return fail(RT_ERR_NO_DEVICE, "scene was built host-only (device < 0)");
    if (material >= s->hs.mats.size()) return fail(RT_ERR_INVALID, "material index out of range");
    HIPCHK(hipSetDevice(s->device));
    DevBuf b_dir, b_nrm, b_uv, b_od, b_oa, b_seed, b_so, b_ok;
    HIPCHK(b_dir.alloc((size_t)n * 12));
    HIPCHK(b_nrm.alloc((size_t)n * 12));
    HIPCHK(b_uv.alloc((size_t)n * 8));
    HIPCHK(b_od.alloc((size_t)n * 12));
    HIPCHK(b_oa.alloc((size_t)n * 12));
    HIPCHK(b_seed.alloc((size_t)n * 4));
    HIPCHK(b_so.alloc((size_t)n * 4));
    HIPCHK(b_ok.alloc((size_t)n));
    HIPCHK(hipMemcpy(b_dir.p, dir, (size_t)n * 12, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b_nrm.p, normal, (size_t)n * 12, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b_uv.p, uv, (size_t)n * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b_seed.p, seed, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe_scatter, dim3((n + 255u) / 256u), dim3(256), 0, 0, s->dev, material, n, b_dir.as<float>(),
                       b_nrm.as<float>(), b_uv.as<float>(), b_seed.as<uint32_t>(), b_ok.as<uint8_t>(), b_od.as<float>(),
                       b_oa.as<float>(), b_so.as<uint32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(ok, b_ok.p, (size_t)n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out_dir, b_od.p, (size_t)n * 12, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out_att, b_oa.p, (size_t)n * 12, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(seed_out, b_so.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

} // extern "C"
