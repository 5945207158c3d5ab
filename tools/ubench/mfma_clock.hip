// Micro-benchmark (diagnostics, not shipped): actual shader clock under load, issue cost of
// v_mfma_f32_32x32x16_f16 (independent / dependent chains), ds_read_b128 and VALU rates with 1..2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_clock.hip -o tools/ubench/mfma_clock && tools/ubench/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// mode 0: 4 independent MFMA chains; 1: one dependent chain; 2: ds_read_b128 stream; 3: v_pk_add_f32 stream; 4: MFMA + ds_read interleaved
__global__ void probe(int mode, int iters, long long* out, float* sink)
{
    __shared__ __attribute__((aligned(16))) char lds[32768];
    const int lane = threadIdx.x & 63;
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(lane * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = i;
    __syncthreads();
    float4 acc4 = {0, 0, 0, 0};
    float2 p = {1.f, 2.f}, q = {0.5f, 0.25f};
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (mode == 0 || mode == 4) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        }
        if (mode == 1) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        }
        if (mode == 2 || mode == 4) {
            const float4 v0 = *reinterpret_cast<const float4*>(lds + ((lane * 16 + it * 1024) & 32767));
            const float4 v1 = *reinterpret_cast<const float4*>(lds + ((lane * 16 + it * 1024 + 16384) & 32767));
            acc4.x += v0.x + v1.x; acc4.y += v0.y + v1.y;
            if (mode == 2) {
                const float4 v2 = *reinterpret_cast<const float4*>(lds + ((lane * 16 + it * 1024 + 8192) & 32767));
                const float4 v3 = *reinterpret_cast<const float4*>(lds + ((lane * 16 + it * 1024 + 24576) & 32767));
                acc4.z += v2.z + v3.z; acc4.w += v2.w + v3.w;
            }
        }
        if (mode == 3) {
#pragma unroll
            for (int u = 0; u < 16; ++u) { p.x = p.x * q.x + q.y; p.y = p.y * q.y + q.x; q.x += 1e-9f; }
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = w1 - w0; }
    float s = acc4.x + acc4.y + acc4.z + acc4.w + p.x + p.y;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 12345.678f) sink[0] = s;
}

int main()
{
    long long* d; float* sink;
    hipMalloc(&d, 4096 * 16); hipMalloc(&sink, 64);
    const char* names[] = {"mfma 4 indep", "mfma dependent", "ds_read_b128 x4", "valu fma x48", "mfma x4 + ds_read x2"};
    const int iters = 20000;
    for (int wps = 1; wps <= 4; ++wps)
        for (int mode = 0; mode < 5; ++mode) {
            const int blocks = 256 * wps, threads = 256;                // wps blocks of 4 waves per CU -> wps waves per SIMD
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, mode, iters, d, sink);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, mode, iters, d, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> h(blocks * 2);
            hipMemcpy(h.data(), d, blocks * 16, hipMemcpyDeviceToHost);
            double cyc = 0, wall = 0;
            for (int i = 0; i < blocks; ++i) { cyc += h[i * 2]; wall += h[i * 2 + 1]; }
            cyc /= blocks; wall /= blocks;
            if (mode == 0 || mode == 1 || mode == 4)
                printf("   -> %.2f PFLOP/s aggregate\n", (double)blocks * 4 * iters * 4 * 32768.0 / (ms * 1e-3) / 1e15);
            printf("%d wave(s)/SIMD  %-22s kernel %.3f ms  clock64 %.0f  wall(100MHz) %.0f  => clock64 rate %.0f MHz, %.1f clock64 ticks/iter, %.1f ns/iter\n",
                   wps, names[mode], ms, cyc, wall, cyc / wall * 100.0, cyc / iters, wall * 10.0 / iters);
        }
    return 0;
}
