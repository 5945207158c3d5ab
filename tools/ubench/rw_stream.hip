// Micro-benchmark (diagnostics): what does HBM sustain for a READ + WRITE mix (the shape of every conv layer: read an
// activation tensor, write one of similar size)?  Streams `rd` KiB-blocks in and `wr` KiB-blocks out per iteration with
// 16-byte per-lane accesses, NLD accesses of each kind in flight per lane, over disjoint 1 GiB regions.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/rw_stream.hip -o tools/ubench/rw_stream && tools/ubench/rw_stream
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NRD, int NWR>
__global__ __launch_bounds__(256) void rw(const char* __restrict__ src, char* __restrict__ dst, size_t span, int iters, float* sink)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
    size_t ro = wave * (NRD > 0 ? NRD : 1) * 1024, wo = wave * (NWR > 0 ? NWR : 1) * 1024;
    const size_t rstep = nwaves * (NRD > 0 ? NRD : 1) * 1024, wstep = nwaves * (NWR > 0 ? NWR : 1) * 1024;
    f4 v[NRD > 0 ? NRD : 1];
    f4 acc = {1.f, 2.f, 3.f, 4.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NRD; ++i) v[i] = *reinterpret_cast<const f4*>(src + ro + i * 1024 + lane * 16);
#pragma unroll
        for (int i = 0; i < NWR; ++i) *reinterpret_cast<f4*>(dst + wo + i * 1024 + lane * 16) = acc;
#pragma unroll
        for (int i = 0; i < NRD; ++i) acc += v[i];
        ro += rstep; if (ro + NRD * 1024 > span) ro -= span - NRD * 1024 > ro ? 0 : (span / rstep) * rstep;
        wo += wstep;
        if (ro + (size_t)NRD * 1024 > span) ro = wave * NRD * 1024;
        if (wo + (size_t)NWR * 1024 > span) wo = wave * NWR * 1024;
    }
    if (acc[0] == 12345.678f) sink[0] = acc[1];
}

template <int NRD, int NWR>
void run(const char* s, char* d, size_t span, int wg_per_cu, float* sink)
{
    const int blocks = 256 * wg_per_cu, iters = 400;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((rw<NRD, NWR>), dim3(blocks), dim3(256), 0, 0, s, d, span, 40, sink);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((rw<NRD, NWR>), dim3(blocks), dim3(256), 0, 0, s, d, span, iters, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double rb = (double)blocks * 4 * iters * NRD * 1024, wb = (double)blocks * 4 * iters * NWR * 1024;
    printf("read %2d KiB + write %2d KiB per wave-iteration, WG/CU %d : read %5.2f + write %5.2f = %5.2f TB/s\n", NRD, NWR, wg_per_cu,
           rb / (ms * 1e-3) / 1e12, wb / (ms * 1e-3) / 1e12, (rb + wb) / (ms * 1e-3) / 1e12);
}

int main()
{
    char *s, *d; float* sink;
    const size_t span = (size_t)1 << 30;
    (void)hipMalloc(&s, span); (void)hipMemset(s, 1, span); (void)hipMalloc(&d, span); (void)hipMemset(d, 0, span); (void)hipMalloc(&sink, 64);
    for (int w : {2, 4}) {
        run<8, 0>(s, d, span, w, sink);
        run<0, 8>(s, d, span, w, sink);
        run<8, 8>(s, d, span, w, sink);
        run<8, 4>(s, d, span, w, sink);
        run<4, 8>(s, d, span, w, sink);
        run<8, 2>(s, d, span, w, sink);
    }
    return 0;
}
