// Micro-benchmark (diagnostics): what does the LDS-DMA path (global_load_lds dwordx4) sustain on MI355X as a function of the
// bytes in flight per CU and of where the data lives (a working set that fits the L2s / one that only fits HBM)?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dma_stream.hip -o tools/ubench/lds_dma_stream && tools/ubench/lds_dma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// Each workgroup (256 threads) streams `iters` tiles of TILE_KB KiB through STAGES LDS stages: STAGES-1 tiles in flight.
template <int TILE_KB, int STAGES>
__global__ __launch_bounds__(256) void stream(const char* __restrict__ src, size_t span, int iters, float* sink)
{
    constexpr int TILE = TILE_KB * 1024, LPT = TILE / 4096;             // DMA instructions per thread per tile (16 B x 256 threads)
    __shared__ __attribute__((aligned(16))) char smem[STAGES * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    size_t off = ((size_t)blockIdx.x * 977 % (span / TILE)) * TILE;     // scattered start, tile-aligned
    auto issue = [&](int stage) {
        const char* g = src + off + (size_t)tid * 16;
        char* s = smem + stage * TILE + wave * 1024;
#pragma unroll
        for (int i = 0; i < LPT; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(g + i * 4096), (lds_void*)(s + i * 4096), 16, 0, 0);
        off += TILE;
        if (off + TILE > span) off = 0;
    };
    for (int s = 0; s < STAGES - 1; ++s) issue(s);
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (STAGES == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (STAGES == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
        __builtin_amdgcn_s_barrier();
        acc += reinterpret_cast<const float*>(smem + (it % STAGES) * TILE)[lane];      // touch the landed tile
        issue((it + STAGES - 1) % STAGES);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 12345.678f) sink[0] = acc;
}

template <int TILE_KB, int STAGES>
void run(const char* d, size_t span, const char* where, int blocks_per_cu, float* sink)
{
    const int blocks = 256 * blocks_per_cu, iters = 400;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((stream<TILE_KB, STAGES>), dim3(blocks), dim3(256), 0, 0, d, span, 50, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream<TILE_KB, STAGES>), dim3(blocks), dim3(256), 0, 0, d, span, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * iters * TILE_KB * 1024;
    printf("%-4s tile %2d KiB  stages %d  blocks/CU %d  in flight/CU %3d KiB : %6.2f TB/s\n", where, TILE_KB, STAGES, blocks_per_cu,
           (STAGES - 1) * TILE_KB * blocks_per_cu, bytes / (ms * 1e-3) / 1e12);
}

int main()
{
    char* d; float* sink;
    const size_t big = (size_t)2 << 30, small = (size_t)64 << 20;      // 2 GiB: HBM; 64 MiB: Infinity Cache, not the 4 MiB L2s
    hipMalloc(&d, big); hipMemset(d, 1, big); hipMalloc(&sink, 64);
    const size_t tiny = (size_t)2 << 20;                                 // 2 MiB: fits every single L2
    for (int pass = 0; pass < 3; ++pass) {
        const size_t span = pass == 0 ? tiny : pass == 1 ? small : big;
        const char* w = pass == 0 ? "L2" : pass == 1 ? "MALL" : "HBM";
        run<8, 2>(d, span, w, 2, sink);  run<8, 2>(d, span, w, 4, sink);  run<8, 2>(d, span, w, 8, sink);
        run<16, 2>(d, span, w, 2, sink); run<16, 2>(d, span, w, 4, sink);
        run<16, 3>(d, span, w, 2, sink); run<16, 3>(d, span, w, 3, sink);
        run<16, 4>(d, span, w, 1, sink); run<16, 4>(d, span, w, 2, sink); run<16, 4>(d, span, w, 3, sink);
        run<32, 2>(d, span, w, 1, sink); run<32, 2>(d, span, w, 2, sink); run<32, 3>(d, span, w, 2, sink);
    }
    return 0;
}
