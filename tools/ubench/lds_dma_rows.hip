// Micro-benchmark (diagnostics): LDS-DMA (global_load_lds dwordx4) throughput per CU for the ROW-SEGMENT access pattern of the conv
// kernels, as a function of the contiguous bytes per row (64 B = one BK=32 chunk of one plane; 128 B = a full cache line),
// of where the rows live (one small matrix every workgroup re-reads = weights in L2; distinct rows per workgroup = activations
// from HBM) and of how many waves issue.   hipcc --offload-arch=gfx950 -O3 lds_dma_rows.hip -o lds_dma_rows && ./lds_dma_rows
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// One workgroup per CU, NW waves.  A "K step" stages ROWS rows x SEG bytes (row stride RS bytes) into one of 3 LDS stages
// (2 steps in flight); after RS/SEG steps the workgroup moves to its next block of ROWS rows.  shared: every workgroup
// walks the SAME rows (weights); else workgroup b owns rows [b*ROWS*blocks_per_wg ...).
template <int SEG, int NW, int ROWS>
__global__ __launch_bounds__(NW * 64) void rows_kernel(const char* __restrict__ src, int RS, int nblocks, int shared, size_t span, float* sink)
{
    constexpr int SPR = SEG / 16, RPW = 64 / SPR, RPR = NW * RPW, LPT = ROWS / RPR;
    constexpr int STAGE = ROWS * SEG;
    __shared__ __attribute__((aligned(16))) char smem[3 * STAGE];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lrow = lane / SPR, lslot = lane % SPR, srow = wave * RPW + lrow;
    const int ksteps = RS / SEG, total = nblocks * ksteps;
    int blk = 0, ks = 0;
    auto issue = [&](int stage) {
        size_t row0 = shared ? (size_t)(blk % 2) * ROWS : ((size_t)blockIdx.x * nblocks + blk) * ROWS;
        const char* g = src + ((row0 + srow) * (size_t)RS + (size_t)ks * SEG + lslot * 16) % span;
        char* s = smem + stage * STAGE + wave * RPW * SEG;
#pragma unroll
        for (int i = 0; i < LPT; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(g + (size_t)i * RPR * RS), (lds_void*)(s + i * RPR * SEG), 16, 0, 0);
        if (++ks == ksteps) { ks = 0; ++blk; }
    };
    issue(0); issue(1);
    float acc = 0.f;
    for (int it = 0; it < total; ++it) {
        if (it + 2 <= total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        acc += reinterpret_cast<const float*>(smem + (it % 3) * STAGE)[lane];
        if (it + 2 < total) issue((it + 2) % 3);
    }
    if (acc == 12345.678f) sink[0] = acc;
}

template <int SEG, int NW, int ROWS>
void run(const char* d, size_t span, int RS, int shared, float* sink)
{
    const int nblocks = 24, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rows_kernel<SEG, NW, ROWS>), dim3(grid), dim3(NW * 64), 0, 0, d, RS, 4, shared, span, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((rows_kernel<SEG, NW, ROWS>), dim3(grid), dim3(NW * 64), 0, 0, d, RS, nblocks, shared, span, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * nblocks * ROWS * RS;
    printf("%-6s seg %3d B  row stride %4d  rows/step %3d  waves %d : %6.2f TB/s  (%5.1f B/clk/CU at 2.0 GHz)\n", shared ? "shared" : "stream",
           SEG, RS, ROWS, NW, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 256 / 2.0e9);
}

int main()
{
    char* d; float* sink;
    const size_t big = (size_t)3 << 30;
    hipMalloc(&d, big); hipMemset(d, 1, big); hipMalloc(&sink, 64);
    for (int shared = 1; shared >= 0; --shared) {
        for (int RS : {512, 1024}) {
            run<64, 4, 384>(d, big, RS, shared, sink);
            run<128, 4, 384>(d, big, RS, shared, sink);
            run<64, 8, 384>(d, big, RS, shared, sink);
            run<128, 8, 384>(d, big, RS, shared, sink);
            run<64, 4, 128>(d, big, RS, shared, sink);
            run<128, 4, 128>(d, big, RS, shared, sink);
        }
    }
    return 0;
}
