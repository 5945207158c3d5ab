// Micro-benchmark (diagnostics): can the MFMA A-fragment access pattern feed from HBM WITHOUT going through LDS?
// A v_mfma_f32_32x32x16_f16 A fragment is, per lane, 16 contiguous bytes of one row (pixel): lanes 0-31 = 32 rows, lanes
// 32-63 = the next 16 bytes of the same rows.  A wave that loads its fragments straight into registers issues, per
// instruction, 32 segments of 32 bytes at the pixel stride; the K loop walks the segments of a row.  This kernel streams a
// [rows][row_bytes] array that way (NLD loads of 16 B per lane in flight, double buffered) and compares with the fully
// coalesced order (lane i -> base + 16 i).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/frag_stream.hip -o tools/ubench/frag_stream && tools/ubench/frag_stream
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NLD, bool FRAG>
__global__ __launch_bounds__(256) void stream(const char* __restrict__ src, size_t rows, int row_bytes, int iters, float* sink)
{
    const int lane = threadIdx.x & 63, l31 = lane & 31, lhi = lane >> 5;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
    const int kgroups = row_bytes / 32 / NLD;               // groups of NLD K steps per row block (pointer walks, no divisions)
    const size_t nblocks = rows / 32, block_bytes = (size_t)32 * row_bytes;
    f4 b0[NLD], b1[NLD];                                     // two register buffers, named (no runtime-indexed arrays)
    float acc = 0.f;
    size_t rb = wave % nblocks;
    int kg = 0;
    // per-lane offset inside a row block: fragment order = (row l31, 16-byte half lhi of the 32-byte K step);
    // coalesced order = lane i -> 16 i (a K "step" is then just the next KiB of the block)
    const size_t lane_off = FRAG ? (size_t)l31 * row_bytes + lhi * 16 : (size_t)lane * 16;
    const size_t step = FRAG ? 32 : 1024;
    auto issue = [&](f4 (&buf)[NLD]) {
        const char* p = src + rb * block_bytes + lane_off + (size_t)kg * NLD * step;
#pragma unroll
        for (int i = 0; i < NLD; ++i) buf[i] = *reinterpret_cast<const f4*>(p + i * step);
        if (++kg == kgroups) { kg = 0; rb += nwaves; if (rb >= nblocks) rb -= nblocks; }
    };
    auto eat = [&](f4 (&buf)[NLD]) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) acc += buf[i][0] + buf[i][3];
    };
    issue(b0);
    for (int it = 0; it < iters; it += 2) {
        issue(b1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
        eat(b0);
        issue(b0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
        eat(b1);
    }
    if (acc == 12345.678f) sink[0] = acc;
}

template <int NLD, bool FRAG>
void run(const char* d, size_t bytes, int row_bytes, int wg_per_cu, float* sink)
{
    const int blocks = 256 * wg_per_cu, iters = 300;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((stream<NLD, FRAG>), dim3(blocks), dim3(256), 0, 0, d, bytes / row_bytes, row_bytes, 30, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream<NLD, FRAG>), dim3(blocks), dim3(256), 0, 0, d, bytes / row_bytes, row_bytes, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double moved = (double)blocks * 4 * iters * NLD * 1024;
    printf("%-9s row %4d B  loads in flight/lane %2d  WG/CU %d  (%3d KiB in flight per CU): %6.2f TB/s\n", FRAG ? "fragment" : "coalesced",
           row_bytes, NLD, wg_per_cu, NLD * wg_per_cu * 4, moved / (ms * 1e-3) / 1e12);
}

int main()
{
    char* d; float* sink;
    const size_t big = (size_t)2 << 30;
    hipMalloc(&d, big); hipMemset(d, 1, big); hipMalloc(&sink, 64);
    for (int rb : {512, 1024, 2048}) {
        run<8, true>(d, big, rb, 2, sink);  run<8, true>(d, big, rb, 4, sink);
        run<16, true>(d, big, rb, 2, sink); run<16, true>(d, big, rb, 4, sink);
    }
    run<8, false>(d, big, 1024, 2, sink); run<8, false>(d, big, 1024, 4, sink);
    run<16, false>(d, big, 1024, 2, sink); run<16, false>(d, big, 1024, 4, sink);
    return 0;
}
