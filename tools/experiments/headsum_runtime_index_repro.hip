// Reproducer attempt for EXPERIMENTS.md R3.6: the round-1/2 head-sum kernel indexed its by-value argument struct with a RUN-TIME
// index (-> scalar loads of kernel arguments inside the pixel loop, a private Lerp array promoted to LDS, the AQL dispatch packet
// read to address it) and returned ~15 wrong values in 10-30 % of the launches that overlapped another stream's kernels.
//   V = 0  the old kernel, verbatim
//   V = 1  run-time index into the argument struct kept, the Lerp array replaced by per-source scalars
//   V = 2  the fixed kernel (compile-time source index)
// Build / run:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/headsum_runtime_index_repro.hip -o /tmp/hs && /tmp/hs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp lerp_index(int dst, int in_size, int out_size)
{
    Lerp r;
    if (in_size == out_size) { r.i0 = dst; r.i1 = dst; r.l0 = 1.f; r.l1 = 0.f; return r; }
    const float scale = out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
    const float src = scale * (float)dst;
    r.i0 = (int)src;
    r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
    float l1 = src - (float)r.i0;
    l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
    r.l1 = l1;
    r.l0 = 1.f - l1;
    return r;
}
struct HeadSrc { const float* p[3]; int h[3], w[3]; int n; };
constexpr int HS_PX = 32;

__device__ __forceinline__ float4 bil(const float* base, const Lerp& ly, const Lerp& lx, int sw, int Cs)
{
    const float4 v00 = *reinterpret_cast<const float4*>(base + ((size_t)ly.i0 * sw + lx.i0) * Cs);
    const float4 v01 = *reinterpret_cast<const float4*>(base + ((size_t)ly.i0 * sw + lx.i1) * Cs);
    const float4 v10 = *reinterpret_cast<const float4*>(base + ((size_t)ly.i1 * sw + lx.i0) * Cs);
    const float4 v11 = *reinterpret_cast<const float4*>(base + ((size_t)ly.i1 * sw + lx.i1) * Cs);
    float4 up;
    up.x = ly.l0 * (lx.l0 * v00.x + lx.l1 * v01.x) + ly.l1 * (lx.l0 * v10.x + lx.l1 * v11.x);
    up.y = ly.l0 * (lx.l0 * v00.y + lx.l1 * v01.y) + ly.l1 * (lx.l0 * v10.y + lx.l1 * v11.y);
    up.z = ly.l0 * (lx.l0 * v00.z + lx.l1 * v01.z) + ly.l1 * (lx.l0 * v10.z + lx.l1 * v11.z);
    up.w = ly.l0 * (lx.l0 * v00.w + lx.l1 * v01.w) + ly.l1 * (lx.l0 * v10.w + lx.l1 * v11.w);
    return up;
}

// V = 0: as shipped in rounds 1-2
__device__ __forceinline__ float4 value_v0(const HeadSrc& s, const Lerp* ly, int b, int y, int x, int c, int Ho, int Wo, int Cs)
{
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < s.n; ++k) {
        const float* base = s.p[k] + (size_t)b * s.h[k] * s.w[k] * Cs + c;
        float4 up;
        if (s.h[k] == Ho && s.w[k] == Wo) up = *reinterpret_cast<const float4*>(base + ((size_t)y * Wo + x) * Cs);
        else up = bil(base, ly[k], lerp_index(x, s.w[k], Wo), s.w[k], Cs);
        if (k == 0) v = up;
        else { v.x += up.x; v.y += up.y; v.z += up.z; v.w += up.w; }
    }
    return v;
}
// V = 1: run-time index into the struct, no private array (the row weights are recomputed per source)
__device__ __forceinline__ float4 value_v1(const HeadSrc& s, int b, int y, int x, int c, int Ho, int Wo, int Cs)
{
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < s.n; ++k) {
        const float* base = s.p[k] + (size_t)b * s.h[k] * s.w[k] * Cs + c;
        float4 up;
        if (s.h[k] == Ho && s.w[k] == Wo) up = *reinterpret_cast<const float4*>(base + ((size_t)y * Wo + x) * Cs);
        else up = bil(base, lerp_index(y, s.h[k], Ho), lerp_index(x, s.w[k], Wo), s.w[k], Cs);
        if (k == 0) v = up;
        else { v.x += up.x; v.y += up.y; v.z += up.z; v.w += up.w; }
    }
    return v;
}
// V = 2: compile-time source index
template <int K>
__device__ __forceinline__ float4 source_v2(const HeadSrc& s, int b, int y, int x, int c, int Ho, int Wo, int Cs)
{
    const int sh = s.h[K], sw = s.w[K];
    const float* base = s.p[K] + (size_t)b * sh * sw * Cs + c;
    if (sh == Ho && sw == Wo) return *reinterpret_cast<const float4*>(base + ((size_t)y * Wo + x) * Cs);
    return bil(base, lerp_index(y, sh, Ho), lerp_index(x, sw, Wo), sw, Cs);
}

template <int V>
__global__ __launch_bounds__(256) void headsum(HeadSrc s, float* __restrict__ out, int Ho, int Wo, int C, int Cs)
{
    __shared__ float tile[48 * (HS_PX + 1)];
    const int b = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * HS_PX, tid = threadIdx.x;
    Lerp ly[3];
    if (V == 0)
        for (int k = 0; k < s.n; ++k) ly[k] = lerp_index(y, s.h[k], Ho);
    const int G = Cs >> 2;
    for (int idx = tid; idx < HS_PX * G; idx += 256) {
        const int px = idx / G, c = (idx - px * G) * 4;
        const int x = x0 + px;
        if (x >= Wo || c >= C) continue;
        float4 v;
        if (V == 0) v = value_v0(s, ly, b, y, x, c, Ho, Wo, Cs);
        else if (V == 1) v = value_v1(s, b, y, x, c, Ho, Wo, Cs);
        else {
            v = source_v2<0>(s, b, y, x, c, Ho, Wo, Cs);
            if (s.n > 1) { const float4 u = source_v2<1>(s, b, y, x, c, Ho, Wo, Cs); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
            if (s.n > 2) { const float4 u = source_v2<2>(s, b, y, x, c, Ho, Wo, Cs); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
        }
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < C) tile[(c + e) * (HS_PX + 1) + px] = vv[e];
    }
    __syncthreads();
    for (int idx = tid; idx < C * HS_PX; idx += 256) {
        const int c = idx / HS_PX, px = idx - c * HS_PX;
        const int x = x0 + px;
        if (x >= Wo) continue;
        out[(((size_t)b * C + c) * Ho + y) * Wo + x] = tile[c * (HS_PX + 1) + px];
    }
}

// the neighbour: a long kernel with a large LDS footprint, like the stem / GEMM kernels of the schedule
__global__ __launch_bounds__(256) void neighbour(float* sink, int iters)
{
    __shared__ float buf[15 * 1024];
    for (int i = threadIdx.x; i < 15 * 1024; i += 256) buf[i] = (float)i;
    __syncthreads();
    float a = (float)threadIdx.x;
    for (int i = 0; i < iters; ++i) a = a * 1.0000001f + buf[(threadIdx.x * 7 + i) % (15 * 1024)];
    if (a == 12345.678f) sink[blockIdx.x] = a;
}

template <int V>
int trial(const HeadSrc& s, float* out, const float* ref_h, size_t n_out, int B, int Ho, int Wo, int C, int Cs, hipStream_t s0, hipStream_t s1,
          float* sink, int rounds, bool overlap)
{
    std::vector<float> h(n_out);
    int bad = 0;
    for (int r = 0; r < rounds; ++r) {
        if (overlap)
            for (int j = 0; j < 6; ++j) hipLaunchKernelGGL(neighbour, dim3(2048), dim3(256), 0, s1, sink, 4000);
        for (int j = 0; j < 8; ++j)
            hipLaunchKernelGGL(headsum<V>, dim3((Wo + HS_PX - 1) / HS_PX, Ho, B), dim3(256), 0, s0, s, out, Ho, Wo, C, Cs);
        (void)hipDeviceSynchronize();
        hipMemcpy(h.data(), out, n_out * 4, hipMemcpyDeviceToHost);
        bad += memcmp(h.data(), ref_h, n_out * 4) != 0;
    }
    return bad;
}

// C entry for tools/debug/headsum_repro_next_to_schedule.py: variant V of the head sum on `stream` (the library's real kernels run
// next to it from Python)
extern "C" int repro_headsum(int V, const float* p0, const float* p1, const float* p2, float* out, int B, void* stream)
{
    const int Ho = 128, Wo = 208, C = 43, Cs = 48;
    HeadSrc s;
    s.n = 3;
    s.p[0] = p0; s.p[1] = p1; s.p[2] = p2;
    s.h[0] = 128; s.h[1] = 64; s.h[2] = 32;
    s.w[0] = 208; s.w[1] = 104; s.w[2] = 52;
    const dim3 grid((Wo + HS_PX - 1) / HS_PX, Ho, B);
    hipStream_t st = (hipStream_t)stream;
    if (V == 0) hipLaunchKernelGGL(headsum<0>, grid, dim3(256), 0, st, s, out, Ho, Wo, C, Cs);
    else if (V == 1) hipLaunchKernelGGL(headsum<1>, grid, dim3(256), 0, st, s, out, Ho, Wo, C, Cs);
    else hipLaunchKernelGGL(headsum<2>, grid, dim3(256), 0, st, s, out, Ho, Wo, C, Cs);
    return (int)hipGetLastError();
}

int main()
{
    const int B = 8, Ho = 128, Wo = 208, C = 43, Cs = 48;
    const int hs[3] = {128, 64, 32}, ws[3] = {208, 104, 52};
    HeadSrc s;
    s.n = 3;
    srand(1);
    for (int k = 0; k < 3; ++k) {
        const size_t n = (size_t)B * hs[k] * ws[k] * Cs;
        std::vector<float> h(n);
        for (auto& v : h) v = (float)(rand() % 20001 - 10000) * 0.01f;
        float* d;
        hipMalloc(&d, n * 4);
        hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        s.p[k] = d; s.h[k] = hs[k]; s.w[k] = ws[k];
    }
    const size_t n_out = (size_t)B * C * Ho * Wo;
    float *out, *sink;
    hipMalloc(&out, n_out * 4);
    hipMalloc(&sink, 2048 * 4);
    hipStream_t s0, s1;
    hipStreamCreate(&s0);
    hipStreamCreate(&s1);
    std::vector<float> ref(n_out);
    hipLaunchKernelGGL(headsum<2>, dim3((Wo + HS_PX - 1) / HS_PX, Ho, B), dim3(256), 0, s0, s, out, Ho, Wo, C, Cs);
    hipDeviceSynchronize();
    hipMemcpy(ref.data(), out, n_out * 4, hipMemcpyDeviceToHost);
    const int R = 40;
    printf("alone      : V0 %d  V1 %d  V2 %d  of %d rounds differ from the reference\n",
           trial<0>(s, out, ref.data(), n_out, B, Ho, Wo, C, Cs, s0, s1, sink, R, false),
           trial<1>(s, out, ref.data(), n_out, B, Ho, Wo, C, Cs, s0, s1, sink, R, false),
           trial<2>(s, out, ref.data(), n_out, B, Ho, Wo, C, Cs, s0, s1, sink, R, false), R);
    printf("overlapped : V0 %d  V1 %d  V2 %d  of %d rounds differ from the reference\n",
           trial<0>(s, out, ref.data(), n_out, B, Ho, Wo, C, Cs, s0, s1, sink, R, true),
           trial<1>(s, out, ref.data(), n_out, B, Ho, Wo, C, Cs, s0, s1, sink, R, true),
           trial<2>(s, out, ref.data(), n_out, B, Ho, Wo, C, Cs, s0, s1, sink, R, true), R);
    return 0;
}
