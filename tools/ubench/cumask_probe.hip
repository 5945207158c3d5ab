// Diagnostics: which XCC / SE / CU does bit i of a hipExtStreamCreateWithCUMask mask select on MI355X?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/cumask_probe.hip -o tools/ubench/cumask_probe && tools/ubench/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void probe(unsigned* out)
{
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // burn a little time so that blocks spread over every allowed CU
    float x = threadIdx.x;
    for (int i = 0; i < 20000; ++i) x = x * 1.000001f + 0.5f;
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hwid; out[blockIdx.x * 2 + 1] = (xcc & 0xf) | (x == 1.f ? 16 : 0); }
}
int main()
{
    unsigned* d; hipMalloc(&d, 4096 * 8);
    auto run = [&](const std::vector<unsigned>& mask, const char* name) {
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, (unsigned)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
        hipLaunchKernelGGL(probe, dim3(2048), dim3(64), 0, s, d);
        hipStreamSynchronize(s);
        std::vector<unsigned> h(4096);
        hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost);
        std::set<unsigned> xccs, cus;
        for (int i = 0; i < 2048; ++i) {
            const unsigned hw = h[i * 2], xcc = h[i * 2 + 1] & 0xf;
            xccs.insert(xcc);
            cus.insert((xcc << 16) | (((hw >> 13) & 0x7) << 8) | ((hw >> 8) & 0xf));     // xcc | se_id | cu_id
        }
        printf("%-28s xccs {", name);
        for (unsigned x : xccs) printf("%u ", x);
        printf("}  distinct (xcc,se,cu) = %zu\n", cus.size());
        hipStreamDestroy(s);
    };
    run(std::vector<unsigned>(8, 0xffffffffu), "all 256 bits");
    { std::vector<unsigned> m(8, 0); for (int i = 0; i < 4; ++i) m[i] = 0xffffffffu; run(m, "bits 0..127"); }
    { std::vector<unsigned> m(8, 0); for (int i = 4; i < 8; ++i) m[i] = 0xffffffffu; run(m, "bits 128..255"); }
    { std::vector<unsigned> m(8, 0); m[0] = 0xffffffffu; run(m, "bits 0..31"); }
    { std::vector<unsigned> m(8, 0); m[1] = 0xffffffffu; run(m, "bits 32..63"); }
    { std::vector<unsigned> m(8, 0x55555555u); run(m, "even bits"); }
    { std::vector<unsigned> m(8, 0x01010101u); run(m, "bits = 0 mod 8"); }
    { std::vector<unsigned> m(8, 0x03030303u); run(m, "bits = 0,1 mod 8"); }
    { std::vector<unsigned> m(8, 0x0f0f0f0fu); run(m, "bits = 0..3 mod 8"); }
    { std::vector<unsigned> m(8, 0); m[0] = 1; run(m, "bit 0"); }
    { std::vector<unsigned> m(8, 0); m[0] = 2; run(m, "bit 1"); }
    { std::vector<unsigned> m(8, 0); m[0] = 0x100; run(m, "bit 8"); }
    return 0;
}
