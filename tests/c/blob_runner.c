/* blob_runner.c -- SMAP.forward (model/smap.py:403-419) from include/smap_hip.h ALONE: no Python, no torch.
 *
 *     blob_runner plan.blob input.f32 output.f32
 *
 * plan.blob   the serialised schedule (smap_amd/engine.py::BackboneEngine.blob(): header | smap_op[n] | weights)
 * input.f32   [frames,3,H,W] fp32 NCHW, raw
 * output.f32  written: the fp32 output buffer (hms | det_d | root_d at the offsets smap_blob_info names, + the status word)
 *
 * Built and run by tests/test_abi_gpu.py (gcc + libamdhip64 + libsmap_hip.so) against tests/golden/backbone_small.npz.
 * Plain C on purpose: it is what a non-Python host of the reference's forward would write. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "smap_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static void* slurp(const char* path, size_t* n)
{
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    void* p = malloc((size_t)sz);
    if (p && fread(p, 1, (size_t)sz, f) != (size_t)sz) { free(p); p = NULL; }
    fclose(f);
    *n = (size_t)sz;
    return p;
}

int main(int argc, char** argv)
{
    if (argc != 4) { fprintf(stderr, "usage: %s plan.blob input.f32 output.f32\n", argv[0]); return 1; }
    size_t nb = 0, ni = 0;
    void* blob = slurp(argv[1], &nb);
    void* in = slurp(argv[2], &ni);
    if (!blob || !in) { fprintf(stderr, "cannot read inputs\n"); return 1; }
    smap_plan* plan = NULL;
    smap_blob_info info;
    int rc = smap_plan_create_from_blob(blob, nb, &plan, &info);
    if (rc) { fprintf(stderr, "smap_plan_create_from_blob: %d\n", rc); return 3; }
    if (ni != (size_t)info.frames * 3 * info.H * info.W * 4) { fprintf(stderr, "input size mismatch\n"); return 1; }
    int64_t arena_b = 0, out_b = 0;
    smap_workspace_bytes(plan, &arena_b, &out_b);
    if (arena_b > info.arena_bytes || out_b > info.out_bytes) { fprintf(stderr, "workspace mismatch\n"); return 3; }
    void *d_arena, *d_w, *d_in, *d_out;
    CHECK_HIP(hipMalloc(&d_arena, (size_t)info.arena_bytes));
    CHECK_HIP(hipMalloc(&d_w, (size_t)info.weights_bytes));
    CHECK_HIP(hipMalloc(&d_in, ni));
    CHECK_HIP(hipMalloc(&d_out, (size_t)info.out_bytes));
    CHECK_HIP(hipMemcpy(d_w, (const char*)blob + info.weights_offset, (size_t)info.weights_bytes, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_in, in, ni, hipMemcpyHostToDevice));
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    rc = smap_plan_run(plan, (const float*)d_in, d_arena, d_w, (float*)d_out, (void*)st);
    if (rc) { fprintf(stderr, "smap_plan_run: %d\n", rc); return 3; }
    CHECK_HIP(hipStreamSynchronize(st));
    void* out = malloc((size_t)info.out_bytes);
    CHECK_HIP(hipMemcpy(out, d_out, (size_t)info.out_bytes, hipMemcpyDeviceToHost));
    FILE* f = fopen(argv[3], "wb");
    if (!f || fwrite(out, 1, (size_t)info.out_bytes, f) != (size_t)info.out_bytes) { fprintf(stderr, "cannot write output\n"); return 1; }
    fclose(f);
    printf("frames %d %dx%d -> maps %dx%d, channels %d/%d/%d, precision %d, arena %lld B, offsets %lld %lld %lld status %lld (%s)\n",
           info.frames, info.H, info.W, info.out_h, info.out_w, info.n_hms, info.n_det, info.n_root, info.precision,
           (long long)info.arena_bytes, (long long)info.hms_off, (long long)info.det_off, (long long)info.root_off,
           (long long)info.status_off, smap_version());
    smap_plan_destroy(plan);
    return 0;
}
