// assoc.hip -- depth-aware part association + 3D lifting for SMAP on gfx950.
//
// Hand-written HIP for CDNA4 (wave64).  Compiled with -ffp-contract=off: every
// float op below rounds exactly once, in the order written, so results are
// bit-identical to the scalar CPU statement of the same algorithm.
//
// Reference semantics (zju3dv/SMAP):
//   nms_kernel        extensions/gpu/nmsBase.cu:10-175
//   paf_score_kernel  extensions/gpu/bodyPartConnectorBase.cu:11-63,104-189
//   group_kernel      extensions/association.cpp:123-233
//   lift_kernel       exps/stage3_root2/test.py:116-134, test_util.py:45-99, lib/utils/post_3d.py
//   refine_kernel     exps/stage3_root2/test_util.py:102-131, model/refinenet.py
//
// Data layout (all fp32 unless noted, all resident in HBM, batched over frames):
//   hms    [B,43,H,W]   0..14 keypoint heat-maps, 15+2l / 16+2l PAF x / y of limb l
//   peaks  [B,15,128,3] slot 0 = (count,0,0); slot r+1 = (x+.5, y+.5, score) of the r-th
//                        raster-order peak; slots > count are zero
//   scores [B,14,127,127]
//   bodys  [B,127,15,4] (x, y, 0, score) heat-map pixels, persons sorted near -> far
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "smap_hip.h"

namespace {

constexpr int NJ = SMAP_NJ, NL = SMAP_NL, MAXP = SMAP_MAXP, PSTRIDE = MAXP + 1;

__constant__ int c_pairs[2 * NL] = {0, 1, 0, 2, 0, 9, 9, 10, 10, 11, 0, 3, 3, 4,
                                    4, 5, 2, 12, 12, 13, 13, 14, 2, 6, 6, 7, 7, 8};
__constant__ float c_bone_length[NL] = {
    26.42178982f, 48.36980909f, 14.88291009f, 31.28002332f, 23.915707f,
    14.97674918f, 31.28002549f, 23.91570732f, 12.4644364f,  48.26604433f,
    39.03553194f, 12.4644364f,  48.19076948f, 39.03553252f};

inline int hip_rc(hipError_t e) { return e == hipSuccess ? 0 : -(1000 + (int)e); }

// ------------------------------------------------------------------ scale --
__global__ void scale_hms_kernel(float* __restrict__ hms, int HW4, int total4)
{
    // one float4 per thread; channel = (i / HW4) % 43 decides the divisor
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += gridDim.x * blockDim.x) {
        const int c = (i / HW4) % SMAP_HMS_C;
        const float d = c < NJ ? 255.f : 127.f;
        float4 v = reinterpret_cast<float4*>(hms)[i];
        v.x = v.x / d; v.y = v.y / d; v.z = v.z / d; v.w = v.w / d;
        reinterpret_cast<float4*>(hms)[i] = v;
    }
}

__global__ void scale_hms_kernel_scalar(float* __restrict__ hms, int HW, int total)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c = (i / HW) % SMAP_HMS_C;
        hms[i] = hms[i] / (c < NJ ? 255.f : 127.f);
    }
}

// -------------------------------------------------------------------- NMS --
// One workgroup (NT threads = NT/64 waves) per (frame, keypoint channel).
// Pass 1: strict 3x3 local-max mask, one wave-ballot per 64 consecutive pixels
//         (coalesced: thread t of chunk i owns pixel i*NT+t), ballots kept in LDS.
// Pass 2: wave 0 turns the per-(chunk,wave) popcounts into exclusive raster-order
//         offsets -- this is the reference's global exclusive scan, restricted to
//         the channel (scan[p] - scan[channel start]).
// Pass 3: every peak finds its rank = offset + popcount(lower lanes) and, if
//         rank < 127, writes the 7x7 score-weighted centroid.
constexpr int NMS_NT = 1024;
constexpr int NMS_NW = NMS_NT / 64;
constexpr int NMS_MAXCHUNK = 32;

__device__ __forceinline__ bool nms_is_peak(const float* __restrict__ s, int p, int H, int W, float thr)
{
    const int x = p % W, y = p / W;
    if (!(0 < x && x < W - 1 && 0 < y && y < H - 1)) return false;
    const float v = s[p];
    if (!(v > thr)) return false;
    return v > s[p - W - 1] && v > s[p - W] && v > s[p - W + 1] && v > s[p - 1] && v > s[p + 1] &&
           v > s[p + W - 1] && v > s[p + W] && v > s[p + W + 1];
}

// Pass 1 alone, one thread per pixel over the WHOLE batch (smap_nms_ws): the single-kernel form below runs B x 15 workgroups -- 120 on 256
// CUs at eight frames -- through 26 serial chunks of nine loads per pixel each (109 us per launch, the longest kernel of the association);
// here the masks of a launch are B x 15 x H x W / 256 workgroups (12 480) of one chunk each.  ballots[(b * 15 + c) * nq + q] = the wave
// ballot of pixels 64 q .. 64 q + 63 of channel c: the same predicate on the same pixels as pass 1 below, hence the same peaks.
__global__ __launch_bounds__(256) void nms_mask_kernel(const float* __restrict__ hms, int C, int H, int W, float thr, int nq,
                                                       unsigned long long* __restrict__ ballots)
{
    const int c = blockIdx.y, b = blockIdx.z;
    const int HW = H * W;
    const float* s = hms + ((size_t)b * C + c) * HW;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool f = p < HW && nms_is_peak(s, p, H, W, thr);
    const unsigned long long m = __ballot(f);
    if ((threadIdx.x & 63) == 0 && (p >> 6) < nq) ballots[((size_t)b * NJ + c) * nq + (p >> 6)] = m;
}

// PRE = true: passes 2 and 3 only, the ballots of pass 1 come from nms_mask_kernel's workspace.
template <bool PRE>
__global__ __launch_bounds__(NMS_NT) void nms_kernel(const float* __restrict__ hms, int C, int H, int W,
                                                      float thr, float* __restrict__ peaks,
                                                      const unsigned long long* __restrict__ ballots)
{
    __shared__ unsigned long long s_ballot[NMS_MAXCHUNK * NMS_NW];
    __shared__ int s_off[NMS_MAXCHUNK * NMS_NW + 1];
    const int c = blockIdx.x % NJ, b = blockIdx.x / NJ;
    const int HW = H * W;
    const float* s = hms + ((size_t)b * C + c) * HW;
    float* out = peaks + ((size_t)b * NJ + c) * PSTRIDE * 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunk = (HW + NMS_NT - 1) / NMS_NT;

    if (PRE) {          // chunk i, wave w of the single-kernel form = pixels 64 (i * 16 + w) .. +63 = ballot i * 16 + w
        const int nq_in = (HW + 63) / 64;
        for (int q = tid; q < nchunk * NMS_NW; q += NMS_NT) s_ballot[q] = q < nq_in ? ballots[((size_t)b * NJ + c) * nq_in + q] : 0ull;
    } else {
        for (int i = 0; i < nchunk; ++i) {
            const int p = i * NMS_NT + tid;
            const bool f = p < HW && nms_is_peak(s, p, H, W, thr);
            const unsigned long long m = __ballot(f);
            if (lane == 0) s_ballot[i * NMS_NW + wave] = m;
        }
    }
    __syncthreads();
    const int nq = nchunk * NMS_NW;
    if (wave == 0) {
        const int per = (nq + 63) / 64;
        int local = 0;
        for (int k = 0; k < per; ++k) {
            const int q = lane * per + k;
            if (q < nq) local += __popcll(s_ballot[q]);
        }
        int incl = local;
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        int run = incl - local;
        for (int k = 0; k < per; ++k) {
            const int q = lane * per + k;
            if (q < nq) {
                s_off[q] = run;
                run += __popcll(s_ballot[q]);
            }
        }
        if (lane == 63) s_off[nq] = incl;   // total peaks of the channel
    }
    __syncthreads();
    const int total = s_off[nq];
    const int count = total < MAXP ? total : MAXP;
    if (tid < PSTRIDE) {
        if (tid == 0) { out[0] = (float)count; out[1] = 0.f; out[2] = 0.f; }
        else if (tid > count) { out[3 * tid] = 0.f; out[3 * tid + 1] = 0.f; out[3 * tid + 2] = 0.f; }
    }
    // Pass 3 (round 6): first the pixel of every kept peak by RANK (a thread still owns pixel i * NT + tid of chunk i), then ONE lane per peak:
    // its window values are requested a row of seven at a time and accumulated in the reference's raster order (nmsBase.cu:96-117: same fp32 additions, same
    // order, bit for bit).  Round 5 computed a centroid inside the chunk loop, in the thread that owned the pixel: a wave that met peaks in three
    // of its 26 chunks walked three 49-load dependency chains one after the other -- 77 us per launch for ~24 peaks per channel.
    __shared__ int s_pos[MAXP];
    for (int i = 0; i < nchunk; ++i) {
        const unsigned long long m = s_ballot[i * NMS_NW + wave];
        if (!((m >> lane) & 1ull)) continue;
        const int rank = s_off[i * NMS_NW + wave] + __popcll(m & ((1ull << lane) - 1ull));
        if (rank < MAXP) s_pos[rank] = i * NMS_NT + tid;
    }
    __syncthreads();
    if (tid < count) {
        const int p = s_pos[tid];
        const int px = p % W, py = p / W;
        float xAcc = 0.f, yAcc = 0.f, sAcc = 0.f;
#pragma unroll 1
        for (int dy = -3; dy <= 3; ++dy) {      // a window ROW at a time: seven loads in flight (all 49 at once cost 110 registers -- a 1024-thread
            const int y = py + dy;              // workgroup of those does not find a CU next to two running backbones: 191 us in situ)
            if (!(0 <= y && y < H)) continue;
            float win[7];
#pragma unroll
            for (int dx = -3; dx <= 3; ++dx) {
                const int x = px + dx;
                win[dx + 3] = (0 <= x && x < W) ? s[y * W + x] : 0.f;          // (outside the map: skipped below like a non-positive value)
            }
#pragma unroll
            for (int dx = -3; dx <= 3; ++dx) {
                const int x = px + dx;
                const float sc = win[dx + 3];
                if (0 <= x && x < W && sc > 0) {
                    xAcc += (float)x * sc;
                    yAcc += (float)y * sc;
                    sAcc += sc;
                }
            }
        }
        const int oi = (tid + 1) * 3;
        out[oi] = xAcc / sAcc + 0.5f;
        out[oi + 1] = yAcc / sAcc + 0.5f;
        out[oi + 2] = s[p];
    }
}

// -------------------------------------------------------------- PAF score --
__device__ __forceinline__ int int_round(float a) { return (int)(a + 0.5f); }

__device__ float paf_process(const float* __restrict__ A, const float* __restrict__ Bp,
                             const float* __restrict__ mapX, const float* __restrict__ mapY,
                             int W, int H)
{
    const float interThreshold = 0.05f, interMinAbove = 0.95f, defaultNms = 0.1f;
    const float dx = Bp[0] - A[0];
    const float dy = Bp[1] - A[1];
    const float dmax = fmaxf(fabsf(dx), fabsf(dy));
    int n = int_round(sqrtf(5 * dmax));
    n = n < 25 ? n : 25;
    n = n > 5 ? n : 5;
    const float norm = sqrtf(dx * dx + dy * dy);
    if ((double)norm > 1e-6) {
        const float sX = A[0], sY = A[1];
        const float ux = dx / norm, uy = dy / norm;
        float sum = 0.f;
        int count = 0;
        const float lx = dx / (float)n, ly = dy / (float)n;
        for (int lm = 0; lm < n; ++lm) {
            int mX = int_round(sX + (float)lm * lx);
            int mY = int_round(sY + (float)lm * ly);
            mX = mX < W - 1 ? mX : W - 1;
            mY = mY < H - 1 ? mY : H - 1;
            const int idx = mY * W + mX;
            const float score = ux * mapX[idx] + uy * mapY[idx];
            if (score > interThreshold) {
                sum += score;
                count++;
            }
        }
        if ((float)count / (float)n > interMinAbove) return sum / (float)count;
        const float l2 = sqrtf(dx * dx + dy * dy);
        const float th = sqrtf((float)(W * H)) / 150;
        if (l2 < th) return (float)((double)defaultNms + 1e-6);
    }
    return -1.f;
}

// grid (127 peakA rows, 14 limbs, B frames), 128 threads: lane = peakB.  Rows with
// no peak A only stream out the -1 fill (coalesced 508 B row).
__global__ __launch_bounds__(128) void paf_score_kernel(const float* __restrict__ hms,
                                                        const float* __restrict__ peaks, int H, int W,
                                                        float* __restrict__ scores)
{
    const int pb = threadIdx.x, pa = blockIdx.x, l = blockIdx.y, b = blockIdx.z;
    if (pb >= MAXP) return;
    const int HW = H * W;
    const float* pk = peaks + (size_t)b * NJ * PSTRIDE * 3;
    const int ja = c_pairs[2 * l], jb = c_pairs[2 * l + 1];
    const float nA = pk[3 * ja * PSTRIDE], nB = pk[3 * jb * PSTRIDE];
    float v = -1.f;
    if ((float)pa < nA && (float)pb < nB) {
        const float* h = hms + (size_t)b * SMAP_HMS_C * HW;
        v = paf_process(pk + 3 * (ja * PSTRIDE + pa + 1), pk + 3 * (jb * PSTRIDE + pb + 1),
                        h + (size_t)(NJ + 2 * l) * HW, h + (size_t)(NJ + 2 * l + 1) * HW, W, H);
    }
    scores[(((size_t)b * NL + l) * MAXP + pa) * MAXP + pb] = v;
}

// ------------------------------------------------------------------ group --
// One wave64 per frame.  Limbs and persons are inherently sequential (depth-ordered
// greedy); the <=127 destination candidates of a (limb, person) step are spread two
// per lane and reduced with 6 xor-shuffles: max score, lowest index on ties == the
// reference's "first strict maximum" scan order.
struct KV { float v; int i; };
struct GroupLds {
    float body[MAXP][NJ][4];
    int remap[NJ][PSTRIDE];
    KV kv[PSTRIDE];            // (root depth, peak index), sorted in place
    float sdepth[PSTRIDE];
    int sidx[PSTRIDE];
};

// Person order = torch's CPU Tensor::sort(0, false) (association.cpp:144): std::sort over
// (value, index) pairs with NaN last -- NOT stable, so equal depths come out in libstdc++
// introsort order.  Same algorithm as bits/stl_algo.h (median-of-3 to first, unguarded
// partition, threshold 16, heap-sort fallback, final insertion sort), run by one lane on LDS;
// the recursion on the right part is an explicit stack (sub-ranges are disjoint, so the
// processing order does not change the result).
__device__ __forceinline__ bool kv_lt(const KV a, const KV b) { return (!(a.v != a.v) && (b.v != b.v)) || (a.v < b.v); }
__device__ __forceinline__ void kv_swap(KV* a, KV* b) { const KV t = *a; *a = *b; *b = t; }

__device__ void kv_adjust_heap(KV* f, int hole, int len, const KV val)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (kv_lt(f[child], f[child - 1])) child--;
        f[hole] = f[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        f[hole] = f[child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && kv_lt(f[parent], val)) { f[hole] = f[parent]; hole = parent; parent = (hole - 1) / 2; }
    f[hole] = val;
}

__device__ void kv_heap_sort(KV* f, int len)
{
    if (len >= 2)
        for (int parent = (len - 2) / 2;; --parent) {
            kv_adjust_heap(f, parent, len, f[parent]);
            if (parent == 0) break;
        }
    while (len > 1) {
        --len;
        const KV val = f[len];
        f[len] = f[0];
        kv_adjust_heap(f, 0, len, val);
    }
}

__device__ __forceinline__ void kv_unguarded_linear_insert(KV* a, int last)
{
    const KV val = a[last];
    int next = last - 1;
    while (kv_lt(val, a[next])) { a[last] = a[next]; last = next; --next; }
    a[last] = val;
}

__device__ void kv_insertion_sort(KV* a, int first, int last)
{
    for (int i = first + 1; i < last; ++i) {
        if (kv_lt(a[i], a[first])) {
            const KV val = a[i];
            for (int k = i; k > first; --k) a[k] = a[k - 1];
            a[first] = val;
        } else kv_unguarded_linear_insert(a, i);
    }
}

__device__ void kv_std_sort(KV* a, int n)
{
    if (n <= 0) return;
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) ++lg;
    int stk_f[32], stk_l[32], stk_d[32], sp = 0;
    stk_f[0] = 0; stk_l[0] = n; stk_d[0] = 2 * lg; sp = 1;
    while (sp > 0) {
        --sp;
        int first = stk_f[sp], last = stk_l[sp], depth = stk_d[sp];
        while (last - first > 16) {
            if (depth == 0) { kv_heap_sort(a + first, last - first); break; }
            --depth;
            const int mid = first + (last - first) / 2;
            KV *pa = a + first + 1, *pb = a + mid, *pc = a + last - 1, *pf = a + first;
            if (kv_lt(*pa, *pb)) {
                if (kv_lt(*pb, *pc)) kv_swap(pf, pb);
                else if (kv_lt(*pa, *pc)) kv_swap(pf, pc);
                else kv_swap(pf, pa);
            } else if (kv_lt(*pa, *pc)) kv_swap(pf, pa);
            else if (kv_lt(*pb, *pc)) kv_swap(pf, pc);
            else kv_swap(pf, pb);
            int lo = first + 1, hi = last;
            for (;;) {
                while (kv_lt(a[lo], a[first])) ++lo;
                --hi;
                while (kv_lt(a[first], a[hi])) --hi;
                if (!(lo < hi)) break;
                kv_swap(a + lo, a + hi);
                ++lo;
            }
            if (sp < 32) { stk_f[sp] = lo; stk_l[sp] = last; stk_d[sp] = depth; ++sp; }   // right part later
            last = lo;
        }
    }
    if (n > 16) {
        kv_insertion_sort(a, 0, 16);
        for (int i = 16; i < n; ++i) kv_unguarded_linear_insert(a, i);
    } else kv_insertion_sort(a, 0, n);
}

// One limb of the greedy grouping (association.cpp:171-229), run by ONE wave64: persons in depth order (sequential: the
// ordinal prior), the <= 127 destination candidates two per lane, arg-max by xor-shuffles with the lowest index on ties
// (== the reference's "first strict maximum" scan).  Reads body[.][src] / remap[src][.] (complete when the limb starts),
// writes body[.][dst] / remap[dst][.]: every joint is the destination of exactly one limb.
__device__ __forceinline__ void group_limb(GroupLds& L, const int i, const int P, const int root_idx, const int dist_flag,
                                           const float* __restrict__ peaks, const float* __restrict__ scores, const int lane)
{
    const float dsScale = 4;
    int src, dst;
    bool flip = false;
    if (root_idx == 2 && i == 1) { src = c_pairs[2 * i + 1]; dst = c_pairs[2 * i]; flip = true; }
    else { src = c_pairs[2 * i]; dst = c_pairs[2 * i + 1]; }
    const float* dp = peaks + 3 * dst * PSTRIDE;
    const int nDst = (int)dp[0];
    if (nDst == 0) return;
    const float* S = scores + (size_t)i * MAXP * MAXP;
    // this lane's two candidates
    const int c0 = lane, c1 = lane + 64;
    const bool v0 = c0 < nDst, v1 = c1 < nDst;
    const float d0x = v0 ? dp[3 * (c0 + 1)] : 0.f, d0y = v0 ? dp[3 * (c0 + 1) + 1] : 0.f;
    const float d1x = v1 ? dp[3 * (c1 + 1)] : 0.f, d1y = v1 ? dp[3 * (c1 + 1) + 1] : 0.f;
    bool used0 = false, used1 = false;
    for (int k1 = 0; k1 < P; ++k1) {
        const float sscore = L.body[k1][src][3];
        if ((double)sscore < 1e-5) continue;        // wave-uniform
        const float sx = L.body[k1][src][0], sy = L.body[k1][src][1];
        const float bone_dist = (float)(1.2 * (double)c_bone_length[i] / (double)L.sdepth[k1]);
        const int rs = L.remap[src][k1];
        float best = 0.0f;
        int bidx = 0x7fffffff;
        if (v0 && !used0) {
            float score = flip ? S[c0 * MAXP + rs] : S[rs * MAXP + c0];
            if (dist_flag && score > 0) {
                const float ddx = sx - d0x, ddy = sy - d0y;
                const float limb = (float)sqrt((double)ddx * (double)ddx + (double)ddy * (double)ddy);
                const float pen = bone_dist / limb / dsScale - 1;
                score += pen < 0.0f ? pen : 0.0f;
            }
            if (score > best) { best = score; bidx = c0; }
        }
        if (v1 && !used1) {
            float score = flip ? S[c1 * MAXP + rs] : S[rs * MAXP + c1];
            if (dist_flag && score > 0) {
                const float ddx = sx - d1x, ddy = sy - d1y;
                const float limb = (float)sqrt((double)ddx * (double)ddx + (double)ddy * (double)ddy);
                const float pen = bone_dist / limb / dsScale - 1;
                score += pen < 0.0f ? pen : 0.0f;
            }
            if (score > best) { best = score; bidx = c1; }   // c1 > c0: strict > keeps the lower index
        }
        for (int d = 32; d >= 1; d >>= 1) {
            const float os = __shfl_xor(best, d);
            const int oi = __shfl_xor(bidx, d);
            if (os > best || (os == best && oi < bidx)) { best = os; bidx = oi; }
        }
        if (best > 0) {                              // wave-uniform after the butterfly
            if (lane == 0) {
                L.body[k1][dst][0] = dp[3 * (bidx + 1)];
                L.body[k1][dst][1] = dp[3 * (bidx + 1) + 1];
                L.body[k1][dst][3] = dp[3 * (bidx + 1) + 2];
                L.remap[dst][k1] = bidx;
            }
            if (bidx == c0) used0 = true;
            if (bidx == c1) used1 = true;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // lane 0's LDS writes before the next limb of this wave reads them
}

// One WORKGROUP per frame, one wave per chain of the limb tree.  The reference walks the 14 limbs in the order
// 1, 0, 2, 3, .. 13 (association.cpp:164-170); a limb only reads the joint its source end was assigned by an EARLIER limb
// (or the root), so with root_idx = 2 (pelvis) the order is a tree:
//     pelvis -1-> neck -0-> head           wave 0: limb 1, then limb 0
//                 neck -2-> 9 -3-> 10 -4-> 11     wave 1 (after limb 1)
//                 neck -5-> 3 -6-> 4 -7-> 5       wave 2 (after limb 1)
//     pelvis -8-> 12 -9-> 13 -10-> 14             wave 3 (at once)
//     pelvis -11-> 6 -12-> 7 -13-> 8              wave 4 (at once)
// Inside a limb nothing changes (persons sequential, lowest candidate index on ties); different limbs write different
// joints; so the result is the reference's, bit for bit, in 4 limb times instead of 14.  Any other root_idx runs the
// reference's flat order on wave 0.
constexpr int GROUP_WAVES = 5;
__global__ __launch_bounds__(GROUP_WAVES * 64) void group_kernel(const float* __restrict__ peaks_all,
                                                   const float* __restrict__ scores_all,
                                                   const float* __restrict__ rdepth_all, int H, int W,
                                                   int root_idx, int dist_flag,
                                                   float* __restrict__ bodys_all, int* __restrict__ counts)
{
    __shared__ GroupLds L;
    __shared__ int neck_done;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NT = GROUP_WAVES * 64;
    const float* peaks = peaks_all + (size_t)b * NJ * PSTRIDE * 3;
    const float* scores = scores_all + (size_t)b * NL * MAXP * MAXP;
    const float* rdepth = rdepth_all + (size_t)b * H * W;
    float* bodys = bodys_all + (size_t)b * MAXP * NJ * 4;

    const float* rp = peaks + 3 * root_idx * PSTRIDE;
    const int P = (int)rp[0];
    if (tid == 0) { counts[b] = P; neck_done = 0; }
    float* lb = &L.body[0][0][0];
    for (int i = tid; i < MAXP * NJ * 4; i += NT) lb[i] = 0.f;
    if (P == 0) {
        for (int i = tid; i < MAXP * NJ * 4; i += NT) bodys[i] = 0.f;
        return;
    }
    for (int i = tid; i < P; i += NT) {
        L.kv[i].v = rdepth[(int)rp[3 * (i + 1) + 1] * W + (int)rp[3 * (i + 1)]];
        L.kv[i].i = i;
    }
    __syncthreads();
    if (tid == 0) kv_std_sort(L.kv, P);            // ordinal prior: near persons first
    __syncthreads();
    for (int i = tid; i < P; i += NT) {
        L.sidx[i] = L.kv[i].i;
        L.sdepth[i] = L.kv[i].v;
    }
    __syncthreads();
    for (int i = tid; i < NJ * PSTRIDE; i += NT) {
        const int j = i / PSTRIDE, k = i % PSTRIDE;
        if (k < P) L.remap[j][k] = (j == root_idx) ? L.sidx[k] : k;
    }
    for (int i = tid; i < P; i += NT) {
        const int s = L.sidx[i];
        L.body[i][root_idx][0] = rp[3 * (s + 1)];
        L.body[i][root_idx][1] = rp[3 * (s + 1) + 1];
        L.body[i][root_idx][3] = rp[3 * (s + 1) + 2];
    }
    __syncthreads();

    if (root_idx != 2) {                              // no tree known: the reference's flat order on one wave
        if (wave == 0)
            for (int j = 0; j < NL; ++j) group_limb(L, (j == 0) ? 1 : (j == 1) ? 0 : j, P, root_idx, dist_flag, peaks, scores, lane);
    } else if (wave == 0) {
        group_limb(L, 1, P, root_idx, dist_flag, peaks, scores, lane);
        if (lane == 0) __hip_atomic_store(&neck_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        group_limb(L, 0, P, root_idx, dist_flag, peaks, scores, lane);
    } else {
        const int first = wave == 1 ? 2 : wave == 2 ? 5 : wave == 3 ? 8 : 11;
        if (wave <= 2)                                // the neck chains wait for limb 1 (wave 0 runs on its own: no deadlock)
            while (__hip_atomic_load(&neck_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(4);
        for (int i = first; i < first + 3; ++i) group_limb(L, i, P, root_idx, dist_flag, peaks, scores, lane);
    }
    __syncthreads();
    for (int i = tid; i < MAXP * NJ * 4; i += NT) bodys[i] = lb[i];
}

// ------------------------------------------------------------------- lift --
// One thread per person (<=127 per frame), one workgroup per frame.
__device__ __forceinline__ double np_lerp(double a, double b, double t)
{
    const double d = (double)((float)b - (float)a);
    return t >= 0.5 ? b - d * (1 - t) : a + d * t;
}

// GT = true: the ground-truth modes of test.py (generate_result / generate_train), where register_pred hands
// back a FLOAT64 person array (test_util.py:37): the Z column, the un-letter-boxed x / y and the absolute depth
// are never rounded to fp32, pred_2d is f64, and the root depth is an all-fp32 product (np.float32 map value x
// Python float scale x np.float32 focal length under numpy 2's weak-scalar promotion).
template <bool GT>
__global__ __launch_bounds__(128) void lift_kernel(const float* __restrict__ bodys_all,
                                                   const int* __restrict__ counts,
                                                   const float* __restrict__ det_d_all,
                                                   const float* __restrict__ root_d_all,
                                                   const double* __restrict__ cams, int H, int W,
                                                   typename std::conditional<GT, double, float>::type* __restrict__ pred_2d_all,
                                                   double* __restrict__ pred_3d_all,
                                                   double* __restrict__ root_z_all)
{
    typedef typename std::conditional<GT, double, float>::type T2;
    constexpr int STRIDE = 4, NPTS = 10, root_n = 2;
    const int b = blockIdx.x, i = threadIdx.x;
    const int P = counts[b];
    T2* p2 = pred_2d_all + ((size_t)b * MAXP + i) * NJ * 4;
    double* o = pred_3d_all + ((size_t)b * MAXP + i) * NJ * 4;
    if (i >= MAXP) return;
    if (i >= P) {
        for (int k = 0; k < NJ * 4; ++k) { p2[k] = (T2)0; o[k] = 0.0; }
        root_z_all[(size_t)b * MAXP + i] = 0.0;
        return;
    }
    const float* src = bodys_all + ((size_t)b * MAXP + i) * NJ * 4;
    const float* det_d = det_d_all + (size_t)b * NL * H * W;
    const float* root_d = root_d_all + (size_t)b * H * W;
    const double* cam = cams + (size_t)b * 9;
    const double scale = cam[0], img_w = cam[1], img_h = cam[2], net_w = cam[3], net_h = cam[4],
                 fx = cam[5], fy = cam[6], cx = cam[7], cy = cam[8];
    float bx[NJ], by[NJ], bs[NJ];
    T2 bz[NJ];                                   // fp32 storage in run_inference, f64 in the ground-truth modes
    for (int j = 0; j < NJ; ++j) {
        bx[j] = src[4 * j] * (float)STRIDE;
        by[j] = src[4 * j + 1] * (float)STRIDE;
        bz[j] = src[4 * j + 2];
        bs[j] = src[4 * j + 3];
    }
    double depth_v[NL];
    for (int k = 0; k < NL; ++k) depth_v[k] = 0.0;
    double rz = 0.0;
    if (bs[root_n] > 0) {
        const int ry = (int)by[root_n], rx = (int)bx[root_n];
        if (GT) rz = (double)((root_d[(ry / STRIDE) * W + rx / STRIDE] * (float)scale) * (float)fx);
        else rz = (double)root_d[(ry / STRIDE) * W + rx / STRIDE] * scale * fx;
        for (int k = 0; k < NL; ++k) {
            const int js = c_pairs[2 * k], jd = c_pairs[2 * k + 1];
            if (!(bs[jd] > 0 && bs[js] > 0)) continue;
            float v[NPTS], sv[NPTS];
            const double dxx = (double)bx[jd] - (double)bx[js], dyy = (double)by[jd] - (double)by[js];
            const double stepx = dxx / (NPTS - 1), stepy = dyy / (NPTS - 1);
            for (int t = 0; t < NPTS; ++t) {
                double px = (double)t * stepx + (double)bx[js];
                double py = (double)t * stepy + (double)by[js];
                if (stepx == 0) px = ((double)t / (NPTS - 1)) * dxx + (double)bx[js];
                if (stepy == 0) py = ((double)t / (NPTS - 1)) * dyy + (double)by[js];
                if (t == NPTS - 1) { px = bx[jd]; py = by[jd]; }
                const long ix = (long)rint(px), iy = (long)rint(py);
                v[t] = det_d[((size_t)k * H + iy / STRIDE) * W + ix / STRIDE];
            }
            for (int t = 0; t < NPTS; ++t) {           // insertion sort
                const float x = v[t];
                int u = t;
                while (u > 0 && sv[u - 1] > x) { sv[u] = sv[u - 1]; --u; }
                sv[u] = x;
            }
            const double vlo = 10.0 / 100.0 * (NPTS - 1), vhi = 90.0 / 100.0 * (NPTS - 1);
            const int ilo = (int)floor(vlo), ihi = (int)floor(vhi);
            const double mn = np_lerp(sv[ilo], sv[ilo + 1], vlo - ilo);
            const double mx = np_lerp(sv[ihi], sv[ihi + 1 < NPTS ? ihi + 1 : ihi], vhi - ihi);
            for (int t = 0; t < NPTS; ++t) {
                if ((double)v[t] < mn) v[t] = (float)mn;
                if ((double)v[t] > mx) v[t] = (float)mx;
            }
            float r = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            r += v[8];
            r += v[9];
            depth_v[k] = (double)(r / (float)NPTS);
        }
        bz[2] = (T2)0;
        bz[0] = (T2)((double)bz[2] - depth_v[1]);
        bz[1] = (T2)((double)bz[0] + depth_v[0]);
        for (int k = 2; k < NL; ++k)
            bz[c_pairs[2 * k + 1]] = (T2)((double)bz[c_pairs[2 * k]] + depth_v[k]);
    }
    root_z_all[(size_t)b * MAXP + i] = rz;
    const bool live = bs[root_n] != 0;
    for (int j = 0; j < NJ; ++j) {
        p2[4 * j] = bx[j]; p2[4 * j + 1] = by[j]; p2[4 * j + 2] = bz[j]; p2[4 * j + 3] = bs[j];
        const T2 x = (T2)((double)bx[j] / scale - (net_w / scale - img_w) / 2);
        const T2 y = (T2)((double)by[j] / scale - (net_h / scale - img_h) / 2);
        double X = 0.0, Y = 0.0, Z = 0.0, Sc = bs[j];
        if (live) {
            const T2 z = (T2)((double)bz[j] + rz);
            X = ((double)x - cx) * (double)z / fx;
            Y = ((double)y - cy) * (double)z / fy;
            Z = z;
        }
        if (Sc == 0) { X = Y = Z = 0.0; }
        o[4 * j] = X; o[4 * j + 1] = Y; o[4 * j + 2] = Z; o[4 * j + 3] = Sc;
    }
}

// ------------------------------------------------------- GT registration --
// register_pred with ground truth (test_util.py:18-42) for the generate_result / generate_train modes: one wave64
// per frame.  dist[g][p] = |gt root g - predicted root p| in fp32 (sqrt(dx*dx + dy*dy), np.linalg.norm(axis=2));
// the greedy loop retires the smallest distance < 30 px -- ties in row-major (g, p) order -- and accepts the pair
// when neither side is taken.  matched[g] = the accepted prediction (heat-map pixels, as connect wrote it) or
// zeros; matched_counts = number of annotations, or 0 when the frame has no prediction or no annotation (the
// caller skips such frames, test.py:81-82 / test_util.py:19-20).
constexpr int REG_MAXG = 64;
__global__ __launch_bounds__(64) void register_gt_kernel(const float* __restrict__ bodys_all,
                                                         const int* __restrict__ counts,
                                                         const float* __restrict__ gt_roots_all,
                                                         const int* __restrict__ gt_counts, int Gmax,
                                                         float* __restrict__ matched_all,
                                                         int* __restrict__ matched_counts)
{
    constexpr int root_n = 2;
    __shared__ float dist[REG_MAXG * MAXP];
    __shared__ int corres[REG_MAXG];
    __shared__ int occupied[MAXP];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int P = counts[b];
    int G = gt_counts[b];
    G = G < 0 ? 0 : (G > Gmax ? Gmax : G);
    const float* bodys = bodys_all + (size_t)b * MAXP * NJ * 4;
    const float* gt = gt_roots_all + (size_t)b * Gmax * 2;
    float* out = matched_all + (size_t)b * MAXP * NJ * 4;
    for (int k = lane; k < MAXP * NJ * 4; k += 64) out[k] = 0.f;
    if (lane == 0) matched_counts[b] = (P > 0 && G > 0) ? G : 0;
    if (P <= 0 || G <= 0) return;
    const int n = G * P;
    for (int k = lane; k < n; k += 64) {
        const int g = k / P, p = k - g * P;
        const float px = bodys[(p * NJ + root_n) * 4] * 4.0f, py = bodys[(p * NJ + root_n) * 4 + 1] * 4.0f;
        const float dx = gt[2 * g] - px, dy = gt[2 * g + 1] - py;
        dist[k] = sqrtf(dx * dx + dy * dy);
    }
    for (int k = lane; k < G; k += 64) corres[k] = -1;
    for (int k = lane; k < P; k += 64) occupied[k] = 0;
    __syncthreads();
    for (;;) {
        float bv = INFINITY;
        int bi = 0x7fffffff;
        for (int k = lane; k < n; k += 64) {
            const float v = dist[k];
            if (v < bv) { bv = v; bi = k; }                 // strided scan: the first hit per lane is its lowest index
        }
        for (int d = 32; d >= 1; d >>= 1) {
            const float ov = __shfl_xor(bv, d);
            const int oi = __shfl_xor(bi, d);
            if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (!(bv < 30.0f)) break;
        const int g = bi / P, p = bi - g * P;
        if (lane == 0) {
            dist[bi] = 50.0f;
            if (corres[g] < 0 && !occupied[p]) { corres[g] = p; occupied[p] = 1; }
        }
        __syncthreads();
    }
    __syncthreads();
    for (int k = lane; k < G * NJ * 4; k += 64) {
        const int g = k / (NJ * 4), e = k - g * (NJ * 4);
        const int p = corres[g];
        if (p >= 0) out[k] = bodys[p * NJ * 4 + e];
    }
}

// ----------------------------------------------------------------- refine --
// One workgroup per (frame, person); the 5 layers run back to back with the
// activations in LDS and the folded, transposed weights [in][out] streamed from L2
// (thread o reads Wt[k][o]: coalesced across the wave).
struct RefineW { const float* wt[5]; const float* bs[5]; };

// T2 = float: run_inference (fp32 2D offsets); T2 = double: ground-truth modes (f64 offsets, rounded once by `.float()`)
template <typename T2>
__global__ __launch_bounds__(256) void refine_kernel(const T2* __restrict__ pred_2d_all,
                                                     const double* __restrict__ pred_3d_all,
                                                     const int* __restrict__ counts, RefineW w,
                                                     double* __restrict__ refined_all)
{
    const int dims[6] = {75, 160, 256, 256, 128, 45};
    constexpr int root_n = 2;
    __shared__ float act[2][256];
    const int b = blockIdx.y, i = blockIdx.x, t = threadIdx.x;
    double* out = refined_all + ((size_t)b * MAXP + i) * NJ * 4;
    if (i >= counts[b]) {
        if (t < NJ * 4) out[t] = 0.0;
        return;
    }
    const T2* p2 = pred_2d_all + ((size_t)b * MAXP + i) * NJ * 4;
    const double* p3 = pred_3d_all + ((size_t)b * MAXP + i) * NJ * 4;
    if (t < 75) {
        const int j = t / 5, d = t % 5;
        double v = 0.0;
        if (j == root_n) v = d < 2 ? (double)p2[4 * j + d] : p3[4 * j + d - 2];
        else if (p3[4 * j + 3] > 0)
            v = d < 2 ? (double)(T2)(p2[4 * j + d] - p2[4 * root_n + d]) : p3[4 * j + d - 2] - p3[4 * root_n + d - 2];
        act[0][t] = (float)v;
    }
    __syncthreads();
    int cur = 0;
    for (int l = 0; l < 5; ++l) {
        const int nin = dims[l], nout = dims[l + 1];
        if (t < nout) {
            float acc = 0.f;
            for (int k = 0; k < nin; ++k) acc += w.wt[l][(size_t)k * nout + t] * act[cur][k];
            acc += w.bs[l][t];
            act[cur ^ 1][t] = (l < 4 && acc < 0.f) ? 0.f : acc;
        }
        __syncthreads();
        cur ^= 1;
    }
    if (t < NJ * 4) {
        const int j = t / 4, d = t % 4;
        double v;
        if (d == 3) v = (p3[4 * root_n + 3] == 0) ? 0.0 : 1.0;
        else if (j != root_n) v = (double)(float)((double)act[cur][j * 3 + d] + p3[4 * root_n + d]);
        else v = (double)(float)p3[4 * j + d];
        out[t] = v;
    }
}

// Plain RefineNet forward (refinenet.py:34-37): x [N,75] -> y [N,45], one workgroup per row.
__global__ __launch_bounds__(256) void refine_mlp_kernel(const float* __restrict__ x, RefineW w,
                                                         float* __restrict__ y)
{
    const int dims[6] = {75, 160, 256, 256, 128, 45};
    __shared__ float act[2][256];
    const int i = blockIdx.x, t = threadIdx.x;
    if (t < 75) act[0][t] = x[(size_t)i * 75 + t];
    __syncthreads();
    int cur = 0;
    for (int l = 0; l < 5; ++l) {
        const int nin = dims[l], nout = dims[l + 1];
        if (t < nout) {
            float acc = 0.f;
            for (int k = 0; k < nin; ++k) acc += w.wt[l][(size_t)k * nout + t] * act[cur][k];
            acc += w.bs[l][t];
            act[cur ^ 1][t] = (l < 4 && acc < 0.f) ? 0.f : acc;
        }
        __syncthreads();
        cur ^= 1;
    }
    if (t < 45) y[(size_t)i * 45 + t] = act[cur][t];
}

// ------------------------------------------------------------- flip-TTA --
// test.py:55-70: outputs_2d[:, i] += sign_i * flip_x(outputs_2d_flip)[:, pair[i]] for all 43 channels
// (PAF-x channels negated), then outputs_2d[:, 15:] *= 0.5 -- key-point channels stay a SUM.
struct FlipTab { int pair[SMAP_HMS_C]; };
__global__ void flip_merge_kernel(float* __restrict__ hms, const float* __restrict__ flip, FlipTab tab, int W,
                                  int HW, long long total)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int pix = (int)(i % HW);
        const long long bc = i / HW;
        const int c = (int)(bc % SMAP_HMS_C);
        const long long b = bc / SMAP_HMS_C;
        const int y = pix / W, x = pix - y * W;
        const float f = flip[((b * SMAP_HMS_C + tab.pair[c]) * (long long)HW) + y * W + (W - 1 - x)];
        float v = hms[i];
        const bool neg = c >= NJ && ((c - NJ) & 1) == 0;
        v = v + (neg ? f * -1.f : f);
        if (c >= NJ) v = v * 0.5f;
        hms[i] = v;
    }
}

// ------------------------------------------------------------ preprocess --
// dataset/custom_dataset.py:41-68 (aug_croppad) + ToTensor + Normalize on the device: bilinear
// resize (half-pixel centres, no anti-aliasing -- the sampling rule of cv2.INTER_LINEAR and of
// F.interpolate(align_corners=False)), round to uint8, centre into the net_h x net_w canvas padded
// with 128, /255, (x - mean) / std.  One thread per canvas pixel, 3 channels; fp32 arithmetic in
// ATen's operation order so that it matches the host path bit for bit.
struct PrepArgs { int h, w, nh, nw, top, left, net_h, net_w; float mean[3], stdv[3]; double scale_x, scale_y; };

// One axis of OpenCV's 8-bit INTER_LINEAR (modules/imgproc/src/resize.cpp, cv::resize -> resizeGeneric_ set-up):
//   fx = (float)((d + 0.5) * scale - 0.5) with scale = 1 / fx_arg (double); s = floor(fx); fx -= s;
//   s < 0 -> (0, 0);  s >= n - 1 -> (n - 1, 0);  coefficients = cvRound(c * 2048) as shorts (INTER_RESIZE_COEF_BITS = 11).
struct CvTap { int s0, s1; int c0, c1; };
__device__ __forceinline__ CvTap cv_tap(int d, double scale, int n)
{
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= n - 1) { f = 0.f; s = n - 1; }
    CvTap t;
    t.s0 = s;
    t.s1 = s + 1 < n ? s + 1 : s;                        // its coefficient is 0 whenever the clamp applies
    t.c0 = (int)rintf((1.f - f) * 2048.f);               // saturate_cast<short>(float) = cvRound: half to even
    t.c1 = (int)rintf(f * 2048.f);
    return t;
}

// dataset/custom_dataset.py:41-68 for one image: cv2.resize(img, (0,0), fx=s, fy=s) [INTER_LINEAR, 8-bit fixed point],
// centre pad with 128, ToTensor, Normalize.  The resize follows OpenCV's published algorithm operation by operation:
// horizontal pass in 11-bit fixed point (int32), vertical pass ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2
// (VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>), and the exact 2x shrink as the 2x2 box mean
// (cv::resize switches INTER_LINEAR to INTER_AREA there).  cv2 is not in this image: parity unpinned by execution.
__global__ void preprocess_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, PrepArgs p)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= p.net_w) return;
    float v[3] = {128.f, 128.f, 128.f};
    const int ry = y - p.top, rx = x - p.left;
    if ((unsigned)ry < (unsigned)p.nh && (unsigned)rx < (unsigned)p.nw) {
        if (p.scale_x == 2.0 && p.scale_y == 2.0) {      // INTER_AREA fast path: (a + b + c + d + 2) >> 2
            const int y0 = ry * 2 < p.h ? ry * 2 : p.h - 1, y1 = ry * 2 + 1 < p.h ? ry * 2 + 1 : p.h - 1;
            const int x0 = rx * 2 < p.w ? rx * 2 : p.w - 1, x1 = rx * 2 + 1 < p.w ? rx * 2 + 1 : p.w - 1;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                v[c] = (float)((src[((size_t)y0 * p.w + x0) * 3 + c] + src[((size_t)y0 * p.w + x1) * 3 + c] +
                                src[((size_t)y1 * p.w + x0) * 3 + c] + src[((size_t)y1 * p.w + x1) * 3 + c] + 2) >> 2);
        } else {
            const CvTap ty = cv_tap(ry, p.scale_y, p.h), tx = cv_tap(rx, p.scale_x, p.w);
            const unsigned char* r0 = src + ((size_t)ty.s0 * p.w) * 3;
            const unsigned char* r1 = src + ((size_t)ty.s1 * p.w) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int h0 = r0[tx.s0 * 3 + c] * tx.c0 + r0[tx.s1 * 3 + c] * tx.c1;
                const int h1 = r1[tx.s0 * 3 + c] * tx.c0 + r1[tx.s1 * 3 + c] * tx.c1;
                const int t = (((ty.c0 * (h0 >> 4)) >> 16) + ((ty.c1 * (h1 >> 4)) >> 16) + 2) >> 2;
                v[c] = (float)(t < 0 ? 0 : (t > 255 ? 255 : t));
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
        dst[((size_t)c * p.net_h + y) * p.net_w + x] = (v[c] / 255.0f - p.mean[c]) / p.stdv[c];
}

}  // namespace

// ---------------------------------------------------------------- C ABI ----
extern "C" {

int smap_scale_hms(float* hms, int B, int H, int W, void* stream)
{
    if (!hms || B <= 0 || H <= 0 || W <= 0) return SMAP_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)B * SMAP_HMS_C * H * W;
    if ((H * W) % 4 == 0 && ((uintptr_t)hms & 15) == 0) {
        const int total4 = (int)(total / 4);
        const int grid = (total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048;
        hipLaunchKernelGGL(scale_hms_kernel, dim3(grid), dim3(256), 0, st, hms, H * W / 4, total4);
    } else {
        const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(scale_hms_kernel_scalar, dim3(grid), dim3(256), 0, st, hms, H * W, (int)total);
    }
    return hip_rc(hipGetLastError());
}

int smap_nms(const float* hms, int B, int C, int H, int W, float threshold, float* peaks, void* stream)
{
    if (!hms || !peaks || B <= 0 || C < NJ || H < 3 || W < 3) return SMAP_E_ARG;
    if ((long long)H * W > (long long)NMS_MAXCHUNK * NMS_NT) return SMAP_E_ARG;
    hipLaunchKernelGGL(nms_kernel<false>, dim3(B * NJ), dim3(NMS_NT), 0, (hipStream_t)stream, hms, C, H, W,
                       threshold, peaks, (const unsigned long long*)nullptr);
    return hip_rc(hipGetLastError());
}

int64_t smap_nms_workspace_bytes(int B, int H, int W)
{
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (int64_t)B * NJ * (((int64_t)H * W + 63) / 64) * 8;
}

int smap_nms_ws(const float* hms, int B, int C, int H, int W, float threshold, float* peaks, void* workspace,
                int64_t workspace_bytes, void* stream)
{
    if (!hms || !peaks || !workspace || B <= 0 || C < NJ || H < 3 || W < 3 || ((uintptr_t)workspace & 7)) return SMAP_E_ARG;
    if ((long long)H * W > (long long)NMS_MAXCHUNK * NMS_NT || B > 65535) return SMAP_E_ARG;
    if (workspace_bytes < smap_nms_workspace_bytes(B, H, W)) return SMAP_E_ARG;
    const int nq = (H * W + 63) / 64;
    unsigned long long* bal = reinterpret_cast<unsigned long long*>(workspace);
    hipLaunchKernelGGL(nms_mask_kernel, dim3((H * W + 255) / 256, NJ, B), dim3(256), 0, (hipStream_t)stream, hms, C, H, W, threshold, nq, bal);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return hip_rc(e);
    hipLaunchKernelGGL(nms_kernel<true>, dim3(B * NJ), dim3(NMS_NT), 0, (hipStream_t)stream, hms, C, H, W,
                       threshold, peaks, (const unsigned long long*)bal);
    return hip_rc(hipGetLastError());
}

int smap_paf_score(const float* hms, const float* peaks, int B, int H, int W, float* scores, void* stream)
{
    if (!hms || !peaks || !scores || B <= 0 || H <= 0 || W <= 0) return SMAP_E_ARG;
    hipLaunchKernelGGL(paf_score_kernel, dim3(MAXP, NL, B), dim3(128), 0, (hipStream_t)stream, hms, peaks,
                       H, W, scores);
    return hip_rc(hipGetLastError());
}

int smap_group(const float* peaks, const float* scores, const float* rdepth, int B, int H, int W,
               int root_idx, int dist_flag, float* bodys, int32_t* counts, void* stream)
{
    if (!peaks || !scores || !rdepth || !bodys || !counts || B <= 0 || root_idx < 0 || root_idx >= NJ)
        return SMAP_E_ARG;
    hipLaunchKernelGGL(group_kernel, dim3(B), dim3(GROUP_WAVES * 64), 0, (hipStream_t)stream, peaks, scores, rdepth, H, W,
                       root_idx, dist_flag, bodys, counts);
    return hip_rc(hipGetLastError());
}

int smap_lift(const float* bodys, const int32_t* counts, const float* det_d, const float* root_d,
              const double* cams, int B, int H, int W, float* pred_2d, double* pred_3d, double* root_z,
              void* stream)
{
    if (!bodys || !counts || !det_d || !root_d || !cams || !pred_2d || !pred_3d || !root_z || B <= 0)
        return SMAP_E_ARG;
    hipLaunchKernelGGL(lift_kernel<false>, dim3(B), dim3(128), 0, (hipStream_t)stream, bodys, counts, det_d, root_d,
                       cams, H, W, pred_2d, pred_3d, root_z);
    return hip_rc(hipGetLastError());
}

int smap_lift_gt(const float* bodys, const int32_t* counts, const float* det_d, const float* root_d,
                 const double* cams, int B, int H, int W, double* pred_2d, double* pred_3d, double* root_z,
                 void* stream)
{
    if (!bodys || !counts || !det_d || !root_d || !cams || !pred_2d || !pred_3d || !root_z || B <= 0)
        return SMAP_E_ARG;
    hipLaunchKernelGGL(lift_kernel<true>, dim3(B), dim3(128), 0, (hipStream_t)stream, bodys, counts, det_d, root_d,
                       cams, H, W, pred_2d, pred_3d, root_z);
    return hip_rc(hipGetLastError());
}

int smap_register_gt(const float* bodys, const int32_t* counts, const float* gt_roots, const int32_t* gt_counts,
                     int B, int G, float* matched, int32_t* matched_counts, void* stream)
{
    if (!bodys || !counts || !gt_roots || !gt_counts || !matched || !matched_counts || B <= 0 || G <= 0 ||
        G > REG_MAXG)
        return SMAP_E_ARG;
    hipLaunchKernelGGL(register_gt_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, bodys, counts, gt_roots,
                       gt_counts, G, matched, matched_counts);
    return hip_rc(hipGetLastError());
}

int smap_refine(const float* pred_2d, const double* pred_3d, const int32_t* counts, int B,
                const float* const* wt, const float* const* bs, double* refined, void* stream)
{
    if (!pred_2d || !pred_3d || !counts || !wt || !bs || !refined || B <= 0) return SMAP_E_ARG;
    RefineW w;
    for (int l = 0; l < 5; ++l) {
        if (!wt[l] || !bs[l]) return SMAP_E_ARG;
        w.wt[l] = wt[l];
        w.bs[l] = bs[l];
    }
    hipLaunchKernelGGL(refine_kernel<float>, dim3(MAXP, B), dim3(256), 0, (hipStream_t)stream, pred_2d, pred_3d,
                       counts, w, refined);
    return hip_rc(hipGetLastError());
}

int smap_refine_gt(const double* pred_2d, const double* pred_3d, const int32_t* counts, int B,
                   const float* const* wt, const float* const* bs, double* refined, void* stream)
{
    if (!pred_2d || !pred_3d || !counts || !wt || !bs || !refined || B <= 0) return SMAP_E_ARG;
    RefineW w;
    for (int l = 0; l < 5; ++l) {
        if (!wt[l] || !bs[l]) return SMAP_E_ARG;
        w.wt[l] = wt[l];
        w.bs[l] = bs[l];
    }
    hipLaunchKernelGGL(refine_kernel<double>, dim3(MAXP, B), dim3(256), 0, (hipStream_t)stream, pred_2d, pred_3d,
                       counts, w, refined);
    return hip_rc(hipGetLastError());
}

}  // extern "C"

extern "C" int smap_refine_mlp(const float* x, int N, const float* const* wt, const float* const* bs, float* y,
                               void* stream)
{
    if (!x || !y || !wt || !bs || N < 0) return SMAP_E_ARG;
    if (N == 0) return 0;
    RefineW w;
    for (int l = 0; l < 5; ++l) {
        if (!wt[l] || !bs[l]) return SMAP_E_ARG;
        w.wt[l] = wt[l];
        w.bs[l] = bs[l];
    }
    hipLaunchKernelGGL(refine_mlp_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, x, w, y);
    return hip_rc(hipGetLastError());
}

extern "C" int smap_flip_merge(float* hms, const float* hms_flip, const int* pair43, int B, int H, int W, void* stream)
{
    if (!hms || !hms_flip || !pair43 || B <= 0 || H <= 0 || W <= 0) return SMAP_E_ARG;
    FlipTab tab;
    for (int c = 0; c < SMAP_HMS_C; ++c) {
        if (pair43[c] < 0 || pair43[c] >= SMAP_HMS_C) return SMAP_E_ARG;
        tab.pair[c] = pair43[c];
    }
    const long long total = (long long)B * SMAP_HMS_C * H * W;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(flip_merge_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, hms, hms_flip, tab, W, H * W,
                       total);
    return hip_rc(hipGetLastError());
}

extern "C" int smap_preprocess(const unsigned char* src, int h, int w, int nh, int nw, int top, int left, float* dst,
                               int net_h, int net_w, const float* mean3, const float* std3, double fx, double fy, void* stream)
{
    if (!src || !dst || !mean3 || !std3 || h <= 0 || w <= 0 || nh <= 0 || nw <= 0 || net_h <= 0 || net_w <= 0 || !(fx > 0) || !(fy > 0))
        return SMAP_E_ARG;
    PrepArgs p{h, w, nh, nw, top, left, net_h, net_w, {mean3[0], mean3[1], mean3[2]}, {std3[0], std3[1], std3[2]}, 1.0 / fx, 1.0 / fy};
    hipLaunchKernelGGL(preprocess_kernel, dim3((net_w + 255) / 256, net_h), dim3(256), 0, (hipStream_t)stream, src, dst, p);
    return hip_rc(hipGetLastError());
}

extern "C" const char* smap_version(void) { return "smap_hip gfx950 r6 (assoc + backbone f16|x3 + flip-TTA + persistent conv + whole-Bottleneck launches + plan blob v2 + windowed arena + N segments + split K + lanes + two-input launches + tap-dot head + scaled head sum + two-launch peak search)"; }
