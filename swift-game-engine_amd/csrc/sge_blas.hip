// Skinned-geometry acceleration structures for gfx950: the per-frame refit the reference issues right after skinning
// (RTAccelerationBuilder.swift:113-145, `encoder.refit(... options: .vertexData)` per dynamic slice) and the reads the
// raytraceKernel makes of the skinned streams at a hit (RayTracing.metalinc:242-296).
//
// Layout (include/sge_amd.h, HostBlas::build): a 64-wide BVH whose topology is shared by all clones of the mesh; what
// changes per character and per frame is one box per entry, float[entryCount + 1][6].
//
// blas_refit_kernel — persistent workgroups, one character at a time. The skinned positions are read ONCE, coalesced, in
// vertex order, a tile of 2-4k vertices at a time into LDS; the routing "vertex -> the clusters it belongs to" (1.6
// clusters per vertex on the Y-Bot) is a schedule built once with the topology and shared by all clones: per tile, the
// vertices of every cluster that reaches into it, in chunks of 16. A lane walks one chunk from LDS with the running
// min / max in registers and folds the result into the cluster's row of an LDS table; no triangle ever gathers its three
// vertices from HBM. The inner entries are then reduced level by level from LDS, one wavefront per wide node, and the
// whole table is written out coalesced. HBM traffic per character: 12 B (16 B padded) per vertex in, 24 B per entry out;
// the index buffer is not read at all. (Versions measured on the way, 10k Y-Bots: every vertex folded into its clusters
// with LDS atomics, 4.5 ms — the LDS atomic rate is about one lane per clock per CU; a 16-lane group per chunk with a DPP
// row reduction, 1.6 ms — four times the instructions of the lane-sequential walk; DESIGN.md section 9 has the table.)
// The same per-tile work runs inside the LBS kernel when the two stages are fused (sge_skin.hip: skin_refit_kernel).
//
// blas_intersect_kernel — one wavefront per ray: a lane tests one entry's box per step, then one triangle of a
// cluster per lane; the closest hit is the wave minimum of (distance, primitive id), so the answer does not depend on
// the visiting order and equals a brute-force scan of the index buffer. Rays that name no character scan the characters'
// world boxes (blas_world_boxes_kernel) 64 per step and descend into the ones they may hit.
#include <algorithm>
#include <cstdlib>
#ifndef SGE_BLAS_EXPERIMENT
#define SGE_BLAS_EXPERIMENT 0
#endif
#if SGE_BLAS_EXPERIMENT == 4 // diagnostic build: stamps inside blasFinishCharacter as well (columns 8..11 of the phase table)
#include <hip/hip_runtime.h>
namespace sge { __device__ unsigned long long g_blasPhase[1024][12]; }
#define SGE_FINISH_STAMP_BEGIN() unsigned long long fstamp_ = __builtin_readcyclecounter()
#define SGE_FINISH_STAMP(k) do { if (tid == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); ::sge::g_blasPhase[blockIdx.x & 1023][8 + (k)] += now_ - fstamp_; fstamp_ = now_; } } while (0)
#endif
#include "sge_blas_dev.hpp"

#ifndef SGE_BLAS_EXPERIMENT
#define SGE_BLAS_EXPERIMENT 0
#endif

namespace sge {

#if SGE_BLAS_EXPERIMENT == 4 // diagnostic build: shader-clock cycles of every phase of a step, summed per workgroup (wave 0's view)
#define SGE_PHASE(k) do { if (tid == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); phase[k] += now_ - stamp; stamp = now_; } } while (0)
#else
#define SGE_PHASE(k) do { } while (0)
#endif

constexpr int kWave = 64;
typedef float v4f __attribute__((ext_vector_type(4)));

// Persistent workgroups (two or three per CU, as the LDS allows), each taking characters blockIdx.x, blockIdx.x + gridDim.x, ... LDS: tab[row * 6 + c]
// (c = 0..2 minima, 3..5 maxima), one tile of positions as X[], Y[], Z[], and the tiles' round ranges. The work is one
// flat sequence of (character, tile) steps. Per step: the workgroup's registers already hold the tile (its loads were
// issued a step earlier and completed behind the previous step's work); they are written to LDS; every wavefront requests
// its round of the schedule and then the NEXT step's positions — possibly the next character's first tile, so the
// end-of-character work below also runs with loads in flight; then every LANE takes one chunk — up to 16 vertices of this
// tile that belong to one cluster — walks it sequentially from LDS with the running min / max in registers (no cross-lane
// step), and folds the result into the cluster's row with six LDS float atomics. 64 chunks of similar length form a round =
// one wavefront's work. Memory loads retire in order: the round's words are requested before the positions, so the walk
// waits only for them.
// After a character's last tile: the inner entries are reduced from LDS level by level, one wavefront per wide node, the
// table is written out coalesced and re-initialised.
template <int STRIDE, int TILE>
__global__ __launch_bounds__(kBlasRefitBlock) void blas_refit_kernel(DevBlas B, const float* __restrict__ positions, long long firstVertex,
                                                                     int chars, float* __restrict__ bounds, int* __restrict__ queue) {
    extern __shared__ float lds[];
    const int rows = B.entryCount + 1, tid = threadIdx.x;
    float* tab = lds;
    float* X = lds + rows * 6; // Y = X + TILE, Z = X + 2 * TILE
    int* trs = reinterpret_cast<int*>(X + 3 * TILE);
    // the next character of this workgroup, behind the round ranges (in the dynamic region: a static __shared__ variable would
    // move the region's base off its 16-byte alignment)
    int& sNextChar = trs[B.tileCount + 1];
    int* topo = trs + B.tileCount + 2;
    blasTableInit(tab, rows, tid, kBlasRefitBlock);
    blasTopoStage(B, topo, tid, kBlasRefitBlock);
    for (int i = tid; i <= B.tileCount; i += kBlasRefitBlock) trs[i] = B.tileRoundStart[i];
    __syncthreads();
    const int lane = tid & (kWave - 1), wave = tid / kWave;
    constexpr int kWaves = kBlasRefitBlock / kWave, kPerThread = TILE / kBlasRefitBlock;
    const int n = B.tileCount, lastRound = trs[n] - 1;
    const float* P0 = positions + (size_t)firstVertex * STRIDE;
    const size_t charStride = (size_t)B.vertexCount * STRIDE;

    float px[kPerThread], py[kPerThread], pz[kPerThread];
    auto fetchPos = [&](int c, int tile) {
        const float* P = P0 + (size_t)c * charStride;
        const int base = tile * B.tileVerts, nv = min(B.tileVerts, B.vertexCount - base);
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            const int v = min(tid + k * kBlasRefitBlock, nv - 1); // past the end: reload the last vertex (never stored), no branch
            const float* p = P + (size_t)(base + v) * STRIDE;
            // streamed once: non-temporal
            if (STRIDE == 4) { const v4f q = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p)); px[k] = q.x; py[k] = q.y; pz[k] = q.z; }
            else { px[k] = __builtin_nontemporal_load(p); py[k] = __builtin_nontemporal_load(p + 1); pz[k] = __builtin_nontemporal_load(p + 2); }
        }
    };
    int c = blockIdx.x;
    if (c >= chars) return;
    // characters start at different tiles, so that the ones in flight at one time do not all read the same offset of their
    // (equally spaced) vertex ranges
    // Characters after the first are handed out through a ticket counter (zeroed by the launcher), gridDim.x + ticket: with a
    // fixed stride the workgroups that get one character more than the rest (10,000 over 768: 14 against 13) are the tail of
    // the launch, and so is every workgroup that found its place on a CU late. The ticket is drawn at the start of a
    // character and is needed in its last step, where the next character's first tile is requested: thread 0 keeps it in a
    // register until then (written to LDS in the first step, the wait for the atomic's answer — behind the tile loads in flight,
    // loads return in order — stood in front of every character's second barrier).
    int tile = c % n, done = 0, ticket = 0;
    fetchPos(c, tile);
#if SGE_BLAS_EXPERIMENT == 4
    unsigned long long phase[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp = __builtin_readcyclecounter();
#endif
    while (true) {
        const int nv = min(B.tileVerts, B.vertexCount - tile * B.tileVerts);
        if (done == 0 && tid == 0) ticket = atomicAdd(queue, 1);
        __syncthreads(); // the previous step's rounds have read X/Y/Z; a finished character's table has been re-initialised
        SGE_PHASE(0); // barrier 1 (the other wavefronts' walks)
#if SGE_BLAS_EXPERIMENT == 4
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SGE_PHASE(1); // the tile's loads
#endif
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            const int v = tid + k * kBlasRefitBlock;
            if (v < nv) { X[v] = px[k]; X[v + TILE] = py[k]; X[v + 2 * TILE] = pz[k]; }
        }
        if (done + 1 == n && tid == 0) sNextChar = (int)gridDim.x + ticket; // read below in this step, behind the barrier
        __syncthreads();
        SGE_PHASE(2); // LDS write + barrier 2
        const int rEnd = trs[tile + 1];
        int r = trs[tile] + wave;
        BlasRound R;
        if (r < rEnd) blasFetchRound(B, r, lastRound, lane, R); // wave-uniform; a wavefront without a round in this tile loads nothing
        // the next step: this character's next tile, or the next character's first
        const bool last = done + 1 == n;
        const int cNext = last ? __builtin_amdgcn_readfirstlane(sNextChar) : c;
        const int tileNext = last ? cNext % n : (tile + 1 == n ? 0 : tile + 1);
#if SGE_BLAS_EXPERIMENT != 2 // (diagnostic builds, tools/build_variant.sh: 1 = no walk, 2 = no position loads after the first; results are wrong)
        fetchPos(min(cNext, chars - 1), tileNext); // unconditional (a branch here would make the waits below conservative); after the last step: unused
#endif
        SGE_PHASE(3); // issue: round words + next tile
#if SGE_BLAS_EXPERIMENT == 4
        if (r < rEnd) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kPerThread) : "memory"); }
        SGE_PHASE(4); // the round words' latency
#endif
#if SGE_BLAS_EXPERIMENT != 1
        if (r < rEnd) blasWalk<TILE>(tab, rows, X, R); // wave-uniform
        for (r += kWaves; r < rEnd; r += kWaves) { // only when a tile has more rounds than the workgroup has wavefronts
            blasFetchRound(B, r, lastRound, lane, R);
            blasWalk<TILE>(tab, rows, X, R);
        }
#else
        if (r < rEnd && R.w[0] == 0xdeadbeefu) tab[0] = (float)R.cluster; // keeps the round's loads alive
#endif
        SGE_PHASE(5); // walk
        if (last) {
            blasFinishCharacter(B, topo, tab, rows, tid, kBlasRefitBlock, bounds + (size_t)c * rows * 6);
            SGE_PHASE(6); // end of character
#if SGE_BLAS_EXPERIMENT == 4
            if (cNext >= chars && tid == 0) { for (int k = 0; k < 8; ++k) g_blasPhase[blockIdx.x & 1023][k] += phase[k]; }
#endif
            if (cNext >= chars) return;
            c = cNext;
            done = 0;
        } else {
            ++done;
        }
        tile = tileNext;
    }
}

#if SGE_BLAS_EXPERIMENT == 4
} // namespace sge
extern "C" int sge_experiment_blas_phases(unsigned long long* out, int reset) { // [1024][12], diagnostic builds only
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sge::g_blasPhase), sizeof(sge::g_blasPhase)) != hipSuccess) return 1;
    if (reset) { static unsigned long long zero[1024][12]; if (hipMemcpyToSymbol(HIP_SYMBOL(sge::g_blasPhase), zero, sizeof(zero)) != hipSuccess) return 1; }
    return 0;
}
namespace sge {
#endif

template <int TILE>
static int launchRefitTile(const DevBlas& B, const float* p, int layout, long long firstVertex, int chars, float* bounds, int* queue, int grid, size_t lds, hipStream_t s) {
    static bool attrSet[kMaxDevices] = {};
    const int devSlot = currentDeviceSlot();
    if (!attrSet[devSlot]) {
        SGE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(blas_refit_kernel<3, TILE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBlasMaxLdsBytes + 64)); // + the ticket slot
        SGE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(blas_refit_kernel<4, TILE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBlasMaxLdsBytes + 64)); // + the ticket slot
        attrSet[devSlot] = true;
    }
    if (layout == SGE_LAYOUT_PADDED16) hipLaunchKernelGGL((blas_refit_kernel<4, TILE>), dim3(grid), dim3(kBlasRefitBlock), lds, s, B, p, firstVertex, chars, bounds, queue);
    else hipLaunchKernelGGL((blas_refit_kernel<3, TILE>), dim3(grid), dim3(kBlasRefitBlock), lds, s, B, p, firstVertex, chars, bounds, queue);
    return SGE_OK;
}

int launch_blas_refit(const DevBlas& B, const void* positions, int layout, long long firstVertex, int chars, float* bounds, int* queue, hipStream_t s) {
    if (chars <= 0) return SGE_OK;
    SGE_HIP(hipMemsetAsync(queue, 0, sizeof(int), s));
    const size_t lds = blasRefitLdsBytes(B.entryCount, B.tileCount, B.tileCap) + blasTopoBytes(B.wideCount, B.levels) + 16; // + the ticket slot
    const int cus = currentDeviceCUs();
    // persistent: as many workgroups as stay resident together (LDS allows floor(160 KB / lds) per CU)
    const int perCU = (int)std::max<size_t>(1, std::min<size_t>(4, (size_t)(160 * 1024) / (lds + 256)));
    const int grid = std::min(chars, cus * perCU);
    const float* p = reinterpret_cast<const float*>(positions);
    switch (B.tileCap) {
    case 3072: return launchRefitTile<3072>(B, p, layout, firstVertex, chars, bounds, queue, grid, lds, s);
    case 2048: return launchRefitTile<2048>(B, p, layout, firstVertex, chars, bounds, queue, grid, lds, s);
    default: return launchRefitTile<4096>(B, p, layout, firstVertex, chars, bounds, queue, grid, lds, s);
    }
}

// ---------------------------------------------------------------------------
// closest hit of one ray against one character
// ---------------------------------------------------------------------------
// Moller-Trumbore as the engine's own rayTriangle (CollisionQuery.swift:1575-1601), also returning the barycentrics
__device__ __forceinline__ bool rayTriangleUV(F3 origin, F3 direction, F3 v0, F3 v1, F3 v2, float eps, float& tOut, float& uOut, float& vOut) {
    F3 e1 = v1 - v0, e2 = v2 - v0;
    F3 pvec = cross(direction, e2);
    float det = dot(e1, pvec);
    if (fabsf(det) < eps) return false;
    float invDet = 1.0f / det;
    F3 tvec = origin - v0;
    float u = dot(tvec, pvec) * invDet;
    if (u < 0 || u > 1) return false;
    F3 qvec = cross(tvec, e1);
    float v = dot(direction, qvec) * invDet;
    if (v < 0 || (u + v) > 1) return false;
    float t = dot(e2, qvec) * invDet;
    if (!(t >= 0)) return false;
    tOut = t + 0.0f; // -0 -> +0
    uOut = u;
    vOut = v;
    return true;
}

struct Inv3 { F3 r0, r1, r2; float ok; }; // rows of the inverse of the 3x3 part
__device__ __forceinline__ Inv3 inverse3(const Aff& m) {
    const F3 a = m.c0, b = m.c1, c = m.c2;
    const F3 bc = cross(b, c), ca = cross(c, a), ab = cross(a, b);
    const float det = dot(a, bc);
    const float r = 1.0f / det;
    return Inv3{bc * r, ca * r, ab * r, det};
}
__device__ __forceinline__ F3 mul3(const Inv3& m, F3 v) { return F3{dot(m.r0, v), dot(m.r1, v), dot(m.r2, v)}; }

template <int STRIDE>
__device__ __forceinline__ F3 loadP(const float* base, uint32_t i) { const float* p = base + (size_t)i * STRIDE; return F3{p[0], p[1], p[2]}; }

constexpr int kBlasStack = kBlasTraversalStackCap;

__device__ __forceinline__ Aff loadInstance(const float* instances, int inst) {
    const float* Mf = instances + (size_t)inst * 16;
    return Aff{F3{Mf[0], Mf[1], Mf[2]}, F3{Mf[4], Mf[5], Mf[6]}, F3{Mf[8], Mf[9], Mf[10]}, F3{Mf[12], Mf[13], Mf[14]}};
}

// widened slab test (CollisionQuery.swift:1603-1630 form): a box is only skipped when it is clearly off the ray or clearly
// behind the best hit
__device__ __forceinline__ bool slabPass(const float* bx, F3 o, F3 inv, float tMin, float bestT) {
    float t0 = (bx[0] - o.x) * inv.x, t1 = (bx[3] - o.x) * inv.x;
    float lo = fminf(t0, t1), hi = fmaxf(t0, t1);
    t0 = (bx[1] - o.y) * inv.y; t1 = (bx[4] - o.y) * inv.y;
    lo = fmaxf(lo, fminf(t0, t1)); hi = fminf(hi, fmaxf(t0, t1));
    t0 = (bx[2] - o.z) * inv.z; t1 = (bx[5] - o.z) * inv.z;
    lo = fmaxf(lo, fminf(t0, t1)); hi = fminf(hi, fmaxf(t0, t1));
    const float slack = 1e-3f * fmaxf(fabsf(lo), fabsf(hi)) + 1e-4f;
    return (lo - slack <= hi + slack) && (hi + slack >= tMin) && (lo - slack <= bestT);
}
__device__ __forceinline__ F3 invDir(F3 d) {
    return F3{d.x != 0 ? 1.0f / d.x : kFloatMax, d.y != 0 ? 1.0f / d.y : kFloatMax, d.z != 0 ? 1.0f / d.z : kFloatMax};
}

// The instances' world-space boxes (what a TLAS build would start from): the box of the eight transformed corners of every
// character's root row, and above them one box per 64 consecutive characters (worldBoxes[chars ..]). One wavefront per
// 64 instances.
__global__ __launch_bounds__(kWave) void blas_world_boxes_kernel(BlasTrace T, float* worldBoxes) {
    const int i = blockIdx.x * kWave + threadIdx.x;
    F3 mn{kFloatMax, kFloatMax, kFloatMax}, mx{-kFloatMax, -kFloatMax, -kFloatMax};
    if (i < T.chars) {
        const Aff M = loadInstance(T.instances, i);
        const float* bx = T.bounds + ((size_t)i * (T.blas.entryCount + 1) + T.blas.entryCount) * 6;
        for (int k = 0; k < 8; ++k) {
            const F3 c = affMulPoint(M, F3{bx[(k & 1) ? 3 : 0], bx[(k & 2) ? 4 : 1], bx[(k & 4) ? 5 : 2]});
            mn = vmin(mn, c);
            mx = vmax(mx, c);
        }
        float* o = worldBoxes + (size_t)i * 6;
        o[0] = mn.x; o[1] = mn.y; o[2] = mn.z; o[3] = mx.x; o[4] = mx.y; o[5] = mx.z;
    }
    for (int off = 32; off > 0; off >>= 1) {
        mn = vmin(mn, F3{__shfl_xor(mn.x, off, kWave), __shfl_xor(mn.y, off, kWave), __shfl_xor(mn.z, off, kWave)});
        mx = vmax(mx, F3{__shfl_xor(mx.x, off, kWave), __shfl_xor(mx.y, off, kWave), __shfl_xor(mx.z, off, kWave)});
    }
    if (threadIdx.x == 0) {
        float* o = worldBoxes + ((size_t)T.chars + blockIdx.x) * 6;
        o[0] = mn.x; o[1] = mn.y; o[2] = mn.z; o[3] = mx.x; o[4] = mx.y; o[5] = mx.z;
    }
}

// One wavefront per ray. `instance >= 0`: that character only. `instance < 0`: every character — the instance level is two
// scans, 64 boxes per step: the boxes of 64 consecutive characters, then the world boxes of the groups the ray may hit (the
// reference rebuilds its TLAS every frame; here the per-frame work is one box reduction, and a crowd's index order is
// largely its spatial order), characters in ascending order, a later one winning only with a strictly smaller distance.
template <int STRIDE>
__global__ __launch_bounds__(kWave) void blas_intersect_kernel(BlasTrace T, const sge_blas_ray* rays, int n, sge_blas_hit* hits) {
    __shared__ int stack[kBlasStack];
    const int lane = threadIdx.x;
    const sge_blas_ray R = rays[blockIdx.x];
    const DevBlas& B = T.blas;
    sge_blas_hit H{};
    H.primitive = -1;
    H.instance = -1;
    if (R.instance >= T.chars) { if (lane == 0) hits[blockIdx.x] = H; return; }
    const F3 wo{R.origin[0], R.origin[1], R.origin[2]}, wd{R.direction[0], R.direction[1], R.direction[2]};
    const float tMin = smax(R.minDistance, 0.0f);

    float bestT = R.maxDistance, bestU = 0, bestV = 0;
    uint32_t bestPrim = 0;
    int bestInst = -1;

    // closest hit inside one character; accepted when closer than bestT (equal: only the first character found keeps it)
    auto traverse = [&](int inst) {
        const Aff M = loadInstance(T.instances, inst);
        const Inv3 Mi = inverse3(M);
        const F3 o = mul3(Mi, wo - M.c3), d = mul3(Mi, wd);
        const F3 inv = invDir(d);
        const float* P = reinterpret_cast<const float*>(T.positions) + (size_t)inst * B.vertexCount * STRIDE;
        const float* boxes = T.bounds + (size_t)inst * (B.entryCount + 1) * 6;
        // (distance bits, primitive id); distances are >= 0, so their bit patterns order like the values. A character found
        // earlier keeps a tie: the starting key carries primitive 0, which nothing is smaller than at equal distance.
        // (with the instances visited in grid order rather than index order, a character with a smaller index than the holder of
        // the best hit so far may still take it at equal distance: its starting key accepts any primitive there)
        unsigned long long best = ((unsigned long long)__float_as_uint(bestT) << 32) | ((bestInst < 0 || inst < bestInst) ? 0xffffffffull : 0ull);
        float u0 = 0, v0 = 0;
        bool found = false;
        int sp = 1;
        if (lane == 0) stack[0] = 0;
        __syncthreads();
        while (sp > 0) {
            const int w = stack[sp - 1];
            --sp;
            __syncthreads();
            const int first = B.wideFirst[w], cnt = B.wideFirst[w + 1] - first;
            bool pass = false;
            int2 link = make_int2(0, 0);
            if (lane < cnt) {
                link = B.entryLink[first + lane];
                pass = slabPass(boxes + (size_t)(first + lane) * 6, o, inv, tMin, __uint_as_float((unsigned)(best >> 32)));
            }
            const unsigned long long inner = __ballot(pass && link.x >= 0), leaves = __ballot(pass && link.x < 0);
            if (pass && link.x >= 0) {
                const int at = sp + __popcll(inner & ((1ull << lane) - 1));
                if (at < kBlasStack) stack[at] = link.x;
            }
            sp = min(sp + __popcll(inner), kBlasStack);
            unsigned long long m = leaves;
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int firstSlot = ~__shfl(link.x, src, kWave), count = __shfl(link.y, src, kWave);
                unsigned long long key = ~0ull;
                float u = 0, v = 0;
                if (lane < count) {
                    const uint32_t* ix = B.slotIndices + (size_t)(firstSlot + lane) * 3;
                    const F3 p0 = loadP<STRIDE>(P, ix[0]), p1 = loadP<STRIDE>(P, ix[1]), p2 = loadP<STRIDE>(P, ix[2]);
                    float t;
                    if (rayTriangleUV(o, d, p0, p1, p2, 1e-6f, t, u, v) && t >= tMin && t <= R.maxDistance)
                        key = ((unsigned long long)__float_as_uint(t) << 32) | B.slotTriangle[firstSlot + lane];
                }
                unsigned long long k = key;
                for (int off = 32; off > 0; off >>= 1) {
                    const unsigned lo32 = __shfl_xor((unsigned)k, off, kWave), hi32 = __shfl_xor((unsigned)(k >> 32), off, kWave);
                    const unsigned long long other = ((unsigned long long)hi32 << 32) | lo32;
                    k = other < k ? other : k;
                }
                if (k != ~0ull && k < best) {
                    best = k;
                    found = true;
                    const int winner = __ffsll((long long)__ballot(key == k)) - 1;
                    u0 = __shfl(u, winner, kWave);
                    v0 = __shfl(v, winner, kWave);
                }
            }
            __syncthreads();
        }
        if (found) {
            bestT = __uint_as_float((unsigned)(best >> 32));
            bestPrim = (uint32_t)best;
            bestU = u0; bestV = v0;
            bestInst = inst;
        }
    };

    if (R.instance >= 0) {
        traverse(R.instance);
    } else if (T.worldBoxesValid && T.instOrder) {
        // three scans of 64 boxes: super-groups (4,096 instances each: 62 of them hold 250,000 characters, one step), the groups of
        // the ones the ray may hit, the instances of those groups — grouped by where they ARE (grid order), not by their index
        const F3 inv = invDir(wd);
        const int groups = (T.chars + kWave - 1) / kWave, supers = (groups + kWave - 1) / kWave;
        for (int sbase = 0; sbase < supers; sbase += kWave) {
            const int sg = sbase + lane;
            unsigned long long sm = __ballot(sg < supers && slabPass(T.superBoxes + (size_t)sg * 6, wo, inv, tMin, bestT));
            while (sm) {
                const int ssrc = __ffsll((long long)sm) - 1;
                sm &= sm - 1;
                const int g = (sbase + ssrc) * kWave + lane;
                unsigned long long gm = __ballot(g < groups && slabPass(T.groupBoxes + (size_t)g * 6, wo, inv, tMin, bestT));
                while (gm) {
                    const int gsrc = __ffsll((long long)gm) - 1;
                    gm &= gm - 1;
                    const int slot = ((sbase + ssrc) * kWave + gsrc) * kWave + lane;
                    const int inst = slot < T.chars ? T.instOrder[slot] : -1;
                    const bool pass = inst >= 0 && slabPass(T.worldBoxes + (size_t)inst * 6, wo, inv, tMin, bestT);
                    unsigned long long m = __ballot(pass);
                    while (m) {
                        const int src = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        traverse(__shfl(inst, src, kWave));
                    }
                }
            }
        }
    } else if (T.worldBoxesValid) {
        const F3 inv = invDir(wd);
        const int groups = (T.chars + kWave - 1) / kWave;
        const float* groupBoxes = T.worldBoxes + (size_t)T.chars * 6;
        for (int gbase = 0; gbase < groups; gbase += kWave) {
            const int g = gbase + lane;
            unsigned long long gm = __ballot(g < groups && slabPass(groupBoxes + (size_t)g * 6, wo, inv, tMin, bestT));
            while (gm) {
                const int gsrc = __ffsll((long long)gm) - 1;
                gm &= gm - 1;
                const int base = (gbase + gsrc) * kWave, i = base + lane;
                const bool pass = i < T.chars && slabPass(T.worldBoxes + (size_t)i * 6, wo, inv, tMin, bestT);
                unsigned long long m = __ballot(pass);
                while (m) {
                    const int src = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    traverse(base + src);
                }
            }
        }
    }
    if (lane != 0) return;
    if (bestInst >= 0) {
        const Aff M = loadInstance(T.instances, bestInst);
        const size_t vbase = (size_t)bestInst * B.vertexCount;
        const float* P = reinterpret_cast<const float*>(T.positions) + vbase * STRIDE;
        const uint32_t* ix = T.indices + (size_t)bestPrim * 3;
        const uint32_t i0 = ix[0], i1 = ix[1], i2 = ix[2];
        // RayTracing.metalinc:258-268
        const F3 w0 = affMulPoint(M, loadP<STRIDE>(P, i0)), w1 = affMulPoint(M, loadP<STRIDE>(P, i1)), w2 = affMulPoint(M, loadP<STRIDE>(P, i2));
        F3 N = normalize(cross(w1 - w0, w2 - w0));
        if (dot(N, wd) > 0.0f) N = -N;
        // :283-300
        const float* NB = reinterpret_cast<const float*>(T.normals) + vbase * STRIDE;
        const float* TB = T.tangents + vbase * 4;
        const float bx = bestU, by = bestV, bw = 1.0f - bx - by;
        const F3 n0 = loadP<STRIDE>(NB, i0), n1 = loadP<STRIDE>(NB, i1), n2 = loadP<STRIDE>(NB, i2);
        const float* q0 = TB + (size_t)i0 * 4; const float* q1 = TB + (size_t)i1 * 4; const float* q2 = TB + (size_t)i2 * 4;
        const F3 nObj = normalize((n0 * bw + n1 * bx) + n2 * by);
        float t4[4];
        for (int k = 0; k < 4; ++k) t4[k] = (q0[k] * bw + q1[k] * bx) + q2[k] * by;
        const float r4 = 1.0f / sqrtf(((t4[0] * t4[0] + t4[1] * t4[1]) + t4[2] * t4[2]) + t4[3] * t4[3]);
        const F3 tObj = normalize(F3{t4[0] * r4, t4[1] * r4, t4[2] * r4});
        const float tw = t4[3] * r4;
        const F3 nW = normalize((M.c0 * nObj.x + M.c1 * nObj.y) + M.c2 * nObj.z);
        const F3 tW = normalize((M.c0 * tObj.x + M.c1 * tObj.y) + M.c2 * tObj.z);
        const F3 bW = normalize(cross(nW, tW) * tw);
        H.hit = 1; H.primitive = (int32_t)bestPrim; H.instance = bestInst; H.distance = bestT; H.bary[0] = bx; H.bary[1] = by;
        H.geomNormal[0] = N.x; H.geomNormal[1] = N.y; H.geomNormal[2] = N.z;
        H.normal[0] = nW.x; H.normal[1] = nW.y; H.normal[2] = nW.z;
        H.tangent[0] = tW.x; H.tangent[1] = tW.y; H.tangent[2] = tW.z;
        H.bitangent[0] = bW.x; H.bitangent[1] = bW.y; H.bitangent[2] = bW.z;
        if (T.uvs) { // interp_uv, RayTracing.metalinc:106-119
            const float* a0 = T.uvs + (size_t)i0 * 2; const float* a1 = T.uvs + (size_t)i1 * 2; const float* a2 = T.uvs + (size_t)i2 * 2;
            H.uv[0] = (a0[0] * bw + a1[0] * bx) + a2[0] * by;
            H.uv[1] = (a0[1] * bw + a1[1] * bx) + a2[1] * by;
        }
    }
    hits[blockIdx.x] = H;
}

__global__ void blas_instance_points_kernel(const float* __restrict__ worldBoxes, int chars, sge_agent_state* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= chars) return;
    const float* b = worldBoxes + (size_t)i * 6;
    sge_agent_state a{};
    a.position[0] = 0.5f * (b[0] + b[3]); a.position[1] = 0.5f * (b[1] + b[4]); a.position[2] = 0.5f * (b[2] + b[5]);
    const float hx = 0.5f * (b[3] - b[0]), hz = 0.5f * (b[5] - b[2]);
    a.radius = fmaxf(sqrtf(hx * hx + hz * hz), 0.0f); // >= 0: every instance is binned (a character without triangles has an inverted box: NaN-free by the fmaxf)
    if (!(a.radius >= 0.0f) || !(a.position[0] == a.position[0]) || !(a.position[2] == a.position[2])) { a.position[0] = a.position[1] = a.position[2] = 0.0f; a.radius = 0.0f; }
    a.halfHeight = 0.5f * (b[4] - b[1]);
    out[i] = a;
}
void launch_blas_instance_points(const float* worldBoxes, int chars, sge_agent_state* out, hipStream_t s) {
    if (chars > 0) hipLaunchKernelGGL(blas_instance_points_kernel, dim3((chars + 255) / 256), dim3(256), 0, s, worldBoxes, chars, out);
}
// level 0: one wavefront per group of 64 entries of `order`; level 1 (second launch, order == nullptr): per 64 group boxes
__global__ __launch_bounds__(kWave) void blas_group_boxes_kernel(const float* __restrict__ boxes, const int* __restrict__ order, int count, float* __restrict__ out) {
    const int slot = blockIdx.x * kWave + threadIdx.x;
    F3 mn{kFloatMax, kFloatMax, kFloatMax}, mx{-kFloatMax, -kFloatMax, -kFloatMax};
    if (slot < count) {
        const float* b = boxes + (size_t)(order ? order[slot] : slot) * 6;
        mn = F3{b[0], b[1], b[2]}; mx = F3{b[3], b[4], b[5]};
    }
    for (int off = 32; off > 0; off >>= 1) {
        mn = vmin(mn, F3{__shfl_xor(mn.x, off, kWave), __shfl_xor(mn.y, off, kWave), __shfl_xor(mn.z, off, kWave)});
        mx = vmax(mx, F3{__shfl_xor(mx.x, off, kWave), __shfl_xor(mx.y, off, kWave), __shfl_xor(mx.z, off, kWave)});
    }
    if (threadIdx.x == 0) {
        float* o = out + (size_t)blockIdx.x * 6;
        o[0] = mn.x; o[1] = mn.y; o[2] = mn.z; o[3] = mx.x; o[4] = mx.y; o[5] = mx.z;
    }
}
void launch_blas_group_boxes(const float* worldBoxes, const int* order, int chars, float* groupBoxes, float* superBoxes, hipStream_t s) {
    if (chars <= 0) return;
    const int groups = (chars + kWave - 1) / kWave, supers = (groups + kWave - 1) / kWave;
    hipLaunchKernelGGL(blas_group_boxes_kernel, dim3(groups), dim3(kWave), 0, s, worldBoxes, order, chars, groupBoxes);
    hipLaunchKernelGGL(blas_group_boxes_kernel, dim3(supers), dim3(kWave), 0, s, groupBoxes, (const int*)nullptr, groups, superBoxes);
}
void launch_blas_world_boxes(const BlasTrace& T, float* worldBoxes, hipStream_t s) {
    if (T.chars > 0) hipLaunchKernelGGL(blas_world_boxes_kernel, dim3((T.chars + kWave - 1) / kWave), dim3(kWave), 0, s, T, worldBoxes);
}

void launch_blas_intersect(BlasTrace T, const sge_blas_ray* d_rays, int n, sge_blas_hit* d_hits, bool anyInstance, hipStream_t s) {
    T.worldBoxesValid = anyInstance && T.worldBoxes != nullptr ? 1 : 0;
    if (n <= 0) return;
    // (the caller has refreshed the world boxes and, when it passes instOrder, the grid-ordered instance level: sge_api.hip)
    if (anyInstance && T.chars > 0 && !T.instOrder && !T.worldBoxes) T.worldBoxesValid = 0; // nothing refreshed: rays without an instance miss
    if (T.layout == SGE_LAYOUT_PADDED16) hipLaunchKernelGGL((blas_intersect_kernel<4>), dim3(n), dim3(kWave), 0, s, T, d_rays, n, d_hits);
    else hipLaunchKernelGGL((blas_intersect_kernel<3>), dim3(n), dim3(kWave), 0, s, T, d_rays, n, d_hits);
}

} // namespace sge
