// 4-weight linear-blend skinning for gfx950 — replaces the reference's Metal
// `skinningKernel` (Game/RayTracing.metalinc:737-776) and the per-job dispatch of
// RTSkinningEncoder.encode (Game/RTSkinningEncoder.swift:27-56).
//
// Mapping: one 256-thread workgroup (4 wavefronts) per character — small crowds split a character over several —
// one thread per vertex, walking the vertex stream 256 at a time with the next chunk's source attributes in flight.
// The character's bone palette (<= 256 x 3 float4 rows, 3.1 KB for the 65-bone Y-Bot) is staged once per workgroup
// in LDS; source streams are SoA and read coalesced (they are shared by all clones and stay L2-resident); the three
// output streams are written coalesced with non-temporal stores and are the HBM traffic (40 B/vertex packed).
//
// Arithmetic: the reference accumulates acc += (palette[idx_j] * v).xyz * w_j over the influences with w_j > 0;
// by linearity this kernel blends the matrices first and transforms once (build with -DSGE_SKIN_PER_INFLUENCE for the
// literal order); normals/tangents are normalised with v_rsq_f32.
// This translation unit is compiled with the default -ffp-contract=fast: the Metal
// original is built with MTL_FAST_MATH (project.pbxproj:328,386), there are no
// thresholds downstream of this arithmetic, and the parity bound is 1e-5 relative.
#include <algorithm>
#include "sge_blas_dev.hpp"

namespace sge {

#ifndef SGE_SKIN_UNCOND
#define SGE_SKIN_UNCOND 2 // influences blended without a `w > 0` test (weights clamped at 0 instead)
#endif
#ifndef SGE_SKIN_BLOCK
#define SGE_SKIN_BLOCK 256
#endif
constexpr int kSkinBlock = SGE_SKIN_BLOCK;

typedef float v4f __attribute__((ext_vector_type(4)));
struct Row3 { float4 r0, r1, r2; };

__device__ __forceinline__ Row3 loadRows(const float4* pal, int bone) {
    Row3 m;
    m.r0 = pal[bone * 3 + 0];
    m.r1 = pal[bone * 3 + 1];
    m.r2 = pal[bone * 3 + 2];
    return m;
}

// (M * (v, w)).xyz with the reference's left-to-right column accumulation
__device__ __forceinline__ float3 xform(const Row3& m, float3 v, float w) {
    float3 r;
    r.x = ((m.r0.x * v.x + m.r0.y * v.y) + m.r0.z * v.z) + m.r0.w * w;
    r.y = ((m.r1.x * v.x + m.r1.y * v.y) + m.r1.z * v.z) + m.r1.w * w;
    r.z = ((m.r2.x * v.x + m.r2.y * v.y) + m.r2.z * v.z) + m.r2.w * w;
    return r;
}

__device__ __forceinline__ float3 normalizeFast(float3 a) {
    float d = (a.x * a.x + a.y * a.y) + a.z * a.z;
    float r = __builtin_amdgcn_rsqf(d);
    return make_float3(a.x * r, a.y * r, a.z * r);
}

struct VertexIn { float3 p, n; float4 t; ushort4 idx; float4 w; };

template <int SRC_STRIDE>
__device__ __forceinline__ VertexIn loadVertex(const SkinLaunch& L, int gid) {
    VertexIn v;
#ifdef SGE_SKIN_EXPERIMENT_SMALL_SOURCE // diagnostic build: every load hits the same 256 source vertices (L1 resident)
    gid &= 255;
#endif
    const float* sp = reinterpret_cast<const float*>(L.srcPos) + (unsigned)gid * SRC_STRIDE;
    const float* sn = reinterpret_cast<const float*>(L.srcNrm) + (unsigned)gid * SRC_STRIDE;
    v.p = make_float3(sp[0], sp[1], sp[2]);
    v.n = make_float3(sn[0], sn[1], sn[2]);
    v.t = reinterpret_cast<const float4*>(L.srcTan)[gid];
    v.idx = reinterpret_cast<const ushort4*>(L.srcIdx)[gid];
    v.w = reinterpret_cast<const float4*>(L.srcWgt)[gid];
    return v;
}

#ifndef SGE_SKIN_MIN_BLOCKS
#define SGE_SKIN_MIN_BLOCKS 1
#endif
typedef float v2f __attribute__((ext_vector_type(2)));

// One workgroup = one character (or 1/splits of its vertices): the palette is staged once,
// then the 256 threads walk the vertex stream 256 at a time with the next chunk's source
// attributes already in flight while the current chunk is transformed and stored.
// The work of one workgroup: vertices [vBegin, vEnd) of character `c` of launch L (palette L.palettes[c]).
//
// LDS palette layout, 12 floats per bone (m_rc = row r, column c of the bone's 3x4 affine part):
//     (m00 m10 | m01 m11) (m02 m12 | m03 m13) (m20 m21 | m22 m23)
// so that one ds_read_b128 lands two ready-made register pairs for v_pk_fma_f32: the matrices are blended pair by pair
// (6 packed ops per influence) and the x / y components of position, normal and tangent come out of packed column sums
// A*v.x + B*v.y + C*v.z (+ D) without any register shuffling (the row-major layout of round 1 cost ~35 % of the kernel's vector
// instructions in v_mov_b32). The kernel shares the SIMDs with the collision kernels of the next step (SGE_OPT_OVERLAP_SKIN), where
// vector-issue slots, not HBM, are the contended resource.
template <int SRC_STRIDE, int DST_STRIDE>
__device__ __forceinline__ void skinRange(const SkinLaunch& L, const int c, const int vBegin, const int vEnd, float4* pal) {
    const int tid = threadIdx.x;
    int gid = vBegin + tid;
    VertexIn cur{};
    if (gid < vEnd) cur = loadVertex<SRC_STRIDE>(L, gid);

    // stage the palette: thread -> one float4 COLUMN (coalesced), scattered into the pair layout
    const float4* gp = reinterpret_cast<const float4*>(L.palettes + (size_t)c * L.paletteCount * 16);
    float* palf = reinterpret_cast<float*>(pal);
    for (int i = tid; i < L.paletteCount * 4; i += kSkinBlock) {
        float4 col = gp[i];
        int bone = i >> 2, cc = i & 3;
        palf[bone * 12 + 2 * cc + 0] = col.x;
        palf[bone * 12 + 2 * cc + 1] = col.y;
        palf[bone * 12 + 8 + cc] = col.z;
    }
    __syncthreads();

    // the character's slice of the three output streams: uniform 64-bit bases, 32-bit per-thread offsets
    const size_t obase = (size_t)L.dstBaseVertex + (size_t)c * L.vertexCount;
    float* const opBase = reinterpret_cast<float*>(L.outPos) + obase * DST_STRIDE;
    float* const onBase = reinterpret_cast<float*>(L.outNrm) + obase * DST_STRIDE;
    v4f* const otBase = reinterpret_cast<v4f*>(L.outTan) + obase;
    const v4f* P = reinterpret_cast<const v4f*>(pal);

    // one vertex: blend the influences' matrices, transform position / normal / tangent once, stream the result out.
    // sum_j w_j (M_j v) = (sum_j w_j M_j) v (the Metal kernel transforms per influence and blends the results, RayTracing.metalinc:
    // 758-775); the first two influences are taken unconditionally with the weight clamped at 0 (a weight <= 0 contributes exactly
    // nothing, as the reference's `w > 0` test does); the rarer third and fourth stay behind their tests.
    auto skinOne = [&](const VertexIn& v, int g) {
        const float w0 = fmaxf(v.w.x, 0.0f), w1 = fmaxf(v.w.y, 0.0f);
        v2f A, B, C, D, E, F;
        {
            const v4f q0 = P[v.idx.x * 3 + 0], q1 = P[v.idx.x * 3 + 1], q2 = P[v.idx.x * 3 + 2];
            A = q0.xy * w0; B = q0.zw * w0; C = q1.xy * w0; D = q1.zw * w0; E = q2.xy * w0; F = q2.zw * w0;
        }
#define SGE_BLEND(BONE, WGT)                                                                   \
        {                                                                                      \
            const v4f q0 = P[(BONE) * 3 + 0], q1 = P[(BONE) * 3 + 1], q2 = P[(BONE) * 3 + 2];  \
            A += q0.xy * (WGT); B += q0.zw * (WGT); C += q1.xy * (WGT); D += q1.zw * (WGT);    \
            E += q2.xy * (WGT); F += q2.zw * (WGT);                                            \
        }
        SGE_BLEND(v.idx.y, w1)
        if (v.w.z > 0.0f) SGE_BLEND(v.idx.z, v.w.z)
        if (v.w.w > 0.0f) SGE_BLEND(v.idx.w, v.w.w)
#undef SGE_BLEND
        // x, y: packed column sums; z: row 2 = (E.x, E.y, F.x, F.y)
        const v2f pxy = A * v.p.x + (B * v.p.y + (C * v.p.z + D));
        const v2f pe = E * v2f{v.p.x, v.p.y};
        const float pz = (pe.x + pe.y) + (F.x * v.p.z + F.y);
        v2f nxy = A * v.n.x + (B * v.n.y + C * v.n.z);
        const v2f ne = E * v2f{v.n.x, v.n.y};
        float nz = (ne.x + ne.y) + F.x * v.n.z;
        v2f txy = A * v.t.x + (B * v.t.y + C * v.t.z);
        const v2f te = E * v2f{v.t.x, v.t.y};
        float tz = (te.x + te.y) + F.x * v.t.z;
        {
            const v2f sq = nxy * nxy;
            const float r = __builtin_amdgcn_rsqf((sq.x + sq.y) + nz * nz);
            nxy *= r; nz *= r;
        }
        {
            const v2f sq = txy * txy;
            const float r = __builtin_amdgcn_rsqf((sq.x + sq.y) + tz * tz);
            txy *= r; tz *= r;
        }
        float* op = opBase + (unsigned)g * DST_STRIDE;
        float* on = onBase + (unsigned)g * DST_STRIDE;
#ifndef SGE_SKIN_PLAIN_STORES
        // streaming output: non-temporal stores (0.95-1.07 ms vs 1.10 ms with plain stores on MI355X)
        if (DST_STRIDE == 4) {
            __builtin_nontemporal_store(v4f{pxy.x, pxy.y, pz, 0.f}, reinterpret_cast<v4f*>(op));
            __builtin_nontemporal_store(v4f{nxy.x, nxy.y, nz, 0.f}, reinterpret_cast<v4f*>(on));
        } else {
            __builtin_nontemporal_store(pxy.x, op); __builtin_nontemporal_store(pxy.y, op + 1); __builtin_nontemporal_store(pz, op + 2);
            __builtin_nontemporal_store(nxy.x, on); __builtin_nontemporal_store(nxy.y, on + 1); __builtin_nontemporal_store(nz, on + 2);
        }
        __builtin_nontemporal_store(v4f{txy.x, txy.y, tz, v.t.w}, otBase + (unsigned)g);
#else
        if (DST_STRIDE == 4) {
            *reinterpret_cast<float4*>(op) = make_float4(pxy.x, pxy.y, pz, 0.f);
            *reinterpret_cast<float4*>(on) = make_float4(nxy.x, nxy.y, nz, 0.f);
        } else {
            op[0] = pxy.x; op[1] = pxy.y; op[2] = pz;
            on[0] = nxy.x; on[1] = nxy.y; on[2] = nz;
        }
        reinterpret_cast<float4*>(otBase)[(unsigned)g] = make_float4(txy.x, txy.y, tz, v.t.w);
#endif
    };

    // two-deep software pipeline written out, so that the chunk in flight never has to be copied between registers
    if (gid >= vEnd) return;
    VertexIn nxt{};
    while (true) {
        const int g1 = gid + kSkinBlock;
        const bool has1 = g1 < vEnd;
        if (has1) nxt = loadVertex<SRC_STRIDE>(L, g1);
        skinOne(cur, gid);
        if (!has1) break;
        const int g2 = g1 + kSkinBlock;
        const bool has2 = g2 < vEnd;
        if (has2) cur = loadVertex<SRC_STRIDE>(L, g2);
        skinOne(nxt, g1);
        if (!has2) break;
        gid = g2;
    }
}

// CPW characters per work unit: every source vertex is loaded ONCE and skinned with CPW palettes (all CPW staged in LDS, the pair
// layout of skinRange). Alone the kernel is as fast as the one-character form (it is bound by its stores); beside the next step's
// collision kernels the per-CU memory pipeline is what the two sides share, and the one-character form puts 64 B of source loads per
// vertex into it next to 40 B of stores — 9 GB of L2 reads per 10k-character launch. With CPW = 4 that is 16 B per vertex.
// Same arithmetic as skinRange (shared through SGE_SKIN_ONE), so the three output streams are bit-identical.
template <int DST_STRIDE, int CPW>
__device__ __forceinline__ void skinRangeMulti(const SkinLaunch& L, const int c0, const int nChars, const int vBegin, const int vEnd, float4* pal) {
    const int tid = threadIdx.x;
    int gid = vBegin + tid;
    VertexIn cur{};
    if (gid < vEnd) cur = loadVertex<3>(L, gid);
    const int rows = L.paletteCount * 3; // float4 rows of one staged palette
    float* palf = reinterpret_cast<float*>(pal);
    for (int k = 0; k < nChars; ++k) {
        const float4* gp = reinterpret_cast<const float4*>(L.palettes + (size_t)(c0 + k) * L.paletteCount * 16);
        for (int i = tid; i < L.paletteCount * 4; i += kSkinBlock) {
            float4 col = gp[i];
            int bone = i >> 2, cc = i & 3;
            float* dst = palf + (size_t)k * rows * 4 + bone * 12;
            dst[2 * cc + 0] = col.x;
            dst[2 * cc + 1] = col.y;
            dst[8 + cc] = col.z;
        }
    }
    __syncthreads();
    const size_t obase = (size_t)L.dstBaseVertex + (size_t)c0 * L.vertexCount;
    float* const opBase = reinterpret_cast<float*>(L.outPos) + obase * DST_STRIDE;
    float* const onBase = reinterpret_cast<float*>(L.outNrm) + obase * DST_STRIDE;
    v4f* const otBase = reinterpret_cast<v4f*>(L.outTan) + obase;
    const size_t charStride = (size_t)L.vertexCount;

    auto skinAll = [&](const VertexIn& v, int g) {
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
            if (k >= nChars) break;
            const v4f* P = reinterpret_cast<const v4f*>(pal) + (size_t)k * rows;
            const float w0 = fmaxf(v.w.x, 0.0f), w1 = fmaxf(v.w.y, 0.0f); // (per character: two registers fewer across the loop than two clamps kept)
            v2f A, B, C, D, E, F;
            {
                const v4f q0 = P[v.idx.x * 3 + 0], q1 = P[v.idx.x * 3 + 1], q2 = P[v.idx.x * 3 + 2];
                A = q0.xy * w0; B = q0.zw * w0; C = q1.xy * w0; D = q1.zw * w0; E = q2.xy * w0; F = q2.zw * w0;
            }
#define SGE_BLEND(BONE, WGT)                                                                   \
            {                                                                                  \
                const v4f q0 = P[(BONE) * 3 + 0], q1 = P[(BONE) * 3 + 1], q2 = P[(BONE) * 3 + 2]; \
                A += q0.xy * (WGT); B += q0.zw * (WGT); C += q1.xy * (WGT); D += q1.zw * (WGT); \
                E += q2.xy * (WGT); F += q2.zw * (WGT);                                        \
            }
            SGE_BLEND(v.idx.y, w1)
            if (v.w.z > 0.0f) SGE_BLEND(v.idx.z, v.w.z)
            if (v.w.w > 0.0f) SGE_BLEND(v.idx.w, v.w.w)
#undef SGE_BLEND
            const v2f pxy = A * v.p.x + (B * v.p.y + (C * v.p.z + D));
            const v2f pe = E * v2f{v.p.x, v.p.y};
            const float pz = (pe.x + pe.y) + (F.x * v.p.z + F.y);
            v2f nxy = A * v.n.x + (B * v.n.y + C * v.n.z);
            const v2f ne = E * v2f{v.n.x, v.n.y};
            float nz = (ne.x + ne.y) + F.x * v.n.z;
            v2f txy = A * v.t.x + (B * v.t.y + C * v.t.z);
            const v2f te = E * v2f{v.t.x, v.t.y};
            float tz = (te.x + te.y) + F.x * v.t.z;
            {
                const v2f sq = nxy * nxy;
                const float r = __builtin_amdgcn_rsqf((sq.x + sq.y) + nz * nz);
                nxy *= r; nz *= r;
            }
            {
                const v2f sq = txy * txy;
                const float r = __builtin_amdgcn_rsqf((sq.x + sq.y) + tz * tz);
                txy *= r; tz *= r;
            }
            float* op = opBase + (size_t)k * charStride * DST_STRIDE + (unsigned)g * DST_STRIDE;
            float* on = onBase + (size_t)k * charStride * DST_STRIDE + (unsigned)g * DST_STRIDE;
            if (DST_STRIDE == 4) {
                __builtin_nontemporal_store(v4f{pxy.x, pxy.y, pz, 0.f}, reinterpret_cast<v4f*>(op));
                __builtin_nontemporal_store(v4f{nxy.x, nxy.y, nz, 0.f}, reinterpret_cast<v4f*>(on));
            } else {
                __builtin_nontemporal_store(pxy.x, op); __builtin_nontemporal_store(pxy.y, op + 1); __builtin_nontemporal_store(pz, op + 2);
                __builtin_nontemporal_store(nxy.x, on); __builtin_nontemporal_store(nxy.y, on + 1); __builtin_nontemporal_store(nz, on + 2);
            }
            __builtin_nontemporal_store(v4f{txy.x, txy.y, tz, v.t.w}, otBase + (size_t)k * charStride + (unsigned)g);
        }
    };

    if (gid >= vEnd) return;
    if (CPW >= 4) {
        // four characters per loaded vertex: the loop body is ~1,200 instructions and 160 B of stores per lane, so the next chunk's
        // source vertex is requested only when the current one is done with — one L2 latency per chunk, 16 registers fewer
        // (the prefetching form needs 113, which leaves a SIMD room for one collision wavefront instead of two)
        while (true) {
            skinAll(cur, gid);
            gid += kSkinBlock;
            if (gid >= vEnd) break;
            cur = loadVertex<3>(L, gid);
        }
        return;
    }
    VertexIn nxt{};
    while (true) {
        const int g1 = gid + kSkinBlock;
        const bool has1 = g1 < vEnd;
        if (has1) nxt = loadVertex<3>(L, g1);
        skinAll(cur, gid);
        if (!has1) break;
        const int g2 = g1 + kSkinBlock;
        const bool has2 = g2 < vEnd;
        if (has2) cur = loadVertex<3>(L, g2);
        skinAll(nxt, g1);
        if (!has2) break;
        gid = g2;
    }
}

// The ticket counter of the resident forms is left at zero by the launch itself: queue[1] counts the workgroups that have drawn
// their last ticket, and the last of them clears both words (every ticket has been answered by then: a workgroup waits for its
// ticket before it goes on). The launcher's memset in front of every launch was a 5 us fill kernel plus a dispatch on the skin
// stream, between two skin launches that follow each other directly in overlap mode. (The words are zeroed once, at allocation.)
__device__ __forceinline__ void ticketsDone(int* __restrict__ queue) {
    if (threadIdx.x == 0 && atomicAdd(queue + 1, 1) == (int)gridDim.x - 1) {
        __hip_atomic_store(queue, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(queue + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The resident form with CPW characters per work unit (unit u = (group of CPW consecutive characters, vertex split)); dynamic LDS:
// CPW palettes of L.paletteCount bones + the ticket slot behind them.
template <int DST_STRIDE, int CPW>
#ifndef SGE_SKIN_MULTI_MIN_BLOCKS
#define SGE_SKIN_MULTI_MIN_BLOCKS SGE_SKIN_MIN_BLOCKS
#endif
__global__ __launch_bounds__(kSkinBlock, SGE_SKIN_MULTI_MIN_BLOCKS) void skin_ticket_multi_kernel(SkinLaunch L, int splits, int vertsPerSplit, int* __restrict__ queue) {
    extern __shared__ float4 palDyn[];
    int* const sNextUnit = reinterpret_cast<int*>(palDyn + (size_t)CPW * L.paletteCount * 3); // (behind the palettes: their 16-byte alignment stays)
    const int groups = (L.chars + CPW - 1) / CPW;
    for (int u = blockIdx.x; u < groups * splits;) {
        const int gidx = u / splits;
        const int sp = u - gidx * splits;
        const int vBegin = sp * vertsPerSplit;
        const int c0 = gidx * CPW;
        int ticket = 0;
        if (threadIdx.x == 0) ticket = atomicAdd(queue, 1);
        skinRangeMulti<DST_STRIDE, CPW>(L, c0, min(CPW, L.chars - c0), vBegin, min(L.vertexCount, vBegin + vertsPerSplit), palDyn);
        if (threadIdx.x == 0) *sNextUnit = (int)gridDim.x + ticket;
        __syncthreads(); // also: every thread is done with the palettes
        u = *sNextUnit;
    }
    ticketsDone(queue);
}

template <int SRC_STRIDE, int DST_STRIDE>
__global__ __launch_bounds__(kSkinBlock, SGE_SKIN_MIN_BLOCKS) void skin_kernel(SkinLaunch L, int splits, int vertsPerSplit, int wavePriority) {
    __shared__ float4 pal[SGE_MAX_BONES * 3];
    // issue priority over the waves of other kernels on the same SIMD (s_setprio takes an immediate)
    if (wavePriority == 1) __builtin_amdgcn_s_setprio(1);
    else if (wavePriority == 2) __builtin_amdgcn_s_setprio(2);
    else if (wavePriority == 3) __builtin_amdgcn_s_setprio(3);
    // persistent form (gridDim.x < chars * splits): a workgroup keeps its place on the CU and walks over the work units
    for (int u = blockIdx.x; u < L.chars * splits; u += gridDim.x) {
        const int c = u / splits;
        const int sp = u - c * splits;
        const int vBegin = sp * vertsPerSplit;
        if (u != (int)blockIdx.x) __syncthreads(); // every thread is done with the previous palette
        skinRange<SRC_STRIDE, DST_STRIDE>(L, c, vBegin, min(L.vertexCount, vBegin + vertsPerSplit), pal);
    }
}

// The resident form: as many workgroups as are meant to stay on the chip, work units after the first drawn from a ticket counter
// (see skin_refit_kernel). The workgroups never leave, so no collision workgroup can take an LBS workgroup's place: beside the
// next step's collision kernels the launch keeps its own rate and the collision side gets what is left. sge_tick uses it for
// large crowds, where the shipped hand-over of places lets the big collision launches hold every place until they drain
// (DESIGN.md 3.5); SGE_SKIN_PERSISTENT=q forces q/4 workgroups per CU, 0 switches it off.
template <int SRC_STRIDE, int DST_STRIDE>
__global__ __launch_bounds__(kSkinBlock, SGE_SKIN_MIN_BLOCKS) void skin_ticket_kernel(SkinLaunch L, int splits, int vertsPerSplit, int* __restrict__ queue) {
    __shared__ float4 pal[SGE_MAX_BONES * 3];
    __shared__ int sNextUnit;
    for (int u = blockIdx.x; u < L.chars * splits;) {
        const int c = u / splits;
        const int sp = u - c * splits;
        const int vBegin = sp * vertsPerSplit;
        int ticket = 0;
        if (threadIdx.x == 0) ticket = atomicAdd(queue, 1);
        skinRange<SRC_STRIDE, DST_STRIDE>(L, c, vBegin, min(L.vertexCount, vBegin + vertsPerSplit), pal);
        if (threadIdx.x == 0) sNextUnit = (int)gridDim.x + ticket;
        __syncthreads(); // also: every thread is done with the palette
        u = sNextUnit; // the next write follows the next unit's barrier in skinRange
    }
    ticketsDone(queue);
}

// RTSkinningEncoder.encode over a heterogeneous job list (RTSkinningEncoder.swift:37-54 dispatches once per job): ONE launch.
// blockJob[b] = (job, first vertex) of workgroup b; every job keeps its own source streams, palette and destination offset.
template <int DST_STRIDE>
__global__ __launch_bounds__(kSkinBlock) void skin_jobs_kernel(const SkinJobDev* jobs, const int2* blockJob, int vertsPerBlock,
                                                               void* outPos, void* outNrm, void* outTan) {
    __shared__ float4 pal[SGE_MAX_BONES * 3];
    const int2 bj = blockJob[blockIdx.x];
    const SkinJobDev J = jobs[bj.x];
    SkinLaunch L{J.srcPos, J.srcNrm, J.srcTan, J.srcIdx, J.srcWgt, J.palette, J.paletteCount, J.vertexCount, 1, J.dstBaseVertex,
                 J.srcStride == 4 ? SGE_LAYOUT_PADDED16 : SGE_LAYOUT_PACKED, DST_STRIDE == 4 ? SGE_LAYOUT_PADDED16 : SGE_LAYOUT_PACKED,
                 outPos, outNrm, outTan};
    const int vEnd = min(J.vertexCount, bj.y + vertsPerBlock);
    if (J.srcStride == 4) skinRange<4, DST_STRIDE>(L, 0, bj.y, vEnd, pal);
    else skinRange<3, DST_STRIDE>(L, 0, bj.y, vEnd, pal);
}

void launch_skin_jobs(const SkinJobDev* d_jobs, const int2* d_blockJob, int blocks, int vertsPerBlock, int dstLayout,
                      void* outPos, void* outNrm, void* outTan, hipStream_t s) {
    if (blocks <= 0) return;
    if (dstLayout == SGE_LAYOUT_PADDED16) hipLaunchKernelGGL((skin_jobs_kernel<4>), dim3(blocks), dim3(kSkinBlock), 0, s, d_jobs, d_blockJob, vertsPerBlock, outPos, outNrm, outTan);
    else hipLaunchKernelGGL((skin_jobs_kernel<3>), dim3(blocks), dim3(kSkinBlock), 0, s, d_jobs, d_blockJob, vertsPerBlock, outPos, outNrm, outTan);
}

// The LBS kernel's store pattern alone (one workgroup per character, three non-temporal streams): sge_api uses it to
// compare candidate placements of the output buffers, see allocCrowdOutputs.
template <int DST_STRIDE>
__global__ __launch_bounds__(kSkinBlock) void store_probe_kernel(float* outPos, float* outNrm, v4f* outTan, int vertexCount) {
    const size_t base = (size_t)blockIdx.x * vertexCount;
    for (int v = threadIdx.x; v < vertexCount; v += kSkinBlock) {
        const size_t o = base + v;
        float* op = outPos + o * DST_STRIDE;
        float* on = outNrm + o * DST_STRIDE;
        if (DST_STRIDE == 4) {
            __builtin_nontemporal_store(v4f{0.f, 0.f, 0.f, 0.f}, reinterpret_cast<v4f*>(op));
            __builtin_nontemporal_store(v4f{0.f, 0.f, 0.f, 0.f}, reinterpret_cast<v4f*>(on));
        } else {
            __builtin_nontemporal_store(0.f, op); __builtin_nontemporal_store(0.f, op + 1); __builtin_nontemporal_store(0.f, op + 2);
            __builtin_nontemporal_store(0.f, on); __builtin_nontemporal_store(0.f, on + 1); __builtin_nontemporal_store(0.f, on + 2);
        }
        __builtin_nontemporal_store(v4f{0.f, 0.f, 0.f, 0.f}, outTan + o);
    }
}
void launch_store_probe(void* outPos, void* outNrm, void* outTan, int chars, int vertexCount, int dstLayout, hipStream_t s) {
    if (chars <= 0 || vertexCount <= 0) return;
    if (dstLayout == SGE_LAYOUT_PADDED16) hipLaunchKernelGGL((store_probe_kernel<4>), dim3(chars), dim3(kSkinBlock), 0, s, (float*)outPos, (float*)outNrm, (v4f*)outTan, vertexCount);
    else hipLaunchKernelGGL((store_probe_kernel<3>), dim3(chars), dim3(kSkinBlock), 0, s, (float*)outPos, (float*)outNrm, (v4f*)outTan, vertexCount);
}

// maxWorkgroupsPerCU > 0 caps the kernel's residency with dynamic-LDS padding: beside the next step's collision kernels
// (SGE_OPT_OVERLAP_SKIN) three workgroups per CU stream as fast as five do alone, and the rest of the register file goes to the
// latency-bound side (measured: 1.50 ms per step uncapped, 1.31 ms capped at three, 1.74 ms without overlap).
int launch_skin(const SkinLaunch& L, hipStream_t s, int maxWorkgroupsPerCU, int* residentQueue, int residentQuarters, int charsPerUnit) {
    if (L.chars <= 0 || L.vertexCount <= 0) return 0;
    // enough workgroups to fill 256 CUs x 8 resident blocks several times over; small crowds split characters
    int splits = 1;
    while ((long long)L.chars * splits < 8192 && splits < 64 && (L.vertexCount + splits - 1) / splits > 2 * kSkinBlock) splits *= 2;
    int vertsPerSplit = ((L.vertexCount + splits - 1) / splits + kSkinBlock - 1) / kSkinBlock * kSkinBlock;
    splits = (L.vertexCount + vertsPerSplit - 1) / vertsPerSplit;
    dim3 grid((unsigned)((size_t)splits * L.chars));
    int ss = L.srcLayout == SGE_LAYOUT_PADDED16 ? 4 : 3, ds = L.dstLayout == SGE_LAYOUT_PADDED16 ? 4 : 3;
    int ldsPad = 0;
    if (maxWorkgroupsPerCU > 0) {
        static const int totalOverride = getenv("SGE_SKIN_LDS_TOTAL") ? atoi(getenv("SGE_SKIN_LDS_TOTAL")) : 0; // experiments
        const int perWorkgroup = totalOverride > 0 ? totalOverride : 160 * 1024 / maxWorkgroupsPerCU, own = (int)sizeof(float4) * SGE_MAX_BONES * 3;
        ldsPad = perWorkgroup > own + 256 ? (perWorkgroup - own - 256) & ~255 : 0;
    }
    if (ldsPad + (int)sizeof(float4) * SGE_MAX_BONES * 3 > 64 * 1024) {
        // a cap of one or two workgroups per CU pads beyond the 64 KB a kernel may ask for by default: raise the limit once per device
        static bool attrSet[kMaxDevices] = {};
        const int devSlot = currentDeviceSlot();
        if (!attrSet[devSlot]) {
            const void* fns[4] = {reinterpret_cast<const void*>(skin_kernel<3, 3>), reinterpret_cast<const void*>(skin_kernel<3, 4>),
                                  reinterpret_cast<const void*>(skin_kernel<4, 3>), reinterpret_cast<const void*>(skin_kernel<4, 4>)};
            bool ok = true;
            for (const void* f : fns) ok &= hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)sizeof(float4) * SGE_MAX_BONES * 3) == hipSuccess;
            if (!ok) { (void)hipGetLastError(); ldsPad = 0; } // no cap rather than a launch that fails
            else attrSet[devSlot] = true;
        }
    }
    // the multi-character form stages charsPerUnit palettes in LDS (48 bytes per bone each): fewer characters per unit when a rig
    // with many bones would pass the 64 KB a kernel may ask for without an attribute (8 x 256 bones = 98 KB; 4 x 256 = 49 KB)
    int cpwSetting = charsPerUnit;
    while (cpwSetting >= 2 && (size_t)cpwSetting * L.paletteCount * 48 + 16 > (size_t)64 * 1024) cpwSetting /= 2;
    if (residentQueue && residentQuarters > 0 && L.srcLayout != SGE_LAYOUT_PADDED16 && (cpwSetting == 2 || cpwSetting == 4 || cpwSetting == 8) && L.chars >= 64) {
        const int cpw = cpwSetting, groups = (L.chars + cpw - 1) / cpw;
        int sp2 = 1;
        while ((long long)groups * sp2 < 8192 && sp2 < 64 && (L.vertexCount + sp2 - 1) / sp2 > 2 * kSkinBlock) sp2 *= 2;
        int vps = ((L.vertexCount + sp2 - 1) / sp2 + kSkinBlock - 1) / kSkinBlock * kSkinBlock;
        sp2 = (L.vertexCount + vps - 1) / vps;
        const size_t lds = (size_t)cpw * L.paletteCount * 48 + 16;
        dim3 pgrid((unsigned)std::min<size_t>((size_t)sp2 * groups, (size_t)currentDeviceCUs() * residentQuarters / 4));
        if (cpw == 2) {
            if (ds == 3) hipLaunchKernelGGL((skin_ticket_multi_kernel<3, 2>), pgrid, dim3(kSkinBlock), lds, s, L, sp2, vps, residentQueue);
            else hipLaunchKernelGGL((skin_ticket_multi_kernel<4, 2>), pgrid, dim3(kSkinBlock), lds, s, L, sp2, vps, residentQueue);
        } else if (cpw == 4) {
            if (ds == 3) hipLaunchKernelGGL((skin_ticket_multi_kernel<3, 4>), pgrid, dim3(kSkinBlock), lds, s, L, sp2, vps, residentQueue);
            else hipLaunchKernelGGL((skin_ticket_multi_kernel<4, 4>), pgrid, dim3(kSkinBlock), lds, s, L, sp2, vps, residentQueue);
        } else {
            if (ds == 3) hipLaunchKernelGGL((skin_ticket_multi_kernel<3, 8>), pgrid, dim3(kSkinBlock), lds, s, L, sp2, vps, residentQueue);
            else hipLaunchKernelGGL((skin_ticket_multi_kernel<4, 8>), pgrid, dim3(kSkinBlock), lds, s, L, sp2, vps, residentQueue);
        }
        return cpw;
    }
    if (residentQueue && residentQuarters > 0 && L.srcLayout != SGE_LAYOUT_PADDED16) { // resident workgroups + ticket counter (left at zero by the launch before)
        dim3 pgrid((unsigned)std::min<size_t>((size_t)splits * L.chars, (size_t)currentDeviceCUs() * residentQuarters / 4));
        if (ds == 3) hipLaunchKernelGGL((skin_ticket_kernel<3, 3>), pgrid, dim3(kSkinBlock), 0, s, L, splits, vertsPerSplit, residentQueue);
        else hipLaunchKernelGGL((skin_ticket_kernel<3, 4>), pgrid, dim3(kSkinBlock), 0, s, L, splits, vertsPerSplit, residentQueue);
        return 1;
    }
    static const int prio = getenv("SGE_SKIN_SETPRIO") ? atoi(getenv("SGE_SKIN_SETPRIO")) : 0;
    const int wp = maxWorkgroupsPerCU > 0 ? prio : 0;
    if (ss == 3 && ds == 3) hipLaunchKernelGGL((skin_kernel<3, 3>), grid, dim3(kSkinBlock), ldsPad, s, L, splits, vertsPerSplit, wp);
    else if (ss == 3 && ds == 4) hipLaunchKernelGGL((skin_kernel<3, 4>), grid, dim3(kSkinBlock), ldsPad, s, L, splits, vertsPerSplit, wp);
    else if (ss == 4 && ds == 3) hipLaunchKernelGGL((skin_kernel<4, 3>), grid, dim3(kSkinBlock), ldsPad, s, L, splits, vertsPerSplit, wp);
    else hipLaunchKernelGGL((skin_kernel<4, 4>), grid, dim3(kSkinBlock), ldsPad, s, L, splits, vertsPerSplit, wp);
    return 0;
}

// ---------------------------------------------------------------------------
// LBS + acceleration-structure refit in one kernel (SGE_OPT_FUSE_BLAS_REFIT): the positions never come back from HBM.
// Persistent workgroups of kBlasRefitBlock threads, each taking characters blockIdx.x, blockIdx.x + gridDim.x, ...; per tile of
// the refit schedule (sge_blas.hip, sge_blas_dev.hpp) the workgroup skins the tile's vertices — same arithmetic and the same
// three non-temporal output streams as skin_kernel — and keeps a copy of every position in the LDS tile; after a barrier
// every lane walks one chunk of the schedule exactly as blas_refit_kernel does. The boxes are therefore the min / max of
// the very floats that were stored.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void blendAndTransform(const float4* pal, const VertexIn& v, float3& acc, float3& nn, float3& tn) {
    const float w0 = fmaxf(v.w.x, 0.0f), w1 = fmaxf(v.w.y, 0.0f);
    Row3 M;
    {
        const Row3 m = loadRows(pal, v.idx.x);
        M.r0 = make_float4(m.r0.x * w0, m.r0.y * w0, m.r0.z * w0, m.r0.w * w0);
        M.r1 = make_float4(m.r1.x * w0, m.r1.y * w0, m.r1.z * w0, m.r1.w * w0);
        M.r2 = make_float4(m.r2.x * w0, m.r2.y * w0, m.r2.z * w0, m.r2.w * w0);
    }
#define SGE_BLEND(BONE, WGT)                                                  \
    {                                                                         \
        const Row3 m = loadRows(pal, (BONE));                                 \
        M.r0.x += m.r0.x * (WGT); M.r0.y += m.r0.y * (WGT); M.r0.z += m.r0.z * (WGT); M.r0.w += m.r0.w * (WGT); \
        M.r1.x += m.r1.x * (WGT); M.r1.y += m.r1.y * (WGT); M.r1.z += m.r1.z * (WGT); M.r1.w += m.r1.w * (WGT); \
        M.r2.x += m.r2.x * (WGT); M.r2.y += m.r2.y * (WGT); M.r2.z += m.r2.z * (WGT); M.r2.w += m.r2.w * (WGT); \
    }
    SGE_BLEND(v.idx.y, w1)
    if (v.w.z > 0.0f) SGE_BLEND(v.idx.z, v.w.z)
    if (v.w.w > 0.0f) SGE_BLEND(v.idx.w, v.w.w)
#undef SGE_BLEND
    acc = xform(M, v.p, 1.0f);
    nn = normalizeFast(xform(M, v.n, 0.0f));
    tn = normalizeFast(xform(M, make_float3(v.t.x, v.t.y, v.t.z), 0.0f));
}

template <int SRC_STRIDE, int DST_STRIDE, int TILE>
__global__ __launch_bounds__(kBlasRefitBlock) void skin_refit_kernel(SkinLaunch L, DevBlas B, float* __restrict__ bounds, int* __restrict__ queue) {
    extern __shared__ float lds[];
    const int rows = B.entryCount + 1, tid = threadIdx.x;
    float* tab = lds;
    float* X = lds + rows * 6; // Y = X + TILE, Z = X + 2 * TILE
    int* trs = reinterpret_cast<int*>(X + 3 * TILE);
    float4* pal = reinterpret_cast<float4*>(lds + (((rows * 6 + 3 * TILE + B.tileCount + 1) + 3) & ~3));
    blasTableInit(tab, rows, tid, kBlasRefitBlock);
    for (int i = tid; i <= B.tileCount; i += kBlasRefitBlock) trs[i] = B.tileRoundStart[i];
    __syncthreads();
    const int lane = tid & (kBlasWave - 1), wave = tid / kBlasWave;
    constexpr int kWaves = kBlasRefitBlock / kBlasWave, kPerThread = TILE / kBlasRefitBlock;
    const int n = B.tileCount, lastRound = trs[n] - 1;
    float* palf = reinterpret_cast<float*>(pal);
    // the ticket's slot lies behind the palette: a static __shared__ variable would shift the dynamic region to offset 4 and
    // with it every 16-byte palette read off its alignment (measured: the kernel takes twice as long)
    int& sNextChar = *reinterpret_cast<int*>(pal + L.paletteCount * 3);
    int* topo = reinterpret_cast<int*>(pal + L.paletteCount * 3 + 1); // the wide tree's shape, behind the ticket's 16-byte slot
    blasTopoStage(B, topo, tid, kBlasRefitBlock);                     // (made visible by the first tile's barrier)

    // Characters are handed out through a ticket counter (zeroed by the launcher): the first is blockIdx.x, every later one
    // gridDim.x + ticket. Beside the collision kernels of the next step (SGE_OPT_OVERLAP_SKIN) some workgroups find their
    // place on a CU late; with a fixed stride they would finish that much later than the rest.
    for (int c = blockIdx.x; c < L.chars;) {
        // stage the palette: thread -> one float4 COLUMN (coalesced), scattered into rows (as skinRange does)
        const float4* gp = reinterpret_cast<const float4*>(L.palettes + (size_t)c * L.paletteCount * 16);
        for (int i = tid; i < L.paletteCount * 4; i += kBlasRefitBlock) {
            const float4 col = gp[i];
            const int bone = i >> 2, cc = i & 3;
            palf[bone * 12 + 0 + cc] = col.x;
            palf[bone * 12 + 4 + cc] = col.y;
            palf[bone * 12 + 8 + cc] = col.z;
        }
        const size_t obase = (size_t)L.dstBaseVertex + (size_t)c * L.vertexCount;
        for (int tile = 0; tile < n; ++tile) {
            const int base = tile * B.tileVerts, nv = min(B.tileVerts, B.vertexCount - base);
            // two vertices in flight per thread, as in skinRange (more would cost the occupancy that hides the store latency)
            VertexIn cur = loadVertex<SRC_STRIDE>(L, base + min(tid, nv - 1));
            // this wavefront's round of the tile, requested before the skinning arithmetic so that it has arrived by the walk
            const int rEnd = trs[tile + 1];
            int r = trs[tile] + wave;
            BlasRound R;
            blasFetchRound(B, r, lastRound, lane, R);
            __syncthreads(); // palette staged (first tile); the previous tile's rounds have read X/Y/Z; the table is initialised
#pragma unroll
            for (int k = 0; k < kPerThread; ++k) {
                const int v = tid + k * kBlasRefitBlock;
                VertexIn nxt{};
                if (k + 1 < kPerThread) nxt = loadVertex<SRC_STRIDE>(L, base + min(v + kBlasRefitBlock, nv - 1));
                if (v < nv) {
                    float3 acc, nn, tn;
                    blendAndTransform(pal, cur, acc, nn, tn);
                    const size_t o = obase + base + v;
                    float* op = reinterpret_cast<float*>(L.outPos) + o * DST_STRIDE;
                    float* on = reinterpret_cast<float*>(L.outNrm) + o * DST_STRIDE;
                    if (DST_STRIDE == 4) {
                        __builtin_nontemporal_store(v4f{acc.x, acc.y, acc.z, 0.f}, reinterpret_cast<v4f*>(op));
                        __builtin_nontemporal_store(v4f{nn.x, nn.y, nn.z, 0.f}, reinterpret_cast<v4f*>(on));
                    } else {
                        __builtin_nontemporal_store(acc.x, op); __builtin_nontemporal_store(acc.y, op + 1); __builtin_nontemporal_store(acc.z, op + 2);
                        __builtin_nontemporal_store(nn.x, on); __builtin_nontemporal_store(nn.y, on + 1); __builtin_nontemporal_store(nn.z, on + 2);
                    }
                    __builtin_nontemporal_store(v4f{tn.x, tn.y, tn.z, cur.t.w}, reinterpret_cast<v4f*>(L.outTan) + o);
                    X[v] = acc.x; X[v + TILE] = acc.y; X[v + 2 * TILE] = acc.z;
                }
                cur = nxt;
            }
            __syncthreads();
            if (r < rEnd) blasWalk<TILE>(tab, rows, X, R); // wave-uniform
            for (r += kWaves; r < rEnd; r += kWaves) {
                blasFetchRound(B, r, lastRound, lane, R);
                blasWalk<TILE>(tab, rows, X, R);
            }
        }
        int ticket = 0;
        if (tid == 0) ticket = atomicAdd(queue, 1); // answered behind the reduction and the write-out below
        blasFinishCharacter(B, topo, tab, rows, tid, kBlasRefitBlock, bounds + (size_t)c * rows * 6);
        // the next character's palette staging writes `pal` only after every wavefront has left the last walk (barriers inside
        // blasFinishCharacter), and the barrier below orders the table re-initialisation before any fold
        if (tid == 0) sNextChar = (int)gridDim.x + ticket;
        __syncthreads();
        c = sNextChar; // next written after the barriers of the next character
    }
}

template <int TILE>
static int launchSkinRefitTile(const SkinLaunch& L, const DevBlas& B, float* bounds, int* queue, int perCU, size_t lds, hipStream_t s) {
    static bool attrSet[kMaxDevices] = {};
    const int devSlot = currentDeviceSlot();
    if (!attrSet[devSlot]) {
        SGE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(skin_refit_kernel<3, 3, TILE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SGE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(skin_refit_kernel<3, 4, TILE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attrSet[devSlot] = true;
    }
    // persistent: exactly the workgroups that are resident together. Registers bind before the LDS does (112 VGPRs: four
    // wavefronts per SIMD, two workgroups per CU); a workgroup beyond that would start when the others leave and do its first
    // character, fixed by its index, as the tail of the launch.
    static int residentCache[kMaxDevices][2] = {};
    static size_t residentLds[kMaxDevices][2] = {};
    const int form = L.dstLayout == SGE_LAYOUT_PADDED16 ? 1 : 0;
    if (!residentCache[devSlot][form] || residentLds[devSlot][form] != lds) { // once per (device, layout, LDS size), not per tick
        int r = 0;
        if (form) SGE_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&r, skin_refit_kernel<3, 4, TILE>, kBlasRefitBlock, lds));
        else SGE_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&r, skin_refit_kernel<3, 3, TILE>, kBlasRefitBlock, lds));
        residentCache[devSlot][form] = r > 0 ? r : 1;
        residentLds[devSlot][form] = lds;
    }
    const int resident = residentCache[devSlot][form];
    const int grid = std::min(L.chars, currentDeviceCUs() * std::max(1, std::min(perCU, resident)));
    if (L.dstLayout == SGE_LAYOUT_PADDED16) hipLaunchKernelGGL((skin_refit_kernel<3, 4, TILE>), dim3(grid), dim3(kBlasRefitBlock), lds, s, L, B, bounds, queue);
    else hipLaunchKernelGGL((skin_refit_kernel<3, 3, TILE>), dim3(grid), dim3(kBlasRefitBlock), lds, s, L, B, bounds, queue);
    return SGE_OK;
}

// the crowd's skin stage and refit stage as one launch; the source mesh is the context's (packed).
// maxWorkgroupsPerCU > 0 launches fewer persistent workgroups than the LDS would hold, which leaves the rest of it to the
// collision kernels of the next step (SGE_OPT_OVERLAP_SKIN): these workgroups never leave, so nothing else frees a place.
int launch_skin_refit(const SkinLaunch& L, const DevBlas& B, float* bounds, int* queue, hipStream_t s, int maxWorkgroupsPerCU) {
    if (L.chars <= 0 || L.vertexCount <= 0) return SGE_OK;
    SGE_HIP(hipMemsetAsync(queue, 0, sizeof(int), s));
    const size_t lds = blasRefitLdsBytes(B.entryCount, B.tileCount, B.tileCap) + 16 + (size_t)L.paletteCount * 48 + 16 + blasTopoBytes(B.wideCount, B.levels); // + the ticket slot + the tree's shape
    int perCU = (int)std::max<size_t>(1, std::min<size_t>(4, (size_t)(160 * 1024) / (lds + 256)));
    if (maxWorkgroupsPerCU > 0) perCU = std::min(perCU, maxWorkgroupsPerCU);
    switch (B.tileCap) {
    case 3072: return launchSkinRefitTile<3072>(L, B, bounds, queue, perCU, lds, s);
    case 2048: return launchSkinRefitTile<2048>(L, B, bounds, queue, perCU, lds, s);
    default: return launchSkinRefitTile<4096>(L, B, bounds, queue, perCU, lds, s);
    }
}

} // namespace sge
