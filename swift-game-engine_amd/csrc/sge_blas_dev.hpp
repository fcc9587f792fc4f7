// Device-side pieces of the acceleration-structure refit shared by blas_refit_kernel (sge_blas.hip: reads the skinned
// positions back) and skin_refit_kernel (sge_skin.hip: the LBS kernel keeps the positions it has just computed).
// Both keep, per workgroup: tab[row * 6 + c] (c = 0..2 minima, 3..5 maxima: the layout of the output, so that the write-out is a
// straight copy) and one tile of positions as
// X[], Y[] = X + TILE, Z[] = X + 2 * TILE in LDS.
#pragma once
#include "sge_internal.hpp"

namespace sge {

constexpr int kBlasWave = 64;

__device__ __forceinline__ float blasWaveMin(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, kBlasWave)); return v; }
__device__ __forceinline__ float blasWaveMax(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kBlasWave)); return v; }

// Wave-wide min / max on the VALU's data-parallel path (DPP): two quad permutes, the two row mirrors, then lane 15 of every row of
// 16 into the next row and lane 31 into the upper half — lane 63 ends up with the reduction of all 64. __shfl_xor goes through
// ds_bpermute_b32, i.e. through the LDS unit, which the other workgroups' walks keep busy: six dependent rounds of it cost the
// end-of-character reduction ~3,000 cycles per level (profiles/r4_refit_phases.txt).
template <bool MIN>
__device__ __forceinline__ float blasWaveReduce(float x) {
#define SGE_DPP_STEP(ctrl, rows_)                                                                                              \
    {                                                                                                                             \
        const float y_ = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x), ctrl, rows_, 0xf, false)); \
        x = MIN ? fminf(x, y_) : fmaxf(x, y_);                                                                                    \
    }
    SGE_DPP_STEP(0xB1, 0xf)  // quad_perm [1,0,3,2]
    SGE_DPP_STEP(0x4E, 0xf)  // quad_perm [2,3,0,1]
    SGE_DPP_STEP(0x141, 0xf) // row_half_mirror
    SGE_DPP_STEP(0x140, 0xf) // row_mirror: every lane of a row holds the row's result
    SGE_DPP_STEP(0x142, 0xa) // row_bcast15 into rows 1 and 3
    SGE_DPP_STEP(0x143, 0xc) // row_bcast31 into rows 2 and 3
#undef SGE_DPP_STEP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

// v_min_f32 / v_max_f32 as written (fminf() on a value loaded from LDS would first canonicalise it: one more instruction each)
#define SGE_BLAS_FOLD(X_, Y_, Z_)                                              \
    asm("v_min_f32 %0, %0, %1" : "+v"(mnx) : "v"(X_)); asm("v_max_f32 %0, %0, %1" : "+v"(mxx) : "v"(X_)); \
    asm("v_min_f32 %0, %0, %1" : "+v"(mny) : "v"(Y_)); asm("v_max_f32 %0, %0, %1" : "+v"(mxy) : "v"(Y_)); \
    asm("v_min_f32 %0, %0, %1" : "+v"(mnz) : "v"(Z_)); asm("v_max_f32 %0, %0, %1" : "+v"(mxz) : "v"(Z_));

// a lane's chunk result into its cluster's row of the table
__device__ __forceinline__ void blasFold(float* tab, int cluster, float mnx, float mny, float mnz, float mxx, float mxy, float mxz) {
    float* t = tab + cluster * 6;
    __hip_atomic_fetch_min(t, mnx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_min(t + 1, mny, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_min(t + 2, mnz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_max(t + 3, mxx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_max(t + 4, mxy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_max(t + 5, mxz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// one wavefront's share of a tile's schedule: every lane's chunk (<= 16 LDS byte offsets, two per word), its cluster, the
// round's step count
struct BlasRound { uint32_t w[8]; int cluster, len; };

__device__ __forceinline__ void blasTableInit(float* tab, int rows, int tid, int threads) {
    const float inf = __builtin_inff();
    for (int i = tid; i < rows * 6; i += threads) tab[i] = (i % 6) < 3 ? inf : -inf;
}

// three loads: the lane's eight schedule words as two 16-byte loads (a wavefront's instruction covers 2 KB of consecutive
// addresses), and its cluster with the round's length in the top byte
__device__ __forceinline__ void blasFetchRound(const DevBlas& B, int r, int lastRound, int lane, BlasRound& R) {
    const int rr = __builtin_amdgcn_readfirstlane(min(r, lastRound)); // past the end: a valid round, loaded but not used
    const uint4* p = reinterpret_cast<const uint4*>(B.roundIds) + ((size_t)rr * 64 + lane) * 2;
    const uint4 a = p[0], b = p[1];
    R.w[0] = a.x; R.w[1] = a.y; R.w[2] = a.z; R.w[3] = a.w; R.w[4] = b.x; R.w[5] = b.y; R.w[6] = b.z; R.w[7] = b.w;
    const int cl = B.roundCluster[(size_t)rr * 64 + lane];
    R.cluster = cl & 0xffffff;
    R.len = cl >> 24; // wave-uniform (the walk takes it through readfirstlane)
}

// The shape of the wide tree in LDS: topo[0 .. levels] = wideLevelStart, then wideFirst[wideCount + 1], then wideParentEntry[wideCount]
__device__ __forceinline__ void blasTopoStage(const DevBlas& B, int* topo, int tid, int threads) {
    const int L = B.levels + 1, W = B.wideCount;
    for (int i = tid; i < L + 2 * W + 1; i += threads)
        topo[i] = i < L ? B.wideLevelStart[i] : (i < L + W + 1 ? B.wideFirst[i - L] : B.wideParentEntry[i - L - W - 1]);
}

// the lane walks its chunk from LDS with the running min / max in registers and folds them into its cluster's row.
// Four vertices at a time: their twelve LDS reads are issued together and folded together, so a round of 16 pays four LDS round
// trips, not sixteen (one read-wait-fold per vertex behind a per-lane `i < len` test was 0.30 ms of the 0.38 ms kernel with the
// position loads taken out: profiles/r4_refit_experiments.txt). The round's length is wave-uniform and lives in a scalar register;
// schedule entries past it repeat the lane's own first vertex (HostBlas::build), so a group may run over the end.
template <int TILE>
__device__ __forceinline__ void blasWalk(float* tab, int rows, const float* X, const BlasRound& R) {
    const float inf = __builtin_inff();
    float mnx = inf, mny = inf, mnz = inf, mxx = -inf, mxy = -inf, mxz = -inf;
    const char* Xb = reinterpret_cast<const char*>(X);
    const int len = __builtin_amdgcn_readfirstlane(R.len);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g * 4 < len) { // scalar branch
            float x[4], y[4], z[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = g * 4 + j;
                const uint32_t off = (i & 1) ? (R.w[i >> 1] >> 16) : (R.w[i >> 1] & 0xffffu);
                const float* q = reinterpret_cast<const float*>(Xb + off);
                x[j] = q[0]; y[j] = q[TILE]; z[j] = q[2 * TILE];
            }
            __builtin_amdgcn_sched_barrier(0); // all twelve reads are issued before the first fold waits (hipcc otherwise re-serialises them to save registers)
#pragma unroll
            for (int j = 0; j < 4; ++j) { SGE_BLAS_FOLD(x[j], y[j], z[j]) }
        }
    }
    blasFold(tab, R.cluster, mnx, mny, mnz, mxx, mxy, mxz);
}

// End of a character, called by every thread of the workgroup: inner entries from their wide nodes, deepest level first, one
// wavefront per wide node; the table goes out coalesced ([entries + 1][6]) and is re-initialised. Leaves a barrier pending:
// the caller's next __syncthreads() orders the re-initialisation before the next character's folds.
// Nothing here loads from global memory (`topo`: blasTopoStage): the position loads of the next tile are in flight at this point and
// loads return in order, so every global load in this function used to wait for a whole tile from HBM. The six components of a wide
// node are reduced side by side (six independent butterfly chains) rather than one after the other.
#ifndef SGE_FINISH_STAMP
#define SGE_FINISH_STAMP(k) do { } while (0)
#define SGE_FINISH_STAMP_BEGIN() do { } while (0)
#endif
__device__ __forceinline__ void blasFinishCharacter(const DevBlas& B, const int* topo, float* tab, int rows, int tid, int threads, float* out) {
    const float inf = __builtin_inff();
    SGE_FINISH_STAMP_BEGIN();
    const int lane = tid & (kBlasWave - 1), wave = tid / kBlasWave, waves = threads / kBlasWave;
    const int* levelStart = topo;
    const int* wideFirst = topo + B.levels + 1;
    const int* wideParent = wideFirst + B.wideCount + 1;
    for (int lvl = B.levels - 1; lvl >= 0; --lvl) {
        __syncthreads();
        SGE_FINISH_STAMP(lvl == B.levels - 1 ? 0 : 2); // 0: the first barrier (the slowest wavefront's walk), 2: a later level's barrier
        for (int w = levelStart[lvl] + wave; w < levelStart[lvl + 1]; w += waves) {
            const int first = wideFirst[w], cnt = wideFirst[w + 1] - first;
            const int parent = wideParent[w];
            const int dst = parent < 0 ? B.entryCount : parent;
            float k[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) k[q] = lane < cnt ? tab[(first + lane) * 6 + q] : (q < 3 ? inf : -inf);
            k[0] = blasWaveReduce<true>(k[0]); k[1] = blasWaveReduce<true>(k[1]); k[2] = blasWaveReduce<true>(k[2]);
            k[3] = blasWaveReduce<false>(k[3]); k[4] = blasWaveReduce<false>(k[4]); k[5] = blasWaveReduce<false>(k[5]);
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 6; ++q) tab[dst * 6 + q] = k[q];
            }
        }
        SGE_FINISH_STAMP(1); // a level's reductions
    }
    __syncthreads();
    SGE_FINISH_STAMP(2);
    for (int i = tid; i < rows * 6; i += threads) { // every table element is read by exactly one thread, which re-initialises it
        out[i] = tab[i];
        tab[i] = (i % 6) < 3 ? inf : -inf;
    }
    SGE_FINISH_STAMP(3); // write-out + re-initialisation
}

} // namespace sge
