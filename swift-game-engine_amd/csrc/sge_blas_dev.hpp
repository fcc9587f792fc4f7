// Device-side pieces of the acceleration-structure refit shared by blas_refit_kernel (sge_blas.hip: reads the skinned
// positions back) and skin_refit_kernel (sge_skin.hip: the LBS kernel keeps the positions it has just computed).
// Both keep, per workgroup: tab[c * rows + row] (c = 0..2 minima, 3..5 maxima) and one tile of positions as
// X[], Y[] = X + TILE, Z[] = X + 2 * TILE in LDS.
#pragma once
#include "sge_internal.hpp"

namespace sge {

constexpr int kBlasWave = 64;

__device__ __forceinline__ float blasWaveMin(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, kBlasWave)); return v; }
__device__ __forceinline__ float blasWaveMax(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kBlasWave)); return v; }

// v_min_f32 / v_max_f32 as written (fminf() on a value loaded from LDS would first canonicalise it: one more instruction each)
#define SGE_BLAS_FOLD(X_, Y_, Z_)                                              \
    asm("v_min_f32 %0, %0, %1" : "+v"(mnx) : "v"(X_)); asm("v_max_f32 %0, %0, %1" : "+v"(mxx) : "v"(X_)); \
    asm("v_min_f32 %0, %0, %1" : "+v"(mny) : "v"(Y_)); asm("v_max_f32 %0, %0, %1" : "+v"(mxy) : "v"(Y_)); \
    asm("v_min_f32 %0, %0, %1" : "+v"(mnz) : "v"(Z_)); asm("v_max_f32 %0, %0, %1" : "+v"(mxz) : "v"(Z_));

// one wavefront's share of a tile's schedule: every lane's chunk (<= 16 LDS byte offsets, two per word), its cluster, the
// round's step count
struct BlasRound { uint32_t w[8]; int cluster, len; };

__device__ __forceinline__ void blasTableInit(float* tab, int rows, int tid, int threads) {
    const float inf = __builtin_inff();
    for (int i = tid; i < rows * 6; i += threads) tab[i] = i < rows * 3 ? inf : -inf;
}

__device__ __forceinline__ void blasFetchRound(const DevBlas& B, int r, int lastRound, int lane, BlasRound& R) {
    const int rr = __builtin_amdgcn_readfirstlane(min(r, lastRound)); // past the end: a valid round, loaded but not used
#pragma unroll
    for (int j = 0; j < 8; ++j) R.w[j] = B.roundIds[((size_t)rr * 8 + j) * 64 + lane];
    R.cluster = B.roundCluster[(size_t)rr * 64 + lane];
    R.len = B.roundLen[rr];
}

// the lane walks its chunk from LDS with the running min / max in registers and folds them into its cluster's row
template <int TILE>
__device__ __forceinline__ void blasWalk(float* tab, int rows, const float* X, const BlasRound& R) {
    const float inf = __builtin_inff();
    float mnx = inf, mny = inf, mnz = inf, mxx = -inf, mxy = -inf, mxz = -inf;
    const char* Xb = reinterpret_cast<const char*>(X);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i < R.len) { // wave-uniform
            const uint32_t off = (i & 1) ? (R.w[i >> 1] >> 16) : (R.w[i >> 1] & 0xffffu);
            const float* q = reinterpret_cast<const float*>(Xb + off);
            const float x = q[0], y = q[TILE], z = q[2 * TILE];
            SGE_BLAS_FOLD(x, y, z)
        }
    }
    float* t = tab + R.cluster;
    __hip_atomic_fetch_min(t, mnx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_min(t + rows, mny, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_min(t + 2 * rows, mnz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_max(t + 3 * rows, mxx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_max(t + 4 * rows, mxy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_max(t + 5 * rows, mxz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The same walk over a tile kept in LDS as it lies in memory (xyz xyz ..., 12 bytes per vertex, the raw-granule refit kernel):
// `tile` = LDS address of the tile's first vertex; the schedule's byte offsets (4 * local vertex id) times three.
__device__ __forceinline__ void blasWalkAoS(float* tab, int rows, const char* tile, const BlasRound& R) {
    const float inf = __builtin_inff();
    float mnx = inf, mny = inf, mnz = inf, mxx = -inf, mxy = -inf, mxz = -inf;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i < R.len) { // wave-uniform
            const uint32_t off = (i & 1) ? (R.w[i >> 1] >> 16) : (R.w[i >> 1] & 0xffffu);
            const float* q = reinterpret_cast<const float*>(tile + off * 3u);
            const float x = q[0], y = q[1], z = q[2];
            SGE_BLAS_FOLD(x, y, z)
        }
    }
    float* t = tab + R.cluster;
    __hip_atomic_fetch_min(t, mnx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_min(t + rows, mny, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_min(t + 2 * rows, mnz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_max(t + 3 * rows, mxx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_max(t + 4 * rows, mxy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_max(t + 5 * rows, mxz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// End of a character, called by every thread of the workgroup: inner entries from their wide nodes, deepest level first, one
// wavefront per wide node; the table goes out coalesced ([entries + 1][6]) and is re-initialised. Leaves a barrier pending:
// the caller's next __syncthreads() orders the re-initialisation before the next character's folds.
__device__ __forceinline__ void blasFinishCharacter(const DevBlas& B, float* tab, int rows, int tid, int threads, float* out) {
    const float inf = __builtin_inff();
    const int lane = tid & (kBlasWave - 1), wave = tid / kBlasWave, waves = threads / kBlasWave;
    for (int lvl = B.levels - 1; lvl >= 0; --lvl) {
        __syncthreads();
        for (int w = B.wideLevelStart[lvl] + wave; w < B.wideLevelStart[lvl + 1]; w += waves) {
            const int first = B.wideFirst[w], cnt = B.wideFirst[w + 1] - first;
            const int parent = B.wideParentEntry[w];
            const int dst = parent < 0 ? B.entryCount : parent;
            for (int q = 0; q < 6; ++q) {
                float k = lane < cnt ? tab[q * rows + first + lane] : (q < 3 ? inf : -inf);
                k = q < 3 ? blasWaveMin(k) : blasWaveMax(k);
                if (lane == 0) tab[q * rows + dst] = k;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < rows * 6; i += threads) { // every table element is read by exactly one thread, which re-initialises it
        const int row = i / 6, q = i - row * 6;
        out[i] = tab[q * rows + row];
        tab[q * rows + row] = q < 3 ? inf : -inf;
    }
}

} // namespace sge
