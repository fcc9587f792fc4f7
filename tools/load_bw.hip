// Diagnostic: what read bandwidth do the access shapes of the refit kernel reach? (10k "characters" of V packed float3 each)
//   hipcc --offload-arch=gfx950 -O3 tools/load_bw.hip -o tools/load_bw && ./tools/load_bw [V]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
// A: flat grid-stride float4 stream
__global__ void streamA(const v4f* __restrict__ p, size_t n, float* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    v4f acc{0, 0, 0, 0};
    for (; i < n; i += stride) acc += p[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = 1;
}
// B: one workgroup per character, float4, 256 threads, many workgroups per CU
__global__ void perCharB(const v4f* __restrict__ p, int vec4PerChar, float* sink) {
    const v4f* q = p + (size_t)blockIdx.x * vec4PerChar;
    v4f acc{0, 0, 0, 0};
    for (int i = threadIdx.x; i < vec4PerChar; i += blockDim.x) acc += q[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = 1;
}
// C/D: the refit kernel's shape: 512 threads, `ldsBytes` of LDS (limits workgroups per CU), tiles of 4096 vertices with two barriers,
// loads as float3 at 12-B stride (D) or float4 (C), DEPTH tiles in flight
template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void tiled(const float* __restrict__ p, int V, float* sink, int rotate) {
    extern __shared__ float lds[];
    const float* P = p + (size_t)blockIdx.x * V * 3;
    const int tiles = (V + 4095) / 4096, tid = threadIdx.x;
    float acc = 0;
    float r[DEPTH][24];
    auto fetch = [&](int t, int slot) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int v = min(t * 4096 + tid + k * 512, V - 1);
                r[slot][k * 3] = P[(size_t)v * 3]; r[slot][k * 3 + 1] = P[(size_t)v * 3 + 1]; r[slot][k * 3 + 2] = P[(size_t)v * 3 + 2];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                int i = min(t * 3072 + tid + k * 512, V * 3 / 4 - 1);
                v4f q = reinterpret_cast<const v4f*>(P)[i];
                r[slot][k * 4] = q.x; r[slot][k * 4 + 1] = q.y; r[slot][k * 4 + 2] = q.z; r[slot][k * 4 + 3] = q.w;
            }
        }
    };
    const int t0 = rotate ? blockIdx.x % tiles : 0;
    auto at = [&](int k) { int t = t0 + k; return t >= tiles ? t - tiles : t; };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) if (d < tiles) fetch(at(d), d);
    for (int s = 0; s < tiles; ++s) {
        __syncthreads();
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (s % DEPTH == d) {
#pragma unroll
                for (int k = 0; k < 24; ++k) lds[tid + k * 512] = r[d][k];
            }
        __syncthreads();
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (s % DEPTH == d && s + DEPTH < tiles) fetch(at(s + DEPTH), d);
        acc += lds[(tid * 7) & 4095];
    }
    if (acc == 12345.678f) *sink = 1;
}
int main(int argc, char** argv) {
    const int chars = 10000, V = argc > 1 ? atoi(argv[1]) : 35440;
    const size_t bytes = (size_t)chars * V * 12;
    float *p, *sink;
    hipMalloc(&p, bytes); hipMalloc(&sink, 4);
    hipMemset(p, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch) {
        launch();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-46s %.3f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6);
    };
    printf("V = %d, %.2f GB\n", V, bytes / 1e9);
    timeit("A flat float4 stream, 8192 x 256", [&] { streamA<<<8192, 256>>>((const v4f*)p, bytes / 16, sink); });
    timeit("B per-character float4, 256 thr", [&] { perCharB<<<chars, 256>>>((const v4f*)p, V * 3 / 4, sink); });
    timeit("B per-character float4, 512 thr", [&] { perCharB<<<chars, 512>>>((const v4f*)p, V * 3 / 4, sink); });
    const int ldsBig = 69 * 1024, ldsSmall = 48 * 1024;
    hipFuncSetAttribute((const void*)tiled<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBig);
    hipFuncSetAttribute((const void*)tiled<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBig);
    hipFuncSetAttribute((const void*)tiled<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBig);
    hipFuncSetAttribute((const void*)tiled<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBig);
    timeit("D tiled float3, depth 1, 69 KB LDS (2 WG/CU)", [&] { tiled<0, 1><<<chars, 512, ldsBig>>>(p, V, sink, 0); });
    timeit("D tiled float3, depth 1, 69 KB, rotated", [&] { tiled<0, 1><<<chars, 512, ldsBig>>>(p, V, sink, 1); });
    timeit("D tiled float3, depth 1, 48 KB LDS (3 WG/CU)", [&] { tiled<0, 1><<<chars, 512, ldsSmall>>>(p, V, sink, 0); });
    timeit("D tiled float3, depth 2, 69 KB LDS", [&] { tiled<0, 2><<<chars, 512, ldsBig>>>(p, V, sink, 0); });
    timeit("D tiled float3, depth 2, 69 KB, rotated", [&] { tiled<0, 2><<<chars, 512, ldsBig>>>(p, V, sink, 1); });
    timeit("C tiled float4, depth 1, 69 KB LDS", [&] { tiled<1, 1><<<chars, 512, ldsBig>>>(p, V, sink, 0); });
    timeit("C tiled float4, depth 2, 69 KB LDS", [&] { tiled<1, 2><<<chars, 512, ldsBig>>>(p, V, sink, 0); });
    timeit("C tiled float4, depth 2, 69 KB, rotated", [&] { tiled<1, 2><<<chars, 512, ldsBig>>>(p, V, sink, 1); });
    return 0;
}
