// Interference probe (GPU box): co-runner kernels that each load ONE resource of a compute unit, launched on their own stream beside
// the product's LBS kernel by tools/interfere.py. Answers "what stretches skin_kernel when the collision kernels run beside it".
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/interfere.hip -o tools/libinterfere.so   (not on the product path)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

namespace {

hipStream_t gStream = nullptr;
uint32_t* gChase = nullptr; // 4 MiB pointer-chase ring (L2 resident)
float* gBig = nullptr;      // 1 GiB streaming buffer
constexpr size_t kChaseWords = 1u << 20;
constexpr size_t kBigFloats = 1u << 28;

// kind 0: one dependent FMA chain per lane (what a latency-bound march looks like to the issue logic)
__global__ void valu_chain(long iters, float* sink) {
    float x = threadIdx.x * 1e-3f, a = 1.0000001f, b = 1e-7f;
    for (long i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) x = __builtin_fmaf(x, a, b);
    }
    if (x == 123.456f) sink[0] = x;
}

// kind 1: eight independent chains per lane (a wave that can issue every cycle it is given)
__global__ void valu_wide(long iters, float* sink) {
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = threadIdx.x * 1e-3f + k;
    for (long i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = __builtin_fmaf(x[k], 1.0000001f, 1e-7f);
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += x[k];
    if (s == 123.456f) sink[0] = s;
}

// kind 2: LDS traffic only
__global__ void lds_spin(long iters, float* sink) {
    __shared__ float buf[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = i;
    __syncthreads();
    float s = 0;
    int at = threadIdx.x;
    for (long i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            s += buf[at];
            at = (at + 65) & 4095;
        }
    }
    if (s == 123.456f) sink[0] = s;
}

// kind 3: dependent loads from an L2-resident ring (what a BVH walk looks like to the memory path)
__global__ void l2_chase(long iters, const uint32_t* ring, float* sink) {
    uint32_t at = (blockIdx.x * 64u + threadIdx.x) * 977u & (kChaseWords - 1);
    for (long i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) at = ring[at];
    }
    if (at == 0xFFFFFFFFu) sink[0] = 1.0f;
}

// kind 4: scalar ALU only
__global__ void salu_spin(long iters, float* sink) {
    uint32_t x = blockIdx.x;
    for (long i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) x = x * 1664525u + 1013904223u;
    }
    if (x == 0x12345u) sink[0] = 1.0f;
}

// kind 5: HBM streaming reads (a competitor for the memory system itself)
__global__ void hbm_read(long iters, const float4* big, float* sink) {
    float s = 0;
    size_t n = kBigFloats / 4;
    size_t at = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (long i = 0; i < iters; ++i) {
        float4 v = big[at % n];
        s += v.x + v.y + v.z + v.w;
        at += (size_t)gridDim.x * blockDim.x;
    }
    if (s == 123.456f) sink[0] = s;
}

// kind 6: the dependent chain with a register footprint that keeps other waves off the SIMD (launch bounds 64 x 3 like the move kernel)
__global__ void __launch_bounds__(64, 3) valu_chain_fat(long iters, float* sink) {
    float x[96];
#pragma unroll
    for (int k = 0; k < 96; ++k) x[k] = threadIdx.x + k;
    for (long i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 96; ++k) x[k] = __builtin_fmaf(x[(k + 95) % 96], 1.0000001f, x[k]);
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 96; ++k) s += x[k];
    if (s == 123.456f) sink[0] = s;
}

__global__ void init_ring(uint32_t* ring) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < kChaseWords) ring[i] = (i * 2654435761u + 12345u) & (kChaseWords - 1);
}

} // namespace

extern "C" int itf_init() {
    if (gStream) return 0;
    if (hipStreamCreateWithFlags(&gStream, hipStreamNonBlocking) != hipSuccess) return 1;
    if (hipMalloc(&gChase, kChaseWords * 4) != hipSuccess) return 2;
    if (hipMalloc(&gBig, kBigFloats * 4) != hipSuccess) return 3;
    hipMemsetAsync(gBig, 0, kBigFloats * 4, gStream);
    init_ring<<<kChaseWords / 256, 256, 0, gStream>>>(gChase);
    return hipStreamSynchronize(gStream) == hipSuccess ? 0 : 4;
}

// launches one co-runner; returns immediately (the kernel runs on the probe's own stream)
extern "C" int itf_launch(int kind, int workgroups, int threads, long iters) {
    float* sink = gBig;
    switch (kind) {
    case 0: valu_chain<<<workgroups, threads, 0, gStream>>>(iters, sink); break;
    case 1: valu_wide<<<workgroups, threads, 0, gStream>>>(iters, sink); break;
    case 2: lds_spin<<<workgroups, threads, 0, gStream>>>(iters, sink); break;
    case 3: l2_chase<<<workgroups, threads, 0, gStream>>>(iters, gChase, sink); break;
    case 4: salu_spin<<<workgroups, threads, 0, gStream>>>(iters, sink); break;
    case 5: hbm_read<<<workgroups, threads, 0, gStream>>>(iters, (const float4*)gBig, sink); break;
    case 6: valu_chain_fat<<<workgroups, 64, 0, gStream>>>(iters, sink); break;
    default: return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

// milliseconds one launch of the co-runner takes by itself (to size `iters`)
extern "C" float itf_time(int kind, int workgroups, int threads, long iters) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipEventRecord(a, gStream);
    itf_launch(kind, workgroups, threads, iters);
    hipEventRecord(b, gStream);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a);
    hipEventDestroy(b);
    return ms;
}

extern "C" int itf_sync() { return hipStreamSynchronize(gStream) == hipSuccess ? 0 : 1; }
