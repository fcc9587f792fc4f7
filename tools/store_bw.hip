// Micro-benchmark: achievable HBM write bandwidth on gfx950 for the store shapes the LBS kernel can emit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void st4(float4* o, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) o[i] = make_float4(i, 1, 2, 3); }
__global__ void st3(float* o, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { float* p = o + i * 3; p[0] = i; p[1] = 1; p[2] = 2; } }
// three output streams like LBS: 12 + 12 + 16 B per element
__global__ void st334(float* a, float* b, float4* c, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) { float* p = a + i * 3; p[0] = i; p[1] = 1; p[2] = 2; float* q = b + i * 3; q[0] = i; q[1] = 3; q[2] = 4; c[i] = make_float4(i, 1, 2, 3); }
}
// same bytes, but float3 streams transposed through LDS so every lane stores 16 B
__global__ void st334_lds(float* a, float* b, float4* c, size_t n) {
    __shared__ float sa[256 * 3], sb[256 * 3];
    size_t base = blockIdx.x * (size_t)blockDim.x;
    size_t i = base + threadIdx.x;
    int t = threadIdx.x;
    sa[t * 3] = i; sa[t * 3 + 1] = 1; sa[t * 3 + 2] = 2;
    sb[t * 3] = i; sb[t * 3 + 1] = 3; sb[t * 3 + 2] = 4;
    __syncthreads();
    if (t < 192) {
        reinterpret_cast<float4*>(a + base * 3)[t] = reinterpret_cast<float4*>(sa)[t];
        reinterpret_cast<float4*>(b + base * 3)[t] = reinterpret_cast<float4*>(sb)[t];
    }
    if (i < n) c[i] = make_float4(i, 1, 2, 3);
}
__global__ void copy4(const float4* in, float4* o, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) o[i] = in[i]; }

int main() {
    size_t n = (size_t)140800000; // 10k chars x 14080 verts
    float *a, *b; float4 *c, *d;
    CK(hipMalloc(&a, n * 12)); CK(hipMalloc(&b, n * 12)); CK(hipMalloc(&c, n * 16)); CK(hipMalloc(&d, n * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    dim3 blk(256), grd((unsigned)((n + 255) / 256));
    auto run = [&](const char* name, double bytes, auto&& f) {
        for (int w = 0; w < 2; ++w) f();
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) f();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-28s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
    };
    run("float4 stores (16B/lane)", n * 16.0, [&] { hipLaunchKernelGGL(st4, grd, blk, 0, 0, c, n); });
    run("float3 stores (12B/lane)", n * 12.0, [&] { hipLaunchKernelGGL(st3, grd, blk, 0, 0, a, n); });
    run("3 streams 12+12+16", n * 40.0, [&] { hipLaunchKernelGGL(st334, grd, blk, 0, 0, a, b, c, n); });
    run("3 streams via LDS 16B/lane", n * 40.0, [&] { hipLaunchKernelGGL(st334_lds, grd, blk, 0, 0, a, b, c, n); });
    run("float4 copy (rd+wr)", n * 32.0, [&] { hipLaunchKernelGGL(copy4, grd, blk, 0, 0, c, d, n); });
    return 0;
}
