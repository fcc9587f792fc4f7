// Diagnostic: does streaming-store bandwidth depend on which hipMalloc'ed buffer is written?
//   hipcc --offload-arch=gfx950 -O3 tools/alloc_bw.hip -o tools/alloc_bw && ./tools/alloc_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void fill(v4f* __restrict__ p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) __builtin_nontemporal_store(v4f{1.f, 2.f, 3.f, 4.f}, p + i);
}
// one workgroup per "character": writes a contiguous 40-B-per-vertex slice of three streams, like the LBS kernel
__global__ void three(float* __restrict__ a, float* __restrict__ b, v4f* __restrict__ c, int verts) {
    size_t base = (size_t)blockIdx.x * verts;
    for (int v = threadIdx.x; v < verts; v += blockDim.x) {
        size_t o = base + v;
        __builtin_nontemporal_store(1.f, a + o * 3); __builtin_nontemporal_store(2.f, a + o * 3 + 1); __builtin_nontemporal_store(3.f, a + o * 3 + 2);
        __builtin_nontemporal_store(1.f, b + o * 3); __builtin_nontemporal_store(2.f, b + o * 3 + 1); __builtin_nontemporal_store(3.f, b + o * 3 + 2);
        __builtin_nontemporal_store(v4f{1.f, 2.f, 3.f, 4.f}, c + o);
    }
}
int main() {
    const int chars = 10000, verts = 14080;
    const size_t nv = (size_t)chars * verts;
    const int K = 6;
    std::vector<void*> A(K), B(K), Cc(K);
    for (int k = 0; k < K; ++k) { hipMalloc(&A[k], nv * 12); hipMalloc(&B[k], nv * 12); hipMalloc(&Cc[k], nv * 16); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
    for (int k = 0; k < K; ++k) {
        float ms[4];
        void* bufs[3] = {A[k], B[k], Cc[k]};
        size_t bytes[3] = {nv * 12, nv * 12, nv * 16};
        for (int s = 0; s < 3; ++s) {
            fill<<<8192, 256>>>((v4f*)bufs[s], bytes[s] / 16);
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) fill<<<8192, 256>>>((v4f*)bufs[s], bytes[s] / 16);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[s], e0, e1); ms[s] /= 5;
        }
        three<<<chars, 256>>>((float*)A[k], (float*)B[k], (v4f*)Cc[k], verts);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) three<<<chars, 256>>>((float*)A[k], (float*)B[k], (v4f*)Cc[k], verts);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[3], e0, e1); ms[3] /= 5;
        printf("set %d: fill pos %.0f GB/s  nrm %.0f GB/s  tan %.0f GB/s | three-stream %.3f ms = %.0f GB/s   (%p %p %p)\n", k,
               bytes[0] / ms[0] / 1e6, bytes[1] / ms[1] / 1e6, bytes[2] / ms[2] / 1e6, ms[3], nv * 40.0 / ms[3] / 1e6, A[k], B[k], Cc[k]);
    }
    return 0;
}
