// Diagnostic: the LBS three-stream store pattern with the positions and normals streams fixed and the TANGENT stream
// placed at successive offsets inside one large arena — tools/alloc_probe.hip showed that the tangent buffer alone decides whether a
// candidate placement is fast (0.81 ms) or slow (1.0-1.1 ms), with a period of several buffer sets.
//   hipcc --offload-arch=gfx950 -O3 tools/tan_scan.hip -o tools/tan_scan && ./tools/tan_scan [arena GiB] [step MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void three(float* __restrict__ a, float* __restrict__ b, v4f* __restrict__ c, int verts) {
    size_t base = (size_t)blockIdx.x * verts;
    for (int v = threadIdx.x; v < verts; v += blockDim.x) {
        size_t o = base + v;
        __builtin_nontemporal_store(1.f, a + o * 3); __builtin_nontemporal_store(2.f, a + o * 3 + 1); __builtin_nontemporal_store(3.f, a + o * 3 + 2);
        __builtin_nontemporal_store(1.f, b + o * 3); __builtin_nontemporal_store(2.f, b + o * 3 + 1); __builtin_nontemporal_store(3.f, b + o * 3 + 2);
        __builtin_nontemporal_store(v4f{1.f, 2.f, 3.f, 4.f}, c + o);
    }
}
int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t arenaGiB = argc > 1 ? (size_t)atoi(argv[1]) : 64, stepMiB = argc > 2 ? (size_t)atoi(argv[2]) : 512;
    const int chars = 10000, verts = 14080;
    const size_t nv = (size_t)chars * verts;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    void *A, *B;
    char* arena;
    if (hipMalloc(&A, nv * 12) != hipSuccess || hipMalloc(&B, nv * 12) != hipSuccess || hipMalloc((void**)&arena, arenaGiB << 30) != hipSuccess) { printf("allocation failed\n"); return 1; }
    printf("pos %p nrm %p arena %p (%zu GiB), tangent stream every %zu MiB\n", A, B, (void*)arena, arenaGiB, stepMiB);
    auto T = [&](char* c) {
        three<<<chars, 256>>>((float*)A, (float*)B, (v4f*)c, verts);
        (void)hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) three<<<chars, 256>>>((float*)A, (float*)B, (v4f*)c, verts);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        return ms / 3;
    };
    const size_t span = (arenaGiB << 30) - nv * 16;
    for (size_t off = 0; off <= span; off += stepMiB << 20) printf("offset %6zu MiB: %.3f ms\n", off >> 20, T(arena + off));
    return 0;
}
