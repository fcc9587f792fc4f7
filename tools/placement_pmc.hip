// Diagnostic (round 4): the LBS three-stream store pattern over a FAST and a SLOW placement of the tangent stream,
// as two differently named kernels, so that `rocprofv3 --pmc` attributes its counters to each (tools/placement_pmc.sh).
// The positions and normals streams are separate allocations; the tangent stream is placed at successive offsets of one large
// arena (tools/tan_scan.hip found: fast anywhere in the arena's first ~64 GiB, slow in the rest), the scan picks the fastest
// and the slowest offset, then both are written 20 times.
//   hipcc --offload-arch=gfx950 -O3 tools/placement_pmc.hip -o tools/placement_pmc && ./tools/placement_pmc [arena GiB] [step GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int WHICH> // 0: fast placement, 1: slow placement, 2: the scan
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
    const size_t arenaGiB = argc > 1 ? (size_t)atoi(argv[1]) : 96, stepGiB = argc > 2 ? (size_t)atoi(argv[2]) : 4;
    const int chars = 10000, verts = 14080;
    const size_t nv = (size_t)chars * verts;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    void *A, *B;
    char* arena;
    if (hipMalloc(&A, nv * 12) != hipSuccess || hipMalloc(&B, nv * 12) != hipSuccess || hipMalloc((void**)&arena, arenaGiB << 30) != hipSuccess) { printf("allocation failed\n"); return 1; }
    printf("pos %p nrm %p arena %p (%zu GiB)\n", A, B, (void*)arena, arenaGiB);
    auto timeIt = [&](auto kernel, char* c, int reps) {
        kernel<<<chars, 256>>>((float*)A, (float*)B, (v4f*)c, verts);
        (void)hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) kernel<<<chars, 256>>>((float*)A, (float*)B, (v4f*)c, verts);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        return ms / reps;
    };
    const size_t span = (arenaGiB << 30) - nv * 16;
    size_t best = 0, worst = 0; float bestMs = 1e9f, worstMs = 0;
    for (size_t off = 0; off <= span; off += stepGiB << 30) {
        const float ms = timeIt(three<2>, arena + off, 3);
        printf("scan offset %3zu GiB (va %p): %.3f ms\n", off >> 30, (void*)(arena + off), ms);
        if (ms < bestMs) { bestMs = ms; best = off; }
        if (ms > worstMs) { worstMs = ms; worst = off; }
    }
    printf("fast placement: offset %zu GiB %.3f ms; slow placement: offset %zu GiB %.3f ms\n", best >> 30, bestMs, worst >> 30, worstMs);
    for (int round = 0; round < 2; ++round) {
        printf("three<0> (fast) %.3f ms\n", timeIt(three<0>, arena + best, 10));
        printf("three<1> (slow) %.3f ms\n", timeIt(three<1>, arena + worst, 10));
    }
    return 0;
}
