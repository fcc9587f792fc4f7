// Diagnostic (round 4): what decides whether a placement of the tangent stream is fast or slow? The three-stream store pattern
// (tools/placement_pmc.hip) with the tangent stream scanned through an arena, under different allocation histories:
//   mode 0: pos, nrm, arena (the arena lies right below the two streams in the address space)
//   mode 1: arena, pos, nrm (the arena lies above them)
//   mode 2: dummy of D GiB first (kept), then pos, nrm, arena
//   mode 3: pos, nrm, dummy of D GiB (kept), arena
//   mode 4: pos, nrm inside the arena's top, tangents scanned below them (everything one allocation)
// and, per offset, the tangent stream written ALONE (is the region slow by itself?).
//   ./tools/placement_rule MODE [arena GiB] [step GiB] [dummy GiB]
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
__global__ void alone(v4f* __restrict__ c, int verts) {
    size_t base = (size_t)blockIdx.x * verts;
    for (int v = threadIdx.x; v < verts; v += blockDim.x) __builtin_nontemporal_store(v4f{1.f, 2.f, 3.f, 4.f}, c + base + v);
}
int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const size_t arenaGiB = argc > 2 ? (size_t)atoi(argv[2]) : 96, stepGiB = argc > 3 ? (size_t)atoi(argv[3]) : 4, dummyGiB = argc > 4 ? (size_t)atoi(argv[4]) : 24;
    const int chars = 10000, verts = 14080;
    const size_t nv = (size_t)chars * verts;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    void *A = nullptr, *B = nullptr, *D = nullptr;
    char* arena = nullptr;
    auto M = [&](void** p, size_t bytes) { if (hipMalloc(p, bytes) != hipSuccess) { printf("allocation of %zu bytes failed\n", bytes); exit(1); } };
    if (mode == 1) { M((void**)&arena, arenaGiB << 30); M(&A, nv * 12); M(&B, nv * 12); }
    else if (mode == 2) { M(&D, dummyGiB << 30); M(&A, nv * 12); M(&B, nv * 12); M((void**)&arena, arenaGiB << 30); }
    else if (mode == 3) { M(&A, nv * 12); M(&B, nv * 12); M(&D, dummyGiB << 30); M((void**)&arena, arenaGiB << 30); }
    else if (mode == 4) { M((void**)&arena, arenaGiB << 30); A = arena + (arenaGiB << 30) - nv * 12; B = (char*)A - nv * 12; }
    else { M(&A, nv * 12); M(&B, nv * 12); M((void**)&arena, arenaGiB << 30); }
    size_t freeB = 0, totalB = 0;
    (void)hipMemGetInfo(&freeB, &totalB);
    printf("mode %d: pos %p nrm %p dummy %p arena %p .. %p (%zu GiB); free %zu of %zu GiB\n", mode, A, B, D, (void*)arena, (void*)(arena + (arenaGiB << 30)), arenaGiB, freeB >> 30, totalB >> 30);
    auto T3 = [&](char* c) {
        three<<<chars, 256>>>((float*)A, (float*)B, (v4f*)c, verts);
        (void)hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) three<<<chars, 256>>>((float*)A, (float*)B, (v4f*)c, verts);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        return ms / 3;
    };
    auto T1 = [&](char* c) {
        alone<<<chars, 256>>>((v4f*)c, verts);
        (void)hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) alone<<<chars, 256>>>((v4f*)c, verts);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        return ms / 3;
    };
    const size_t span = (arenaGiB << 30) - nv * 16 - (mode == 4 ? 2 * nv * 12 : 0);
    for (size_t off = 0; off <= span; off += stepGiB << 30) {
        char* c = arena + off;
        const long long dist = (long long)((char*)B - c) / (1ll << 30);
        printf("offset %3zu GiB  (nrm - tan = %4lld GiB)  three streams %.3f ms   tangents alone %.3f ms\n", off >> 30, dist, T3(c), T1(c));
    }
    return 0;
}
