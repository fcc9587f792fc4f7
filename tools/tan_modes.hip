// Diagnostic: the three-stream pattern at a FAST and at a SLOW placement of the tangent stream (tools/tan_scan.hip: first
// 64 GiB / rest of a 96 GiB allocation), with the tangent store issued in different flavours, and with the roles of the buffers exchanged.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void three(float* __restrict__ a, float* __restrict__ b, float* __restrict__ c, int verts) {
    size_t base = (size_t)blockIdx.x * verts;
    for (int v = threadIdx.x; v < verts; v += blockDim.x) {
        size_t o = base + v;
        __builtin_nontemporal_store(1.f, a + o * 3); __builtin_nontemporal_store(2.f, a + o * 3 + 1); __builtin_nontemporal_store(3.f, a + o * 3 + 2);
        __builtin_nontemporal_store(1.f, b + o * 3); __builtin_nontemporal_store(2.f, b + o * 3 + 1); __builtin_nontemporal_store(3.f, b + o * 3 + 2);
        if (MODE == 0) __builtin_nontemporal_store(v4f{1.f, 2.f, 3.f, 4.f}, (v4f*)(c + o * 4));
        else if (MODE == 1) *(v4f*)(c + o * 4) = v4f{1.f, 2.f, 3.f, 4.f};
        else if (MODE == 2) { __builtin_nontemporal_store(v2f{1.f, 2.f}, (v2f*)(c + o * 4)); __builtin_nontemporal_store(v2f{3.f, 4.f}, (v2f*)(c + o * 4 + 2)); }
        else if (MODE == 3) { __builtin_nontemporal_store(1.f, c + o * 4); __builtin_nontemporal_store(2.f, c + o * 4 + 1); __builtin_nontemporal_store(3.f, c + o * 4 + 2); __builtin_nontemporal_store(4.f, c + o * 4 + 3); }
        else if (MODE == 4) __builtin_nontemporal_store(v4f{1.f, 2.f, 3.f, 4.f}, (v4f*)(c + ((size_t)(gridDim.x - 1 - blockIdx.x) * verts + v) * 4)); // characters in reverse order
    }
}
static hipEvent_t e0, e1;
template <int MODE> static float T(void* a, void* b, void* c) {
    const int chars = 10000, verts = 14080;
    three<MODE><<<chars, 256>>>((float*)a, (float*)b, (float*)c, verts);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) three<MODE><<<chars, 256>>>((float*)a, (float*)b, (float*)c, verts);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t nv = (size_t)10000 * 14080;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    void *A, *B;
    char* arena;
    if (hipMalloc(&A, nv * 12) != hipSuccess || hipMalloc(&B, nv * 12) != hipSuccess || hipMalloc((void**)&arena, (size_t)96 << 30) != hipSuccess) { printf("allocation failed\n"); return 1; }
    char* fast = arena + ((size_t)8 << 30);
    char* slow = arena + ((size_t)80 << 30);
    char* slow2 = arena + ((size_t)88 << 30);
    char* fast2 = arena + ((size_t)16 << 30);
    printf("tangent stream in the fast / slow region, store flavour:\n");
    printf("  nt dwordx4        %.3f / %.3f ms\n", T<0>(A, B, fast), T<0>(A, B, slow));
    printf("  plain dwordx4     %.3f / %.3f ms\n", T<1>(A, B, fast), T<1>(A, B, slow));
    printf("  nt 2 x dwordx2    %.3f / %.3f ms\n", T<2>(A, B, fast), T<2>(A, B, slow));
    printf("  nt 4 x dword      %.3f / %.3f ms\n", T<3>(A, B, fast), T<3>(A, B, slow));
    printf("  nt dwordx4, characters in reverse order  %.3f / %.3f ms\n", T<4>(A, B, fast), T<4>(A, B, slow));
    printf("all three streams inside the arena:\n");
    printf("  pos fast, nrm fast, tan fast   %.3f ms\n", T<0>(fast2, fast2 + ((size_t)4 << 30), fast));
    printf("  pos slow, nrm slow, tan fast   %.3f ms\n", T<0>(slow2, slow2 + ((size_t)4 << 30), fast));
    printf("  pos fast, nrm fast, tan slow   %.3f ms\n", T<0>(fast2, fast2 + ((size_t)4 << 30), slow));
    printf("  pos slow, nrm slow, tan slow   %.3f ms\n", T<0>(slow2, slow2 + ((size_t)4 << 30), slow));
    printf("  pos slow, nrm fast, tan slow   %.3f ms\n", T<0>(slow2, fast2, slow));
    return 0;
}
