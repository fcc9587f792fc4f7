// Diagnostic for the placement-dependent store rate of the LBS output streams: is a slow candidate slow because of
// ONE of its buffers, or because of the combination of the three?
//   hipcc --offload-arch=gfx950 -O3 tools/alloc_probe.hip -o tools/alloc_probe && ./tools/alloc_probe
// Allocates K sets of the three output streams (all held together, as allocCrowdOutputs does while it probes), times
//   - each buffer alone with the same per-character store pattern (one stream),
//   - each set with the LBS three-stream pattern,
//   - mixed triples (stream 1 of set i, stream 2 of set j, stream 3 of set k).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
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
__global__ void one12(float* __restrict__ a, int verts) {
    size_t base = (size_t)blockIdx.x * verts;
    for (int v = threadIdx.x; v < verts; v += blockDim.x) {
        size_t o = base + v;
        __builtin_nontemporal_store(1.f, a + o * 3); __builtin_nontemporal_store(2.f, a + o * 3 + 1); __builtin_nontemporal_store(3.f, a + o * 3 + 2);
    }
}
__global__ void one16(v4f* __restrict__ c, int verts) {
    size_t base = (size_t)blockIdx.x * verts;
    for (int v = threadIdx.x; v < verts; v += blockDim.x) __builtin_nontemporal_store(v4f{1.f, 2.f, 3.f, 4.f}, c + base + v);
}
static hipEvent_t e0, e1;
template <class F> static float timeIt(F f) {
    f();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}
int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int K = argc > 1 ? atoi(argv[1]) : 12;
    const int chars = 10000, verts = 14080;
    const size_t nv = (size_t)chars * verts;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<void*> A(K), B(K), C(K);
    for (int k = 0; k < K; ++k)
        if (hipMalloc(&A[k], nv * 12) != hipSuccess || hipMalloc(&B[k], nv * 12) != hipSuccess || hipMalloc(&C[k], nv * 16) != hipSuccess) { printf("allocation %d failed\n", k); return 1; }
    std::vector<float> set(K), a1(K), b1(K), c1(K);
    for (int k = 0; k < K; ++k) {
        set[k] = timeIt([&] { three<<<chars, 256>>>((float*)A[k], (float*)B[k], (v4f*)C[k], verts); });
        a1[k] = timeIt([&] { one12<<<chars, 256>>>((float*)A[k], verts); });
        b1[k] = timeIt([&] { one12<<<chars, 256>>>((float*)B[k], verts); });
        c1[k] = timeIt([&] { one16<<<chars, 256>>>((v4f*)C[k], verts); });
        printf("set %2d: three streams %.3f ms | alone: pos %.3f (%.0f GB/s) nrm %.3f (%.0f GB/s) tan %.3f (%.0f GB/s) | %p %p %p\n", k, set[k],
               a1[k], nv * 12 / a1[k] * 1e-6, b1[k], nv * 12 / b1[k] * 1e-6, c1[k], nv * 16 / c1[k] * 1e-6, A[k], B[k], C[k]);
    }
    // mixed triples: best and worst sets exchange one stream at a time
    int best = 0, worst = 0;
    for (int k = 1; k < K; ++k) { if (set[k] < set[best]) best = k; if (set[k] > set[worst]) worst = k; }
    printf("best set %d (%.3f), worst set %d (%.3f)\n", best, set[best], worst, set[worst]);
    auto T = [&](int i, int j, int k) { return timeIt([&] { three<<<chars, 256>>>((float*)A[i], (float*)B[j], (v4f*)C[k], verts); }); };
    printf("pos of worst, rest of best: %.3f\n", T(worst, best, best));
    printf("nrm of worst, rest of best: %.3f\n", T(best, worst, best));
    printf("tan of worst, rest of best: %.3f\n", T(best, best, worst));
    printf("pos of best, rest of worst: %.3f\n", T(best, worst, worst));
    printf("nrm of best, rest of worst: %.3f\n", T(worst, best, worst));
    printf("tan of best, rest of worst: %.3f\n", T(worst, worst, best));
    // again, to see whether the numbers are stable
    for (int k = 0; k < K; ++k) printf("set %2d again: %.3f\n", k, timeIt([&] { three<<<chars, 256>>>((float*)A[k], (float*)B[k], (v4f*)C[k], verts); }));
    return 0;
}
