// Diagnostic for the placement-dependent store rate of the LBS output streams: is it the relative alignment of the
// three streams (HBM channel / stack aliasing) or where the driver put each allocation?
//   hipcc --offload-arch=gfx950 -O3 tools/arena_bw.hip -o tools/arena_bw && ./tools/arena_bw
// For K arenas (one hipMalloc each, large enough for the three streams + shifts): the LBS store pattern with the streams carved at
// 2 MiB-aligned offsets, then with streams 2 / 3 shifted by s / 2s for a range of s; beside it K sets of three separate hipMallocs.
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
static float timeThree(void* a, void* b, void* c, int chars, int verts, hipEvent_t e0, hipEvent_t e1) {
    three<<<chars, 256>>>((float*)a, (float*)b, (v4f*)c, verts);
    hipEventRecord(e0);
    for (int r = 0; r < 4; ++r) three<<<chars, 256>>>((float*)a, (float*)b, (v4f*)c, verts);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 4;
}
int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const bool vmmOnly = argc > 1;
    const int chars = 10000, verts = 14080;
    const size_t nv = (size_t)chars * verts, MiB2 = 2u << 20;
    auto up = [&](size_t x) { return (x + MiB2 - 1) / MiB2 * MiB2; };
    const size_t s12 = up(nv * 12), s16 = up(nv * 16), slack = 64u << 20;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t shifts[] = {0, 4096, 65536, 262144, 1u << 20, 3u << 20, 17u << 20};
    const int K = vmmOnly ? 0 : 4;
    std::vector<char*> arenas(K);
    for (int k = 0; k < K; ++k) if (hipMalloc((void**)&arenas[k], 2 * s12 + s16 + slack) != hipSuccess) { printf("arena %d: allocation failed\n", k); return 1; }
    for (int k = 0; k < K; ++k) {
        printf("arena %d (%p):", k, (void*)arenas[k]);
        for (size_t s : shifts) printf("  shift %zuK %.3f ms", s >> 10, timeThree(arenas[k], arenas[k] + s12 + s, arenas[k] + 2 * s12 + 2 * s, chars, verts, e0, e1));
        printf("\n");
    }
    for (int k = 0; k < K; ++k) hipFree(arenas[k]);
    std::vector<void*> A(K), B(K), C(K);
    for (int k = 0; k < K; ++k) { hipMalloc(&A[k], nv * 12); hipMalloc(&B[k], nv * 12); hipMalloc(&C[k], nv * 16); }
    for (int k = 0; k < K; ++k) printf("separate set %d: %.3f ms (%p %p %p)\n", k, timeThree(A[k], B[k], C[k], chars, verts, e0, e1), A[k], B[k], C[k]);
    // the same separate sets again after freeing / re-allocating in another order
    for (int k = 0; k < K; ++k) { hipFree(B[k]); }
    for (int k = K - 1; k >= 0; --k) hipMalloc(&B[k], nv * 12);
    for (int k = 0; k < K; ++k) printf("re-allocated set %d: %.3f ms (%p %p %p)\n", k, timeThree(A[k], B[k], C[k], chars, verts, e0, e1), A[k], B[k], C[k]);
    for (int k = 0; k < K; ++k) { hipFree(A[k]); hipFree(B[k]); hipFree(C[k]); }
    // virtual-memory API: one reserved range per stream, backed by physical chunks of a chosen size
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) { printf("no VMM\n"); return 0; }
    printf("VMM granularity %zu KiB\n", gran >> 10);
    const size_t chunkSizes[] = {16u << 20, 64u << 20, 128u << 20, 256u << 20, 512u << 20, 0 /* whole stream */};
    for (size_t chunk : chunkSizes) {
        for (int rep = 0; rep < 3; ++rep) {
            void* ptr[3] = {nullptr, nullptr, nullptr};
            std::vector<hipMemGenericAllocationHandle_t> handles;
            const size_t want[3] = {nv * 12, nv * 12, nv * 16};
            size_t mapped[3];
            bool ok = true;
            for (int sidx = 0; sidx < 3 && ok; ++sidx) {
                const size_t c = chunk ? chunk : (want[sidx] + gran - 1) / gran * gran;
                const size_t total = (want[sidx] + c - 1) / c * c;
                mapped[sidx] = total;
                ok = hipMemAddressReserve(&ptr[sidx], total, 0, nullptr, 0) == hipSuccess;
                for (size_t off = 0; off < total && ok; off += c) {
                    hipMemGenericAllocationHandle_t h;
                    ok = hipMemCreate(&h, c, &prop, 0) == hipSuccess && hipMemMap((char*)ptr[sidx] + off, c, 0, h, 0) == hipSuccess;
                    if (ok) handles.push_back(h);
                }
                hipMemAccessDesc acc{};
                acc.location = prop.location;
                acc.flags = hipMemAccessFlagsProtReadWrite;
                ok = ok && hipMemSetAccess(ptr[sidx], total, &acc, 1) == hipSuccess;
            }
            if (ok) printf("VMM chunk %zu MiB (rep %d): %.3f ms\n", chunk >> 20, rep, timeThree(ptr[0], ptr[1], ptr[2], chars, verts, e0, e1));
            else printf("VMM chunk %zu MiB: failed (%s)\n", chunk >> 20, hipGetErrorString(hipGetLastError()));
            for (int sidx = 0; sidx < 3; ++sidx) if (ptr[sidx]) { hipMemUnmap(ptr[sidx], mapped[sidx]); hipMemAddressFree(ptr[sidx], mapped[sidx]); }
            for (auto h : handles) hipMemRelease(h);
        }
    }
    return 0;
}
