// Probe (runs on the GPU box): does a kernel launched with hipExtAnyOrderLaunch start while the kernel in front of it ON THE SAME
// STREAM is still running, and does the next ordinary launch wait for both?  (hip_ext.h says the flag is "not supported on GFX9xx"
// for the module-launch entry point; the move stage would use it to put its multi-wave launch and its grouped launch on one queue.)
//   hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o /tmp/anyorder_probe && /tmp/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>

__global__ void spin_kernel(unsigned long long* stamps, int slot, long long cycles) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[slot * 2] = t0;
    while ((long long)(wall_clock64() - t0) < cycles) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[slot * 2 + 1] = wall_clock64();
}
__global__ void stamp_kernel(unsigned long long* stamps, int slot) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[slot * 2] = wall_clock64(); stamps[slot * 2 + 1] = stamps[slot * 2]; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    unsigned long long* d; CK(hipMalloc(&d, 64 * 8)); CK(hipMemset(d, 0, 64 * 8));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int rate = 0; CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0)); // kHz
    const long long us200 = (long long)rate * 200 / 1000, us50 = (long long)rate * 50 / 1000;
    for (int rep = 0; rep < 3; ++rep) {
        // A: 200 us (ordinary) ; B: 50 us, any order ; C: stamp, ordinary (must start after A and B have both ended)
        hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, s, d, 0, us200);
        hipExtLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d, 1, us50);
        hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, d, 2);
        // the same three without the flag
        hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, s, d, 3, us200);
        hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, s, d, 4, us50);
        hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, d, 5);
        CK(hipStreamSynchronize(s));
        unsigned long long h[12]; CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        auto us = [&](unsigned long long t) { return (double)(long long)(t - h[0]) * 1000.0 / rate; };
        printf("rep %d  any-order:  A %.1f..%.1f  B %.1f..%.1f  C %.1f   | ordinary:  A %.1f..%.1f  B %.1f..%.1f  C %.1f  (us)\n", rep,
               us(h[0]), us(h[1]), us(h[2]), us(h[3]), us(h[4]), us(h[6]) - us(h[6]), us(h[7]) - us(h[6]), us(h[8]) - us(h[6]), us(h[9]) - us(h[6]), us(h[10]) - us(h[6]));
        const bool overlapped = h[2] < h[1], joined = h[4] >= h[1] && h[4] >= h[3];
        printf("        B started %s A ended; C started %s both ended\n", overlapped ? "BEFORE" : "after", joined ? "after" : "BEFORE (!)");
    }
    return 0;
}
