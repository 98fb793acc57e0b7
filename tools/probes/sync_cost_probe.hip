// Probe: what does a cross-stream dependency cost the stream that SIGNALS it?  A chain of N short kernels on stream A,
// with after each kernel (a) nothing, (b) hipEventRecord (+ hipStreamWaitEvent on stream B), (c) hipStreamWriteValue32
// (+ hipStreamWaitValue32 on B), (d) a 1-thread flag kernel (+ hipStreamWaitValue32 on B), (e) round 4: the event attached to
// the kernel dispatch itself (hipExtLaunchKernelGGL's stopEvent: no packet of its own) (+ hipStreamWaitEvent on B).
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/sync_cost_probe.hip -o /tmp/sync_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
__global__ void work(float* x) { float v = x[threadIdx.x]; for (int i = 0; i < 2000; ++i) v = v * 1.0001f + 0.5f; x[threadIdx.x] = v; }
__global__ void side(float* x) { x[threadIdx.x] += 1.f; }
__global__ void flagk(volatile unsigned* f, unsigned v) { *f = v; __threadfence_system(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int N = 200;
    float *a, *b; CK(hipMalloc(&a, 4096)); CK(hipMalloc(&b, 4096));
    hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, 1));
    std::vector<hipEvent_t> ev(N);
    for (auto& evi : ev) CK(hipEventCreateWithFlags(&evi, (unsigned)(hipEventDisableTiming | hipEventDisableSystemFence)));
    unsigned* flag = nullptr;
    hipError_t fe = hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory);
    if (fe != hipSuccess) { printf("signal memory: %s\n", hipGetErrorString(fe)); (void)hipGetLastError(); flag = nullptr; }
    else CK(hipMemset(flag, 0, 8));
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    unsigned epoch = 0;
    for (int mode = 0; mode < 5; ++mode) {
        if ((mode == 2 || mode == 3) && !flag) continue;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, sa));
            for (int i = 0; i < N; ++i) {
                if (mode == 4) {
                    hipExtLaunchKernelGGL(work, dim3(64), dim3(256), 0, sa, nullptr, ev[i], 0, a);
                    CK(hipStreamWaitEvent(sb, ev[i], 0)); side<<<1, 64, 0, sb>>>(b);
                    continue;
                }
                work<<<64, 256, 0, sa>>>(a);
                if (mode == 1) { CK(hipEventRecord(ev[i], sa)); CK(hipStreamWaitEvent(sb, ev[i], 0)); side<<<1, 64, 0, sb>>>(b); }
                if (mode == 2) { ++epoch; CK(hipStreamWriteValue32(sa, flag, epoch, 0)); CK(hipStreamWaitValue32(sb, flag, epoch, hipStreamWaitValueGte, 0xFFFFFFFFu)); side<<<1, 64, 0, sb>>>(b); }
                if (mode == 3) { ++epoch; flagk<<<1, 1, 0, sa>>>(flag, epoch); CK(hipStreamWaitValue32(sb, flag, epoch, hipStreamWaitValueGte, 0xFFFFFFFFu)); side<<<1, 64, 0, sb>>>(b); }
            }
            CK(hipEventRecord(t1, sa));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, t0, t1));
            if (rep == 2) printf("mode %d (%s): %.2f us per kernel on the signalling stream\n", mode,
                                 mode == 0 ? "no sync" : mode == 1 ? "event record" : mode == 2 ? "hipStreamWriteValue32" : mode == 3 ? "flag kernel" : "stop event on the dispatch (hipExtLaunchKernelGGL)", 1e3 * ms / N);
        }
    }
    return 0;
}
