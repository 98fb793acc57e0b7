// Price of ONE synchronisation between two dependent phases of the deep-level chain, measured both ways on the box
// ("one persistent launch per deep-level group ... XCD-hierarchical grid barrier"):
//   (a) a dependent KERNEL BOUNDARY: N back-to-back launches of a 256-workgroup kernel on one stream;
//   (b) the XCD-hierarchical GRID BARRIER of MI355X_MICROARCH.md ("barrier-xcd": per-XCC arrival counter, the last
//       arriver of an XCC bumps a top counter, the last XCC publishes a generation word per XCC; payload published with
//       write-through sc1 stores + s_waitcnt vmcnt(0), consumers read with sc1 loads: no buffer_wbl2 / buffer_inv) inside
//       ONE persistent launch of 256 workgroups running N phases.
// Each phase does the same toy work (a workgroup writes 4 KiB, then -- after the synchronisation -- reads the 4 KiB ANOTHER
// workgroup, on another XCD, wrote in the previous phase and checks it: stale reads are counted), idle and BESIDE a second
// stream that keeps every CU busy with an MFMA-bound kernel: the situation of the real step, where the weight gradients
// and the deferred skip-window convs run beside the chain.
// build + run (GPU box): hipcc -O3 --offload-arch=gfx950 tools/probes/xcd_barrier_probe.hip -o /tmp/xcd_barrier_probe && /tmp/xcd_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st_sc1(float* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ f32x4 ld_sc1(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

struct Bar { unsigned xcc_cnt[8 * 16]; unsigned top[16]; unsigned gen[8 * 16]; };     // one word per 64-byte line

// all workgroups of the grid must be co-resident (grid <= what the device holds); wg_per_xcc[x] = workgroups of THIS launch
// on XCC x (counted by the launch itself, below); every spin gives up after ~0.2 s (reported as 1 << 20 "stale" reads)
// so that a mis-count can never hang the box
#define SPIN_LIMIT 4000000
__device__ __forceinline__ void grid_barrier(Bar* bar, unsigned epoch, const unsigned* wg_per_xcc, int* stale) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's write-through stores have left the CU
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;
        const unsigned old = __hip_atomic_fetch_add(&bar->xcc_cnt[xcc * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == wg_per_xcc[xcc] * epoch) {               // last arriver of this XCC
            const unsigned t = __hip_atomic_fetch_add(&bar->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t + 1 == wg_per_xcc[8] * epoch)                 // last populated XCC: open the barrier for everybody
                for (int x = 0; x < 8; ++x) __hip_atomic_store(&bar->gen[x * 16], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int spins = 0;
        while (__hip_atomic_load(&bar->gen[xcc * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > SPIN_LIMIT) { atomicAdd(stale, 1 << 20); break; }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void phase_write(float* buf, int wg, int ph) {
    float* p = buf + ((size_t)(ph & 1) * 4096 + wg) * 1024 + threadIdx.x * 4;      // 4 KiB per workgroup, two generations
    const float v = (float)(ph * 4096 + wg);
    st_sc1(p, (f32x4){v, v + 0.25f, v + 0.5f, v + 0.75f});
}
__device__ __forceinline__ int phase_check(const float* buf, int wg, int nwg, int ph) {   // the block `nwg/2 + 3` further on: another XCD
    const int src = (wg + nwg / 2 + 3) % nwg;
    const f32x4 v = ld_sc1(buf + ((size_t)(ph & 1) * 4096 + src) * 1024 + threadIdx.x * 4);
    return v[0] == (float)(ph * 4096 + src) ? 0 : 1;
}

// cen[0..7]: census of this launch (zeroed by the host), cen[8]: number of populated XCCs, cen[9]: flat arrival counter
__global__ __launch_bounds__(256) void persistent_kernel(float* buf, Bar* bar, unsigned* cen, int nph, int* stale) {
    const int wg = blockIdx.x, nwg = gridDim.x;
    __shared__ unsigned wpx[9];
    if (threadIdx.x == 0) {
        // where did THIS launch's workgroups land?  (placement is not a contract: count, then one flat barrier)
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;
        __hip_atomic_fetch_add(&cen[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&cen[9], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(&cen[9], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nwg) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > SPIN_LIMIT) { atomicAdd(stale, 1 << 20); break; }
        }
        unsigned pop = 0;
        for (int x = 0; x < 8; ++x) {
            wpx[x] = __hip_atomic_load(&cen[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pop += wpx[x] > 0 ? 1u : 0u;
        }
        wpx[8] = pop;
    }
    __syncthreads();
    int bad = 0;
    for (int ph = 1; ph <= nph; ++ph) {
        phase_write(buf, wg, ph);
        grid_barrier(bar, (unsigned)ph, wpx, stale);
        bad += phase_check(buf, wg, nwg, ph);
    }
    if (bad) atomicAdd(stale, bad);
}

__global__ __launch_bounds__(256) void phase_kernel(float* buf, int ph, int* stale) {        // the same phase as its own launch
    const int wg = blockIdx.x, nwg = gridDim.x;
    if (ph > 1 && phase_check(buf, wg, nwg, ph - 1)) atomicAdd(stale, 1);
    phase_write(buf, wg, ph);
}

__global__ void census_kernel(unsigned* wg_per_xcc) {
    if (threadIdx.x == 0) atomicAdd(&wg_per_xcc[__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7], 1u);
}

// background load: fp32 MFMA chains, two workgroups per CU, ~`iters` x 32 cycles x 8 MFMAs per wave
__global__ __launch_bounds__(256) void mfma_load_kernel(float* sink, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float a = (float)threadIdx.x * 1e-6f, b = 1.0001f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 12345.f) sink[0] = s;
}

static double med(std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    const int NWG = 256, NPH = 64, REP = 15;
    float* buf; Bar* bar; unsigned* wpx; unsigned* cen; int* stale; float* sink;
    CHECK(hipMalloc(&buf, (size_t)2 * 4096 * 1024 * 4));
    CHECK(hipMalloc(&bar, sizeof(Bar)));
    CHECK(hipMalloc(&wpx, 8 * 4));
    CHECK(hipMalloc(&cen, 64));
    CHECK(hipMalloc(&stale, 4));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(wpx, 0, 32));
    CHECK(hipMemset(stale, 0, 4));
    hipStream_t s, s2;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipLaunchKernelGGL(census_kernel, dim3(NWG), dim3(256), 0, s, wpx);
    CHECK(hipStreamSynchronize(s));
    unsigned h[8];
    CHECK(hipMemcpy(h, wpx, 32, hipMemcpyDeviceToHost));
    printf("workgroups per XCC of a %d-workgroup grid:", NWG);
    for (int i = 0; i < 8; ++i) printf(" %u", h[i]);
    printf("\n");
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int loaded = 0; loaded < 2; ++loaded) {
        std::vector<float> tk, tp;
        for (int r = 0; r < REP; ++r) {
            // (a) N dependent launches
            if (loaded) hipLaunchKernelGGL(mfma_load_kernel, dim3(512), dim3(256), 0, s2, sink, 60000);
            CHECK(hipEventRecord(e0, s));
            for (int ph = 1; ph <= NPH; ++ph) hipLaunchKernelGGL(phase_kernel, dim3(NWG), dim3(256), 0, s, buf, ph, stale);
            CHECK(hipEventRecord(e1, s));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            tk.push_back(1e3f * ms / NPH);
            CHECK(hipDeviceSynchronize());
            // (b) one persistent launch, N barriers
            CHECK(hipMemsetAsync(bar, 0, sizeof(Bar), s));
            CHECK(hipMemsetAsync(cen, 0, 64, s));
            if (loaded) hipLaunchKernelGGL(mfma_load_kernel, dim3(512), dim3(256), 0, s2, sink, 60000);
            CHECK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(persistent_kernel, dim3(NWG), dim3(256), 0, s, buf, bar, cen, NPH, stale);
            CHECK(hipEventRecord(e1, s));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            tp.push_back(1e3f * ms / NPH);
            CHECK(hipDeviceSynchronize());
        }
        printf("%-26s per phase: %6.2f us as %d dependent launches | %6.2f us as one persistent launch with the XCD-hierarchical barrier\n",
               loaded ? "beside an MFMA-bound stream" : "idle device", med(tk), NPH, med(tp));
    }
    int hs = 0;
    CHECK(hipMemcpy(&hs, stale, 4, hipMemcpyDeviceToHost));
    printf("stale reads: %d\n", hs);
    return 0;
}
