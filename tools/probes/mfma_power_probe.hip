// Probe: what bounds a dense exact-fp32 MFMA stream on real (random) data -- pipe issue rate or the power-managed clock?
// One wave per SIMD (256-thread blocks, 1 per CU) or two; operands stream from LDS (16-byte reads, one per 4 k-steps and
// tile), 18 accumulator tiles (the 6 x 3 register tile of the weight-gradient kernel); LDS holds zeros, a constant or
// random floats.  Reports TFLOP/s, shader cycles (s_memtime) per MFMA and s_memtime ticks per 100 MHz tick.
// hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_power_probe.hip -o /tmp/mfma_power_probe && /tmp/mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(MODE >= 2 ? 512 : 256) void probe(const float* src, float* out, unsigned long long* stamps, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = src[i];
    __syncthreads();
    if (MODE >= 2 && threadIdx.x >= 256) {              // idle partner wave on every SIMD
        if (MODE == 3) { for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(100); }
        __syncthreads();
        return;
    }
    f32x4 acc[6][3];
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63;
    const float* base = lds + lane * 4;
    f32x4 a[2][6], b[2][3];
    auto ld = [&](int set, int it) {
        const float* p = base + ((it * 5) & 7) * 256;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (MODE == 1) {
                const float* q = p + i * 1024 + 2048 + (lane & 3) + 1;       // 4-byte aligned only: two ds_read2_b32
#pragma unroll
                for (int e = 0; e < 4; ++e) a[set][i][e] = q[e];
            } else {
                a[set][i] = *reinterpret_cast<const f32x4*>(p + i * 1024 + 2048);
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) b[set][i] = *reinterpret_cast<const f32x4*>(p + 9 * 1024 + i * 2048);
    };
    ld(0, 0);
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            ld(h ^ 1, it + h + 1);
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int m = 0; m < 6; ++m)
#pragma unroll
                    for (int n = 0; n < 3; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[h][m][w], b[h][n][w], acc[m][n], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (MODE >= 2) __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        stamps[2 * w] = t1 - t0; stamps[2 * w + 1] = r1 - r0;
    }
}

int main() {
    const int iters = 3000;
    float* src; float* out; unsigned long long* st;
    hipMalloc(&src, 16384 * 4); hipMalloc(&out, 512 * 512 * 4); hipMalloc(&st, 512 * 8 * 16);
    std::vector<float> h(16384);
    const char* names[4] = {"zeros", "ones", "random [-1,1)", "random, small exponent spread"};
    for (int data = 2; data < 3; ++data) {
        for (int i = 0; i < 16384; ++i) {
            const float r = (float)rand() / RAND_MAX;
            h[i] = data == 0 ? 0.f : data == 1 ? 1.f : data == 2 ? 2.f * r - 1.f : 1.f + 0.001f * r;
        }
        hipMemcpy(src, h.data(), 16384 * 4, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 4; ++mode)
        for (int bpc = 1; bpc <= (mode >= 2 ? 1 : 2); ++bpc) {
            const int grid = 256 * bpc;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 0, 0, src, out, st, iters);
                else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, 0, src, out, st, iters);
                else if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(512), 0, 0, src, out, st, iters);
                else hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(512), 0, 0, src, out, st, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> hs(2 * grid * 8);
            hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> cyc, clk;
            const int wpb = mode >= 2 ? 8 : 4;
            for (int w = 0; w < grid * wpb; ++w) {
                if (mode >= 2 && (w % 8) >= 4) continue; cyc.push_back((double)hs[2 * w] / (iters * 72.0)); clk.push_back((double)hs[2 * w] / (double)hs[2 * w + 1] * 0.1); }
            std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
            const double flops = (double)grid * 4 * iters * 72.0 * 2048.0;
            printf("mode %d %-30s waves/SIMD %d: %7.3f ms %6.1f TFLOP/s | s_memtime ticks per MFMA per wave %.2f | ticks per ns %.3f\n",
                   mode, names[data], bpc, ms, flops / ms / 1e9, cyc[cyc.size() / 2], clk[clk.size() / 2]);
        }
    }
    return 0;
}
