// Micro-benchmark: issue rate of the exact-fp32 MFMA shapes on gfx950.
// hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_ubench.hip -o tools/probes/mfma_ubench && ./tools/probes/mfma_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// variants of the 12-accumulator loop approximating the conv kernel's inner loop
// MODE 1: 4 distinct A x 3 distinct B operands; MODE 2: + AGPR accumulators (inline asm);
// MODE 3: MODE 1 + 7 VALU moves refreshing the operands each iteration (stand-in for ds_read results)
template <int MODE>
__global__ void kmix(float* out, int iters, float a0, float b0) {
    f32x4 acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a[4], b[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = a0 + threadIdx.x + i;
#pragma unroll
    for (int i = 0; i < 3; ++i) b[i] = b0 + threadIdx.x * 2 + i;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[i]));
#pragma unroll
            for (int i = 0; i < 3; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(b[i]) : "v"(b[i]));
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                if (MODE == 2) {
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[m * 3 + n]) : "v"(a[m]), "v"(b[n]));
                } else {
                    acc[m * 3 + n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], acc[m * 3 + n], 0, 0, 0);
                }
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// LDS-fed variants: per k-step 4 A + 3 B operands come from LDS (conflict-free addresses that
// move with the iteration), software pipelined one operand set ahead of the 12 MFMAs that use it.
// W = floats per ds_read (1: b32, 2: b64, 4: b128); one read feeds W k-steps.
template <int W>
__global__ void klds(float* out, int iters, float a0, float b0) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = a0 + i * 1e-6f;
    __syncthreads();
    typedef float vec __attribute__((ext_vector_type(W)));
    f32x4 acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63;
    const float* base = lds + lane * W;                 // 64 lanes x W floats contiguous: conflict free
    vec a[2][4], b[2][3];
    auto ld = [&](int set, int it) {
        const float* p = base + ((it * 7) & 7) * 64 * W;     // moves with the iteration (no hoisting)
#pragma unroll
        for (int i = 0; i < 4; ++i) a[set][i] = *reinterpret_cast<const vec*>(p + i * 512);
#pragma unroll
        for (int i = 0; i < 3; ++i) b[set][i] = *reinterpret_cast<const vec*>(p + 2048 + 64 * W * 8 + i * 512);
    };
    ld(0, 0);
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            ld(h ^ 1, it + h + 1);
#pragma unroll
            for (int w = 0; w < W; ++w)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 3; ++n) {
                        float av, bv;
                        av = a[h][m][w]; bv = b[h][n][w];
                        acc[m * 3 + n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[m * 3 + n], 0, 0, 0);
                    }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static void run(const char* name, K kern, int nacc, double flop_per_mfma, int block, int blocks_per_cu, int iters = 4000) {
    float* out;
    const int grid = 256 * blocks_per_cu;
    hipMalloc(&out, (size_t)grid * block * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)grid * block / 64;
    const double flops = waves * iters * nacc * flop_per_mfma;
    printf("%-10s acc=%2d block=%4d blocks/CU=%d (waves/SIMD=%.1f): %.3f ms  %.1f TFLOP/s\n", name, nacc, block,
           blocks_per_cu, (double)block * blocks_per_cu / 256, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    const double f16 = 2.0 * 16 * 16 * 4, f32 = 2.0 * 32 * 32 * 2;
    for (int bpc = 1; bpc <= 4; ++bpc) {
        run("16x16x4", k16<1>, 1, f16, 256, bpc);
        run("16x16x4", k16<2>, 2, f16, 256, bpc);
        run("16x16x4", k16<4>, 4, f16, 256, bpc);
        run("16x16x4", k16<12>, 12, f16, 256, bpc);
        run("mix1 4Ax3B", kmix<1>, 12, f16, 256, bpc);
        run("mix2 +AGPR", kmix<2>, 12, f16, 256, bpc);
        run("mix3 +vmov", kmix<3>, 12, f16, 256, bpc);
        run("lds b32", klds<1>, 12, f16, 256, bpc, 4000);
        run("lds b64", klds<2>, 24, f16, 256, bpc, 2000);
        run("lds b128", klds<4>, 48, f16, 256, bpc, 1000);
        run("32x32x2", k32<1>, 1, f32, 256, bpc);
        run("32x32x2", k32<2>, 2, f32, 256, bpc);
        run("32x32x2", k32<4>, 4, f32, 256, bpc);
    }
    return 0;
}
