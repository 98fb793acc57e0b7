// Stand-alone reproducer for DESIGN.md 5.3: do packed fp32 VALU instructions
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) return different results from run to run while waves of ANOTHER kernel stream
// bf16 MFMAs (v_mfma_f32_16x16x32_bf16) on the same CUs?
//
//   kernel A<PK>  : every workgroup copies the same 32 KiB of fixed pseudo-random floats to LDS, then each thread runs an
//                   accumulate loop over 16-byte LDS reads (the shape of narrow_wgrad_kernel's inner loop: one dz vector, one
//                   x vector, FMAs into register accumulators) -- PK = 1: v_pk_fma_f32 + v_pk_mul_f32 + v_pk_add_f32 on register
//                   pairs (inline asm, so the instruction is certain), PK = 0: the same per-element IEEE operations as
//                   v_fma_f32 / v_mul_f32 / v_add_f32.  The workgroup writes a checksum of its staged LDS data and a checksum
//                   of all its accumulators' bits.  Both variants must give the same bits in every launch.
//   kernel B      : v_mfma_f32_16x16x32_bf16 spinner, two workgroups per CU, on a second stream (back to back for the whole run).
//   kernel C      : v_mfma_f32_16x16x4_f32 spinner, same shape (the control: the exact-fp32 step never showed the defect).
// Arms: A alone, A beside B, A beside C; each N launches of A<1> and N of A<0> (default 10000), every launch's per-workgroup
// checksums compared with the first launch's.  Reports launches / workgroups that differ.
//   kernel N      : the library kernel's own inner loop (C++, see below) -- the compiler's packed instruction mix, or, built with
//                   -DNOPK_BUILD -Xclang -target-feature -Xclang -packed-fp32-ops, what the library ships now.
// build + run (GPU box): hipcc -O3 --offload-arch=gfx950 tools/probes/pk_fma_probe.hip -o /tmp/pk_fma_probe && /tmp/pk_fma_probe [N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define NLDS 8192           // floats of fixed data per workgroup
#define NWG 256             // workgroups of A (one per CU, like the 130 - 256 workgroups of the narrow kernels)
#define RING 128            // launches between two host-side comparisons

template <int PK>
__global__ __launch_bounds__(256) void probe_a(const float* src, unsigned* out, int slot, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[NLDS];
    __shared__ unsigned red[2][256];
    const int tid = threadIdx.x;
    for (int i = tid; i < NLDS / 4; i += 256)
        *reinterpret_cast<f32x4*>(&lds[4 * i]) = *reinterpret_cast<const f32x4*>(&src[4 * i]);
    __syncthreads();
    unsigned stage_sum = 0;
    for (int i = tid; i < NLDS; i += 256) stage_sum += __builtin_bit_cast(unsigned, lds[i]) * (unsigned)(2 * i + 1);
    f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f}, bs = {0.f, 0.f};
    // thread (pair p, lane group g) of the narrow kernel: row pointers differ per thread, 16-byte reads
    const float* xr = lds + (tid & 15) * 256;
    const float* zr = lds + 4096 + ((tid >> 4) & 15) * 256;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 4
        for (int q4 = 0; q4 < 64; ++q4) {
            const f32x4 z = *reinterpret_cast<const f32x4*>(zr + 4 * ((q4 + it) & 63));
            const f32x4 x = *reinterpret_cast<const f32x4*>(xr + 4 * q4);
            f32x2 zlo = {z[0], z[1]}, zhi = {z[2], z[3]}, xlo = {x[0], x[1]}, xhi = {x[2], x[3]};
#ifndef NOPK_BUILD
            if (PK == 1) {
                f32x2 t;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc0) : "v"(xlo), "v"(zlo));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc1) : "v"(xhi), "v"(zhi));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(zlo), "v"(zhi));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(bs) : "v"(t));
            } else if (PK == 2) {
                // the op_sel forms the compiler emits: src1's LOW half for both lanes / src0 low + src1 high
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc0) : "v"(xlo), "v"(zlo));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc1) : "v"(xhi), "v"(zhi));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(bs) : "v"(zlo));
            } else if (PK == 5) {
                // ONLY op_sel_hi:[1,0,1] (the HIGH lane reads the LOW half of src1)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc0) : "v"(xlo), "v"(zlo));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc1) : "v"(xhi), "v"(zhi));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(bs) : "v"(zlo));
            } else if (PK == 6) {
                // ONLY op_sel:[0,1,0] (the LOW lane reads the HIGH half of src1)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc0) : "v"(xlo), "v"(zlo));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc1) : "v"(xhi), "v"(zhi));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(bs) : "v"(zlo));
            } else if (PK == 3) {
                // v_pk_mov_b32 assembling an operand pair from halves of two pairs, consumed by the next packed FMA
                f32x2 t;
                asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(t) : "v"(xlo), "v"(xhi));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc0) : "v"(t), "v"(zlo));
                asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(t) : "v"(zlo), "v"(zhi));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc1) : "v"(xhi), "v"(t));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(bs) : "v"(t));
            } else if (PK == 4) {
                // one half of an operand pair written by a 32-bit v_mov right before the packed instruction reads the pair
                f32x2 t = xlo, u = zhi;
                asm volatile("v_mov_b32 %0, %1" : "+v"(t[0]) : "v"(xhi[1]));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc0) : "v"(t), "v"(zlo));
                asm volatile("v_mov_b32 %0, %1" : "+v"(u[1]) : "v"(zlo[0]));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc1) : "v"(xhi), "v"(u));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(bs[0]) : "v"(t[1]), "v"(u[0]));
            } else
#endif
            {
                float t0, t1;
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc0[0]) : "v"(xlo[0]), "v"(zlo[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc0[1]) : "v"(xlo[1]), "v"(zlo[1]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc1[0]) : "v"(xhi[0]), "v"(zhi[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc1[1]) : "v"(xhi[1]), "v"(zhi[1]));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(zlo[0]), "v"(zhi[0]));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(zlo[1]), "v"(zhi[1]));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(bs[0]) : "v"(t0));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(bs[1]) : "v"(t1));
            }
        }
        // keep the magnitudes bounded without changing what the packed instructions see per launch
        acc0 *= 0.5f; acc1 *= 0.5f; bs *= 0.5f;
    }
    unsigned acc_sum = 0;
    acc_sum += __builtin_bit_cast(unsigned, acc0[0]) * 3u + __builtin_bit_cast(unsigned, acc0[1]) * 5u;
    acc_sum += __builtin_bit_cast(unsigned, acc1[0]) * 7u + __builtin_bit_cast(unsigned, acc1[1]) * 11u;
    acc_sum += __builtin_bit_cast(unsigned, bs[0]) * 13u + __builtin_bit_cast(unsigned, bs[1]) * 17u;
    red[0][tid] = stage_sum; red[1][tid] = acc_sum * (unsigned)(2 * tid + 1);
    __syncthreads();
    if (tid == 0) {
        unsigned s0 = 0, s1 = 0;
        for (int i = 0; i < 256; ++i) { s0 += red[0][i]; s1 += red[1][i]; }
        out[((size_t)slot * NWG + blockIdx.x) * 2] = s0;
        out[((size_t)slot * NWG + blockIdx.x) * 2 + 1] = s1;
    }
}

// The inner loop of narrow_wgrad_kernel<KT = 3, SI = 1> (wun_narrow.hip: the output head's weight gradient, the kernel whose
// accumulators differed), VERBATIM as C++ -- compiled with the default flags it becomes the instruction mix the library had
// (ds_read_b128 x 3, v_pk_fma_f32 with op_sel, v_pk_add_f32, v_pk_mov_b32, v_fmac_f32, v_mov_b32 of half a register pair right
// before a packed read); built with -DNOPK_BUILD ... -Xclang -target-feature -Xclang -packed-fp32-ops it is the library's
// current build.  G lane groups x NP pairs as in the head of M4 (52 pairs: 26 input channels x 2 output rows).
__global__ __launch_bounds__(256) void probe_n(const float* src, unsigned* out, int slot, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[NLDS];
    __shared__ unsigned red[2][256];
    constexpr int KT = 3, SI = 1, TQ = 256, XWIN = 3 * SI + KT, XV = (XWIN + 3) / 4, XW = (TQ * SI + KT - 1 + 4 * XV + 3) / 4 * 4, ZP = TQ + 4;
    const int tid = threadIdx.x;
    for (int i = tid; i < NLDS / 4; i += 256)
        *reinterpret_cast<f32x4*>(&lds[4 * i]) = *reinterpret_cast<const f32x4*>(&src[4 * i]);
    __syncthreads();
    unsigned stage_sum = 0;
    for (int i = tid; i < NLDS; i += 256) stage_sum += __builtin_bit_cast(unsigned, lds[i]) * (unsigned)(2 * i + 1);
    const int N = 2, Ctot = 26, NP = Ctot * N, G = 256 / NP;
    const bool live = tid < G * NP;
    const int p = live ? tid % NP : 0, g = live ? tid / NP : 0;
    const int ci = p / N, n = p - ci * N;
    const float* Xs = lds;                 // [Ctot][XW]   (26 x 272 floats = 7072)
    const float* Zs = lds + Ctot * XW;     // [N][ZP]      (2 x 260)
    float acc[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) acc[k] = 0.f;
    float accb = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (live) {
            const float* xr = Xs + ci * XW;
            const float* zr = Zs + n * ZP;
            for (int q4 = g; q4 < TQ / 4; q4 += G) {
                const f32x4 z = *reinterpret_cast<const f32x4*>(zr + 4 * q4);
                float xv[4 * XV];
#pragma unroll
                for (int v = 0; v < XV; ++v) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(xr + 4 * q4 * SI + 4 * v);
                    xv[4 * v] = t4[0]; xv[4 * v + 1] = t4[1]; xv[4 * v + 2] = t4[2]; xv[4 * v + 3] = t4[3];
                }
#pragma unroll
                for (int k = 0; k < KT; ++k)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[k] = fmaf(xv[r * SI + k], z[r], acc[k]);
                accb += (z[0] + z[1]) + (z[2] + z[3]);
            }
        }
#pragma unroll
        for (int k = 0; k < KT; ++k) acc[k] *= 0.5f;
        accb *= 0.5f;
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(accb));
    }
    unsigned acc_sum = __builtin_bit_cast(unsigned, acc[0]) * 3u + __builtin_bit_cast(unsigned, acc[1]) * 5u +
                       __builtin_bit_cast(unsigned, acc[2]) * 7u + __builtin_bit_cast(unsigned, accb) * 11u;
    red[0][tid] = stage_sum; red[1][tid] = acc_sum * (unsigned)(2 * tid + 1);
    __syncthreads();
    if (tid == 0) {
        unsigned s0 = 0, s1 = 0;
        for (int i = 0; i < 256; ++i) { s0 += red[0][i]; s1 += red[1][i]; }
        out[((size_t)slot * NWG + blockIdx.x) * 2] = s0;
        out[((size_t)slot * NWG + blockIdx.x) * 2 + 1] = s1;
    }
}

// bf16 / fp32 MFMA spinners: 8 independent accumulator tiles, operands in registers (no memory traffic), `iters` rounds
__global__ __launch_bounds__(256) void spin_bf16(float* sink, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned seed = threadIdx.x * 2654435761u + 12345u;
    unsigned w[4];
    for (int i = 0; i < 4; ++i) { seed = seed * 1664525u + 1013904223u; w[i] = (seed & 0x007F007Fu) | 0x3F003F00u; }   // bf16 pairs in [0.5, 1)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const bf16x8 a = __builtin_bit_cast(bf16x8, (u32x4){w[0], w[1], w[2], w[3]});
    const bf16x8 b = __builtin_bit_cast(bf16x8, (u32x4){w[3], w[2], w[1], w[0]});
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        if ((it & 1023) == 1023)
            for (int i = 0; i < 8; ++i) acc[i] *= 1e-30f;
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;
}
__global__ __launch_bounds__(256) void spin_f32(float* sink, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float a = 0.5f + (float)(threadIdx.x & 63) / 128.f, b = 1.f - (float)(threadIdx.x & 31) / 64.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        if ((it & 1023) == 1023)
            for (int i = 0; i < 8; ++i) acc[i] *= 1e-30f;
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 10000;
    const int iters_a = 8;                               // ~ 2 k packed instructions per thread and launch
    float* src; unsigned* out; float* sink;
    CHECK(hipMalloc(&src, NLDS * 4)); CHECK(hipMalloc(&out, (size_t)RING * NWG * 2 * 4)); CHECK(hipMalloc(&sink, 64));
    std::vector<float> h(NLDS);
    unsigned s = 1337u;
    for (int i = 0; i < NLDS; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((float)(s >> 8) / 16777216.f) * 2.f - 1.f; }
    CHECK(hipMemcpy(src, h.data(), NLDS * 4, hipMemcpyHostToDevice));
    hipStream_t sa, sb;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    // spinner length: ~ 2 ms per launch, so a handful are always queued beside A
    const int spin_iters = 60000;
    std::vector<unsigned> ref[7], got((size_t)RING * NWG * 2);
    const char* arm_name[3] = {"A alone", "A beside the bf16-MFMA spinner (v_mfma_f32_16x16x32_bf16)", "A beside the fp32-MFMA spinner (v_mfma_f32_16x16x4_f32)"};
#ifdef NOPK_BUILD
    const char* kind_name[8] = {"asm: plain v_pk_fma/mul/add     ", "asm: scalar VALU                ", "narrow loop, built WITHOUT pk ops",
#else
    const char* kind_name[8] = {"asm: plain v_pk_fma/mul/add     ", "asm: scalar VALU                ", "narrow loop, compiler's pk ops   ",
#endif
                                "asm: v_pk_fma with op_sel forms ", "asm: v_pk_mov_b32 -> v_pk_fma    ", "asm: v_mov half -> v_pk_fma      ",
                                "asm: only op_sel_hi:[1,0,1]     ", "asm: only op_sel:[0,1,0]        "};
    for (int arm = 0; arm < 3; ++arm) {
        for (int kind = 0; kind < 8; ++kind) {
#ifdef NOPK_BUILD
            if (kind != 1 && kind != 2) continue;          // (the assembler of this build refuses the packed instructions)
#endif
            std::vector<unsigned>& rf = ref[kind < 2 ? 0 : kind - 1];
            long bad_launches = 0, bad_wgs = 0, bad_stage = 0, launches = 0;
            double a_ms = 0.0;
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            for (int base = 0; base < N; base += RING) {
                const int n = (N - base) < RING ? (N - base) : RING;
                if (arm > 0) {
                    // keep the spinner stream busy for the whole ring (a ring of A launches takes ~ 2 ms alone, several times that
                    // beside the spinner): four spinner launches of >= 2 ms each per ring
                    for (int k = 0; k < 4; ++k) {
                        if (arm == 1) hipLaunchKernelGGL(spin_bf16, dim3(512), dim3(256), 0, sb, sink, spin_iters);
                        else hipLaunchKernelGGL(spin_f32, dim3(512), dim3(256), 0, sb, sink, spin_iters / 4);
                    }
                }
                CHECK(hipEventRecord(e0, sa));
                for (int i = 0; i < n; ++i) {
                    if (kind == 0) hipLaunchKernelGGL(probe_a<1>, dim3(NWG), dim3(256), 0, sa, src, out, i, iters_a);
                    else if (kind == 1) hipLaunchKernelGGL(probe_a<0>, dim3(NWG), dim3(256), 0, sa, src, out, i, iters_a);
                    else if (kind == 2) hipLaunchKernelGGL(probe_n, dim3(NWG), dim3(256), 0, sa, src, out, i, iters_a * 4);
                    else if (kind == 3) hipLaunchKernelGGL(probe_a<2>, dim3(NWG), dim3(256), 0, sa, src, out, i, iters_a);
                    else if (kind == 4) hipLaunchKernelGGL(probe_a<3>, dim3(NWG), dim3(256), 0, sa, src, out, i, iters_a);
                    else if (kind == 5) hipLaunchKernelGGL(probe_a<4>, dim3(NWG), dim3(256), 0, sa, src, out, i, iters_a);
                    else if (kind == 6) hipLaunchKernelGGL(probe_a<5>, dim3(NWG), dim3(256), 0, sa, src, out, i, iters_a);
                    else hipLaunchKernelGGL(probe_a<6>, dim3(NWG), dim3(256), 0, sa, src, out, i, iters_a);
                }
                CHECK(hipEventRecord(e1, sa));
                CHECK(hipStreamSynchronize(sa));
                float ms = 0.f; CHECK(hipEventElapsedTime(&ms, e0, e1)); a_ms += ms;
                CHECK(hipMemcpy(got.data(), out, (size_t)n * NWG * 2 * 4, hipMemcpyDeviceToHost));
                if (rf.empty()) rf.assign(got.begin(), got.begin() + NWG * 2);
                for (int i = 0; i < n; ++i) {
                    int bw = 0;
                    for (int w = 0; w < NWG; ++w) {
                        if (got[((size_t)i * NWG + w) * 2] != rf[w * 2]) ++bad_stage;
                        if (got[((size_t)i * NWG + w) * 2 + 1] != rf[w * 2 + 1]) ++bw;
                    }
                    bad_wgs += bw; bad_launches += bw > 0; ++launches;
                }
                if (arm > 0) CHECK(hipStreamSynchronize(sb));
            }
            printf("%-60s %s: %ld launches x %d workgroups, %.1f us per launch; accumulator checksums differ from the reference in "
                   "%ld launches (%ld workgroups); staged-LDS checksums differ in %ld workgroups\n",
                   arm_name[arm], kind_name[kind], launches, NWG, 1e3 * a_ms / launches, bad_launches, bad_wgs, bad_stage);
            fflush(stdout);
        }
    }
    printf("reference = first launch of the first arm of each kernel (the two inline-asm kernels share one: the scalar one must match the packed one bit for bit);\n"
           "us per launch beside a spinner vs alone shows that the kernels really shared the CUs\n");
    return 0;
}
