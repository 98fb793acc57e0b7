// gfx950 (MI355X / CDNA4): the elementwise / bandwidth-bound kernels of the Wave-U-Net hot path that BOTH modes share --
// 2x upsampling and its adjoint (UnetAudioSeparator.py:109-118, InterpolationLayer.py:4-40), the output head, its loss and
// gradients (OutputLayer.py:5-23, Training.py:50-63), the [B,T,C] -> NCW layout change, the transposed weight copies for the
// input-gradient convs, TF-Adam (Training.py:77).
//
// Its own translation unit (round 5) so that it can be built WITHOUT the packed fp32 VALU instructions (csrc/Makefile,
// NO_PK_FP32): in the bf16 mode these kernels run on CUs that bf16 MFMA kernels occupy, where packed fp32 instructions of a
// neighbouring wave were measured to miscompute.  The exact-fp32 MFMA kernels stay in wun_kernels.hip,
// built as before (their device code is unchanged by the split: tools/kernel_resources.sh, profiles).
#include "wun_internal.h"

#include <cstdio>
#include <cstdlib>

namespace wun {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
// HIP-event bracket of one launch for bench.py's per-kernel figures (the bracket list lives in wun_kernels.hip)
struct ProfScope {
    hipStream_t s;
    ProfScope(const char* name, double flops, hipStream_t st, const char* tag = "", double bytes = 0.0) : s(st) {
        prof_scope_begin(name, flops, st, tag, bytes);
    }
    ~ProfScope() { prof_scope_end(s); }
};
}  // namespace

// =====================================================================================
// upsampling (linear / learned), forward and backward
//   UnetAudioSeparator.py:109-118, InterpolationLayer.py:4-40
// =====================================================================================
// (ET: element type of x and y -- float, or bf16_t in the bf16 mode whose activations live in HBM as bf16; the arithmetic
//  is fp32 either way, ET = float compiles to the plain accesses)
template <typename ET>
__global__ void upsample_kernel(UpsampleArgs a) {
    const long long total = (long long)a.B * a.C * a.tup;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(i % a.tup);
        const long long bc = i / a.tup;
        const int c = (int)(bc % a.C), b = (int)(bc / a.C);
        const ET* x = reinterpret_cast<const ET*>(a.x) + (long long)b * a.xbs + (long long)c * a.xpitch;
        const int j = t >> 1;
        float v;
        if ((t & 1) == 0) {
            v = ld1<ET>(x, j);
        } else if (a.w != nullptr) {
            const float s = 1.f / (1.f + __expf(-a.w[c]));
            const float x1 = (j + 1 < a.n) ? ld1<ET>(x, j + 1) : 0.f;      // SAME: one zero on the right
            v = s * ld1<ET>(x, j) + (1.f - s) * x1;
        } else {
            const float x1 = (j + 1 < a.n) ? ld1<ET>(x, j + 1) : ld1<ET>(x, j);     // legacy bilinear clamps
            v = 0.5f * (ld1<ET>(x, j) + x1);
        }
        st1<ET>(reinterpret_cast<ET*>(a.y), (long long)b * a.ybs + (long long)c * a.ypitch + t, v);
    }
}

// Vector form (rows 16-byte aligned: the plan's buffers): block = 4 (b, c) rows x 64 lanes, a lane produces 4
// consecutive outputs from x[2i], x[2i+1], x[2i+2] -- one 8-byte + one 4-byte load, one 16-byte store, no integer
// division (the scalar kernel spends its time on three 64-bit divisions per element).  Same arithmetic per element.
template <typename ET>
__global__ __launch_bounds__(256) void upsample_vec_kernel(UpsampleArgs a) {
    const int row = (int)blockIdx.y * 4 + (int)(threadIdx.x >> 6);
    const int i = (int)blockIdx.x * 64 + (int)(threadIdx.x & 63);          // vector index: outputs 4i .. 4i+3
    if (row >= a.B * a.C || 4 * i >= a.tup) return;
    const int b = row / a.C, c = row - b * a.C;
    const ET* x = reinterpret_cast<const ET*>(a.x) + (long long)b * a.xbs + (long long)c * a.xpitch;
    ET* y = reinterpret_cast<ET*>(a.y) + (long long)b * a.ybs + (long long)c * a.ypitch;
    const int j = 2 * i;
    const float x0 = ld1<ET>(x, j);
    const float x1r = (j + 1 < a.n) ? ld1<ET>(x, j + 1) : 0.f, x2r = (j + 2 < a.n) ? ld1<ET>(x, j + 2) : 0.f;
    float o1, o3;
    if (a.w != nullptr) {
        const float sg = 1.f / (1.f + __expf(-a.w[c]));
        o1 = sg * x0 + (1.f - sg) * x1r;                                    // SAME: one zero on the right
        o3 = sg * x1r + (1.f - sg) * x2r;
    } else {
        o1 = 0.5f * (x0 + ((j + 1 < a.n) ? x1r : x0));                      // legacy bilinear clamps
        o3 = 0.5f * (x1r + ((j + 2 < a.n) ? x2r : x1r));
    }
    const int t = 4 * i;
    if (t + 3 < a.tup) {
        st4<ET>(y, t, (f32x4){x0, o1, x1r, o3});
    } else {
        st1<ET>(y, t, x0);
        if (t + 1 < a.tup) st1<ET>(y, t + 1, o1);
        if (t + 2 < a.tup) st1<ET>(y, t + 2, x1r);
    }
}

// bf16 rows: a lane produces 8 consecutive outputs y[8i .. 8i+7] from x[4i .. 4i+4] -- one 8-byte load + one 2-byte load,
// one 16-byte store (the generic vector form would issue three 2-byte loads per 8-byte store).  Same arithmetic per element.
__global__ __launch_bounds__(256) void upsample_vec8_bf16_kernel(UpsampleArgs a) {
    const int row = (int)blockIdx.y * 4 + (int)(threadIdx.x >> 6);
    const int i = (int)blockIdx.x * 64 + (int)(threadIdx.x & 63);
    if (row >= a.B * a.C || 8 * i >= a.tup) return;
    const int b = row / a.C, c = row - b * a.C;
    const bf16_t* x = reinterpret_cast<const bf16_t*>(a.x) + (long long)b * a.xbs + (long long)c * a.xpitch;
    bf16_t* y = reinterpret_cast<bf16_t*>(a.y) + (long long)b * a.ybs + (long long)c * a.ypitch;
    const int j = 4 * i;
    float xv[5];
    if (j + 3 < a.xpitch) {
        const f32x4 v = ld4<bf16_t>(x, j);                                 // (elements past n inside the padded row: masked below)
#pragma unroll
        for (int k = 0; k < 4; ++k) xv[k] = v[k];
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) xv[k] = j + k < a.n ? ld1<bf16_t>(x, j + k) : 0.f;
    }
    xv[4] = j + 4 < a.n ? ld1<bf16_t>(x, j + 4) : 0.f;
#pragma unroll
    for (int k = 1; k < 4; ++k) xv[k] = j + k < a.n ? xv[k] : 0.f;
    float o[8];
    float sg = 0.5f;
    if (a.w != nullptr) sg = 1.f / (1.f + __expf(-a.w[c]));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        o[2 * k] = xv[k];
        if (a.w != nullptr) o[2 * k + 1] = sg * xv[k] + (1.f - sg) * xv[k + 1];                     // SAME: one zero on the right
        else o[2 * k + 1] = 0.5f * (xv[k] + ((j + k + 1 < a.n) ? xv[k + 1] : xv[k]));               // legacy bilinear clamps
    }
    const int t = 8 * i;
    if (t + 7 < a.tup) {
        st4<bf16_t>(y, t, (f32x4){o[0], o[1], o[2], o[3]});
        st4<bf16_t>(y, t + 4, (f32x4){o[4], o[5], o[6], o[7]});
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (t + k < a.tup) st1<bf16_t>(y, t + k, o[k]);
    }
}

hipError_t launch_upsample(const UpsampleArgs& a, hipStream_t s) {
    ProfScope ps("upsample_kernel", 0.0, s, "", (a.bf ? 2.0 : 4.0) * (double)a.B * a.C * ((double)a.n + a.tup));
    if ((a.ypitch & 3) == 0 && (a.ybs & 3) == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0 && (long long)a.B * a.C <= 4 * 65535ll) {
        const int nvec = (a.tup + 3) / 4;
        const dim3 grid((unsigned)((nvec + 63) / 64), (unsigned)((a.B * a.C + 3) / 4));
        if (a.bf && (a.xpitch & 3) == 0 && (a.xbs & 3) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (a.ypitch & 7) == 0 && (a.ybs & 7) == 0) {
            const dim3 grid8((unsigned)(((a.tup + 7) / 8 + 63) / 64), (unsigned)((a.B * a.C + 3) / 4));
            hipLaunchKernelGGL(upsample_vec8_bf16_kernel, grid8, dim3(256), 0, s, a);
        } else if (a.bf) hipLaunchKernelGGL(upsample_vec_kernel<bf16_t>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(upsample_vec_kernel<float>, grid, dim3(256), 0, s, a);
        return hipGetLastError();
    }
    const long long total = (long long)a.B * a.C * a.tup;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (a.bf) hipLaunchKernelGGL(upsample_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(upsample_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

__device__ __forceinline__ float sigmoidf_exact(float w) { return 1.f / (1.f + expf(-w)); }

// dz[b][c][i] = lrelu'(x) * ( dy[2i] + wa*dy[2i+1] + wb*dy[2i-1] )
template <typename ET>
__global__ void upsample_bwd_kernel(UpsampleBwdArgs a) {
    const long long total = (long long)a.B * a.C * a.n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(idx % a.n);
        const long long bc = idx / a.n;
        const int c = (int)(bc % a.C), b = (int)(bc / a.C);
        const ET* dy = reinterpret_cast<const ET*>(a.dy) + (long long)b * a.ybs + (long long)c * a.ypitch;
        float wa = 0.5f, wb = 0.5f;
        if (a.w != nullptr) { wa = 1.f / (1.f + __expf(-a.w[c])); wb = 1.f - wa; }
        float g = ld1<ET>(dy, 2 * i);
        if (2 * i + 1 < a.tup) {
            float wgt = wa;
            if (a.w == nullptr && i == a.n - 1) wgt = 1.f;       // same-mode legacy clamp: out[2n-1] = x[n-1]
            g += wgt * ld1<ET>(dy, 2 * i + 1);
        }
        if (i >= 1) g += wb * ld1<ET>(dy, 2 * i - 1);
        const long long xi = (long long)b * a.xbs + (long long)c * a.xpitch + i;
        g *= (ld1<ET>(reinterpret_cast<const ET*>(a.x), xi) > 0.f) ? 1.f : 0.2f;
        st1<ET>(reinterpret_cast<ET*>(a.dz), xi, g);
    }
}

// dw[c] = sigmoid'(w[c]) * sum_{b,i} dy[2i+1] * (x[i] - x[i+1])   (x[n] = 0 in same mode)
template <typename ET>
__global__ __launch_bounds__(256) void interp_grad_kernel(UpsampleBwdArgs a) {
    __shared__ float red[256];
    const int c = blockIdx.x;
    float s = 0.f;
    const int nmid = a.tup / 2;               // number of odd outputs
    for (long long idx = threadIdx.x; idx < (long long)a.B * nmid; idx += 256) {
        const int i = (int)(idx % nmid), b = (int)(idx / nmid);
        const ET* x = reinterpret_cast<const ET*>(a.x) + (long long)b * a.xbs + (long long)c * a.xpitch;
        const float x1 = (i + 1 < a.n) ? ld1<ET>(x, i + 1) : 0.f;
        s += ld1<ET>(reinterpret_cast<const ET*>(a.dy), (long long)b * a.ybs + (long long)c * a.ypitch + 2 * i + 1) * (ld1<ET>(x, i) - x1);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float sg = sigmoidf_exact(a.w[c]);
        a.dw[c] = red[0] * sg * (1.f - sg);
    }
}

// Two-stage form of interp_grad_kernel (round 6).  One workgroup per CHANNEL (above) leaves 48 .. 312 workgroups walking
// B x n/2 strided elements each: 0.58 ms per step on M5 `full` for 12 launches that move 0.3 GB.  Stage 1: one workgroup per
// (excerpt, channel) row; a thread takes 4 consecutive mid-points -- dy[8k .. 8k+7] as two 16-byte loads (the odd elements
// are the mid-points' gradients), x[4k .. 4k+4] as one 16-byte load + one element -- block tree in fixed order -> partial[b][c].
// Stage 2: dw[c] = sigmoid'(w[c]) * sum_b partial[b][c] in excerpt order.  Deterministic; no atomics.
template <typename ET>
__global__ __launch_bounds__(256) void interp_grad_rows_kernel(UpsampleBwdArgs a) {
    __shared__ float red[256];
    const int c = blockIdx.x, b = blockIdx.y;
    const int nmid = a.tup / 2;                                              // number of odd outputs
    const ET* dy = reinterpret_cast<const ET*>(a.dy) + (long long)b * a.ybs + (long long)c * a.ypitch;
    const ET* x = reinterpret_cast<const ET*>(a.x) + (long long)b * a.xbs + (long long)c * a.xpitch;
    float s = 0.f;
    for (int i0 = 4 * (int)threadIdx.x; i0 < nmid; i0 += 1024) {
        if (i0 + 3 < nmid && i0 + 4 < a.n + 1 && 2 * i0 + 7 < a.tup) {
            const f32x4 d0 = ld4<ET>(dy, 2 * i0), d1 = ld4<ET>(dy, 2 * i0 + 4), xv = ld4<ET>(x, i0);
            const float x4 = (i0 + 4 < a.n) ? ld1<ET>(x, i0 + 4) : 0.f;
            s += d0[1] * (xv[0] - xv[1]);
            s += d0[3] * (xv[1] - xv[2]);
            s += d1[1] * (xv[2] - xv[3]);
            s += d1[3] * (xv[3] - x4);
        } else {
            for (int i = i0; i < i0 + 4 && i < nmid; ++i) {
                const float x1 = (i + 1 < a.n) ? ld1<ET>(x, i + 1) : 0.f;
                s += ld1<ET>(dy, 2 * i + 1) * (ld1<ET>(x, i) - x1);
            }
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.dw_partial[(long long)b * a.C + c] = red[0];
}

__global__ __launch_bounds__(256) void interp_grad_finish_kernel(UpsampleBwdArgs a) {
    const int c = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (c >= a.C) return;
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) s += a.dw_partial[(long long)b * a.C + c];
    const float sg = sigmoidf_exact(a.w[c]);
    a.dw[c] = s * sg * (1.f - sg);
}

// Vector form of upsample_bwd_kernel: a lane produces dz[4i .. 4i+3] from dy[8i-1 .. 8i+7] (two 16-byte loads + one
// scalar) and the mask vector; same arithmetic and summation order per element.
template <typename ET>
__global__ __launch_bounds__(256) void upsample_bwd_vec_kernel(UpsampleBwdArgs a) {
    const int row = (int)blockIdx.y * 4 + (int)(threadIdx.x >> 6);
    const int iv = (int)blockIdx.x * 64 + (int)(threadIdx.x & 63);         // vector index: dz[4iv .. 4iv+3]
    if (row >= a.B * a.C || 4 * iv >= a.n) return;
    const int b = row / a.C, c = row - b * a.C;
    const ET* dy = reinterpret_cast<const ET*>(a.dy) + (long long)b * a.ybs + (long long)c * a.ypitch;
    const ET* const xall = reinterpret_cast<const ET*>(a.x);
    ET* const dzall = reinterpret_cast<ET*>(a.dz);
    float wa = 0.5f, wb = 0.5f;
    if (a.w != nullptr) { wa = 1.f / (1.f + __expf(-a.w[c])); wb = 1.f - wa; }
    float d[9];                                                             // d[k] = dy[8iv - 1 + k] (0 outside)
    const int t0 = 8 * iv;
    d[0] = (t0 >= 1) ? ld1<ET>(dy, t0 - 1) : 0.f;
    if (t0 + 7 < a.tup) {
        const f32x4 v0 = ld4<ET>(dy, t0), v1 = ld4<ET>(dy, t0 + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { d[1 + k] = v0[k]; d[5 + k] = v1[k]; }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) d[1 + k] = (t0 + k < a.tup) ? ld1<ET>(dy, t0 + k) : 0.f;
    }
    const long long xi = (long long)b * a.xbs + (long long)c * a.xpitch + 4 * iv;
    float g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = 4 * iv + k;
        float v = d[1 + 2 * k];                                             // dy[2i]
        if (2 * i + 1 < a.tup) {
            float wgt = wa;
            if (a.w == nullptr && i == a.n - 1) wgt = 1.f;                  // same-mode legacy clamp: out[2n-1] = x[n-1]
            v += wgt * d[2 + 2 * k];
        }
        if (i >= 1) v += wb * d[2 * k];
        g[k] = v;
    }
    if (4 * iv + 3 < a.n) {
        const f32x4 xm = ld4<ET>(xall, xi);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = g[k] * ((xm[k] > 0.f) ? 1.f : 0.2f);
        st4<ET>(dzall, xi, o);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (4 * iv + k < a.n) st1<ET>(dzall, xi + k, g[k] * ((ld1<ET>(xall, xi + k) > 0.f) ? 1.f : 0.2f));
    }
}

hipError_t launch_upsample_bwd(const UpsampleBwdArgs& a, hipStream_t s) {
    ProfScope ps("upsample_bwd_kernel", 0.0, s, "", (a.bf ? 2.0 : 4.0) * (double)a.B * a.C * (2.0 * a.n + a.tup));
    const long long total = (long long)a.B * a.C * a.n;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    const bool vecok = (a.ypitch & 3) == 0 && (a.ybs & 3) == 0 && (reinterpret_cast<uintptr_t>(a.dy) & 15) == 0 &&
                       (a.xpitch & 3) == 0 && (a.xbs & 3) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(a.dz) & 15) == 0 && (long long)a.B * a.C <= 4 * 65535ll;
    if (vecok) {
        const int nvec = (a.n + 3) / 4;
        const dim3 grid((unsigned)((nvec + 63) / 64), (unsigned)((a.B * a.C + 3) / 4));
        if (a.bf) hipLaunchKernelGGL(upsample_bwd_vec_kernel<bf16_t>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(upsample_bwd_vec_kernel<float>, grid, dim3(256), 0, s, a);
    } else {
        if (a.bf) hipLaunchKernelGGL(upsample_bwd_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(upsample_bwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

// gradient of the learned interpolation weights of one level (InterpolationLayer.py:19-23): reads d(upsampled tensor) and the
// level's input, writes dw -- nothing the input-gradient chain waits for, so the plan runs it on a side stream
hipError_t launch_interp_grad(const UpsampleBwdArgs& a, hipStream_t s) {
    hipError_t e = hipSuccess;
    const bool vecok = (a.ypitch & 3) == 0 && (a.ybs & 3) == 0 && (reinterpret_cast<uintptr_t>(a.dy) & 15) == 0 &&
                       (a.xpitch & 3) == 0 && (a.xbs & 3) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;
    if (a.w != nullptr && a.dw != nullptr) {
        ProfScope ps("interp_grad_kernel", 0.0, s, "", (a.bf ? 2.0 : 4.0) * (double)a.B * a.C * (a.n + a.tup));
        // rows 16-byte aligned (the plan's buffers) and a scratch vector given: the two-stage form
        if (a.dw_partial != nullptr && vecok && a.B <= 65535) {
            const dim3 grid((unsigned)a.C, (unsigned)a.B);
            if (a.bf) hipLaunchKernelGGL(interp_grad_rows_kernel<bf16_t>, grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL(interp_grad_rows_kernel<float>, grid, dim3(256), 0, s, a);
            hipLaunchKernelGGL(interp_grad_finish_kernel, dim3((unsigned)((a.C + 255) / 256)), dim3(256), 0, s, a);
        } else {
            if (a.bf) hipLaunchKernelGGL(interp_grad_kernel<bf16_t>, dim3((unsigned)a.C), dim3(256), 0, s, a);
            else hipLaunchKernelGGL(interp_grad_kernel<float>, dim3((unsigned)a.C), dim3(256), 0, s, a);
        }
        e = hipGetLastError();
    }
    return e;
}

// =====================================================================================
// output head (OutputLayer.py:5-23, UnetAudioSeparator.py:127-142) + loss (Training.py:50-63)
// =====================================================================================
#define WUN_MAX_HEAD_ACC 8   // Sh*C <= 4*2

__device__ __forceinline__ int head_block_floats(const HeadArgs& a) { return a.Ko * (a.C + a.F) * a.C + a.C; }

// (FT: element type of the feature map -- float, or bf16_t in the bf16 mode)
// one output position (b, t) of every source: hw = the heads' weights in LDS, blk = floats per source
template <typename FT>
__device__ __forceinline__ void head_fwd_pos(const HeadArgs& a, const float* hw, const FT* featp, int b, int t) {
    const int Cin = a.C + a.F;
    const int blk = a.Ko * Cin * a.C + a.C;
    float acc[WUN_MAX_HEAD_ACC];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int c = 0; c < 2; ++c)
            acc[s * 2 + c] = (s < a.Sh && c < a.C) ? hw[s * blk + a.Ko * Cin * a.C + c] : 0.f;
    if (a.Ko == 1 && t - a.padl >= 0 && t - a.padl < a.Tfeat) {
        // 1-tap head: the feature rows eight at a time, loads first (the generic loop is one dependent load per channel)
        const int tf = t - a.padl;
        for (int ci = 0; ci < a.C; ++ci) {
            const float xv = a.mix_ncw[(long long)b * a.mbs + (long long)ci * a.mpitch + a.moff_feat + tf];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    if (s < a.Sh && c < a.C) acc[s * 2 + c] += hw[s * blk + ci * a.C + c] * xv;
        }
        const FT* __restrict__ fr = featp + (long long)b * a.fbs + tf;
        for (int f0 = 0; f0 < a.F; f0 += 8) {
            float xf[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[j] = f0 + j < a.F ? ld1<FT>(fr, (long long)(f0 + j) * a.fpitch) : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (f0 + j < a.F) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int c = 0; c < 2; ++c)
                            if (s < a.Sh && c < a.C) acc[s * 2 + c] += hw[s * blk + (a.C + f0 + j) * a.C + c] * xf[j];
                }
        }
    } else
    for (int k = 0; k < a.Ko; ++k) {
        const int tf = t + k - a.padl;
        if (tf < 0 || tf >= a.Tfeat) continue;
        for (int ci = 0; ci < Cin; ++ci) {
            const float xv = (ci < a.C)
                ? a.mix_ncw[(long long)b * a.mbs + (long long)ci * a.mpitch + a.moff_feat + tf]
                : ld1<FT>(featp, (long long)b * a.fbs + (long long)(ci - a.C) * a.fpitch + tf);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    if (s < a.Sh && c < a.C)
                        acc[s * 2 + c] += hw[s * blk + (k * Cin + ci) * a.C + c] * xv;
        }
    }
    float tot0 = 0.f, tot1 = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (s < a.Sh && c < a.C) {
                float v = acc[s * 2 + c];
                if (a.tanh_act) v = tanhf(v);
                else if (!a.training) v = fminf(fmaxf(v, -1.f), 1.f);
                a.out[(((long long)s * a.B + b) * a.Tout + t) * a.C + c] = v;
                if (c == 0) tot0 += v; else tot1 += v;
            }
        }
    if (a.difference) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (c < a.C) {
                float v = a.mix_ncw[(long long)b * a.mbs + (long long)c * a.mpitch + a.moff_diff + t] -
                          (c == 0 ? tot0 : tot1);
                if (!a.training) v = fminf(fmaxf(v, -1.f), 1.f);
                a.out[(((long long)(a.S - 1) * a.B + b) * a.Tout + t) * a.C + c] = v;
            }
        }
    }
}

template <typename FT>
__global__ __launch_bounds__(256) void head_fwd_kernel(HeadArgs a, long long h0, long long h1,
                                                       long long h2, long long h3) {
    extern __shared__ float hw[];
    const FT* const featp = reinterpret_cast<const FT*>(a.feat);
    const long long hoff[4] = {h0, h1, h2, h3};
    const int blk = a.Ko * (a.C + a.F) * a.C + a.C;
    for (int s = 0; s < a.Sh; ++s)
        for (int i = threadIdx.x; i < blk; i += 256) hw[s * blk + i] = a.Wh[hoff[s] + i];
    __syncthreads();
    const long long total = (long long)a.B * a.Tout;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * 256) {
        const int t = (int)(idx % a.Tout), b = (int)(idx / a.Tout);
        head_fwd_pos<FT>(a, hw, featp, b, t);
    }
}


// 1-tap head (every shipped config), FOUR consecutive positions per thread: a feature row is one 16-byte (fp32) / 8-byte
// (bf16) load per thread instead of four element loads in four threads, the per-source weights are read from LDS once
// per four positions, the outputs of a source are one (mono) or two (stereo) 16-byte stores.  Same arithmetic and the same
// summation order per element as head_fwd_kernel's 1-tap path: bit-identical outputs.  Tout % 4 tail positions, an
// unaligned feature map or Ko > 1 take head_fwd_kernel.  (The [S][B][Tout][C] output rows start at (s B + b) Tout C floats: only
// dword-aligned for odd Tout -- the output vectors are stored through a 4-byte-aligned vector type.)
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
template <typename FT>
__global__ __launch_bounds__(256) void head_fwd4_kernel(HeadArgs a, long long h0, long long h1, long long h2, long long h3) {
    extern __shared__ float hw[];
    const FT* const featp = reinterpret_cast<const FT*>(a.feat);
    const long long hoff[4] = {h0, h1, h2, h3};
    const int Cin = a.C + a.F;
    const int blk = Cin * a.C + a.C;                       // Ko == 1
    for (int s = 0; s < a.Sh; ++s)
        for (int i = threadIdx.x; i < blk; i += 256) hw[s * blk + i] = a.Wh[hoff[s] + i];
    __syncthreads();
    const int nq = a.Tout >> 2;                            // whole groups of four positions per excerpt
    const long long total = (long long)a.B * nq;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int qi = (int)(idx % nq), b = (int)(idx / nq);
        const int t = 4 * qi;                              // padl == 0: feature position == output position
        float acc[4][WUN_MAX_HEAD_ACC];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    acc[r][s * 2 + c] = (s < a.Sh && c < a.C) ? hw[s * blk + Cin * a.C + c] : 0.f;
        for (int ci = 0; ci < a.C; ++ci) {
            const float* mr = a.mix_ncw + (long long)b * a.mbs + (long long)ci * a.mpitch + a.moff_feat + t;
            const float xv[4] = {mr[0], mr[1], mr[2], mr[3]};
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    if (s < a.Sh && c < a.C) {
                        const float w = hw[s * blk + ci * a.C + c];
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r][s * 2 + c] += w * xv[r];
                    }
        }
        const FT* __restrict__ fr = featp + (long long)b * a.fbs + t;
        for (int f0 = 0; f0 < a.F; f0 += 4) {
            f32x4 xf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) xf[j] = f0 + j < a.F ? ld4<FT>(fr, (long long)(f0 + j) * a.fpitch) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (f0 + j < a.F) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int c = 0; c < 2; ++c)
                            if (s < a.Sh && c < a.C) {
                                const float w = hw[s * blk + (a.C + f0 + j) * a.C + c];
#pragma unroll
                                for (int r = 0; r < 4; ++r) acc[r][s * 2 + c] += w * xf[j][r];
                            }
                }
        }
        float tot[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r) { tot[r][0] = 0.f; tot[r][1] = 0.f; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < a.Sh) {
                float v[4][2];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        float x = acc[r][s * 2 + c];
                        if (a.tanh_act) x = tanhf(x);
                        else if (!a.training) x = fminf(fmaxf(x, -1.f), 1.f);
                        v[r][c] = x;
                        if (c < a.C) tot[r][c] += x;
                    }
                float* op = a.out + (((long long)s * a.B + b) * a.Tout + t) * a.C;
                if (a.C == 1) {
                    *reinterpret_cast<f32x4u*>(op) = (f32x4u){v[0][0], v[1][0], v[2][0], v[3][0]};
                } else {
                    *reinterpret_cast<f32x4u*>(op) = (f32x4u){v[0][0], v[0][1], v[1][0], v[1][1]};
                    *reinterpret_cast<f32x4u*>(op + 4) = (f32x4u){v[2][0], v[2][1], v[3][0], v[3][1]};
                }
            }
        }
        if (a.difference) {
            float v[4][2];
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (c < a.C) {
                    const float* mr = a.mix_ncw + (long long)b * a.mbs + (long long)c * a.mpitch + a.moff_diff + t;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x = mr[r] - tot[r][c];
                        if (!a.training) x = fminf(fmaxf(x, -1.f), 1.f);
                        v[r][c] = x;
                    }
                }
            float* op = a.out + (((long long)(a.S - 1) * a.B + b) * a.Tout + t) * a.C;
            if (a.C == 1) {
                *reinterpret_cast<f32x4u*>(op) = (f32x4u){v[0][0], v[1][0], v[2][0], v[3][0]};
            } else {
                *reinterpret_cast<f32x4u*>(op) = (f32x4u){v[0][0], v[0][1], v[1][0], v[1][1]};
                *reinterpret_cast<f32x4u*>(op + 4) = (f32x4u){v[2][0], v[2][1], v[3][0], v[3][1]};
            }
        }
    }
    // the Tout % 4 last positions of every excerpt: one position per thread, the generic path
    const int tail = a.Tout & 3;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < (long long)a.B * tail; idx += (long long)gridDim.x * 256)
        head_fwd_pos<FT>(a, hw, featp, (int)(idx / tail), 4 * nq + (int)(idx % tail));
}

// ... and the same for d(feature map): the Sh * C gradient samples of four positions are loaded once (16-byte loads), each
// feature row is one vector load (the LeakyReLU mask) and one vector store.  Bit-identical to head_dfeat_kernel's 1-tap path.
template <typename FT>
__global__ __launch_bounds__(256) void head_dfeat4_kernel(HeadArgs a, long long h0, long long h1, long long h2, long long h3) {
    extern __shared__ float hw[];
    const FT* const featp = reinterpret_cast<const FT*>(a.feat);
    FT* const dzp = reinterpret_cast<FT*>(a.dzfeat);
    const long long hoff[4] = {h0, h1, h2, h3};
    const int Cin = a.C + a.F;
    const int blk = Cin * a.C + a.C;
    for (int s = 0; s < a.Sh; ++s)
        for (int i = threadIdx.x; i < blk; i += 256) hw[s * blk + i] = a.Wh[hoff[s] + i];
    __syncthreads();
    const int nq = a.Tfeat >> 2;
    const long long total = (long long)a.B * nq;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int qi = (int)(idx % nq), b = (int)(idx / nq);
        const int u = 4 * qi;                              // padl == 0 and Tfeat == Tout (1-tap head)
        f32x4 dp[WUN_MAX_HEAD_ACC];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                dp[s * 2 + c] = (s < a.Sh && c < a.C)
                    ? *reinterpret_cast<const f32x4*>(a.dpre + (long long)s * a.dps + (long long)b * a.dpbs + (long long)c * a.dppitch + u)
                    : (f32x4){0.f, 0.f, 0.f, 0.f};
        const FT* __restrict__ fr = featp + (long long)b * a.fbs + u;
        FT* __restrict__ dr = dzp + (long long)b * a.fbs + u;
        for (int f0 = 0; f0 < a.F; f0 += 4) {
            f32x4 xf[4], g[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) xf[j] = f0 + j < a.F ? ld4<FT>(fr, (long long)(f0 + j) * a.fpitch) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                g[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (f0 + j < a.F) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int c = 0; c < 2; ++c)
                            if (s < a.Sh && c < a.C) {
                                const float w = hw[s * blk + (a.C + f0 + j) * a.C + c];
#pragma unroll
                                for (int r = 0; r < 4; ++r) g[j][r] += w * dp[s * 2 + c][r];
                            }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (f0 + j < a.F) {
                    f32x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = g[j][r] * ((xf[j][r] > 0.f) ? 1.f : 0.2f);
                    st4<FT>(dr, (long long)(f0 + j) * a.fpitch, o);
                }
        }
    }
    const int tail = a.Tfeat & 3;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < (long long)a.B * tail; idx += (long long)gridDim.x * 256)
        head_dfeat_pos<FT>(a, hw, featp, dzp, (int)(idx / tail), 4 * nq + (int)(idx % tail));
}

// loss partials + dpre (gradient wrt the pre-activation of each head conv output)
__global__ __launch_bounds__(256) void head_bwd_kernel(HeadArgs a) {
    __shared__ float red[4];
    float lsum = 0.f;
    const long long total = (long long)a.B * a.Tout;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * 256) {
        const int t = (int)(idx % a.Tout), b = (int)(idx / a.Tout);
        for (int c = 0; c < a.C; ++c) {
            float glast = 0.f;
            if (a.difference) {
                const long long o = (((long long)(a.S - 1) * a.B + b) * a.Tout + t) * a.C + c;
                const float d = a.out[o] - a.tgt[o];
                lsum += d * d;
                glast = a.gscale * d;
            }
            for (int s = 0; s < a.Sh; ++s) {
                const long long o = (((long long)s * a.B + b) * a.Tout + t) * a.C + c;
                const float y = a.out[o];
                const float d = y - a.tgt[o];
                lsum += d * d;
                float g = a.gscale * d - glast;
                if (a.tanh_act) g *= (1.f - y * y);
                a.dpre[(long long)s * a.dps + (long long)b * a.dpbs + (long long)c * a.dppitch + t] = g;
            }
        }
    }
    // deterministic block reduction
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_down(lsum, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) a.loss_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// dzfeat[b][f][u] = lrelu'(feat) * sum_{k,s,c} W[s][k][C+f][c] * dpre[s][b][c][u - k + padl]
// one feature-map position (b, u), all F channels
template <typename FT>
__device__ __forceinline__ void head_dfeat_pos(const HeadArgs& a, const float* hw, const FT* featp, FT* dzp, int b, int u) {
    const int Cin = a.C + a.F;
    const int blk = a.Ko * Cin * a.C + a.C;
    if (a.Ko == 1) {
        // 1-tap head (every shipped config): the Sh * C gradient samples of this position are loaded once, the
        // feature rows four at a time with all loads issued before the first store (the generic loop below re-reads
        // the gradient per feature channel and serialises load -> store per channel: 25 us for 27 MB)
        const int t = u + a.padl;
        float dp[WUN_MAX_HEAD_ACC];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                dp[s * 2 + c] = (s < a.Sh && c < a.C && t >= 0 && t < a.Tout)
                    ? a.dpre[(long long)s * a.dps + (long long)b * a.dpbs + (long long)c * a.dppitch + t] : 0.f;
        const FT* __restrict__ fr = featp + (long long)b * a.fbs + u;
        FT* __restrict__ dr = dzp + (long long)b * a.fbs + u;
        for (int f0 = 0; f0 < a.F; f0 += 4) {
            float xf[4], g[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) xf[j] = f0 + j < a.F ? ld1<FT>(fr, (long long)(f0 + j) * a.fpitch) : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                g[j] = 0.f;
                if (f0 + j < a.F) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int c = 0; c < 2; ++c)
                            if (s < a.Sh && c < a.C) g[j] += hw[s * blk + (a.C + f0 + j) * a.C + c] * dp[s * 2 + c];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (f0 + j < a.F) st1<FT>(dr, (long long)(f0 + j) * a.fpitch, g[j] * ((xf[j] > 0.f) ? 1.f : 0.2f));
        }
        return;
    }
    for (int f = 0; f < a.F; ++f) {
        float g = 0.f;
        for (int k = 0; k < a.Ko; ++k) {
            const int t = u - k + a.padl;
            if (t < 0 || t >= a.Tout) continue;
            for (int s = 0; s < a.Sh; ++s)
                for (int c = 0; c < a.C; ++c)
                    g += hw[s * blk + (k * Cin + a.C + f) * a.C + c] *
                         a.dpre[(long long)s * a.dps + (long long)b * a.dpbs + (long long)c * a.dppitch + t];
        }
        const long long fi = (long long)b * a.fbs + (long long)f * a.fpitch + u;
        g *= (ld1<FT>(featp, fi) > 0.f) ? 1.f : 0.2f;
        st1<FT>(dzp, fi, g);
    }
}

template <typename FT>
__global__ __launch_bounds__(256) void head_dfeat_kernel(HeadArgs a, long long h0, long long h1,
                                                         long long h2, long long h3) {
    extern __shared__ float hw[];
    const FT* const featp = reinterpret_cast<const FT*>(a.feat);
    FT* const dzp = reinterpret_cast<FT*>(a.dzfeat);
    const long long hoff[4] = {h0, h1, h2, h3};
    const int blk = a.Ko * (a.C + a.F) * a.C + a.C;
    for (int s = 0; s < a.Sh; ++s)
        for (int i = threadIdx.x; i < blk; i += 256) hw[s * blk + i] = a.Wh[hoff[s] + i];
    __syncthreads();
    const long long total = (long long)a.B * a.Tfeat;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * 256) {
        const int u = (int)(idx % a.Tfeat), b = (int)(idx / a.Tfeat);
        head_dfeat_pos<FT>(a, hw, featp, dzp, b, u);
    }
}

__global__ __launch_bounds__(256) void loss_finish_kernel(const float* partial, int n, float scale, float* loss) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];     // fixed order -> deterministic
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = red[0] * scale;
}

hipError_t launch_loss_finish(const float* partial, int n, float scale, float* loss, hipStream_t s) {
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, s, partial, n, scale, loss);
    return hipGetLastError();
}

// =====================================================================================
// small utilities
// =====================================================================================
__global__ void btc_to_ncw_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int T,
                                  int C, int pitch) {
    const long long total = (long long)B * T * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long bt = i / C;
        const int t = (int)(bt % T), b = (int)(bt / T);
        dst[((long long)b * C + c) * pitch + t] = src[i];
    }
}

// The same copy for the audio tensors the reference feeds (1 or 2 channels, UnetAudioSeparator.py:27): grid y = excerpt,
// a thread owns 4 consecutive time steps of every channel -- no index divisions, one 16-byte store per channel row
// (rows of the NCW buffer are 16-byte aligned; the [B, T, C] source rows are not for odd T, so it is read by dwords,
// which coalesce across the wave all the same).  For C = 1 the two layouts hold the same samples in the same order;
// the copy only re-pitches the rows to the 16-byte aligned form every consumer's vector / DMA loads are written for.
template <int C>
__global__ __launch_bounds__(256) void btc_to_ncw_vec_kernel(const float* __restrict__ src, float* __restrict__ dst, int T,
                                                            int pitch) {
    const int b = blockIdx.y;
    const int t0 = ((int)blockIdx.x * 256 + (int)threadIdx.x) * 4;
    if (t0 >= T) return;
    const float* sp = src + ((long long)b * T + t0) * C;
    float v[4][C];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < C; ++c) v[r][c] = (t0 + r < T) ? sp[r * C + c] : 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float* dp = dst + ((long long)b * C + c) * pitch + t0;
        if (t0 + 3 < T) {
            *reinterpret_cast<f32x4*>(dp) = (f32x4){v[0][c], v[1][c], v[2][c], v[3][c]};
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (t0 + r < T) dp[r] = v[r][c];
        }
    }
}

hipError_t launch_btc_to_ncw(const float* src, float* dst, int B, int T, int C, int pitch,
                             hipStream_t s) {
    const long long total = (long long)B * T * C;
    if ((C == 1 || C == 2) && (pitch & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0 && B <= 65535) {
        ProfScope ps("btc_to_ncw_kernel", 0.0, s, "", 8.0 * (double)total);
        const dim3 grid((unsigned)((T + 1023) / 1024), (unsigned)B);
        if (C == 1) hipLaunchKernelGGL(btc_to_ncw_vec_kernel<1>, grid, dim3(256), 0, s, src, dst, T, pitch);
        else hipLaunchKernelGGL(btc_to_ncw_vec_kernel<2>, grid, dim3(256), 0, s, src, dst, T, pitch);
        return hipGetLastError();
    }
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    ProfScope ps("btc_to_ncw_kernel", 0.0, s, "", 8.0 * (double)total);
    hipLaunchKernelGGL(btc_to_ncw_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, B, T, C, pitch);
    return hipGetLastError();
}

__device__ __forceinline__ void make_wt_body(const float* __restrict__ src, float* __restrict__ dst,
                                             const WtDesc& d) {
    if (d.mode == 0) {
        const long long total = (long long)d.J * d.N * d.C;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
             i += (long long)gridDim.x * blockDim.x) {
            const int c = (int)(i % d.C);
            const long long jn = i / d.C;
            const int n = (int)(jn % d.N), j = (int)(jn / d.N);
            const int k = d.k_last - j * d.k_step;
            dst[i] = src[((long long)k * d.C + c) * d.N + n];
        }
    } else {
        // fused output phases: dst[j][n][p][c] = src[k_last - 2j + p][c][n]  (0 where that tap does not exist)
        const long long total = (long long)d.J * d.N * 2 * d.C;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
             i += (long long)gridDim.x * blockDim.x) {
            const int c = (int)(i % d.C);
            long long r = i / d.C;
            const int ph = (int)(r % 2); r /= 2;
            const int n = (int)(r % d.N), j = (int)(r / d.N);
            const int k = d.k_last - 2 * j + ph;
            dst[i] = (k >= 0 && k < d.k_step) ? src[((long long)k * d.C + c) * d.N + n] : 0.f;
        }
    }
}

// tiled [C][N] -> [N][C] transpose per tap through LDS: coalesced on both sides.
// grid.x = tiles, blockIdx.y = descriptor; block = 256 threads = 32 x 8
__global__ __launch_bounds__(256) void make_wt_kernel(const float* __restrict__ params, float* __restrict__ ws,
                                                      const WtDesc* __restrict__ descs) {
    __shared__ float tile[32][33];
    const WtDesc d = descs[blockIdx.y];
    const float* src = params + d.src_off;
    float* dst = ws + d.dst_off;
    const int tc = (d.C + 31) / 32, tn = (d.N + 31) / 32;
    const int nph = d.mode == 1 ? 2 : 1;
    const int ntiles = d.J * nph * tc * tn;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int r = t;
        const int in_ = r % tn; r /= tn;
        const int ic = r % tc; r /= tc;
        const int ph = r % nph; const int j = r / nph;
        const int k = d.mode == 1 ? d.k_last - 2 * j + ph : d.k_last - j * d.k_step;
        const bool kok = d.mode == 1 ? (k >= 0 && k < d.k_step) : true;
        __syncthreads();
#pragma unroll
        for (int yy = 0; yy < 4; ++yy) {
            const int c = ic * 32 + ty + yy * 8, n = in_ * 32 + tx;
            tile[ty + yy * 8][tx] = (kok && c < d.C && n < d.N) ? src[((long long)k * d.C + c) * d.N + n] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int yy = 0; yy < 4; ++yy) {
            const int n = in_ * 32 + ty + yy * 8, c = ic * 32 + tx;
            if (n < d.N && c < d.C) {
                const long long o = d.mode == 1 ? (((long long)j * d.N + n) * 2 + ph) * d.C + c
                                                : ((long long)j * d.N + n) * d.C + c;
                dst[o] = tile[tx][ty + yy * 8];
            }
        }
    }
}

hipError_t launch_make_wt(const float* params, float* ws, const WtDesc* dev_descs, int ndesc,
                          int max_elems, hipStream_t s) {
    if (ndesc <= 0) return hipSuccess;
    int bx = (max_elems + 1023) / 1024;
    if (bx > 128) bx = 128;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(make_wt_kernel, dim3(bx, ndesc), dim3(256), 0, s, params, ws, dev_descs);
    return hipGetLastError();
}

__global__ void make_wt_one_kernel(const float* __restrict__ src, float* __restrict__ dst, WtDesc d) {
    make_wt_body(src, dst, d);
}

hipError_t launch_make_wt_one(const float* src, float* dst, WtDesc d, hipStream_t s) {
    const long long total = (long long)d.J * d.N * d.C * (d.mode == 1 ? 2 : 1);
    long long blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) return hipSuccess;
    hipLaunchKernelGGL(make_wt_one_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, d);
    return hipGetLastError();
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float lr_t, float b1, float b2,
                            float eps, float gscale) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
    }
}

hipError_t launch_adam(float* p, const float* g, float* m, float* v, long long n, float lr_t,
                       float b1, float b2, float eps, float gscale, hipStream_t s) {
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    ProfScope ps("adam_kernel", 0.0, s, "", 28.0 * (double)n);       // p, m, v read + written, g read
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, n, lr_t, b1, b2,
                       eps, gscale);
    return hipGetLastError();
}

__global__ void fill_kernel(float* p, long long n, float val) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) p[i] = val;
}

hipError_t launch_fill(float* p, long long n, float val, hipStream_t s) {
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) return hipSuccess;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, n, val);
    return hipGetLastError();
}

// ---- head launch wrappers (need the per-source offsets) --------------------------------
static size_t head_lds(const HeadArgs& a) {
    return sizeof(float) * (size_t)a.Sh * (a.Ko * (a.C + a.F) * a.C + a.C);
}

// the four-positions-per-thread forms: 1-tap head without padding (feature position == output position), feature rows
// and output rows aligned for 16-byte (fp32) / 8-byte (bf16) vectors
static bool head_vec4_ok(const HeadArgs& a) {
    return a.Ko == 1 && a.padl == 0 && a.Tfeat == a.Tout && a.Tout >= 4 && (a.fpitch & 3) == 0 && (a.fbs & 3) == 0 &&
           (reinterpret_cast<uintptr_t>(a.feat) & 15) == 0;
}

hipError_t launch_head_fwd_off(const HeadArgs& a, const long long* hoff, hipStream_t s) {
    const long long total = (long long)a.B * a.Tout;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    // feature map + mix window read once, all sources written
    const double fb = a.featbf ? 2.0 : 4.0;
    ProfScope ps("head_fwd_kernel", 0.0, s, "", (double)a.B * ((double)a.Tfeat * (fb * a.F + 4.0 * a.C) + 4.0 * (double)a.Tout * a.S * a.C));
    if (head_vec4_ok(a)) {
        long long b4 = ((long long)a.B * (a.Tout >> 2) + 255) / 256;
        if (b4 > 4096) b4 = 4096;
        if (b4 < 1) b4 = 1;
        if (a.featbf) hipLaunchKernelGGL(head_fwd4_kernel<bf16_t>, dim3((unsigned)b4), dim3(256), head_lds(a), s, a, hoff[0], hoff[1], hoff[2], hoff[3]);
        else hipLaunchKernelGGL(head_fwd4_kernel<float>, dim3((unsigned)b4), dim3(256), head_lds(a), s, a, hoff[0], hoff[1], hoff[2], hoff[3]);
        return hipGetLastError();
    }
    if (a.featbf) hipLaunchKernelGGL(head_fwd_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), head_lds(a), s, a, hoff[0], hoff[1], hoff[2], hoff[3]);
    else hipLaunchKernelGGL(head_fwd_kernel<float>, dim3((unsigned)blocks), dim3(256), head_lds(a), s, a, hoff[0], hoff[1], hoff[2], hoff[3]);
    return hipGetLastError();
}

int head_bwd_blocks(const HeadArgs& a) {
    const long long total = (long long)a.B * a.Tout;
    long long blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    return (int)blocks;
}

hipError_t launch_head_bwd_off(const HeadArgs& a, const long long* hoff, hipStream_t s) {
    {
        // outputs + targets read, d(pre-activation) written
        ProfScope ps("head_bwd_kernel", 0.0, s, "", 4.0 * (double)a.B * a.Tout * a.C * (2.0 * a.S + a.Sh));
        hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)head_bwd_blocks(a)), dim3(256), 0, s, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // (d(pre-activation) read, the feature map read for its LeakyReLU mask, d(feature map) written)
    ProfScope ps("head_dfeat_kernel", 0.0, s, "", (double)a.B * (4.0 * (double)a.Tout * a.Sh * a.C + (a.featbf ? 4.0 : 8.0) * (double)a.Tfeat * a.F));
    const long long total = (long long)a.B * a.Tfeat;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (head_vec4_ok(a) && (a.dppitch & 3) == 0 && (a.dpbs & 3) == 0 && (a.dps & 3) == 0 && (reinterpret_cast<uintptr_t>(a.dpre) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(a.dzfeat) & 15) == 0) {
        long long b4 = ((long long)a.B * (a.Tfeat >> 2) + 255) / 256;
        if (b4 > 4096) b4 = 4096;
        if (b4 < 1) b4 = 1;
        if (a.featbf) hipLaunchKernelGGL(head_dfeat4_kernel<bf16_t>, dim3((unsigned)b4), dim3(256), head_lds(a), s, a, hoff[0], hoff[1], hoff[2], hoff[3]);
        else hipLaunchKernelGGL(head_dfeat4_kernel<float>, dim3((unsigned)b4), dim3(256), head_lds(a), s, a, hoff[0], hoff[1], hoff[2], hoff[3]);
        return hipGetLastError();
    }
    if (a.featbf) hipLaunchKernelGGL(head_dfeat_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), head_lds(a), s, a, hoff[0], hoff[1], hoff[2], hoff[3]);
    else hipLaunchKernelGGL(head_dfeat_kernel<float>, dim3((unsigned)blocks), dim3(256), head_lds(a), s, a, hoff[0], hoff[1], hoff[2], hoff[3]);
    return hipGetLastError();
}

}  // namespace wun
