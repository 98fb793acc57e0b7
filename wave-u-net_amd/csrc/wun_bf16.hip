// gfx950 bf16-MFMA speed mode of the implicit-GEMM conv (forward convs and input gradients).
//
//   D[time 16][cout 16] += A[time][k] * B[k][cout]   on v_mfma_f32_16x16x32_bf16 (fp32 accumulate)
//   k = (tap, input channel); one MFMA consumes 32 input channels of one tap.
//
// Operands are rounded to bf16 (round-to-nearest-even) when they are staged into LDS; everything
// in HBM -- activations, gradients, master weights, Adam state -- stays fp32, and the accumulators,
// bias, activation, mask and accumulate of the epilogue are fp32 exactly as in the exact-fp32
// kernel (the accumulator fragment layout of the two MFMA shapes is identical, so the epilogue is
// the same code).  Lane l of a wave supplies A[i = l&15][k = 8*(l>>4) .. +7] and
// B[k = 8*(l>>4) .. +7][j = l&15] and receives D[i = 4*(l>>4)+r][j = l&15].
//
// LDS images (per pipeline buffer):
//   X  [plane][row = time][32 channels] bf16, row pitch 96 B (64 B of data + 32 B pad: conflict-free
//      for the ds_read_b128 lane groups of gfx950 at every tap offset); the stride-2 loader keeps
//      even / odd input samples in two planes so tap k reads plane k&1 at row q + (k>>1)
//   W  [tap][channel group of 8][cout][8 channels] bf16 -- copied verbatim from the pre-packed
//      bf16 weight image (pack_bf16_kernel), so a B fragment is one aligned 16-byte read and 16
//      lanes read 256 contiguous bytes
// Pipeline: stage = (32-channel chunk, group of TG taps); weights of stage s+1 and the input window
// of chunk c+1 are fetched into registers while the MFMAs of stage s run and written to the other
// LDS buffer afterwards; one barrier per stage.
#include "wun_internal.h"

#include <cstdio>

namespace wun {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ unsigned pack2(float lo, float hi) { return bf16_rne(lo) | (bf16_rne(hi) << 16); }

__device__ __forceinline__ int xcd_block(int bid, int grid) {
    const int per = grid >> 3, rem = grid & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
}

#define WUN_BF_XPB 96          // bytes per X row in LDS
#define WUN_BF_KMAX 15         // taps

template <int MT, int NW>
__global__ __launch_bounds__(256) void conv_bf16_kernel(ConvArgs a, int nTT, int nNT, int TG, int ROWS) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WT = 4;
    constexpr int TT = WT * MT * 16;
    constexpr int NT = NW * 16;
    const bool deint = (a.loader == LOADER_DEINT);
    const int planes = deint ? 2 : 1;
    const int KW = a.KW;
    const int G = (KW + TG - 1) / TG;                       // tap groups per chunk
    // compile-time staging trip counts (upper bounds; the live count is checked at run time)
    constexpr int XIT = (8 * (TT + (WUN_BF_KMAX + 1) / 2) + 255) / 256 > (4 * (TT + WUN_BF_KMAX - 1) + 255) / 256
                            ? (8 * (TT + (WUN_BF_KMAX + 1) / 2) + 255) / 256
                            : (4 * (TT + WUN_BF_KMAX - 1) + 255) / 256;
    constexpr int WITMAX = (5 * 4 * NT + 255) / 256;        // TG <= 5

    const int xbytes = planes * ROWS * WUN_BF_XPB;
    const int wbytes = TG * 4 * NT * 16;
    unsigned char* Xs = smem;                               // two X buffers, then two W buffers
    unsigned char* Ws = smem + 2 * xbytes;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = xcd_block((int)blockIdx.x, (int)gridDim.x);
    const int nt = bid % nNT; bid /= nNT;
    const int tt = bid % nTT; bid /= nTT;
    const int b = bid;
    const int q0 = tt * TT, n0 = nt * NT;
    const int wt0 = wave * MT * 16;
    const int Ctot = a.C0 + a.C1;
    const int nchunks = (Ctot + 31) / 32;
    const int S = nchunks * G;

    const float* src0b = a.src0 + (long long)b * a.bs0 + a.off0;
    const float* src1b = (a.src1 != nullptr) ? a.src1 + (long long)b * a.bs1 + a.off1 : src0b;

    f32x4 acc[MT][NW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float xreg[XIT][8];
    u32x4 wreg[WITMAX];
    const int nxitems = planes * ROWS * 4;                  // (channel group, plane, row) items per chunk
    const int nwitems = TG * 4 * NT;                        // 16-byte weight items per stage

    // chunk-invariant state of the X items this thread stages, two registers per item:
    //   xti[i] = clamped source time | channel group << 26 | time-valid << 28 | item-live << 29
    //   xlo[i] = byte offset of the item's 16-byte slot in the LDS image
    int xti[XIT], xlo[XIT];
#pragma unroll
    for (int i = 0; i < XIT; ++i) {
        const int it = tid + i * 256;
        const bool live = it < nxitems;
        const int itc = live ? it : 0;
        const int c8l = itc / (planes * ROWS);
        const int pr = itc - c8l * (planes * ROWS);
        const int pl = deint ? pr / ROWS : 0;
        const int row = pr - pl * ROWS;
        const int t = (deint ? 2 * (q0 + row) + pl : q0 + row) - a.shift;
        const bool tok = t >= 0 && t < a.Tin;
        const int tc = t < 0 ? 0 : (t > a.Tin - 1 ? a.Tin - 1 : t);          // Tin < 2^26 (checked by the launcher)
        xti[i] = tc | (c8l << 26) | ((tok ? 1 : 0) << 28) | ((live ? 1 : 0) << 29);
        xlo[i] = (pl * ROWS + row) * WUN_BF_XPB + c8l * 16;
    }

    // ---- global -> registers ----
    auto load_x = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < XIT; ++i) {
            if (i * 256 < nxitems) {                         // uniform
                const int t = xti[i] & 0x3FFFFFF;
                const int cbase = chunk * 32 + ((xti[i] >> 26) & 3) * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    int c = cbase + e;
                    c = c < Ctot ? c : Ctot - 1;
                    const float* p = c < a.C0 ? src0b + (long long)c * a.pitch0 : src1b + (long long)(c - a.C0) * a.pitch1;
                    xreg[i][e] = p[t];
                }
            }
        }
    };
    auto store_x = [&](int chunk, int buf) {
        unsigned char* xb = Xs + buf * xbytes;
#pragma unroll
        for (int i = 0; i < XIT; ++i) {
            if (i * 256 < nxitems && ((xti[i] >> 29) & 1)) {
                const bool tok = (xti[i] >> 28) & 1;
                const int cbase = chunk * 32 + ((xti[i] >> 26) & 3) * 8;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (tok && cbase + e < Ctot) ? xreg[i][e] : 0.f;
                u32x4 pk = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
                *reinterpret_cast<u32x4*>(xb + xlo[i]) = pk;
            }
        }
    };
    // weights: a.W points at the packed bf16 image [KW][C8p][Npad][8]; a.wb_c8p / a.wb_npad give its shape
    const unsigned short* Wb = reinterpret_cast<const unsigned short*>(a.W);
    auto load_w = [&](int s) {
        const int chunk = s / G, g = s - chunk * G;
        const int j0 = g * TG;
#pragma unroll
        for (int i = 0; i < WITMAX; ++i) {
            const int it = tid + i * 256;
            if (i * 256 < nwitems) {
                const int itc = it < nwitems ? it : nwitems - 1;
                const int jg = itc / (4 * NT);
                const int r = itc - jg * (4 * NT);
                const int c8l = r / NT, n = r - c8l * NT;
                int j = j0 + jg;
                j = j < KW ? j : KW - 1;                     // taps past the filter are never read by the MFMA loop
                const long long off = (((long long)j * a.wb_c8p + chunk * 4 + c8l) * a.wb_npad + n0 + n) * 8;
                wreg[i] = *reinterpret_cast<const u32x4*>(Wb + off);
            }
        }
    };
    auto store_w = [&](int buf) {
        unsigned char* wbuf = Ws + buf * wbytes;
#pragma unroll
        for (int i = 0; i < WITMAX; ++i) {
            const int it = tid + i * 256;
            if (i * 256 < nwitems && it < nwitems) *reinterpret_cast<u32x4*>(wbuf + it * 16) = wreg[i];
        }
    };

    // ---- MFMA over the taps of one stage ----
    auto run_stage = [&](int s) {
        const int chunk = s / G, g = s - chunk * G;
        const int j0 = g * TG;
        int j1 = j0 + TG; if (j1 > KW) j1 = KW;
        const unsigned char* xb = Xs + (chunk & 1) * xbytes;
        const unsigned char* wbuf = Ws + (s & 1) * wbytes;
        for (int j = j0; j < j1; ++j) {
            const int pl = deint ? (j & 1) : 0;
            const int ro = deint ? (j >> 1) : j;
            const unsigned char* xa = xb + (pl * ROWS + wt0 + li + ro) * WUN_BF_XPB + lg * 16;
            const unsigned char* wp = wbuf + (((j - j0) * 4 + lg) * NT + li) * 16;
            bf16x8 av[MT], bv[NW];
#pragma unroll
            for (int m = 0; m < MT; ++m) av[m] = *reinterpret_cast<const bf16x8*>(xa + m * 16 * WUN_BF_XPB);
#pragma unroll
            for (int n = 0; n < NW; ++n) bv[n] = *reinterpret_cast<const bf16x8*>(wp + n * 16 * 16);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NW; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[m], bv[n], acc[m][n], 0, 0, 0);
        }
    };

    // ---- pipeline ----
    load_x(0);
    load_w(0);
    store_x(0, 0);
    store_w(0);
    __syncthreads();
    for (int s = 0; s < S; ++s) {
        const int chunk = s / G, g = s - chunk * G;
        const bool has_next = s + 1 < S;
        const bool next_chunk = chunk + 1 < nchunks;
        if (has_next) load_w(s + 1);
        if (g == 0 && next_chunk) load_x(chunk + 1);
        run_stage(s);
        if (has_next) store_w((s + 1) & 1);
        if (g == G - 1 && next_chunk) store_x(chunk + 1, (chunk + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue (fp32; same as the exact-fp32 kernel's) ----
    const bool lrelu = (a.flags & F_LRELU) != 0;
    const bool accum = (a.flags & F_ACCUM) != 0;
    const bool vec = (a.flags & F_VEC4) != 0;
#pragma unroll
    for (int n = 0; n < NW; ++n) {
        const int ncol = n0 + n * 16 + li;
        if (ncol >= a.N) continue;
        const float bvv = (a.bias != nullptr) ? a.bias[ncol] : 0.f;
        float* dst; const float* msk; long long rowbase;
        if (ncol < a.N0) {
            rowbase = (long long)b * a.obs0 + (long long)ncol * a.opitch0 + a.ooff0;
            dst = a.dst0; msk = a.msk0;
        } else {
            rowbase = (long long)b * a.obs1 + (long long)(ncol - a.N0) * a.opitch1 + a.ooff1;
            dst = a.dst1; msk = a.msk1;
        }
        float* decrow = (a.dec != nullptr && ncol < a.N0)
                            ? a.dec + (long long)b * a.decbs + (long long)ncol * a.decpitch : nullptr;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int q = q0 + wt0 + m * 16 + lg * 4;
            if (vec && q + 3 < a.Tout) {
                f32x4 v = acc[m][n];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] += bvv;
                    if (lrelu) v[r] = fmaxf(0.2f * v[r], v[r]);
                }
                const long long idx = rowbase + q;
                if (msk != nullptr) {
                    const f32x4 mk = *reinterpret_cast<const f32x4*>(&msk[idx]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= (mk[r] > 0.f) ? 1.f : 0.2f;
                }
                if (accum) v += *reinterpret_cast<const f32x4*>(&dst[idx]);
                *reinterpret_cast<f32x4*>(&dst[idx]) = v;
                if (decrow != nullptr) {
                    decrow[q >> 1] = v[0];
                    decrow[(q >> 1) + 1] = v[2];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (q + r < a.Tout) {
                        float v = acc[m][n][r] + bvv;
                        if (lrelu) v = fmaxf(0.2f * v, v);
                        const long long idx = rowbase + (long long)(q + r) * a.ostride;
                        if (msk != nullptr) v *= (msk[idx] > 0.f) ? 1.f : 0.2f;
                        if (accum) v += dst[idx];
                        dst[idx] = v;
                        if (decrow != nullptr && ((q + r) & 1) == 0) decrow[(q + r) >> 1] = v;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------
bool conv_bf16_supported(const ConvArgs& a) {
    if (a.flags & F_PHASE2) return false;
    if (a.C0 + a.C1 < 8) return false;                    // the 1-/2-channel audio input stays on the exact-fp32 kernel
    if (a.KW < 1 || a.KW > WUN_BF_KMAX) return false;
    if (a.Tin >= (1 << 26)) return false;
    return true;
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int MT, int NW>
static hipError_t conv_bf16_launch_t(ConvArgs a, hipStream_t s) {
    constexpr int TT = 4 * MT * 16, NT = NW * 16;
    const bool deint = a.loader == LOADER_DEINT;
    const int ROWS = deint ? TT + (a.KW + 1) / 2 : TT + a.KW - 1;
    const int TG = a.KW <= 5 ? a.KW : (a.KW <= 10 ? (a.KW + 1) / 2 : (a.KW + 2) / 3);
    const int nTT = (a.Tout + TT - 1) / TT, nNT = (a.N + NT - 1) / NT;
    const size_t lds = 2 * ((size_t)(deint ? 2 : 1) * ROWS * WUN_BF_XPB + (size_t)TG * 4 * NT * 16);
    auto kern = conv_bf16_kernel<MT, NW>;
    static size_t lds_allowed = 64 * 1024;
    if (lds > lds_allowed) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_allowed = lds;
    }
    const long long grid = (long long)nTT * nNT * a.B;
    if (grid <= 0) return hipSuccess;
    char nm[64], tag[160];
    snprintf(nm, sizeof(nm), "conv_bf16_kernel<%d, %d>", MT, NW);
    snprintf(tag, sizeof(tag), "C=%d N=%d T=%d K=%d ld=%d B=%d grid=%lld", a.C0 + a.C1, a.N, a.Tout, a.KW, a.loader, a.B, grid);
    prof_scope_begin(nm, conv_flops(a), s, tag);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, a, nTT, nNT, TG, ROWS);
    prof_scope_end(s);
    return hipGetLastError();
}

// a.W must point at the packed bf16 image of the layer's weights (pack_bf16_kernel), a.wb_c8p / a.wb_npad set
hipError_t launch_conv_bf16(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    if (!conv_bf16_supported(a) || a.wb_c8p <= 0 || a.wb_npad <= 0 || !al16(a.W)) return hipErrorInvalidValue;
    bool vec = a.ostride == 1 && al16(a.dst0) && (a.obs0 & 3) == 0 && (a.opitch0 & 3) == 0 && (a.ooff0 & 3) == 0;
    if (a.dst1 != nullptr) vec = vec && al16(a.dst1) && (a.obs1 & 3) == 0 && (a.opitch1 & 3) == 0 && (a.ooff1 & 3) == 0;
    if (a.msk0 != nullptr) vec = vec && al16(a.msk0);
    if (a.msk1 != nullptr) vec = vec && al16(a.msk1);
    if (a.dec != nullptr) vec = vec && (a.decpitch & 1) == 0 && (a.decbs & 1) == 0;
    if (vec) a.flags |= F_VEC4;
    // tile: fewest padded columns among 64/48/32; rows by how many tiles the launch has
    int bestnw = 4, bestpad = 1 << 30;
    const int cands[3] = {4, 3, 2};
    for (int i = 0; i < 3; ++i) {
        const int ntile = cands[i] * 16;
        const int padded = ((a.N + ntile - 1) / ntile) * ntile;
        if (padded < bestpad) { bestpad = padded; bestnw = cands[i]; }
    }
    const long long cols = (a.N + bestnw * 16 - 1) / (bestnw * 16);
    int mt = 4;
    while (mt > 1 && ((long long)((a.Tout + 64 * mt - 1) / (64 * mt)) * cols * a.B < 512 || a.Tout <= 32 * mt)) mt >>= 1;
#define WUN_BF(M, N) if (mt == M && bestnw == N) return conv_bf16_launch_t<M, N>(a, s);
    WUN_BF(4, 4) WUN_BF(4, 3) WUN_BF(4, 2)
    WUN_BF(2, 4) WUN_BF(2, 3) WUN_BF(2, 2)
    WUN_BF(1, 4) WUN_BF(1, 3) WUN_BF(1, 2)
#undef WUN_BF
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------
// weight packing: fp32 [KW][C][N] (TF layout, cout contiguous) -> bf16 [KW][C8p][Npad][8]
//   dst[((k*C8p + c/8)*Npad + n)*8 + c%8] = bf16(src[(k*C + c)*N + n]), zero where c >= C or n >= N
// One launch packs every conv of the plan (descriptor table in device memory).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ params, float* __restrict__ ws,
                                                        const PackDesc* __restrict__ descs) {
    const PackDesc d = descs[blockIdx.y];
    const float* src = (d.src_in_ws ? ws : params) + d.src_off;
    unsigned short* dst = reinterpret_cast<unsigned short*>(ws + d.dst_off);
    const long long total = (long long)d.KW * d.C8p * d.Npad;          // 16-byte items
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int n = (int)(i % d.Npad);
        const long long r = i / d.Npad;
        const int c8 = (int)(r % d.C8p), k = (int)(r / d.C8p);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c8 * 8 + e;
            v[e] = (c < d.C && n < d.N) ? src[((long long)k * d.C + c) * d.N + n] : 0.f;
        }
        u32x4 pk = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
        *reinterpret_cast<u32x4*>(dst + i * 8) = pk;
    }
}

hipError_t launch_pack_bf16(const float* params, float* ws, const PackDesc* dev_descs, int ndesc, long long max_items,
                            hipStream_t s) {
    if (ndesc <= 0) return hipSuccess;
    long long bx = (max_items + 255) / 256;
    if (bx > 256) bx = 256;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)bx, (unsigned)ndesc), dim3(256), 0, s, params, ws, dev_descs);
    return hipGetLastError();
}

// lane-layout probe of v_mfma_f32_16x16x32_bf16: d[16][16] = a[16][32] * b[32][16] (row-major fp32 in / out,
// operands rounded to bf16 in the kernel)
__global__ void mfma_bf16_probe_kernel(const float* a, const float* b, float* d) {
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, lg = lane >> 4;
    float av[8], bv[8];
    for (int e = 0; e < 8; ++e) { av[e] = a[li * 32 + lg * 8 + e]; bv[e] = b[(lg * 8 + e) * 16 + li]; }
    u32x4 ap = {pack2(av[0], av[1]), pack2(av[2], av[3]), pack2(av[4], av[5]), pack2(av[6], av[7])};
    u32x4 bp = {pack2(bv[0], bv[1]), pack2(bv[2], bv[3]), pack2(bv[4], bv[5]), pack2(bv[6], bv[7])};
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ap), __builtin_bit_cast(bf16x8, bp), c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[(4 * lg + r) * 16 + li] = c[r];
}

hipError_t launch_mfma_bf16_probe(const float* a, const float* b, float* d, hipStream_t s) {
    hipLaunchKernelGGL(mfma_bf16_probe_kernel, dim3(1), dim3(64), 0, s, a, b, d);
    return hipGetLastError();
}

}  // namespace wun
