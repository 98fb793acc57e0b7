// gfx950 bf16-MFMA speed mode of the weight / bias gradient.
//
//   D[(cin,tap) 16][cout 16] += A[(cin,tap)][q] * B[q][cout]   on v_mfma_f32_16x16x32_bf16 (fp32 accumulate),
//   k = output position q, 32 positions per MFMA.
//
// Same decomposition as the exact-fp32 kernel (wun_kernels.hip, wgrad_mfma_kernel): same (row group,
// column group, split) tiles, same units of <= 128 positions, same tile-major split partials summed in
// fixed order by wgrad_reduce_kernel, an all-ones A row for the bias gradient.  What differs:
//   * the input rows and the dz rows are rounded to bf16 (nearest-even) when they are written to LDS
//     (HBM tensors stay fp32); the accumulators and the partials are fp32;
//   * a B fragment is one aligned 16-byte LDS read (8 consecutive positions of one dz row; row pitch
//     2*TK + 32 bytes keeps the ds_read_b128 lane groups conflict-free);
//   * an A fragment is 8 consecutive positions of input row `cin` starting at tap + alignment shift, i.e.
//     at an arbitrary 2-byte offset: eight 16-bit LDS reads (the price of serving all 15 taps from ONE
//     staged copy of the row; lanes of one channel read neighbouring halves of the same dwords, which
//     the LDS broadcasts).
#include "wun_internal.h"

#include <cstdio>

namespace wun {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

typedef __bf16 wb_bf16x2 __attribute__((ext_vector_type(2)));
typedef float wb_f32x2 __attribute__((ext_vector_type(2)));
// two fp32 -> packed bf16 pair, round-to-nearest-even: one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned wb_pack2(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((wb_f32x2){lo, hi}, wb_bf16x2));
}

__device__ __forceinline__ int wb_xcd_block(int bid, int grid) {
    const int per = grid >> 3, rem = grid & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
}

#define WUN_WGB_XIT 8      // float4 X loads per thread and unit (same staging bound as the fp32 kernel)

// XPe / ZPe: row pitches in bf16 ELEMENTS; ONESPe: length of the all-ones row
template <int MTW, int NW>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(WgradArgs a, int nMG, int nNG, int TK, int XPe, int ZPe,
                                                            int nChMax, int ONESPe, int XW4) {
    extern __shared__ __attribute__((aligned(16))) unsigned short wlds[];
    constexpr int MG = 4 * MTW * 16;
    constexpr int NG = NW * 16;
    constexpr int ZIT = (NG * 32 + 255) / 256;          // TK/4 <= 32 float4 per dz row
    const bool deint = (a.loader == LOADER_DEINT);
    const int planes = deint ? 2 : 1;
    unsigned short* Xs = wlds + ONESPe;
    unsigned short* Zs = Xs + ((nChMax * planes * XPe + 7) & ~7);      // 16-byte aligned

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = wb_xcd_block((int)blockIdx.x, (int)gridDim.x);
    const int ng = bid % nNG; bid /= nNG;
    const int mg = bid % nMG;
    const int split = bid / nMG;

    const int Ctot = a.C0 + a.C1;
    const int Mtot = Ctot * a.KW;                      // row Mtot is the bias (all-ones) row
    const int rlo = mg * MG;
    const int cLo = rlo / a.KW;
    int cHi = (rlo + MG - 1) / a.KW;
    if (cHi > Ctot - 1) cHi = Ctot - 1;
    const int nCh = cHi - cLo + 1;                     // may be <= 0 (bias-only group)

    const int delta0 = ((a.off0 - a.shift) % 4 + 4) % 4;
    const int delta1 = ((a.off1 - a.shift) % 4 + 4) % 4;

    int rowoff[MTW];                                   // element offset of this lane's A row in wlds (0 = ones row)
    int nact = 0;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int rt = rlo + (wave * MTW + mt) * 16;
        if (rt <= Mtot) nact = mt + 1;
        const int r = rt + li;
        int off = 0;
        if (r < Mtot) {
            const int c = r / a.KW, k = r - c * a.KW;
            const int kd = k + (c < a.C0 ? delta0 : delta1);
            off = ONESPe + (c - cLo) * planes * XPe + (deint ? ((kd & 1) * XPe + (kd >> 1)) : kd);
        }
        rowoff[mt] = off;
    }
    (void)nact;
    for (int i = tid; i < ONESPe; i += 256) wlds[i] = 0x3F80;          // bf16 1.0
    // a short last k-step reads input positions past the staged window (against zeroed dz): keep every
    // element of the input rows finite from the start (0 * NaN would poison the accumulators)
    for (int i = tid; i < ((nChMax * planes * XPe + 7) & ~7); i += 256) Xs[i] = 0;

    f32x4 acc[MTW][NW];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 xreg[WUN_WGB_XIT];
    f32x4 zreg[ZIT];
    const int TK4 = TK >> 2;
    const float inv_xw4 = 1.0f / (float)XW4, inv_tk4 = 1.0f / (float)TK4;

    // packed per-vector staging state (see the fp32 kernel): bit 31 live | row << 23 | c4 << 16 | LDS element offset
    int xpk[WUN_WGB_XIT];
    int zpk[ZIT];
#pragma unroll
    for (int i = 0; i < WUN_WGB_XIT; ++i) {
        const int f = tid + i * 256;
        const int row = (int)(((float)f + 0.5f) * inv_xw4);
        const int c4 = f - row * XW4;
        const bool rok = row < nCh;
        const int ldsoff = deint ? (row * 2) * XPe + 2 * c4 : row * XPe + 4 * c4;
        xpk[i] = rok ? (int)(0x80000000u | ((unsigned)row << 23) | ((unsigned)c4 << 16) | (unsigned)ldsoff)
                     : (int)((unsigned)c4 << 16);
    }
#pragma unroll
    for (int i = 0; i < ZIT; ++i) {
        const int f = tid + i * 256;
        const int row = (int)(((float)f + 0.5f) * inv_tk4);
        const int c4 = f - row * TK4;
        zpk[i] = row < NG ? (int)(0x80000000u | ((unsigned)row << 23) | ((unsigned)c4 << 16) | (unsigned)(row * ZPe + 4 * c4))
                          : (int)((unsigned)c4 << 16);
    }

    auto load_unit = [&](int u) {
        const int b = u / a.nQT, qt = u - b * a.nQT;
        const int q0 = qt * TK;
        const int tb = (deint ? 2 * q0 : q0) - a.shift;
        const float* base0 = a.src0 + (long long)b * a.bs0;
        const float* base1 = (a.C1 > 0) ? a.src1 + (long long)b * a.bs1 : base0;
        const int e00 = (tb + a.off0) & ~3, e01 = (tb + a.off1) & ~3;
#pragma unroll
        for (int i = 0; i < WUN_WGB_XIT; ++i) {
            int pk = xpk[i];
            asm volatile("" : "+v"(pk));
            int c = cLo + ((pk >> 23) & 255);
            c = c < Ctot ? c : Ctot - 1;
            const bool s1 = c >= a.C0;
            const int xro = s1 ? (c - a.C0) * a.pitch1 : c * a.pitch0;
            int e = (s1 ? e01 : e00) + (((pk >> 16) & 127) << 2);
            const int emax = (s1 ? a.pitch1 : a.pitch0) - 4;
            e = e < 0 ? 0 : (e > emax ? emax : e);
            xreg[i] = *reinterpret_cast<const f32x4*>((s1 ? base1 : base0) + xro + e);
        }
        const float* zb = a.dz + (long long)b * a.dzbs;
        const int qmax = a.dzpitch - 4;
#pragma unroll
        for (int i = 0; i < ZIT; ++i) {
            int pk = zpk[i];
            asm volatile("" : "+v"(pk));
            const int nn = ng * NG + ((pk >> 23) & 255);
            const int zro = (pk < 0 && nn < a.N ? nn : 0) * a.dzpitch;
            int q = q0 + (((pk >> 16) & 127) << 2);
            q = q > qmax ? qmax : q;
            zreg[i] = *reinterpret_cast<const f32x4*>(zb + zro + q);
        }
    };
    auto store_unit = [&](int u) {
        const int b = u / a.nQT, qt = u - b * a.nQT;
        const int q0 = qt * TK;
        const int tb = (deint ? 2 * q0 : q0) - a.shift;
        int nq = a.Tq - q0; if (nq > TK) nq = TK;
        const int span = 4 * XW4;
        const int t00 = ((tb + a.off0) & ~3) - a.off0, t01 = ((tb + a.off1) & ~3) - a.off1;
        const bool xedge = t00 < 0 || t00 + span > a.Tin || (a.C1 > 0 && (t01 < 0 || t01 + span > a.Tin));
#pragma unroll
        for (int i = 0; i < WUN_WGB_XIT; ++i) {
            int pk = xpk[i];
            asm volatile("" : "+v"(pk));
            if (pk < 0) {
                f32x4 v = xreg[i];
                if (xedge) {
                    const bool s1 = cLo + ((pk >> 23) & 255) >= a.C0;
                    const int t0 = (s1 ? t01 : t00) + (((pk >> 16) & 127) << 2);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (t0 + k < 0 || t0 + k >= a.Tin) v[k] = 0.f;
                }
                unsigned short* dstp = Xs + (pk & 0xFFFF);
                if (!deint) {
                    *reinterpret_cast<u32x2*>(dstp) = (u32x2){wb_pack2(v[0], v[1]), wb_pack2(v[2], v[3])};
                } else {
                    *reinterpret_cast<unsigned*>(dstp) = wb_pack2(v[0], v[2]);
                    *reinterpret_cast<unsigned*>(dstp + XPe) = wb_pack2(v[1], v[3]);
                }
            }
        }
        // positions beyond nq are zero in dz, so whatever the input rows hold there contributes nothing
#pragma unroll
        for (int i = 0; i < ZIT; ++i) {
            int pk = zpk[i];
            asm volatile("" : "+v"(pk));
            if (pk < 0) {
                f32x4 v = zreg[i];
                const int c4x = ((pk >> 16) & 127) << 2;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (c4x + k >= nq) v[k] = 0.f;
                *reinterpret_cast<u32x2*>(Zs + (pk & 0xFFFF)) = (u32x2){wb_pack2(v[0], v[1]), wb_pack2(v[2], v[3])};
            }
        }
        (void)b;
    };

    const int nunits = a.B * a.nQT;
    const int u0 = split * a.units_per_split;
    int u1 = u0 + a.units_per_split;
    if (u1 > nunits) u1 = nunits;
    // the dz tile of a short last unit is only written up to TK: clear the tail k-step once if TK is not a multiple of 32
    const int TKr = (TK + 31) & ~31;
    if (TKr != TK)
        for (int i = tid; i < NG * (TKr - TK); i += 256) Zs[(i / (TKr - TK)) * ZPe + TK + i % (TKr - TK)] = 0;
    if (u0 < u1) load_unit(u0);
    for (int u = u0; u < u1; ++u) {
        __syncthreads();
        store_unit(u);
        __syncthreads();
        if (u + 1 < u1) load_unit(u + 1);
        const int qt = u % a.nQT;
        int nq = a.Tq - qt * TK; if (nq > TK) nq = TK;
        const int nsteps = (nq + 31) >> 5;                 // k-steps of 32 positions
        for (int st = 0; st < nsteps; ++st) {
            bf16x8 av[MTW], bv[NW];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                const unsigned short* xp = wlds + rowoff[mt] + st * 32 + 8 * lg;
                u32x4 w = {(unsigned)xp[0] | ((unsigned)xp[1] << 16), (unsigned)xp[2] | ((unsigned)xp[3] << 16),
                           (unsigned)xp[4] | ((unsigned)xp[5] << 16), (unsigned)xp[6] | ((unsigned)xp[7] << 16)};
                av[mt] = __builtin_bit_cast(bf16x8, w);
            }
#pragma unroll
            for (int n = 0; n < NW; ++n)
                bv[n] = *reinterpret_cast<const bf16x8*>(Zs + (n * 16 + li) * ZPe + st * 32 + 8 * lg);
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int n = 0; n < NW; ++n)
                    acc[mt][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mt], bv[n], acc[mt][n], 0, 0, 0);
        }
    }

    if (!a.direct) {
        f32x4* tile = reinterpret_cast<f32x4*>(a.out) +
                      ((((long long)(a.split_base + split) * nMG + mg) * nNG + ng) * (MG * NG / 4)) +
                      wave * (MTW * NW * 64) + lane;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int n = 0; n < NW; ++n) tile[(mt * NW + n) * 64] = acc[mt][n];
        return;
    }
    float* outp = a.out;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            const int col = ng * NG + n * 16 + li;
            if (col >= a.N) continue;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int r = rlo + (wave * MTW + mt) * 16 + lg * 4 + r4;
                if (r < Mtot) {
                    const int c = r / a.KW, k = r - c * a.KW;
                    outp[((long long)k * Ctot + c) * a.N + col] = acc[mt][n][r4];
                } else if (r == Mtot) {
                    outp[(long long)Mtot * a.N + col] = acc[mt][n][r4];
                }
            }
        }
    }
}

template <int MTW, int NW>
static hipError_t wgrad_bf16_launch_t(WgradArgs a, const WgradGeom& g, hipStream_t s) {
    a.nQT = (a.Tq + g.TK - 1) / g.TK;
    const long long units = (long long)a.B * a.nQT;
    a.units_per_split = (int)((units + a.nsplit - 1) / a.nsplit);
    const bool deint = a.loader == LOADER_DEINT;
    const int TKr = (g.TK + 31) & ~31;
    // element pitches: input rows hold 4*XW4 (stride 1) or 2*XW4 per plane (stride 2) staged positions, plus the read
    // overhang of the last k-step (tap + shift + up to 31 positions of a short unit); dz rows 2*TKr + 32 bytes
    const int xneed = (deint ? 2 * g.XW4 : 4 * g.XW4) + 32 + 24;
    const int XPe = (xneed + 3) / 4 * 4 + 4;
    const int ZPe = TKr + 16;
    const int ONESPe = TKr + 32;
    const size_t lds = 2 * ((size_t)ONESPe + (((size_t)g.nChMax * (deint ? 2 : 1) * XPe + 7) & ~(size_t)7) + (size_t)(NW * 16) * ZPe);
    if (lds > 160 * 1024 || (size_t)g.nChMax * (deint ? 2 : 1) * XPe > 65535 || (size_t)(NW * 16) * ZPe > 65535) return hipErrorInvalidValue;
    auto kern = wgrad_bf16_kernel<MTW, NW>;
    static size_t lds_allowed = 64 * 1024;
    if (lds > lds_allowed) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_allowed = lds;
    }
    const long long grid = (long long)g.nMG * g.nNG * a.nsplit;
    char nm[64], tag[160];
    snprintf(nm, sizeof(nm), "wgrad_bf16_kernel<%d, %d>", MTW, NW);
    snprintf(tag, sizeof(tag), "C=%d N=%d T=%d K=%d ld=%d B=%d nsplit=%d grid=%lld", a.C0 + a.C1, a.N, a.Tq, a.KW, a.loader, a.B,
             a.nsplit, grid);
    prof_scope_begin(nm, 2.0 * a.KW * (double)(a.C0 + a.C1) * a.N * (double)a.Tq * a.B, s, tag);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, a, g.nMG, g.nNG, g.TK, XPe, ZPe, g.nChMax, ONESPe, g.XW4);
    prof_scope_end(s);
    return hipGetLastError();
}

hipError_t launch_wgrad_bf16(const WgradArgs& a, hipStream_t s) {
    if ((a.pitch0 & 3) || (a.bs0 & 3) || (reinterpret_cast<uintptr_t>(a.src0) & 15) || a.pitch0 < 4) return hipErrorInvalidValue;
    if (a.C1 > 0 && ((a.pitch1 & 3) || (a.bs1 & 3) || (reinterpret_cast<uintptr_t>(a.src1) & 15) || a.pitch1 < 4)) return hipErrorInvalidValue;
    if ((a.dzpitch & 3) || (a.dzbs & 3) || (reinterpret_cast<uintptr_t>(a.dz) & 15) || a.dzpitch < 4) return hipErrorInvalidValue;
    const WgradGeom g = wgrad_geom(a);
    if ((long long)g.nChMax * g.XW4 > (long long)WUN_WGB_XIT * 256) return hipErrorInvalidValue;
#define WUN_WGB(M, N) if (g.MTW == M && g.NW == N) return wgrad_bf16_launch_t<M, N>(a, g, s);
    WUN_WGB(1, 1) WUN_WGB(1, 2) WUN_WGB(1, 3)
    WUN_WGB(2, 1) WUN_WGB(2, 2) WUN_WGB(2, 3)
    WUN_WGB(4, 1) WUN_WGB(4, 2) WUN_WGB(4, 3)
    WUN_WGB(6, 1) WUN_WGB(6, 2) WUN_WGB(6, 3)
    WUN_WGB(1, 4) WUN_WGB(2, 4) WUN_WGB(4, 4)
    WUN_WGB(1, 5) WUN_WGB(2, 5) WUN_WGB(4, 5)
#undef WUN_WGB
    return hipErrorInvalidValue;
}

}  // namespace wun
