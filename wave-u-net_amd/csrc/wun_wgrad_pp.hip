// gfx950: ping-pong form of the exact-fp32 weight / bias gradient (wgrad_mfma_kernel's arithmetic, tile geometry and
// partial layout; Training.py:77 backward of UnetAudioSeparator.py:97-125).
//
// wgrad_mfma_kernel runs two 256-thread workgroups per CU whose phases coincide: both stage a unit (global loads, LDS
// stores, two barriers) while the matrix pipe idles, then both share it.  Here ONE 512-thread workgroup owns the CU and
// its two wave sets (waves 0-3 / 4-7; wave w and w+4 sit on the same SIMD) alternate BY CONSTRUCTION: in segment k the
// set (k & 1) stages unit k into ITS OWN LDS buffer (loads issued, awaited and written inside the segment: no staging
// registers live across an MFMA phase) while the other set runs the MFMAs of unit k-1 out of its buffer; one s_barrier
// per segment.  The matrix pipe of every SIMD therefore always has exactly one wave in its MFMA stream; staging costs
// only the issue slots it takes from that wave.  Both sets accumulate the SAME output tile over alternating units; at
// the end set 1 hands its accumulators to set 0 through LDS (fixed order: set0 + set1) and set 0 stores the tile --
// tile-major split partial (summed by wgrad_reduce_kernel) or the final layout.
#include "wun_internal.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace wun {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 pp_mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int pp_xcd_block(int bid, int grid) {
    const int per = grid >> 3, rem = grid & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
}

#ifdef WUN_PP_TRACE
// diagnostic builds only (tools/pp_trace.py): per workgroup and wave set {segment start, work done, barrier passed} shader-clock
// stamps of the first 20 segments + the constant 100 MHz clock at entry / exit
#define WUN_PP_TRACE_WGS 1024
#define WUN_PP_TRACE_SEGS 20
__device__ unsigned long long g_pp_trace[WUN_PP_TRACE_WGS * 2 * (4 + 3 * WUN_PP_TRACE_SEGS)];
#endif

#ifndef WUN_PP_MFMA_PRIO
#define WUN_PP_MFMA_PRIO 1       /* wave priority inside an MFMA segment (the stager runs at 0) */
#endif

template <int MTW, int NW>
__global__ __launch_bounds__(512, 2) void wgrad_pp_kernel(WgradArgs a, int nMG, int nNG, int TK, int XP, int ZP,
                                                          int nChMax, int ONESP, int XW4) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int MG = 4 * MTW * 16;
    constexpr int NG = NW * 16;
    constexpr int XIT = WUN_WG_PP_XIT;
    constexpr int ZIT = (NG * 32 + 255) / 256;          // TK/4 <= 32 float4 per dz row
    const bool deint = (a.loader == LOADER_DEINT);
    const int planes = deint ? 2 : 1;
    const int SB = nChMax * planes * XP + NG * ZP;      // floats per set buffer {input rows, dz rows}

    const int tid = threadIdx.x;
    const int stid = tid & 255;                         // thread index inside the wave set
    const int lane = tid & 63;
    const int li = lane & 15, lg = lane >> 4;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int set = wave8 >> 2, wave = wave8 & 3;
    float* Xs = lds + ONESP + set * SB;
    float* Zs = Xs + nChMax * planes * XP;

    int bid = pp_xcd_block((int)blockIdx.x, (int)gridDim.x);
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

    int rowoff[MTW];                                   // LDS float offset of this lane's A row per M tile (0 = ones row)
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
            off = ONESP + set * SB + (c - cLo) * planes * XP + (deint ? ((kd & 1) * XP + (kd >> 1)) : kd);
        }
        rowoff[mt] = off;
    }
    for (int i = tid; i < ONESP; i += 512) lds[i] = 1.f;

    f32x4 acc[MTW][NW];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int TK4 = TK >> 2;
    const float inv_xw4 = 1.0f / (float)XW4, inv_tk4 = 1.0f / (float)TK4;

    // unit-invariant staging state, one packed register per vector (see wgrad_mfma_kernel):
    //   bit 31: this thread stages vector i | row (8 bits) << 23 | c4 (7 bits) << 16 | LDS float offset (16 bits)
    int xpk[XIT];
    int zpk[ZIT];
#pragma unroll
    for (int i = 0; i < XIT; ++i) {
        const int f = stid + i * 256;
        const int row = (int)(((float)f + 0.5f) * inv_xw4);
        const int c4 = f - row * XW4;
        const bool rok = row < nCh;
        const int ldsoff = deint ? (row * 2) * XP + 2 * c4 : row * XP + 4 * c4;
        xpk[i] = rok ? (int)(0x80000000u | ((unsigned)row << 23) | ((unsigned)c4 << 16) | (unsigned)ldsoff)
                     : (int)((unsigned)c4 << 16);
    }
#pragma unroll
    for (int i = 0; i < ZIT; ++i) {
        const int f = stid + i * 256;
        const int row = (int)(((float)f + 0.5f) * inv_tk4);
        const int c4 = f - row * TK4;
        zpk[i] = row < NG ? (int)(0x80000000u | ((unsigned)row << 23) | ((unsigned)c4 << 16) | (unsigned)(row * ZP + 4 * c4))
                          : (int)((unsigned)c4 << 16);
    }

    // global -> registers -> LDS of one unit, all inside the calling set's staging segment
    auto stage_unit = [&](int u) {
        f32x4 xreg[XIT];
        f32x4 zreg[ZIT];
        const int b = u / a.nQT, qt = u - b * a.nQT;
        const int q0 = qt * TK;
        const int tb = (deint ? 2 * q0 : q0) - a.shift;
        const float* base0 = a.src0 + (long long)b * a.bs0;
        const float* base1 = (a.C1 > 0) ? a.src1 + (long long)b * a.bs1 : base0;
        const int e00 = (tb + a.off0) & ~3, e01 = (tb + a.off1) & ~3;   // uniform element shift per source
#pragma unroll
        for (int i = 0; i < XIT; ++i) {
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
        int nq = a.Tq - q0; if (nq > TK) nq = TK;
        const int span = 4 * XW4;
        const int t00 = e00 - a.off0, t01 = e01 - a.off1;
        const bool xedge = t00 < 0 || t00 + span > a.Tin || (a.C1 > 0 && (t01 < 0 || t01 + span > a.Tin));
#pragma unroll
        for (int i = 0; i < XIT; ++i) {
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
                float* dstp = Xs + (pk & 0xFFFF);
                if (!deint) {
                    *reinterpret_cast<f32x4*>(dstp) = v;
                } else {
                    *reinterpret_cast<float2*>(dstp) = make_float2(v[0], v[2]);
                    *reinterpret_cast<float2*>(dstp + XP) = make_float2(v[1], v[3]);
                }
            }
        }
        const bool zedge = nq < TK;
#pragma unroll
        for (int i = 0; i < ZIT; ++i) {
            int pk = zpk[i];
            asm volatile("" : "+v"(pk));
            if (pk < 0) {
                f32x4 v = zreg[i];
                if (zedge) {
                    const int c4x = ((pk >> 16) & 127) << 2;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (c4x + k >= nq) v[k] = 0.f;
                }
                *reinterpret_cast<f32x4*>(Zs + (pk & 0xFFFF)) = v;
            }
        }
    };

    // Branch-free MFMA stream of one unit.  Every instruction a SIMD issues beside its MFMAs costs matrix-pipe time
    // (measured: ~10 cycles per LDS read, whichever wave issues it), so the stream is built for the FEWEST instructions:
    // the k index of an MFMA is a summation index, so lane group lg of a block of four k-steps (16 positions) takes the
    // positions 4 lg + s, s = 0..3 -- FOUR CONSECUTIVE floats of its A row and of its dz row: one 16-byte LDS read per
    // dz column tile, two 8-byte-pair reads (ds_read2_b32; the tap shift leaves A rows only 4-byte aligned) per A row
    // tile and block, all at compile-time offsets from per-tile base addresses, issued one block ahead.
    // Dead tiles of the ragged last row group read the ones row and are never stored; positions past the end of a row
    // multiply zero-filled dz.
    auto mfma_unit = [&](int u) {
        const int qt = u % a.nQT;
        int nq = a.Tq - qt * TK; if (nq > TK) nq = TK;
        const int nblk = (nq + 15) >> 4;
        const float* ap[MTW];
        const float* bp[NW];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) ap[mt] = lds + rowoff[mt] + 4 * lg;
#pragma unroll
        for (int n = 0; n < NW; ++n) bp[n] = Zs + (n * 16 + li) * ZP + 4 * lg;
        float av[2][MTW][4];
        f32x4 bv[2][NW];
        auto ldblk = [&](int buf) {
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int e = 0; e < 4; ++e) av[buf][mt][e] = ap[mt][e];
#pragma unroll
            for (int n = 0; n < NW; ++n) bv[buf][n] = *reinterpret_cast<const f32x4*>(bp[n]);
        };
        ldblk(0);
        constexpr int NM = 4 * MTW * NW;                 // MFMAs per block
        constexpr int NR = 2 * MTW + NW;                 // LDS reads per block
        for (int blk = 0; blk < nblk; blk += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h == 1 && blk + 1 >= nblk) break;
                const int inc = (blk + h + 1 < nblk) ? 16 : 0;   // (the last block re-reads itself: unused)
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) ap[mt] += inc;
#pragma unroll
                for (int n = 0; n < NW; ++n) bp[n] += inc;
                ldblk(h ^ 1);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                        for (int n = 0; n < NW; ++n) acc[mt][n] = pp_mfma16(av[h][mt][s], bv[h][n][s], acc[mt][n]);
                // spread: the address increments + reads of the next block over the first MFMAs of this one
                constexpr int PER = NM / (NR + 1) > 0 ? NM / (NR + 1) : 1;
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NM - PER * NR > 0 ? NM - PER * NR : 0, 0);
            }
        }
    };

    const int nunits = a.B * a.nQT;
    const int u0 = split * a.units_per_split;
    int u1 = u0 + a.units_per_split;
    if (u1 > nunits) u1 = nunits;
    const int nu = u1 > u0 ? u1 - u0 : 0;

#ifdef WUN_PP_TRACE
    const bool tr_on = (tid & 255) == 0 && blockIdx.x < WUN_PP_TRACE_WGS;
    unsigned long long* trp = g_pp_trace + ((size_t)(blockIdx.x < WUN_PP_TRACE_WGS ? blockIdx.x : 0) * 2 + set) * (4 + 3 * WUN_PP_TRACE_SEGS);
    if (tr_on) { trp[0] = wall_clock64(); trp[1] = __builtin_readcyclecounter(); trp[3] = (unsigned long long)nu; }
#define PP_STAMP(seg, i) do { if (tr_on && (seg) < WUN_PP_TRACE_SEGS) trp[4 + 3 * (seg) + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define PP_STAMP(seg, i) do { } while (0)
#endif
    // segment k: set (k & 1) stages unit k, the other set multiplies unit k - 1
    for (int seg = 0; seg <= nu; ++seg) {
        PP_STAMP(seg, 0);
        if ((seg & 1) == set) {
#ifdef WUN_PP_TRACE
            switch ((a.ablate >> 2) & 3) { case 1: __builtin_amdgcn_s_setprio(1); break; case 2: __builtin_amdgcn_s_setprio(2); break;
                                           case 3: __builtin_amdgcn_s_setprio(3); break; default: __builtin_amdgcn_s_setprio(0); }
#endif
#ifdef WUN_PP_TRACE
            if (seg < nu && (!(a.ablate & 16) || seg < 2)) stage_unit(u0 + seg);
#else
            if (seg < nu) stage_unit(u0 + seg);
#endif
        } else if (seg >= 1) {
#ifdef WUN_PP_TRACE
            if (a.ablate & 32) { PP_STAMP(seg, 1); __syncthreads(); PP_STAMP(seg, 2); continue; }
            switch (a.ablate & 3) { case 1: __builtin_amdgcn_s_setprio(1); break; case 2: __builtin_amdgcn_s_setprio(2); break;
                                    case 3: __builtin_amdgcn_s_setprio(3); break; default: __builtin_amdgcn_s_setprio(0); }
#else
            __builtin_amdgcn_s_setprio(WUN_PP_MFMA_PRIO);
#endif
            mfma_unit(u0 + seg - 1);
#ifndef WUN_PP_TRACE
            __builtin_amdgcn_s_setprio(0);
#endif
        }
        PP_STAMP(seg, 1);
        __syncthreads();
        PP_STAMP(seg, 2);
    }
#ifdef WUN_PP_TRACE
    if (tr_on) trp[2] = wall_clock64();
#endif

    // set 1 -> set 0 through LDS (register order: one f32x4 per lane and tile), fixed order set0 + set1
    f32x4* cb = reinterpret_cast<f32x4*>(lds) + (wave * (MTW * NW)) * 64 + lane;
    if (nu > 1) {
        if (set == 1) {
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int n = 0; n < NW; ++n) cb[(mt * NW + n) * 64] = acc[mt][n];
        }
        __syncthreads();
        if (set == 0) {
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int n = 0; n < NW; ++n) acc[mt][n] += cb[(mt * NW + n) * 64];
        }
    }
    if (set != 0) return;

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
        if (mt >= nact) continue;
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
static hipError_t wgrad_pp_launch_t(WgradArgs a, const WgradGeom& g, hipStream_t s) {
    a.nQT = (a.Tq + g.TK - 1) / g.TK;
    const long long units = (long long)a.B * a.nQT;
    a.units_per_split = (int)((units + a.nsplit - 1) / a.nsplit);
    auto kern = wgrad_pp_kernel<MTW, NW>;
    static size_t lds_allowed = 64 * 1024;
    if (g.lds > lds_allowed) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
        if (e != hipSuccess) return e;
        lds_allowed = g.lds;
    }
    const long long grid = (long long)g.nMG * g.nNG * a.nsplit;
    char nm[64];
    snprintf(nm, sizeof(nm), "wgrad_pp_kernel<%d, %d>", MTW, NW);
    char tag[160];
    snprintf(tag, sizeof(tag), "C=%d N=%d T=%d K=%d ld=%d B=%d nsplit=%d grid=%lld", a.C0 + a.C1, a.N, a.Tq, a.KW,
             a.loader, a.B, a.nsplit, grid);
    prof_scope_begin(nm, 2.0 * a.KW * (double)(a.C0 + a.C1) * a.N * (double)a.Tq * a.B, s, tag);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), g.lds, s, a, g.nMG, g.nNG, g.TK, g.XP, g.ZP, g.nChMax,
                       g.ONESP, g.XW4);
    prof_scope_end(s);
    return hipGetLastError();
}

hipError_t launch_wgrad_pp(const WgradArgs& a_in, const WgradGeom& g, hipStream_t s) {
    WgradArgs a = a_in;
#ifdef WUN_PP_TRACE
    if (const char* e = getenv("WUN_PP_PRIO")) a.ablate = atoi(e);      // bits 0-1: MFMA segment priority, bits 2-3: staging segment priority
#endif
#define WUN_WGP(M, N) if (g.MTW == M && g.NW == N) return wgrad_pp_launch_t<M, N>(a, g, s);
    WUN_WGP(1, 1) WUN_WGP(1, 2) WUN_WGP(1, 3)
    WUN_WGP(2, 1) WUN_WGP(2, 2) WUN_WGP(2, 3)
    WUN_WGP(4, 1) WUN_WGP(4, 2) WUN_WGP(4, 3)
    WUN_WGP(6, 1) WUN_WGP(6, 2) WUN_WGP(6, 3)
    WUN_WGP(1, 4) WUN_WGP(2, 4) WUN_WGP(4, 4) WUN_WGP(6, 4)
    WUN_WGP(1, 5) WUN_WGP(2, 5) WUN_WGP(4, 5) WUN_WGP(6, 5)
#undef WUN_WGP
    return hipErrorInvalidValue;
}

}  // namespace wun

#ifdef WUN_PP_TRACE
extern "C" int wun_dbg_pp_trace_read(unsigned long long* host, int nwords) {
    const int cap = (int)(sizeof(wun::g_pp_trace) / sizeof(unsigned long long));
    if (nwords > cap) nwords = cap;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(wun::g_pp_trace), (size_t)nwords * sizeof(unsigned long long)) != hipSuccess) return -2;
    return nwords;
}
#endif
