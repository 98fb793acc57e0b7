// gfx950: "register-window" form of the exact-fp32 weight / bias gradient (Training.py:77 backward of the convs of
// UnetAudioSeparator.py:97-125):   dW[k][c][n] = sum_{b,q} x[b][c][S q + k - shift] * dz[b][n][q],   db[n] = sum dz.
//
// What the measurements of round 4 say about the fp32 matrix pipe (tools/probes/mfma_power_probe.hip): a
// dense v_mfma_f32_16x16x4_f32 stream fed by ALIGNED 16-byte LDS reads runs at 33.2 cycles per MFMA from one wave per
// SIMD (32.3 from two); wgrad_mfma_kernel's stream -- one 4-byte LDS read per operand and k-step, rows at arbitrary
// 4-byte alignment -- needs 36.5-37.6 whoever else shares the SIMD, and every VALU instruction of the staging code
// (address arithmetic, masks, LDS stores) is issued at the expense of an MFMA.  So this kernel removes the instructions:
//
//  * MFMA rows = RC channels x G tap groups (taps k and k + 8), columns = 16 output channels, and ONE ACCUMULATOR TILE
//    PER TAP k: the A operand of (tap k, k-step s) is x[c][S (q0 + 16 blk + 4 lg + s) + k] -- for a lane a fixed element
//    S s + k of a WINDOW of 3 S + KT consecutive floats that it loads once per block of 16 positions with two to four
//    aligned 16-byte reads and then uses, straight out of the registers, for all KT taps x 4 k-steps x NW column tiles
//    (the k index of an MFMA is a summation index: lane group lg takes positions 4 lg + s).  dz: one 16-byte read per
//    column tile and block.  7 LDS reads per 96 MFMAs instead of 48 (K = 15, three column tiles).
//  * the input window and the dz tile of a unit go global -> LDS by DMA (global_load_lds, 16 bytes per lane, source at
//    any 4-byte alignment -- probed: tools/probes/dma_align_probe): the LDS image starts exactly at the first sample the unit
//    needs, so crop offsets / 'same' padding shifts cost nothing; no staging registers, no LDS stores, per unit a few
//    DMA instructions with unit-invariant per-lane offsets; edge units clamp their addresses and zero the samples
//    outside [0, Tin) / beyond Tq in LDS after landing.  Stride-2 convs need no de-interleave (window element 2 s + k).
//  * a workgroup is 4 waves per column group -- one per SIMD, always: K = 15: 24 channels = three row tiles of 8 channels
//    x {taps k, k + 8}; waves 0-2 take the taps {0-3, 6, 7} (+8) of one row tile each, wave 3 the taps {4, 5} (+8) of all
//    three (six accumulator tiles per column tile for every wave; wave 3's windows are the two aligned vectors 4..11); K = 5: 64 channels = four row tiles of 16, all taps.
//  * the bias gradient rides in a dead MFMA row (tap 15 of a 15-tap filter / a channel past Cin): its A operand is 1.
//  * split partials are written in the FINAL layout ([K][Cin][Cout] + bias) and summed by a flat, fixed-order pairwise
//    reduction; both parts of a down level (decimated + window positions) share it whatever their tiles.
#include "wun_internal.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace wun {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

// wave-uniform values the compiler cannot prove uniform -> SGPRs (inline-asm "s" operands)
__device__ __forceinline__ unsigned win_sgpr(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ const float* win_sgpr_ptr(const float* q) {
    const unsigned long long v = (unsigned long long)q;
    const unsigned lo = win_sgpr((unsigned)v), hi = win_sgpr((unsigned)(v >> 32));
    return (const float*)(((unsigned long long)hi << 32) | lo);
}
// one LDS-DMA instruction in its scalar-base form: 64 lanes x 16 bytes from sbase + voff[lane] to LDS bytes m0 + 16 lane
// (M0 is written inside the statement and NOT declared as a clobber: M0 is a reserved register for LLVM's AMDGPU backend,
//  naming it in a clobber list is diagnosed as "may lead to undefined behaviour" (-Winline-asm) -- the backend never keeps
//  a live value in M0 across an asm statement, it re-materialises M0 immediately before every instruction of its own that
//  reads it (checked in the disassembly of the __builtin_amdgcn_global_load_lds calls of the edge / slow paths below:
//  each is preceded by its own s_mov_b32 m0).  tools/m0_check.sh verifies that pairing on the disassembly: 1378 LDS-DMA
//  instructions in this file, none without a fresh M0 write.)
__device__ __forceinline__ void win_dma16(unsigned m0v, unsigned voff, const float* sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(voff), "s"(sbase) : "memory");
}

__device__ __forceinline__ int win_xcd_block(int bid, int grid) {
    const int per = grid >> 3, rem = grid & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
}

// granules (16 bytes) a thread may stage per unit: X rows x live granules + dz rows x TK/4 over 64 W threads
#define WUN_WIN_XIT 3
#define WUN_WIN_ZIT 4


struct WinParams {
    int RTW, CGW;        // row tiles x column groups per workgroup (waves = RTW * CGW)
    int nMG, nNG;        // workgroups along rows / columns
    int TK;              // positions per unit (multiple of 16)
    int XP, ZP;          // LDS row pitches (floats, multiples of 4)
    int XGL;             // live granules per X row
    int GOFF;            // float offset between tap groups in the X row (8)
    int zoff;            // float offset of the dz rows inside a buffer (X region padded to whole 1 KiB DMA blocks)
    int bufFloats;       // floats per LDS buffer {X rows, dz rows}
    long long pstride;   // floats per split in the partial buffer
};

// which (row tiles, taps) a wave owns: K = 15 -- waves 0-2: one row tile, taps {0, 1, 2, 3, 6, 7}; wave 3: three row tiles,
// taps {4, 5}; K = 5 -- one row tile, all five taps
struct WinTapsA { static constexpr int JR = 1, NK = 6; static constexpr int k(int i) { return i < 4 ? i : i + 2; } };
struct WinTapsB { static constexpr int JR = 3, NK = 2; static constexpr int k(int i) { return 4 + i; } };
struct WinTaps5 { static constexpr int JR = 1, NK = 5; static constexpr int k(int i) { return i; } };

// K15 = true: 15-tap scheme (G = 2 tap groups, 8 channels per row tile, 3 row tiles + the tap split above);
// K15 = false: 5-tap scheme (16 channels per row tile, 4 row tiles).  NA = accumulator tiles per column tile and wave.
template <bool K15, int NW, int S>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu((K15 ? NW <= 2 : NW <= 4) ? 3 : 2))) void wgrad_win_kernel(WgradArgs a, WinParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int G = K15 ? 2 : 1;
    constexpr int RC = 16 / G;                          // channels per row tile
    constexpr int NA = K15 ? 6 : 5;
    const int W = 4 * p.CGW;
    const int nthr = (int)win_sgpr((unsigned)(64 * W));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = wave & 3, cg = wave >> 2;            // K15: rt == 3 is the wave of taps 4-5 over the three row tiles
    const bool roleB = K15 && rt == 3;

    int bid = win_xcd_block((int)blockIdx.x, (int)gridDim.x);
    const int ng = bid % p.nNG; bid /= p.nNG;
    const int mg = bid % p.nMG;
    const int split = bid / p.nMG;

    const int Ctot = a.C0 + a.C1;
    const int cWG = mg * p.RTW * RC;                    // first channel of the workgroup's X tile
    const int nXrows = p.RTW * RC, nZrows = p.CGW * NW * 16;
    const int cT = cWG + (roleB ? 0 : rt) * RC;         // first channel of this wave's (first) row tile
    const int n0 = (ng * p.CGW + cg) * NW * 16;         // first column of this wave
    const bool wave_live = cT < Ctot + (G == 1 ? 1 : 0) && n0 < a.N;      // (G == 1: a tile of dead rows carries the bias)
    const int zoff = p.zoff;

    // ---- unit-invariant DMA state: per granule its source offset (elements, from the unit's first sample of the row)
    // and which source; granule f of the tile lands at LDS float 4 f (lane-linear) ----
    const int XG = p.XP >> 2, ZG = p.ZP >> 2;
    // Unit-invariant DMA state: one byte offset per 16-byte slot of this thread.  Every lane fetches -- lanes of pad slots
    // re-fetch a valid granule into their pad -- from ONE uniform base pointer + these offsets: global_load_lds in its
    // scalar-base form, no address VALU, no exec masks (interior units of single-source tiles; the other units clamp
    // per lane, below).  Slot f of a region lands at LDS float 4 f (lane-linear).
    unsigned xb[WUN_WIN_XIT], zbo[WUN_WIN_ZIT];
    unsigned xs1 = 0;                                   // bit i: slot i of this thread reads the second source
    const float inv_xg = 1.0f / (float)XG, inv_zg = 1.0f / (float)ZG;
    // slot -> (row, granule) of the X / dz region
    auto xslot = [&](int f, int& row, int& g) __attribute__((always_inline)) {
        row = (int)(((float)f + 0.5f) * inv_xg);
        g = f - row * XG;
        if (g < 0) { --row; g += XG; } else if (g >= XG) { ++row; g -= XG; }
    };
    auto zslot = [&](int f, int& row, int& g) __attribute__((always_inline)) {
        row = (int)(((float)f + 0.5f) * inv_zg);
        g = f - row * ZG;
        if (g < 0) { --row; g += ZG; } else if (g >= ZG) { ++row; g -= ZG; }
    };
    // source row of X-tile row `row`: element offset of its first sample from the source base; s1 = second source
    auto xrow_src = [&](int row, bool& s1) __attribute__((always_inline)) {
        int c = cWG + row;
        c = c < Ctot ? c : Ctot - 1;                     // dead channels: any valid row (results never stored)
        s1 = c >= a.C0;
        return s1 ? (c - a.C0) * a.pitch1 : c * a.pitch0;
    };
#pragma unroll
    for (int i = 0; i < WUN_WIN_XIT; ++i) {
        int row, g;
        xslot(tid + i * nthr, row, g);
        bool s1;
        const int ro = xrow_src(row < nXrows ? row : nXrows - 1, s1);
        xb[i] = 4u * (unsigned)(ro + 4 * (g < p.XGL ? g : p.XGL - 1));
        xs1 |= (s1 ? 1u : 0u) << i;
    }
#pragma unroll
    for (int i = 0; i < WUN_WIN_ZIT; ++i) {
        int row, g;
        zslot(tid + i * nthr, row, g);
        int n = ng * p.CGW * NW * 16 + (row < nZrows ? row : nZrows - 1);
        n = n < a.N ? n : a.N - 1;                       // padded columns: any valid row
        zbo[i] = 4u * (unsigned)(n * a.dzpitch + 4 * (g < (p.TK >> 2) ? g : (p.TK >> 2) - 1));
    }
    // single-source tile: all of the workgroup's channels come from src0 or all from src1
    const int cLast = (cWG + nXrows - 1 < Ctot ? cWG + nXrows - 1 : Ctot - 1);
    const bool one_src = cWG >= a.C0 || cLast < a.C0;
    const bool tile_s1 = cWG >= a.C0;
    const unsigned lds_base = (unsigned)(unsigned long long)(lds_void_t*)lds;
    const int nxslots = nXrows * XG, nzslots = nZrows * ZG;
    const int nxpad = (int)win_sgpr((unsigned)((nxslots + 63) & ~63)), nzpad = (int)win_sgpr((unsigned)((nzslots + 63) & ~63));

    // (the rare paths -- edge units, tiles that straddle the two sources -- read the launch arguments from the kernarg
    //  segment through an opaque pointer, so those fields do not occupy scalar registers across the MFMA loop)
    auto slow_args = [&]() __attribute__((always_inline)) {
        const char* kp = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        return (const WgradArgs*)kp;
    };
    // DMA instructions of this wave per region and unit (wave-uniform; slot rounds of 64 W lanes)
    const int nxi = (int)win_sgpr((unsigned)((nxpad - wave * 64 + nthr - 1) / nthr));
    const int nzi = (int)win_sgpr((unsigned)((nzpad - wave * 64 + nthr - 1) / nthr));
    const unsigned m0w = win_sgpr(lds_base + 16u * (unsigned)(wave * 64));
    const unsigned nthr16 = win_sgpr(16u * (unsigned)nthr);
    const int t4 = 4 * p.XGL;
    const bool shift_edge = a.shift > 0;
    // is unit (b, qt) an edge unit (samples outside [0, Tin) / positions beyond Tq)?
    auto unit_edge = [&](int qt) __attribute__((always_inline)) {
        const int q0 = qt * p.TK, t0 = S * q0 - a.shift;
        return (shift_edge && qt == 0) || t0 + t4 > a.Tin || q0 + p.TK > a.Tq;
    };
    // one unit = p.TK output positions of one excerpt
    auto unit_fast = [&](int qt) __attribute__((always_inline)) { return !unit_edge(qt); };
    // interior unit of a single-source tile
    auto dma_fast = [&](int b, int qt, int bo) __attribute__((always_inline)) {
        const int q0 = qt * p.TK;
        const int t0 = S * q0 - a.shift;                 // time index of the first staged sample
        {
            const unsigned m0x = win_sgpr(m0w + 4u * (unsigned)bo);
            if (one_src) {
                const float* xbs = win_sgpr_ptr(tile_s1 ? a.src1 + (long long)b * a.bs1 + a.off1 + t0
                                                        : a.src0 + (long long)b * a.bs0 + a.off0 + t0);
#pragma unroll
                for (int i = 0; i < WUN_WIN_XIT; ++i)
                    if (i < nxi) win_dma16(m0x + (unsigned)i * nthr16, xb[i], xbs);
            } else {
                // the tile straddles the two sources of a channel concat: every instruction once per source, lanes split
                const float* xb0 = win_sgpr_ptr(a.src0 + (long long)b * a.bs0 + a.off0 + t0);
                const float* xb1 = win_sgpr_ptr(a.src1 + (long long)b * a.bs1 + a.off1 + t0);
#pragma unroll
                for (int i = 0; i < WUN_WIN_XIT; ++i)
                    if (i < nxi) {
                        if ((xs1 >> i) & 1u) win_dma16(m0x + (unsigned)i * nthr16, xb[i], xb1);
                        else win_dma16(m0x + (unsigned)i * nthr16, xb[i], xb0);
                    }
            }
            const float* zbs = win_sgpr_ptr(a.dz + (long long)b * a.dzbs + q0);
            const unsigned m0z = win_sgpr(m0w + 4u * (unsigned)(bo + zoff));
#pragma unroll
            for (int i = 0; i < WUN_WIN_ZIT; ++i)
                if (i < nzi) win_dma16(m0z + (unsigned)i * nthr16, zbo[i], zbs);
        }
    };
    // edge units / tiles that straddle the two sources: per-lane source, addresses clamped into the source row
    auto dma_slow = [&](int b, int qt, int bo) __attribute__((always_inline)) {
        const int q0 = qt * p.TK;
        const int t0 = S * q0 - a.shift;
        const WgradArgs& sa = *slow_args();
        float* buf = lds + bo;
#pragma unroll
        for (int i = 0; i < WUN_WIN_XIT; ++i) {
            if (i * nthr + wave * 64 < nxslots) {        // wave-uniform
                int row, g, fo = tid + i * nthr;
                asm volatile("" : "+v"(fo));             // (keeps the slot arithmetic inside the unit loop: no hoisted copies)
                xslot(fo, row, g);
                if (row < nXrows && g < p.XGL) {
                    int c = cWG + row;
                    c = c < Ctot ? c : Ctot - 1;
                    const bool s1 = c >= sa.C0;
                    const int ro = s1 ? (c - sa.C0) * sa.pitch1 : c * sa.pitch0;
                    const int pitch = s1 ? sa.pitch1 : sa.pitch0, off = s1 ? sa.off1 : sa.off0;
                    int er = 4 * g + off + t0;                                 // element index inside the source row
                    er = er < 0 ? 0 : (er > pitch - 4 ? pitch - 4 : er);
                    const float* base = s1 ? sa.src1 + (long long)b * sa.bs1 : sa.src0 + (long long)b * sa.bs0;
                    __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(base + ro + er),
                                                     (lds_void_t*)(buf + (i * nthr + wave * 64) * 4), 16, 0, 0);
                }
            }
        }
        const float* zb = sa.dz + (long long)b * sa.dzbs;
#pragma unroll
        for (int i = 0; i < WUN_WIN_ZIT; ++i) {
            if (i * nthr + wave * 64 < nzslots) {
                int row, g, fo = tid + i * nthr;
                asm volatile("" : "+v"(fo));
                zslot(fo, row, g);
                if (row < nZrows && g < (p.TK >> 2)) {
                    int n = ng * p.CGW * NW * 16 + row;
                    n = n < sa.N ? n : sa.N - 1;
                    int er = q0 + 4 * g;
                    er = er > sa.dzpitch - 4 ? sa.dzpitch - 4 : er;
                    __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(zb + (long long)n * sa.dzpitch + er),
                                                     (lds_void_t*)(buf + zoff + (i * nthr + wave * 64) * 4), 16, 0, 0);
                }
            }
        }
    };
    // edge units, after landing: zero what lies outside [0, Tin) / beyond Tq (clamped granules hold other samples)
    auto zero_fix = [&](int qt, int bo) __attribute__((always_inline)) {
        const WgradArgs& a = *slow_args();               // (shadows the kernel argument inside this rare path)
        float* buf = lds + bo;
        const int q0 = qt * p.TK;
        const int t0 = S * q0 - a.shift;
        if (t0 < 0 || t0 + 4 * p.XGL > a.Tin) {
#pragma unroll
            for (int i = 0; i < WUN_WIN_XIT; ++i) {
                int f = tid + i * nthr;
                asm volatile("" : "+v"(f));
                int row, g;
                xslot(f, row, g);
                if (f < nxslots && row < nXrows && g < p.XGL) {
                    // the granule was fetched from a clamped position: re-derive each element
                    int c = cWG + row;
                    c = c < Ctot ? c : Ctot - 1;
                    const bool s1 = c >= a.C0;
                    const int pitch = s1 ? a.pitch1 : a.pitch0, off = s1 ? a.off1 : a.off0;
                    const int er0 = off + t0 + 4 * g;
                    const int erc = er0 < 0 ? 0 : (er0 > pitch - 4 ? pitch - 4 : er0);
                    f32x4 v = *reinterpret_cast<f32x4*>(buf + 4 * f);
                    f32x4 w;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int t = t0 + 4 * g + k;               // wanted time index
                        const int j = er0 + k - erc;                 // where it sits in the fetched granule (if at all)
                        float x = 0.f;
                        if (t >= 0 && t < a.Tin && j >= 0 && j < 4) x = j == 0 ? v[0] : j == 1 ? v[1] : j == 2 ? v[2] : v[3];
                        w[k] = x;
                    }
                    *reinterpret_cast<f32x4*>(buf + 4 * f) = w;
                }
            }
        }
        if (q0 + p.TK > a.Tq) {
#pragma unroll
            for (int i = 0; i < WUN_WIN_ZIT; ++i) {
                int f = tid + i * nthr;
                asm volatile("" : "+v"(f));
                int row, g;
                zslot(f, row, g);
                if (f < nzslots && row < nZrows && g < (p.TK >> 2)) {
                    const int er0 = q0 + 4 * g;
                    const int erc = er0 > a.dzpitch - 4 ? a.dzpitch - 4 : er0;
                    f32x4 v = *reinterpret_cast<f32x4*>(buf + zoff + 4 * f);
                    f32x4 w;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int q = q0 + 4 * g + k;
                        const int j = er0 + k - erc;
                        float x = 0.f;
                        if (q < a.Tq && j >= 0 && j < 4) x = j == 0 ? v[0] : j == 1 ? v[1] : j == 2 ? v[2] : v[3];
                        w[k] = x;
                    }
                    *reinterpret_cast<f32x4*>(buf + zoff + 4 * f) = w;
                }
            }
        }
    };

    f32x4 acc[NA][NW];
#pragma unroll
    for (int k = 0; k < NA; ++k)
#pragma unroll
        for (int n = 0; n < NW; ++n) acc[k][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // lane's LDS read positions: X row = channel li % RC of a row tile, tap group li / RC, positions 4 lg ..;
    // dz row = column li of each column tile
    const int lc = li % RC, lgp = li / RC;
    const int xrd = lc * p.XP + p.GOFF * lgp + 4 * S * lg;          // (+ row tile * RC * XP)
    const int zrd = zoff + (cg * NW * 16 + li) * p.ZP + 4 * lg;
    // the bias rides in a dead row: (K15) tap group 1 of tap 7 = tap 15; (K5) a channel past Cin, tap 0.  Its A operand is 1.
    const bool ones_lane = K15 ? (lgp == 1) : (cT + lc >= Ctot);

    // One unit for a wave that owns JR row tiles x the taps TAPS::k[0 .. NK) (accumulator tile j * NK + kk): per block of
    // 16 positions JR windows (16-byte reads from the aligned float below the first tap) + NW dz reads, registers
    // double-buffered one block ahead.
    auto mfma_unit = [&](int bo, auto taps, auto&& mid) __attribute__((always_inline)) {
        using TP = decltype(taps);
        constexpr int JR = TP::JR, NK = TP::NK;
        constexpr int AS = TP::k(0) & ~3;                            // aligned start of the window
        constexpr int NRD = (3 * S + TP::k(NK - 1) - AS + 4) / 4;    // 16-byte reads per window
        constexpr int KBIAS = K15 ? 7 : 0;                           // tap whose dead rows carry the bias
        const float* xp = lds + bo + xrd + (JR == 1 ? rt * RC * p.XP : 0) + AS;
        const float* zp = lds + bo + zrd;
        f32x4 wv[2][JR][NRD];
        f32x4 bv[2][NW];
        auto ldblk = [&](int bsel, int blk) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < JR; ++j)
#pragma unroll
                for (int q = 0; q < NRD; ++q)
                    wv[bsel][j][q] = *reinterpret_cast<const f32x4*>(xp + j * RC * p.XP + 16 * S * blk + 4 * q);
#pragma unroll
            for (int n = 0; n < NW; ++n) bv[bsel][n] = *reinterpret_cast<const f32x4*>(zp + n * 16 * p.ZP + 16 * blk);
        };
        const int nblk = p.TK >> 4;
        ldblk(0, 0);
        for (int blk = 0; blk < nblk; blk += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h == 1 && blk + 1 >= nblk) break;
                const int nb = blk + h + 1 < nblk ? blk + h + 1 : blk + h;      // (the last block re-reads itself: unused)
                ldblk(h ^ 1, nb);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    // the next unit's DMA is issued from INSIDE the MFMA stream: instructions of a wave that is streaming
                    // MFMAs issue in their shadow; the same instructions from a wave that is not cost one MFMA slot of the
                    // SIMD's other waves each (measured: 5-8k cycles per unit as a separate phase)
                    if (s == 1 && h == 0 && blk == 0) mid();
#pragma unroll
                    for (int j = 0; j < JR; ++j)
#pragma unroll
                        for (int kk = 0; kk < NK; ++kk) {
                            const int e = S * s + TP::k(kk) - AS;
                            float av = wv[h][j][e >> 2][e & 3];
                            if (TP::k(kk) == KBIAS) av = ones_lane ? 1.0f : av;
#pragma unroll
                            for (int n = 0; n < NW; ++n)
                                acc[j * NK + kk][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[h][n][s], acc[j * NK + kk][n], 0, 0, 0);
                        }
                }
            }
        }
    };

    const int nunits = a.B * a.nQT;
    const int u0 = split * a.units_per_split;
    int u1 = u0 + a.units_per_split;
    if (u1 > nunits) u1 = nunits;

#define WT_STAMP(k, i) do { } while (0)
    // (the unit loop is instantiated once per wave role: one MFMA stream, one register allocation problem each)
    auto unit_loop = [&](auto taps) __attribute__((always_inline)) {
        if (u0 >= u1) return;
        int ub = u0 / a.nQT, uq = u0 - ub * a.nQT;                      // unit being multiplied: (excerpt, position tile)
        if (unit_fast(uq)) dma_fast(ub, uq, 0); else dma_slow(ub, uq, 0);
        for (int u = u0; u < u1; ++u) {
            const int cur = ((u - u0) & 1) * p.bufFloats;
            WT_STAMP(u - u0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's granules of unit u have landed
            __syncthreads();                                            // ... everybody's; the other buffer is free
            WT_STAMP(u - u0, 1);
            if (unit_edge(uq)) {                                        // uniform
                zero_fix(uq, cur);
                __syncthreads();
            }
            int nq = uq + 1, nb = ub;
            if (nq == a.nQT) { nq = 0; ++nb; }
            const bool more = u + 1 < u1;
            const bool nfast = more && unit_fast(nq) && wave_live;      // -> issued from inside the MFMA stream
            if (more && !nfast) { if (unit_fast(nq)) dma_fast(nb, nq, p.bufFloats - cur); else dma_slow(nb, nq, p.bufFloats - cur); }
            WT_STAMP(u - u0, 2);
            if (wave_live) mfma_unit(cur, taps, [&]() __attribute__((always_inline)) { if (nfast) dma_fast(nb, nq, p.bufFloats - cur); });
            WT_STAMP(u - u0, 3);
            ub = nb; uq = nq;
        }
    };
    if constexpr (K15) {
        if (roleB) unit_loop(WinTapsB{});
        else unit_loop(WinTapsA{});
    } else {
        unit_loop(WinTaps5{});
    }
    if (!wave_live) return;

    // ---- store: final layout, one split = [K][Cin][Cout] followed by the bias row.  Lane (li, lg) holds rows 4 lg + r of
    // every tile: one base pointer per (row tile, r), the tap loop advances it by the uniform stride Cin * Cout ----
    float* outp = a.out + (long long)(a.direct ? 0 : a.split_base + split) * p.pstride;
    const long long tstride = (long long)Ctot * a.N;
    auto store_tiles = [&](auto taps) __attribute__((always_inline)) {
        using TP = decltype(taps);
        constexpr int JR = TP::JR, NK = TP::NK;
#pragma unroll
        for (int j = 0; j < JR; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * lg + r;                            // MFMA row
                const int c = cT + j * RC + i % RC, tg = 8 * (i / RC);
                float* rowp = outp + ((long long)tg * Ctot + c) * a.N + n0 + li;
#pragma unroll
                for (int n = 0; n < NW; ++n) {
                    if (n0 + n * 16 + li < a.N && c < Ctot) {
#pragma unroll
                        for (int kk = 0; kk < NK; ++kk)
                            if (tg + TP::k(kk) < a.KW) rowp[n * 16 + (long long)TP::k(kk) * tstride] = acc[j * NK + kk][n][r];
                    }
                }
            }
    };
    if constexpr (K15) {
        if (roleB) store_tiles(WinTapsB{});
        else store_tiles(WinTapsA{});
    } else {
        store_tiles(WinTaps5{});
    }
    // the bias row: (K15) row RC (= tap 7 + 8) of the first row tile, accumulator 5 of its wave; (K5) the first channel
    // past Cin, tap 0
    {
        const bool mine = K15 ? (!roleB && cT == 0) : (cT <= Ctot && Ctot < cT + RC);
        if (mine) {
            const int brow = K15 ? RC : Ctot - cT;
            constexpr int KB = K15 ? 5 : 0;
            if ((brow >> 2) == lg) {
                const long long boff = (long long)a.KW * Ctot * a.N;
#pragma unroll
                for (int n = 0; n < NW; ++n) {
                    const int col = n0 + n * 16 + li;
                    if (col < a.N) {
                        const f32x4 v = acc[KB][n];
                        const int rr = brow & 3;
                        outp[boff + col] = rr == 0 ? v[0] : rr == 1 ? v[1] : rr == 2 ? v[2] : v[3];
                    }
                }
            }
        }
    }
}

// out[i] = sum over splits of partial[s][i] (i over the final [K][Cin][Cout] + bias block): SL split lanes per element,
// each summing its splits {sl, sl + SL, ...} in order into one accumulator per residue class mod 4, then a fixed binary
// tree over the 4 * SL partial sums: deterministic, and the error grows like the tree depth rather than the split count
template <int SL>
__global__ __launch_bounds__(256) void wgrad_win_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out_w,
                                                               float* __restrict__ out_b, long long nw, long long n,
                                                               long long pstride, int nsplit) {
    const long long n4 = (n + 3) >> 2;
    __shared__ f32x4 red[256];
    constexpr int VPB = 256 / SL;
    const int slot = threadIdx.x % VPB, sl = threadIdx.x / VPB;
    const long long v = (long long)blockIdx.x * VPB + slot;
    f32x4 s4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (v < n4) {
        const f32x4* pp = reinterpret_cast<const f32x4*>(partial) + v;
        const long long st4 = pstride >> 2;
        int k = sl;
        for (; k + 3 * SL < nsplit; k += 4 * SL) {
            f32x4 t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = pp[(long long)(k + j * SL) * st4];
#pragma unroll
            for (int j = 0; j < 4; ++j) s4[j] += t[j];
        }
        for (int j = 0; k < nsplit; k += SL, ++j) s4[j] += pp[(long long)k * st4];
    }
    f32x4 sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    if constexpr (SL > 1) {
        red[threadIdx.x] = sum;
        __syncthreads();
#pragma unroll
        for (int half = SL / 2; half >= 1; half >>= 1) {
            if (sl < half) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + half * VPB];
            __syncthreads();
        }
        if (sl != 0) return;
        sum = red[slot];
    }
    if (v >= n4) return;
    const long long e0 = 4 * v;
    if (e0 + 3 < nw && (reinterpret_cast<uintptr_t>(out_w) & 15) == 0) {
        reinterpret_cast<f32x4*>(out_w)[v] = sum;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long e = e0 + k;
            if (e < nw) out_w[e] = sum[k];
            else if (e < n) out_b[e - nw] = sum[k];
        }
    }
}

// smallest row pitch >= need (floats, multiple of 4) for which the 16-byte reads of a wave -- lane (li, lg) at
// row(li) * pitch + goff * group(li) + step * lg -- are free of bank conflicts (gfx950 serves a ds_read_b128 in four
// groups of 16 lanes over 64 four-byte banks)
static int win_pitch(int need, int rc, int goff, int step) {
    static const int grp[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                   {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    const int lo = (need + 3) & ~3;
    for (int pitch = lo; pitch < lo + 260; pitch += 4) {
        bool ok = true;
        for (int half = 0; half < 2 && ok; ++half)
            for (int g = 0; g < 2 && ok; ++g) {
                unsigned long long used = 0;
                for (int k = 0; k < 16; ++k) {
                    const int lane = grp[g][k] + 32 * half;
                    const int li = lane & 15, lg = lane >> 4;
                    const int b0 = ((li % rc) * pitch + goff * (li / rc) + step * lg) & 63;
                    const unsigned long long m = 0xFull << b0;
                    if (used & m) { ok = false; break; }
                    used |= m;
                }
            }
        if (ok) return pitch;
    }
    return lo;
}

bool wgrad_win_supported(const WgradArgs& a) {
    const int Ctot = a.C0 + a.C1;
    if (a.bf16) return false;
    if (!(a.KW == 15 || a.KW == 5)) return false;                 // instantiated tap counts (the shipped 15 / 5)
    if (a.KW == 5 && a.loader == LOADER_DEINT) return false;
    if (Ctot < 8 || a.N < 16) return false;                        // narrow layers have their own kernel
    if (a.KW == 15 && (Ctot % 8) != 0) return false;               // row tiles of 8 channels x 2 tap groups
    if (a.C1 > 0 && (a.C0 % (a.KW == 15 ? 8 : 1)) != 0) return false;
    if ((long long)Ctot * std::max(a.pitch0, a.pitch1) >= (1ll << 30)) return false;   // 30-bit element offsets
    if ((long long)a.N * a.dzpitch >= (1ll << 30)) return false;                        // ... of the dz rows of one excerpt too
    return true;
}

struct WinGeom { int K15, NW, S; WinParams p; size_t lds; int threads; };

// tile choice: K = 15 -> 8 channels x {taps k, k + 8}, three row tiles (24 channels) per workgroup; K = 5 -> 16 channels,
// four row tiles.  Column tiles per wave: the count in {3, 4, 5} (K = 15) / {2 .. 6} (K = 5) that pads N least, ties to
// the smaller (<= 24 accumulator tiles fit 3 waves per SIMD).  force_mtw / force_nw override (autotuner, test hook).
static WinGeom wgrad_win_geom(const WgradArgs& a) {
    WinGeom g;
    const int Ctot = a.C0 + a.C1;
    g.S = a.loader == LOADER_DEINT ? 2 : 1;
    g.K15 = a.KW == 15 ? 1 : 0;
    const int G = g.K15 ? 2 : 1, NA = g.K15 ? 6 : 5, KT = g.K15 ? 8 : 5;
    const int RC = 16 / G;
    const int ntile = (a.N + 15) / 16;
    int nw = 0;
    {
        const int lo = g.K15 ? 3 : 2, hi = g.K15 ? 5 : 6;
        int best = 1 << 30;
        for (int c = lo; c <= hi; ++c) {
            const int padded = (ntile + c - 1) / c * c;
            if (padded < best) { best = padded; nw = c; }
        }
        if (ntile < lo) nw = ntile;
    }
    if (a.force_nw >= 1 && a.force_nw <= (g.K15 ? 5 : 6)) nw = a.force_nw;
    g.NW = nw;
    WinParams& p = g.p;
    p.RTW = g.K15 ? 3 : 4;                                               // row tiles per workgroup (always 4 waves per column group)
    const int rowtiles = (Ctot + (G == 1 ? 1 : 0) + RC - 1) / RC;       // (K = 5: room for the bias row past Cin)
    const int colgroups = (ntile + nw - 1) / nw;
    p.CGW = 1;                                                           // (8-wave workgroups leave room for only one per CU)
    if (a.force_mtw == 1 || a.force_mtw == 2) p.CGW = a.force_mtw <= colgroups ? a.force_mtw : colgroups;
    p.nMG = (rowtiles + p.RTW - 1) / p.RTW;
    p.nNG = (colgroups + p.CGW - 1) / p.CGW;
    p.GOFF = 8;
    g.threads = 64 * 4 * p.CGW;
    // LDS budget: enough workgroups per CU for 3 (<= 24 accumulator tiles: <= 168 registers) or 2 waves per SIMD
    const int wps = (g.K15 ? nw <= 2 : nw <= 4) ? 3 : 2;
    (void)NA;
    int wgs = wps / p.CGW;
    if (wgs < 1) wgs = 1;
    const size_t lds_cap = (size_t)(160 * 1024) / (size_t)wgs;
    int tk = 64;
    if (const char* e = getenv("WUN_WIN_TK")) { const int v = atoi(e); if (v == 16 || v == 32 || v == 64 || v == 128) tk = v; }
    while (tk > 16 && tk / 2 >= a.Tq) tk /= 2;
    for (;;) {
        p.TK = tk;
        const int WN = 4 * ((3 * g.S + KT + 3) / 4);
        const int need = g.S * (tk - 4) + p.GOFF * (G - 1) + WN;         // floats a window read may touch
        p.XGL = (g.S * (tk - 1) + a.KW + 3) / 4;                          // granules that carry needed samples
        if (4 * p.XGL < need) p.XGL = (need + 3) / 4;                     // (window reads stay inside fetched data)
        p.XP = win_pitch(4 * p.XGL, RC, p.GOFF, 4 * g.S);
        p.ZP = win_pitch(tk, 16, 0, 4);
        p.zoff = 4 * ((p.RTW * RC * (p.XP / 4) + 63) & ~63);                       // whole 1 KiB DMA blocks per region
        p.bufFloats = p.zoff + 4 * ((p.CGW * nw * 16 * (p.ZP / 4) + 63) & ~63);
        g.lds = sizeof(float) * 2 * (size_t)p.bufFloats;
        const long long xs = p.zoff / 4, zs = (p.bufFloats - p.zoff) / 4;
        const bool fits = xs <= (long long)WUN_WIN_XIT * g.threads && zs <= (long long)WUN_WIN_ZIT * g.threads &&
                          g.lds <= lds_cap;
        if (fits || tk == 16) break;
        tk /= 2;
    }
    p.pstride = (((long long)a.KW * Ctot * a.N + a.N) + 3) & ~3ll;
    return g;
}

long long wgrad_win_partial_floats(const WgradArgs& a) { return wgrad_win_geom(a).p.pstride; }
int wgrad_win_units(const WgradArgs& a) { const WinGeom g = wgrad_win_geom(a); return a.B * ((a.Tq + g.p.TK - 1) / g.p.TK); }
int wgrad_win_tiles(const WgradArgs& a) { const WinGeom g = wgrad_win_geom(a); return g.p.nMG * g.p.nNG; }

template <bool K15, int NW, int S>
static hipError_t win_launch_t(WgradArgs a, const WinGeom& g, hipStream_t s) {
    a.nQT = (a.Tq + g.p.TK - 1) / g.p.TK;
    const long long units = (long long)a.B * a.nQT;
    a.units_per_split = (int)((units + a.nsplit - 1) / a.nsplit);
    auto kern = wgrad_win_kernel<K15, NW, S>;
    static size_t lds_allowed = 64 * 1024;
    if (g.lds > lds_allowed) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
        if (e != hipSuccess) return e;
        lds_allowed = g.lds;
    }
    const long long grid = (long long)g.p.nMG * g.p.nNG * a.nsplit;
    char nm[64];
    snprintf(nm, sizeof(nm), "wgrad_win_kernel<%d, %d, %d>", K15 ? 15 : 5, NW, S);
    char tag[200];
    int occ = -1;
    if (getenv("WUN_WIN_OCC")) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, g.threads, g.lds);
    snprintf(tag, sizeof(tag), "C=%d N=%d T=%d K=%d ld=%d B=%d nsplit=%d grid=%lld w=%dx%d tk=%d lds=%zu occ=%d", a.C0 + a.C1, a.N, a.Tq,
             a.KW, a.loader, a.B, a.nsplit, grid, 4, g.p.CGW, g.p.TK, g.lds, occ);
    if (getenv("WUN_WIN_OCC")) fprintf(stderr, "[win] %s %s\n", nm, tag);
    prof_scope_begin(nm, 2.0 * a.KW * (double)(a.C0 + a.C1) * a.N * (double)a.Tq * a.B, s, tag);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(g.threads), g.lds, s, a, g.p);
    prof_scope_end(s);
    return hipGetLastError();
}

hipError_t launch_wgrad_win(const WgradArgs& a, hipStream_t s) {
    if (!wgrad_win_supported(a)) return hipErrorInvalidValue;
    const WinGeom g = wgrad_win_geom(a);
    if (g.lds > 160 * 1024) return hipErrorInvalidValue;
    {
        const long long xs = g.p.zoff / 4, zs = (g.p.bufFloats - g.p.zoff) / 4;
        if (xs > (long long)WUN_WIN_XIT * g.threads || zs > (long long)WUN_WIN_ZIT * g.threads) return hipErrorInvalidValue;
    }
#define WUN_WIN(k15, nw, ss) if (g.K15 == k15 && g.NW == nw && g.S == ss) return win_launch_t<k15 != 0, nw, ss>(a, g, s);
    WUN_WIN(1, 1, 1) WUN_WIN(1, 2, 1) WUN_WIN(1, 3, 1) WUN_WIN(1, 4, 1) WUN_WIN(1, 5, 1)
    WUN_WIN(1, 1, 2) WUN_WIN(1, 2, 2) WUN_WIN(1, 3, 2) WUN_WIN(1, 4, 2) WUN_WIN(1, 5, 2)
    WUN_WIN(0, 1, 1) WUN_WIN(0, 2, 1) WUN_WIN(0, 3, 1) WUN_WIN(0, 4, 1) WUN_WIN(0, 5, 1) WUN_WIN(0, 6, 1)
#undef WUN_WIN
    return hipErrorInvalidValue;
}

hipError_t launch_wgrad_win_reduce(const WgradArgs& a, const float* partial, int nsplit, float* out_w, float* out_b,
                                   hipStream_t s) {
    const WinGeom g = wgrad_win_geom(a);
    const long long nw = (long long)a.KW * (a.C0 + a.C1) * a.N, n = nw + a.N;
    const long long n4 = (n + 3) / 4;
    char tag[96];
    snprintf(tag, sizeof(tag), "bytes=%lld nsplit=%d", (long long)(nsplit + 1) * n * 4, nsplit);
    prof_scope_begin("wgrad_win_reduce_kernel", 0.0, s, tag, (double)(nsplit + 1) * (double)n * 4.0);
    const int sl = nsplit >= 128 ? 16 : (nsplit >= 16 ? 4 : 1);
    const int vpb = 256 / sl;
    const long long blocks = (n4 + vpb - 1) / vpb;
    if (sl == 16) hipLaunchKernelGGL(wgrad_win_reduce_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, partial, out_w, out_b, nw, n, g.p.pstride, nsplit);
    else if (sl == 4) hipLaunchKernelGGL(wgrad_win_reduce_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, partial, out_w, out_b, nw, n, g.p.pstride, nsplit);
    else hipLaunchKernelGGL(wgrad_win_reduce_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, partial, out_w, out_b, nw, n, g.p.pstride, nsplit);
    prof_scope_end(s);
    return hipGetLastError();
}

}  // namespace wun

