// gfx950 bf16-MFMA speed mode of the weight / bias gradient.
//
//   dW[tap][cin][cout] = sum over (excerpt, q) of X[cin][q*stride + tap] * dZ[cout][q]
//   on v_mfma_f32_16x16x32_bf16 (fp32 accumulate): MFMA rows = 16 input channels of ONE tap, columns = 16 output
//   channels, k = 32 output positions.
//
// The reduction runs over POSITIONS, and the A operand of tap j is the input row shifted by j samples: in a
// channel-major LDS image that is 8 consecutive bf16 at an arbitrary 2-byte offset (the first version of this
// kernel assembled every A fragment from eight 16-bit LDS reads and was LDS-issue-bound at 9-17 % of the bf16
// MFMA peak).  Here the input tile is staged POSITION-major -- Xs[channel block][position][16 channels], 32-byte
// rows -- and read with gfx950's transposing LDS read (ds_read_b64_tr_b16: a 16-lane group fetches a
// [4 positions][16 channels] block and every lane receives the 4 positions of its channel): the tap shift is a ROW
// offset, an A fragment is two such reads (4 LDS cycles, like the aligned B fragment), and one staged tile serves
// all taps.  Lane group g of a k-step holds positions {4g..4g+3, 16+4g..16+4g+3} in BOTH operands (k is a
// summation index, any consistent assignment works), which makes the A reads of a wave cover 512 contiguous bytes
// (no bank conflicts) and the B fragment two 8-byte reads of a dz row.
//
// Workgroup = 4 waves x MTW row tiles = 4*MTW slots: slot t < NCB*K is (channel block t / K, tap t % K) with
// NCB = (4*MTW - 1) / K channel blocks per workgroup, the last slot is the bias gradient (an all-ones A operand in
// registers; row group 0 only).  Columns: NW tiles of 16 output channels shared by all waves.  The position axis is
// split over workgroups in units of TK <= 128 positions of one excerpt; every workgroup writes its raw tiles to the
// tile-major partial buffer and wgrad_bf16_reduce_kernel sums the splits in fixed order (deterministic, no atomics).
// Activations and their gradients live in HBM as bf16 (round 5; WgradArgs.sbf -- the single-operator entry point converts
// its fp32 arguments first): an item of the input stage is (channel pair, 4 positions) = two 8-byte loads, transposed to
// four (position, channel pair) dwords with v_perm_b32; a dz item is one 8-byte load written to LDS as it is.  Staging
// writes are rotated per position group so that the 32-byte rows of 4 neighbouring groups land on different banks.
#include "wun_internal.h"

#include <cstdio>

namespace wun {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

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

// 4 positions x 16 channels block -> the 4 positions of this lane's channel (see the header)
__device__ __forceinline__ u32x2 wb_tr_read(const unsigned short* p) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p));
}

#define WUN_WGB_XITP 4     // staged (channel pair, 4 positions) items per thread and unit: 2 float4 loads each

struct WgBfK { int NCB, nCB, nMG, nNG, TK, XW4, XROWS, ZPe; };

template <int MTW, int NW>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(WgradArgs a, WgBfK g) {
    extern __shared__ __attribute__((aligned(16))) unsigned short wl[];
    constexpr int NG = NW * 16;
    constexpr int SLOTS = 4 * MTW;
    constexpr int ZIT = (NG * 32 + 255) / 256;          // TK/4 <= 32 float4 per dz row
    const bool deint = (a.loader == LOADER_DEINT);
    const int planes = deint ? 2 : 1;
    const int xsub = planes * g.XROWS * 16;             // elements of one channel block's image (multiple of 64)
    // wl[0, 512): 32 rows x 16 of bf16 1.0 -- the A operand of the bias slot (and of idle slots, whose results are
    // discarded): every slot of the MFMA loop is the same pair of reads, so the loop has no branches and the
    // compiler can issue all LDS reads of a k-step ahead of its MFMAs
    unsigned short* Xs = wl + 512;
    unsigned short* Zs = Xs + g.NCB * xsub + 64;           // (+ a trash row for staged positions before the image start)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = wb_xcd_block((int)blockIdx.x, (int)gridDim.x);
    const int ng = bid % g.nNG; bid /= g.nNG;
    const int mg = bid % g.nMG;
    const int split = bid / g.nMG;
    const int cb0 = mg * g.NCB;
    const int Ctot = a.C0 + a.C1;
    const int K = a.KW;

    // alignment of the staged window: vector loads start at a multiple of 4 floats of the source row
    const int delta0 = ((a.off0 - a.shift) % 4 + 4) % 4;
    const int delta1 = ((a.off1 - a.shift) % 4 + 4) % 4;

    // per slot of this wave: LDS element offset of this lane's transposing read (row 4*lg + (li >> 2) of the tap's
    // first k-step, columns 4*(li & 3)..) and its advance per k-step (0 for the ones block); kind: 0 real, 1 bias, 2 idle
    int abase[MTW], astep[MTW], kind[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int slot = wave * MTW + mt;
        const int lane_off = (4 * lg + (li >> 2)) * 16 + 4 * (li & 3);
        int off = lane_off, step = 0, kd = 2;
        if (slot == SLOTS - 1) {
            kd = (mg == 0) ? 1 : 2;
        } else {
            const int cbl = slot / K, tap = slot - cbl * K;
            if (cbl < g.NCB && cb0 + cbl < g.nCB) {
                const int row0 = deint ? (tap & 1) * g.XROWS + (tap >> 1) : tap;
                off = 512 + cbl * xsub + row0 * 16 + lane_off;
                step = 32 * 16;
                kd = 0;
            }
        }
        abase[mt] = off; astep[mt] = step; kind[mt] = __builtin_amdgcn_readfirstlane(kd);
    }
    for (int i = tid; i < 256; i += 256) reinterpret_cast<unsigned*>(wl)[i] = 0x3F803F80u;      // bf16 1.0 pairs
    // a short last k-step reads input rows past the staged window (against zeroed dz): keep every element of the
    // input image finite from the start (0 * NaN would poison the accumulators)
    for (int i = tid; i < g.NCB * xsub / 2; i += 256) reinterpret_cast<unsigned*>(Xs)[i] = 0u;

    f32x4 acc[MTW][NW];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    u32x2 xra[WUN_WGB_XITP], xrb[WUN_WGB_XITP];       // channels 2*pl and 2*pl + 1 of an item: 4 bf16 positions each
    u32x2 zreg[ZIT];
    const int TK4 = g.TK >> 2;
    const float inv_tk4 = 1.0f / (float)TK4;

    // packed per-item staging state: bit 31 live | rot << 28 | cbl << 24 | pl << 20 | c4 (position group)
    int xpk[WUN_WGB_XITP];
    int zpk[ZIT];
    {
        const float inv_ncb = 1.0f / (float)g.NCB;
#pragma unroll
        for (int i = 0; i < WUN_WGB_XITP; ++i) {
            const int f = tid + i * 256;
            const int pl = f & 7, m = f >> 3;
            const int c4 = (int)(((float)m + 0.5f) * inv_ncb);
            const int cbl = m - c4 * g.NCB;
            const bool live = c4 < g.XW4 && cb0 + cbl < g.nCB;
            xpk[i] = (int)((live ? 0x80000000u : 0u) | ((unsigned)(m & 3) << 28) | ((unsigned)cbl << 24) | ((unsigned)pl << 20) |
                           (unsigned)(c4 & 0xFFFFF));
        }
    }
#pragma unroll
    for (int i = 0; i < ZIT; ++i) {
        const int f = tid + i * 256;
        const int row = (int)(((float)f + 0.5f) * inv_tk4);
        const int c4 = f - row * TK4;
        zpk[i] = row < NG ? (int)(0x80000000u | ((unsigned)row << 23) | ((unsigned)c4 << 16) | (unsigned)(row * g.ZPe + 4 * c4))
                          : (int)((unsigned)c4 << 16);
    }

    auto load_unit = [&](int u) {
        const int b = u / a.nQT, qt = u - b * a.nQT;
        const int q0 = qt * g.TK;
        const int tb = (deint ? 2 * q0 : q0) - a.shift;
        const bf16_t* base0 = reinterpret_cast<const bf16_t*>(a.src0) + (long long)b * a.bs0;
        const bf16_t* base1 = (a.C1 > 0) ? reinterpret_cast<const bf16_t*>(a.src1) + (long long)b * a.bs1 : base0;
        const int e00 = (tb + a.off0) & ~3, e01 = (tb + a.off1) & ~3;
#pragma unroll
        for (int i = 0; i < WUN_WGB_XITP; ++i) {
            int pk = xpk[i];
            asm volatile("" : "+v"(pk));
            int c = (cb0 + ((pk >> 24) & 15)) * 16 + 2 * ((pk >> 20) & 15);
            int c1 = c + 1;
            c = c < Ctot ? c : Ctot - 1;
            c1 = c1 < Ctot ? c1 : Ctot - 1;
            const bool s1 = c >= a.C0;                  // (C0 is even: both channels of a pair come from one source)
            const int pitch = s1 ? a.pitch1 : a.pitch0;
            const bf16_t* base = s1 ? base1 : base0;
            const int cs = s1 ? a.C0 : 0;
            int e = (s1 ? e01 : e00) + ((pk & 0xFFFFF) << 2);
            const int emax = pitch - 4;
            e = e < 0 ? 0 : (e > emax ? emax : e);
            xra[i] = *reinterpret_cast<const u32x2*>(base + (long long)(c - cs) * pitch + e);
            xrb[i] = *reinterpret_cast<const u32x2*>(base + (long long)(c1 - cs) * pitch + e);
        }
        const bf16_t* zb = reinterpret_cast<const bf16_t*>(a.dz) + (long long)b * a.dzbs;
        const int qmax = a.dzpitch - 4;
#pragma unroll
        for (int i = 0; i < ZIT; ++i) {
            int pk = zpk[i];
            asm volatile("" : "+v"(pk));
            const int nn = ng * NG + ((pk >> 23) & 255);
            const int zro = (pk < 0 && nn < a.N ? nn : 0) * a.dzpitch;
            int q = q0 + (((pk >> 16) & 127) << 2);
            q = q > qmax ? qmax : q;
            zreg[i] = *reinterpret_cast<const u32x2*>(zb + zro + q);
        }
    };
    // per item: the 4 LDS destinations (element offsets from Xs; unit-invariant) in WRITE order -- write instruction k
    // stores position (k + rot) & 3 of the item, so the 32-byte rows of 4 neighbouring position groups differ mod 4
    // (positions before the image start, r < 0, go to a trash row behind the image)
    const int trash = g.NCB * xsub;                       // 16 elements behind the last channel block (reserved by the launcher)
    unsigned xdst[WUN_WGB_XITP][2];
#pragma unroll
    for (int i = 0; i < WUN_WGB_XITP; ++i) {
        const int pk = xpk[i];
        const int cbl = (pk >> 24) & 15, pl = (pk >> 20) & 15, c4 = pk & 0xFFFFF, rot = (pk >> 28) & 3;
        const int c = (cb0 + cbl) * 16 + 2 * pl;
        const int r0 = 4 * c4 - (c >= a.C0 ? delta1 : delta0);
        unsigned o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + ((k + rot) & 3);
            const int row = deint ? (r & 1) * g.XROWS + (r >> 1) : r;
            o[k] = (unsigned)(r >= 0 ? cbl * xsub + row * 16 + 2 * pl : trash + 2 * pl);
        }
        xdst[i][0] = o[0] | (o[1] << 16);
        xdst[i][1] = o[2] | (o[3] << 16);
    }
    const bool cedge = (cb0 + g.NCB) * 16 > Ctot;         // the last channel block of this row group is partly empty
    auto store_unit = [&](int u) {
        const int b = u / a.nQT, qt = u - b * a.nQT;
        const int q0 = qt * g.TK;
        const int tb = (deint ? 2 * q0 : q0) - a.shift;
        int nq = a.Tq - q0; if (nq > g.TK) nq = g.TK;
        // interior units (every staged position inside the input, full channel blocks) skip the per-element tests
        const bool edge = cedge || tb - 3 < 0 || tb + 4 * g.XW4 > a.Tin;
#pragma unroll
        for (int i = 0; i < WUN_WGB_XITP; ++i) {
            int pk = xpk[i];
            asm volatile("" : "+v"(pk));
            if (pk < 0) {
                const u32x2 ua = xra[i], ub = xrb[i];
                // d[k] = {channel c, channel c + 1} at position k of the item: low / high halves of the two rows' dwords
                unsigned d[4];
                d[0] = __builtin_amdgcn_perm(ub[0], ua[0], 0x05040100u);
                d[1] = __builtin_amdgcn_perm(ub[0], ua[0], 0x07060302u);
                d[2] = __builtin_amdgcn_perm(ub[1], ua[1], 0x05040100u);
                d[3] = __builtin_amdgcn_perm(ub[1], ua[1], 0x07060302u);
                if (edge) {
                    const int cbl = (pk >> 24) & 15, pl = (pk >> 20) & 15, c4 = pk & 0xFFFFF;
                    const int c = (cb0 + cbl) * 16 + 2 * pl;
                    const int t0 = tb + 4 * c4 - (c >= a.C0 ? delta1 : delta0);    // position in the source's own time axis
                    const unsigned cm = (c < Ctot ? 0x0000FFFFu : 0u) | (c + 1 < Ctot ? 0xFFFF0000u : 0u);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool in = t0 + k >= 0 && t0 + k < a.Tin;
                        d[k] = in ? (d[k] & cm) : 0u;
                    }
                }
                const int rot = (pk >> 28) & 3;
                if (rot & 1) { const unsigned t = d[0]; d[0] = d[1]; d[1] = d[2]; d[2] = d[3]; d[3] = t; }
                if (rot & 2) { unsigned t = d[0]; d[0] = d[2]; d[2] = t; t = d[1]; d[1] = d[3]; d[3] = t; }
                unsigned w0 = xdst[i][0], w1 = xdst[i][1];
                asm volatile("" : "+v"(w0), "+v"(w1));
                *reinterpret_cast<unsigned*>(Xs + (w0 & 0xFFFF)) = d[0];
                *reinterpret_cast<unsigned*>(Xs + (w0 >> 16)) = d[1];
                *reinterpret_cast<unsigned*>(Xs + (w1 & 0xFFFF)) = d[2];
                *reinterpret_cast<unsigned*>(Xs + (w1 >> 16)) = d[3];
            }
        }
        // positions beyond nq are zero in dz, so whatever the input rows hold there contributes nothing
        const bool zedge = nq < g.TK;
#pragma unroll
        for (int i = 0; i < ZIT; ++i) {
            int pk = zpk[i];
            asm volatile("" : "+v"(pk));
            if (pk < 0) {
                u32x2 v = zreg[i];
                if (zedge) {
                    const int c4x = ((pk >> 16) & 127) << 2;
                    if (c4x >= nq) v[0] &= 0xFFFF0000u;
                    if (c4x + 1 >= nq) v[0] &= 0x0000FFFFu;
                    if (c4x + 2 >= nq) v[1] &= 0xFFFF0000u;
                    if (c4x + 3 >= nq) v[1] &= 0x0000FFFFu;
                }
                *reinterpret_cast<u32x2*>(Zs + (pk & 0xFFFF)) = v;
            }
        }
        (void)b;
    };

    const int nunits = a.B * a.nQT;
    const int u0 = split * a.units_per_split;
    int u1 = u0 + a.units_per_split;
    if (u1 > nunits) u1 = nunits;
    // the dz tile of a unit is only written up to TK: clear the tail of the last k-step once if TK is not a multiple of 32
    const int TKr = (g.TK + 31) & ~31;
    if (TKr != g.TK)
        for (int i = tid; i < NG * (TKr - g.TK); i += 256) Zs[(i / (TKr - g.TK)) * g.ZPe + g.TK + i % (TKr - g.TK)] = 0;
    if (u0 < u1) load_unit(u0);
    for (int u = u0; u < u1; ++u) {
        __syncthreads();
        store_unit(u);
        __syncthreads();
        if (u + 1 < u1) load_unit(u + 1);
        const int qt = u % a.nQT;
        int nq = a.Tq - qt * g.TK; if (nq > g.TK) nq = g.TK;
        const int nsteps = (nq + 31) >> 5; // k-steps of 32 positions
        for (int st = 0; st < nsteps; ++st) {
            bf16x8 bv[NW];
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                const unsigned short* zp = Zs + (n * 16 + li) * g.ZPe + st * 32 + 4 * lg;
                const u32x2 lo = *reinterpret_cast<const u32x2*>(zp);
                const u32x2 hi = *reinterpret_cast<const u32x2*>(zp + 16);
                bv[n] = __builtin_bit_cast(bf16x8, (u32x4){lo[0], lo[1], hi[0], hi[1]});
            }
            bf16x8 av[MTW];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                const unsigned short* xp = wl + abase[mt];
                const u32x2 lo = wb_tr_read(xp);
                const u32x2 hi = wb_tr_read(xp + 16 * 16);
                av[mt] = __builtin_bit_cast(bf16x8, (u32x4){lo[0], lo[1], hi[0], hi[1]});
                abase[mt] += astep[mt];
            }
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int n = 0; n < NW; ++n)
                    acc[mt][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mt], bv[n], acc[mt][n], 0, 0, 0);
        }
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) abase[mt] -= nsteps * astep[mt];
    }

    if (!a.direct) {
        f32x4* tile = reinterpret_cast<f32x4*>(a.out) +
                      ((((long long)(a.split_base + split) * g.nMG + mg) * g.nNG + ng) * (SLOTS * 16 * NG / 4)) +
                      wave * (MTW * NW * 64) + lane;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            if (kind[mt] == 2) continue;                   // idle: never read by the reduction
#pragma unroll
            for (int n = 0; n < NW; ++n) tile[(mt * NW + n) * 64] = acc[mt][n];
        }
        return;
    }
    float* outp = a.out;                                   // single split: final layout [K][Cin][Cout] + bias row
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int slot = wave * MTW + mt;
        if (kind[mt] == 2) continue;
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            const int col = ng * NG + n * 16 + li;
            if (col >= a.N) continue;
            if (kind[mt] == 1) {
                if (lg == 0) outp[(long long)K * Ctot * a.N + col] = acc[mt][n][0];
                continue;
            }
            const int cbl = slot / K, tap = slot - cbl * K;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int c = (cb0 + cbl) * 16 + lg * 4 + r4;
                if (c < Ctot) outp[((long long)tap * Ctot + c) * a.N + col] = acc[mt][n][r4];
            }
        }
    }
}

// Sums the tile-major split partials in split order (SL split lanes per element, combined in lane order:
// deterministic) and scatters to out_w[K][Cin][Cout], out_b[Cout].  One thread = one f32x4 accumulator register.
struct WgradBfReduceArgs {
    const float* partial; float* out_w; float* out_b;
    int nsplit, MTW, NW, NCB, nCB, nMG, nNG, KW, Ctot, N;
};

template <int SL>
__global__ __launch_bounds__(256) void wgrad_bf16_reduce_kernel(WgradBfReduceArgs a) {
    __shared__ f32x4 red[SL > 1 ? 256 : 1];
    constexpr int VPB = 256 / SL;                                  // vector slots per block
    const int SLOTS = 4 * a.MTW, NG = a.NW * 16;
    const int tile_v = SLOTS * 16 * NG / 4;                        // f32x4 slots per (row group, column group) tile
    const int slot_v = threadIdx.x % VPB, sl = threadIdx.x / VPB;
    const long long gv = (long long)blockIdx.x * VPB + slot_v;     // global slot over all tiles
    const long long ntile = (long long)a.nMG * a.nNG;
    bool live = gv < ntile * tile_v;
    // decode first: idle slots were never written
    int mg = 0, col = 0, slot = 0, lg = 0;
    if (live) {
        const int t = (int)(gv / tile_v), v = (int)(gv % tile_v);
        mg = t / a.nNG;
        const int ng = t % a.nNG;
        const int per_wave = a.MTW * a.NW * 64;
        const int wave = v / per_wave, rem = v % per_wave;
        const int tn = rem / 64, lane = rem % 64;
        const int mt = tn / a.NW, n = tn % a.NW;
        lg = lane >> 4;
        col = ng * NG + n * 16 + (lane & 15);
        slot = wave * a.MTW + mt;
        if (slot == SLOTS - 1) live = mg == 0;
        else { const int cbl = slot / a.KW; live = cbl < a.NCB && mg * a.NCB + cbl < a.nCB; }
    }
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const f32x4* p = reinterpret_cast<const f32x4*>(a.partial) + gv;
        const long long sstride = ntile * tile_v;
        int k = sl;
        for (; k + 7 * SL < a.nsplit; k += 8 * SL) {               // batches of independent loads, summed in split order
            f32x4 t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = p[(long long)(k + j * SL) * sstride];
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += t[j];
        }
        for (; k < a.nsplit; k += SL) sum += p[(long long)k * sstride];
    }
    if constexpr (SL > 1) {
        red[threadIdx.x] = sum;
        __syncthreads();
        if (sl != 0) return;
        sum = red[slot_v];
#pragma unroll
        for (int k = 1; k < SL; ++k) sum += red[k * VPB + slot_v];
    }
    if (!live || col >= a.N) return;
    if (slot == SLOTS - 1) {
        if (lg == 0) a.out_b[col] = sum[0];
        return;
    }
    const int cbl = slot / a.KW, tap = slot - cbl * a.KW;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int c = (mg * a.NCB + cbl) * 16 + lg * 4 + r4;
        if (c < a.Ctot) a.out_w[((long long)tap * a.Ctot + c) * a.N + col] = sum[r4];
    }
}

// ---------------------------------------------------------------------------------------
// host side: geometry, launchers
// ---------------------------------------------------------------------------------------
bool wgrad_bf16_supported(const WgradArgs& a) {
    if (a.KW < 1 || a.KW > 15 || a.C0 + a.C1 < 8) return false;
    if (a.C1 > 0 && (a.C0 & 1)) return false;          // a staged channel pair must not straddle the two sources
    return true;
}

WgradBfGeom wgrad_bf16_geom(const WgradArgs& a) {
    WgradBfGeom g;
    const int Ctot = a.C0 + a.C1, K = a.KW;
    const bool deint = a.loader == LOADER_DEINT;
    g.nCB = (Ctot + 15) / 16;
    // columns: the tile count (2..5) that pads the output channels least, wider on ties
    int bestnw = 4, bestpad = 1 << 30;
    for (int nw = 5; nw >= 2; --nw) {
        const int padded = ((a.N + nw * 16 - 1) / (nw * 16)) * nw * 16;
        if (padded < bestpad) { bestpad = padded; bestnw = nw; }
    }
    if (a.N <= 16) bestnw = 1;
    g.NW = (a.force_nw >= 1 && a.force_nw <= 5) ? a.force_nw : bestnw;
    // rows: 4 tiles per wave; 8 (half as many row groups, each of which re-reads the dz tile, but 2 instead of 3
    // resident workgroups per CU) only when the autotuner measured it faster
    const int ncb4 = std::max(1, std::min(g.nCB, 15 / K)), ncb8 = std::max(1, std::min(g.nCB, 31 / K));
    int mtw = 4;
    if (a.force_mtw == 4 || a.force_mtw == 8) mtw = a.force_mtw;
    if (mtw == 8 && g.NW > 3) mtw = 4;                  // accumulator budget
    g.MTW = mtw;
    g.NCB = mtw == 8 ? ncb8 : ncb4;
    g.nMG = (g.nCB + g.NCB - 1) / g.NCB;
    g.nNG = (a.N + g.NW * 16 - 1) / (g.NW * 16);
    int tk = (a.Tq + 3) & ~3;
    if (tk > 128) tk = 128;
    if (tk < 4) tk = 4;
    for (;;) {
        const int need = deint ? 2 * tk + K - 1 : tk + K - 1;         // staged positions of a unit (+ 3 of alignment slack)
        g.XW4 = (need + 3 + 3) / 4;
        if ((long long)8 * g.NCB * g.XW4 <= (long long)WUN_WGB_XITP * 256 || tk <= 32) break;
        tk = (tk / 2 + 3) & ~3;
    }
    while ((long long)8 * g.NCB * g.XW4 > (long long)WUN_WGB_XITP * 256 && g.NCB > 1) --g.NCB;    // (1-tap filters)
    g.nMG = (g.nCB + g.NCB - 1) / g.NCB;
    g.TK = tk;
    const int tkr = (tk + 31) & ~31;
    const int rows = deint ? std::max(tkr + (K + 1) / 2, 2 * g.XW4 + 1) : std::max(tkr + K - 1, 4 * g.XW4);
    g.XROWS = (rows + 3) & ~3;
    g.ZPe = tkr + 8;
    g.lds = 2 * (512 + (size_t)g.NCB * (deint ? 2 : 1) * g.XROWS * 16 + 64 + (size_t)g.NW * 16 * g.ZPe);
    return g;
}

template <int MTW, int NW>
static hipError_t wgrad_bf16_launch_t(WgradArgs a, const WgradBfGeom& g, hipStream_t s) {
    a.nQT = (a.Tq + g.TK - 1) / g.TK;
    const long long units = (long long)a.B * a.nQT;
    a.units_per_split = (int)((units + a.nsplit - 1) / a.nsplit);
    if (g.lds > 160 * 1024 || (size_t)(NW * 16) * g.ZPe > 65535 || g.XW4 >= (1 << 20)) return hipErrorInvalidValue;
    if ((size_t)g.NCB * (a.loader == LOADER_DEINT ? 2 : 1) * g.XROWS * 16 + 80 > 65535) return hipErrorInvalidValue;   // 16-bit LDS destinations
    if ((long long)8 * g.NCB * g.XW4 > (long long)WUN_WGB_XITP * 256) return hipErrorInvalidValue;
    auto kern = wgrad_bf16_kernel<MTW, NW>;
    static size_t lds_allowed = 64 * 1024;
    if (g.lds > lds_allowed) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
        if (e != hipSuccess) return e;
        lds_allowed = g.lds;
    }
    const long long grid = (long long)g.nMG * g.nNG * a.nsplit;
    char nm[64], tag[160];
    snprintf(nm, sizeof(nm), "wgrad_bf16_kernel<%d, %d>", MTW, NW);
    snprintf(tag, sizeof(tag), "C=%d N=%d T=%d K=%d ld=%d B=%d nsplit=%d grid=%lld", a.C0 + a.C1, a.N, a.Tq, a.KW, a.loader, a.B,
             a.nsplit, grid);
    prof_scope_begin(nm, 2.0 * a.KW * (double)(a.C0 + a.C1) * a.N * (double)a.Tq * a.B, s, tag);
    WgBfK k;
    k.NCB = g.NCB; k.nCB = g.nCB; k.nMG = g.nMG; k.nNG = g.nNG; k.TK = g.TK; k.XW4 = g.XW4; k.XROWS = g.XROWS; k.ZPe = g.ZPe;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), g.lds, s, a, k);
    prof_scope_end(s);
    return hipGetLastError();
}

hipError_t launch_wgrad_bf16(const WgradArgs& a, hipStream_t s) {
    if (!wgrad_bf16_supported(a) || !a.sbf) return hipErrorInvalidValue;     // bf16 rows in HBM (the operator entry point converts first)
    if ((a.pitch0 & 3) || (a.bs0 & 3) || (reinterpret_cast<uintptr_t>(a.src0) & 15) || a.pitch0 < 4) return hipErrorInvalidValue;
    if (a.C1 > 0 && ((a.pitch1 & 3) || (a.bs1 & 3) || (reinterpret_cast<uintptr_t>(a.src1) & 15) || a.pitch1 < 4)) return hipErrorInvalidValue;
    if ((a.dzpitch & 3) || (a.dzbs & 3) || (reinterpret_cast<uintptr_t>(a.dz) & 15) || a.dzpitch < 4) return hipErrorInvalidValue;
    const WgradBfGeom g = wgrad_bf16_geom(a);
#define WUN_WGB(M, N) if (g.MTW == M && g.NW == N) return wgrad_bf16_launch_t<M, N>(a, g, s);
    WUN_WGB(4, 1) WUN_WGB(4, 2) WUN_WGB(4, 3) WUN_WGB(4, 4) WUN_WGB(4, 5)
    WUN_WGB(8, 1) WUN_WGB(8, 2) WUN_WGB(8, 3)
#undef WUN_WGB
    return hipErrorInvalidValue;
}

hipError_t launch_wgrad_bf16_reduce(const WgradArgs& a, const float* partial, int nsplit, float* out_w, float* out_b,
                                    hipStream_t s) {
    const WgradBfGeom g = wgrad_bf16_geom(a);
    WgradBfReduceArgs r;
    r.partial = partial; r.out_w = out_w; r.out_b = out_b;
    r.nsplit = nsplit; r.MTW = g.MTW; r.NW = g.NW; r.NCB = g.NCB; r.nCB = g.nCB; r.nMG = g.nMG; r.nNG = g.nNG;
    r.Ctot = a.C0 + a.C1; r.KW = a.KW; r.N = a.N;
    const long long slots = (long long)g.nMG * g.nNG * (4 * g.MTW * 16) * (g.NW * 16) / 4;
    const int sl = (nsplit >= 64 && slots < (1 << 16)) ? 16 : (nsplit >= 8 && slots < (1 << 18) ? 4 : 1);
    const int vpb = 256 / sl;
    const long long blocks = (slots + vpb - 1) / vpb;
    if (sl == 16) hipLaunchKernelGGL(wgrad_bf16_reduce_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, r);
    else if (sl == 4) hipLaunchKernelGGL(wgrad_bf16_reduce_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, r);
    else hipLaunchKernelGGL(wgrad_bf16_reduce_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, r);
    return hipGetLastError();
}

}  // namespace wun
