// Weight gradients of the two "narrow" convolutions of the Wave-U-Net -- the audio-input conv (1 or 2
// input channels, /root/reference/Models/UnetAudioSeparator.py:98 at i = 0) and the output head (1 or 2
// output channels per source, /root/reference/Models/OutputLayer.py:8,15) -- as direct reductions on the
// vector pipe.  Their contraction has no dense channel x channel face (Cin*Cout <= a few dozen), so an MFMA
// tile is > 90 % padding there (the MFMA weight-gradient kernel ran them at 0.6 and 10 TFLOP/s); they are
// bound by streaming dz once from HBM.
//
//   dW[k][ci][n] = sum_{b,q} x[b][ci][q*SI + k - shift] * dz[b][n][q],   db[n] = sum_{b,q} dz[b][n][q]
//
// One workgroup owns a contiguous range of (excerpt, 256-position tile) units.  Per unit the dz rows and
// the input rows (+ halo) are staged in LDS with coalesced loads; thread (pair p = (ci, n), lane group g)
// keeps the KT tap accumulators of its pair in registers and walks the tile 4 positions at a time with
// 16-byte LDS reads (one dz vector, a contiguous x window), i.e. 4*K FMAs per ~6 LDS reads.  At the end the
// lane groups of a pair are summed through LDS in a fixed order and the workgroup writes one partial
// vector; narrow_wgrad_reduce_kernel sums the partials in split order (deterministic, no atomics) and
// scatters to the TF layout [K][Cin][Cout] + bias of each source.
#include "wun_internal.h"

#include <cstdio>
#include <cstdlib>

namespace wun {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WUN_NW_TQ 256            // output positions per unit

// ETM: element types of the bf16 mode (activations and their gradients live in HBM as bf16) -- bit 0: src0, bit 1: src1,
// bit 2: dz hold bf16_t instead of float (NarrowWgradArgs.et); values are widened as they are staged, the arithmetic is fp32.
template <int B> struct NwEt { typedef float T; };
template <> struct NwEt<1> { typedef bf16_t T; };

template <int KT, int SI, int ETM>
__global__ __launch_bounds__(256) void narrow_wgrad_kernel(NarrowWgradArgs a) {
    typedef typename NwEt<ETM & 1>::T X0T;
    typedef typename NwEt<(ETM >> 1) & 1>::T X1T;
    typedef typename NwEt<(ETM >> 2) & 1>::T ZT;
    extern __shared__ __attribute__((aligned(16))) float nlds[];
    constexpr int TQ = WUN_NW_TQ;
    constexpr int XWIN = 3 * SI + KT;                       // x values one thread needs for 4 positions
    constexpr int XV = (XWIN + 3) / 4;                      // as 16-byte vectors
    constexpr int XW = (TQ * SI + KT - 1 + 4 * XV + 3) / 4 * 4;   // floats per staged input row (window + read slack)
    const int Ctot = a.C0 + a.C1;
    const int NP = Ctot * a.N;                              // (ci, n) pairs, <= 256
    float* Xs = nlds;                                       // [Ctot][XW]
    constexpr int ZP = TQ + 4;                              // dz row pitch: lanes of a wave read DIFFERENT rows at the same
                                                            // column -- a pitch of 256 floats would put them all on one bank
    float* Zs = Xs + Ctot * XW;                             // [N][ZP]
    float* red = Zs + a.N * ZP;                             // [G][NP][KT + 1]
    const int tid = threadIdx.x;
    const int G = 256 / NP;                                 // lane groups per pair
    const bool live = tid < G * NP;
    const int p = live ? tid % NP : 0, g = live ? tid / NP : 0;
    const int ci = p / a.N, n = p - ci * a.N;

    float acc[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) acc[k] = 0.f;
    float accb = 0.f;

    const int nunits = a.B * a.nQT;
    const int u0 = blockIdx.x * a.units_per_split;
    int u1 = u0 + a.units_per_split;
    if (u1 > nunits) u1 = nunits;
    for (int u = u0; u < u1; ++u) {
        const int b = u / a.nQT, qt = u - b * a.nQT;
        const int q0 = qt * TQ;
        const int t0 = q0 * SI - a.shift;
        __syncthreads();
        // ---- stage: input rows (zero outside [0, Tin)), dz rows (zero beyond Tq).  All global loads of a
        // batch are issued before the first LDS write, so a unit costs one memory round trip, not one per element;
        // dz rows are read as 16-byte vectors (row pitch and q0 are multiples of 4) ----
        for (int i0 = 0; i0 < Ctot * XW; i0 += 8 * 256) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int i = i0 + e * 256 + tid;
                const int c = i / XW, x = i - c * XW;
                const int t = t0 + x;
                v[e] = 0.f;
                if (i < Ctot * XW && t >= 0 && t < a.Tin)
                    v[e] = c < a.C0 ? ld1<X0T>(reinterpret_cast<const X0T*>(a.src0), (long long)b * a.bs0 + (long long)c * a.pitch0 + a.off0 + t)
                                    : ld1<X1T>(reinterpret_cast<const X1T*>(a.src1), (long long)b * a.bs1 + (long long)(c - a.C0) * a.pitch1 + a.off1 + t);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int i = i0 + e * 256 + tid;
                if (i < Ctot * XW) Xs[i] = v[e];
            }
        }
        for (int i0 = 0; i0 < a.N * (TQ / 4); i0 += 8 * 256) {
            f32x4 v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int i = i0 + e * 256 + tid;
                const int r = i / (TQ / 4), q = (i - r * (TQ / 4)) * 4;
                const int s = r / a.Nper, c = r - s * a.Nper;
                v[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (i < a.N * (TQ / 4) && q0 + q < a.Tq) {
                    const ZT* zp = reinterpret_cast<const ZT*>(a.dz) + (long long)s * a.zss + (long long)b * a.dzbs + (long long)c * a.dzpitch + q0 + q;
                    if (q0 + q + 3 < a.dzpitch) v[e] = ld4<ZT>(zp, 0);
                    else for (int k = 0; k < 4; ++k) if (q0 + q + k < a.Tq) v[e][k] = ld1<ZT>(zp, k);
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (q0 + q + k >= a.Tq) v[e][k] = 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int i = i0 + e * 256 + tid;
                const int r = i / (TQ / 4), q = (i - r * (TQ / 4)) * 4;
                if (i < a.N * (TQ / 4)) *reinterpret_cast<f32x4*>(&Zs[r * ZP + q]) = v[e];
            }
        }
        __syncthreads();
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
    }
    // ---- sum the lane groups of each pair (fixed order), one partial vector per workgroup ----
    __syncthreads();
    if (live) {
        float* r = red + (g * NP + p) * (KT + 1);
#pragma unroll
        for (int k = 0; k < KT; ++k) r[k] = acc[k];
        r[KT] = accb;
    }
    __syncthreads();
    const int P = (a.KW * Ctot + 1) * a.N;
    float* out = a.partial + (long long)(a.split_base + blockIdx.x) * P;
    for (int i = tid; i < NP * (a.KW + 1); i += 256) {
        const int pp = i / (a.KW + 1), k = i - pp * (a.KW + 1);
        const int cc = pp / a.N, nn = pp - cc * a.N;
        if (k == a.KW && cc != 0) continue;                 // the bias sum is kept by the ci == 0 pairs
        float s = 0.f;
        for (int gg = 0; gg < G; ++gg) s += red[(gg * NP + pp) * (KT + 1) + (k == a.KW ? KT : k)];
        if (k < a.KW) out[(k * Ctot + cc) * a.N + nn] = s;
        else out[a.KW * Ctot * a.N + nn] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Streaming form for ONE or TWO input channels (the mono / stereo audio-input conv: dz is 24 .. 48 rows of up to 73715 positions
// -- 24 rows per launch --, x one or two rows; round 5: the stereo form, 31 accumulators per dz row):
// no LDS, no barrier in the unit loop.  A unit is 256 output positions; lane l of every wave owns positions
// q0 + 4 l .. + 3, wave w the dz rows [w NPW, (w + 1) NPW).  Per unit a lane loads its dz quads straight from global
// memory (one 16-byte load per row: a wave instruction reads 1 KiB contiguous) and its x window of 3 SI + KT samples
// (dword loads at immediate offsets from one address; the four waves read the same window: L1 hits), then 4 KT FMAs per
// row into KT + 1 accumulators per row.  At the end the 64 lanes of a wave are summed per accumulator with a fixed DPP
// tree (row_shr 1, 2, 4, 8, then the four row totals in order) and the workgroup writes one partial vector -- the layout
// narrow_wgrad_reduce_kernel expects.
template <int KT, int SI, int NPW, bool PF, typename ZT, int CIN>
__global__ __launch_bounds__(64 * (24 / NPW)) void narrow_stream_kernel(NarrowWgradArgs a) {
    const ZT* const dzp = reinterpret_cast<const ZT*>(a.dz);
    constexpr int XWIN = 3 * SI + KT;
    constexpr int NA = CIN * KT + 1;                       // accumulators per dz row: CIN x KT taps + the bias sum
    static_assert(NPW * NA <= 128, "one accumulator per lane of two registers");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = a.nrow0 + wave * NPW;                   // first dz row of this wave (nrow0: rows [nrow0, nrow0 + 24) per launch)

    float acc[NPW][NA];
#pragma unroll
    for (int n = 0; n < NPW; ++n)
#pragma unroll
        for (int k = 0; k < NA; ++k) acc[n][k] = 0.f;

    // row bases of this wave's dz rows (wave-uniform): source s = n / Nper, channel c = n % Nper
    long long zrow[NPW];
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
        const int nn = n0 + n < a.N ? n0 + n : 0;              // rows past N: loaded, never stored
        const int sidx = nn / a.Nper, c = nn - sidx * a.Nper;
        zrow[n] = (long long)sidx * a.zss + (long long)c * a.dzpitch;
    }
    const int nunits = a.B * a.nQT;
    const int u0 = blockIdx.x * a.units_per_split;
    int u1 = u0 + a.units_per_split;
    if (u1 > nunits) u1 = nunits;
    auto load_unit = [&](int u, float (&xv)[CIN][XWIN], f32x4 (&z)[NPW]) __attribute__((always_inline)) {
        const int b = u / a.nQT, qt = u - b * a.nQT;
        const int q = qt * 256 + 4 * lane;                      // first of this lane's four positions
        const int t0 = q * SI - a.shift;                        // input sample under tap 0 of position q
        const float* xr = a.src0 + (long long)b * a.bs0 + a.off0;
        const ZT* zb = dzp + (long long)b * a.dzbs + q;
        // interior unit (wave-uniform): every sample of every lane inside the row, every position below Tq
        const int tq_lo = qt * 256 * SI - a.shift, tq_hi = (qt * 256 + 255) * SI - a.shift + KT - 1;
        const bool interior = tq_lo >= 0 && tq_hi < a.Tin && qt * 256 + 255 < a.Tq && qt * 256 + 255 < a.dzpitch;
        if (interior) {
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                const float* xp = xr + (long long)ci * a.pitch0 + t0;
#pragma unroll
                for (int i = 0; i < XWIN; ++i) xv[ci][i] = xp[i];
            }
#pragma unroll
            for (int n = 0; n < NPW; ++n) z[n] = ld4<ZT>(zb, zrow[n]);
        } else {
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                for (int i = 0; i < XWIN; ++i) {
                    const int t = t0 + i;
                    const int tc = t < 0 ? 0 : (t > a.Tin - 1 ? a.Tin - 1 : t);
                    const float v = xr[(long long)ci * a.pitch0 + tc];
                    xv[ci][i] = (t >= 0 && t < a.Tin) ? v : 0.f;
                }
#pragma unroll
            for (int n = 0; n < NPW; ++n) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qq = q + r < a.Tq ? q + r : a.Tq - 1;
                    const float v = ld1<ZT>(dzp, (long long)b * a.dzbs + zrow[n] + qq);
                    z[n][r] = q + r < a.Tq ? v : 0.f;
                }
            }
        }
    };
    // the loads of unit u + 1 are in flight while the FMAs of unit u run (two register sets)
    float xa[CIN][XWIN], xb[CIN][XWIN];
    f32x4 za[NPW], zb2[NPW];
    if (PF && u0 < u1) load_unit(u0, xa, za);
    for (int u = u0; u < u1; ++u) {
        const bool more = PF && u + 1 < u1;
        if (!PF) load_unit(u, xa, za);
        if (more) load_unit(u + 1, xb, zb2);
#pragma unroll
        for (int n = 0; n < NPW; ++n) {
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                for (int k = 0; k < KT; ++k)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[n][ci * KT + k] = fmaf(xa[ci][r * SI + k], za[n][r], acc[n][ci * KT + k]);
            acc[n][CIN * KT] += (za[n][0] + za[n][1]) + (za[n][2] + za[n][3]);
        }
        if (more) {
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                for (int i = 0; i < XWIN; ++i) xa[ci][i] = xb[ci][i];
#pragma unroll
            for (int n = 0; n < NPW; ++n) za[n] = zb2[n];
        }
    }
    // ---- 64 lanes -> 1 per accumulator: rows of 16 lanes by DPP shifts (lane 15 of a row ends up with the row's sum),
    // then the four row totals in row order; total j is kept by lane j (two registers: 64 + the rest) ----
    const int P = (a.KW * CIN + 1) * a.N;
    float* out = a.partial + (long long)(a.split_base + blockIdx.x) * P;
    float keep0 = 0.f, keep1 = 0.f;
#pragma unroll
    for (int n = 0; n < NPW; ++n)
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            float v = acc[n][k];
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, false));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xF, 0xF, false));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xF, 0xF, false));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xF, 0xF, false));
            const int vi = __builtin_bit_cast(int, v);
            const float tot = ((__builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 15)) +
                                __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 31))) +
                               __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 47))) +
                              __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 63));
            const int j = n * NA + k;
            if (j < 64) keep0 = lane == j ? tot : keep0;
            else keep1 = lane == j - 64 ? tot : keep1;
        }
    // lane j holds accumulator j = n * NA + (ci * KT + k) of this wave: weights out[(k * CIN + ci) * N + n0 + n] (k < KW),
    // bias out[KW * CIN * N + n0 + n]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = lane + 64 * h;
        if (j < NPW * NA) {
            const int n = j / NA, kk = j - n * NA;
            const float v = h ? keep1 : keep0;
            if (n0 + n < a.N && wave * NPW + n < 24) {
                if (kk == CIN * KT) out[a.KW * CIN * a.N + n0 + n] = v;
                else {
                    const int ci = kk / KT, k = kk - ci * KT;
                    if (k < a.KW) out[(k * CIN + ci) * a.N + n0 + n] = v;
                }
            }
        }
    }
}

#ifndef WUN_NS_NPW
#define WUN_NS_NPW 4      /* dz rows per wave (24 / NPW waves per workgroup); measured 3 / 4 / 6: 42.9 / 36.8 / 37.5 us on the stride-2 part */
#define WUN_NS_PF false  /* two register sets (loads of the next unit during the FMAs) */
#endif
// shapes the streaming form serves: one input channel, all rows on four waves of NPW = 6, taps <= 15
static bool narrow_stream_ok(const NarrowWgradArgs& a) {
    static const bool off = getenv("WUN_NO_NARROW_STREAM") != nullptr && atoi(getenv("WUN_NO_NARROW_STREAM")) != 0;
    // (the kernel reads src0 only, in 256-position units)
    static_assert(WUN_NW_TQ == 256, "narrow_stream_kernel walks units of 4 positions x 64 lanes");
    // one or two input channels (the reference's mono / stereo audio), up to 48 dz rows (two launches of 24)
    return !off && (a.C0 == 1 || a.C0 == 2) && a.C1 == 0 && a.N <= 48 && a.KW >= 4 && a.KW <= 15;
}

// out element (k, ci, n = s*Nper + c) -> source s: weights [K][Ctot][Nper] at woff[s], bias at boff[s].
// One workgroup per output element: 256 threads sum the splits (thread t takes splits t, t+256, ...), then a
// fixed-order tree through LDS -- deterministic, and ~nsplit/256 dependent loads per thread instead of nsplit.
__global__ __launch_bounds__(256) void narrow_wgrad_reduce_kernel(const float* partial, int nsplit, int KW, int Ctot, int N,
                                                                  int Nper, float* grads, long long w0, long long w1,
                                                                  long long w2, long long w3, long long b0, long long b1,
                                                                  long long b2, long long b3) {
    __shared__ float red[256];
    const long long woff[4] = {w0, w1, w2, w3}, boff[4] = {b0, b1, b2, b3};
    const int P = (KW * Ctot + 1) * N;
    const int i = blockIdx.x;
    float s = 0.f;
    for (int k = threadIdx.x; k < nsplit; k += 256) s += partial[(long long)k * P + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const int n = i % N, r = i / N;
    const int src = n / Nper, c = n - src * Nper;
    if (r < KW * Ctot) grads[woff[src] + (long long)r * Nper + c] = red[0];
    else grads[boff[src] + c] = red[0];
}

// (the plan keeps the LDS-staged form off the CUs the bf16 weight gradient is using: wun_plan.hip, run_narrow_wgrad)
bool narrow_wgrad_uses_lds(const NarrowWgradArgs& a) { return !narrow_stream_ok(a); }

bool narrow_wgrad_supported(const NarrowWgradArgs& a) {
    const int Ctot = a.C0 + a.C1;
    if (Ctot * a.N > 256 || Ctot * a.N < 1) return false;
    if (a.KW < 1 || a.KW > 15) return false;
    if (a.stride != 1 && a.stride != 2) return false;
    if (a.stride == 2 && a.KW > 15) return false;
    if (a.N / a.Nper > 4 || a.N % a.Nper != 0) return false;
    // dz rows are read as 16-byte vectors
    if ((a.dzpitch & 3) || (a.dzbs & 3) || (a.zss & 3) || (reinterpret_cast<uintptr_t>(a.dz) & 15)) return false;
    return true;
}

static inline int narrow_kt(int KW) { return KW <= 3 ? 3 : 15; }

size_t narrow_wgrad_lds(const NarrowWgradArgs& a) {
    const int KT = narrow_kt(a.KW), SI = a.stride;
    const int XV = (3 * SI + KT + 3) / 4;
    const int XW = (WUN_NW_TQ * SI + KT - 1 + 4 * XV + 3) / 4 * 4;
    const int Ctot = a.C0 + a.C1, NP = Ctot * a.N, G = 256 / NP;
    return sizeof(float) * ((size_t)Ctot * XW + (size_t)a.N * (WUN_NW_TQ + 4) + (size_t)G * NP * (KT + 1));
}

int narrow_wgrad_units(const NarrowWgradArgs& a) { return a.B * ((a.Tq + WUN_NW_TQ - 1) / WUN_NW_TQ); }

long long narrow_wgrad_partial_floats(const NarrowWgradArgs& a) { return (long long)(a.KW * (a.C0 + a.C1) + 1) * a.N; }

int narrow_wgrad_pick_nsplit(const NarrowWgradArgs& a) {
    const int units = narrow_wgrad_units(a);
    // ~4 workgroups per CU, each streaming >= 1 unit; the streaming form pays a 64-lane reduction of every accumulator
    // per workgroup: one workgroup per CU, >= 4 units each (measured 256 / 512 / 1024: 36.8 / 44.6 / 51.3 us)
    int cap = narrow_stream_ok(a) ? 256 : 1024;
    if (const char* e = getenv("WUN_NARROW_SPLITS")) cap = atoi(e);
    int ns = units < cap ? units : cap;
    return ns < 1 ? 1 : ns;
}

// a.nsplit / a.split_base / a.partial set by the caller
hipError_t launch_narrow_wgrad(NarrowWgradArgs a, hipStream_t s) {
    if (!narrow_wgrad_supported(a)) return hipErrorInvalidValue;
    a.nQT = (a.Tq + WUN_NW_TQ - 1) / WUN_NW_TQ;
    const int units = a.B * a.nQT;
    a.units_per_split = (units + a.nsplit - 1) / a.nsplit;
    if (narrow_stream_ok(a)) {
        char tag[160];
        snprintf(tag, sizeof(tag), "C=%d N=%d T=%d K=%d stride=%d B=%d nsplit=%d stream", a.C0, a.N, a.Tq, a.KW, a.stride, a.B, a.nsplit);
        prof_scope_begin("narrow_wgrad_kernel", 2.0 * a.KW * (double)a.C0 * a.N * (double)a.Tq * a.B, s, tag,
                         (double)a.B * a.Tq * (((a.et & 4) ? 2.0 : 4.0) * a.N + 4.0 * a.C0 * a.stride));
        if (a.et & ~4) { prof_scope_end(s); return hipErrorInvalidValue; }      // (the audio itself is always fp32)
        const dim3 blk(64 * (24 / WUN_NS_NPW));
        // 24 dz rows per launch (the second launch of a 48-row layer writes the other rows of the same partial vectors)
        for (int r0 = 0; r0 < a.N; r0 += 24) {
            a.nrow0 = r0;
#define WUN_NSK(SI, ZT, CI) hipLaunchKernelGGL((narrow_stream_kernel<15, SI, WUN_NS_NPW, WUN_NS_PF, ZT, CI>), dim3((unsigned)a.nsplit), blk, 0, s, a)
            if (a.C0 == 1) {
                if (a.et & 4) { if (a.stride == 2) WUN_NSK(2, bf16_t, 1); else WUN_NSK(1, bf16_t, 1); }
                else { if (a.stride == 2) WUN_NSK(2, float, 1); else WUN_NSK(1, float, 1); }
            } else {
                if (a.et & 4) { if (a.stride == 2) WUN_NSK(2, bf16_t, 2); else WUN_NSK(1, bf16_t, 2); }
                else { if (a.stride == 2) WUN_NSK(2, float, 2); else WUN_NSK(1, float, 2); }
            }
#undef WUN_NSK
        }
        prof_scope_end(s);
        return hipGetLastError();
    }
    const size_t lds = narrow_wgrad_lds(a);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const int KT = narrow_kt(a.KW);
    char tag[160];
    snprintf(tag, sizeof(tag), "C=%d N=%d T=%d K=%d stride=%d B=%d nsplit=%d", a.C0 + a.C1, a.N, a.Tq, a.KW, a.stride, a.B, a.nsplit);
    // (bandwidth-bound: dz rows + the input rows they touch streamed once)
    prof_scope_begin("narrow_wgrad_kernel", 2.0 * a.KW * (double)(a.C0 + a.C1) * a.N * (double)a.Tq * a.B, s, tag,
                     (double)a.B * a.Tq * (((a.et & 4) ? 2.0 : 4.0) * a.N + (((a.et & 1) ? 2.0 : 4.0) * a.C0 + ((a.et & 2) ? 2.0 : 4.0) * a.C1) * a.stride));
    // element-type combinations the plan produces: all fp32; the bf16 mode's audio-input conv (fp32 audio, bf16 dz);
    // the bf16 mode's head (fp32 audio + bf16 feature map, fp32 d(pre-activation))
    if (a.et != 0 && a.et != 4 && a.et != 2) { prof_scope_end(s); return hipErrorInvalidValue; }
#define WUN_NWL(K, S) WUN_NWE(K, S, 0) WUN_NWE(K, S, 4) WUN_NWE(K, S, 2)
#define WUN_NWE(K, S, E) \
    if (KT == K && a.stride == S && a.et == E) { \
        auto kern = narrow_wgrad_kernel<K, S, E>; \
        static size_t allowed = 64 * 1024; \
        if (lds > allowed) { \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e; \
            allowed = lds; \
        } \
        hipLaunchKernelGGL(kern, dim3((unsigned)a.nsplit), dim3(256), lds, s, a); \
    }
    WUN_NWL(3, 1) WUN_NWL(3, 2) WUN_NWL(15, 1) WUN_NWL(15, 2)
#undef WUN_NWL
#undef WUN_NWE
    prof_scope_end(s);
    return hipGetLastError();
}

hipError_t launch_narrow_wgrad_reduce(const NarrowWgradArgs& a, const float* partial, int nsplit, float* grads,
                                      const long long* woff, const long long* boff, hipStream_t s) {
    const int P = (int)narrow_wgrad_partial_floats(a);
    hipLaunchKernelGGL(narrow_wgrad_reduce_kernel, dim3((unsigned)P), dim3(256), 0, s, partial, nsplit, a.KW,
                       a.C0 + a.C1, a.N, a.Nper, grads, woff[0], woff[1], woff[2], woff[3], boff[0], boff[1], boff[2], boff[3]);
    return hipGetLastError();
}

}  // namespace wun
