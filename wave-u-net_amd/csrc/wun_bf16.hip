// gfx950 bf16-MFMA speed mode of the implicit-GEMM conv (forward convs and input gradients).
//
//   D[time 16][cout 16] += A[time][k] * B[k][cout]   on v_mfma_f32_16x16x32_bf16 (fp32 accumulate)
//   k = (tap, input channel); one MFMA consumes 32 input channels of one tap.
//
// ACTIVATIONS AND THEIR GRADIENTS LIVE IN HBM AS bf16 (round 5): the input rows this kernel reads are bf16 NCW rows
// (rounded once, by the epilogue that produced them: nearest even, v_cvt_pk_bf16_f32), and its own epilogue -- fp32
// accumulators, bias, activation, mask, accumulate exactly as in the exact-fp32 kernel (the accumulator fragment layout
// of the two MFMA shapes is identical) -- rounds once more on the store (ConvArgs.obf; obf = 0 stores fp32: the
// single-operator tests).  Master weights, weight gradients and the Adam state stay fp32.  Lane l of a wave supplies A[i = l&15][k = 8*(l>>4) .. +7] and
// B[k = 8*(l>>4) .. +7][j = l&15] and receives D[i = 4*(l>>4)+r][j = l&15].
//
// LDS images (per pipeline buffer):
//   X  [plane][row = time][32*NCK channels] bf16, row pitch 64*NCK + 32 B (the 32 B pad makes the
//      ds_read_b128 lane groups of gfx950 conflict-free at every tap offset); the stride-2 loader keeps
//      even / odd input samples in two planes so tap k reads plane k&1 at row q + (k>>1)
//   W  [tap][channel group of 8][cout][8 channels] bf16 -- copied verbatim from the pre-packed
//      bf16 weight image (pack_bf16_kernel), so a B fragment is one aligned 16-byte read and 16
//      lanes read 256 contiguous bytes
// Pipeline: stage = (NCK chunks of 32 channels) x all taps; while the MFMAs of stage s run, the weights
// of stage s+1 stream global -> LDS (global_load_lds) and its input window is fetched into registers
// (dword loads: two consecutive time steps of one channel), transposed to [time][8 channels] with v_perm_b32 and
// written to the other LDS buffer afterwards; one barrier per stage.
#include "wun_internal.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace wun {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two fp32 -> packed bf16 pair, round-to-nearest-even: one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){lo, hi}, bf16x2));
}

__device__ __forceinline__ int xcd_block(int bid, int grid) {
    const int per = grid >> 3, rem = grid & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
}

#define WUN_BF_KMAX 15         // taps

// F_ACCUM applies to the row positions [lo, lo + len) only (ConvArgs.acc_lo / acc_len)
__device__ __forceinline__ bool conv_acc_pos(int lo, unsigned len, int pos) { return (unsigned)(pos - lo) < len; }

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

// X items (PAIRS of 16-byte slots = 8 channels of two consecutive time steps) a thread stages per stage, by tile height
template <int MT> struct BfXit { static constexpr int v = MT == 4 ? 6 : (MT == 2 ? 4 : 3); };
static inline int bf16_xit(int mt) { return mt == 4 ? 6 : (mt == 2 ? 4 : 3); }
// staged item count of a stage: channel groups x time pairs (one spare pair for an odd first sample), the pairs of a
// group padded to whole waves
__host__ __device__ static inline int bf16_pairs64(int pr) { return ((pr + 2) / 2 + 63) & ~63; }

// Stage = NCK chunks of 32 input channels x ALL taps (so a stage carries enough MFMAs -- ~12 tap-chunks --
// to cover a global-memory round trip).  Weights go global -> LDS directly (global_load_lds, 16 bytes per
// lane: no registers, the image is already in LDS order); the input window is fetched into registers while
// the previous MFMAs run and written (bf16-rounded) afterwards.
//
// One output tile per workgroup; with S > 1 stages the weights and the input window of stage s+1 stream in while
// stage s computes (both double-buffered).  (A weights-stationary walk over several time tiles of one column tile was
// built and measured -- no gain on the one-stage layers it applies to -- and removed.)
// MODE: 0 stride-1 loader, 1 stride-2 (even / odd planes) loader, 2 fused two-phase transposed conv (stride-1 loader).
// Compile-time, like the schedule: at bf16 MFMA rates the narrow layers are bound by the instruction stream around the
// MFMAs, and four uniform run-time branches in the stage loop were measured to cost them 4 %.
// NCK (chunks of 32 input channels per stage: 1..3) is compile-time as well: row pitch, slab sizes and the k-step walk
// of the MFMA loop become immediates.
// KT: the tap count when it is one of the reference's shipped filter sizes (15 / 5, 8 = a phase of the transposed
// 15-tap conv), 0 = run-time taps: tap loops, slab sizes and the row count of the input window are then constants too.
template <int MT, int NW, int MODE, int NCK, int KT>
__global__ __launch_bounds__(256, (MT == 4 ? 2 : (MT == 2 ? (NW <= 3 ? 3 : 2) : (NW <= 3 ? 4 : 3)))) void conv_bf16_kernel(ConvArgs a, int nTT, int nNT, int ROWS_arg) {
    const int ROWS = KT > 0 ? (MODE == 1 ? 4 * MT * 16 + (KT + 1) / 2 : 4 * MT * 16 + KT - 1) : ROWS_arg;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WT = 4;
    constexpr int TT = WT * MT * 16;
    constexpr int NT = NW * 16;
    constexpr int XIT = BfXit<MT>::v;
    constexpr bool deint = MODE == 1;
    constexpr int planes = deint ? 2 : 1;
    const int KW = KT > 0 ? KT : a.KW;
    constexpr int XPB = 64 * NCK + 32;                      // bytes per X row: conflict-free for ds_read_b128 at every tap
    constexpr int C8S = 4 * NCK;                            // 8-channel groups per stage
    const int Ctot = a.C0 + a.C1;
    constexpr int CKW = 32 * NCK;
    const int S = (Ctot + CKW - 1) / CKW;

    const int xbytes = planes * ROWS * XPB;
    const int wbytes = KW * C8S * NT * 16;
    const int nxb = S > 1 ? 2 : 1;                          // X / W buffers: 2 only when there are several stages
    unsigned char* Xs = smem;
    unsigned char* Ws = smem + nxb * xbytes;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = xcd_block((int)blockIdx.x, (int)gridDim.x);
    const int nt = bid % nNT;
    const int seg = bid / nNT;
    // fused two-phase transposed stride-2 conv (F_PHASE2): the weight matrix has 2*N columns (phase 0 | phase 1 of the N
    // output channels); a workgroup takes NT/2 channels x both phases -- column tiles [0, NW/2) are phase 0 (outputs
    // t = 2q), tiles [NW/2, NW) phase 1 (t = 2q + 1) of the SAME channels, so a lane owns 8 consecutive output samples
    constexpr bool phase2 = MODE == 2;
    const int n0 = phase2 ? nt * (NT / 2) : nt * NT;
    const int wt0 = wave * MT * 16;
    const int tix0 = seg;

    f32x4 acc[MT][NW];
    unsigned xreg[XIT][8];
    // X items: item it = tid + i*256 is (channel group of 8, time PAIR), the pairs of a group padded to a multiple of 64
    // so that the channel group of an item is uniform across a wave (the 8 row offsets of a load batch are SCALAR values,
    // a lane contributes only its 32-bit time offset).  Lane l of a load instruction fetches ONE dword = the bf16 samples
    // (u, u + 1) of one channel row at an EVEN element index u of the row (rows are 16-byte aligned), so a wave reads 256
    // contiguous bytes per channel; the 8 dwords of an item are transposed with v_perm_b32 into two 16-byte LDS slots
    // (8 channels of time u, 8 channels of time u + 1).  `par` = parity of the row element under the tile's first
    // sample: with par = 1 the first pair starts one sample early and its first element is dropped.
    const int PR = planes * ROWS;
    const int PRP64 = bf16_pairs64(PR);
    const int nxitems = C8S * PRP64;                        // <= XIT*256 (launcher)
    const int par = (a.off0 - a.shift) & 1;                 // tile-invariant: q0 is even (off0 == off1 for two sources: launcher)
    // tile-invariant per-item state: xfl[i] = pair index | channel group << 24 | live << 29 | element e inside the image << (30 + e);
    // xlo0/xlo1[i] = LDS byte offsets of the two slots.  Per tile: xti[i] = clamped row element of the pair | sample e
    // inside [0, Tin) << (28 + e).
    unsigned xfl[XIT];
    int xlo0[XIT], xlo1[XIT], xti[XIT];
#pragma unroll
    for (int i = 0; i < XIT; ++i) {
        const int it0 = wave * 64 + i * 256;                // wave-uniform
        const int c8l = it0 / PRP64;
        const int pp = it0 - c8l * PRP64 + lane;
        const bool live = it0 < nxitems;
        const int tr0 = 2 * pp - par, tr1 = tr0 + 1;        // conv-relative sample index of the two elements
        const bool in0 = live && tr0 >= 0 && tr0 < PR, in1 = live && tr1 >= 0 && tr1 < PR;
        xfl[i] = (unsigned)pp | ((unsigned)c8l << 24) | ((live ? 1u : 0u) << 29) | ((in0 ? 1u : 0u) << 30) | ((in1 ? 1u : 0u) << 31);
        const int r0 = in0 ? tr0 : 0, r1 = in1 ? tr1 : 0;
        xlo0[i] = (deint ? ((r0 & 1) * ROWS + (r0 >> 1)) : r0) * XPB + c8l * 16;
        xlo1[i] = (deint ? ((r1 & 1) * ROWS + (r1 >> 1)) : r1) * XPB + c8l * 16;
    }
    // Input rows are fetched with raw buffer loads: resource = this excerpt's tensor, SGPR offset = channel row
    // (one s_mul per load), VGPR offset = the lane's byte offset inside the row.  A flat load needs ~8 scalar instructions
    // of 64-bit address arithmetic per row pointer; at bf16 MFMA rates that arithmetic, not the matrix pipe, set the stage time.
    const unsigned char* const xsrc0 = reinterpret_cast<const unsigned char*>(a.src0);
    const unsigned char* const xsrc1 = reinterpret_cast<const unsigned char*>(a.src1);
    __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)xsrc0, 0, 0x7FFFFFFE, 0x00020000);
    __amdgpu_buffer_rsrc_t rs1 = rs0;
    const int pb0 = a.pitch0 * 2, pb1 = a.pitch1 * 2;      // row pitches in bytes
    const int umax = (a.pitch0 < a.pitch1 || a.src1 == nullptr ? a.pitch0 : a.pitch1) - 2;   // last pair inside a row (pitches are even)
    auto set_tile = [&](int tix, int& b, int& q0) {
        b = tix / nTT;
        q0 = (tix - b * nTT) * TT;
        rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)(xsrc0 + (long long)b * a.bs0 * 2), 0, 0x7FFFFFFE, 0x00020000);
        rs1 = (a.src1 != nullptr)
                  ? __builtin_amdgcn_make_buffer_rsrc((void*)(xsrc1 + (long long)b * a.bs1 * 2), 0, 0x7FFFFFFE, 0x00020000)
                  : rs0;
        const int s0e = a.off0 + (deint ? 2 * q0 : q0) - a.shift - par;      // even row element under conv-relative sample -par
#pragma unroll
        for (int i = 0; i < XIT; ++i) {
            const int u = s0e + 2 * (int)(xfl[i] & 0xFFFFFFu);
            const int t = u - a.off0;                                        // time of element 0 on the source's own axis
            const int uc = u < 0 ? 0 : (u > umax ? umax : u);                // Tin + off < 2^23 (checked by the launcher)
            xti[i] = uc | ((t >= 0 && t < a.Tin ? 1 : 0) << 28) | ((t + 1 >= 0 && t + 1 < a.Tin ? 1 : 0) << 29);
        }
    };

    // ---- X: global -> registers, registers -> LDS ([time][8 channels] slots, zero fill) ----
    auto load_x = [&](int st) {
#pragma unroll
        for (int i = 0; i < XIT; ++i) {
            if (i * 256 < nxitems) {                         // uniform
                const int tb = (xti[i] & 0xFFFFFF) * 2;      // byte offset of the pair inside its row
                const int cbase = st * CKW + __builtin_amdgcn_readfirstlane((int)((xfl[i] >> 24) & 15u)) * 8;     // wave-uniform
                // a group of 8 channels lies in ONE source (C0 % 8 == 0, checked by the launcher)
                const bool s1 = cbase >= a.C0;
                const __amdgpu_buffer_rsrc_t rs = s1 ? rs1 : rs0;
                const int c0 = s1 ? cbase - a.C0 : cbase, cmax = (s1 ? a.C1 : a.C0) - 1, pb = s1 ? pb1 : pb0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    int c = c0 + e;
                    c = c < cmax ? c : cmax;                 // channels past the tensor: any valid row (zero-filled at the store)
                    xreg[i][e] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, tb, c * pb, 0);
                }
            }
        }
    };
    auto store_x = [&](int st, int buf) {
        unsigned char* xb = Xs + buf * xbytes;
#pragma unroll
        for (int i = 0; i < XIT; ++i) {
            if (i * 256 < nxitems && ((xfl[i] >> 29) & 1u)) {
                const int cbase = st * CKW + __builtin_amdgcn_readfirstlane((int)((xfl[i] >> 24) & 15u)) * 8;
                unsigned d[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) d[e] = (cbase + e < Ctot) ? xreg[i][e] : 0u;       // (uniform selects)
                // dword e = samples (u, u + 1) of channel e: low halves -> slot of time u, high halves -> slot of time u + 1
                const bool t0ok = (xti[i] >> 28) & 1, t1ok = (xti[i] >> 29) & 1;
                u32x4 lo = {__builtin_amdgcn_perm(d[1], d[0], 0x05040100u), __builtin_amdgcn_perm(d[3], d[2], 0x05040100u),
                            __builtin_amdgcn_perm(d[5], d[4], 0x05040100u), __builtin_amdgcn_perm(d[7], d[6], 0x05040100u)};
                u32x4 hi = {__builtin_amdgcn_perm(d[1], d[0], 0x07060302u), __builtin_amdgcn_perm(d[3], d[2], 0x07060302u),
                            __builtin_amdgcn_perm(d[5], d[4], 0x07060302u), __builtin_amdgcn_perm(d[7], d[6], 0x07060302u)};
                if (!t0ok) lo = (u32x4){0u, 0u, 0u, 0u};
                if (!t1ok) hi = (u32x4){0u, 0u, 0u, 0u};
                if ((xfl[i] >> 30) & 1u) *reinterpret_cast<u32x4*>(xb + xlo0[i]) = lo;
                if ((xfl[i] >> 31) & 1u) *reinterpret_cast<u32x4*>(xb + xlo1[i]) = hi;
            }
        }
    };
    // ---- W: packed bf16 image [KW][C8p][Npad][8] -> LDS [tap][channel group][cout][8], 16 bytes per lane ----
    const unsigned short* Wb = reinterpret_cast<const unsigned short*>(a.W);
    constexpr int per_tap = C8S * NT;                       // 16-byte items per tap (<= 768)
    int wofs[3];                                            // tap-invariant element offset of this lane's items
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int r = k * 256 + tid;
        const int c8l = r / NT, n = r - c8l * NT;           // NT is a compile-time constant: no integer division
        const int gcol = !phase2 ? n0 + n : (n < NT / 2 ? n0 + n : a.N + n0 + (n - NT / 2));   // column in the image
        wofs[k] = r < per_tap ? (c8l * a.wb_npad + gcol) * 8 : -1;
    }
    auto dma_w = [&](int st, int buf) {
        unsigned char* wbuf = Ws + buf * wbytes;
        for (int j = 0; j < KW; ++j) {
            const unsigned short* wj = Wb + (((long long)j * a.wb_c8p + st * C8S) * a.wb_npad) * 8;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k * 256 < per_tap && wofs[k] >= 0)
                    __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(wj + wofs[k]),
                                                     (lds_void_t*)(wbuf + (j * per_tap + k * 256 + wave * 64) * 16), 16, 0, 0);
            }
        }
    };

    // ---- MFMA over all (tap, channel sub-chunk) k-steps of one stage; the operands of k-step i+1 are read from
    // LDS before the MFMAs of k-step i issue (two register sets), so LDS latency hides behind the matrix pipe ----
    auto run_stage = [&](int xbuf, int wbufi) {
        const unsigned char* xb = Xs + xbuf * xbytes + (wt0 + li) * XPB + lg * 16;
        const unsigned char* wbuf = Ws + wbufi * wbytes + (lg * NT + li) * 16;
        const int nsteps = KW * NCK;
        auto ldops = [&](int j, int sc, bf16x8 (&av)[MT], bf16x8 (&bv)[NW]) {
            const int pl = deint ? (j & 1) : 0;
            const int ro = deint ? (j >> 1) : j;
            const unsigned char* xa = xb + (pl * ROWS + ro) * XPB + sc * 64;
            const unsigned char* wp = wbuf + ((j * C8S + sc * 4) * NT) * 16;
#pragma unroll
            for (int m = 0; m < MT; ++m) av[m] = *reinterpret_cast<const bf16x8*>(xa + m * 16 * XPB);
#pragma unroll
            for (int n = 0; n < NW; ++n) bv[n] = *reinterpret_cast<const bf16x8*>(wp + n * 16 * 16);
        };
        auto mm = [&](const bf16x8 (&av)[MT], const bf16x8 (&bv)[NW]) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NW; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[m], bv[n], acc[m][n], 0, 0, 0);
        };
        bf16x8 a0[MT], b0[NW], a1[MT], b1[NW];
        int j = 0, sc = 0;
        auto advance = [&]() { if (++sc == NCK) { sc = 0; ++j; } };
        ldops(0, 0, a0, b0);
#pragma unroll 1                                             // (fully unrolled, 30-45 k-steps of hoisted LDS reads spill)
        for (int i = 0; i < nsteps; i += 2) {
            advance();
            if (i + 1 < nsteps) ldops(j, sc, a1, b1);
            mm(a0, b0);
            advance();
            if (i + 2 < nsteps) ldops(j, sc, a0, b0);
            if (i + 1 < nsteps) mm(a1, b1);
        }
    };

    // ---- epilogue of one tile (fp32 arithmetic, the exact-fp32 kernel's; OB: destinations, masks and the decimated copy
    // hold bf16 -- rounded to nearest even on the store, 4 values per 8-byte store).  Every launch argument it needs
    // is read into a local first: selecting between `a.dst0` and `a.dst1` per lane otherwise compiles to a VECTOR load
    // of the kernel-argument segment followed by s_waitcnt vmcnt(0) -- which also drains the input prefetch of the next
    // tile and serialises the whole pipeline (measured: phases became exactly additive) ----
    const bool lrelu = (a.flags & F_LRELU) != 0;
    const bool accum = (a.flags & F_ACCUM) != 0;
    const bool vec = (a.flags & F_VEC4) != 0;
    float* const e_dst0 = a.dst0; float* const e_dst1 = a.dst1;
    const float* const e_msk0 = a.msk0; const float* const e_msk1 = a.msk1;
    const float* const e_bias = a.bias;
    float* const e_dec = a.dec;
    const long long e_obs0 = a.obs0, e_obs1 = a.obs1, e_decbs = a.decbs;
    const int e_op0 = a.opitch0, e_op1 = a.opitch1, e_oo0 = a.ooff0, e_oo1 = a.ooff1, e_N = a.N, e_N0 = a.N0,
              e_Tout = a.Tout, e_os = a.ostride, e_decp = a.decpitch;
    const int e_acc_lo = a.acc_lo; const unsigned e_acc_len = a.acc_len;
    float e_bv[NW];                                         // this lane's bias values, loaded once (a load inside the
#pragma unroll                                              // tile loop would wait on vmcnt(0) and drain the input prefetch)
    for (int n = 0; n < NW; ++n) {
        const int ncol = n0 + n * 16 + li;
        e_bv[n] = (e_bias != nullptr && ncol < e_N) ? e_bias[ncol] : 0.f;
    }
    const int e_Tlim = a.Tlim;
    auto epilogue = [&](int b, int q0, auto obtag) {
        constexpr bool OB = decltype(obtag)::value;
        using ET = std::conditional_t<OB, bf16_t, float>;
        if (phase2) {
            if constexpr ((NW % 2) == 0) {
                ET* const dst = reinterpret_cast<ET*>(e_dst0);
                const ET* const msk = reinterpret_cast<const ET*>(e_msk0);
#pragma unroll
                for (int n = 0; n < NW / 2; ++n) {
                    const int ncol = n0 + n * 16 + li;
                    if (ncol >= e_N) continue;
                    const long long rowbase = (long long)b * e_obs0 + (long long)ncol * e_op0 + e_oo0;
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const int q = q0 + wt0 + m * 16 + lg * 4;
                        const int t0 = 2 * q;
                        float v[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v[2 * r] = acc[m][n][r]; v[2 * r + 1] = acc[m][NW / 2 + n][r]; }
                        if (vec && t0 + 7 < e_Tlim) {
                            const long long idx = rowbase + t0;
                            if (msk != nullptr) {
                                const f32x4 m0 = ld4<ET>(msk, idx), m1 = ld4<ET>(msk, idx + 4);
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    v[r] *= (m0[r] > 0.f) ? 1.f : 0.2f;
                                    v[4 + r] *= (m1[r] > 0.f) ? 1.f : 0.2f;
                                }
                            }
                            if (accum) {
                                const f32x4 o0 = ld4<ET>(dst, idx), o1 = ld4<ET>(dst, idx + 4);
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    if (conv_acc_pos(e_acc_lo, e_acc_len, e_oo0 + t0 + r)) v[r] += o0[r];
                                    if (conv_acc_pos(e_acc_lo, e_acc_len, e_oo0 + t0 + 4 + r)) v[4 + r] += o1[r];
                                }
                            }
                            st4<ET>(dst, idx, (f32x4){v[0], v[1], v[2], v[3]});
                            st4<ET>(dst, idx + 4, (f32x4){v[4], v[5], v[6], v[7]});
                        } else {
#pragma unroll
                            for (int r = 0; r < 8; ++r) {
                                if (t0 + r < e_Tlim) {
                                    const long long idx = rowbase + t0 + r;
                                    float x = v[r];
                                    if (msk != nullptr) x *= (ld1<ET>(msk, idx) > 0.f) ? 1.f : 0.2f;
                                    if (accum && conv_acc_pos(e_acc_lo, e_acc_len, e_oo0 + t0 + r)) x += ld1<ET>(dst, idx);
                                    st1<ET>(dst, idx, x);
                                }
                            }
                        }
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            const int ncol = n0 + n * 16 + li;
            if (ncol >= e_N) continue;
            const float bvv = e_bv[n];
            const bool first = ncol < e_N0;
            ET* const dst = reinterpret_cast<ET*>(first ? e_dst0 : e_dst1);
            const ET* const msk = reinterpret_cast<const ET*>(first ? e_msk0 : e_msk1);
            const long long rowbase = first ? (long long)b * e_obs0 + (long long)ncol * e_op0 + e_oo0
                                            : (long long)b * e_obs1 + (long long)(ncol - e_N0) * e_op1 + e_oo1;
            const int oo = first ? e_oo0 : e_oo1;
            ET* decrow = (e_dec != nullptr && first) ? reinterpret_cast<ET*>(e_dec) + (long long)b * e_decbs + (long long)ncol * e_decp : nullptr;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int q = q0 + wt0 + m * 16 + lg * 4;
                if (vec && q + 3 < e_Tout) {
                    f32x4 v = acc[m][n];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] += bvv;
                        if (lrelu) v[r] = fmaxf(0.2f * v[r], v[r]);
                    }
                    const long long idx = rowbase + q;
                    if (msk != nullptr) {
                        const f32x4 mk = ld4<ET>(msk, idx);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] *= (mk[r] > 0.f) ? 1.f : 0.2f;
                    }
                    if (accum) {
                        const f32x4 o = ld4<ET>(dst, idx);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (conv_acc_pos(e_acc_lo, e_acc_len, oo + q + r)) v[r] += o[r];
                    }
                    st4<ET>(dst, idx, v);
                    if (decrow != nullptr) {
                        if constexpr (OB) {
                            *reinterpret_cast<unsigned*>(decrow + (q >> 1)) = bf_pack2(v[0], v[2]);
                        } else {
                            decrow[q >> 1] = v[0];
                            decrow[(q >> 1) + 1] = v[2];
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (q + r < e_Tout) {
                            float v = acc[m][n][r] + bvv;
                            if (lrelu) v = fmaxf(0.2f * v, v);
                            const long long idx = rowbase + (long long)(q + r) * e_os;
                            if (msk != nullptr) v *= (ld1<ET>(msk, idx) > 0.f) ? 1.f : 0.2f;
                            if (accum && conv_acc_pos(e_acc_lo, e_acc_len, oo + (q + r) * e_os)) v += ld1<ET>(dst, idx);
                            st1<ET>(dst, idx, v);
                            if (decrow != nullptr && ((q + r) & 1) == 0) st1<ET>(decrow, (q + r) >> 1, v);
                        }
                    }
                }
            }
        }
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NW; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };

    int b, q0;
    set_tile(tix0, b, q0);
    dma_w(0, 0);
    { load_x(0); store_x(0, 0); }
    __syncthreads();
    // ---- one output tile: weights and input window of stage st+1 stream in under the MFMAs of stage st ----
    zero_acc();
    for (int st = 0; st < S; ++st) {
        const bool has_next = st + 1 < S;
        if (has_next) {
            dma_w(st + 1, (st + 1) & 1);
            load_x(st + 1);
        }
        run_stage(st & 1, st & 1);
        if (has_next) {
            store_x(st + 1, (st + 1) & 1);
            __syncthreads();
        }
    }
    {
        if (a.obf) epilogue(b, q0, std::true_type{});
        else epilogue(b, q0, std::false_type{});
    }
}

// ---------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------
bool conv_bf16_supported(const ConvArgs& a) {
    if ((a.flags & F_PHASE2) && !(a.ostride == 1 && a.dst1 == nullptr && a.loader == LOADER_DIRECT)) return false;
    if (a.C0 + a.C1 < 8) return false;                    // the 1-/2-channel audio input has its own kernel (first_conv_kernel)
    if (a.KW < 1 || a.KW > WUN_BF_KMAX) return false;
    if ((long long)a.Tin + a.off0 >= (1 << 23) || (long long)a.Tin + a.off1 >= (1 << 23) || a.off0 < 0) return false;
    if (a.C1 > 0 && (a.C0 & 7) != 0) return false;         // an 8-channel group must not straddle the two sources
    if (a.C1 > 0 && a.off0 != a.off1) return false;        // one time-pair alignment for both sources (always 0 / 0 in the plan)
    if ((a.pitch0 & 1) != 0 || (a.bs0 & 1) != 0 || (a.C1 > 0 && ((a.pitch1 & 1) != 0 || (a.bs1 & 1) != 0))) return false;   // dword pairs
    if (a.pitch0 < 2 || (a.C1 > 0 && a.pitch1 < 2)) return false;
    // the excerpt's tensor is addressed with 32-bit byte offsets
    if ((long long)a.C0 * a.pitch0 * 2 >= (1ll << 31) || (long long)a.C1 * a.pitch1 * 2 >= (1ll << 31)) return false;
    return true;
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static inline int bf16_rows(const ConvArgs& a, int TT) {
    return a.loader == LOADER_DEINT ? TT + (a.KW + 1) / 2 : TT + a.KW - 1;
}
static inline size_t bf16_lds(const ConvArgs& a, int TT, int NT, int nck) {
    const int planes = a.loader == LOADER_DEINT ? 2 : 1;
    const int S = (a.C0 + a.C1 + 32 * nck - 1) / (32 * nck);
    const int nxb = S > 1 ? 2 : 1, nwb = S > 1 ? 2 : 1;
    return nxb * ((size_t)planes * bf16_rows(a, TT) * (64 * nck + 32)) + nwb * ((size_t)a.KW * 4 * nck * NT * 16);
}
// channel chunks per stage: enough tap-chunks (~12) per stage to cover a memory round trip, within the
// X-staging register budget and 160 KiB of LDS
static int bf16_pick_nck(const ConvArgs& a, int TT, int NT, int xit) {
    const int planes = a.loader == LOADER_DEINT ? 2 : 1;
    const int maxck = (a.C0 + a.C1 + 31) / 32;
    // (the HEURISTIC keeps the chunk counts of the round-2..4 kernel, whose single-sample items allowed fewer chunks per
    //  stage -- 9 / 7 / 4 items of 64-padded rows per thread: they keep the LDS footprint at two workgroups per CU; the
    //  autotuner may pick any count the pair items hold, conv_bf16_choice_ok)
    const int xit_old = TT >= 256 ? 9 : (TT >= 128 ? 7 : 4);
    auto fits = [&](int nck) {
        return bf16_pairs64(planes * bf16_rows(a, TT)) * 4 * nck <= xit * 256 &&
               ((planes * bf16_rows(a, TT) + 63) & ~63) * 4 * nck <= xit_old * 256 && bf16_lds(a, TT, NT, nck) <= 160 * 1024;
    };
    // all input channels in one stage -> weights-stationary schedule
    if (maxck <= 3 && fits(maxck)) return maxck;
    int nck = (12 + a.KW - 1) / a.KW;
    if (nck > 3) nck = 3;
    if (nck > maxck) nck = maxck;
    while (nck > 1 && !fits(nck)) --nck;
    return nck;
}

template <int MT, int NW, int MODE, int NCK, int KT>
static hipError_t conv_bf16_launch_k(const ConvArgs& a, int nTT, int nNT, int ROWS, size_t lds, long long grid, hipStream_t s) {
    auto kern = conv_bf16_kernel<MT, NW, MODE, NCK, KT>;
    static size_t lds_allowed = 64 * 1024;
    if (lds > lds_allowed) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_allowed = lds;
    }
    char nm[64], tag[160];
    snprintf(nm, sizeof(nm), "conv_bf16_kernel<%d, %d>", MT, NW);
    snprintf(tag, sizeof(tag), "C=%d N=%d T=%d K=%d ld=%d B=%d nck=%d ph2=%d grid=%lld", a.C0 + a.C1, a.N, a.Tout, a.KW, a.loader,
             a.B, NCK, MODE == 2 ? 1 : 0, grid);
    prof_scope_begin(nm, conv_flops(a), s, tag);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, a, nTT, nNT, ROWS);
    prof_scope_end(s);
    return hipGetLastError();
}

template <int MT, int NW, int MODE>
static hipError_t conv_bf16_launch_m(const ConvArgs& a, int NCK, int nTT, int nNT, int ROWS, size_t lds, long long grid, hipStream_t s) {
    // the shipped filter sizes get their own instantiation: stride-1 loader 15 / 5 taps, stride-2 loader 15 taps, the
    // fused two-phase conv 8 taps per phase (of 15); anything else takes the run-time-tap kernel
    constexpr int K1 = MODE == 2 ? 8 : 15, K2 = MODE == 0 ? 5 : 0;
#define WUN_BF_K(N) \
    if (NCK == N) { \
        if (a.KW == K1) return conv_bf16_launch_k<MT, NW, MODE, N, K1>(a, nTT, nNT, ROWS, lds, grid, s); \
        if (K2 > 0 && a.KW == K2) return conv_bf16_launch_k<MT, NW, MODE, N, K2>(a, nTT, nNT, ROWS, lds, grid, s); \
        return conv_bf16_launch_k<MT, NW, MODE, N, 0>(a, nTT, nNT, ROWS, lds, grid, s); \
    }
    WUN_BF_K(1) WUN_BF_K(2) WUN_BF_K(3)
#undef WUN_BF_K
    return hipErrorInvalidValue;
}

template <int MT, int NW>
static hipError_t conv_bf16_launch_t(ConvArgs a, hipStream_t s, int nck_force = 0) {
    constexpr int TT = 4 * MT * 16, NT = NW * 16;
    const int NCK = nck_force > 0 ? nck_force : bf16_pick_nck(a, TT, NT, BfXit<MT>::v);
    const int ROWS = bf16_rows(a, TT);
    const bool deint = a.loader == LOADER_DEINT;
    const int planes = deint ? 2 : 1;
    const size_t lds = bf16_lds(a, TT, NT, NCK);
    if (lds > 160 * 1024 || bf16_pairs64(planes * ROWS) * 4 * NCK > BfXit<MT>::v * 256) return hipErrorInvalidValue;
    const bool phase2 = (a.flags & F_PHASE2) != 0;
    if (phase2 && ((NW % 2) != 0 || deint)) return hipErrorInvalidValue;
    const int nTT = (a.Tout + TT - 1) / TT, nNT = phase2 ? (a.N + NT / 2 - 1) / (NT / 2) : (a.N + NT - 1) / NT;
    const long long grid = (long long)nTT * a.B * nNT;
    if (grid <= 0) return hipSuccess;
    if constexpr ((NW % 2) == 0) {
        if (phase2) return conv_bf16_launch_m<MT, NW, 2>(a, NCK, nTT, nNT, ROWS, lds, grid, s);
    }
    return deint ? conv_bf16_launch_m<MT, NW, 1>(a, NCK, nTT, nNT, ROWS, lds, grid, s)
                 : conv_bf16_launch_m<MT, NW, 0>(a, NCK, nTT, nNT, ROWS, lds, grid, s);
}

// Tile choices of the autotuner: code = (log2 MT) * 9 + (NW - 2) * 3 + (NCK - 1), MT in {1, 2, 4} tiles of 16
// positions per wave, NW in {2, 3, 4} column tiles, NCK in {1, 2, 3} channel chunks per stage.
bool conv_bf16_choice_ok(const ConvArgs& a, int variant) {
    const int code = variant - kBf16VariantBase;
    if (code < 0 || code >= 27 || !conv_bf16_supported(a)) return false;
    const int mt = 1 << (code / 9), nw = 2 + (code / 3) % 3, nck = 1 + code % 3;
    const bool phase2 = (a.flags & F_PHASE2) != 0;
    if (phase2 && (nw & 1)) return false;
    const int TT = 64 * mt, NT = 16 * nw;
    const int chan = phase2 ? NT / 2 : NT;
    const int padded = (a.N + chan - 1) / chan * chan;
    if (padded * 3 > a.N * 4 + 48) return false;                       // > ~33 % padded columns
    // the packed weight image is padded to a multiple of 64 columns: a tile whose last column group runs past that
    // would DMA the next channel group's rows (results land in discarded columns, but it is an out-of-image read)
    if (!phase2 && padded > (a.N + 63) / 64 * 64) return false;
    if (TT > 64 && TT >= 2 * a.Tout) return false;                     // mostly padding in time
    if (nck > (a.C0 + a.C1 + 31) / 32) return false;
    const int planes = a.loader == LOADER_DEINT ? 2 : 1;
    if (bf16_pairs64(planes * bf16_rows(a, TT)) * 4 * nck > bf16_xit(mt) * 256) return false;
    return bf16_lds(a, TT, NT, nck) <= 160 * 1024;
}
int conv_bf16_list_candidates(const ConvArgs& a, ConvChoice* out, int maxn) {
    int n = 0;
    for (int code = 0; code < 27 && n < maxn; ++code)
        if (conv_bf16_choice_ok(a, kBf16VariantBase + code)) { out[n].variant = kBf16VariantBase + code; out[n].ksplit = 1; ++n; }
    return n;
}

// a.W must point at the packed bf16 image of the layer's weights (pack_bf16_kernel), a.wb_c8p / a.wb_npad set
hipError_t launch_conv_bf16(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    if (!a.xbf || !conv_bf16_supported(a) || a.wb_c8p <= 0 || a.wb_npad <= 0 || !al16(a.W)) return hipErrorInvalidValue;
    if (!al16(a.src0) || (a.src1 != nullptr && !al16(a.src1))) return hipErrorInvalidValue;      // dword pairs at even row elements
    if (a.acc_len == 0) { a.acc_lo = 0; a.acc_len = 0x7FFFFFFFu; }                               // F_ACCUM over the whole row (default)
    bool vec = a.ostride == 1 && al16(a.dst0) && (a.obs0 & 3) == 0 && (a.opitch0 & 3) == 0 && (a.ooff0 & 3) == 0;
    if (a.dst1 != nullptr) vec = vec && al16(a.dst1) && (a.obs1 & 3) == 0 && (a.opitch1 & 3) == 0 && (a.ooff1 & 3) == 0;
    if (a.msk0 != nullptr) vec = vec && al16(a.msk0);
    if (a.msk1 != nullptr) vec = vec && al16(a.msk1);
    if (a.dec != nullptr) vec = vec && (a.decpitch & 1) == 0 && (a.decbs & 1) == 0 && al16(a.dec);
    if (vec) a.flags |= F_VEC4;
    // autotuned choice (ConvChoice.variant = kBf16VariantBase + tile code; checked by conv_bf16_choice_ok)
    if (a.force_variant > kBf16VariantBase) {
        const int code = a.force_variant - 1 - kBf16VariantBase;
        if (!conv_bf16_choice_ok(a, a.force_variant - 1)) return hipErrorInvalidValue;
        const int mt = 1 << (code / 9), nw = 2 + (code / 3) % 3, nck = 1 + code % 3;
#define WUN_BF(M, N) if (mt == M && nw == N) return conv_bf16_launch_t<M, N>(a, s, nck);
        WUN_BF(4, 4) WUN_BF(4, 3) WUN_BF(4, 2)
        WUN_BF(2, 4) WUN_BF(2, 3) WUN_BF(2, 2)
        WUN_BF(1, 4) WUN_BF(1, 3) WUN_BF(1, 2)
#undef WUN_BF
        return hipErrorInvalidValue;
    }
    // tile: fewest padded columns among 64/48/32; rows by how many tiles the launch has
    const bool phase2 = (a.flags & F_PHASE2) != 0;
    int bestnw = 4, bestpad = 1 << 30;
    const int cands[3] = {4, 3, 2};
    for (int i = 0; i < 3; ++i) {
        if (phase2 && (cands[i] & 1)) continue;                // both phases of a channel live in one workgroup
        const int ntile = phase2 ? cands[i] * 8 : cands[i] * 16;   // output channels per workgroup
        const int padded = ((a.N + ntile - 1) / ntile) * ntile;
        if (padded < bestpad) { bestpad = padded; bestnw = cands[i]; }
    }
    const int chan_per_wg = phase2 ? bestnw * 8 : bestnw * 16;
    const long long cols = (a.N + chan_per_wg - 1) / chan_per_wg;
    int mt = 4;
    while (mt > 1 && ((long long)((a.Tout + 64 * mt - 1) / (64 * mt)) * cols * a.B < 512 || a.Tout <= 32 * mt ||
                      bf16_lds(a, 64 * mt, bestnw * 16, 1) > 160 * 1024)) mt >>= 1;
    // A workgroup runs its phases (fetch, MFMA, store) back to back, so the CU needs a second resident workgroup to
    // overlap them: take the tallest tile whose LDS footprint still lets two workgroups share a CU
    static const int lds_cap = getenv("WUN_BF16_LDS_CAP") ? atoi(getenv("WUN_BF16_LDS_CAP")) : 80;
    {
        int m2 = mt;
        while (m2 > 1 && bf16_lds(a, 64 * m2, bestnw * 16, bf16_pick_nck(a, 64 * m2, bestnw * 16, bf16_xit(m2))) >
                             (size_t)lds_cap * 1024) m2 >>= 1;
        if (bf16_lds(a, 64 * m2, bestnw * 16, bf16_pick_nck(a, 64 * m2, bestnw * 16, bf16_xit(m2))) <= (size_t)lds_cap * 1024)
            mt = m2;
    }
#define WUN_BF(M, N) if (mt == M && bestnw == N) return conv_bf16_launch_t<M, N>(a, s);
    WUN_BF(4, 4) WUN_BF(4, 3) WUN_BF(4, 2)
    WUN_BF(2, 4) WUN_BF(2, 3) WUN_BF(2, 2)
    WUN_BF(1, 4) WUN_BF(1, 3) WUN_BF(1, 2)
#undef WUN_BF
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------
// The audio-input conv of the bf16 mode (UnetAudioSeparator.py:98 at i = 0: 1 or 2 input channels): fp32 audio in,
// bf16 activations out.  No dense channel x channel face (Cin <= 2), so it runs as a direct conv on the vector pipe,
// bound by writing the 24 .. 48 output rows: a thread owns 4 consecutive output positions of EVERY output channel -- its
// input window of 3 SI + K samples per channel sits in registers, the weights of one output channel are uniform
// (scalar loads), 4 K Cin FMAs per output channel, one 8-byte (bf16) / 16-byte (fp32) store per channel row: a wave
// writes 512 contiguous bytes per row.  Same epilogue as the MFMA convs: bias, LeakyReLU, optional compact copy of the
// even positions (the [:, ::2, :] of :100 in same-padding mode).  ConvArgs: src0 fp32 (xbf = 0), C1 = 0, W fp32 in TF
// layout [K][Cin][Cout], dst0 / dec bf16 (obf = 1) or fp32; loader DIRECT = stride 1, DEINT = stride 2.
// ---------------------------------------------------------------------------------------
// KT: the tap count when it is the shipped 15 (tap loop fully unrolled, no branches), 0 = run-time taps <= 15.  The
// weights sit in LDS as [cout][tap][cin] (one uniform-address read per tap: a broadcast), staged once per workgroup.
template <int SI, int CIN, bool OB, int KT>
__global__ __launch_bounds__(256) void first_conv_kernel(ConvArgs a) {
    using ET = std::conditional_t<OB, bf16_t, float>;
    __shared__ float wl[48 * WUN_BF_KMAX * 2 + 48];          // N <= 48 output channels per pass (launcher), + their bias
    constexpr int KM = KT > 0 ? KT : WUN_BF_KMAX;
    constexpr int XW = 3 * SI + KM;                          // window for 4 positions
    const int K = KT > 0 ? KT : a.KW;
    const int b = blockIdx.y;
    const int q = ((int)blockIdx.x * 256 + (int)threadIdx.x) * 4;
    const bool lrelu = (a.flags & F_LRELU) != 0;
    ET* const dst = reinterpret_cast<ET*>(a.dst0) + (long long)b * a.obs0 + a.ooff0;
    ET* const dec = a.dec != nullptr ? reinterpret_cast<ET*>(a.dec) + (long long)b * a.decbs : nullptr;
    float xv[CIN][XW];
    const int t0 = q * SI - a.shift;
    if (q < a.Tout) {
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
            const float* xr = a.src0 + (long long)b * a.bs0 + (long long)c * a.pitch0 + a.off0;
            if (t0 >= 0 && t0 + XW <= a.Tin) {               // interior: plain loads
#pragma unroll
                for (int i = 0; i < XW; ++i) xv[c][i] = xr[t0 + i];
            } else {
#pragma unroll
                for (int i = 0; i < XW; ++i) {
                    const int t = t0 + i;
                    const int tc = t < 0 ? 0 : (t > a.Tin - 1 ? a.Tin - 1 : t);
                    const float v = xr[tc];
                    xv[c][i] = (t >= 0 && t < a.Tin) ? v : 0.f;
                }
            }
        }
    }
    for (int nb = 0; nb < a.N; nb += 48) {
        const int nn = a.N - nb < 48 ? a.N - nb : 48;
        __syncthreads();
        for (int i = threadIdx.x; i < nn * K * CIN; i += 256) {
            const int n = i / (K * CIN), r = i - n * (K * CIN);          // r = tap * CIN + cin
            wl[n * (KM * CIN) + r] = a.W[(long long)r * a.N + nb + n];
        }
        for (int i = threadIdx.x; i < nn; i += 256) wl[48 * KM * CIN + i] = a.bias != nullptr ? a.bias[nb + i] : 0.f;
        __syncthreads();
        if (q >= a.Tout) continue;
        for (int n = 0; n < nn; ++n) {
            const float* wn = wl + n * (KM * CIN);
            float acc[4];
            const float bv = wl[48 * KM * CIN + n];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = bv;
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                if (KT > 0 || k < K) {
#pragma unroll
                    for (int c = 0; c < CIN; ++c) {
                        const float w = wn[k * CIN + c];
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r] = fmaf(xv[c][r * SI + k], w, acc[r]);
                    }
                }
            }
            if (lrelu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = fmaxf(0.2f * acc[r], acc[r]);
            }
            const long long row = (long long)(nb + n) * a.opitch0;
            if (q + 3 < a.Tout) {
                st4<ET>(dst, row + q, (f32x4){acc[0], acc[1], acc[2], acc[3]});
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (q + r < a.Tout) st1<ET>(dst, row + q + r, acc[r]);
            }
            if (dec != nullptr) {
                ET* dr = dec + (long long)(nb + n) * a.decpitch;
                st1<ET>(dr, q >> 1, acc[0]);
                if (q + 2 < a.Tout) st1<ET>(dr, (q >> 1) + 1, acc[2]);
            }
        }
    }
}

bool first_conv_supported(const ConvArgs& a) {
    return a.C1 == 0 && (a.C0 == 1 || a.C0 == 2) && a.KW >= 1 && a.KW <= WUN_BF_KMAX && !a.xbf && a.ostride == 1 &&
           a.dst1 == nullptr && a.msk0 == nullptr && (a.flags & (F_ACCUM | F_PHASE2)) == 0 && a.N0 == a.N && a.B <= 65535 &&
           (a.opitch0 & 3) == 0 && (a.obs0 & 3) == 0 && (a.ooff0 & 3) == 0 && al16(a.dst0);
}

hipError_t launch_first_conv(const ConvArgs& a, hipStream_t s) {
    if (!first_conv_supported(a)) return hipErrorInvalidValue;
    if (a.Tout <= 0) return hipSuccess;
    const dim3 grid((unsigned)((a.Tout + 1023) / 1024), (unsigned)a.B);
    char tag[128];
    snprintf(tag, sizeof(tag), "C=%d N=%d T=%d K=%d ld=%d B=%d", a.C0, a.N, a.Tout, a.KW, a.loader, a.B);
    // (bandwidth-bound: the output rows written once, the audio read once)
    prof_scope_begin("first_conv_kernel", conv_flops(a), s, tag,
                     (double)a.B * a.Tout * ((a.obf ? 2.0 : 4.0) * a.N * (a.dec != nullptr ? 1.5 : 1.0) + 4.0 * a.C0 * (a.loader == LOADER_DEINT ? 2 : 1)));
#define WUN_FC(SI, CI) \
    if (a.KW == 15) { \
        if (a.obf) hipLaunchKernelGGL((first_conv_kernel<SI, CI, true, 15>), grid, dim3(256), 0, s, a); \
        else hipLaunchKernelGGL((first_conv_kernel<SI, CI, false, 15>), grid, dim3(256), 0, s, a); \
    } else { \
        if (a.obf) hipLaunchKernelGGL((first_conv_kernel<SI, CI, true, 0>), grid, dim3(256), 0, s, a); \
        else hipLaunchKernelGGL((first_conv_kernel<SI, CI, false, 0>), grid, dim3(256), 0, s, a); \
    }
    if (a.loader == LOADER_DEINT) { if (a.C0 == 1) { WUN_FC(2, 1) } else { WUN_FC(2, 2) } }
    else { if (a.C0 == 1) { WUN_FC(1, 1) } else { WUN_FC(1, 2) } }
#undef WUN_FC
    prof_scope_end(s);
    return hipGetLastError();
}

// fp32 rows -> bf16 rows (nearest even), re-pitched: dst[r][t] = bf16(src[r][t]), t < T.  Used by the single-operator entry
// points (their callers hand over fp32 tensors) -- the plan never converts: its tensors are born bf16.
__global__ __launch_bounds__(256) void cast_rows_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int T,
                                                            long long spitch, long long dpitch) {
    const long long r = blockIdx.y;
    const int t = ((int)blockIdx.x * 256 + (int)threadIdx.x) * 4;
    if (t >= T) return;
    const float* sp = src + r * spitch + t;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = t + k < T ? sp[k] : 0.f;
    bf16_t* dp = dst + r * dpitch + t;
    if (t + 3 < dpitch && ((dpitch & 3) == 0)) {
        st4<bf16_t>(dp, 0, (f32x4){v[0], v[1], v[2], v[3]});
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (t + k < T) st1<bf16_t>(dp, k, v[k]);
    }
}

hipError_t launch_cast_rows_bf16(const float* src, void* dst, long long rows, int T, long long spitch, long long dpitch, hipStream_t s) {
    if (rows <= 0 || T <= 0) return hipSuccess;
    for (long long r0 = 0; r0 < rows; r0 += 65535) {
        const long long nr = rows - r0 < 65535 ? rows - r0 : 65535;
        hipLaunchKernelGGL(cast_rows_bf16_kernel, dim3((unsigned)((T + 1023) / 1024), (unsigned)nr), dim3(256), 0, s,
                           src + r0 * spitch, reinterpret_cast<bf16_t*>(dst) + r0 * dpitch, T, spitch, dpitch);
    }
    return hipGetLastError();
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
