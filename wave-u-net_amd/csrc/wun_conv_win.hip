// gfx950: "register-window" form of the exact-fp32 conv1d (UnetAudioSeparator.py:97-125 forward convs and the stride-1
// input gradients), the conv counterpart of wun_wgrad_win.hip:
//
//   out[b][n][q] = epi( sum_{k < K, c} W[k][c][n] * in[b][c][S q + k - shift] ),   K in {15, 5}, S in {1, 2}
//
// MFMA rows = output positions, columns = 16 output channels, k index = 4 input channels (lane group lg <-> channel).
// Four MFMA row tiles are INTERLEAVED in time -- row li of tile m is position q0 + 4 li + m -- so that a lane's A
// operands of all four tiles and all K taps are the 3 S + K consecutive floats in[c][S (q0 + 4 li) - shift ..] of ITS
// channel: a window it loads once per 4-channel chunk with aligned 16-byte LDS reads and then feeds to 4 K NW MFMAs
// straight from the registers (tile m, tap k -> window element S m + k).  The weights come in the "window layout"
// [c][tap group][n][4 taps] (written once per step by pack_win_kernel), so a B operand read is one 16-byte vector holding
// four taps of one (channel, output channel).  K = 15, three column tiles: 17 LDS reads per 180 MFMAs where
// conv_mfma_kernel issues 120.  Both operands of a chunk go global -> LDS by DMA in the scalar-base form (no address
// VALU, no LDS stores), issued from inside the MFMA stream of the previous chunk; only tiles that touch the zero
// padding of a transposed / 'same' conv clamp per lane and zero the out-of-range samples after landing.
// Epilogue: a lane's four tiles hold four CONSECUTIVE positions -> 16-byte stores; bias, LeakyReLU, LeakyReLU-derivative
// mask, (windowed) accumulate, two destinations -- the semantics of conv_mfma_kernel's vector epilogue.
// Not served here (launch_conv keeps conv_mfma_kernel): split-K, the fused two-phase transposed conv, batch-folded deep
// levels, strided outputs, the decimated copy of 'same' padding, channel counts that are not multiples of 4.
#include "wun_internal.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace wun {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

__device__ __forceinline__ unsigned cw_sgpr(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ const float* cw_sgpr_ptr(const float* q) {
    const unsigned long long v = (unsigned long long)q;
    const unsigned lo = cw_sgpr((unsigned)v), hi = cw_sgpr((unsigned)(v >> 32));
    return (const float*)(((unsigned long long)hi << 32) | lo);
}
// one LDS-DMA instruction, scalar-base form: 64 lanes x 16 bytes from sbase + voff[lane] to LDS bytes m0 + 16 lane
__device__ __forceinline__ void cw_dma16(unsigned m0v, unsigned voff, const float* sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ int cw_xcd_block(int bid, int grid) {
    const int per = grid >> 3, rem = grid & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
}

#define WUN_CW_XIT 5      // DMA instructions per input row (row pitch <= 768 floats)
#define WUN_CW_WIT 6      // DMA instructions per weight channel (tap groups x NT <= 384 slots)

#ifdef WUN_CW_TRACE
// diagnostic builds only (tools/cw_trace.py): per workgroup and wave {100 MHz clock at entry / exit} + per chunk
// {loop top, data landed (barrier passed), MFMAs done} shader-clock stamps, then epilogue end
#define WUN_CWT_WGS 2048
#define WUN_CWT_CH 12
#define WUN_CWT_WORDS (6 + 3 * WUN_CWT_CH)
__device__ unsigned long long g_cw_trace[WUN_CWT_WGS * 4 * WUN_CWT_WORDS];
#endif

struct ConvWinParams {
    int WT, WN;          // waves along time / along output channels (WT * WN == 4)
    int TT, NT;          // workgroup tile: positions x output channels
    int nTT, nNT;
    int XP;              // LDS row pitch of the input rows (floats, multiple of 256: whole 1 KiB DMA blocks, conflict-free reads)
    int XGL;             // live 16-byte granules per input row
    int WR;              // 16-byte slots per weight channel in LDS (tap groups x NT, padded to whole DMA blocks)
    int woff;            // float offset of the weight region in a buffer
    int bufFloats;       // floats per LDS buffer {4 CQ input rows, weights [4 CQ][KG][NT][4]}
    unsigned nt_inv;     // ceil(65536 / NT): slot -> (tap group, column) without a division
};

// CQ = channel quads (MFMA k-steps worth of channels) per chunk, i.e. per barrier
template <int K, int S, int MB, int NW, int CQ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MB * NW <= 3 ? 3 : 2)))
void conv_win_kernel(ConvArgs a, ConvWinParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int KG = (K + 3) / 4;                     // tap groups of 4
    constexpr int NWIN = (3 * S + K + 3) / 4;           // 16-byte reads per window
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wt = wave % p.WT, wn = wave / p.WT;

    int bid = cw_xcd_block((int)blockIdx.x, (int)gridDim.x);
    const int nt = bid % p.nNT; bid /= p.nNT;
    const int tt = bid % p.nTT;
    const int b = bid / p.nTT;
    const int q0 = tt * p.TT, n0 = nt * p.NT;
    const int wq0 = q0 + wt * MB * 64, wn0 = n0 + wn * NW * 16;
    const int Ctot = a.C0 + a.C1;
    const int nchunks = Ctot / (4 * CQ);
    const int t0 = S * q0 - a.shift;                    // time index of the first staged input sample

    // ---- staging map, no per-thread division: wave w stages input rows / weight channels w, w + 4, ...; lane l of DMA
    // instruction i covers slot 64 i + l of its row / channel.  Chunk-invariant byte offsets from uniform base pointers:
    const int XG = p.XP >> 2;                           // slots per input row (multiple of 64)
    const int nxi = XG >> 6, nwi = p.WR >> 6;           // DMA instructions per row / per channel
    unsigned xb[WUN_CW_XIT], wb[WUN_CW_WIT];
#pragma unroll
    for (int i = 0; i < WUN_CW_XIT; ++i) {
        const int g = 64 * i + lane;
        xb[i] = 16u * (unsigned)(g < p.XGL ? g : p.XGL - 1);
    }
#pragma unroll
    for (int i = 0; i < WUN_CW_WIT; ++i) {
        int sl = 64 * i + lane;
        sl = sl < KG * p.NT ? sl : KG * p.NT - 1;        // pad slots re-fetch the last one
        const int kg = (int)(((unsigned)sl * p.nt_inv) >> 16);
        int ng = n0 + sl - kg * p.NT;
        ng = ng < a.N ? ng : a.N - 1;                    // padded columns: any valid weights (results never stored)
        wb[i] = 16u * (unsigned)(kg * a.N + ng);         // window layout [c][kg][n][4]
    }
    const unsigned lds_base = (unsigned)(unsigned long long)(lds_void_t*)lds;
    const bool edge = t0 < 0 || t0 + 4 * p.XGL > a.Tin;           // tile touches the zero padding (or the row end)

    // source rows of chunk `ch` (channels 4 CQ ch .., one source)
    auto chunk_src = [&](int ch, int& pitch, int& off) __attribute__((always_inline)) -> const float* {
        const int c0 = 4 * CQ * ch;
        if (c0 < a.C0) { pitch = a.pitch0; off = a.off0; return a.src0 + (long long)b * a.bs0 + (long long)c0 * a.pitch0; }
        pitch = a.pitch1; off = a.off1;
        return a.src1 + (long long)b * a.bs1 + (long long)(c0 - a.C0) * a.pitch1;
    };
    auto dma_chunk = [&](int ch, int bo) __attribute__((always_inline)) {
        int pitch, off;
        const float* rows = chunk_src(ch, pitch, off);
        const unsigned m0b = cw_sgpr(lds_base + 4u * (unsigned)bo);
#pragma unroll
        for (int r = 0; r < CQ; ++r) {
            const int row = wave + 4 * r;
            const unsigned m0x = m0b + 16u * (unsigned)(row * XG);
            if (!edge) {
                const float* xbs = cw_sgpr_ptr(rows + (long long)row * pitch + off + t0);
#pragma unroll
                for (int i = 0; i < WUN_CW_XIT; ++i)
                    if (i < nxi) cw_dma16(m0x + 1024u * (unsigned)i, xb[i], xbs);
            } else {
                // clamp every 16 bytes into its source row; zero_fix() repairs the samples outside [0, Tin)
#pragma unroll
                for (int i = 0; i < WUN_CW_XIT; ++i)
                    if (i < nxi) {
                        int er = (int)(xb[i] >> 2) + off + t0;
                        er = er < 0 ? 0 : (er > pitch - 4 ? pitch - 4 : er);
                        __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(rows + (long long)row * pitch + er),
                                                         (lds_void_t*)(lds + bo + 4 * (row * XG + 64 * i)), 16, 0, 0);
                    }
            }
            const float* wbs = cw_sgpr_ptr(a.Wwin + ((long long)(4 * CQ * ch + row) * KG) * a.N * 4);
            const unsigned m0v = m0b + 4u * (unsigned)p.woff + 16u * (unsigned)(row * p.WR);
#pragma unroll
            for (int i = 0; i < WUN_CW_WIT; ++i)
                if (i < nwi) cw_dma16(m0v + 1024u * (unsigned)i, wb[i], wbs);
        }
    };
    auto zero_fix = [&](int ch, int bo) __attribute__((always_inline)) {
        int pitch, off;
        (void)chunk_src(ch, pitch, off);
#pragma unroll
        for (int r = 0; r < CQ; ++r)
#pragma unroll
            for (int i = 0; i < WUN_CW_XIT; ++i) {
                int g = 64 * i + lane;
                asm volatile("" : "+v"(g));
                if (i < nxi && g < p.XGL) {
                    float* q = lds + bo + 4 * ((wave + 4 * r) * XG + g);
                    const int er0 = off + t0 + 4 * g;
                    const int erc = er0 < 0 ? 0 : (er0 > pitch - 4 ? pitch - 4 : er0);
                    f32x4 v = *reinterpret_cast<f32x4*>(q);
                    f32x4 w;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int t = t0 + 4 * g + k;
                        const int j = er0 + k - erc;
                        float x = 0.f;
                        if (t >= 0 && t < a.Tin && j >= 0 && j < 4) x = j == 0 ? v[0] : j == 1 ? v[1] : j == 2 ? v[2] : v[3];
                        w[k] = x;
                    }
                    *reinterpret_cast<f32x4*>(q) = w;
                }
            }
    };

    f32x4 acc[MB][4][NW];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < NW; ++n) acc[mb][m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // lane's LDS read positions: input row lg (+ 4 per quad), window at float S (wave offset + 4 li); weights of channel
    // lg, column li
    const int xrd = lg * p.XP + S * (wt * MB * 64 + 4 * li);
    const int wrd = p.woff + 4 * (lg * p.WR + wn * NW * 16 + li);

    auto mfma_chunk = [&](int bo, auto&& mid) __attribute__((always_inline)) {
#pragma unroll
        for (int qd = 0; qd < CQ; ++qd) {
            const float* xp = lds + bo + xrd + 4 * qd * p.XP;
            const float* wp = lds + bo + wrd + 16 * qd * p.WR;
            f32x4 win[MB][NWIN];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int j = 0; j < NWIN; ++j) win[mb][j] = *reinterpret_cast<const f32x4*>(xp + S * 64 * mb + 4 * j);
            f32x4 bv[2][NW];
#pragma unroll
            for (int n = 0; n < NW; ++n) bv[0][n] = *reinterpret_cast<const f32x4*>(wp + 64 * n);
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) {
                if (kg + 1 < KG) {
#pragma unroll
                    for (int n = 0; n < NW; ++n) bv[(kg + 1) & 1][n] = *reinterpret_cast<const f32x4*>(wp + 4 * (kg + 1) * p.NT + 64 * n);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 4 * kg + j;
                    if (k >= K) continue;
                    if (qd == 0 && kg == 0 && j == 1) mid();     // the next chunk's DMA: issued from inside the MFMA stream
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            const int e = S * m + k;
                            const float av = win[mb][e >> 2][e & 3];
#pragma unroll
                            for (int n = 0; n < NW; ++n)
                                acc[mb][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[kg & 1][n][j], acc[mb][m][n], 0, 0, 0);
                        }
                }
            }
        }
    };

#ifdef WUN_CW_TRACE
    const bool tr_on = lane == 0 && blockIdx.x < WUN_CWT_WGS;
    unsigned long long* trp = g_cw_trace + ((size_t)(blockIdx.x < WUN_CWT_WGS ? blockIdx.x : 0) * 4 + wave) * WUN_CWT_WORDS;
    if (tr_on) { trp[0] = wall_clock64(); trp[1] = __builtin_readcyclecounter(); trp[3] = (unsigned long long)nchunks; }
#define CWT_STAMP(k, i) do { if (tr_on && (k) < WUN_CWT_CH) trp[6 + 3 * (k) + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define CWT_STAMP(k, i) do { } while (0)
#endif
    if (nchunks > 0) dma_chunk(0, 0);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int cur = (ch & 1) * p.bufFloats;
        CWT_STAMP(ch, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        CWT_STAMP(ch, 1);
        if (edge) {
            zero_fix(ch, cur);
            __syncthreads();
        }
        const bool more = ch + 1 < nchunks;
        if (more && edge) dma_chunk(ch + 1, p.bufFloats - cur);
        mfma_chunk(cur, [&]() __attribute__((always_inline)) { if (more && !edge) dma_chunk(ch + 1, p.bufFloats - cur); });
        CWT_STAMP(ch, 2);
    }
#ifdef WUN_CW_TRACE
    if (tr_on) trp[4] = __builtin_readcyclecounter();
#endif

    // ---- epilogue: lane (li, lg) holds, for column wn0 + 16 n + li, the positions wq0 + 64 mb + 16 lg + 4 r + {0..3}.
    // FULL = the whole workgroup tile lies inside [0, Tout) x [0, N): no bounds tests (all but the last tiles) ----
    const bool lrelu = (a.flags & F_LRELU) != 0;
    const bool accum = (a.flags & F_ACCUM) != 0;
    auto epilogue = [&](auto full_c) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            const int ncol = wn0 + n * 16 + li;
            if (!FULL && ncol >= a.N) continue;
            const float bvv = (a.bias != nullptr) ? a.bias[ncol] : 0.f;
            float* dst; const float* msk; int ooff;
            if (ncol < a.N0) {
                dst = a.dst0 + (long long)b * a.obs0 + (long long)ncol * a.opitch0 + a.ooff0;
                msk = a.msk0 != nullptr ? a.msk0 + (long long)b * a.obs0 + (long long)ncol * a.opitch0 + a.ooff0 : nullptr;
                ooff = a.ooff0;
            } else {
                dst = a.dst1 + (long long)b * a.obs1 + (long long)(ncol - a.N0) * a.opitch1 + a.ooff1;
                msk = a.msk1 != nullptr ? a.msk1 + (long long)b * a.obs1 + (long long)(ncol - a.N0) * a.opitch1 + a.ooff1 : nullptr;
                ooff = a.ooff1;
            }
            const int qb = wq0 + 16 * lg;
            dst += qb;
            if (msk != nullptr) msk += qb;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int dq = 64 * mb + 4 * r;                  // compile-time offset from qb
                    const int q = qb + dq;
                    if (!FULL && q >= a.Tout) continue;
                    f32x4 v = {acc[mb][0][n][r], acc[mb][1][n][r], acc[mb][2][n][r], acc[mb][3][n][r]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] += bvv;
                        if (lrelu) v[e] = fmaxf(0.2f * v[e], v[e]);
                    }
                    if (FULL || q + 3 < a.Tout) {
                        if (msk != nullptr) {
                            const f32x4 mk = *reinterpret_cast<const f32x4*>(msk + dq);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= (mk[e] > 0.f) ? 1.f : 0.2f;
                        }
                        const int pos0 = ooff + q;
                        if (accum && (conv_acc_at(a, pos0) || conv_acc_at(a, pos0 + 3))) {
                            const f32x4 old = *reinterpret_cast<const f32x4*>(dst + dq);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (conv_acc_at(a, pos0 + e)) v[e] += old[e];
                        }
                        *reinterpret_cast<f32x4*>(dst + dq) = v;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (q + e < a.Tout) {
                                float x = v[e];
                                if (msk != nullptr) x *= (msk[dq + e] > 0.f) ? 1.f : 0.2f;
                                if (accum && conv_acc_at(a, ooff + q + e)) x += dst[dq + e];
                                dst[dq + e] = x;
                            }
                        }
                    }
                }
        }
    };
    if (q0 + p.TT <= a.Tout && n0 + p.NT <= a.N) epilogue(std::true_type{});
    else epilogue(std::false_type{});
#ifdef WUN_CW_TRACE
    if (tr_on) { trp[5] = __builtin_readcyclecounter(); trp[2] = wall_clock64(); }
#endif
}

// [K][C][N] weights -> window layout [C][KG][N][4]: dst[((c KG + kg) N + n) 4 + j] = src[(4 kg + j)][c][n] (0 for taps >= K).
// One thread = one 16-byte output vector; the four taps are four reads, each coalesced along n.
__global__ __launch_bounds__(256) void pack_win_kernel(const float* __restrict__ params, float* __restrict__ ws,
                                                       const WinPackDesc* __restrict__ descs) {
    const WinPackDesc d = descs[blockIdx.y];
    const float* src = (d.src_in_ws ? ws : params) + d.src_off;
    f32x4* dst = reinterpret_cast<f32x4*>(ws + d.dst_off);
    const int KG = (d.K + 3) / 4;
    const long long total = (long long)d.C * KG * d.N;
    const long long cn = (long long)d.C * d.N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i % d.N);
        const long long r = i / d.N;
        const int kg = (int)(r % KG), c = (int)(r / KG);
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * kg + j;
            v[j] = k < d.K ? src[(long long)k * cn + (long long)c * d.N + n] : 0.f;
        }
        dst[i] = v;
    }
}

hipError_t launch_pack_win(const float* params, float* ws, const WinPackDesc* dev_descs, int ndesc, long long max_vecs,
                           hipStream_t s) {
    if (ndesc <= 0) return hipSuccess;
    long long bx = (max_vecs + 255) / 256;
    if (bx > 64) bx = 64;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(pack_win_kernel, dim3((unsigned)bx, (unsigned)ndesc), dim3(256), 0, s, params, ws, dev_descs);
    return hipGetLastError();
}

// ---- tile menu: (MB, NW, WT, WN): positions per workgroup = WT * MB * 64, output channels = WN * NW * 16 ----
struct ConvWinVariant { int MB, NW, WT, WN, CQ; };
static const ConvWinVariant kWinVariants[] = {
    {1, 3, 4, 1, 1},      // 256 x 48, 4 channels per barrier
    {1, 2, 4, 1, 1},      // 256 x 32
    {1, 4, 4, 1, 1},      // 256 x 64
    {1, 5, 4, 1, 1},      // 256 x 80
    {2, 3, 4, 1, 1},      // 512 x 48
    {1, 3, 2, 2, 1},      // 128 x 96
    {2, 3, 2, 2, 1},      // 256 x 96
    {2, 2, 4, 1, 1},      // 512 x 32
    {1, 3, 4, 1, 2},      // 256 x 48, 8 channels per barrier
    {2, 3, 4, 1, 2},      // 512 x 48
    {1, 2, 4, 1, 2},      // 256 x 32
    {2, 2, 4, 1, 2},      // 512 x 32
    {1, 3, 2, 2, 2},      // 128 x 96
    {2, 3, 2, 2, 2},      // 256 x 96
};
int conv_win_num_variants() { return (int)(sizeof(kWinVariants) / sizeof(kWinVariants[0])); }

static bool cw_aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

static bool conv_win_geom(const ConvArgs& a, int wv, ConvWinParams& p, size_t& lds) {
    if (wv < 0 || wv >= conv_win_num_variants()) return false;
    const ConvWinVariant& v = kWinVariants[wv];
    const int S = a.loader == LOADER_DEINT ? 2 : 1;
    const int K = a.KW, KG = (K + 3) / 4;
    const int NWIN = (3 * S + K + 3) / 4;
    const int CQ = v.CQ;
    p.WT = v.WT; p.WN = v.WN;
    p.TT = v.WT * v.MB * 64; p.NT = v.WN * v.NW * 16;
    p.nTT = (a.Tout + p.TT - 1) / p.TT; p.nNT = (a.N + p.NT - 1) / p.NT;
    const int need = std::max(S * (p.TT - 1) + K, S * (p.TT - 4) + 4 * NWIN);
    p.XGL = (need + 3) / 4;
    p.XP = ((4 * p.XGL + 255) / 256) * 256;
    p.WR = (KG * p.NT + 63) & ~63;
    p.woff = 4 * CQ * p.XP;
    p.bufFloats = p.woff + 4 * (4 * CQ * p.WR);
    p.nt_inv = (65536u + (unsigned)p.NT - 1) / (unsigned)p.NT;
    lds = sizeof(float) * 2 * (size_t)p.bufFloats;
    if ((p.XP >> 8) > WUN_CW_XIT || (p.WR >> 6) > WUN_CW_WIT) return false;
    return lds <= 96 * 1024;
}

// may this launch run window variant `wv`?  (the one rule set of dispatcher, tuner, hooks and imported tables)
bool conv_win_ok(const ConvArgs& a, int wv) {
    static const bool off = getenv("WUN_NO_CONV_WIN") != nullptr;
    if (off || a.Wwin == nullptr) return false;
    if (!(a.KW == 15 || a.KW == 5)) return false;
    const int Ctot = a.C0 + a.C1;
    if (wv < 0 || wv >= conv_win_num_variants()) return false;
    const int cq4 = 4 * kWinVariants[wv].CQ;
    if (Ctot < 8 || (Ctot % cq4) != 0 || (a.C0 % cq4) != 0 || (a.N & 3) != 0 || a.N < 16) return false;
    if (a.flags & F_PHASE2) return false;
    if (a.loader == LOADER_DEINT && a.KW != 15) return false;
    if (a.ostride != 1 || a.dec != nullptr || a.ups_y != nullptr || a.ubw_dz != nullptr) return false;
    // 16-byte vector epilogue: aligned destinations / masks
    bool vec = cw_aligned16(a.dst0) && (a.obs0 & 3) == 0 && (a.opitch0 & 3) == 0 && (a.ooff0 & 3) == 0;
    if (a.dst1 != nullptr) vec = vec && cw_aligned16(a.dst1) && (a.obs1 & 3) == 0 && (a.opitch1 & 3) == 0 && (a.ooff1 & 3) == 0;
    if (a.msk0 != nullptr) vec = vec && cw_aligned16(a.msk0);
    if (a.msk1 != nullptr) vec = vec && cw_aligned16(a.msk1);
    if (!vec || !cw_aligned16(a.Wwin)) return false;
    if ((long long)4 * ((a.KW + 3) / 4) * a.N * 16 >= (1ll << 31)) return false;
    ConvWinParams p;
    size_t lds;
    if (!conv_win_geom(a, wv, p, lds)) return false;
    if (p.TT >= 2 * a.Tout && p.TT > 64) return false;                                       // mostly padding in time
    const int padded = p.nNT * p.NT;
    if (padded * 3 > a.N * 4 + 48) return false;                                             // > ~33 % padded columns
    return true;
}

size_t conv_win_lds_bytes(const ConvArgs& a, int wv) {
    ConvWinParams p;
    size_t lds = 0;
    (void)conv_win_geom(a, wv, p, lds);
    return lds;
}

// heuristic choice among the window variants (or -1): fewest padded columns, then the 256-position tiles
int conv_win_pick(const ConvArgs& a) {
    int best = -1;
    long long bestcost = 1ll << 60;
    for (int wv = 0; wv < conv_win_num_variants(); ++wv) {
        if (!conv_win_ok(a, wv)) continue;
        ConvWinParams p;
        size_t lds;
        conv_win_geom(a, wv, p, lds);
        const long long cost = (long long)p.nTT * p.TT * p.nNT * p.NT * 16 + wv;            // padded work, ties to the menu order
        if (cost < bestcost) { bestcost = cost; best = wv; }
    }
    return best;
}

template <int K, int S, int MB, int NW, int CQ>
static hipError_t conv_win_launch_t(const ConvArgs& a, const ConvWinParams& p, size_t lds, hipStream_t s) {
    auto kern = conv_win_kernel<K, S, MB, NW, CQ>;
    static size_t lds_allowed = 64 * 1024;
    if (lds > lds_allowed) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_allowed = lds;
    }
    const long long grid = (long long)p.nTT * p.nNT * a.B;
    if (grid <= 0) return hipSuccess;
    char nm[64];
    snprintf(nm, sizeof(nm), "conv_win_kernel<%d, %d, %d, %d, %d>", K, S, MB, NW, CQ);
    char tag[160];
    snprintf(tag, sizeof(tag), "C=%d N=%d T=%d K=%d ld=%d B=%d tile=%dx%d acc=%d grid=%lld", a.C0 + a.C1, a.N, a.Tout, a.KW,
             a.loader, a.B, p.TT, p.NT, (a.flags & F_ACCUM) ? 1 : 0, grid);
    prof_scope_begin(nm, conv_flops(a), s, tag);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, a, p);
    prof_scope_end(s);
    return hipGetLastError();
}

hipError_t launch_conv_win(const ConvArgs& a_in, int wv, hipStream_t s) {
    if (!conv_win_ok(a_in, wv)) return hipErrorInvalidValue;
    ConvArgs a = a_in;
    if (a.acc_len == 0) { a.acc_lo = 0; a.acc_len = 0x7FFFFFFFu; }
    ConvWinParams p;
    size_t lds;
    conv_win_geom(a, wv, p, lds);
    const ConvWinVariant& v = kWinVariants[wv];
    const int S = a.loader == LOADER_DEINT ? 2 : 1;
#define WUN_CWL(k, ss, mb, nw, cq) if (a.KW == k && S == ss && v.MB == mb && v.NW == nw && v.CQ == cq) return conv_win_launch_t<k, ss, mb, nw, cq>(a, p, lds, s);
#define WUN_CWL_ALL(k, ss) WUN_CWL(k, ss, 1, 2, 1) WUN_CWL(k, ss, 1, 3, 1) WUN_CWL(k, ss, 1, 4, 1) WUN_CWL(k, ss, 1, 5, 1) WUN_CWL(k, ss, 2, 2, 1) WUN_CWL(k, ss, 2, 3, 1) \
                           WUN_CWL(k, ss, 1, 2, 2) WUN_CWL(k, ss, 1, 3, 2) WUN_CWL(k, ss, 2, 2, 2) WUN_CWL(k, ss, 2, 3, 2)
    WUN_CWL_ALL(15, 1) WUN_CWL_ALL(15, 2) WUN_CWL_ALL(5, 1)
#undef WUN_CWL_ALL
#undef WUN_CWL
    return hipErrorInvalidValue;
}

}  // namespace wun

#ifdef WUN_CW_TRACE
extern "C" int wun_dbg_cw_trace_read(unsigned long long* host, int nwords) {
    const int cap = (int)(sizeof(wun::g_cw_trace) / sizeof(unsigned long long));
    if (nwords > cap) nwords = cap;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(wun::g_cw_trace), (size_t)nwords * sizeof(unsigned long long)) != hipSuccess) return -2;
    return nwords;
}
#endif
