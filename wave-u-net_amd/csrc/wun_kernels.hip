// gfx950 (MI355X / CDNA4) kernels of the Wave-U-Net hot path.
//
// The two heavy kernels are implicit GEMMs on the exact-fp32 matrix instruction
// v_mfma_f32_16x16x4_f32 (same 157 TFLOP/s peak as the fp32 vector pipe, bit-equal to an
// fmaf chain, no reduced precision):
//
//   conv_mfma_kernel   D[time 16][cout 16] += A[time][k] * B[k][cout],  k = (tap, cin)
//        A comes from an LDS tile of the NCW input (time contiguous, halo included,
//        zero filled outside the valid range -> crop / 'same' padding / concat are
//        address arithmetic), B from an LDS tile of the TF-layout kernel [K][Cin][Cout].
//        Serves the forward convs (stride 1, and stride 2 through a de-interleaving
//        loader that fuses the [:, ::2, :] decimation into the conv so the never-observed
//        odd outputs are not computed) and, with tap-flipped/transposed weights, every
//        input-gradient.
//   wgrad_mfma_kernel  D[(cin,tap) 16][cout 16] += A[(cin,tap)][t] * B[t][cout]
//        reduction over batch*time, split over workgroups, deterministic two-stage sum;
//        an extra all-ones A row yields the bias gradient in the same pass.
//
// Wavefront = 64 lanes; one 16x16x4 MFMA takes A[i = lane&15][k = lane>>4],
// B[k = lane>>4][j = lane&15] and returns D[i = 4*(lane>>4)+r][j = lane&15], r = 0..3.
#include "wun_internal.h"

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace wun {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__host__ __device__ static inline int fit_pitch(int width, int mod32) {
    return width + (((mod32 - (width % 32)) % 32) + 32) % 32;
}

// =====================================================================================
// optional per-launch timing (HIP events on the launch stream), used by bench.py
// =====================================================================================
struct ProfSlot { std::string name; double flops; hipEvent_t e0, e1; };
static std::vector<ProfSlot> g_prof;
static bool g_prof_on = false;
static std::mutex g_prof_mu;

void prof_begin() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& sl : g_prof) { (void)hipEventDestroy(sl.e0); (void)hipEventDestroy(sl.e1); }
    g_prof.clear();
    g_prof_on = true;
}

struct ProfScope {
    bool on; hipStream_t s; size_t idx;
    ProfScope(const char* name, double flops, hipStream_t st) : on(g_prof_on), s(st), idx(0) {
        if (!on) return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        ProfSlot sl; sl.name = name; sl.flops = flops;
        if (hipEventCreate(&sl.e0) != hipSuccess || hipEventCreate(&sl.e1) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(sl.e0, s);
        g_prof.push_back(sl);
        idx = g_prof.size() - 1;
    }
    ~ProfScope() {
        if (!on) return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        (void)hipEventRecord(g_prof[idx].e1, s);
    }
};

// JSON: {"kernels": [{"name":..., "launches": n, "ms": total, "flops": total}, ...]}
std::string prof_end() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = false;
    struct Agg { long n = 0; double ms = 0, flops = 0; };
    std::map<std::string, Agg> agg;
    for (auto& sl : g_prof) {
        (void)hipEventSynchronize(sl.e1);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sl.e0, sl.e1) == hipSuccess) {
            Agg& a = agg[sl.name]; a.n += 1; a.ms += ms; a.flops += sl.flops;
        }
        (void)hipEventDestroy(sl.e0); (void)hipEventDestroy(sl.e1);
    }
    g_prof.clear();
    std::string out = "{\"kernels\": [";
    bool first = true;
    for (auto& kv : agg) {
        char buf[512];
        snprintf(buf, sizeof(buf), "%s{\"name\": \"%s\", \"launches\": %ld, \"ms\": %.6f, \"flops\": %.6e}",
                 first ? "" : ", ", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops);
        out += buf;
        first = false;
    }
    out += "]}";
    return out;
}

// =====================================================================================
// implicit-GEMM conv
// =====================================================================================
// Workgroup = 4 waves; tile = (WT*MT*16 time) x (WN*NW*16 cout).  Per channel chunk (CK LDS
// rows) the input window + halo and the weight slab [J][CK][NT] are staged through
// REGISTERS one chunk ahead (global loads of chunk c+1 are in flight while the MFMAs of
// chunk c run; they are written to LDS after the barrier), weights as 16-byte loads.
// Low-parallelism launches (deep levels: few time tiles) are split over channel chunks
// (split-K): raw partial tiles go to a scratch buffer and conv_splitk_epilogue_kernel sums
// them in a fixed order and applies the epilogue.
#define WUN_JMAX 15

template <int MT, int NW, int WT, int WN, int CK, bool VECW>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a, int nTT, int nNT, int J,
                                                        int XP, int WP) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TT = WT * MT * 16;
    constexpr int NT = WN * NW * 16;
    constexpr int NT4 = NT / 4;
    constexpr int CH = CK / 2;
    // staging trip counts (compile time, sized for J <= WUN_JMAX)
    constexpr int TPR = 256 / CK;                                   // DIRECT: threads per LDS row
    constexpr int XIT_D = (TT + WUN_JMAX - 1 + TPR - 1) / TPR;
    constexpr int TPC = 256 / CH;                                   // DEINT: threads per channel
    constexpr int XIT_I = (2 * (TT + (WUN_JMAX + 1) / 2 - 1) + TPC - 1) / TPC;
    constexpr int XIT = XIT_D > XIT_I ? XIT_D : XIT_I;
    constexpr int WIT = (WUN_JMAX * CK * NT4 + 255) / 256;

    float* Xs = lds;
    float* Ws = lds + CK * XP;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    const int nt = bid % nNT; bid /= nNT;
    const int tt = bid % nTT; bid /= nTT;
    const int b = bid % a.B;
    const int ksp = bid / a.B;
    const int q0 = tt * TT, n0 = nt * NT;
    const int wt0 = (wave % WT) * MT * 16;
    const int wn0 = (wave / WT) * NW * 16;
    const int Ctot = a.C0 + a.C1;
    const bool deint = (a.loader == LOADER_DEINT);
    const int UW = TT + J - 1;
    const int CKC = deint ? CH : CK;                 // input channels per chunk
    const int nchunks = (Ctot + CKC - 1) / CKC;
    const int ch_lo = ksp * a.cps;
    int ch_hi = ch_lo + a.cps;
    if (ch_hi > nchunks) ch_hi = nchunks;

    const float* src0b = a.src0 + (long long)b * a.bs0 + a.off0;
    const float* src1b = (a.src1 != nullptr) ? a.src1 + (long long)b * a.bs1 + a.off1 : src0b;

    f32x4 acc[MT][NW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float xreg[XIT];
    f32x4 wreg[WIT];          // VECW: one 16-byte load each; else 4 scalar loads each
    unsigned xmask = 0, wmask = 0;   // validity bits of the chunk held in xreg / wreg

    // ---- chunk-invariant staging state, computed once: clamped time offsets + validity of the
    // input window elements this thread stages, element offsets + validity of its weight
    // vectors.  Per chunk only a uniform base (c0) changes. ----
    int xt[XIT];
    unsigned xmask_s = 0, wmask_s = 0;
    const int xrow = deint ? tid / TPC : tid / TPR;       // LDS row (DIRECT) / channel (DEINT) staged by this thread
    {
        const int lr = deint ? tid % TPC : tid % TPR;
        const int stride_i = deint ? TPC : TPR;
        const int lim = deint ? 2 * UW : UW;
        const int tbase = (deint ? 2 * q0 : q0) - a.shift;
#pragma unroll
        for (int i = 0; i < XIT; ++i) {
            const int u = lr + i * stride_i;
            const int t = tbase + u;
            const bool ok = u < lim && t >= 0 && t < a.Tin;
            int tc = t < 0 ? 0 : t;
            if (tc > a.Tin - 1) tc = a.Tin - 1;
            xt[i] = tc;
            xmask_s |= (ok ? 1u : 0u) << i;
        }
    }
    int wofs[WIT];            // element offset of this thread's i-th weight vector for channel chunk 0
    int wr[WIT];              // channel index within the chunk (for the channel-tail mask)
    const int nwvec = J * CK * NT4;                       // weight vectors per chunk (uniform)
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
        const int f = tid + i * 256;
        const int row = f / NT4, c4 = f % NT4;
        const int j = row / CK, r = row % CK;
        int k, cch;
        if (!deint) { k = j; cch = r; }
        else { k = 2 * j + r / CH; cch = r % CH; }
        const bool ok = f < nwvec && k < a.KW && n0 + c4 * 4 < a.N;
        wofs[i] = ok ? (k * Ctot + cch) * a.N + n0 + c4 * 4 : 0;
        wr[i] = cch;
        wmask_s |= (ok ? 1u : 0u) << i;
    }

    // ---- global -> registers for one chunk.  Loads are unconditional (clamped addresses);
    // the zero fill is applied when the registers are written to LDS, so nothing here
    // waits for the loads and they stay in flight during the MFMAs of the previous chunk ----
    auto load_chunk = [&](int chunk) {
        const int c0 = chunk * CKC;
        {
            const int c = c0 + xrow;
            const bool cok = c < Ctot;
            const float* p = (!cok || c < a.C0) ? src0b + (long long)(cok ? c : 0) * a.pitch0
                                                : src1b + (long long)(c - a.C0) * a.pitch1;
            const int nx = deint ? XIT_I : XIT_D;
#pragma unroll
            for (int i = 0; i < XIT; ++i)
                if (i < nx) xreg[i] = p[xt[i]];
            xmask = cok ? xmask_s : 0u;
        }
        const bool tail = c0 + CKC > Ctot;                // uniform; only the last chunk of odd configs
        const float* wc = a.W + (long long)c0 * a.N;      // uniform base of this chunk's weight rows
        wmask = wmask_s;
#pragma unroll
        for (int i = 0; i < WIT; ++i) {
            if (i * 256 < nwvec) {                        // uniform: exact trip count for this J
                const bool cok = !tail || c0 + wr[i] < Ctot;
                const float* wp = cok ? wc + wofs[i] : a.W;
                if constexpr (VECW) {
                    wreg[i] = *reinterpret_cast<const f32x4*>(wp);
                } else {
                    const int f = tid + i * 256;
                    const int c4 = f % NT4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool eok = n0 + c4 * 4 + e < a.N;
                        wreg[i][e] = eok ? wp[e] : 0.f;
                    }
                }
                if (!cok) wmask &= ~(1u << i);
            }
        }
    };
    // ---- registers -> LDS (zero fill applied here) ----
    auto store_chunk = [&]() {
        if (!deint) {
            const int lr = tid % TPR;
            float* xd = Xs + xrow * XP + lr;
#pragma unroll
            for (int i = 0; i < XIT_D; ++i)
                if (lr + i * TPR < UW) xd[i * TPR] = ((xmask >> i) & 1u) ? xreg[i] : 0.f;
        } else {
            const int le = tid % TPC;
            float* xd = Xs + ((le & 1) * CH + xrow) * XP + (le >> 1);
#pragma unroll
            for (int i = 0; i < XIT_I; ++i)
                if (le + i * TPC < 2 * UW) xd[i * (TPC / 2)] = ((xmask >> i) & 1u) ? xreg[i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < WIT; ++i) {
            if (i * 256 < nwvec) {
                const int f = tid + i * 256;
                const int row = f / NT4, c4 = f % NT4;
                if (f < nwvec)
                    *reinterpret_cast<f32x4*>(&Ws[row * WP + c4 * 4]) =
                        ((wmask >> i) & 1u) ? wreg[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };

#ifdef WUN_ABLATION
    const bool ab_noload = a.flags & 256, ab_nostore = a.flags & 512, ab_nomfma = a.flags & 1024,
               ab_noepi = a.flags & 2048, ab_nobar = a.flags & 4096;
#else
    constexpr bool ab_noload = false, ab_nostore = false, ab_nomfma = false, ab_noepi = false, ab_nobar = false;
#endif
    if (ch_lo < ch_hi && !ab_noload) load_chunk(ch_lo);
    for (int chunk = ch_lo; chunk < ch_hi; ++chunk) {
        if (!ab_nobar) __syncthreads();                 // every wave is done reading the previous chunk
        if (!ab_nostore) store_chunk();
        if (!ab_nobar) __syncthreads();
        if (chunk + 1 < ch_hi && !ab_noload) load_chunk(chunk + 1);   // in flight during the MFMAs below
        if (!ab_nomfma) {
            // flat k-steps s = j*KS + ks (4 LDS rows each); operands of step s+1 are read from
            // LDS while the MFMAs of step s issue (double-buffered registers)
            constexpr int KS = CK / 4;
            int nsteps = J * KS;
            if (deint && CK == 8 && (a.KW & 1)) nsteps -= 1;       // the odd phase has no tap KW
            const float* xb = Xs + lg * XP + wt0 + li;
            const float* wbp = Ws + lg * WP + wn0 + li;
            float a0[MT], b0[NW], a1[MT], b1[NW];
            auto ldop = [&](int st, float (&av)[MT], float (&bv)[NW]) {
                const int j = st / KS, ks = st % KS;
                const float* xa = xb + j + ks * 4 * XP;
                const float* wb = wbp + (j * CK + ks * 4) * WP;
#pragma unroll
                for (int m = 0; m < MT; ++m) av[m] = xa[m * 16];
#pragma unroll
                for (int n = 0; n < NW; ++n) bv[n] = wb[n * 16];
            };
            auto mm = [&](const float (&av)[MT], const float (&bv)[NW]) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NW; ++n) acc[m][n] = mfma16(av[m], bv[n], acc[m][n]);
            };
            if (nsteps > 0) {
                ldop(0, a0, b0);
                int st = 0;
                for (; st + 1 < nsteps; st += 2) {
                    ldop(st + 1, a1, b1);
                    mm(a0, b0);
                    if (st + 2 < nsteps) ldop(st + 2, a0, b0);
                    mm(a1, b1);
                }
                if (nsteps & 1) mm(a0, b0);
            }
        }
    }

    // ---- split-K: raw partial tile, epilogue runs in conv_splitk_epilogue_kernel ----
    if (a.part != nullptr) {
        const int TP = (a.Tout + 3) & ~3;
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            const int ncol = n0 + wn0 + n * 16 + li;
            if (ncol >= a.N) continue;
            float* prow = a.part + (((long long)ksp * a.B + b) * a.N + ncol) * TP;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int q = q0 + wt0 + m * 16 + lg * 4;
                if (q < TP) *reinterpret_cast<f32x4*>(&prow[q]) = acc[m][n];
            }
        }
        return;
    }

    // ---- epilogue ----
    const bool lrelu = (a.flags & F_LRELU) != 0;
    const bool accum = (a.flags & F_ACCUM) != 0;
    const bool vec = (a.flags & F_VEC4) != 0;
    if (ab_noepi && acc[0][0][0] != 12345.678f) return;
#pragma unroll
    for (int n = 0; n < NW; ++n) {
        const int ncol = n0 + wn0 + n * 16 + li;
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
                if (accum) {
                    const f32x4 old = *reinterpret_cast<const f32x4*>(&dst[idx]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += old[r];
                }
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

// sums the split-K partial tiles in a fixed order and applies the conv epilogue
__global__ void conv_splitk_epilogue_kernel(ConvArgs a, int ksplit) {
    const int TP = (a.Tout + 3) & ~3;
    const long long total = (long long)a.B * a.N * a.Tout;
    const bool lrelu = (a.flags & F_LRELU) != 0;
    const bool accum = (a.flags & F_ACCUM) != 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(i % a.Tout);
        const long long bn = i / a.Tout;
        const int ncol = (int)(bn % a.N), b = (int)(bn / a.N);
        float v = 0.f;
        for (int ks = 0; ks < ksplit; ++ks)
            v += a.part[(((long long)ks * a.B + b) * a.N + ncol) * TP + q];
        if (a.bias != nullptr) v += a.bias[ncol];
        if (lrelu) v = fmaxf(0.2f * v, v);
        float* dst; const float* msk; long long idx;
        if (ncol < a.N0) {
            idx = (long long)b * a.obs0 + (long long)ncol * a.opitch0 + a.ooff0 + (long long)q * a.ostride;
            dst = a.dst0; msk = a.msk0;
        } else {
            idx = (long long)b * a.obs1 + (long long)(ncol - a.N0) * a.opitch1 + a.ooff1 + (long long)q * a.ostride;
            dst = a.dst1; msk = a.msk1;
        }
        if (msk != nullptr) v *= (msk[idx] > 0.f) ? 1.f : 0.2f;
        if (accum) v += dst[idx];
        dst[idx] = v;
        if (a.dec != nullptr && ncol < a.N0 && (q & 1) == 0)
            a.dec[(long long)b * a.decbs + (long long)ncol * a.decpitch + (q >> 1)] = v;
    }
}

// variant table -------------------------------------------------------------------------
struct ConvVariant { int MT, NW, WT, WN, CK; };
static const ConvVariant kConvVariants[] = {
    {4, 2, 4, 1, 8}, {4, 3, 4, 1, 8}, {4, 4, 4, 1, 8}, {4, 5, 4, 1, 8},   //  0..3 : 256 x 32/48/64/80
    {2, 2, 4, 1, 8}, {2, 3, 4, 1, 8}, {2, 4, 4, 1, 8}, {2, 5, 4, 1, 8},   //  4..7 : 128 x 32/48/64/80
    {1, 2, 4, 1, 8}, {1, 3, 4, 1, 8},                                     //  8..9 :  64 x 32/48
    {1, 2, 2, 2, 8}, {1, 3, 2, 2, 8},                                     // 10..11:  32 x 64/96
    {1, 2, 1, 4, 8},                                                      // 12    :  16 x 128
    {4, 2, 4, 1, 4}, {1, 2, 4, 1, 4},                                     // 13..14: 1-/2-channel audio input
};

static inline int conv_J(const ConvArgs& a) { return a.loader == LOADER_DEINT ? (a.KW + 1) / 2 : a.KW; }

static int pick_nw(int N, const int* cands, int ncand) {
    // fewest padded columns; ties resolved by the order of `cands`
    int best = cands[0], bestpad = 1 << 30;
    for (int i = 0; i < ncand; ++i) {
        const int nt = cands[i] * 16;
        const int padded = ((N + nt - 1) / nt) * nt;
        if (padded < bestpad) { bestpad = padded; best = cands[i]; }
    }
    return best;
}

int conv_pick_variant(const ConvArgs& a) {
    const int Ctot = a.C0 + a.C1;
    if (Ctot <= 4) return a.Tout > 64 ? 13 : 14;
    if (a.Tout <= 16) return 12;
    if (a.Tout <= 32) { const int c[2] = {3, 2}; return pick_nw(a.N, c, 2) == 3 ? 11 : 10; }
    if (a.Tout <= 64) { const int c[2] = {3, 2}; return pick_nw(a.N, c, 2) == 3 ? 9 : 8; }
    const int pad256 = ((a.Tout + 255) / 256) * 256, pad128 = ((a.Tout + 127) / 128) * 128;
    const int base = (pad128 < pad256) ? 4 : 0;
    const int c[4] = {3, 5, 4, 2};
    const int nw = pick_nw(a.N, c, 4);
    return base + (nw - 2);
}

static void conv_geom(const ConvArgs& a, int variant, int& TT, int& NT, int& J, int& XP, int& WP) {
    const ConvVariant& v = kConvVariants[variant];
    TT = v.WT * v.MT * 16;
    NT = v.WN * v.NW * 16;
    J = conv_J(a);
    XP = fit_pitch(TT + J - 1, 16);
    WP = fit_pitch(NT, 16);
}

size_t conv_lds_bytes(const ConvArgs& a, int variant) {
    int TT, NT, J, XP, WP;
    conv_geom(a, variant, TT, NT, J, XP, WP);
    const int CK = kConvVariants[variant].CK;
    return sizeof(float) * ((size_t)CK * XP + (size_t)J * CK * WP);
}

double conv_flops(const ConvArgs& a) {
    return 2.0 * a.KW * (double)(a.C0 + a.C1) * a.N * (double)a.Tout * a.B;
}

// split-K decision: (ksplit, chunks per split) from shapes only (deterministic)
void conv_splitk(const ConvArgs& a, int variant, long long part_cap_floats, int& ksplit, int& cps) {
    int TT, NT, J, XP, WP;
    conv_geom(a, variant, TT, NT, J, XP, WP);
    const int CK = kConvVariants[variant].CK;
    const int CKC = a.loader == LOADER_DEINT ? CK / 2 : CK;
    const int nchunks = (a.C0 + a.C1 + CKC - 1) / CKC;
    const long long natural = (long long)((a.Tout + TT - 1) / TT) * ((a.N + NT - 1) / NT) * a.B;
    ksplit = 1; cps = nchunks;
    if (natural >= 512 || nchunks < 2 || part_cap_floats <= 0) return;
    long long want = (1024 + natural - 1) / natural;
    if (want > nchunks) want = nchunks;
    const long long per = (long long)a.B * a.N * ((a.Tout + 3) & ~3);
    if (want * per > part_cap_floats) want = part_cap_floats / per;
    if (want < 2) return;
    cps = (int)((nchunks + want - 1) / want);
    ksplit = (nchunks + cps - 1) / cps;
    if (ksplit < 2) { ksplit = 1; cps = nchunks; }
}

template <int MT, int NW, int WT, int WN, int CK, bool VECW>
static hipError_t conv_launch_t(ConvArgs a, int variant, float* part, long long part_cap, hipStream_t s) {
    int TT, NT, J, XP, WP;
    conv_geom(a, variant, TT, NT, J, XP, WP);
    const int nTT = (a.Tout + TT - 1) / TT, nNT = (a.N + NT - 1) / NT;
    const size_t lds = conv_lds_bytes(a, variant);
    auto kern = conv_mfma_kernel<MT, NW, WT, WN, CK, VECW>;
    static size_t lds_allowed = 64 * 1024;
    if (lds > lds_allowed) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_allowed = lds;
    }
    int ksplit, cps;
    conv_splitk(a, variant, part != nullptr ? part_cap : 0, ksplit, cps);
    a.cps = cps;
    a.part = ksplit > 1 ? part : nullptr;
    const long long grid = (long long)nTT * nNT * a.B * ksplit;
    if (grid <= 0) return hipSuccess;
    char nm[64];
    snprintf(nm, sizeof(nm), "conv_mfma_kernel<%d, %d, %d, %d, %d, %s>", MT, NW, WT, WN, CK, VECW ? "true" : "false");
    {
        ProfScope ps(nm, conv_flops(a), s);
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, a, nTT, nNT, J, XP, WP);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || ksplit == 1) return e;
    const long long total = (long long)a.B * a.N * a.Tout;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv_splitk_epilogue_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, ksplit);
    return hipGetLastError();
}

// Vector (16-byte) paths need aligned bases / pitches; everything the plan allocates is.
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

hipError_t launch_conv(const ConvArgs& a_in, float* part, long long part_cap, hipStream_t s) {
    ConvArgs a = a_in;
    if (conv_J(a) > WUN_JMAX) return hipErrorInvalidValue;       // rejected at plan creation
    const bool vecw = (a.N & 3) == 0 && aligned16(a.W);
    bool vec = a.ostride == 1 && aligned16(a.dst0) && (a.obs0 & 3) == 0 && (a.opitch0 & 3) == 0 && (a.ooff0 & 3) == 0;
    if (a.dst1 != nullptr)
        vec = vec && aligned16(a.dst1) && (a.obs1 & 3) == 0 && (a.opitch1 & 3) == 0 && (a.ooff1 & 3) == 0;
    if (a.msk0 != nullptr) vec = vec && aligned16(a.msk0);
    if (a.msk1 != nullptr) vec = vec && aligned16(a.msk1);
    if (a.dec != nullptr) vec = vec && (a.decpitch & 1) == 0 && (a.decbs & 1) == 0;
    if (vec) a.flags |= F_VEC4;
#ifdef WUN_ABLATION
    if (const char* e = getenv("WUN_ABLATE")) a.flags |= atoi(e) << 8;
    if (const char* e = getenv("WUN_NOVEC")) if (atoi(e)) a.flags &= ~F_VEC4;
#endif
    int v = conv_pick_variant(a);
#ifdef WUN_ABLATION
    if (const char* e = getenv("WUN_VARIANT")) v = atoi(e);
#endif
    if (!vecw) {
        // channel counts that are not multiples of 4 (non-shipped configs): scalar weight loads,
        // restricted tile menu
        const int Ctot = a.C0 + a.C1;
        if (Ctot <= 4) v = a.Tout > 64 ? 13 : 14;
        else v = a.Tout <= 16 ? 12 : (a.Tout <= 64 ? 9 : 1);
        switch (v) {
            case 1: return conv_launch_t<4, 3, 4, 1, 8, false>(a, v, part, part_cap, s);
            case 9: return conv_launch_t<1, 3, 4, 1, 8, false>(a, v, part, part_cap, s);
            case 12: return conv_launch_t<1, 2, 1, 4, 8, false>(a, v, part, part_cap, s);
            case 13: return conv_launch_t<4, 2, 4, 1, 4, false>(a, v, part, part_cap, s);
            default: return conv_launch_t<1, 2, 4, 1, 4, false>(a, v, part, part_cap, s);
        }
    }
#define WUN_CV(i, MT, NW, WT, WN, CK) case i: return conv_launch_t<MT, NW, WT, WN, CK, true>(a, v, part, part_cap, s);
    switch (v) {
        WUN_CV(0, 4, 2, 4, 1, 8) WUN_CV(1, 4, 3, 4, 1, 8) WUN_CV(2, 4, 4, 4, 1, 8) WUN_CV(3, 4, 5, 4, 1, 8)
        WUN_CV(4, 2, 2, 4, 1, 8) WUN_CV(5, 2, 3, 4, 1, 8) WUN_CV(6, 2, 4, 4, 1, 8) WUN_CV(7, 2, 5, 4, 1, 8)
        WUN_CV(8, 1, 2, 4, 1, 8) WUN_CV(9, 1, 3, 4, 1, 8)
        WUN_CV(10, 1, 2, 2, 2, 8) WUN_CV(11, 1, 3, 2, 2, 8)
        WUN_CV(12, 1, 2, 1, 4, 8)
        WUN_CV(13, 4, 2, 4, 1, 4) WUN_CV(14, 1, 2, 4, 1, 4)
        default: return hipErrorInvalidValue;
    }
#undef WUN_CV
}

// =====================================================================================
// weight / bias gradient
// =====================================================================================
// D[(cin,tap) 16][cout 16] += A[(cin,tap)][q] * B[q][cout]; reduction over (batch, q) split
// over workgroups ("units" of TK <= 128 output positions), partial sums to scratch, summed
// in a fixed order by reduce_splits_*.  Row (cin,tap) of A is the input row `cin` read at
// offset `tap`, so one staged input row serves all taps.  Staging is 16-byte global loads
// into registers one unit ahead of the MFMAs (aligned vectors; the sub-vector shift
// `delta` is folded into the LDS read offset), de-interleaved into even/odd planes for the
// stride-2 convs.  An extra all-ones A row produces the bias gradient.
#define WUN_WG_XIT 8      // float4 X loads per thread and unit (geometry guarantees it suffices)

template <int MTW, int NW>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(WgradArgs a, int nMG, int nNG, int TK,
                                                         int XP, int ZP, int nChMax, int ONESP,
                                                         int XW4) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int MG = 4 * MTW * 16;
    constexpr int NG = NW * 16;
    constexpr int ZIT = (NG * 32 + 255) / 256;          // TK/4 <= 32 float4 per dz row
    const bool deint = (a.loader == LOADER_DEINT);
    const int planes = deint ? 2 : 1;
    float* Xs = lds + ONESP;
    float* Zs = Xs + nChMax * planes * XP;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
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

    // (off - shift) mod 4 fixes the sub-vector shift of every unit (q0 is a multiple of 4)
    const int delta0 = ((a.off0 - a.shift) % 4 + 4) % 4;
    const int delta1 = ((a.off1 - a.shift) % 4 + 4) % 4;

    int rowoff[MTW];
    int nact = 0;                                      // wave-uniform count of live M tiles
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int rt = rlo + (wave * MTW + mt) * 16;
        if (rt <= Mtot) nact = mt + 1;
        const int r = rt + li;
        int off = 0;                                   // ones row
        if (r < Mtot) {
            const int c = r / a.KW, k = r - c * a.KW;
            const int kd = k + (c < a.C0 ? delta0 : delta1);
            off = ONESP + (c - cLo) * planes * XP + (deint ? ((kd & 1) * XP + (kd >> 1)) : kd);
        }
        rowoff[mt] = off;
    }
    for (int i = tid; i < ONESP; i += 256) lds[i] = 1.f;

    f32x4 acc[MTW][NW];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 xreg[WUN_WG_XIT];
    f32x4 zreg[ZIT];
    const int TK4 = TK >> 2;
    const float inv_xw4 = 1.0f / (float)XW4, inv_tk4 = 1.0f / (float)TK4;

    // Buffers are in the plan's canonical layout: row pitch % 4 == 0, 16-byte aligned rows,
    // off + Tin <= pitch.  Every vector is loaded from an in-row clamped position; the zero
    // fill (virtual time outside [0, Tin), rows beyond the group) is applied when the
    // registers are written to LDS, so the loads stay in flight during the previous unit.
    auto load_unit = [&](int u) {
        const int b = u / a.nQT, qt = u - b * a.nQT;
        const int q0 = qt * TK;
        const int tb = (deint ? 2 * q0 : q0) - a.shift;
        const float* base0 = a.src0 + (long long)b * a.bs0;
        const float* base1 = (a.C1 > 0) ? a.src1 + (long long)b * a.bs1 : base0;
#pragma unroll
        for (int i = 0; i < WUN_WG_XIT; ++i) {
            const int f = tid + i * 256;
            const int row = (int)(((float)f + 0.5f) * inv_xw4);
            const int c4 = f - row * XW4;
            const int c = cLo + (row < nCh ? row : 0);
            const bool s0 = c < a.C0;
            const int off = s0 ? a.off0 : a.off1;
            const int pitch = s0 ? a.pitch0 : a.pitch1;
            const int e0 = ((tb + off) & ~3) + 4 * c4;           // element index in the row (multiple of 4)
            int e0c = e0 < 0 ? 0 : e0;
            if (e0c > pitch - 4) e0c = pitch - 4;
            const int rel = (s0 ? c * a.pitch0 : (c - a.C0) * a.pitch1) + e0c;
            xreg[i] = *reinterpret_cast<const f32x4*>((s0 ? base0 : base1) + rel);
        }
        const float* zb = a.dz + (long long)b * a.dzbs + q0;
#pragma unroll
        for (int i = 0; i < ZIT; ++i) {
            const int f = tid + i * 256;
            const int row = (int)(((float)f + 0.5f) * inv_tk4);
            const int c4 = f - row * TK4;
            int nn = ng * NG + row;
            if (!(row < NG && nn < a.N)) nn = 0;
            int qq = 4 * c4;
            if (q0 + qq > a.dzpitch - 4) qq = a.dzpitch - 4 - q0;
            zreg[i] = *reinterpret_cast<const f32x4*>(zb + (long long)nn * a.dzpitch + qq);
        }
    };
    auto store_unit = [&](int u) {
        const int b = u / a.nQT, qt = u - b * a.nQT;
        const int q0 = qt * TK;
        const int tb = (deint ? 2 * q0 : q0) - a.shift;
        int nq = a.Tq - q0; if (nq > TK) nq = TK;
#pragma unroll
        for (int i = 0; i < WUN_WG_XIT; ++i) {
            const int f = tid + i * 256;
            const int row = (int)(((float)f + 0.5f) * inv_xw4);
            const int c4 = f - row * XW4;
            if (row < nCh) {
                const int c = cLo + row;
                const int off = (c < a.C0) ? a.off0 : a.off1;
                const int t0 = ((tb + off) & ~3) + 4 * c4 - off;     // virtual time of element 0
                f32x4 v = xreg[i];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (t0 + k < 0 || t0 + k >= a.Tin) v[k] = 0.f;
                if (!deint) {
                    *reinterpret_cast<f32x4*>(&Xs[row * XP + 4 * c4]) = v;
                } else {
                    float* p0 = &Xs[(row * 2) * XP + 2 * c4];
                    float* p1 = &Xs[(row * 2 + 1) * XP + 2 * c4];
                    *reinterpret_cast<float2*>(p0) = make_float2(v[0], v[2]);
                    *reinterpret_cast<float2*>(p1) = make_float2(v[1], v[3]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < ZIT; ++i) {
            const int f = tid + i * 256;
            const int row = (int)(((float)f + 0.5f) * inv_tk4);
            const int c4 = f - row * TK4;
            if (row < NG) {
                const bool rok = ng * NG + row < a.N;
                f32x4 v = zreg[i];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (!rok || 4 * c4 + k >= nq) v[k] = 0.f;
                float* p = &Zs[row * ZP + 4 * c4];
                *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
                *reinterpret_cast<float2*>(p + 2) = make_float2(v[2], v[3]);
            }
        }
    };

    const int nunits = a.B * a.nQT;
    const int u0 = split * a.units_per_split;
    int u1 = u0 + a.units_per_split;
    if (u1 > nunits) u1 = nunits;
    if (u0 < u1) load_unit(u0);
    for (int u = u0; u < u1; ++u) {
        const int qt = u % a.nQT;
        int nq = a.Tq - qt * TK; if (nq > TK) nq = TK;
        const int nsteps = (nq + 3) >> 2;
        __syncthreads();
        store_unit(u);
        __syncthreads();
        if (u + 1 < u1) load_unit(u + 1);
        {
            float a0[MTW], b0[NW], a1[MTW], b1[NW];
            auto ldop = [&](int st, float (&av)[MTW], float (&bv)[NW]) {
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) av[mt] = lds[rowoff[mt] + 4 * st + lg];
#pragma unroll
                for (int n = 0; n < NW; ++n) bv[n] = Zs[(n * 16 + li) * ZP + 4 * st + lg];
            };
            auto mm = [&](const float (&av)[MTW], const float (&bv)[NW]) {
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    if (mt < nact) {
#pragma unroll
                        for (int n = 0; n < NW; ++n) acc[mt][n] = mfma16(av[mt], bv[n], acc[mt][n]);
                    }
                }
            };
            ldop(0, a0, b0);
            int st = 0;
            for (; st + 1 < nsteps; st += 2) {
                ldop(st + 1, a1, b1);
                mm(a0, b0);
                if (st + 2 < nsteps) ldop(st + 2, a0, b0);
                mm(a1, b1);
            }
            if (nsteps & 1) mm(a0, b0);
        }
    }

    float* outp = a.out + (long long)split * a.split_stride;
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

struct WgradGeom { int MTW, NW, nMG, nNG, TK, XP, ZP, nChMax, ONESP, XW4; size_t lds; };

static WgradGeom wgrad_geom(const WgradArgs& a) {
    WgradGeom g;
    const int Ctot = a.C0 + a.C1;
    const int mtiles = (Ctot * a.KW + 1 + 15) / 16;
    int bestnw = 3, bestpad = 1 << 30;
    for (int nw = 3; nw >= 1; --nw) {
        const int padded = ((a.N + nw * 16 - 1) / (nw * 16)) * nw * 16;
        if (padded < bestpad) { bestpad = padded; bestnw = nw; }
    }
    g.NW = bestnw;
    int tk = (a.Tq + 3) & ~3;
    if (tk > 128) tk = 128;
    if (tk < 4) tk = 4;
    g.TK = tk;
    const bool deint = a.loader == LOADER_DEINT;
    const int Jx = deint ? (a.KW + 1) / 2 : a.KW;
    // float4 per staged input row: span of the unit + up to 3 elements of alignment slack
    g.XW4 = deint ? (2 * (tk + Jx - 1) + 3 + 3) / 4 : (tk + a.KW - 1 + 3 + 3) / 4;
    const int mod = deint ? 10 : (a.KW >= 9 ? 16 : (a.KW >= 5 ? 8 : 4));
    g.XP = fit_pitch(deint ? 2 * g.XW4 : 4 * g.XW4, mod);
    g.ZP = fit_pitch(tk, 2);
    // the M-group (rows of 4*MTW*16 (cin,tap) pairs) must stage within WUN_WG_XIT vectors/thread
    int mtw = mtiles <= 4 ? 1 : (mtiles <= 8 ? 2 : 6);
    for (;;) {
        const int MG = 4 * mtw * 16;
        int nch = (MG + a.KW - 2) / a.KW + 1;
        if (nch > Ctot) nch = Ctot;
        g.nChMax = nch;
        if ((long long)nch * g.XW4 <= (long long)WUN_WG_XIT * 256 || mtw == 1) break;
        mtw = mtw == 6 ? 4 : (mtw == 4 ? 2 : 1);
    }
    g.MTW = mtw;
    const int MG = 4 * g.MTW * 16, NG = g.NW * 16;
    g.nMG = (Ctot * a.KW + 1 + MG - 1) / MG;
    g.nNG = (a.N + NG - 1) / NG;
    g.ONESP = (tk + 15) & ~15;
    g.lds = sizeof(float) * ((size_t)g.ONESP + (size_t)g.nChMax * (deint ? 2 : 1) * g.XP + (size_t)NG * g.ZP);
    return g;
}

int wgrad_pick_nsplit(const WgradArgs& a) {
    WgradGeom g = wgrad_geom(a);
    const int nQT = (a.Tq + g.TK - 1) / g.TK;
    const long long units = (long long)a.B * nQT;
    const long long per = (long long)g.nMG * g.nNG;
    long long ns = (1024 + per - 1) / per;
    if (ns > units) ns = units;
    if (ns < 1) ns = 1;
    // keep each split at >= 2 units when there is plenty of work, to amortise the epilogue
    if (units >= 8 && ns > units / 2) ns = units / 2;
    const long long ups = (units + ns - 1) / ns;
    ns = (units + ups - 1) / ups;
    return (int)ns;
}

template <int MTW, int NW>
static hipError_t wgrad_launch_t(WgradArgs a, const WgradGeom& g, hipStream_t s) {
    a.nQT = (a.Tq + g.TK - 1) / g.TK;
    const long long units = (long long)a.B * a.nQT;
    a.units_per_split = (int)((units + a.nsplit - 1) / a.nsplit);
    auto kern = wgrad_mfma_kernel<MTW, NW>;
    static size_t lds_allowed = 64 * 1024;
    if (g.lds > lds_allowed) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
        if (e != hipSuccess) return e;
        lds_allowed = g.lds;
    }
    const long long grid = (long long)g.nMG * g.nNG * a.nsplit;
    char nm[64];
    snprintf(nm, sizeof(nm), "wgrad_mfma_kernel<%d, %d>", MTW, NW);
    ProfScope ps(nm, 2.0 * a.KW * (double)(a.C0 + a.C1) * a.N * (double)a.Tq * a.B, s);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), g.lds, s, a, g.nMG, g.nNG, g.TK, g.XP,
                       g.ZP, g.nChMax, g.ONESP, g.XW4);
    return hipGetLastError();
}

hipError_t launch_wgrad(const WgradArgs& a, hipStream_t s) {
    // canonical layout required (the plan's buffers are; the single-op entry points repack)
    if ((a.pitch0 & 3) || (a.bs0 & 3) || (reinterpret_cast<uintptr_t>(a.src0) & 15) || a.pitch0 < 4) return hipErrorInvalidValue;
    if (a.C1 > 0 && ((a.pitch1 & 3) || (a.bs1 & 3) || (reinterpret_cast<uintptr_t>(a.src1) & 15) || a.pitch1 < 4)) return hipErrorInvalidValue;
    if ((a.dzpitch & 3) || (a.dzbs & 3) || (reinterpret_cast<uintptr_t>(a.dz) & 15) || a.dzpitch < 4) return hipErrorInvalidValue;
    const WgradGeom g = wgrad_geom(a);
    if ((long long)g.nChMax * g.XW4 > (long long)WUN_WG_XIT * 256) return hipErrorInvalidValue;
#define WUN_WG(M, N) if (g.MTW == M && g.NW == N) return wgrad_launch_t<M, N>(a, g, s);
    WUN_WG(1, 1) WUN_WG(1, 2) WUN_WG(1, 3)
    WUN_WG(2, 1) WUN_WG(2, 2) WUN_WG(2, 3)
    WUN_WG(4, 1) WUN_WG(4, 2) WUN_WG(4, 3)
    WUN_WG(6, 1) WUN_WG(6, 2) WUN_WG(6, 3)
#undef WUN_WG
    return hipErrorInvalidValue;
}

// out[e] = sum_s partial[s*stride + e]  (fixed order -> deterministic)
__global__ void reduce_splits_kernel(const float* __restrict__ partial, long long stride, int nsplit,
                                     float* __restrict__ out, long long n) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n;
         e += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += partial[(long long)k * stride + e];
        out[e] = s;
    }
}

// many splits, few elements: 16 element lanes x 16 split lanes per block, fixed summation order
__global__ __launch_bounds__(256) void reduce_splits_wide_kernel(const float* __restrict__ partial,
                                                                 long long stride, int nsplit,
                                                                 float* __restrict__ out, long long n) {
    __shared__ float red[16][17];
    const int ex = threadIdx.x & 15, y = threadIdx.x >> 4;
    const long long e = (long long)blockIdx.x * 16 + ex;
    float s = 0.f;
    if (e < n)
        for (int k = y; k < nsplit; k += 16) s += partial[(long long)k * stride + e];
    red[y][ex] = s;
    __syncthreads();
    if (y == 0 && e < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][ex];
        out[e] = t;
    }
}

hipError_t launch_reduce(const float* partial, long long stride, int nsplit, float* out,
                         long long n, hipStream_t s) {
    if (nsplit >= 8) {
        const long long blocks = (n + 15) / 16;
        hipLaunchKernelGGL(reduce_splits_wide_kernel, dim3((unsigned)blocks), dim3(256), 0, s, partial,
                           stride, nsplit, out, n);
        return hipGetLastError();
    }
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)blocks), dim3(256), 0, s, partial, stride,
                       nsplit, out, n);
    return hipGetLastError();
}

// =====================================================================================
// upsampling (linear / learned), forward and backward
//   UnetAudioSeparator.py:109-118, InterpolationLayer.py:4-40
// =====================================================================================
__global__ void upsample_kernel(UpsampleArgs a) {
    const long long total = (long long)a.B * a.C * a.tup;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(i % a.tup);
        const long long bc = i / a.tup;
        const int c = (int)(bc % a.C), b = (int)(bc / a.C);
        const float* x = a.x + (long long)b * a.xbs + (long long)c * a.xpitch;
        const int j = t >> 1;
        float v;
        if ((t & 1) == 0) {
            v = x[j];
        } else if (a.w != nullptr) {
            const float s = 1.f / (1.f + __expf(-a.w[c]));
            const float x1 = (j + 1 < a.n) ? x[j + 1] : 0.f;      // SAME: one zero on the right
            v = s * x[j] + (1.f - s) * x1;
        } else {
            const float x1 = (j + 1 < a.n) ? x[j + 1] : x[j];     // legacy bilinear clamps
            v = 0.5f * (x[j] + x1);
        }
        a.y[(long long)b * a.ybs + (long long)c * a.ypitch + t] = v;
    }
}

hipError_t launch_upsample(const UpsampleArgs& a, hipStream_t s) {
    const long long total = (long long)a.B * a.C * a.tup;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(upsample_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

__device__ __forceinline__ float sigmoidf_exact(float w) { return 1.f / (1.f + expf(-w)); }

// dz[b][c][i] = lrelu'(x) * ( dy[2i] + wa*dy[2i+1] + wb*dy[2i-1] )
__global__ void upsample_bwd_kernel(UpsampleBwdArgs a) {
    const long long total = (long long)a.B * a.C * a.n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(idx % a.n);
        const long long bc = idx / a.n;
        const int c = (int)(bc % a.C), b = (int)(bc / a.C);
        const float* dy = a.dy + (long long)b * a.ybs + (long long)c * a.ypitch;
        float wa = 0.5f, wb = 0.5f;
        if (a.w != nullptr) { wa = 1.f / (1.f + __expf(-a.w[c])); wb = 1.f - wa; }
        float g = dy[2 * i];
        if (2 * i + 1 < a.tup) {
            float wgt = wa;
            if (a.w == nullptr && i == a.n - 1) wgt = 1.f;       // same-mode legacy clamp: out[2n-1] = x[n-1]
            g += wgt * dy[2 * i + 1];
        }
        if (i >= 1) g += wb * dy[2 * i - 1];
        const long long xi = (long long)b * a.xbs + (long long)c * a.xpitch + i;
        g *= (a.x[xi] > 0.f) ? 1.f : 0.2f;
        a.dz[xi] = g;
    }
}

// dw[c] = sigmoid'(w[c]) * sum_{b,i} dy[2i+1] * (x[i] - x[i+1])   (x[n] = 0 in same mode)
__global__ __launch_bounds__(256) void interp_grad_kernel(UpsampleBwdArgs a) {
    __shared__ float red[256];
    const int c = blockIdx.x;
    float s = 0.f;
    const int nmid = a.tup / 2;               // number of odd outputs
    for (long long idx = threadIdx.x; idx < (long long)a.B * nmid; idx += 256) {
        const int i = (int)(idx % nmid), b = (int)(idx / nmid);
        const float* x = a.x + (long long)b * a.xbs + (long long)c * a.xpitch;
        const float x1 = (i + 1 < a.n) ? x[i + 1] : 0.f;
        s += a.dy[(long long)b * a.ybs + (long long)c * a.ypitch + 2 * i + 1] * (x[i] - x1);
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

hipError_t launch_upsample_bwd(const UpsampleBwdArgs& a, hipStream_t s) {
    const long long total = (long long)a.B * a.C * a.n;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (a.w != nullptr && a.dw != nullptr) {
        hipLaunchKernelGGL(interp_grad_kernel, dim3((unsigned)a.C), dim3(256), 0, s, a);
        e = hipGetLastError();
    }
    return e;
}

// =====================================================================================
// output head (OutputLayer.py:5-23, UnetAudioSeparator.py:127-142) + loss (Training.py:50-63)
// =====================================================================================
#define WUN_MAX_HEAD_ACC 8   // Sh*C <= 4*2

__device__ __forceinline__ int head_block_floats(const HeadArgs& a) { return a.Ko * (a.C + a.F) * a.C + a.C; }

__global__ __launch_bounds__(256) void head_fwd_kernel(HeadArgs a, long long h0, long long h1,
                                                       long long h2, long long h3) {
    extern __shared__ float hw[];
    const long long hoff[4] = {h0, h1, h2, h3};
    const int blk = a.Ko * (a.C + a.F) * a.C + a.C;
    for (int s = 0; s < a.Sh; ++s)
        for (int i = threadIdx.x; i < blk; i += 256) hw[s * blk + i] = a.Wh[hoff[s] + i];
    __syncthreads();
    const int Cin = a.C + a.F;
    const long long total = (long long)a.B * a.Tout;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * 256) {
        const int t = (int)(idx % a.Tout), b = (int)(idx / a.Tout);
        float acc[WUN_MAX_HEAD_ACC];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                acc[s * 2 + c] = (s < a.Sh && c < a.C) ? hw[s * blk + a.Ko * Cin * a.C + c] : 0.f;
        for (int k = 0; k < a.Ko; ++k) {
            const int tf = t + k - a.padl;
            if (tf < 0 || tf >= a.Tfeat) continue;
            for (int ci = 0; ci < Cin; ++ci) {
                const float xv = (ci < a.C)
                    ? a.mix_ncw[(long long)b * a.mbs + (long long)ci * a.mpitch + a.moff_feat + tf]
                    : a.feat[(long long)b * a.fbs + (long long)(ci - a.C) * a.fpitch + tf];
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
__global__ __launch_bounds__(256) void head_dfeat_kernel(HeadArgs a, long long h0, long long h1,
                                                         long long h2, long long h3) {
    extern __shared__ float hw[];
    const long long hoff[4] = {h0, h1, h2, h3};
    const int blk = a.Ko * (a.C + a.F) * a.C + a.C;
    for (int s = 0; s < a.Sh; ++s)
        for (int i = threadIdx.x; i < blk; i += 256) hw[s * blk + i] = a.Wh[hoff[s] + i];
    __syncthreads();
    const int Cin = a.C + a.F;
    const long long total = (long long)a.B * a.Tfeat;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * 256) {
        const int u = (int)(idx % a.Tfeat), b = (int)(idx / a.Tfeat);
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
            g *= (a.feat[fi] > 0.f) ? 1.f : 0.2f;
            a.dzfeat[fi] = g;
        }
    }
}

__global__ void loss_finish_kernel(const float* partial, int n, float scale, float* loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += partial[i];
        *loss = s * scale;
    }
}

hipError_t launch_loss_finish(const float* partial, int n, float scale, float* loss, hipStream_t s) {
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, s, partial, n, scale, loss);
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

hipError_t launch_btc_to_ncw(const float* src, float* dst, int B, int T, int C, int pitch,
                             hipStream_t s) {
    const long long total = (long long)B * T * C;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(btc_to_ncw_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, B, T, C, pitch);
    return hipGetLastError();
}

__global__ void make_wt_kernel(const float* __restrict__ params, float* __restrict__ ws,
                               const WtDesc* __restrict__ descs) {
    const WtDesc d = descs[blockIdx.y];
    const long long total = (long long)d.J * d.N * d.C;
    const float* src = params + d.src_off;
    float* dst = ws + d.dst_off;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % d.C);
        const long long jn = i / d.C;
        const int n = (int)(jn % d.N), j = (int)(jn / d.N);
        const int k = d.k_last - j * d.k_step;
        dst[i] = src[((long long)k * d.C + c) * d.N + n];
    }
}

hipError_t launch_make_wt(const float* params, float* ws, const WtDesc* dev_descs, int ndesc,
                          int max_elems, hipStream_t s) {
    if (ndesc <= 0) return hipSuccess;
    int bx = (max_elems + 255) / 256;
    if (bx > 256) bx = 256;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(make_wt_kernel, dim3(bx, ndesc), dim3(256), 0, s, params, ws, dev_descs);
    return hipGetLastError();
}

__global__ void make_wt_one_kernel(const float* __restrict__ src, float* __restrict__ dst, WtDesc d) {
    const long long total = (long long)d.J * d.N * d.C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % d.C);
        const long long jn = i / d.C;
        const int n = (int)(jn % d.N), j = (int)(jn / d.N);
        const int k = d.k_last - j * d.k_step;
        dst[i] = src[((long long)k * d.C + c) * d.N + n];
    }
}

hipError_t launch_make_wt_one(const float* src, float* dst, WtDesc d, hipStream_t s) {
    const long long total = (long long)d.J * d.N * d.C;
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

// one wave: d[16][16] = a[16][4] * b[4][16] through the documented lane layout
__global__ void mfma_probe_kernel(const float* a, const float* b, float* d) {
    const int lane = threadIdx.x & 63;
    const float av = a[(lane & 15) * 4 + (lane >> 4)];
    const float bv = b[(lane >> 4) * 16 + (lane & 15)];
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = mfma16(av, bv, c);
    for (int r = 0; r < 4; ++r) d[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = c[r];
}

hipError_t launch_mfma_probe(const float* a, const float* b, float* d, hipStream_t s) {
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, s, a, b, d);
    return hipGetLastError();
}

// ---- head launch wrappers (need the per-source offsets) --------------------------------
static size_t head_lds(const HeadArgs& a) {
    return sizeof(float) * (size_t)a.Sh * (a.Ko * (a.C + a.F) * a.C + a.C);
}

hipError_t launch_head_fwd_off(const HeadArgs& a, const long long* hoff, hipStream_t s) {
    const long long total = (long long)a.B * a.Tout;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)blocks), dim3(256), head_lds(a), s, a, hoff[0],
                       hoff[1], hoff[2], hoff[3]);
    return hipGetLastError();
}

int head_bwd_blocks(const HeadArgs& a) {
    const long long total = (long long)a.B * a.Tout;
    long long blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    return (int)blocks;
}

hipError_t launch_head_bwd_off(const HeadArgs& a, const long long* hoff, hipStream_t s) {
    hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)head_bwd_blocks(a)), dim3(256), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const long long total = (long long)a.B * a.Tfeat;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(head_dfeat_kernel, dim3((unsigned)blocks), dim3(256), head_lds(a), s, a, hoff[0],
                       hoff[1], hoff[2], hoff[3]);
    return hipGetLastError();
}

}  // namespace wun
