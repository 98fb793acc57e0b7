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
template <int MT, int NW, int WT, int WN, int CK>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a, int nTT, int nNT, int J,
                                                        int XP, int WP) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TT = WT * MT * 16;
    constexpr int NT = WN * NW * 16;
    float* Xs = lds;
    float* Ws = lds + CK * XP;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    const int nt = bid % nNT; bid /= nNT;
    const int tt = bid % nTT;
    const int b = bid / nTT;
    const int q0 = tt * TT, n0 = nt * NT;
    const int wt0 = (wave % WT) * MT * 16;
    const int wn0 = (wave / WT) * NW * 16;
    const int Ctot = a.C0 + a.C1;
    const bool deint = (a.loader == LOADER_DEINT);
    const int UW = TT + J - 1;
    const int CKC = deint ? CK / 2 : CK;       // input channels per chunk

    f32x4 acc[MT][NW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int c0 = 0; c0 < Ctot; c0 += CKC) {
        __syncthreads();
        // ---- stage the input window (with halo) ----
        if (!deint) {
            constexpr int TPR = 256 / CK;
            const int r = tid / TPR, lr = tid % TPR;
            const int c = c0 + r;
            const float* p = nullptr;
            if (c < a.C0) p = a.src0 + (long long)b * a.bs0 + (long long)c * a.pitch0 + a.off0;
            else if (c < Ctot) p = a.src1 + (long long)b * a.bs1 + (long long)(c - a.C0) * a.pitch1 + a.off1;
            const int tbase = q0 - a.shift;
            for (int u = lr; u < UW; u += TPR) {
                const int t = tbase + u;
                float v = 0.f;
                if (p != nullptr && t >= 0 && t < a.Tin) v = p[t];
                Xs[r * XP + u] = v;
            }
        } else {
            constexpr int CH = CK / 2;
            constexpr int TPR = 256 / CH;
            const int cc = tid / TPR, le = tid % TPR;
            const int c = c0 + cc;
            const float* p = nullptr;
            if (c < a.C0) p = a.src0 + (long long)b * a.bs0 + (long long)c * a.pitch0 + a.off0;
            else if (c < Ctot) p = a.src1 + (long long)b * a.bs1 + (long long)(c - a.C0) * a.pitch1 + a.off1;
            const int tbase = 2 * q0 - a.shift;
            for (int e = le; e < 2 * UW; e += TPR) {
                const int t = tbase + e;
                float v = 0.f;
                if (p != nullptr && t >= 0 && t < a.Tin) v = p[t];
                Xs[((e & 1) * CH + cc) * XP + (e >> 1)] = v;
            }
        }
        // ---- stage the weight slab [J][CK][NT] ----
        for (int row = tid >> 4; row < J * CK; row += 16) {
            const int j = row / CK, r = row % CK;
            const float* wp = nullptr;
            if (!deint) {
                const int c = c0 + r;
                if (c < Ctot) wp = a.W + ((long long)j * Ctot + c) * a.N + n0;
            } else {
                const int ph = r / (CK / 2), cc = r % (CK / 2);
                const int k = 2 * j + ph, c = c0 + cc;
                if (k < a.KW && c < Ctot) wp = a.W + ((long long)k * Ctot + c) * a.N + n0;
            }
            for (int n = tid & 15; n < NT; n += 16)
                Ws[row * WP + n] = (wp != nullptr && n0 + n < a.N) ? wp[n] : 0.f;
        }
        __syncthreads();
        // ---- MFMA over (tap, channel-quad) ----
        for (int j = 0; j < J; ++j) {
            const float* xa = Xs + lg * XP + wt0 + li + j;
            const float* wb = Ws + (j * CK + lg) * WP + wn0 + li;
#pragma unroll
            for (int ks = 0; ks < CK / 4; ++ks) {
                if (deint && CK == 8 && ks == 1 && 2 * j + 1 >= a.KW) continue;  // odd phase has no such tap
                float av[MT], bv[NW];
#pragma unroll
                for (int m = 0; m < MT; ++m) av[m] = xa[ks * 4 * XP + m * 16];
#pragma unroll
                for (int n = 0; n < NW; ++n) bv[n] = wb[ks * 4 * WP + n * 16];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NW; ++n) acc[m][n] = mfma16(av[m], bv[n], acc[m][n]);
            }
        }
    }

    // ---- epilogue ----
    const bool lrelu = (a.flags & F_LRELU) != 0;
    const bool accum = (a.flags & F_ACCUM) != 0;
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

// variant table -------------------------------------------------------------------------
struct ConvVariant { int MT, NW, WT, WN, CK; };
static const ConvVariant kConvVariants[] = {
    {4, 2, 4, 1, 8},   // 0: 256 x 32
    {4, 3, 4, 1, 8},   // 1: 256 x 48
    {4, 4, 4, 1, 8},   // 2: 256 x 64
    {4, 5, 4, 1, 8},   // 3: 256 x 80
    {4, 6, 4, 1, 8},   // 4: 256 x 96
    {2, 3, 2, 2, 8},   // 5:  64 x 96
    {1, 2, 1, 4, 8},   // 6:  16 x 128
    {4, 2, 4, 1, 4},   // 7: 256 x 32, 4-channel chunks (1- or 2-channel audio input)
    {1, 3, 2, 2, 8},   // 8:  32 x 96
};
static const int kNumConvVariants = sizeof(kConvVariants) / sizeof(kConvVariants[0]);

static inline int conv_J(const ConvArgs& a) { return a.loader == LOADER_DEINT ? (a.KW + 1) / 2 : a.KW; }

int conv_pick_variant(const ConvArgs& a) {
    const int Ctot = a.C0 + a.C1;
    if (Ctot <= 4) return 7;
    if (a.Tout <= 24) return 6;
    if (a.Tout <= 48) return 8;
    if (a.Tout <= 160) return 5;
    // wide-time variants: choose the cout tile that wastes the fewest padded columns
    int best = 1, bestpad = 1 << 30;
    for (int v = 4; v >= 0; --v) {
        const int nt = kConvVariants[v].NW * 16;
        const int padded = ((a.N + nt - 1) / nt) * nt;
        if (padded < bestpad) { bestpad = padded; best = v; }
    }
    return best;
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

template <int MT, int NW, int WT, int WN, int CK>
static hipError_t conv_launch_t(const ConvArgs& a, int variant, hipStream_t s) {
    int TT, NT, J, XP, WP;
    conv_geom(a, variant, TT, NT, J, XP, WP);
    const int nTT = (a.Tout + TT - 1) / TT, nNT = (a.N + NT - 1) / NT;
    const size_t lds = conv_lds_bytes(a, variant);
    auto kern = conv_mfma_kernel<MT, NW, WT, WN, CK>;
    static size_t lds_allowed = 64 * 1024;
    if (lds > lds_allowed) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_allowed = lds;
    }
    const long long grid = (long long)nTT * nNT * a.B;
    if (grid <= 0) return hipSuccess;
    char nm[64];
    snprintf(nm, sizeof(nm), "conv_mfma_kernel<%d, %d, %d, %d, %d>", MT, NW, WT, WN, CK);
    ProfScope ps(nm, conv_flops(a), s);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, a, nTT, nNT, J, XP, WP);
    return hipGetLastError();
}

hipError_t launch_conv(const ConvArgs& a, hipStream_t s) {
    const int v = conv_pick_variant(a);
    switch (v) {
        case 0: return conv_launch_t<4, 2, 4, 1, 8>(a, v, s);
        case 1: return conv_launch_t<4, 3, 4, 1, 8>(a, v, s);
        case 2: return conv_launch_t<4, 4, 4, 1, 8>(a, v, s);
        case 3: return conv_launch_t<4, 5, 4, 1, 8>(a, v, s);
        case 4: return conv_launch_t<4, 6, 4, 1, 8>(a, v, s);
        case 5: return conv_launch_t<2, 3, 2, 2, 8>(a, v, s);
        case 6: return conv_launch_t<1, 2, 1, 4, 8>(a, v, s);
        case 7: return conv_launch_t<4, 2, 4, 1, 4>(a, v, s);
        default: return conv_launch_t<1, 3, 2, 2, 8>(a, v, s);
    }
}

// =====================================================================================
// weight / bias gradient
// =====================================================================================
template <int MTW, int NW>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgradArgs a, int nMG, int nNG, int TK,
                                                         int XP, int ZP, int nChMax, int ONESP) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int MG = 4 * MTW * 16;
    constexpr int NG = NW * 16;
    const bool deint = (a.loader == LOADER_DEINT);
    const int planes = deint ? 2 : 1;
    const int Jx = deint ? (a.KW + 1) / 2 : a.KW;      // halo width per plane
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
            off = ONESP + (c - cLo) * planes * XP + (deint ? ((k & 1) * XP + (k >> 1)) : k);
        }
        rowoff[mt] = off;
    }
    for (int i = tid; i < ONESP; i += 256) lds[i] = 1.f;

    f32x4 acc[MTW][NW];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nunits = a.B * a.nQT;
    int u1 = (split + 1) * a.units_per_split;
    if (u1 > nunits) u1 = nunits;
    for (int u = split * a.units_per_split; u < u1; ++u) {
        const int b = u / a.nQT, qt = u % a.nQT;
        const int q0 = qt * TK;
        int nq = a.Tq - q0; if (nq > TK) nq = TK;
        const int nq4 = (nq + 3) & ~3;
        __syncthreads();
        // stage X rows (one wave per channel row, lanes along time)
        for (int ci = wave; ci < nCh; ci += 4) {
            const int c = cLo + ci;
            const float* p = (c < a.C0)
                ? a.src0 + (long long)b * a.bs0 + (long long)c * a.pitch0 + a.off0
                : a.src1 + (long long)b * a.bs1 + (long long)(c - a.C0) * a.pitch1 + a.off1;
            if (!deint) {
                const int tbase = q0 - a.shift;
                const int width = nq4 + a.KW - 1;
                for (int x = lane; x < width; x += 64) {
                    const int t = tbase + x;
                    Xs[ci * XP + x] = (t >= 0 && t < a.Tin) ? p[t] : 0.f;
                }
            } else {
                const int tbase = 2 * q0 - a.shift;
                const int width = 2 * (nq4 + Jx - 1);
                for (int e = lane; e < width; e += 64) {
                    const int t = tbase + e;
                    Xs[(ci * 2 + (e & 1)) * XP + (e >> 1)] = (t >= 0 && t < a.Tin) ? p[t] : 0.f;
                }
            }
        }
        // stage dZ rows
        for (int n = wave; n < NG; n += 4) {
            const int nn = ng * NG + n;
            const float* p = a.dz + (long long)b * a.dzbs + (long long)nn * a.dzpitch + q0;
            for (int q = lane; q < nq4; q += 64)
                Zs[n * ZP + q] = (nn < a.N && q < nq) ? p[q] : 0.f;
        }
        __syncthreads();
        const int nsteps = nq4 >> 2;
        for (int s = 0; s < nsteps; ++s) {
            float av[MTW], bv[NW];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) av[mt] = lds[rowoff[mt] + 4 * s + lg];
#pragma unroll
            for (int n = 0; n < NW; ++n) bv[n] = Zs[(n * 16 + li) * ZP + 4 * s + lg];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                if (mt < nact) {
#pragma unroll
                    for (int n = 0; n < NW; ++n) acc[mt][n] = mfma16(av[mt], bv[n], acc[mt][n]);
                }
            }
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

struct WgradGeom { int MTW, NW, nMG, nNG, TK, XP, ZP, nChMax, ONESP; size_t lds; };

static WgradGeom wgrad_geom(const WgradArgs& a) {
    WgradGeom g;
    const int Ctot = a.C0 + a.C1;
    const int mtiles = (Ctot * a.KW + 1 + 15) / 16;
    g.MTW = mtiles <= 4 ? 1 : (mtiles <= 8 ? 2 : 6);
    int bestnw = 3, bestpad = 1 << 30;
    for (int nw = 3; nw >= 1; --nw) {
        const int padded = ((a.N + nw * 16 - 1) / (nw * 16)) * nw * 16;
        if (padded < bestpad) { bestpad = padded; bestnw = nw; }
    }
    g.NW = bestnw;
    const int MG = 4 * g.MTW * 16, NG = g.NW * 16;
    g.nMG = (Ctot * a.KW + 1 + MG - 1) / MG;
    g.nNG = (a.N + NG - 1) / NG;
    int tk = (a.Tq + 3) & ~3;
    if (tk > 128) tk = 128;
    if (tk < 4) tk = 4;
    g.TK = tk;
    const bool deint = a.loader == LOADER_DEINT;
    const int Jx = deint ? (a.KW + 1) / 2 : a.KW;
    const int mod = deint ? 10 : (a.KW >= 9 ? 16 : (a.KW >= 5 ? 8 : (a.KW >= 3 ? 4 : 2)));
    g.XP = fit_pitch(tk + Jx - 1, mod);
    g.ZP = fit_pitch(tk, 2);
    int nch = (MG + a.KW - 2) / a.KW + 1;
    if (nch > Ctot) nch = Ctot;
    g.nChMax = nch;
    g.ONESP = (tk + 15) & ~15;
    g.lds = sizeof(float) * ((size_t)g.ONESP + (size_t)nch * (deint ? 2 : 1) * g.XP + (size_t)NG * g.ZP);
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
                       g.ZP, g.nChMax, g.ONESP);
    return hipGetLastError();
}

hipError_t launch_wgrad(const WgradArgs& a, hipStream_t s) {
    const WgradGeom g = wgrad_geom(a);
#define WUN_WG(M, N) if (g.MTW == M && g.NW == N) return wgrad_launch_t<M, N>(a, g, s);
    WUN_WG(1, 1) WUN_WG(1, 2) WUN_WG(1, 3)
    WUN_WG(2, 1) WUN_WG(2, 2) WUN_WG(2, 3)
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

hipError_t launch_reduce(const float* partial, long long stride, int nsplit, float* out,
                         long long n, hipStream_t s) {
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
