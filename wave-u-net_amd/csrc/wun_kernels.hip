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
#include <algorithm>
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
// Every bracket {e0, launch, e1} is followed by an EMPTY bracket {c0, c1} on the same stream: an event pair costs ~5 us of
// packet processing that rocprofv3's kernel durations do not contain; prof_end() reports the median empty bracket so the
// caller can state durations net of the measuring method's own cost (bench.py: roofline.event_bracket_overhead_us).
struct ProfSlot { std::string name; std::string tag; double flops; double bytes; hipEvent_t e0, e1, c0, c1; };
static std::vector<ProfSlot> g_prof;
static bool g_prof_on = false;
static std::mutex g_prof_mu;

void prof_begin() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& sl : g_prof) { (void)hipEventDestroy(sl.e0); (void)hipEventDestroy(sl.e1); (void)hipEventDestroy(sl.c0); (void)hipEventDestroy(sl.c1); }
    g_prof.clear();
    g_prof_on = true;
}

struct ProfScope {
    bool on; hipStream_t s; size_t idx;
    // bytes: ALGORITHMIC HBM bytes of a bandwidth-bound launch (what a perfect implementation must move), 0 for MFMA kernels
    ProfScope(const char* name, double flops, hipStream_t st, const char* tag = "", double bytes = 0.0) : on(g_prof_on), s(st), idx(0) {
        if (!on) return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        ProfSlot sl; sl.name = name; sl.tag = tag; sl.flops = flops; sl.bytes = bytes;
        if (hipEventCreate(&sl.e0) != hipSuccess || hipEventCreate(&sl.e1) != hipSuccess ||
            hipEventCreate(&sl.c0) != hipSuccess || hipEventCreate(&sl.c1) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(sl.e0, s);
        g_prof.push_back(sl);
        idx = g_prof.size() - 1;
    }
    ~ProfScope() {
        if (!on) return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        (void)hipEventRecord(g_prof[idx].e1, s);
        (void)hipEventRecord(g_prof[idx].c0, s);
        (void)hipEventRecord(g_prof[idx].c1, s);
    }
};

// explicit begin / end form for launchers in other translation units (one bracket at a time)
static size_t g_prof_open = (size_t)-1;
void prof_scope_begin(const char* name, double flops, hipStream_t s, const char* tag, double bytes) {
    g_prof_open = (size_t)-1;
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfSlot sl; sl.name = name; sl.tag = tag; sl.flops = flops; sl.bytes = bytes;
    if (hipEventCreate(&sl.e0) != hipSuccess || hipEventCreate(&sl.e1) != hipSuccess ||
        hipEventCreate(&sl.c0) != hipSuccess || hipEventCreate(&sl.c1) != hipSuccess) return;
    (void)hipEventRecord(sl.e0, s);
    g_prof.push_back(sl);
    g_prof_open = g_prof.size() - 1;
}
void prof_scope_end(hipStream_t s) {
    if (g_prof_open == (size_t)-1) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_prof[g_prof_open].e1, s);
    (void)hipEventRecord(g_prof[g_prof_open].c0, s);
    (void)hipEventRecord(g_prof[g_prof_open].c1, s);
    g_prof_open = (size_t)-1;
}

// JSON: {"kernels": [{"name":..., "launches": n, "ms": total, "flops": total}, ...]}
std::string prof_end() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = false;
    struct Agg { long n = 0; double ms = 0, flops = 0, bytes = 0; };
    std::map<std::string, Agg> agg;
    std::string detail;
    std::vector<float> empty_ms;
    for (auto& sl : g_prof) {
        (void)hipEventSynchronize(sl.c1);
        float ms = 0.f, cms = 0.f;
        if (hipEventElapsedTime(&cms, sl.c0, sl.c1) == hipSuccess) empty_ms.push_back(cms);
        if (hipEventElapsedTime(&ms, sl.e0, sl.e1) == hipSuccess) {
            Agg& a = agg[sl.name]; a.n += 1; a.ms += ms; a.flops += sl.flops; a.bytes += sl.bytes;
            if (getenv("WUN_PROFILE_DETAIL") != nullptr) {
                char buf[512];
                snprintf(buf, sizeof(buf), "%s{\"name\": \"%s\", \"tag\": \"%s\", \"ms\": %.6f, \"flops\": %.6e}",
                         detail.empty() ? "" : ", ", sl.name.c_str(), sl.tag.c_str(), ms, sl.flops);
                detail += buf;
            }
        }
        (void)hipEventDestroy(sl.e0); (void)hipEventDestroy(sl.e1); (void)hipEventDestroy(sl.c0); (void)hipEventDestroy(sl.c1);
    }
    g_prof.clear();
    double overhead_ms = 0.0;
    if (!empty_ms.empty()) {
        std::sort(empty_ms.begin(), empty_ms.end());
        overhead_ms = empty_ms[empty_ms.size() / 2];
    }
    char obuf[96];
    snprintf(obuf, sizeof(obuf), "{\"bracket_overhead_ms\": %.6f, ", overhead_ms);
    std::string out = std::string(obuf) + "\"launches\": [" + detail + "], \"kernels\": [";
    bool first = true;
    for (auto& kv : agg) {
        char buf[512];
        snprintf(buf, sizeof(buf), "%s{\"name\": \"%s\", \"launches\": %ld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}",
                 first ? "" : ", ", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops, kv.second.bytes);
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
// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  Remap the hardware block
// id so that every XCD works on a CONTIGUOUS range of logical tiles: the tiles that share an input
// window (the N tiles of one time tile, neighbouring time tiles' halos; for the weight gradient all
// (row group, column group) tiles of one split) then hit the same L2 instead of fetching the window
// once per XCD.
__device__ __forceinline__ int xcd_contiguous_block(int bid, int grid) {
    const int per = grid >> 3, rem = grid & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
}

#define WUN_JMAX 15


// floor(n / d) for 0 <= n < 2^22, d >= 1 with inv = 1.0f / d: one multiply + a +/-1 fix-up instead of the ~35
// instruction integer division (the batch-folded tiles of the deep, launch-latency-bound levels do a dozen of
// these per thread before their first load)
__device__ __forceinline__ int fast_div(int n, int d, float inv) {
    int q = (int)(((float)n + 0.5f) * inv);
    const int r = n - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

//
// FOLD variants (deep levels, few output positions per excerpt): the GEMM M axis is the flattened
// (excerpt, position) index, so one workgroup tile spans several excerpts and every weight slab
// is fetched once per tile instead of once per excerpt.  The LDS input row then holds one
// full-length segment per excerpt of the tile; only the staging addresses, the per-lane A-operand
// offsets and the epilogue's (excerpt, position) decode differ from the plain kernel.
#define WUN_FOLD_XCAP(TT) (3 * (TT) + 32)      // LDS input-row capacity of a FOLD tile (floats per plane)

// XVEC ("DMA" instantiations, stride-1 loader): the input window and the weight slab of a chunk go global -> LDS
// DIRECTLY (global_load_lds, 16 bytes per lane, one contiguous KiB of LDS per wave instruction): no staging
// registers, no LDS store instructions, no per-chunk mask / address VALU -- per chunk a wave issues a handful of
// DMA instructions right after the barrier and then runs ONE uninterrupted tap loop.  The window starts at the
// aligned element below its first sample (rows of the plan's buffers are 16-byte aligned) and the sub-vector shift
// (launch-constant per source: (off - shift) mod 4) is folded into the A-operand LDS offset.  Samples outside
// [0, Tin) ('same' padding, the halo of a transposed conv) are zeroed in LDS after the DMA has landed, by the edge
// tiles only.  The element-wise register path stays for the stride-2 loader, FOLD tiles, channel tails and
// unaligned test tensors.
// copies of a dst0 output element / quad beside its main store (ConvArgs.dec / dec_exp / dec1): b = excerpt, n = dst0 channel
__device__ __forceinline__ void conv_copy1(const ConvArgs& a, int b, int n, int q, float v) {
    if (a.dec != nullptr) {
        float* row = a.dec + (long long)b * a.decbs + (long long)n * a.decpitch + a.dec_off;
        if (a.dec_exp) {
            const unsigned pos = (unsigned)(2 * q - a.dec_lo);
            if (pos < a.dec_len) row[pos] = v;
        } else if ((q & 1) == 0) {
            row[q >> 1] = v;
        }
    }
    if (a.dec1 != nullptr && (q & 1) != 0) a.dec1[(long long)b * a.dec1bs + (long long)n * a.dec1pitch + a.dec1_off + (q >> 1)] = v;
}
__device__ __forceinline__ void conv_copy4(const ConvArgs& a, int b, int n, int q, f32x4 v) {       // q % 4 == 0, all four valid
    if (a.dec != nullptr) {
        float* row = a.dec + (long long)b * a.decbs + (long long)n * a.decpitch + a.dec_off;
        if (a.dec_exp) {
            const int pos0 = 2 * q - a.dec_lo;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if ((unsigned)(pos0 + 2 * r) < a.dec_len) row[pos0 + 2 * r] = v[r];
        } else {
            row[q >> 1] = v[0];
            row[(q >> 1) + 1] = v[2];
        }
    }
    if (a.dec1 != nullptr) {
        float* row = a.dec1 + (long long)b * a.dec1bs + (long long)n * a.dec1pitch + a.dec1_off;
        row[q >> 1] = v[1];
        row[(q >> 1) + 1] = v[3];
    }
}
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;
// KG > 1 ("in-workgroup split-K", the deep levels): the workgroup is KG groups of 256 threads; group kg runs the whole
// kernel body on its own LDS region for the channel chunks of sub-split ksg * KG + kg of the SAME output tile, the groups'
// accumulators are summed through LDS in the fixed order ((g0 + g1) + g2) and group 0 runs the epilogue -- a launch with
// few output tiles gets KG x the waves per tile without partial tiles in HBM or an epilogue launch on the dependent chain.
// Every group runs the same number of chunks (launcher rule: chunks % (KG * splits) == 0), so the barriers stay uniform.
template <int MT, int NW, int WT, int WN, int CK, bool VECW, bool FOLD = false, bool XVEC = false, int KG = 1>
__global__ __launch_bounds__(256 * KG) void conv_mfma_kernel(ConvArgs a, int nTT, int nNT, int J,
                                                             int XP, int WP) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int kg = KG > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;
    constexpr int TT = WT * MT * 16;
    constexpr int NT = WN * NW * 16;
    constexpr int NT4 = NT / 4;
    constexpr int CH = CK / 2;
    // staging trip counts (compile time, sized for J <= WUN_JMAX)
    constexpr int TPR = 256 / CK;                                   // DIRECT: threads per LDS row
    constexpr int XIT_D = FOLD ? (WUN_FOLD_XCAP(TT) + TPR - 1) / TPR : (TT + WUN_JMAX - 1 + TPR - 1) / TPR;
    constexpr int TPC = 256 / CH;                                   // DEINT: threads per channel
    constexpr int XIT_I = FOLD ? (2 * WUN_FOLD_XCAP(TT) + TPC - 1) / TPC
                               : (2 * (TT + (WUN_JMAX + 1) / 2 - 1) + TPC - 1) / TPC;
    constexpr int XIT = XVEC ? 1 : (XIT_D > XIT_I ? XIT_D : XIT_I);
    constexpr int WIT = XVEC ? 1 : (WUN_JMAX * CK * NT4 + 255) / 256;
    // XVEC: 16-byte granules per thread -- X: CK rows of at most (TT + 48) / 4 granules (the LDS pitch), W: J * CK rows of
    // at most NT / 4 + 4 granules
    constexpr int XDIT = XVEC ? (CK * ((TT + 48) / 4) + 255) / 256 : 1;
    constexpr int WDIT = XVEC ? (WUN_JMAX * CK * (NT4 + 4) + 255) / 256 : 1;
    static_assert(!(XVEC && FOLD), "DMA staging is not implemented for batch-folded tiles");

    // two LDS buffers {input window, weight slab}: chunk c+1 is written while chunk c is read
    const int XB = CK * XP, LB = CK * XP + J * CK * WP;      // floats per X tile / per buffer
    float* lds = lds_all + (KG > 1 ? kg * 2 * LB : 0);
    float* Xs = lds;
    float* Ws = lds + XB;

    const int tid = KG > 1 ? (int)(threadIdx.x & 255) : (int)threadIdx.x;
    const int lane = tid & 63;
    const int li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // raised wave priority for prologue / staging / epilogue code (a workgroup that starts beside 3-4 MFMA-bound older waves
    // otherwise gets only their left-over issue slots): same-box A/B 8.90 -> 8.87 ms per step
#define WUN_PRIO_HI() __builtin_amdgcn_s_setprio(2)
#define WUN_PRIO_LO() __builtin_amdgcn_s_setprio(0)
    WUN_PRIO_HI();
    int bid = xcd_contiguous_block((int)blockIdx.x, (int)gridDim.x);
    const int nt = bid % nNT; bid /= nNT;
    const int tt = bid % nTT; bid /= nTT;
    const float inv_tout = 1.0f / (float)a.Tout;
    const int b = FOLD ? fast_div(tt * TT, a.Tout, inv_tout) : bid % a.B;      // FOLD: first excerpt of the tile
    const int ksg = FOLD ? bid : bid / a.B;                   // split whose partial tile this workgroup produces
    const int ksp = ksg * KG + kg;                            // sub-split of this group of 256 threads
    const int q0 = tt * TT, n0 = nt * NT;                     // FOLD: q0 = first flattened row
    // FOLD: excerpts touched by this tile and the per-excerpt segment length in the LDS row
    int fold_nb = 1;
    const int fold_seg = a.Tout + J - 1;
    if constexpr (FOLD) {
        int last = fast_div(q0 + TT - 1, a.Tout, inv_tout);
        if (last > a.B - 1) last = a.B - 1;
        fold_nb = last - b + 1;
    }
    const int wt0 = (wave % WT) * MT * 16;
    const int wn0 = (wave / WT) * NW * 16;
    const int Ctot = a.C0 + a.C1;
    const bool deint = (a.loader == LOADER_DEINT);
    const bool phase2 = (a.flags & F_PHASE2) != 0;      // only launched on WN == 1, even NW variants
    const int n0h = nt * (NT / 2);
    const int UW = FOLD ? fold_nb * fold_seg : TT + J - 1;
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
    // XVEC state: sub-vector shift of each source, aligned element index of the window start in a source row, granules
    // per LDS row that are fetched, and per granule of this thread its global element offset (chunk-invariant; the
    // chunk only moves a uniform base) or -1 for a lane that fetches nothing
    const int dl0 = XVEC ? (((a.off0 - a.shift) % 4) + 4) % 4 : 0;
    const int dl1 = (XVEC && a.C1 > 0) ? (((a.off1 - a.shift) % 4) + 4) % 4 : dl0;
    int xdo[XDIT], wdo[WDIT];                                            // xdo: row << 20 | 4 * granule, or -1 (lane fetches nothing)
    int nvr = 1, e00 = 0, e01 = 0;
    bool xedge = false;
    const float* vb0 = a.src0 + (long long)b * a.bs0;                 // row bases WITHOUT the crop offset (aligned)
    const float* vb1 = (a.src1 != nullptr) ? a.src1 + (long long)b * a.bs1 : vb0;
    const int XG = XP >> 2, WG4 = WP >> 2;                               // granules per LDS row
    if constexpr (XVEC) {
        const int tbase = q0 - a.shift;
        nvr = (UW + (dl0 > dl1 ? dl0 : dl1) + 3) >> 2;
        e00 = a.off0 + tbase - dl0;
        e01 = a.off1 + tbase - dl1;
        const int tf0 = tbase - dl0, tf1 = tbase - dl1;                  // time of the first staged element
        xedge = tf0 < 0 || tf0 + 4 * nvr > a.Tin || (a.C1 > 0 && (tf1 < 0 || tf1 + 4 * nvr > a.Tin));
        const float inv_xg = 1.0f / (float)XG;
#pragma unroll
        for (int i = 0; i < XDIT; ++i) {
            const int f = tid + i * 256;
            const int row = fast_div(f, XG, inv_xg), g = f - row * XG;
            xdo[i] = (row < CK && g < nvr) ? (row << 20) | (4 * g) : -1;
        }
        const float inv_wg = 1.0f / (float)WG4;
#pragma unroll
        for (int i = 0; i < WDIT; ++i) {
            const int f = tid + i * 256;
            const int row = fast_div(f, WG4, inv_wg), c4 = f - row * WG4;
            const int j = row / CK, r = row - j * CK;
            const bool ok = j < J && c4 < NT4;
            int wo;
            if (!phase2) {
                int col = n0 + c4 * 4;
                col = col > a.N - 4 ? a.N - 4 : col;                     // padded columns: any valid (finite) weights
                wo = (j * Ctot + r) * a.N + col;
            } else {
                const int x = c4 * 4, ph = x / (NT / 2), cl = x % (NT / 2);
                int col = n0h + cl;
                col = col > a.N - 4 ? a.N - 4 : col;
                wo = (j * Ctot + r) * (2 * a.N) + ph * a.N + col;
            }
            wdo[i] = ok ? wo : -1;
        }
    }
    if constexpr (!XVEC) {
        const int lr = deint ? tid % TPC : tid % TPR;
        const int stride_i = deint ? TPC : TPR;
        const int lim = deint ? 2 * UW : UW;
        const int tbase = (deint ? 2 * q0 : q0) - a.shift;
        const int seglen = deint ? 2 * fold_seg : fold_seg;
        const float inv_seglen = 1.0f / (float)seglen;
#pragma unroll
        for (int i = 0; i < XIT; ++i) {
            const int u = lr + i * stride_i;
            int t = tbase + u, bl = 0;
            if constexpr (FOLD) { bl = fast_div(u, seglen, inv_seglen); t = u - bl * seglen - a.shift; }
            const bool ok = u < lim && t >= 0 && t < a.Tin;
            int tc = t < 0 ? 0 : t;
            if (tc > a.Tin - 1) tc = a.Tin - 1;
            if constexpr (FOLD) {
                if (bl > fold_nb - 1) bl = fold_nb - 1;
                tc |= bl << 20;                       // excerpt within the tile (Tin < 2^20, checked by the launcher)
            }
            xt[i] = tc;
            xmask_s |= (ok ? 1u : 0u) << i;
        }
    }
    int wofs[WIT];            // element offset of this thread's i-th weight vector for channel chunk 0
    int wr[WIT];              // channel index within the chunk (for the channel-tail mask)
    const int nwvec = J * CK * NT4;                       // weight vectors per chunk (uniform)
#pragma unroll
    for (int i = 0; i < (XVEC ? 0 : WIT); ++i) {
        const int f = tid + i * 256;
        const int row = f / NT4, c4 = f % NT4;
        const int j = row / CK, r = row % CK;
        int k, cch;
        if (!deint) { k = j; cch = r; }
        else { k = 2 * j + r / CH; cch = r % CH; }
        bool ok;
        int wo;
        if (!phase2) {
            ok = f < nwvec && k < a.KW && n0 + c4 * 4 < a.N;
            wo = (k * Ctot + cch) * a.N + n0 + c4 * 4;
        } else {
            // fused output phases: LDS columns [0, NT/2) hold phase-0 weights of channels
            // n0h.., columns [NT/2, NT) the phase-1 weights of the SAME channels; global rows
            // are [2][N] (make_wt mode 1)
            const int x = c4 * 4, ph = x / (NT / 2), cl = x % (NT / 2);
            ok = f < nwvec && k < a.KW && n0h + cl < a.N;
            wo = (k * Ctot + cch) * (2 * a.N) + ph * a.N + n0h + cl;
        }
        wofs[i] = ok ? wo : 0;
        wr[i] = cch;
        wmask_s |= (ok ? 1u : 0u) << i;
    }

    // ---- global -> registers for one chunk.  Loads are unconditional (clamped addresses);
    // the zero fill is applied when the registers are written to LDS, so nothing here
    // waits for the loads and they stay in flight during the MFMAs of the previous chunk ----
    auto load_chunk = [&](int chunk) {
        const int c0 = chunk * CKC;
        if constexpr (XVEC) {
            return;                                        // (DMA instantiations stage through dma_chunk)
        } else {
            const int c = c0 + xrow;
            const bool cok = c < Ctot;
            const float* p = (!cok || c < a.C0) ? src0b + (long long)(cok ? c : 0) * a.pitch0
                                                : src1b + (long long)(c - a.C0) * a.pitch1;
            const int nx = deint ? XIT_I : XIT_D;
            const int bsel = (!cok || c < a.C0) ? (int)a.bs0 : (int)a.bs1;
#pragma unroll
            for (int i = 0; i < XIT; ++i)
                if (i < nx) {
                    if constexpr (FOLD) xreg[i] = p[(xt[i] >> 20) * bsel + (xt[i] & 0xFFFFF)];
                    else xreg[i] = p[xt[i]];
                }
            xmask = cok ? xmask_s : 0u;
        }
        const bool tail = c0 + CKC > Ctot;                // uniform; only the last chunk of odd configs
        const float* wc = a.W + (long long)c0 * (phase2 ? 2 * a.N : a.N);   // uniform base of this chunk's weight rows
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
    auto store_chunk = [&](int bufoff, int chunk) {
        if constexpr (XVEC) {
            (void)chunk;
            return;
        } else if (!deint) {
            const int lr = tid % TPR;
            float* xd = Xs + bufoff + xrow * XP + lr;
#pragma unroll
            for (int i = 0; i < XIT_D; ++i)
                if (lr + i * TPR < UW) xd[i * TPR] = ((xmask >> i) & 1u) ? xreg[i] : 0.f;
        } else {
            const int le = tid % TPC;
            float* xd = Xs + bufoff + ((le & 1) * CH + xrow) * XP + (le >> 1);
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
                    *reinterpret_cast<f32x4*>(&Ws[bufoff + row * WP + c4 * 4]) =
                        ((wmask >> i) & 1u) ? wreg[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    // MFMA loop over the taps [j_begin, j_end) of the chunk in LDS buffer `bufoff`.  A k-step is
    // 4 LDS rows (4 input channels) of one tap: KS = CK/4 k-steps per tap.  Operands of the
    // next k-step are read from LDS right after the first MFMA of the current one
    // (double-buffered registers, pointer-increment addressing), so LDS latency is covered by
    // the remaining MFMAs.
    constexpr int KS = CK / 4;
    // FOLD: LDS position (segment of its excerpt + output position) of this lane's A row per M tile
    int arow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        arow[m] = 0;
        if constexpr (FOLD) {
            const int row = q0 + wt0 + m * 16 + li;
            int g = fast_div(row, a.Tout, inv_tout);
            const int q = row - g * a.Tout;
            g -= b;
            arow[m] = g < fold_nb ? g * fold_seg + q : 0;      // rows past the last excerpt: results discarded
        }
    }
    const bool skip_last_odd = deint && CK == 8 && (a.KW & 1);     // the odd phase has no tap KW
    auto run_taps = [&](int bufoff, int j_begin, int j_end, int dlt) {
        if (j_end <= j_begin) return;
        // (XVEC: dlt = sub-vector shift of the chunk's source; stride-2 planes hold it as dlt>>1 / (dlt+1)>>1)
        const float* xa = Xs + bufoff + lg * XP + (FOLD ? 0 : wt0 + li) + j_begin + (deint ? (dlt >> 1) : dlt);   // tap j, rows 0..3
        const int x2off = 4 * XP + (deint ? (dlt & 1) : 0);                          // rows 4..7 of the same tap
        (void)x2off;
        const float* wb = Ws + bufoff + (j_begin * CK + lg) * WP + wn0 + li;
        const int wstep = CK * WP;
        float a0[MT], b0[NW], a1[MT], b1[NW];
        auto ldop = [&](const float* x, const float* w, float (&av)[MT], float (&bv)[NW]) {
#pragma unroll
            for (int m = 0; m < MT; ++m) av[m] = FOLD ? x[arow[m]] : x[m * 16];
#pragma unroll
            for (int n = 0; n < NW; ++n) bv[n] = w[n * 16];
        };
        auto mm_first = [&](const float (&av)[MT], const float (&bv)[NW]) {
            acc[0][0] = mfma16(av[0], bv[0], acc[0][0]);
        };
        auto mm_rest = [&](const float (&av)[MT], const float (&bv)[NW]) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NW; ++n)
                    if (m + n > 0) acc[m][n] = mfma16(av[m], bv[n], acc[m][n]);
        };
        // pin the per-k-step interleave: 1 MFMA, the LDS reads of the next operands, the rest
        auto pin = [&]() {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, MT + NW, 0);        // DS reads (<= MT+NW instrs)
            __builtin_amdgcn_sched_group_barrier(0x008, MT * NW - 1, 0);    // MFMA
        };
        ldop(xa, wb, a0, b0);
        if constexpr (KS == 2) {
            // the stride-2 loader has no odd-phase rows for its last tap when KW is odd: that
            // half k-step is peeled off so the loop body stays branch-free
            const int j_full = (skip_last_odd && j_end == J) ? j_end - 1 : j_end;
            for (int j = j_begin; j < j_full; ++j) {
                const bool more = j + 1 < j_end;
                mm_first(a0, b0);
                ldop(xa + x2off, wb + 4 * WP, a1, b1);                  // rows 4..7 of tap j
                mm_rest(a0, b0);
                pin();
                xa += more ? 1 : 0;                                     // next tap (clamped at the end)
                wb += more ? wstep : 0;
                mm_first(a1, b1);
                ldop(xa, wb, a0, b0);
                mm_rest(a1, b1);
                pin();
            }
            if (j_full < j_end) {
                mm_first(a0, b0);
                mm_rest(a0, b0);
            }
        } else {
            // one k-step per tap: two taps per iteration, odd tail peeled
            const int npair = (j_end - j_begin) / 2;
            int j = j_begin;
            for (int it = 0; it < npair; ++it, j += 2) {
                const bool more2 = j + 2 < j_end;
                mm_first(a0, b0);
                ldop(xa + 1, wb + wstep, a1, b1);
                mm_rest(a0, b0);
                pin();
                xa += more2 ? 2 : 1;
                wb += more2 ? 2 * wstep : wstep;
                mm_first(a1, b1);
                ldop(xa, wb, a0, b0);
                mm_rest(a1, b1);
                pin();
            }
            if (j < j_end) {
                mm_first(a0, b0);
                mm_rest(a0, b0);
            }
        }
    };

    // ---- XVEC: chunk -> LDS buffer by DMA.  Wave w issues the KiB blocks {w, w+4, ...} of the X region and of the W
    // region; every lane supplies the global address of its 16-byte granule (or sits the instruction out) ----
    auto dma_chunk = [&](int chunk, int bufoff) __attribute__((always_inline)) {
        if constexpr (XVEC) {
            const int c0 = chunk * CKC;
            const bool s1 = c0 >= a.C0;                                  // (a chunk lies in ONE source: launcher rule)
            const float* xb = s1 ? vb1 + (long long)(c0 - a.C0) * a.pitch1 : vb0 + (long long)c0 * a.pitch0;
            const int pitch = s1 ? a.pitch1 : a.pitch0, e0s = s1 ? e01 : e00;
            const int nxg = CK * XG;
#pragma unroll
            for (int i = 0; i < XDIT; ++i) {
                if (i * 256 + wave * 64 < nxg) {                         // wave-uniform
                    const int pk = xdo[i];
                    int e = e0s + (pk & 0xFFFFF);
                    e = e < 0 ? 0 : (e > pitch - 4 ? pitch - 4 : e);     // clamped into the row; zeroed later if outside [0, Tin)
                    if (pk >= 0)
                        __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(xb + (pk >> 20) * pitch + e),
                                                         (lds_void_t*)(Xs + bufoff + (i * 256 + wave * 64) * 4), 16, 0, 0);
                }
            }
            const float* wc = a.W + (long long)c0 * (phase2 ? 2 * a.N : a.N);
            const int nwg = J * CK * WG4;
#pragma unroll
            for (int i = 0; i < WDIT; ++i) {
                if (i * 256 + wave * 64 < nwg) {
                    const int o = wdo[i];
                    if (o >= 0)
                        __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(wc + o),
                                                         (lds_void_t*)(Ws + bufoff + (i * 256 + wave * 64) * 4), 16, 0, 0);
                }
            }
        }
    };
    // edge tiles: zero the staged samples whose time index lies outside [0, Tin)
    auto zero_fix = [&](int chunk, int bufoff) __attribute__((always_inline)) {
        if constexpr (XVEC) {
            const bool s1 = chunk * CKC >= a.C0;
            const int t00 = (s1 ? e01 - a.off1 : e00 - a.off0);
            const float inv_xg = 1.0f / (float)XG;
#pragma unroll
            for (int i = 0; i < XDIT; ++i) {
                const int f = tid + i * 256;
                const int row = fast_div(f, XG, inv_xg), g = f - row * XG;
                if (row < CK && g < nvr) {
                    const int t0 = t00 + 4 * g;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (t0 + k < 0 || t0 + k >= a.Tin) Xs[bufoff + row * XP + 4 * g + k] = 0.f;
                }
            }
        }
    };

    if constexpr (XVEC) {
        if (ch_lo < ch_hi) dma_chunk(ch_lo, 0);
        for (int chunk = ch_lo; chunk < ch_hi; ++chunk) {
            const int cur = ((chunk - ch_lo) & 1) * LB;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's granules of `chunk` have landed
            __syncthreads();                                            // ... everybody's have; buffer LB - cur is free
            if (chunk == ch_lo) WUN_PRIO_LO();
            if (xedge) {
                zero_fix(chunk, cur);
                __syncthreads();
            }
            if (chunk + 1 < ch_hi) dma_chunk(chunk + 1, LB - cur);
            run_taps(cur, 0, J, (chunk * CKC < a.C0) ? dl0 : dl1);
        }
    } else {
    // Pipeline: global loads of chunk c+1 are issued before the MFMAs of chunk c; their LDS
    // writes (to the other buffer) sit in the middle of chunk c's MFMA loop, so they issue in
    // the shadow of the matrix pipe; one barrier per chunk.
    if (ch_lo < ch_hi) load_chunk(ch_lo);
    if (ch_lo < ch_hi) store_chunk(0, ch_lo);
    __syncthreads();
    WUN_PRIO_LO();
    const int half = (J * 2) / 4;    // the next chunk is written to LDS half-way through the tap loop
    for (int chunk = ch_lo; chunk < ch_hi; ++chunk) {
        const int cur = ((chunk - ch_lo) & 1) * LB;
        const bool has_next = chunk + 1 < ch_hi;
        WUN_PRIO_HI();
        if (has_next) load_chunk(chunk + 1);
        WUN_PRIO_LO();
        run_taps(cur, 0, half, 0);
        WUN_PRIO_HI();
        if (has_next) store_chunk(LB - cur, chunk + 1);
        WUN_PRIO_LO();
        run_taps(cur, half, J, 0);
        __syncthreads();
    }
    }
    WUN_PRIO_HI();

    if constexpr (KG > 1) {
        // ---- in-workgroup split-K: groups 1 .. KG-1 hand their accumulators over through LDS (the staging buffers are
        // dead: barrier first), group 0 adds them in group order and carries on alone ----
        __syncthreads();
        float* red = lds_all;                                  // [KG - 1][MT * NW][256 threads] x 16 bytes
        if (kg > 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NW; ++n)
                    *reinterpret_cast<f32x4*>(&red[((((kg - 1) * MT + m) * NW + n) * 256 + tid) * 4]) = acc[m][n];
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int g = 1; g < KG; ++g)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NW; ++n)
                    acc[m][n] += *reinterpret_cast<const f32x4*>(&red[((((g - 1) * MT + m) * NW + n) * 256 + tid) * 4]);
    }

    // ---- split-K: raw partial tile, epilogue runs in conv_splitk_epilogue_kernel ----
    if (a.part != nullptr) {
        const int TP = (a.Tout + 3) & ~3;
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            const int ncol = n0 + wn0 + n * 16 + li;
            if (ncol >= a.N) continue;
            float* prow = a.part + (((long long)ksg * a.B + b) * a.N + ncol) * TP;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int q = q0 + wt0 + m * 16 + lg * 4;
                if constexpr (FOLD) {
                    int g = fast_div(q, a.Tout, inv_tout), qq = q - g * a.Tout;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (g < a.B) a.part[(((long long)ksg * a.B + g) * a.N + ncol) * TP + qq] = acc[m][n][r];
                        if (++qq == a.Tout) { qq = 0; ++g; }
                    }
                } else {
                    if (q < TP) *reinterpret_cast<f32x4*>(&prow[q]) = acc[m][n];
                }
            }
        }
        return;
    }

    // ---- epilogue of the fused two-phase transposed stride-2 conv: tiles [0, NW/2) hold output
    // phase 0 (t = 2q), tiles [NW/2, NW) phase 1 (t = 2q+1) of the same channels, so a lane
    // owns 8 consecutive output samples ----
    if (phase2) {
        if constexpr (WN == 1 && ((NW % 2) == 0 || NW == 3)) {
            const bool vec2 = (a.flags & F_VEC4) != 0;
            const bool accum2 = (a.flags & F_ACCUM) != 0;
            // 8 consecutive outputs t = 2q .. 2q+7 of channel `ncol` (v[2r] = phase 0, v[2r+1] = phase 1 of position q + r)
            auto store8 = [&](int ncol, int m, float (&v)[8]) {
                const long long rowbase = (long long)b * a.obs0 + (long long)ncol * a.opitch0 + a.ooff0;
                const int q = q0 + wt0 + m * 16 + lg * 4;
                const int t0 = 2 * q;
                if (vec2 && t0 + 7 < a.Tlim) {
                    const long long idx = rowbase + t0;
                    if (a.msk0 != nullptr) {
                        const f32x4 m0 = *reinterpret_cast<const f32x4*>(&a.msk0[idx]);
                        const f32x4 m1 = *reinterpret_cast<const f32x4*>(&a.msk0[idx + 4]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[r] *= (m0[r] > 0.f) ? 1.f : 0.2f;
                            v[4 + r] *= (m1[r] > 0.f) ? 1.f : 0.2f;
                        }
                    }
                    const int pos0 = a.ooff0 + t0;
                    if (accum2 && (conv_acc_at(a, pos0) || conv_acc_at(a, pos0 + 7))) {      // (the window is wider than 8)
                        const f32x4 o0 = *reinterpret_cast<const f32x4*>(&a.dst0[idx]);
                        const f32x4 o1 = *reinterpret_cast<const f32x4*>(&a.dst0[idx + 4]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (conv_acc_at(a, pos0 + r)) v[r] += o0[r];
                            if (conv_acc_at(a, pos0 + 4 + r)) v[4 + r] += o1[r];
                        }
                    }
                    *reinterpret_cast<f32x4*>(&a.dst0[idx]) = (f32x4){v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(&a.dst0[idx + 4]) = (f32x4){v[4], v[5], v[6], v[7]};
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        if (t0 + r < a.Tlim) {
                            const long long idx = rowbase + t0 + r;
                            float x = v[r];
                            if (a.msk0 != nullptr) x *= (a.msk0[idx] > 0.f) ? 1.f : 0.2f;
                            if (accum2 && conv_acc_at(a, a.ooff0 + t0 + r)) x += a.dst0[idx];
                            a.dst0[idx] = x;
                        }
                    }
                }
            };
            if constexpr ((NW % 2) == 0) {
#pragma unroll
                for (int n = 0; n < NW / 2; ++n) {
                    const int ncol = n0h + n * 16 + li;
                    if (ncol >= a.N) continue;
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        float v[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v[2 * r] = acc[m][n][r]; v[2 * r + 1] = acc[m][NW / 2 + n][r]; }
                        store8(ncol, m, v);
                    }
                }
            } else {
                // 24 channels per workgroup (the channel counts of this network are multiples of 24: no padded columns,
                // where 16-channel tiles pad 24 -> 32, 72 -> 80, 120 -> 128, 168 -> 176): columns [0, 24) = phase 0,
                // [24, 48) = phase 1 of the same channels, i.e. MFMA column blocks {ph0 ch 0-15}, {ph0 ch 16-23 | ph1 ch 0-7},
                // {ph1 ch 8-23}.  The phase-1 partner of a lane's phase-0 value sits 8 lanes away in the same 16-lane row
                // (DPP row rotate by 8): lanes 0-7 own channels li and 16 + li, lanes 8-15 channel li.
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    float p1[4], p2[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // (copy the vector element to a scalar first: __builtin_bit_cast applied to an ext-vector element
                        // lvalue reads element 0 with this compiler)
                        const float s1 = acc[m][1][r], s2 = acc[m][2][r];
                        p1[r] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0x128, 0xF, 0xF, false));
                        p2[r] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2), 0x128, 0xF, 0xF, false));
                    }
                    const int cA = n0h + li;
                    if (cA < a.N) {
                        float v[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v[2 * r] = acc[m][0][r]; v[2 * r + 1] = li < 8 ? p1[r] : p2[r]; }
                        store8(cA, m, v);
                    }
                    const int cB = n0h + 16 + li;
                    if (li < 8 && cB < a.N) {
                        float v[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v[2 * r] = acc[m][1][r]; v[2 * r + 1] = p2[r]; }
                        store8(cB, m, v);
                    }
                }
            }
        }
        return;
    }

    // ---- epilogue ----
    const bool lrelu = (a.flags & F_LRELU) != 0;
    const bool accum = (a.flags & F_ACCUM) != 0;
    const bool vec = (a.flags & F_VEC4) != 0;
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
        const bool copies = (a.dec != nullptr || a.dec1 != nullptr) && ncol < a.N0;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int q = q0 + wt0 + m * 16 + lg * 4;
            if constexpr (FOLD) {
                int g = fast_div(q, a.Tout, inv_tout), qq = q - g * a.Tout;
                const bool first = ncol < a.N0;
                const long long colbase = first ? (long long)ncol * a.opitch0 + a.ooff0
                                                : (long long)(ncol - a.N0) * a.opitch1 + a.ooff1;
                const long long obs = first ? a.obs0 : a.obs1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (g < a.B) {
                        float v = acc[m][n][r] + bvv;
                        if (lrelu) v = fmaxf(0.2f * v, v);
                        const long long idx = (long long)g * obs + colbase + (long long)qq * a.ostride;
                        if (msk != nullptr) v *= (msk[idx] > 0.f) ? 1.f : 0.2f;
                        if (accum && conv_acc_at(a, (first ? a.ooff0 : a.ooff1) + qq * a.ostride)) v += dst[idx];
                        dst[idx] = v;
                        if (copies) conv_copy1(a, g, ncol, qq, v);
                    }
                    if (++qq == a.Tout) { qq = 0; ++g; }
                }
            } else if (vec && q + 3 < a.Tout) {
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
                const int pos0 = (ncol < a.N0 ? a.ooff0 : a.ooff1) + q;                 // vector path: ostride == 1
                if (accum && (conv_acc_at(a, pos0) || conv_acc_at(a, pos0 + 3))) {
                    const f32x4 old = *reinterpret_cast<const f32x4*>(&dst[idx]);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (conv_acc_at(a, pos0 + r)) v[r] += old[r];
                }
                *reinterpret_cast<f32x4*>(&dst[idx]) = v;
                if (copies) conv_copy4(a, b, ncol, q, v);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (q + r < a.Tout) {
                        float v = acc[m][n][r] + bvv;
                        if (lrelu) v = fmaxf(0.2f * v, v);
                        const long long idx = rowbase + (long long)(q + r) * a.ostride;
                        if (msk != nullptr) v *= (msk[idx] > 0.f) ? 1.f : 0.2f;
                        if (accum && conv_acc_at(a, (ncol < a.N0 ? a.ooff0 : a.ooff1) + (q + r) * a.ostride)) v += dst[idx];
                        dst[idx] = v;
                        if (copies) conv_copy1(a, b, ncol, q + r, v);
                    }
                }
            }
        }
    }
}

// sums the split-K partial tiles in a fixed order and applies the conv epilogue.  One thread = 4
// consecutive output positions of one (excerpt, channel) row: the partials are read as 16-byte
// vectors (rows are padded to 4 floats) and, when the destination allows (F_VEC4), stored as one.
__global__ __launch_bounds__(256) void conv_splitk_epilogue_kernel(ConvArgs a, int ksplit) {
    const int TP = (a.Tout + 3) & ~3;
    const int TP4 = TP >> 2;
    const long long total = (long long)a.B * a.N * TP4;
    const bool lrelu = (a.flags & F_LRELU) != 0;
    const bool accum = (a.flags & F_ACCUM) != 0;
    const bool vec = (a.flags & F_VEC4) != 0;
    const long long sstride = (long long)a.B * a.N * TP;        // floats between consecutive splits
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(i % TP4) * 4;
        const long long bn = i / TP4;
        const int ncol = (int)(bn % a.N), b = (int)(bn / a.N);
        const float bvv = (a.bias != nullptr) ? a.bias[ncol] : 0.f;
        float* dst; const float* msk; long long rowbase;
        if (ncol < a.N0) {
            rowbase = (long long)b * a.obs0 + (long long)ncol * a.opitch0 + a.ooff0;
            dst = a.dst0; msk = a.msk0;
        } else {
            rowbase = (long long)b * a.obs1 + (long long)(ncol - a.N0) * a.opitch1 + a.ooff1;
            dst = a.dst1; msk = a.msk1;
        }
        // mask / old values of the vector path are requested before the partials (independent loads)
        const bool vpath = vec && q + 3 < a.Tout;
        f32x4 mk = {1.f, 1.f, 1.f, 1.f}, old = {0.f, 0.f, 0.f, 0.f};
        if (vpath && msk != nullptr) mk = *reinterpret_cast<const f32x4*>(&msk[rowbase + q]);
        const int pos0 = (ncol < a.N0 ? a.ooff0 : a.ooff1) + q * a.ostride;
        const bool acc_v = accum && (conv_acc_at(a, pos0) || conv_acc_at(a, pos0 + 3));       // vector path: ostride == 1
        if (vpath && acc_v) old = *reinterpret_cast<const f32x4*>(&dst[rowbase + q]);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const float* pp = a.part + bn * TP + q;
        // the loads of a batch are all in flight before the first add (a load + wait per split made this kernel
        // ksplit memory latencies long: 12-16 us on the critical path of every deep level); the sum order is unchanged.
        // Batches of FOUR: this kernel runs 46 times per step beside the MFMA kernels and has to find room in their
        // register files -- batches of 8 (92 VGPRs) 8.41 ms per step, of 4 (69 VGPRs) 8.39, one batch of 16 (~130) 8.56
        int ks = 0;
        for (; ks + 4 <= ksplit; ks += 4) {
            f32x4 t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = *reinterpret_cast<const f32x4*>(pp + (long long)(ks + j) * sstride);
#pragma unroll
            for (int j = 0; j < 4; ++j) v += t[j];
        }
        {
            f32x4 t[3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
                t[j] = (ks + j < ksplit) ? *reinterpret_cast<const f32x4*>(pp + (long long)(ks + j) * sstride) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (ks + j < ksplit) v += t[j];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] += bvv;
            if (lrelu) v[r] = fmaxf(0.2f * v[r], v[r]);
        }
        if (a.ubw_dz != nullptr && ncol >= a.N0) {
            // fused adjoint of the 2x linear upsampling (no mask / accumulate on these columns: v = d_up[q .. q+3]).
            // d_up[q-1] belongs to the previous thread: its partials are summed again here in the same order.
            const int n = a.ubw_n, tup = a.Tout, c = ncol - a.N0;
            float prev = 0.f;
            if (q >= 1) {
                float t = 0.f;
                for (int k2 = 0; k2 < ksplit; ++k2) t += pp[(long long)k2 * sstride - 1];
                prev = t + bvv;
            }
            const long long xr = (long long)b * a.ubw_bs + (long long)c * a.ubw_pitch;
            const float d0 = v[0], d1 = q + 1 < tup ? v[1] : 0.f, d2 = q + 2 < tup ? v[2] : 0.f, d3 = q + 3 < tup ? v[3] : 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = (q >> 1) + h;
                if (i < n) {
                    float g = h ? d2 : d0;                                           // d_up[2i]
                    if (2 * i + 1 < tup) g += ((i == n - 1) ? 1.f : 0.5f) * (h ? d3 : d1);   // same-mode clamp: out[2n-1] = x[n-1]
                    if (i >= 1) g += 0.5f * (h ? d1 : prev);
                    a.ubw_dz[xr + i] = g * ((a.ubw_x[xr + i] > 0.f) ? 1.f : 0.2f);
                }
            }
        } else if (vpath) {
            const long long idx = rowbase + q;
            if (msk != nullptr) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= (mk[r] > 0.f) ? 1.f : 0.2f;
            }
            if (acc_v) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (conv_acc_at(a, pos0 + r)) v[r] += old[r];
            }
            *reinterpret_cast<f32x4*>(&dst[idx]) = v;
            if ((a.dec != nullptr || a.dec1 != nullptr) && ncol < a.N0) conv_copy4(a, b, ncol, q, v);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (q + r < a.Tout) {
                    const long long idx = rowbase + (long long)(q + r) * a.ostride;
                    float x = v[r];
                    if (msk != nullptr) x *= (msk[idx] > 0.f) ? 1.f : 0.2f;
                    if (accum && conv_acc_at(a, pos0 + r * a.ostride)) x += dst[idx];
                    dst[idx] = x;
                    if ((a.dec != nullptr || a.dec1 != nullptr) && ncol < a.N0) conv_copy1(a, b, ncol, q + r, x);
                }
            }
        }
        if (a.ups_y != nullptr && ncol < a.N0 && q < a.Tout) {
            // fused 2x upsampling of this row segment (forward launches: no mask, no accumulate, so v is the stored
            // activation): out[2i] = y[i], out[2i+1] = interp(y[i], y[i+1]).  y[q+4] belongs to the next thread: its
            // partials are summed again here in the same order (bit-identical to what that thread stores).
            const int n = a.Tout;
            float y[5];
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = v[r];
            y[4] = 0.f;
            if (q + 4 < n) {
                float t = 0.f;
                for (int k2 = 0; k2 < ksplit; ++k2) t += pp[(long long)k2 * sstride + 4];
                t += bvv;
                if (lrelu) t = fmaxf(0.2f * t, t);
                y[4] = t;
            }
            float* up = a.ups_y + (long long)b * a.ups_bs + (long long)ncol * a.ups_pitch;
            float sg = 0.f;
            if (a.ups_w != nullptr) sg = 1.f / (1.f + __expf(-a.ups_w[ncol]));
            float o[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = q + r;
                const bool has_next = i + 1 < n;
                const float x0 = y[r], x1 = has_next ? y[r + 1] : 0.f;
                o[2 * r] = x0;
                o[2 * r + 1] = (a.ups_w != nullptr) ? sg * x0 + (1.f - sg) * x1            // SAME: one zero on the right
                                                    : 0.5f * (x0 + (has_next ? x1 : x0)); // legacy bilinear clamps
            }
            const int t0 = 2 * q;
            if (t0 + 7 < a.ups_tup && q + 3 < n && (a.ups_pitch & 3) == 0 && (a.ups_bs & 3) == 0) {
                *reinterpret_cast<f32x4*>(up + t0) = (f32x4){o[0], o[1], o[2], o[3]};
                *reinterpret_cast<f32x4*>(up + t0 + 4) = (f32x4){o[4], o[5], o[6], o[7]};
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    if (q + (r >> 1) < n && t0 + r < a.ups_tup) up[t0 + r] = o[r];
            }
        }
    }
}

static thread_local int t_last_fused_ups = 0;       // conv_last_fused_ups()

// variant table -------------------------------------------------------------------------
struct ConvVariant { int MT, NW, WT, WN, CK, fold; };
static const ConvVariant kConvVariants[] = {
    {4, 2, 4, 1, 8}, {4, 3, 4, 1, 8}, {4, 4, 4, 1, 8}, {4, 5, 4, 1, 8},   //  0..3 : 256 x 32/48/64/80
    {2, 2, 4, 1, 8}, {2, 3, 4, 1, 8}, {2, 4, 4, 1, 8}, {2, 5, 4, 1, 8},   //  4..7 : 128 x 32/48/64/80
    {1, 2, 4, 1, 8}, {1, 3, 4, 1, 8},                                     //  8..9 :  64 x 32/48
    {1, 2, 2, 2, 8}, {1, 3, 2, 2, 8},                                     // 10..11:  32 x 64/96
    {1, 2, 1, 4, 8},                                                      // 12    :  16 x 128
    {4, 2, 4, 1, 4}, {1, 2, 4, 1, 4},                                     // 13..14: 1-/2-channel audio input
    {2, 6, 4, 1, 8}, {4, 6, 4, 1, 8},                                     // 15..16: 128/256 x 96 (fused two-phase dgrad)
    // non-power-of-two tile heights: let the autotuner land the workgroup count on a multiple of
    // the 256 CUs (e.g. T = 9203: 96 tiles of 192 rows x 16 excerpts = 6 workgroups per CU)
    {3, 2, 4, 1, 8}, {3, 3, 4, 1, 8}, {3, 4, 4, 1, 8}, {3, 5, 4, 1, 8},   // 17..20: 192 x 32/48/64/80
    {3, 6, 4, 1, 8},                                                      // 21    : 192 x 96
    {5, 2, 4, 1, 8}, {5, 3, 4, 1, 8},                                     // 22..23: 320 x 32/48
    {6, 2, 4, 1, 8}, {6, 3, 4, 1, 8},                                     // 24..25: 384 x 32/48
    // 4-channel chunks for stride-1 launches: half the LDS weight slab -> 3+ workgroups per CU
    {4, 3, 4, 1, 4}, {2, 3, 4, 1, 4}, {3, 3, 4, 1, 4}, {2, 2, 4, 1, 4}, {3, 2, 4, 1, 4},   // 26..30
    {4, 5, 4, 1, 4}, {3, 5, 4, 1, 4}, {2, 5, 4, 1, 4},                                    // 31..33
    // FOLD tiles for the deep levels: M = flattened (excerpt, position)
    {1, 2, 4, 1, 8, 1}, {1, 3, 4, 1, 8, 1},                               // 34..35:  64 x 32/48
    {2, 2, 4, 1, 8, 1}, {2, 3, 4, 1, 8, 1},                               // 36..37: 128 x 32/48
    {2, 3, 2, 2, 8, 1}, {4, 3, 2, 2, 8, 1},                               // 38..39:  64/128 x 96
    {2, 2, 2, 2, 8, 1}, {4, 2, 2, 2, 8, 1},                               // 40..41:  64/128 x 64
};
#define WUN_FIRST_FOLD_VARIANT 34

// In-workgroup split-K variants (conv_mfma_kernel<..., KG = 3>): variant WUN_FIRST_K3_VARIANT + i runs the tile of
// kK3Base[i] with three 256-thread groups per workgroup.  3 divides the chunk count of every layer of this network
// (channel counts are multiples of 24 = 3 chunks of 8), 4 does not.
#define WUN_KG 3
static const int kK3Base[] = {8, 9, 10, 11, 4, 5, 34, 35, 36, 37, 40};
static const int kNumK3 = (int)(sizeof(kK3Base) / sizeof(kK3Base[0]));
static int first_k3_variant();
static inline bool is_k3(int v) { return v >= first_k3_variant() && v < first_k3_variant() + kNumK3; }
static inline int conv_tile_variant(int v) { return is_k3(v) ? kK3Base[v - first_k3_variant()] : v; }   // index into kConvVariants

// can this launch use FOLD variant `v`?  (segments of every excerpt a tile touches must fit the LDS row)
static bool conv_fold_ok(const ConvArgs& a, int v) {
    const ConvVariant& cv = kConvVariants[v];
    if (!cv.fold || (a.flags & F_PHASE2)) return false;
    const int TT = cv.WT * cv.MT * 16;
    const int J = a.loader == LOADER_DEINT ? (a.KW + 1) / 2 : a.KW;
    long long nb = (TT - 1) / a.Tout + 2;
    if (nb > a.B) nb = a.B;
    if (nb * (a.Tout + J - 1) > WUN_FOLD_XCAP(TT)) return false;
    if (a.Tin >= (1 << 20)) return false;
    const long long span0 = a.bs0 * (long long)a.B, span1 = a.src1 ? a.bs1 * (long long)a.B : 0;
    return span0 < (1ll << 31) && span1 < (1ll << 31);
}

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
    if (a.Tout <= 96 && a.B > 1 && (a.N & 3) == 0) {
        // deep levels: fold the batch into M; 128-row tiles when there are enough rows to fill the chip
        const int c[2] = {3, 2};
        const int nw = pick_nw(a.N, c, 2);
        const int v = ((long long)a.B * a.Tout >= 1024 ? 36 : 34) + (nw == 3 ? 1 : 0);
        if (conv_fold_ok(a, v)) return v;
    }
    if (a.Tout <= 16) return 12;
    if (a.Tout <= 32) { const int c[2] = {3, 2}; return pick_nw(a.N, c, 2) == 3 ? 11 : 10; }
    if (a.Tout <= 64) { const int c[2] = {3, 2}; return pick_nw(a.N, c, 2) == 3 ? 9 : 8; }
    const int pad256 = ((a.Tout + 255) / 256) * 256, pad128 = ((a.Tout + 127) / 128) * 128;
    const int base = (pad128 < pad256) ? 4 : 0;
    const int c[4] = {3, 5, 4, 2};
    const int nw = pick_nw(a.N, c, 4);
    return base + (nw - 2);
}

// fused two-phase transposed conv: per workgroup NT/2 channels x 2 phases; needs even NW, WN == 1
int conv_pick_variant_phase2(const ConvArgs& a) {
    const int pad256 = ((a.Tout + 255) / 256) * 256, pad128 = ((a.Tout + 127) / 128) * 128;
    const bool t128 = pad128 < pad256;
    int best = 2, bestpad = 1 << 30;                 // channels per workgroup = NW*8
    const int cands[4] = {6, 4, 3, 2};               // (3: the packed 24-channel tile -- exact for this network's widths)
    for (int i = 0; i < 4; ++i) {
        const int half = cands[i] * 8;
        const int padded = ((a.N + half - 1) / half) * half;
        if (padded < bestpad) { bestpad = padded; best = cands[i]; }
    }
    if (best == 3) return t128 ? 5 : 1;              // 128 / 256 x 48
    if (best == 6) return 15;                        // 128 x 96 keeps 3 waves/SIMD
    if (best == 4) return t128 ? 6 : 2;
    return t128 ? 4 : 0;
}

long long conv_natural_wgs_phase2(const ConvArgs& a) {
    const int v = conv_pick_variant_phase2(a);
    const int TT = kConvVariants[v].WT * kConvVariants[v].MT * 16;
    const int half = kConvVariants[v].NW * 8;
    return (long long)((a.Tout + TT - 1) / TT) * ((a.N + half - 1) / half) * a.B;
}

static inline int conv_J(const ConvArgs& a);
// Vector (16-byte) paths need aligned bases / pitches; everything the plan allocates is.
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// may this launch use the DMA-staging instantiation (XVEC) of tile `variant`?
static int conv_dma_pitch(int width) { return fit_pitch((width + 3) & ~3, 16); }
static bool conv_xvec_ok(const ConvArgs& a, int variant) {
    const ConvVariant& cv = kConvVariants[variant];
    const int Ctot = a.C0 + a.C1;
    if (cv.fold || cv.CK != 8 || getenv("WUN_NO_DMA") != nullptr) return false;
    // (a stride-2 variant -- contiguous window, the four k lanes = four consecutive taps -- was built and measured in
    //  round 3: its chunk loop is 14 % faster but the step 1 % slower; not in the tree)
    if (a.loader != LOADER_DIRECT) return false;
    const int CKC = cv.CK;
    if ((Ctot % CKC) != 0 || (a.C0 % CKC) != 0 || (a.N & 3) != 0 || a.N < 4) return false;       // whole chunks, one source each
    if (!aligned16(a.src0) || (a.bs0 & 3) != 0 || (a.pitch0 & 3) != 0 || a.pitch0 < 4) return false;
    if (a.src1 != nullptr && (!aligned16(a.src1) || (a.bs1 & 3) != 0 || (a.pitch1 & 3) != 0 || a.pitch1 < 4)) return false;
    if (a.off0 < 0 || a.off1 < 0 || !aligned16(a.W)) return false;
    if ((long long)cv.CK * a.pitch0 >= (1ll << 31) || (long long)cv.CK * a.pitch1 >= (1ll << 31)) return false;
    return true;
}

static void conv_geom(const ConvArgs& a, int variant, int& TT, int& NT, int& J, int& XP, int& WP) {
    const ConvVariant& v = kConvVariants[variant];
    TT = v.WT * v.MT * 16;
    NT = v.WN * v.NW * 16;
    J = conv_J(a);
    XP = fit_pitch(v.fold ? WUN_FOLD_XCAP(TT) : TT + J - 1, 16);
    if (conv_xvec_ok(a, variant)) XP = conv_dma_pitch(TT + J - 1 + 3);       // whole 16-byte granules incl. the sub-vector shift
    WP = fit_pitch(NT, 16);
}

// workgroup tiles along M (per excerpt, or over the flattened batch for FOLD) and the excerpt factor of the grid
static inline long long conv_mtiles(const ConvArgs& a, int variant, int TT, int& bfac) {
    if (kConvVariants[variant].fold) { bfac = 1; return ((long long)a.B * a.Tout + TT - 1) / TT; }
    bfac = a.B;
    return (a.Tout + TT - 1) / TT;
}

size_t conv_lds_bytes(const ConvArgs& a, int variant) {
    if (is_k3(variant)) {
        const int bv = conv_tile_variant(variant);
        const size_t stage = WUN_KG * conv_lds_bytes(a, bv);
        const size_t red = (size_t)(WUN_KG - 1) * kConvVariants[bv].MT * kConvVariants[bv].NW * 256 * 16;   // accumulator hand-over
        return stage > red ? stage : red;
    }
    if (variant >= WUN_FIRST_RETIRED_VARIANT) return (size_t)1 << 30;             // retired index range: never launched
    int TT, NT, J, XP, WP;
    conv_geom(a, variant, TT, NT, J, XP, WP);
    const int CK = kConvVariants[variant].CK;
    return 2 * sizeof(float) * ((size_t)CK * XP + (size_t)J * CK * WP);     // double buffered
}

double conv_flops(const ConvArgs& a) {
    if (a.flags & F_PHASE2)   // useful work of the transposed conv: every forward tap once per forward output
        return 2.0 * a.kw_full * (double)(a.C0 + a.C1) * a.N * (double)a.Tin * a.B;
    return 2.0 * a.KW * (double)(a.C0 + a.C1) * a.N * (double)a.Tout * a.B;
}

// split-K decision: (ksplit, chunks per split) from shapes only (deterministic)
void conv_splitk(const ConvArgs& a, int variant, long long part_cap_floats, int& ksplit, int& cps) {
    int TT, NT, J, XP, WP;
    conv_geom(a, variant, TT, NT, J, XP, WP);
    const int CK = kConvVariants[variant].CK;
    const int CKC = a.loader == LOADER_DEINT ? CK / 2 : CK;
    const int nchunks = (a.C0 + a.C1 + CKC - 1) / CKC;
    int bfac;
    const long long natural = conv_mtiles(a, variant, TT, bfac) * ((a.N + NT - 1) / NT) * bfac;
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

template <int MT, int NW, int WT, int WN, int CK, bool VECW, bool FOLD = false, bool XVEC = false, int KG = 1>
static hipError_t conv_launch_t(ConvArgs a, int variant, float* part, long long part_cap, hipStream_t s) {
    int TT, NT, J, XP, WP;
    conv_geom(a, variant, TT, NT, J, XP, WP);
    const bool phase2 = (a.flags & F_PHASE2) != 0;
    int bfac;
    const int nTT = (int)conv_mtiles(a, variant, TT, bfac);
    const int nNT = phase2 ? (a.N + NT / 2 - 1) / (NT / 2) : (a.N + NT - 1) / NT;
    size_t lds = conv_lds_bytes(a, variant);
    if (KG > 1) {
        const size_t red = (size_t)(KG - 1) * MT * NW * 256 * 16;
        lds = KG * lds > red ? KG * lds : red;
        if (phase2 || lds > 160 * 1024) return hipErrorInvalidValue;
    }
    if (FOLD && !conv_fold_ok(a, variant)) return hipErrorInvalidValue;
    auto kern = conv_mfma_kernel<MT, NW, WT, WN, CK, VECW, FOLD, XVEC, KG>;
    static size_t lds_allowed = 64 * 1024;
    if (lds > lds_allowed) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_allowed = lds;
    }
    const size_t lds_launch = lds;
    int ksplit, cps;
    conv_splitk(a, variant, (part != nullptr && !phase2) ? part_cap : 0, ksplit, cps);
    if (a.force_ksplit > 0 && !phase2) {                         // autotuned choice
        const int CKC = a.loader == LOADER_DEINT ? CK / 2 : CK;
        const int nchunks = (a.C0 + a.C1 + CKC - 1) / CKC;
        int want = a.force_ksplit > nchunks ? nchunks : a.force_ksplit;
        if (part == nullptr) want = 1;
        const long long per = (long long)a.B * a.N * ((a.Tout + 3) & ~3);
        if (want > 1 && (long long)want * per > part_cap) return hipErrorInvalidValue;   // would overrun the scratch
        cps = (nchunks + want - 1) / want;
        ksplit = (nchunks + cps - 1) / cps;
    }
    if (KG > 1) {
        // every 256-thread group of every split gets the same whole number of chunks (uniform barriers)
        const int CKC = a.loader == LOADER_DEINT ? CK / 2 : CK;
        const int nchunks = (a.C0 + a.C1) / CKC;
        const int want = (a.force_ksplit > 0 && part != nullptr) ? a.force_ksplit : 1;
        if ((a.C0 + a.C1) % CKC != 0 || nchunks % (KG * want) != 0) return hipErrorInvalidValue;
        const long long per = (long long)a.B * a.N * ((a.Tout + 3) & ~3);
        if (want > 1 && (long long)want * per > part_cap) return hipErrorInvalidValue;
        ksplit = want;
        cps = nchunks / (KG * want);
    }
    a.cps = cps;
    a.part = ksplit > 1 ? part : nullptr;
    const long long grid = (long long)nTT * nNT * bfac * ksplit;
    if (grid <= 0) return hipSuccess;
    char nm[64];
    snprintf(nm, sizeof(nm), "conv_mfma_kernel<%d, %d, %d, %d, %d, %s%s%s%s>", MT, NW, WT, WN, CK, VECW ? "true" : "false",
             FOLD ? ", fold" : "", XVEC ? ", dma" : "", KG > 1 ? ", k3" : "");
    {
        char tag[160];
        snprintf(tag, sizeof(tag), "C=%d N=%d T=%d K=%d ld=%d B=%d ks=%d ph2=%d acc=%d os=%d grid=%lld", a.C0 + a.C1, a.N,
                 a.Tout, a.KW, a.loader, a.B, ksplit, (a.flags & F_PHASE2) ? 1 : 0, (a.flags & F_ACCUM) ? 1 : 0,
                 a.ostride, grid);
        ProfScope ps(nm, conv_flops(a), s, tag);
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256 * KG), lds_launch, s, a, nTT, nNT, J, XP, WP);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || ksplit == 1) return e;
    const long long total = (long long)a.B * a.N * (((a.Tout + 3) & ~3) >> 2);
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    {
        const double elems = (double)a.B * a.N * a.Tout;
        double by = 4.0 * elems * (ksplit + 1 + (a.msk0 != nullptr ? 1 : 0) + ((a.flags & F_ACCUM) ? 1 : 0));
        if (a.ups_y != nullptr) by += 4.0 * (double)a.B * a.N * a.ups_tup;
        ProfScope ps("conv_splitk_epilogue_kernel", 0.0, s, "", by);
        hipLaunchKernelGGL(conv_splitk_epilogue_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, ksplit);
    }
    t_last_fused_ups = (a.ups_y != nullptr || a.ubw_dz != nullptr) ? 1 : 0;
    return hipGetLastError();
}

// Is (tile variant, split-K factor) a choice the dispatcher may use for this launch?  The single
// rule set behind the autotuner's candidate list, the test hook's forced choices and imported
// tuning tables: a choice that fails here is never launched (a stale table cannot push a split
// past the partial scratch or select a tile the loader does not support).
bool conv_choice_ok(const ConvArgs& a, long long part_cap, int v, int ks) {
    const int nvar = (int)(sizeof(kConvVariants) / sizeof(kConvVariants[0]));
    if (is_k3(v)) {
        // the tile's own rules, whole chunks for every group of every split, the three staging regions inside the CU's LDS
        const int bv = conv_tile_variant(v);
        static const bool off = getenv("WUN_NO_K3") != nullptr && atoi(getenv("WUN_NO_K3")) != 0;
        if (off || (a.flags & F_PHASE2) || ks < 1 || !conv_choice_ok(a, part_cap, bv, 1)) return false;
        const ConvVariant& cv = kConvVariants[bv];
        const int CKC = a.loader == LOADER_DEINT ? cv.CK / 2 : cv.CK;
        const int Ctot = a.C0 + a.C1;
        if (Ctot % CKC) return false;
        const int nchunks = Ctot / CKC;
        if (nchunks % (WUN_KG * ks) != 0) return false;
        if (conv_lds_bytes(a, v) > 160 * 1024) return false;
        const int TT = cv.WT * cv.MT * 16, NT = cv.WN * cv.NW * 16;
        int bfac;
        const long long natural = conv_mtiles(a, bv, TT, bfac) * ((a.N + NT - 1) / NT) * bfac;
        if (natural * ks > 2048) return false;                         // (a launch that fills the chip needs no help)
        const long long per = (long long)a.B * a.N * ((a.Tout + 3) & ~3);
        return ks == 1 || (long long)ks * per <= part_cap;
    }
    if (v >= WUN_FIRST_RETIRED_VARIANT) return false;                // retired index range (wun_internal.h)
    if (v < 0 || v >= nvar || ks < 1) return false;
    const int Ctot = a.C0 + a.C1;
    const bool phase2 = (a.flags & F_PHASE2) != 0;
    if ((a.N & 3) != 0) return false;                                // scalar-weight fallback: fixed menu
    const ConvVariant& cv = kConvVariants[v];
    if (Ctot <= 4 && cv.CK != 4) return false;
    if (Ctot > 4 && cv.CK == 4 && (a.loader == LOADER_DEINT || phase2 || v < 26)) return false;
    if (phase2 && !(cv.WN == 1 && ((cv.NW % 2) == 0 || cv.NW == 3))) return false;
    if (cv.fold && !conv_fold_ok(a, v)) return false;
    const int TT = cv.WT * cv.MT * 16;
    const int NT = phase2 ? cv.WN * cv.NW * 8 : cv.WN * cv.NW * 16;   // channels per workgroup
    if (!cv.fold && TT > 16 && TT >= 2 * a.Tout) return false;       // mostly padding in time
    if (cv.fold && (long long)TT >= 2ll * a.B * a.Tout) return false;
    const int padded = ((a.N + NT - 1) / NT) * NT;
    if (padded * 3 > a.N * 4 + 48) return false;                      // > ~33 % padded columns
    if (conv_lds_bytes(a, v) > 150 * 1024) return false;
    if (ks == 1) return true;
    const int CKC = a.loader == LOADER_DEINT ? cv.CK / 2 : cv.CK;
    const int nchunks = (Ctot + CKC - 1) / CKC;
    int bfac;
    const long long natural = conv_mtiles(a, v, TT, bfac) * ((a.N + NT - 1) / NT) * bfac;
    const long long per = (long long)a.B * a.N * ((a.Tout + 3) & ~3);
    return !(phase2 || ks > nchunks || natural * ks > 6144 || (long long)ks * per > part_cap);
}

// Candidate (tile variant, split-K) choices for the autotuner.  Returns the number written.
int conv_list_candidates(const ConvArgs& a, long long part_cap, ConvChoice* out, int maxn) {
    int n = 0;
    const int nvar = conv_num_variants();
    static const int ks_menu[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48};
    for (int v = 0; v < nvar && n < maxn; ++v)
        for (unsigned i = 0; i < sizeof(ks_menu) / sizeof(ks_menu[0]) && n < maxn; ++i) {
            if (!conv_choice_ok(a, part_cap, v, ks_menu[i])) {
                if (is_k3(v) && ks_menu[i] < 8) continue;            // (divisibility: a larger split may fit again)
                break;                                               // larger splits fail the same bounds
            }
            out[n].variant = v; out[n].ksplit = ks_menu[i]; ++n;
        }
    return n;
}

static_assert(sizeof(kConvVariants) / sizeof(kConvVariants[0]) == WUN_FIRST_RETIRED_VARIANT, "the retired index range follows the tile table");
static int first_k3_variant() { return WUN_FIRST_RETIRED_VARIANT + WUN_NUM_RETIRED_VARIANTS; }
int conv_num_variants() { return first_k3_variant() + kNumK3; }


int conv_last_fused_ups() { return t_last_fused_ups; }

hipError_t launch_conv(const ConvArgs& a_in, float* part, long long part_cap, hipStream_t s) {
    ConvArgs a = a_in;
    t_last_fused_ups = 0;
    // the fused upsampled copy is written by the split-K epilogue of plain forward launches only
    if (a.ups_y != nullptr && (a.dst1 != nullptr || a.msk0 != nullptr || a.ostride != 1 || a.ooff0 != 0 ||
                               (a.flags & (F_ACCUM | F_PHASE2)) != 0 || a.N0 != a.N || !aligned16(a.ups_y)))
        return hipErrorInvalidValue;
    // ... the fused adjoint by the split-K epilogue of an input-gradient launch with two destinations, dst1 unmasked
    if (a.ubw_dz != nullptr && (a.dst1 == nullptr || a.msk1 != nullptr || a.ostride != 1 || a.bias != nullptr ||
                                (a.flags & (F_ACCUM | F_PHASE2 | F_LRELU)) != 0 || a.N0 >= a.N || a.ubw_x == nullptr))
        return hipErrorInvalidValue;
    if (a.acc_len == 0) { a.acc_lo = 0; a.acc_len = 0x7FFFFFFFu; }          // F_ACCUM over the whole row (default)
    if (conv_J(a) > WUN_JMAX) return hipErrorInvalidValue;       // rejected at plan creation
    if (a.KW <= 0) {
        // no taps (odd output phase of a transposed stride-2 conv with filter_size 1): the conv is the
        // epilogue of an empty sum -- bias / activation / mask / accumulate through the split-K epilogue
        const long long total = (long long)a.B * a.N * (((a.Tout + 3) & ~3) >> 2);
        if (total <= 0) return hipSuccess;
        long long blocks = (total + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(conv_splitk_epilogue_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, 0);
        return hipGetLastError();
    }
    const bool vecw = (a.N & 3) == 0 && aligned16(a.W);
    bool vec = a.ostride == 1 && aligned16(a.dst0) && (a.obs0 & 3) == 0 && (a.opitch0 & 3) == 0 && (a.ooff0 & 3) == 0;
    if (a.dst1 != nullptr)
        vec = vec && aligned16(a.dst1) && (a.obs1 & 3) == 0 && (a.opitch1 & 3) == 0 && (a.ooff1 & 3) == 0;
    if (a.msk0 != nullptr) vec = vec && aligned16(a.msk0);
    if (a.msk1 != nullptr) vec = vec && aligned16(a.msk1);
    if (a.dec != nullptr) vec = vec && (a.decpitch & 1) == 0 && (a.decbs & 1) == 0;
    if ((a.flags & F_PHASE2) && !(a.ostride == 1 && a.dst1 == nullptr && vecw)) return hipErrorInvalidValue;
    if (vec) a.flags |= F_VEC4;
    int v = (a.flags & F_PHASE2) ? conv_pick_variant_phase2(a) : conv_pick_variant(a);
    if (a.force_variant > 0 && vecw) {
        v = a.force_variant - 1;
        if (!conv_choice_ok(a, part != nullptr ? part_cap : 0, v, a.force_ksplit > 0 ? a.force_ksplit : 1)) return hipErrorInvalidValue;
    }
    if (is_k3(v)) {
        const int bv = conv_tile_variant(v);
        const bool xv3 = conv_xvec_ok(a, bv);
#define WUN_K3(i, MT, NW, WT, WN) case i: \
        if (xv3) return conv_launch_t<MT, NW, WT, WN, 8, true, false, true, WUN_KG>(a, bv, part, part_cap, s); \
        return conv_launch_t<MT, NW, WT, WN, 8, true, false, false, WUN_KG>(a, bv, part, part_cap, s);
#define WUN_K3F(i, MT, NW, WT, WN) case i: return conv_launch_t<MT, NW, WT, WN, 8, true, true, false, WUN_KG>(a, bv, part, part_cap, s);
        switch (bv) {
            WUN_K3(8, 1, 2, 4, 1) WUN_K3(9, 1, 3, 4, 1) WUN_K3(10, 1, 2, 2, 2) WUN_K3(11, 1, 3, 2, 2)
            WUN_K3(4, 2, 2, 4, 1) WUN_K3(5, 2, 3, 4, 1)
            WUN_K3F(34, 1, 2, 4, 1) WUN_K3F(35, 1, 3, 4, 1) WUN_K3F(36, 2, 2, 4, 1) WUN_K3F(37, 2, 3, 4, 1)
            WUN_K3F(40, 2, 2, 2, 2)
            default: return hipErrorInvalidValue;
        }
#undef WUN_K3
#undef WUN_K3F
    }
    if (v >= WUN_FIRST_RETIRED_VARIANT) return hipErrorInvalidValue;
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
    const bool xv = conv_xvec_ok(a, v);
#define WUN_CV(i, MT, NW, WT, WN, CK) case i: \
        if (xv) return conv_launch_t<MT, NW, WT, WN, CK, true, false, true>(a, v, part, part_cap, s); \
        return conv_launch_t<MT, NW, WT, WN, CK, true>(a, v, part, part_cap, s);
#define WUN_CV4(i, MT, NW, WT, WN, CK) case i: return conv_launch_t<MT, NW, WT, WN, CK, true>(a, v, part, part_cap, s);
    switch (v) {
        WUN_CV(0, 4, 2, 4, 1, 8) WUN_CV(1, 4, 3, 4, 1, 8) WUN_CV(2, 4, 4, 4, 1, 8) WUN_CV(3, 4, 5, 4, 1, 8)
        WUN_CV(4, 2, 2, 4, 1, 8) WUN_CV(5, 2, 3, 4, 1, 8) WUN_CV(6, 2, 4, 4, 1, 8) WUN_CV(7, 2, 5, 4, 1, 8)
        WUN_CV(8, 1, 2, 4, 1, 8) WUN_CV(9, 1, 3, 4, 1, 8)
        WUN_CV(10, 1, 2, 2, 2, 8) WUN_CV(11, 1, 3, 2, 2, 8)
        WUN_CV(12, 1, 2, 1, 4, 8)
        WUN_CV4(13, 4, 2, 4, 1, 4) WUN_CV4(14, 1, 2, 4, 1, 4)
        WUN_CV(15, 2, 6, 4, 1, 8) WUN_CV(16, 4, 6, 4, 1, 8)
        WUN_CV(17, 3, 2, 4, 1, 8) WUN_CV(18, 3, 3, 4, 1, 8) WUN_CV(19, 3, 4, 4, 1, 8) WUN_CV(20, 3, 5, 4, 1, 8)
        WUN_CV(21, 3, 6, 4, 1, 8)
        WUN_CV(22, 5, 2, 4, 1, 8) WUN_CV(23, 5, 3, 4, 1, 8)
        WUN_CV(24, 6, 2, 4, 1, 8) WUN_CV(25, 6, 3, 4, 1, 8)
        WUN_CV4(26, 4, 3, 4, 1, 4) WUN_CV4(27, 2, 3, 4, 1, 4) WUN_CV4(28, 3, 3, 4, 1, 4) WUN_CV4(29, 2, 2, 4, 1, 4)
        WUN_CV4(30, 3, 2, 4, 1, 4) WUN_CV4(31, 4, 5, 4, 1, 4) WUN_CV4(32, 3, 5, 4, 1, 4) WUN_CV4(33, 2, 5, 4, 1, 4)
#define WUN_CF(i, MT, NW, WT, WN, CK) case i: return conv_launch_t<MT, NW, WT, WN, CK, true, true>(a, v, part, part_cap, s);
        WUN_CF(34, 1, 2, 4, 1, 8) WUN_CF(35, 1, 3, 4, 1, 8) WUN_CF(36, 2, 2, 4, 1, 8) WUN_CF(37, 2, 3, 4, 1, 8)
        WUN_CF(38, 2, 3, 2, 2, 8) WUN_CF(39, 4, 3, 2, 2, 8) WUN_CF(40, 2, 2, 2, 2, 8) WUN_CF(41, 4, 2, 2, 2, 8)
#undef WUN_CF
        default: return hipErrorInvalidValue;
    }
#undef WUN_CV
#undef WUN_CV4
}

// =====================================================================================
// weight / bias gradient
// =====================================================================================
// D[(cin,tap) 16][cout 16] += A[(cin,tap)][q] * B[q][cout]; reduction over (batch, q) split
// over workgroups ("units" of TK <= 128 output positions), partial sums to scratch, summed
// in a fixed order by wgrad_reduce_kernel.  Row (cin,tap) of A is the input row `cin` read at
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
    int bid = xcd_contiguous_block((int)blockIdx.x, (int)gridDim.x);
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
    // off + Tin <= pitch.  Every vector is loaded from an in-row clamped position; zero fill
    // (virtual time outside [0, Tin), q beyond the batch) is applied at the LDS write and only
    // for units that touch a boundary, so interior units stage with almost no VALU work.
    // Unit-invariant per-vector state is ONE packed register per vector (the accumulator-heavy
    // geometries have no registers to spare; the few VALU ops that unpack it per unit are noise
    // next to the unit's hundreds of MFMAs):
    //   xpk[i] = bit 31: this thread stages vector i | row (8 bits) << 23 | c4 (7 bits) << 16 | LDS float offset (16 bits)
    //   zpk[i] = same fields for the dz tile
    int xpk[WUN_WG_XIT];
    int zpk[ZIT];
#pragma unroll
    for (int i = 0; i < WUN_WG_XIT; ++i) {
        const int f = tid + i * 256;
        const int row = (int)(((float)f + 0.5f) * inv_xw4);
        const int c4 = f - row * XW4;
        const bool rok = row < nCh;
        const int ldsoff = deint ? (row * 2) * XP + 2 * c4 : row * XP + 4 * c4;
        xpk[i] = rok ? (int)(0x80000000u | ((unsigned)row << 23) | ((unsigned)c4 << 16) | (unsigned)ldsoff)
                     : (int)((unsigned)c4 << 16);
    }
#pragma unroll
    for (int i = 0; i < ZIT; ++i) {
        const int f = tid + i * 256;
        const int row = (int)(((float)f + 0.5f) * inv_tk4);
        const int c4 = f - row * TK4;
        zpk[i] = row < NG ? (int)(0x80000000u | ((unsigned)row << 23) | ((unsigned)c4 << 16) | (unsigned)(row * ZP + 4 * c4))
                          : (int)((unsigned)c4 << 16);
    }

    auto load_unit = [&](int u) {
        const int b = u / a.nQT, qt = u - b * a.nQT;
        const int q0 = qt * TK;
        const int tb = (deint ? 2 * q0 : q0) - a.shift;
        const float* base0 = a.src0 + (long long)b * a.bs0;
        const float* base1 = (a.C1 > 0) ? a.src1 + (long long)b * a.bs1 : base0;
        const int e00 = (tb + a.off0) & ~3, e01 = (tb + a.off1) & ~3;   // uniform element shift per source
#pragma unroll
        for (int i = 0; i < WUN_WG_XIT; ++i) {
            int pk = xpk[i];
            asm volatile("" : "+v"(pk));          // keep the unpacking inside the unit loop (no hoisted copies)
            int c = cLo + ((pk >> 23) & 255);                         // rows past nCh carry row 0
            c = c < Ctot ? c : Ctot - 1;                              // bias-only row group: cLo == Ctot (nothing staged)
            const bool s1 = c >= a.C0;
            const int xro = s1 ? (c - a.C0) * a.pitch1 : c * a.pitch0; // element offset of the source row
            int e = (s1 ? e01 : e00) + (((pk >> 16) & 127) << 2);      // element index inside the row
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
        // interior unit: every staged element has 0 <= t < Tin for both sources -> no masking
        const int span = 4 * XW4;
        const int t00 = ((tb + a.off0) & ~3) - a.off0, t01 = ((tb + a.off1) & ~3) - a.off1;
        const bool xedge = t00 < 0 || t00 + span > a.Tin || (a.C1 > 0 && (t01 < 0 || t01 + span > a.Tin));
#pragma unroll
        for (int i = 0; i < WUN_WG_XIT; ++i) {
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
                float* p = Zs + (pk & 0xFFFF);
                *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
                *reinterpret_cast<float2*>(p + 2) = make_float2(v[2], v[3]);
            }
        }
        (void)b;
    };

    const int nunits = a.B * a.nQT;
    const int u0 = split * a.units_per_split;
    int u1 = u0 + a.units_per_split;
    if (u1 > nunits) u1 = nunits;
    constexpr bool wb_noload = false, wb_nostore = false, wb_nomfma = false, wb_noepi = false;
    if (u0 < u1 && !wb_noload) load_unit(u0);
    for (int u = u0; u < u1; ++u) {
        const int qt = u % a.nQT;
        int nq = a.Tq - qt * TK; if (nq > TK) nq = TK;
        const int nsteps = (nq + 3) >> 2;
        __syncthreads();
        if (!wb_nostore) store_unit(u);
        __syncthreads();
        if (u + 1 < u1 && !wb_noload) load_unit(u + 1);
        if (!wb_nomfma) {
            // Branch-free MFMA stream over all MTW tiles of the wave (the geometry keeps the ragged
            // last group's dead tiles few).
            auto run_unit = [&](auto full_tag) {
                constexpr bool FULL = decltype(full_tag)::value;
                float a0[MTW], b0[NW], a1[MTW], b1[NW];
                auto ldop = [&](int st, float (&av)[MTW], float (&bv)[NW]) {
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) av[mt] = lds[rowoff[mt] + 4 * st + lg];
#pragma unroll
                    for (int n = 0; n < NW; ++n) bv[n] = Zs[(n * 16 + li) * ZP + 4 * st + lg];
                };
                // same order as the conv kernel: first MFMA, LDS reads of the next k-step, the rest
                auto mm_first = [&](const float (&av)[MTW], const float (&bv)[NW]) {
                    acc[0][0] = mfma16(av[0], bv[0], acc[0][0]);
                };
                auto mm_rest = [&](const float (&av)[MTW], const float (&bv)[NW]) {
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) {
                        if (FULL || mt < nact) {
#pragma unroll
                            for (int n = 0; n < NW; ++n)
                                if (mt + n > 0) acc[mt][n] = mfma16(av[mt], bv[n], acc[mt][n]);
                        }
                    }
                };
                auto pin = [&]() {
                    if constexpr (FULL) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, MTW + NW, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, MTW * NW - 1, 0);
                    }
                };
                const int npair = nsteps >> 1;
                const int last = nsteps - 1;
                ldop(0, a0, b0);
                int st = 0;
                for (int it = 0; it < npair; ++it, st += 2) {
                    mm_first(a0, b0);
                    ldop(st + 1, a1, b1);
                    mm_rest(a0, b0);
                    pin();
                    mm_first(a1, b1);
                    ldop(st + 2 < last ? st + 2 : last, a0, b0);
                    mm_rest(a1, b1);
                    pin();
                }
                if (nsteps & 1) {
                    mm_first(a0, b0);
                    mm_rest(a0, b0);
                }
            };
            run_unit(std::true_type{});      // dead tiles of a ragged last group read the ones row and are never stored
        }
    }

    if (wb_noepi && acc[0][0][0] != 12345.678f) return;
    if (!a.direct) {
        // split partial, tile-major: each wave stores its accumulator tiles as contiguous 1 KiB
        // runs (register order); wgrad_reduce_kernel sums the splits and does the scatter to the
        // [K][Cin][Cout] layout once.  Dead tiles / padded columns are written too (finite) and
        // ignored by the reduction.
        f32x4* tile = reinterpret_cast<f32x4*>(a.out) +
                      ((((long long)(a.split_base + split) * nMG + mg) * nNG + ng) * (MG * NG / 4)) +
                      wave * (MTW * NW * 64) + lane;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int n = 0; n < NW; ++n) tile[(mt * NW + n) * 64] = acc[mt][n];
        return;
    }
    // single split: final layout directly (weights [K][Cin][Cout], then the bias row)
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

// Sums the tile-major split partials of one weight gradient in split order (SL split lanes per
// element, combined in lane order: deterministic) and scatters to the final layout:
// out_w[K][Cin][Cout], out_b[Cout].  One thread = one f32x4 accumulator register of the tile.
struct WgradReduceArgs {
    const float* partial; float* out_w; float* out_b;
    int nsplit, MTW, NW, nMG, nNG, Mtot, KW, Ctot, N;
};

template <int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradReduceArgs a) {
    __shared__ f32x4 red[SL > 1 ? 256 : 1];
    constexpr int VPB = 256 / SL;                                  // vector slots per block
    const int MG = 4 * a.MTW * 16, NG = a.NW * 16;
    const int tile_v = MG * NG / 4;                                // f32x4 slots per tile
    const int slot = threadIdx.x % VPB, sl = threadIdx.x / VPB;
    const long long gv = (long long)blockIdx.x * VPB + slot;       // global slot over all tiles
    const long long ntile = (long long)a.nMG * a.nNG;
    const bool live = gv < ntile * tile_v;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const f32x4* p = reinterpret_cast<const f32x4*>(a.partial) + gv;
        const long long sstride = ntile * tile_v;
        // batches of independent loads, summed in split order (a load + wait per split serialised nsplit / SL
        // memory latencies: 58 us for 922 splits)
        int k = sl;
        for (; k + 7 * SL < a.nsplit; k += 8 * SL) {
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
        sum = red[slot];
#pragma unroll
        for (int k = 1; k < SL; ++k) sum += red[k * VPB + slot];
    }
    if (!live) return;
    const int t = (int)(gv / tile_v), v = (int)(gv % tile_v);
    const int mg = t / a.nNG, ng = t % a.nNG;
    const int per_wave = a.MTW * a.NW * 64;
    const int wave = v / per_wave, rem = v % per_wave;
    const int tn = rem / 64, lane = rem % 64;
    const int mt = tn / a.NW, n = tn % a.NW;
    const int li = lane & 15, lg = lane >> 4;
    const int col = ng * NG + n * 16 + li;
    if (col >= a.N) return;
    const int r0 = mg * MG + (wave * a.MTW + mt) * 16 + lg * 4;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int r = r0 + r4;
        if (r < a.Mtot) {
            const int c = r / a.KW, kk = r - c * a.KW;
            a.out_w[((long long)kk * a.Ctot + c) * a.N + col] = sum[r4];
        } else if (r == a.Mtot) {
            a.out_b[col] = sum[r4];
        }
    }
}


WgradGeom wgrad_geom(const WgradArgs& a) {
    WgradGeom g;
    const int Ctot = a.C0 + a.C1;
    const int mtiles = (Ctot * a.KW + 1 + 15) / 16;
    int bestnw = 3, bestpad = 1 << 30;
    for (int nw = 3; nw >= 1; --nw) {
        const int padded = ((a.N + nw * 16 - 1) / (nw * 16)) * nw * 16;
        if (padded < bestpad) { bestpad = padded; bestnw = nw; }
    }
    g.NW = (a.force_nw >= 1 && a.force_nw <= 5) ? a.force_nw : bestnw;
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
    if (a.force_mtw == 1 || a.force_mtw == 2 || a.force_mtw == 4 || a.force_mtw == 6) mtw = a.force_mtw;
    if (g.NW > 3 && mtw > 4) mtw = 4;          // accumulator budget: MTW * NW <= 20 tiles
    g.ONESP = (tk + 15) & ~15;
    const int xit = WUN_WG_XIT;
    for (;;) {
        const int MG = 4 * mtw * 16;
        int nch = (MG + a.KW - 2) / a.KW + 1;
        if (nch > Ctot) nch = Ctot;
        g.nChMax = nch;
        if ((long long)nch * g.XW4 <= (long long)xit * 256 || mtw == 1) break;
        mtw = mtw == 6 ? 4 : (mtw == 4 ? 2 : 1);
    }
    g.MTW = mtw;
    const int MG = 4 * g.MTW * 16, NG = g.NW * 16;
    g.nMG = (Ctot * a.KW + 1 + MG - 1) / MG;
    g.nNG = (a.N + NG - 1) / NG;
    const size_t setb = (size_t)g.nChMax * (deint ? 2 : 1) * g.XP + (size_t)NG * g.ZP;
    g.lds = sizeof(float) * ((size_t)g.ONESP + setb);
    return g;
}

// (bf16 speed mode: the same questions answered for the bf16 kernel's own tiling)
int wgrad_max_units(const WgradArgs& a) {
    if (a.win) return wgrad_win_units(a);
    const int TK = a.bf16 ? wgrad_bf16_geom(a).TK : wgrad_geom(a).TK;
    return a.B * ((a.Tq + TK - 1) / TK);
}

int wgrad_pick_nsplit(const WgradArgs& a) {
    if (a.win) {
        const long long units = wgrad_win_units(a), per = wgrad_win_tiles(a);
        long long ns = (1024 + per - 1) / per;
        if (ns > units) ns = units;
        if (units >= 8 && ns > units / 2) ns = units / 2;
        if (ns < 1) ns = 1;
        const long long ups = (units + ns - 1) / ns;
        return (int)((units + ups - 1) / ups);
    }
    int TK, nMG, nNG;
    if (a.bf16) { const WgradBfGeom b = wgrad_bf16_geom(a); TK = b.TK; nMG = b.nMG; nNG = b.nNG; }
    else { const WgradGeom f = wgrad_geom(a); TK = f.TK; nMG = f.nMG; nNG = f.nNG; }
    const int nQT = (a.Tq + TK - 1) / TK;
    const long long units = (long long)a.B * nQT;
    const long long per = (long long)nMG * nNG;
    long long ns = (1024 + per - 1) / per;
    // big weight blocks: every split writes+reads the whole block once, so aim lower (512 workgroups)
    const long long blk = (long long)a.KW * (a.C0 + a.C1) * a.N;
    if (blk > (1ll << 18)) ns = (512 + per - 1) / per;
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
    char tag[160];
    snprintf(tag, sizeof(tag), "C=%d N=%d T=%d K=%d ld=%d B=%d nsplit=%d grid=%lld", a.C0 + a.C1, a.N, a.Tq, a.KW,
             a.loader, a.B, a.nsplit, grid);
    ProfScope ps(nm, 2.0 * a.KW * (double)(a.C0 + a.C1) * a.N * (double)a.Tq * a.B, s, tag);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), g.lds, s, a, g.nMG, g.nNG, g.TK, g.XP,
                       g.ZP, g.nChMax, g.ONESP, g.XW4);
    return hipGetLastError();
}

hipError_t launch_wgrad(const WgradArgs& a, hipStream_t s) {
    if (a.bf16) return launch_wgrad_bf16(a, s);
    if (a.win) return launch_wgrad_win(a, s);
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
    WUN_WG(1, 4) WUN_WG(2, 4) WUN_WG(4, 4)
    WUN_WG(1, 5) WUN_WG(2, 5) WUN_WG(4, 5)
#undef WUN_WG
    return hipErrorInvalidValue;
}

void wgrad_resolved_geom(const WgradArgs& a, int& mtw, int& nw) {
    if (a.win) { mtw = a.force_mtw; nw = a.force_nw; return; }      // (own geometry; parts need not agree)
    if (a.bf16) { const WgradBfGeom b = wgrad_bf16_geom(a); mtw = b.MTW; nw = b.NW; return; }
    const WgradGeom g = wgrad_geom(a);
    mtw = g.MTW; nw = g.NW;
}

// floats one split of this weight gradient occupies in the tile-major partial buffer
long long wgrad_partial_floats(const WgradArgs& a) {
    if (a.win) return wgrad_win_partial_floats(a);
    if (a.bf16) { const WgradBfGeom b = wgrad_bf16_geom(a); return (long long)b.nMG * b.nNG * (4 * b.MTW * 16) * (b.NW * 16); }
    const WgradGeom g = wgrad_geom(a);
    return (long long)g.nMG * g.nNG * (4 * g.MTW * 16) * (g.NW * 16);
}

// partial: `nsplit` consecutive splits written by launch_wgrad calls that share `a`'s tile geometry
hipError_t launch_wgrad_reduce(const WgradArgs& a, const float* partial, int nsplit, float* out_w, float* out_b,
                               hipStream_t s) {
    if (a.bf16) return launch_wgrad_bf16_reduce(a, partial, nsplit, out_w, out_b, s);
    if (a.win) return launch_wgrad_win_reduce(a, partial, nsplit, out_w, out_b, s);
    const WgradGeom g = wgrad_geom(a);
    WgradReduceArgs r;
    r.partial = partial; r.out_w = out_w; r.out_b = out_b;
    r.nsplit = nsplit; r.MTW = g.MTW; r.NW = g.NW; r.nMG = g.nMG; r.nNG = g.nNG;
    r.Ctot = a.C0 + a.C1; r.KW = a.KW; r.Mtot = r.Ctot * a.KW; r.N = a.N;
    const long long slots = (long long)g.nMG * g.nNG * (4 * g.MTW * 16) * (g.NW * 16) / 4;
    // few elements but many splits: spread the splits over 4 / 16 lanes per element
    const int sl = (nsplit >= 64 && slots < (1 << 16)) ? 16 : (nsplit >= 8 && slots < (1 << 18) ? 4 : 1);
    const int vpb = 256 / sl;
    const long long blocks = (slots + vpb - 1) / vpb;
    char tag[96];
    snprintf(tag, sizeof(tag), "bytes=%lld nsplit=%d", (long long)((nsplit + 1) * slots * 16), nsplit);
    ProfScope ps("wgrad_reduce_kernel", 0.0, s, tag, (double)((nsplit + 1) * slots * 16));
    if (sl == 16) hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, r);
    else if (sl == 4) hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, r);
    else hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, r);
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

}  // namespace wun

