// Internal: kernel argument blocks + launcher prototypes shared by wun_kernels.hip
// (device code) and wun_plan.hip (host plan / C ABI).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace wun {

enum { LOADER_DIRECT = 0, LOADER_DEINT = 1 };

// Epilogue flag bits
enum { F_LRELU = 1, F_ACCUM = 2, F_VEC4 = 4, F_PHASE2 = 8 };

// One implicit-GEMM 1-D convolution launch.  The input is a virtual channel-concat of
// up to two NCW sources, zero outside [0, Tin) (this is how crop+concat, 'same' zero
// padding and the halo of valid convolutions are expressed without copies).
//   DIRECT: out[b][n][q] = epi( sum_{j<J, c} W[j][c][n] * in[b][c][q + j - shift] )
//   DEINT : out[b][n][q] = epi( sum_{k<KW,c} W[k][c][n] * in[b][c][2q + k - shift] )
// Outputs n < N0 go to dst0, the rest to dst1 (dgrad of a concat).  Output element q is
// stored at dst[b*obs + n*opitch + ooff + q*ostride]; msk (same geometry as dst) applies
// the LeakyReLU derivative of the forward activation stored there; dec receives a
// compact copy of the even q (the [:, ::2, :] decimation) of dst0.
struct ConvArgs {
    const float* src0; const float* src1;
    long long bs0, bs1;
    int pitch0, pitch1;
    int off0, off1;
    int C0, C1;
    int Tin;
    int shift;
    const float* W;
    const float* bias;
    int KW;          // taps (DIRECT: J = KW; DEINT: J = ceil(KW/2))
    int N;           // output channels
    int N0;          // split point between dst0 / dst1
    float* dst0; float* dst1;
    long long obs0, obs1;
    int opitch0, opitch1;
    int ooff0, ooff1;
    const float* msk0; const float* msk1;
    int ostride;
    int Tout;
    float* dec; long long decbs; int decpitch;
    // Round 6 (context plans of the exact-fp32 mode: the decimated stream IS a slice of the encoder output,
    // UnetAudioSeparator.py:98-100, computed once): two more copies of dst0, same validity rules as `dec`.
    //   dec_exp != 0: `dec` is an EXPANDED copy instead of a compact one -- output q goes to dec[b][n][2q - dec_lo] when
    //                 0 <= 2q - dec_lo < dec_len (the stride-2 conv of a down level writes the even positions of the skip window);
    //   dec1        : compact copy of the ODD q of dst0, dec1[b][n][q >> 1] (an up level's input gradient splits the skip
    //                 window's gradient by the parity of the absolute conv position: one half is added to the decimated
    //                 stream's gradient, the other feeds the odd-position launches).
    int dec_exp; int dec_lo; unsigned dec_len;
    float* dec1; long long dec1bs; int dec1pitch;
    int dec_off, dec1_off;       // first element of the copies inside their rows, in ELEMENTS of the copy's tensor (fp32 or bf16)
    int flags;
    int B;
    int loader;
    int Tlim;        // F_PHASE2: true output length (outputs t = 2q+p < Tlim)
    int kw_full;     // F_PHASE2: taps of the forward conv (FLOP accounting)
    int force_variant;   // autotuner: tile variant index + 1 (0 = heuristic)
    int force_ksplit;    // autotuner: split-K factor (0 = heuristic)
    int cps;         // split-K: channel chunks per split (set by the launcher)
    float* part;     // split-K partial buffer (set by the launcher) or null
    int wb_c8p, wb_npad;   // bf16 mode: W points at the packed image [KW][wb_c8p][wb_npad][8] (wun_bf16.hip)
    // F_ACCUM applies to the row positions pos = ooff + q*ostride with acc_lo <= pos < acc_lo + acc_len only; elsewhere
    // the result is stored.  (A down level's input gradient = transposed stride-2 conv over the whole row + the
    // skip-window conv over the crop window: the window part is computed EARLY on a side stream, the row-wide part
    // adds it inside the window.)  acc_len == 0 is normalised by launch_conv to "the whole row".
    int acc_lo; unsigned acc_len;
    // Fused 2x upsampling of the dst0 output (UnetAudioSeparator.py:109-118, InterpolationLayer.py:19-39): when the launch
    // ends in the split-K epilogue kernel, that kernel also writes ups_y[b][c][0 .. ups_tup) = {y[i], interp(y[i], y[i+1])}
    // (ups_w: learned per-channel weights before the sigmoid, or null = linear; ups_tup = 2n - 1 with context, 2n without)
    // -- the same arithmetic as upsample_vec_kernel, which the caller then does not launch (conv_last_fused_ups()).
    float* ups_y; long long ups_bs; int ups_pitch; int ups_tup; const float* ups_w;
    // ... and its adjoint for the dst1 columns of an up conv's input gradient (linear interpolation only): instead of
    // storing d_up, the split-K epilogue writes ubw_dz[b][c][i] = (d_up[2i] + wa d_up[2i+1] + 1/2 d_up[2i-1]) *
    // LeakyReLU'(ubw_x[b][c][i]) -- the arithmetic of upsample_bwd_vec_kernel; rows of ubw_dz / ubw_x share one geometry.
    float* ubw_dz; const float* ubw_x; long long ubw_bs; int ubw_pitch; int ubw_n;
    // bf16 storage (compute_dtype = 1, section "bf16 activations in HBM" of DESIGN.md): xbf -- src0 / src1 point at bf16
    // elements; obf -- dst0 / dst1 / msk0 / msk1 / dec point at bf16 elements.  All pitches, batch strides and offsets of a
    // tensor are in ITS elements.  Only the bf16 kernels (wun_bf16.hip) read these.
    int xbf, obf;
};
// Conv tile variants [42, 56) were the register-window conv tiles of round 4 (per launch on par with the DMA-staged
// conv_mfma_kernel tiles, 0.5 % slower per step; removed in round 5).  The index range stays RETIRED -- never a legal
// choice -- so that the in-workgroup split-K tiles keep their numbers (56 ..) and committed tuning tables stay valid.
#define WUN_FIRST_RETIRED_VARIANT 42
#define WUN_NUM_RETIRED_VARIANTS 14
// did the last launch_conv() on this thread write the fused upsampled copy?  (only split-K launches do)
int conv_last_fused_ups();
__host__ __device__ static inline bool conv_acc_at(const ConvArgs& a, int pos) { return (unsigned)(pos - a.acc_lo) < a.acc_len; }

// Weight/bias gradient launch:  P[split][ (k*C + c)*N + n ] and bias row P[split][KW*C*N + n]
//   = sum over this split's (b, q-tile) units of in[b][c][q*SI + k - shift] * dz[b][n][q]
struct WgradArgs {
    const float* src0; const float* src1;
    long long bs0, bs1;
    int pitch0, pitch1;
    int off0, off1;
    int C0, C1;
    int Tin;
    int shift;
    int KW;
    int loader;      // DIRECT (SI = 1) / DEINT (SI = 2)
    const float* dz; long long dzbs; int dzpitch; int N; int Tq;
    float* out;      // direct: final [K][Cin][Cout](+bias row) block; else base of the tile-major split partials
    int direct;      // 1: single split, write the final layout
    int split_base;  // index of this launch's first split in the partial buffer
    int nsplit; int units_per_split; int nQT; int B;
    int ablate;      // debugging switches (only read when built with -DWUN_ABLATION)
    int force_mtw, force_nw;   // autotuner: geometry overrides (0 = heuristic)
    int bf16;        // speed mode: operands rounded to bf16 in LDS, v_mfma_f32_16x16x32_bf16 (wun_wgrad_bf16.hip)
    int win;         // exact fp32, register-window form (wun_wgrad_win.hip) instead of the LDS-tiled wgrad_mfma_kernel
    int sbf;         // bf16 kernel: src0 / src1 AND dz point at bf16 elements (pitches / strides / offsets in elements)
};

// tile geometry of an exact-fp32 weight-gradient launch
struct WgradGeom { int MTW, NW, nMG, nNG, TK, XP, ZP, nChMax, ONESP, XW4; size_t lds; };
WgradGeom wgrad_geom(const WgradArgs& a);
// register-window weight gradient (wun_wgrad_win.hip; WgradArgs.win): aligned 16-byte operand reads, DMA staging,
// split partials in the final layout
bool wgrad_win_supported(const WgradArgs& a);
long long wgrad_win_partial_floats(const WgradArgs& a);
int wgrad_win_units(const WgradArgs& a);
int wgrad_win_tiles(const WgradArgs& a);
hipError_t launch_wgrad_win(const WgradArgs& a, hipStream_t s);
hipError_t launch_wgrad_win_reduce(const WgradArgs& a, const float* partial, int nsplit, float* out_w, float* out_b,
                                   hipStream_t s);
// bf16 speed mode (wun_wgrad_bf16.hip): own tiling -- MFMA rows = 16 input channels of one tap; a workgroup holds
// 4*MTW slots = NCB channel blocks x KW taps + the bias slot
struct WgradBfGeom { int MTW, NW, NCB, nCB, nMG, nNG, TK, XW4, XROWS, ZPe; size_t lds; };
bool wgrad_bf16_supported(const WgradArgs& a);
WgradBfGeom wgrad_bf16_geom(const WgradArgs& a);
hipError_t launch_wgrad_bf16(const WgradArgs& a, hipStream_t s);
hipError_t launch_wgrad_bf16_reduce(const WgradArgs& a, const float* partial, int nsplit, float* out_w, float* out_b,
                                    hipStream_t s);

// Narrow weight gradient (wun_narrow.hip): audio-input conv / output head.  Input = virtual concat of two
// NCW sources as in WgradArgs; dz row n = (s, c), s = n / Nper, at dz + s*zss + b*dzbs + c*dzpitch.
struct NarrowWgradArgs {
    const float* src0; const float* src1;
    long long bs0, bs1;
    int pitch0, pitch1, off0, off1, C0, C1;
    int Tin, shift, KW, stride;
    const float* dz; long long zss, dzbs; int dzpitch;
    int N, Nper, Tq, B;
    float* partial;      // [splits][(KW*Ctot + 1) * N]
    int split_base, nsplit, units_per_split, nQT;
    int et;              // element types: bit 0 src0, bit 1 src1, bit 2 dz stored as bf16 (geometry in elements)
    int nrow0;           // streaming form: first dz row of this launch (set by the launcher)
};

struct ConvChoice { int variant; int ksplit; };
// bf16 conv (wun_bf16.hip): tile choices share the tuning tables with the fp32 variants, offset by kBf16VariantBase
static const int kBf16VariantBase = 1000;
bool conv_bf16_choice_ok(const ConvArgs& a, int variant);
int conv_bf16_list_candidates(const ConvArgs& a, ConvChoice* out, int maxn);
struct WgradChoice { int mtw, nw, nsplit[2]; };   // per layer: shared tile geometry, split count per part

struct UpsampleArgs {
    const float* x; long long xbs; int xpitch; int n;     // [B][C][n]
    float* y; long long ybs; int ypitch; int tup;          // [B][C][tup]
    const float* w;                                        // interp weights [C] or null (linear)
    int C; int B; int context;
    int bf;                                                // x and y hold bf16 elements
};

struct UpsampleBwdArgs {
    const float* dy; long long ybs; int ypitch; int tup;   // gradient wrt upsampled tensor
    const float* x; long long xbs; int xpitch; int n;      // forward input (post-activation)
    float* dz;                                             // out: dL/d(pre-activation of x), geometry of x
    const float* w; float* dw;                             // interp weights / their gradient (or null)
    float* dw_partial;                                     // [B][C] scratch of the two-stage interp-weight gradient (or null: one workgroup per channel)
    int C; int B; int context;
    int bf;                                                // dy, x and dz hold bf16 elements
};

struct HeadArgs {
    const float* mix_ncw; long long mbs; int mpitch; int moff_feat; int moff_diff; // crop offsets
    const float* feat; long long fbs; int fpitch;          // last up-conv output [B][F][Tfeat]
    const float* Wh;                                       // head params: per source {kernel [Ko][C+F][C], bias [C]}
    int C, F, S, Sh, Ko, padl;
    int Tfeat, Tout, B;
    int tanh_act, difference, training;
    float* out;                                            // [S][B][Tout][C]
    // backward only
    const float* tgt;                                      // [S][B][Tout][C]
    float* dpre; long long dps; long long dpbs; int dppitch; // [Sh][B][C][ToutP]
    float* dzfeat;                                         // geometry of feat
    float* loss_partial;                                   // [gridDim.x]
    float gscale;                                          // 2 / (S*B*Tout*C)
    int featbf;                                            // feat and dzfeat hold bf16 elements (fbs / fpitch in elements)
};

// bf16 mode: one conv's weights, fp32 [KW][C][N] (from the parameter arena or a transposed copy in
// the workspace) -> packed bf16 image [KW][C8p][Npad][8] in the workspace
struct PackDesc {
    long long src_off;   // floats, into params (src_in_ws = 0) or the workspace (1)
    long long dst_off;   // floats, into the workspace (the image holds KW*C8p*Npad*8 bf16 = half as many floats)
    int KW, C, N, C8p, Npad;
    int src_in_ws;
};

struct WtDesc {      // mode 0: dst[j][n][c] = src[k_last - j*k_step][c][n]
                     // mode 1: dst[j][n][p][c] = src[k_last - 2j + p][c][n] (0 if tap >= k_step (= KW))
    long long src_off;   // into params
    long long dst_off;   // into workspace
    int J, C, N, k_last, k_step;
    int mode;
};

// ---- bf16 storage helpers (device): a tensor element type ET is float or bf16_t; values are converted to float on load
// and rounded to nearest-even (v_cvt_pk_bf16_f32) on store.  ET = float compiles to the plain accesses.
typedef unsigned short bf16_t;
typedef float wun_f32x4 __attribute__((ext_vector_type(4)));
typedef float wun_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned wun_u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 wun_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf_pack2(float lo, float hi) {       // two fp32 -> packed bf16 pair (RNE)
    return __builtin_bit_cast(unsigned, __builtin_convertvector((wun_f32x2){lo, hi}, wun_bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf_hi(unsigned v) { return __builtin_bit_cast(float, v & 0xFFFF0000u); }
template <typename ET> __device__ __forceinline__ float ld1(const ET* p, long long i);
template <> __device__ __forceinline__ float ld1<float>(const float* p, long long i) { return p[i]; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p, long long i) { return __builtin_bit_cast(float, (unsigned)p[i] << 16); }
template <typename ET> __device__ __forceinline__ void st1(ET* p, long long i, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, long long i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void st1<bf16_t>(bf16_t* p, long long i, float v) { p[i] = (bf16_t)(bf_pack2(v, 0.f) & 0xFFFFu); }
// four consecutive elements; p + i must be aligned to 4 elements (16 bytes fp32 / 8 bytes bf16)
template <typename ET> __device__ __forceinline__ wun_f32x4 ld4(const ET* p, long long i);
template <> __device__ __forceinline__ wun_f32x4 ld4<float>(const float* p, long long i) { return *reinterpret_cast<const wun_f32x4*>(p + i); }
template <> __device__ __forceinline__ wun_f32x4 ld4<bf16_t>(const bf16_t* p, long long i) {
    const wun_u32x2 v = *reinterpret_cast<const wun_u32x2*>(p + i);
    return (wun_f32x4){bf_lo(v[0]), bf_hi(v[0]), bf_lo(v[1]), bf_hi(v[1])};
}
template <typename ET> __device__ __forceinline__ void st4(ET* p, long long i, wun_f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, long long i, wun_f32x4 v) { *reinterpret_cast<wun_f32x4*>(p + i) = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, long long i, wun_f32x4 v) {
    *reinterpret_cast<wun_u32x2*>(p + i) = (wun_u32x2){bf_pack2(v[0], v[1]), bf_pack2(v[2], v[3])};
}

// ---- launchers (wun_kernels.hip) ---------------------------------------------------
size_t conv_lds_bytes(const ConvArgs& a, int variant);
int  conv_pick_variant(const ConvArgs& a);
hipError_t launch_conv(const ConvArgs& a, float* part, long long part_cap, hipStream_t s);
double conv_flops(const ConvArgs& a);        // useful FLOPs (2*MACs) of the launch
long long conv_natural_wgs_phase2(const ConvArgs& a);
int conv_list_candidates(const ConvArgs& a, long long part_cap, ConvChoice* out, int maxn);
bool conv_choice_ok(const ConvArgs& a, long long part_cap, int variant, int ksplit);
int conv_num_variants();
int wgrad_max_units(const WgradArgs& a);

int  wgrad_pick_nsplit(const WgradArgs& a);
hipError_t launch_wgrad(const WgradArgs& a, hipStream_t s);
void wgrad_resolved_geom(const WgradArgs& a, int& mtw, int& nw);
long long wgrad_partial_floats(const WgradArgs& a);
hipError_t launch_wgrad_reduce(const WgradArgs& a, const float* partial, int nsplit, float* out_w, float* out_b,
                               hipStream_t s);
hipError_t launch_upsample(const UpsampleArgs& a, hipStream_t s);
hipError_t launch_upsample_bwd(const UpsampleBwdArgs& a, hipStream_t s);
hipError_t launch_interp_grad(const UpsampleBwdArgs& a, hipStream_t s);               // dw of the learned interpolation weights
hipError_t launch_head_fwd_off(const HeadArgs& a, const long long* hoff, hipStream_t s);
int head_bwd_blocks(const HeadArgs& a);
hipError_t launch_head_bwd_off(const HeadArgs& a, const long long* hoff, hipStream_t s);
hipError_t launch_loss_finish(const float* partial, int n, float scale, float* loss, hipStream_t s);
hipError_t launch_btc_to_ncw(const float* src, float* dst, int B, int T, int C, int pitch,
                             hipStream_t s);
hipError_t launch_make_wt(const float* params, float* ws, const WtDesc* dev_descs, int ndesc,
                          int max_elems, hipStream_t s);
hipError_t launch_adam(float* p, const float* g, float* m, float* v, long long n, float lr_t,
                       float b1, float b2, float eps, float gscale, hipStream_t s);
hipError_t launch_fill(float* p, long long n, float val, hipStream_t s);
hipError_t launch_mfma_probe(const float* a, const float* b, float* d, hipStream_t s);
void prof_begin();
std::string prof_end();
void prof_scope_begin(const char* name, double flops, hipStream_t s, const char* tag, double bytes = 0.0);   // HIP-event bracket of the next launch (bytes: algorithmic HBM bytes of a bandwidth-bound launch)
void prof_scope_end(hipStream_t s);

// ---- narrow weight gradients (wun_narrow.hip) ----
bool narrow_wgrad_supported(const NarrowWgradArgs& a);
bool narrow_wgrad_uses_lds(const NarrowWgradArgs& a);      // the LDS-staged kernel (else the streaming form)
int narrow_wgrad_units(const NarrowWgradArgs& a);
int narrow_wgrad_pick_nsplit(const NarrowWgradArgs& a);
long long narrow_wgrad_partial_floats(const NarrowWgradArgs& a);
hipError_t launch_narrow_wgrad(NarrowWgradArgs a, hipStream_t s);
hipError_t launch_narrow_wgrad_reduce(const NarrowWgradArgs& a, const float* partial, int nsplit, float* grads,
                                      const long long* woff, const long long* boff, hipStream_t s);

// ---- bf16-MFMA speed mode (wun_bf16.hip) ----
// 8-channel groups of a packed bf16 weight image for C input channels: the conv kernel reads whole stages of
// 4, 8 or 12 groups (1..3 chunks of 32 channels), so the image is zero-padded to the next multiple of each
static inline int bf16_image_groups(int C) {
    const int g = (C + 7) / 8;
    int best = 0;
    for (int m = 4; m <= 12; m += 4) { const int r = (g + m - 1) / m * m; best = r > best ? r : best; }
    return best;
}
bool conv_bf16_supported(const ConvArgs& a);
hipError_t launch_conv_bf16(const ConvArgs& a, hipStream_t s);
bool first_conv_supported(const ConvArgs& a);
hipError_t launch_first_conv(const ConvArgs& a, hipStream_t s);                 // audio-input conv of the bf16 mode
hipError_t launch_cast_rows_bf16(const float* src, void* dst, long long rows, int T, long long spitch, long long dpitch, hipStream_t s);
hipError_t launch_pack_bf16(const float* params, float* ws, const PackDesc* dev_descs, int ndesc, long long max_items,
                            hipStream_t s);
hipError_t launch_mfma_bf16_probe(const float* a, const float* b, float* d, hipStream_t s);

}  // namespace wun
