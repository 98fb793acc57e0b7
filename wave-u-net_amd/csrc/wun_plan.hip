// Host side of libwun.so: static plan (the equivalent of the reference building its TF
// graph once, /root/reference/Training.py:47) and the C ABI declared in include/wun.h.
#include "../../include/wun.h"
#include "wun_internal.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

using namespace wun;

namespace wun {
hipError_t launch_make_wt_one(const float* src, float* dst, WtDesc d, hipStream_t s);
}

static thread_local std::string g_err;
static bool g_profiling = false;   // while wun_profile_* is active everything runs on the caller's stream

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return fail(WUN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));      \
    } while (0)

// ---------------------------------------------------------------------------------------
// shapes (UnetAudioSeparator.py:34-83, Utils.py:104-123)
// ---------------------------------------------------------------------------------------
static int check_config(const wun_config* c) {
    if (!c) return fail(WUN_ERR_INVALID, "config is null");
    if (c->num_layers < 1 || c->num_layers > 24) return fail(WUN_ERR_INVALID, "num_layers out of range");
    if (c->num_initial_filters < 1) return fail(WUN_ERR_INVALID, "num_initial_filters < 1");
    if (c->filter_size < 1 || c->merge_filter_size < 1 || c->output_filter_size < 1 || c->input_filter_size < 1)
        return fail(WUN_ERR_INVALID, "filter sizes must be >= 1");
    if (c->filter_size > 15 || c->merge_filter_size > 15)
        return fail(WUN_ERR_UNSUPPORTED, "filter_size / merge_filter_size > 15 not supported by the gfx950 kernels");
    if (c->upsampling != 0 && c->upsampling != 1) return fail(WUN_ERR_UNSUPPORTED, "upsampling must be linear|learned");
    if (c->output_type != 0 && c->output_type != 1) return fail(WUN_ERR_UNSUPPORTED, "output_type must be direct|difference");
    if (c->output_activation != 0 && c->output_activation != 1)
        return fail(WUN_ERR_UNSUPPORTED, "output_activation must be tanh|linear");   // UnetAudioSeparator.py:136
    if (c->num_channels != 1 && c->num_channels != 2) return fail(WUN_ERR_INVALID, "num_channels must be 1 or 2");
    if (c->num_sources < 1 || c->num_sources > 4) return fail(WUN_ERR_UNSUPPORTED, "num_sources must be 1..4");
    if (c->output_type == 1 && c->num_sources < 2) return fail(WUN_ERR_INVALID, "difference output needs >= 2 sources");
    if (c->compute_dtype != 0 && c->compute_dtype != 1) return fail(WUN_ERR_UNSUPPORTED, "compute_dtype must be 0 (f32) or 1 (bf16)");
    {
        // the head kernels keep every source's output filter in LDS (default 64 KiB dynamic limit)
        const long long sh = c->output_type == 0 ? c->num_sources : c->num_sources - 1;
        const long long cin = (long long)c->num_channels + c->num_initial_filters;
        const long long head_lds = 4 * sh * ((long long)c->output_filter_size * cin * c->num_channels + c->num_channels);
        if (head_lds > 64 * 1024)
            return fail(WUN_ERR_UNSUPPORTED, "output_filter_size x (num_channels + num_initial_filters) too large for the head kernels' LDS");
    }
    return WUN_OK;
}

extern "C" int wun_get_padding(const wun_config* cfg, int64_t desired, int64_t* in_frames, int64_t* out_frames) {
    int rc = check_config(cfg);
    if (rc) return rc;
    if (!in_frames || !out_frames) return fail(WUN_ERR_INVALID, "null output pointer");
    if (!cfg->context) { *in_frames = desired; *out_frames = desired; return WUN_OK; }   // :83
    double rem = (double)desired;                       // :43
    rem = rem - cfg->output_filter_size + 1;            // :46
    for (int i = 0; i < cfg->num_layers; ++i) {         // :49-51
        rem = rem + cfg->merge_filter_size - 1;
        rem = (rem + 1.0) / 2.0;
    }
    const int64_t x = (int64_t)std::ceil(rem);          // :54
    if (x < 2) return fail(WUN_ERR_INVALID, "get_padding: bottleneck feature map < 2 (reference assert)");
    int64_t out = x, inp = x + cfg->filter_size - 1;    // :58-62
    for (int i = 0; i < cfg->num_layers; ++i) {         // :65-73
        out = 2 * out - 1;
        out = out - cfg->merge_filter_size + 1;
        inp = 2 * inp - 1;
        inp += (i < cfg->num_layers - 1 ? cfg->filter_size : cfg->input_filter_size) - 1;
    }
    out = out - cfg->output_filter_size + 1;            // :76
    *in_frames = inp;
    *out_frames = out;
    return WUN_OK;
}

// ---------------------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------------------
// A workspace tensor [B][C][pitch] of fp32 (eb = 4) or bf16 (eb = 2: the activations and their gradients of the bf16
// mode) elements; `off` is in FLOATS from the workspace base, pitch and bs are in ELEMENTS; rows are 16-byte aligned.
struct Buf { long long off = -1; int C = 0; int T = 0; int pitch = 0; long long bs = 0; int eb = 4; };
struct ConvLayer { long long woff = 0, boff = 0; int KW = 0, Cin = 0, Cout = 0;
                   long long wt_full = -1, wt_ph[2] = {-1, -1}, wt_ph2 = -1; int Jp[2] = {0, 0}; int J0 = 0;
                   // dedup plans: the two-phase image of the filter SHIFTED by one tap (W''[k + 1] = W[k], W''[0] = 0), for the
                   // odd-window input gradient: its outputs start at an odd row position; with the shifted filter the launch
                   // starts one position earlier, on a 16-byte boundary, and takes the vector epilogue
                   long long wt_ph2s = -1; int J0s = 0; };
// tc / cs: length / start of the centre crop the skip connection takes (Utils.py:104-123), in conv-output positions.
// Round 6 (dedup plans): the crop window split by the parity of the ABSOLUTE conv position -- even positions are elements
// of the decimated stream (computed once, by the stride-2 launch), odd positions get their own stride-2 launches:
// t_ev0 / n_even, t_odd0 / n_odd = first position and count of each parity inside [cs, cs + tc).
struct DownShape { int cin, cout, t_in, t_conv, t_dec, tc, cs, t_ev0, n_even, t_odd0, n_odd; };
struct UpShape { int c_skip, c_cur, cout, t_cur, t_up, t_conv, crop_start; };

struct wun_plan {
    wun_config cfg;
    int B = 0, Tin = 0, Tout = 0;
    int L = 0, C = 0, S = 0, Sh = 0;
    bool same = false;
    std::vector<wun_tensor_info> tensors;
    long long arena = 0, ws = 0;
    std::vector<DownShape> dsh;
    std::vector<UpShape> ush;
    int t_b_in = 0, t_b = 0, c_b = 0;
    int t_feat = 0, in_crop_start = 0, mix_diff_off = 0;
    std::vector<ConvLayer> down, up, head;
    ConvLayer bott;
    std::vector<long long> interp;
    Buf mix_ncw, bott_out, dz_bott;
    // bf16 mode, output layer too wide for ONE narrow weight-gradient launch ((C + F) * Sh * C > 256: the deep variant):
    // its weight gradient runs on the bf16 MFMA kernel, which reads bf16 rows only -- bf16 copies of the audio (made in the
    // forward pass) and of the head's d(pre-activation) (made after head_bwd_kernel); both are a few rows
    bool head16 = false;
    Buf mix16;
    long long dpre16_off = -1; int dp16_pitch = 0;
    std::vector<Buf> dec, skip, ups, upo, dz_dec, dz_skip, d_ups, dz_upo;
    // Round 6, context plans of the exact-fp32 mode ("dedup"): the reference's decimated stream is a SLICE of the encoder
    // output (UnetAudioSeparator.py:98-100: one tensor, one rounding).  The stride-2 launch of a down level writes its
    // outputs into dec[i] AND into the even positions of the skip window; a second stride-2 launch computes only the odd
    // window positions (rounds 1 - 5 ran a stride-1 conv over the whole window: every even window position was computed
    // twice, 7.8 % of the step's FLOPs).  Backward: an up level's input gradient splits the window's gradient by parity --
    // even part stored into dz_dec[i] (the transposed conv that fills the rest of dz_dec[i] later ADDS inside that range),
    // odd part compact in dz_odd[i] -- and the window's input gradient / weight gradient run over the odd positions only.
    bool dedup = false;
    std::vector<Buf> dz_odd;
    long long dpre_off = -1; int dp_pitch = 0;
    long long partial_off = -1, partial_floats = 0;
    long long loss_partial_off = -1;
    std::vector<long long> interp_partial_off;
    long long conv_part_off = -1, conv_part_floats = 0;
    std::vector<WtDesc> wt;
    WtDesc* dev_wt = nullptr;
    int wt_max = 0;
    double fwd_flops = 0, bwd_flops = 0, fwd_dense = 0, fwd_unique = 0, bwd_unique = 0;
    // second HIP stream: independent launches (weight gradients vs the input-gradient chain;
    // skip-window convs vs the decimating convs) run concurrently so that one kernel's tail and
    // epilogue overlap another kernel's MFMA phase
    // autotuner state: per-launch choices in launch order (forward / backward), filled by wun_plan_tune
    mutable int tune_mode = 0;                  // 0 = heuristics, 1 = measuring, 2 = tuned
    mutable std::vector<ConvChoice> conv_fwd, conv_bwd;
    mutable std::vector<WgradChoice> wg_bwd;
    mutable size_t ci = 0, wi = 0;
    mutable bool in_bwd = false;
    mutable hipEvent_t tev0 = nullptr, tev1 = nullptr;
    mutable hipStream_t side = nullptr, side2 = nullptr;
    mutable std::vector<hipEvent_t> events;
    mutable size_t ev_next = 0;
    // transposed weight copies for the input-gradient convs: produced on the side stream during the
    // forward pass (training mode) so that the backward pass does not start with a 50 us transpose
    mutable hipEvent_t wt_ev = nullptr;
    mutable bool wt_ready = false;
    mutable std::vector<hipEvent_t> skip_ev;             // forward: skip window i is complete (deferred window convs)
    mutable std::vector<hipEvent_t> win_ev;              // backward: the early skip-window input gradient of level i is complete
    // bf16-MFMA speed mode (cfg.compute_dtype == 1): packed bf16 images of the conv weights in the
    // workspace, keyed by where the fp32 weights of a launch live (params arena / transposed copy in ws)
    struct BfImg { long long off; int c8p, npad; };
    bool bf16 = false;
    std::map<std::pair<int, long long>, BfImg> bf_img;       // (1 = in workspace, float offset) -> image
    std::vector<PackDesc> pack;                              // forward images first, then the dgrad images
    int npack_fwd = 0;
    long long pack_max = 0;
    PackDesc* dev_pack = nullptr;
    mutable const float* cur_params = nullptr;
    mutable const float* cur_ws = nullptr;
};

static long long bump(long long& cur, long long n) {
    const long long off = (cur + 63) / 64 * 64;
    cur = off + n;
    return off;
}

static Buf make_buf(long long& cur, int B, int C, int T, const char* name = nullptr, int idx = 0, int eb = 4) {
    Buf b;
    const int per16 = 16 / eb;                                        // elements per 16 bytes
    b.C = C; b.T = T; b.pitch = (T + per16 - 1) / per16 * per16; b.eb = eb;
    b.bs = (long long)C * b.pitch;
    b.off = bump(cur, ((long long)B * b.bs * eb + 3) / 4);
    if (name != nullptr && getenv("WUN_DUMP_LAYOUT") != nullptr)      // (tools/ws_diff.py: workspace map for debugging)
        fprintf(stderr, "[wun-layout] %s %d off=%lld B=%d C=%d T=%d pitch=%d eb=%d\n", name, idx, b.off, B, C, T, b.pitch, eb);
    return b;
}

// Can this plan run the bf16 mode (compute_dtype = 1)?  Every conv from level 1 on must be served by the bf16 MFMA
// kernels -- the activations then live in HBM as bf16 and nothing but those kernels (and the element-type aware
// elementwise / head / audio-input kernels) reads them: channel counts in whole groups of 8 (num_initial_filters % 8 == 0;
// every shipped config: 24, the deep variant 48), no tap-less conv phase (a 1-tap filter with context), rows and tensors
// inside the kernels' 32-bit offsets.  A plan that does not qualify is the exact-fp32 plan (wun_plan_query reports it).
static bool bf16_plan_ok(const wun_config* c, int B, const std::vector<DownShape>& dsh, int t_b,
                         const std::vector<UpShape>& ush, int c_b, int Sh) {
    if ((c->num_initial_filters & 7) != 0) return false;
    if (c->context && (c->filter_size < 2)) return false;
    if (c->filter_size > 15 || c->merge_filter_size > 15 || c->output_filter_size > 15) return false;
    (void)B;
    // a row of `t` elements (+ halo) inside the kernels' 23-bit row elements, an excerpt's tensor inside 32-bit byte offsets
    auto fits = [](long long ch, long long t) { return t + 64 < (1ll << 23) && ch * ((t + 15) & ~7ll) * 2 < (1ll << 31); };
    for (const DownShape& d : dsh)
        if (!fits(d.cout, d.t_conv) || !fits(d.cin, d.t_in)) return false;
    if (!fits(c_b, t_b)) return false;
    // up path (these were only checked at launch): the two sources of an up conv -- skip window and upsampled
    // tensor, twice the channels of the down level at the same length --, its output and the gradient tensors of the same shapes
    for (const UpShape& u : ush) {
        if (!fits(u.c_skip, u.t_up) || !fits(u.c_cur, u.t_up) || !fits(u.cout, u.t_conv)) return false;
        if ((u.c_skip & 7) != 0) return false;                  // an 8-channel group must not straddle the two sources
    }
    // output head: the narrow weight-gradient kernels (all sources in one launch, or one launch per source), or -- even
    // channel counts -- the bf16 MFMA weight gradient on bf16 copies of the audio and the head's d(pre-activation)
    if (Sh > 0) {
        const int C = c->num_channels, F = c->num_initial_filters;
        NarrowWgradArgs t;
        memset(&t, 0, sizeof(t));
        t.C0 = C; t.C1 = F; t.KW = c->output_filter_size; t.stride = 1; t.N = Sh * C; t.Nper = C;
        bool ok = narrow_wgrad_supported(t);
        if (!ok && (C & 1) == 0) {
            WgradArgs w;
            memset(&w, 0, sizeof(w));
            w.C0 = C; w.C1 = F; w.KW = c->output_filter_size; w.N = C;
            ok = wgrad_bf16_supported(w);
        }
        if (!ok) { t.N = t.Nper = C; ok = narrow_wgrad_supported(t); }
        if (!ok) return false;
    }
    // audio-input conv: the narrow kernels read fp32 audio against bf16 gradients
    {
        NarrowWgradArgs t;
        memset(&t, 0, sizeof(t));
        t.C0 = c->num_channels; t.KW = c->filter_size; t.stride = c->context ? 2 : 1; t.N = t.Nper = dsh.empty() ? 0 : dsh[0].cout;
        if (!narrow_wgrad_supported(t)) return false;
    }
    return true;
}

static void crop_offsets(int from, int to, int& start, int& end) {   // Utils.py:120-121
    const int diff = from - to;
    start = diff / 2;
    end = diff - start;
}

static void add_tensor(wun_plan* p, const std::string& name, std::vector<int64_t> shape, long long& cur,
                       long long* off_out) {
    wun_tensor_info t;
    memset(&t, 0, sizeof(t));
    snprintf(t.name, sizeof(t.name), "%s", name.c_str());
    t.offset = cur;
    t.ndim = (int)shape.size();
    long long n = 1;
    for (size_t i = 0; i < shape.size(); ++i) { t.shape[i] = shape[i]; n *= shape[i]; }
    *off_out = cur;
    cur += n;
    p->tensors.push_back(t);
}

static void add_conv(wun_plan* p, int& counter, int K, int cin, int cout, long long& cur, ConvLayer* out) {
    std::string base = counter == 0 ? "separator/conv1d" : "separator/conv1d_" + std::to_string(counter);
    ++counter;
    out->KW = K; out->Cin = cin; out->Cout = cout;
    add_tensor(p, base + "/kernel", {K, cin, cout}, cur, &out->woff);
    add_tensor(p, base + "/bias", {cout}, cur, &out->boff);
}

static WgradArgs wgrad_shape_only(int B, int C0, int C1, int KW, int loader, int N, int Tq) {
    WgradArgs w;
    memset(&w, 0, sizeof(w));
    w.B = B; w.C0 = C0; w.C1 = C1; w.KW = KW; w.loader = loader; w.N = N; w.Tq = Tq;
    return w;
}

extern "C" int wun_plan_create(const wun_config* cfg, int64_t batch, int64_t input_frames, wun_plan** out) {
    int rc = check_config(cfg);
    if (rc) return rc;
    if (!out) return fail(WUN_ERR_INVALID, "out is null");
    if (batch < 1 || input_frames < 1 || input_frames > (1ll << 30)) return fail(WUN_ERR_INVALID, "bad batch / input_frames");
    wun_plan* p = new (std::nothrow) wun_plan();
    if (!p) return fail(WUN_ERR_NOMEM, "out of host memory");
    p->cfg = *cfg;
    p->B = (int)batch; p->Tin = (int)input_frames;
    const int L = p->L = cfg->num_layers, F = cfg->num_initial_filters;
    const int Kd = cfg->filter_size, Ku = cfg->merge_filter_size, Ko = cfg->output_filter_size;
    const int C = p->C = cfg->num_channels;
    p->S = cfg->num_sources;
    p->Sh = cfg->output_type == 0 ? p->S : p->S - 1;
    const bool same = p->same = !cfg->context;

    // ---- shape walk of get_output (UnetAudioSeparator.py:97-142) ----
    auto conv_len = [&](int t, int k) { return same ? t : t - k + 1; };
    p->dsh.resize(L);
    int t = p->Tin, cin = C;
    for (int i = 0; i < L; ++i) {
        DownShape& d = p->dsh[i];
        d.cin = cin; d.cout = F + F * i; d.t_in = t; d.t_conv = conv_len(t, Kd);
        if (d.t_conv < 1) { delete p; return fail(WUN_ERR_INVALID, "input too short for the down path"); }
        d.t_dec = (d.t_conv + 1) / 2;                       // [:, ::2, :]  :100
        t = d.t_dec; cin = d.cout;
    }
    p->c_b = F + F * L; p->t_b_in = t; p->t_b = conv_len(t, Kd);
    if (p->t_b < 1) { delete p; return fail(WUN_ERR_INVALID, "input too short for the bottleneck conv"); }
    p->ush.resize(L);
    t = p->t_b; int ccur = p->c_b;
    for (int j = 0; j < L; ++j) {
        UpShape& u = p->ush[j];
        const DownShape& d = p->dsh[L - 1 - j];
        u.c_skip = d.cout; u.c_cur = ccur; u.cout = F + F * (L - j - 1);
        u.t_cur = t; u.t_up = cfg->context ? 2 * t - 1 : 2 * t;      // :115/:117, InterpolationLayer.py:32
        if (same && d.t_conv != u.t_up) {                             // :121
            delete p;
            return fail(WUN_ERR_INVALID, "same-padding input length must be divisible by 2^num_layers (reference assert UnetAudioSeparator.py:121)");
        }
        if (d.t_conv < u.t_up) { delete p; return fail(WUN_ERR_INVALID, "crop with negative difference (Utils.py:117)"); }
        int ce;
        crop_offsets(d.t_conv, u.t_up, u.crop_start, ce);
        u.t_conv = conv_len(u.t_up, Ku);
        if (u.t_conv < 1) { delete p; return fail(WUN_ERR_INVALID, "input too short for the up path"); }
        t = u.t_conv; ccur = u.cout;
    }
    for (int i = 0; i < L; ++i) {
        DownShape& d = p->dsh[i];
        d.tc = p->ush[L - 1 - i].t_up;
        d.cs = p->ush[L - 1 - i].crop_start;
        d.t_ev0 = d.cs + (d.cs & 1); d.t_odd0 = d.cs + 1 - (d.cs & 1);
        d.n_even = d.cs + d.tc > d.t_ev0 ? (d.cs + d.tc - d.t_ev0 + 1) / 2 : 0;
        d.n_odd = d.tc - d.n_even;
    }
    p->t_feat = t;
    if (p->Tin < p->t_feat) { delete p; return fail(WUN_ERR_INVALID, "crop with negative difference (Utils.py:117)"); }
    { int ce; crop_offsets(p->Tin, p->t_feat, p->in_crop_start, ce); }     // :127
    p->Tout = conv_len(p->t_feat, Ko);
    if (p->Tout < 1) { delete p; return fail(WUN_ERR_INVALID, "input too short for the output layer"); }
    { int s2, e2; crop_offsets(p->t_feat, p->Tout, s2, e2); p->mix_diff_off = p->in_crop_start + s2; }  // OutputLayer.py:20

    // ---- variable arena in TF creation order ----
    long long cur = 0;
    int counter = 0;
    p->down.resize(L); p->up.resize(L); p->head.resize(p->Sh);
    for (int i = 0; i < L; ++i) add_conv(p, counter, Kd, p->dsh[i].cin, p->dsh[i].cout, cur, &p->down[i]);
    add_conv(p, counter, Kd, p->dsh[L - 1].cout, p->c_b, cur, &p->bott);
    p->interp.assign(L, -1);
    for (int j = 0; j < L; ++j) {
        if (cfg->upsampling == 1)
            add_tensor(p, "separator/interp_" + std::to_string(j), {p->ush[j].c_cur}, cur, &p->interp[j]);
        add_conv(p, counter, Ku, p->ush[j].c_skip + p->ush[j].c_cur, p->ush[j].cout, cur, &p->up[j]);
    }
    for (int s = 0; s < p->Sh; ++s) add_conv(p, counter, Ko, C + F, C, cur, &p->head[s]);
    p->arena = cur;

    // ---- workspace ----
    long long w = 0;
    const int B = p->B;
    // bf16 mode: every activation and activation-gradient tensor is born bf16 (the audio itself, the head's
    // d(pre-activation), weights, weight gradients and all scratch stay fp32)
    p->bf16 = cfg->compute_dtype == 1 && bf16_plan_ok(cfg, B, p->dsh, p->t_b, p->ush, p->c_b, p->Sh);
    const int eb = p->bf16 ? 2 : 4;
    // (exact-fp32 mode only.  The bf16 mode was built and measured with it at the end of round 6 -- bf16 epilogues with the
    //  expanded / parity copies, the audio-input conv with a strided output, the emulation restated; all 62 bf16-mode tests
    //  green -- and was 3 - 4 % SLOWER (M4 3.27 -> 3.39 ms): its kernels are bound by the instruction stream around the MFMAs,
    //  the scalar strided epilogues of the odd-position launches cost more than the halved window work saves.  Not adopted.)
    p->dedup = !same && !p->bf16 && getenv("WUN_NO_DEDUP") == nullptr;
    p->mix_ncw = make_buf(w, B, C, p->Tin, "mix_ncw");
    p->dec.resize(L); p->skip.resize(L); p->dz_dec.resize(L); p->dz_skip.resize(L);
    for (int i = 0; i < L; ++i) {
        p->dec[i] = make_buf(w, B, p->dsh[i].cout, p->dsh[i].t_dec, "dec", i, eb);
        p->skip[i] = make_buf(w, B, p->dsh[i].cout, p->dsh[i].tc, "skip", i, eb);
        if (!same) p->dz_dec[i] = make_buf(w, B, p->dsh[i].cout, p->dsh[i].t_dec, "dz_dec", i, eb);
        p->dz_skip[i] = make_buf(w, B, p->dsh[i].cout, p->dsh[i].tc, "dz_skip", i, eb);
    }
    if (p->dedup) {
        p->dz_odd.resize(L);
        for (int i = 0; i < L; ++i) p->dz_odd[i] = make_buf(w, B, p->dsh[i].cout, std::max(p->dsh[i].n_odd, 1), "dz_odd", i, eb);
    }
    p->bott_out = make_buf(w, B, p->c_b, p->t_b, "bott_out", 0, eb);
    p->dz_bott = make_buf(w, B, p->c_b, p->t_b, "dz_bott", 0, eb);
    p->ups.resize(L); p->upo.resize(L); p->d_ups.resize(L); p->dz_upo.resize(L);
    for (int j = 0; j < L; ++j) {
        p->ups[j] = make_buf(w, B, p->ush[j].c_cur, p->ush[j].t_up, "ups", j, eb);
        p->d_ups[j] = make_buf(w, B, p->ush[j].c_cur, p->ush[j].t_up, "d_ups", j, eb);
        p->upo[j] = make_buf(w, B, p->ush[j].cout, p->ush[j].t_conv, "upo", j, eb);
        p->dz_upo[j] = make_buf(w, B, p->ush[j].cout, p->ush[j].t_conv, "dz_upo", j, eb);
    }
    p->dp_pitch = (p->Tout + 3) / 4 * 4;
    p->dpre_off = bump(w, (long long)p->Sh * B * C * p->dp_pitch);
    if (p->bf16 && p->Sh > 0) {
        NarrowWgradArgs t;
        memset(&t, 0, sizeof(t));
        t.C0 = C; t.C1 = F; t.KW = Ko; t.stride = 1; t.N = p->Sh * C; t.Nper = C;
        if (!narrow_wgrad_supported(t) && (C & 1) == 0) {
            p->head16 = true;
            p->mix16 = make_buf(w, B, C, p->Tin, "mix16", 0, 2);
            p->dp16_pitch = (p->Tout + 7) / 8 * 8;
            p->dpre16_off = bump(w, ((long long)p->Sh * B * C * p->dp16_pitch + 1) / 2);
        }
    }
    p->loss_partial_off = bump(w, 1024);
    if (cfg->upsampling == 1) {                                   // [B][C] partials of every interp_<j> gradient (own block per
        p->interp_partial_off.assign(L, -1);                      // level: consecutive levels run on different side streams)
        for (int j = 0; j < L; ++j) p->interp_partial_off[j] = bump(w, (long long)B * p->ush[j].c_cur);
    }
    p->conv_part_floats = 16ll << 20;                       // split-K scratch (64 MiB)
    p->conv_part_off = bump(w, p->conv_part_floats);

    // ---- transposed / tap-flipped weights for the input-gradient convs ----
    auto add_wt = [&](const ConvLayer& cl, int J, int k_last, int k_step) -> long long {
        WtDesc d;
        d.src_off = cl.woff; d.J = J; d.C = cl.Cin; d.N = cl.Cout; d.k_last = k_last; d.k_step = k_step; d.mode = 0;
        d.dst_off = bump(w, (long long)J * cl.Cin * cl.Cout);
        p->wt.push_back(d);
        const long long n = (long long)J * cl.Cin * cl.Cout;
        if (n > p->wt_max) p->wt_max = (int)n;
        return d.dst_off;
    };
    // both output phases of the transposed stride-2 conv side by side: [J0][Cout][2][Cin]
    auto add_wt2 = [&](const ConvLayer& cl, int J0) -> long long {
        WtDesc d;
        d.src_off = cl.woff; d.J = J0; d.C = cl.Cin; d.N = cl.Cout; d.k_last = 2 * (J0 - 1); d.k_step = cl.KW; d.mode = 1;
        const long long n = (long long)J0 * cl.Cin * cl.Cout * 2;
        d.dst_off = bump(w, n);
        p->wt.push_back(d);
        if (n > p->wt_max) p->wt_max = (int)n;
        return d.dst_off;
    };
    for (int i = 1; i < L; ++i) {
        ConvLayer& cl = p->down[i];
        cl.wt_full = add_wt(cl, cl.KW, cl.KW - 1, 1);
        if (!same) {
            for (int ph = 0; ph < 2; ++ph) {
                const int Jp = (cl.KW - ph + 1) / 2;          // taps k = 2j + ph < KW
                cl.Jp[ph] = Jp;
                cl.wt_ph[ph] = add_wt(cl, Jp, 2 * (Jp - 1) + ph, 2);
            }
            cl.J0 = (cl.KW + 1) / 2;
            cl.wt_ph2 = add_wt2(cl, cl.J0);
            if (p->dedup) {
                cl.J0s = (cl.KW + 2) / 2;
                cl.wt_ph2s = add_wt2(cl, cl.J0s);
                p->wt.back().k_last = 2 * (cl.J0s - 1) - 1;      // tap of W behind tap 2 (J0s - 1) of the shifted filter
            }
        }
    }
    p->bott.wt_full = add_wt(p->bott, Kd, Kd - 1, 1);
    for (int j = 0; j < L; ++j) p->up[j].wt_full = add_wt(p->up[j], Ku, Ku - 1, 1);

    // ---- bf16 mode: packed weight images (every conv with >= 8 input channels; the audio-input conv and the
    // output head stay exact fp32) ----
    if (p->bf16) {
        auto add_img = [&](int in_ws, long long src_off, int K, int Cc, int Nn, int slack = 0) {
            if (Cc < 8 || K < 1) return;
            PackDesc d;
            d.src_off = src_off; d.src_in_ws = in_ws; d.KW = K; d.C = Cc; d.N = Nn;
            d.C8p = bf16_image_groups(Cc); d.Npad = (Nn + slack + 63) / 64 * 64;
            const long long items = (long long)K * d.C8p * d.Npad;
            d.dst_off = bump(w, items * 4);                     // 8 bf16 = 4 floats per item
            p->pack.push_back(d);
            p->pack_max = std::max(p->pack_max, items);
            p->bf_img[{in_ws, src_off}] = wun_plan::BfImg{d.dst_off, d.C8p, d.Npad};
        };
        for (int i = 0; i < L; ++i) add_img(0, p->down[i].woff, Kd, p->down[i].Cin, p->down[i].Cout);
        add_img(0, p->bott.woff, Kd, p->bott.Cin, p->bott.Cout);
        for (int j = 0; j < L; ++j) add_img(0, p->up[j].woff, Ku, p->up[j].Cin, p->up[j].Cout);
        p->npack_fwd = (int)p->pack.size();
        // input-gradient convs: "input channels" = the forward conv's Cout, outputs = its Cin
        for (int i = 1; i < L; ++i) {
            const ConvLayer& cl = p->down[i];
            add_img(1, cl.wt_full, cl.KW, cl.Cout, cl.Cin);
            if (!same) for (int ph = 0; ph < 2; ++ph) add_img(1, cl.wt_ph[ph], cl.Jp[ph], cl.Cout, cl.Cin);
            // fused two-phase image: the transposed copy [J0][Cout][2][Cin] read as a [J0][Cout][2*Cin] matrix
            if (!same) add_img(1, cl.wt_ph2, cl.J0, cl.Cout, 2 * cl.Cin, 32);   // (+32: a phase-1 column tile may overhang)
        }
        add_img(1, p->bott.wt_full, Kd, p->bott.Cout, p->bott.Cin);
        for (int j = 0; j < L; ++j) add_img(1, p->up[j].wt_full, Ku, p->up[j].Cout, p->up[j].Cin);
    }

    // ---- wgrad partial buffer: max over layers of (total splits) * (kernel + bias floats) ----
    long long pmax = 0;
    // floats per split in the tile-major partial buffer, worst-case tile padding (384 rows x 80 columns)
    // (bf16 speed mode: rows come in slots of 16 channels x 1 tap, up to 32 slots per row group)
    const long long rowpad = p->bf16 ? 1024 : 384;
    auto blockf = [rowpad](const ConvLayer& cl) { return ((long long)cl.KW * (cl.Cin + 15) + 1 + rowpad) * (cl.Cout + 80); };
    // one part of a layer's weight gradient under BOTH exact-fp32 kernels: the LDS-tiled one (tile-major partial
    // blocks) and -- where it serves the shape -- the register-window one, whose split policy differs (it always aims
    // at 1024 workgroups and counts units of its own length) and whose partial blocks have the final layout
    auto need = [&](const WgradArgs& w, const ConvLayer& cl) {
        long long n = (long long)wgrad_pick_nsplit(w) * blockf(cl);
        WgradArgs ww = w;
        ww.win = 1;
        if (!p->bf16 && wgrad_win_supported(ww)) n = std::max(n, (long long)wgrad_pick_nsplit(ww) * wgrad_win_partial_floats(ww));
        return n;
    };
    for (int i = 0; i < L; ++i) {
        const DownShape& d = p->dsh[i];
        long long n;
        if (same) {
            n = need(wgrad_shape_only(B, d.cin, 0, Kd, LOADER_DIRECT, d.cout, d.t_conv), p->down[i]);
        } else {
            n = need(wgrad_shape_only(B, d.cin, 0, Kd, LOADER_DEINT, d.cout, d.t_dec), p->down[i]);
            if (!p->dedup) n += need(wgrad_shape_only(B, d.cin, 0, Kd, LOADER_DIRECT, d.cout, d.tc), p->down[i]);
            else if (d.n_odd > 0) n += need(wgrad_shape_only(B, d.cin, 0, Kd, LOADER_DEINT, d.cout, d.n_odd), p->down[i]);
        }
        pmax = std::max(pmax, n);
    }
    pmax = std::max(pmax, need(wgrad_shape_only(B, p->bott.Cin, 0, Kd, LOADER_DIRECT, p->c_b, p->t_b), p->bott));
    for (int j = 0; j < L; ++j)
        pmax = std::max(pmax, need(wgrad_shape_only(B, p->ush[j].c_skip, p->ush[j].c_cur, Ku, LOADER_DIRECT, p->ush[j].cout, p->ush[j].t_conv), p->up[j]));
    if (p->Sh > 0)
        pmax = std::max(pmax, need(wgrad_shape_only(B, C, F, Ko, LOADER_DIRECT, C, p->Tout), p->head[0]));
    pmax *= 4;                                              // two streams' halves, each with 2x headroom for the autotuner's split counts
    p->partial_floats = pmax;
    p->partial_off = bump(w, pmax);
    p->ws = (w + 63) / 64 * 64;

    // ---- FLOP accounting (2*MAC of the conv contractions) ----
    auto cf = [&](int K, double ci, double co, double tt) { return 2.0 * K * ci * co * tt * B; };
    double fwd = 0, dense = 0, bwd = 0, uniq_f = 0, uniq_b = 0;
    for (int i = 0; i < L; ++i) {
        const DownShape& d = p->dsh[i];
        dense += cf(Kd, d.cin, d.cout, d.t_conv);
        // every observed conv output once: the even positions + the odd positions inside the crop window
        const double once = same ? cf(Kd, d.cin, d.cout, d.t_conv) : cf(Kd, d.cin, d.cout, d.t_dec) + cf(Kd, d.cin, d.cout, d.n_odd);
        const double live = (same || p->dedup) ? once : cf(Kd, d.cin, d.cout, d.t_dec) + cf(Kd, d.cin, d.cout, d.tc);
        fwd += live; uniq_f += once;
        bwd += live * (i > 0 ? 2.0 : 1.0); uniq_b += once * (i > 0 ? 2.0 : 1.0);
    }
    { const double f = cf(Kd, p->bott.Cin, p->c_b, p->t_b); fwd += f; dense += f; bwd += 2 * f; uniq_f += f; uniq_b += 2 * f; }
    for (int j = 0; j < L; ++j) {
        const double f = cf(Ku, p->ush[j].c_skip + p->ush[j].c_cur, p->ush[j].cout, p->ush[j].t_conv);
        fwd += f; dense += f; bwd += 2 * f; uniq_f += f; uniq_b += 2 * f;
    }
    { const double f = p->Sh * cf(Ko, C + F, C, p->Tout); fwd += f; dense += f; bwd += 2 * f; uniq_f += f; uniq_b += 2 * f; }
    p->fwd_flops = fwd; p->bwd_flops = bwd; p->fwd_dense = dense; p->fwd_unique = uniq_f; p->bwd_unique = uniq_b;

    // ---- device-side descriptor table (the only device memory the plan owns) ----
    if (!p->wt.empty()) {
        hipError_t e = hipMalloc((void**)&p->dev_wt, p->wt.size() * sizeof(WtDesc));
        if (e == hipSuccess) e = hipMemcpy(p->dev_wt, p->wt.data(), p->wt.size() * sizeof(WtDesc), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            // no device available (e.g. CPU-only build box): the plan is still usable for queries
            p->dev_wt = nullptr;
            (void)hipGetLastError();
        }
    }
    if (!p->pack.empty()) {
        hipError_t e = hipMalloc((void**)&p->dev_pack, p->pack.size() * sizeof(PackDesc));
        if (e == hipSuccess) e = hipMemcpy(p->dev_pack, p->pack.data(), p->pack.size() * sizeof(PackDesc), hipMemcpyHostToDevice);
        if (e != hipSuccess) { p->dev_pack = nullptr; (void)hipGetLastError(); }
    }
    *out = p;
    return WUN_OK;
}

extern "C" void wun_plan_destroy(wun_plan* p) {
    if (!p) return;
    if (p->dev_wt) (void)hipFree(p->dev_wt);
    if (p->dev_pack) (void)hipFree(p->dev_pack);
    for (auto e : p->events) (void)hipEventDestroy(e);
    if (p->tev0) { (void)hipEventDestroy(p->tev0); (void)hipEventDestroy(p->tev1); }
    if (p->wt_ev) (void)hipEventDestroy(p->wt_ev);
    for (auto e : p->skip_ev) (void)hipEventDestroy(e);
    for (auto e : p->win_ev) (void)hipEventDestroy(e);
    if (p->side) (void)hipStreamDestroy(p->side);
    if (p->side2) (void)hipStreamDestroy(p->side2);
    delete p;
}

extern "C" int wun_plan_query(const wun_plan* p, wun_plan_info* info) {
    if (!p || !info) return fail(WUN_ERR_INVALID, "null argument");
    info->batch = p->B; info->input_frames = p->Tin; info->output_frames = p->Tout;
    long long n = 0;
    for (const auto& t : p->tensors) { long long k = 1; for (int i = 0; i < t.ndim; ++i) k *= t.shape[i]; n += k; }
    info->num_params = n; info->arena_floats = p->arena; info->workspace_floats = p->ws;
    info->num_tensors = (int64_t)p->tensors.size(); info->num_outputs = p->S;
    info->fwd_flops = p->fwd_flops; info->bwd_flops = p->bwd_flops; info->fwd_flops_dense = p->fwd_dense;
    info->fwd_flops_unique = p->fwd_unique; info->bwd_flops_unique = p->bwd_unique;
    info->compute_dtype_effective = p->bf16 ? 1 : 0;
    return WUN_OK;
}

extern "C" int wun_plan_tensor(const wun_plan* p, int64_t index, wun_tensor_info* info) {
    if (!p || !info) return fail(WUN_ERR_INVALID, "null argument");
    if (index < 0 || index >= (int64_t)p->tensors.size()) return fail(WUN_ERR_INVALID, "tensor index out of range");
    *info = p->tensors[(size_t)index];
    return WUN_OK;
}

extern "C" int wun_plan_activation(const wun_plan* p, int32_t kind, int32_t index, wun_activation_info* info) {
    if (!p || !info) return fail(WUN_ERR_INVALID, "null argument");
    const Buf* b = nullptr;
    long long t0 = 0, tstep = 1;
    if (kind == 0 && index >= 0 && index < p->L) { b = &p->dec[(size_t)index]; tstep = 2; }
    else if (kind == 1 && index >= 0 && index < p->L) { b = &p->skip[(size_t)index]; t0 = p->same ? 0 : p->dsh[(size_t)index].cs; }
    else if (kind == 2 && index == 0) b = &p->bott_out;
    else if (kind == 3 && index >= 0 && index < p->L) b = &p->upo[(size_t)index];
    else if (kind == 4 && index >= 0 && index < p->L) b = &p->ups[(size_t)index];
    else if (kind == 5 && index >= 0 && index < p->L) b = &p->dz_upo[(size_t)index];
    else if (kind == 6 && index >= 0 && index < p->L) b = &p->d_ups[(size_t)index];
    else if (kind == 7 && index >= 0 && index < p->L) { b = &p->dz_skip[(size_t)index]; t0 = p->same ? 0 : p->dsh[(size_t)index].cs; }
    else if (kind == 8 && index >= 0 && index < p->L && !p->same) { b = &p->dz_dec[(size_t)index]; tstep = 2; }
    else if (kind == 9 && index == 0) b = &p->dz_bott;
    if (b == nullptr) return fail(WUN_ERR_INVALID, "wun_plan_activation: unknown kind / index");
    info->offset = b->off; info->batch_stride = b->bs; info->pitch = b->pitch;
    info->channels = b->C; info->frames = b->T; info->t0 = t0; info->tstep = tstep; info->elem_bytes = b->eb;
    return WUN_OK;
}

// ---------------------------------------------------------------------------------------
// helpers to fill argument blocks
// ---------------------------------------------------------------------------------------
static ConvArgs conv_base(const wun_plan* p) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.B = p->B; a.ostride = 1; a.loader = LOADER_DIRECT;
    return a;
}
static void set_src0(ConvArgs& a, const float* ws, const Buf& b, int off, int C) {
    a.src0 = ws + b.off; a.bs0 = b.bs; a.pitch0 = b.pitch; a.off0 = off; a.C0 = C;
}
static void set_src1(ConvArgs& a, const float* ws, const Buf& b, int off, int C) {
    a.src1 = ws + b.off; a.bs1 = b.bs; a.pitch1 = b.pitch; a.off1 = off; a.C1 = C;
}
static void set_dst0(ConvArgs& a, float* ws, const Buf& b, int off, const Buf* mask) {
    a.dst0 = ws + b.off; a.obs0 = b.bs; a.opitch0 = b.pitch; a.ooff0 = off;
    a.msk0 = mask ? ws + mask->off : nullptr;
}
static void set_dst1(ConvArgs& a, float* ws, const Buf& b, int off, const Buf* mask) {
    a.dst1 = ws + b.off; a.obs1 = b.bs; a.opitch1 = b.pitch; a.ooff1 = off;
    a.msk1 = mask ? ws + mask->off : nullptr;
}
static WgradArgs wgrad_base(const wun_plan* p) {
    WgradArgs w;
    memset(&w, 0, sizeof(w));
    w.B = p->B; w.loader = LOADER_DIRECT;
    return w;
}
static void wset_src0(WgradArgs& a, const float* ws, const Buf& b, int off, int C) {
    a.src0 = ws + b.off; a.bs0 = b.bs; a.pitch0 = b.pitch; a.off0 = off; a.C0 = C;
}
static void wset_src1(WgradArgs& a, const float* ws, const Buf& b, int off, int C) {
    a.src1 = ws + b.off; a.bs1 = b.bs; a.pitch1 = b.pitch; a.off1 = off; a.C1 = C;
}
static void wset_dz(WgradArgs& a, const float* base, long long bs, int pitch, int N, int Tq) {
    a.dz = base; a.dzbs = bs; a.dzpitch = pitch; a.N = N; a.Tq = Tq;
}

static HeadArgs head_args(const wun_plan* p, const float* params, float* ws, float* outputs, int training) {
    HeadArgs h;
    memset(&h, 0, sizeof(h));
    const int L = p->L;
    h.mix_ncw = ws + p->mix_ncw.off; h.mbs = p->mix_ncw.bs; h.mpitch = p->mix_ncw.pitch;
    h.moff_feat = p->in_crop_start; h.moff_diff = p->mix_diff_off;
    h.feat = ws + p->upo[L - 1].off; h.fbs = p->upo[L - 1].bs; h.fpitch = p->upo[L - 1].pitch;
    h.Wh = params;
    h.C = p->C; h.F = p->cfg.num_initial_filters; h.S = p->S; h.Sh = p->Sh; h.Ko = p->cfg.output_filter_size;
    h.padl = p->same ? (h.Ko - 1) / 2 : 0;
    h.Tfeat = p->t_feat; h.Tout = p->Tout; h.B = p->B;
    h.tanh_act = p->cfg.output_activation == 0; h.difference = p->cfg.output_type == 1; h.training = training;
    h.out = outputs;
    h.dpre = ws + p->dpre_off; h.dppitch = p->dp_pitch; h.dpbs = (long long)p->C * p->dp_pitch;
    h.dps = (long long)p->B * h.dpbs;
    h.dzfeat = ws + p->dz_upo[L - 1].off;
    h.loss_partial = ws + p->loss_partial_off;
    h.gscale = 2.0f / ((float)p->S * (float)p->B * (float)p->Tout * (float)p->C);
    h.featbf = p->bf16 ? 1 : 0;
    return h;
}


// ---------------------------------------------------------------------------------------
// two-stream helpers
// ---------------------------------------------------------------------------------------
// Flags of the plan's cross-stream events.  They only order kernels of ONE device against each other: the kernel
// packets' own end-of-kernel release / start-of-kernel acquire (agent scope, needed between any two dependent kernels
// on a part whose 8 L2s are not coherent) already make the data visible, so the event itself carries no system-scope
// fence (hipEventDisableSystemFence; host-side consumers synchronise through the caller's stream, never through
// these events).  A/B with pinned tilings: 9.25 -> 9.14 ms per step; the whole GPU suite (bit-exact determinism,
// B=16 vs oracle) passes in both modes.  WUN_EVENT_SCOPE=system|device: fall-back switch.
static unsigned event_flags() {
    static const unsigned f = [] {
        const char* e = getenv("WUN_EVENT_SCOPE");
        if (e && e[0] == 's') return (unsigned)hipEventDisableTiming;
        if (e && e[0] == 'd') return (unsigned)(hipEventDisableTiming | hipEventReleaseToDevice);
        return (unsigned)(hipEventDisableTiming | hipEventDisableSystemFence);
    }();
    return f;
}

static int side_init(const wun_plan* p) {
    if (p->side != nullptr) return WUN_OK;
    if (getenv("WUN_SINGLE_STREAM") != nullptr) return WUN_OK;      // debugging: everything on one stream
    // The side streams carry the off-critical-path work (weight gradients, deferred skip-window convs): lowest queue
    // priority, so their workgroups fill the drain of the dependent chain on the caller's stream instead of sharing the
    // CUs with it (A/B, pinned tilings: 9.32 -> 9.21 ms per step; "high" 9.39).  WUN_SIDE_PRIO=normal|high: experiment switch.
    // Only with wun_config.exclusive_streams: beside a communication stream (RCCL all-reduce on one GPU, same box) the
    // low-priority queues made the step 13.1 ms instead of 9.2 -- and a process that has ever created them stays slow.
    int least = 0, greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    int prio = p->cfg.exclusive_streams ? least : 0;
    if (const char* e = getenv("WUN_SIDE_PRIO")) prio = (e[0] == 'l') ? least : (e[0] == 'h') ? greatest : 0;
    HIP_TRY(hipStreamCreateWithPriority(&p->side, hipStreamNonBlocking, prio));
    HIP_TRY(hipStreamCreateWithPriority(&p->side2, hipStreamNonBlocking, prio));
    p->events.resize(160);
    for (auto& e : p->events) HIP_TRY(hipEventCreateWithFlags(&e, event_flags()));
    return WUN_OK;
}
// `to` waits for everything issued so far on `from`
static int stream_dep(const wun_plan* p, hipStream_t from, hipStream_t to) {
    if (from == to) return WUN_OK;
    hipEvent_t e = p->events[p->ev_next++ % p->events.size()];
    HIP_TRY(hipEventRecord(e, from));
    HIP_TRY(hipStreamWaitEvent(to, e, 0));
    return WUN_OK;
}


// ---------------------------------------------------------------------------------------
// autotuned dispatch: every conv / wgrad launch of a step has a fixed position in the launch
// order; wun_plan_tune measures candidate (tile variant, split-K) / (geometry, split count)
// choices for each position on the real buffers and caches the fastest.
// ---------------------------------------------------------------------------------------
static float time_launch(const wun_plan* p, hipStream_t s, const std::function<hipError_t()>& fn) {
    if (fn() != hipSuccess) { (void)hipGetLastError(); return 1e30f; }      // warm-up / validity
    float best = 1e30f;
    for (int r = 0; r < 2; ++r) {
        (void)hipEventRecord(p->tev0, s);
        if (fn() != hipSuccess) { (void)hipGetLastError(); return 1e30f; }
        (void)hipEventRecord(p->tev1, s);
        if (hipEventSynchronize(p->tev1) != hipSuccess) return 1e30f;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, p->tev0, p->tev1);
        if (ms < best) best = ms;
    }
    return best;
}

// One event-bracketed run (no warm-up); 1e30 on failure.
static float time_once(const wun_plan* p, hipStream_t s, const std::function<hipError_t()>& fn) {
    (void)hipEventRecord(p->tev0, s);
    if (fn() != hipSuccess) { (void)hipGetLastError(); return 1e30f; }
    (void)hipEventRecord(p->tev1, s);
    if (hipEventSynchronize(p->tev1) != hipSuccess) return 1e30f;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, p->tev0, p->tev1);
    return ms;
}

// WUN_TUNE_ALTS log: how many near-best candidates per launch position, and how near (tools/step_tune.py)
static size_t alts_max() { const char* e = getenv("WUN_TUNE_ALTS_MAX"); const int v = e ? atoi(e) : 6; return (size_t)(v > 0 ? v : 6); }
static float alts_tol() { const char* e = getenv("WUN_TUNE_ALTS_TOL"); const float v = e ? (float)atof(e) : 1.15f; return v > 1.f ? v : 1.15f; }

// Times `n` candidate launches of ONE launch position against each other: every candidate is warmed up once, then
// the candidates are run round-robin for WUN_TUNE_ROUNDS rounds (default 4) and each keeps its fastest run.  The shader
// clock of a busy MI355X drifts by ~10 % over milliseconds (DVFS); timing candidates one after the other in a single
// pass -- the round-1/2 tuner -- lets that drift decide between tiles that differ by a few per cent.  best[i] = 1e30 for
// candidates that failed.
static void time_candidates(const wun_plan* p, hipStream_t s, int n, const std::function<hipError_t(int)>& launch, float* best) {
    static const int rounds = getenv("WUN_TUNE_ROUNDS") ? std::max(1, atoi(getenv("WUN_TUNE_ROUNDS"))) : 4;
    for (int i = 0; i < n; ++i) {
        best[i] = 1e30f;
        if (launch(i) != hipSuccess) { (void)hipGetLastError(); best[i] = -1.f; }    // warm-up / validity
    }
    for (int r = 0; r < rounds; ++r)
        for (int i = 0; i < n; ++i) {
            if (best[i] < 0.f) continue;
            const float ms = time_once(p, s, [&]() { return launch(i); });
            if (ms < best[i]) best[i] = ms;
        }
    for (int i = 0; i < n; ++i)
        if (best[i] < 0.f) best[i] = 1e30f;
}

// at < 0: the launch takes the next position of the step's launch order; at >= 0: a position reserved earlier
// (deferred launches keep the position they have in the canonical order, so tuned tables stay aligned)
static hipError_t conv_dispatch(const wun_plan* p, ConvArgs a, float* part, long long cap, hipStream_t s, long long at = -1) {
    std::vector<ConvChoice>& vec = p->in_bwd ? p->conv_bwd : p->conv_fwd;
    const size_t idx = at >= 0 ? (size_t)at : p->ci++;
    if (p->bf16) {
        // bf16 mode: every tensor this launch touches holds bf16 elements; the bf16 MFMA kernel is the ONLY kernel that can
        // serve it (bf16_plan_ok admitted the plan on that condition) -- weights from the packed image
        if (a.C0 + a.C1 < 8) {
            // the audio-input conv: fp32 audio in, bf16 activations out, direct conv on the vector pipe (wun_bf16.hip)
            a.xbf = 0; a.obf = 1;
            return launch_first_conv(a, s);
        }
        a.xbf = 1; a.obf = 1;
        if (!conv_bf16_supported(a)) return hipErrorInvalidValue;
        const bool in_ws = a.W >= p->cur_ws && a.W < p->cur_ws + p->ws;
        auto it = p->bf_img.find({in_ws ? 1 : 0, (long long)(a.W - (in_ws ? p->cur_ws : p->cur_params))});
        if (it == p->bf_img.end()) return hipErrorInvalidValue;
        {
            a.W = p->cur_ws + it->second.off;
            a.wb_c8p = it->second.c8p; a.wb_npad = it->second.npad;
            // tile (positions x columns x channel chunks per stage) autotuned like the fp32 variants
            if (p->tune_mode == 1) {
                if (vec.size() <= idx) vec.resize(idx + 1, ConvChoice{-1, 0});
                ConvChoice cands[32];
                const int n = conv_bf16_list_candidates(a, cands, 32);
                float best = time_launch(p, s, [&]() { return launch_conv_bf16(a, s); });
                const float base = best;
                ConvChoice bc{-1, 0};
                for (int i = 0; i < n; ++i) {
                    ConvArgs b = a;
                    b.force_variant = cands[i].variant + 1;
                    const float ms = time_launch(p, s, [&]() { return launch_conv_bf16(b, s); });
                    if (ms < best * 0.98f) { best = ms; bc = cands[i]; }
                }
                vec[idx] = bc;
                if (getenv("WUN_TUNE_LOG"))
                    fprintf(stderr, "[tune conv-bf16 %s#%zu] C=%d N=%d T=%d K=%d ld=%d ph2=%d cands=%d base %.3f ms -> code=%d %.3f ms\n",
                            p->in_bwd ? "bwd" : "fwd", idx, a.C0 + a.C1, a.N, a.Tout, a.KW, a.loader, (a.flags & F_PHASE2) ? 1 : 0, n,
                            base, bc.variant, best);
            }
            if (p->tune_mode >= 1 && idx < vec.size() && vec[idx].variant >= kBf16VariantBase && conv_bf16_choice_ok(a, vec[idx].variant))
                a.force_variant = vec[idx].variant + 1;
            return launch_conv_bf16(a, s);
        }
    }
    if (p->tune_mode == 1) {
        if (vec.size() <= idx) vec.resize(idx + 1, ConvChoice{-1, 0});
        std::vector<ConvChoice> cands_v(640);               // (per call: two plans may be tuned from different threads)
        std::vector<float> tms_v(641);
        ConvChoice* cands = cands_v.data();
        float* tms = tms_v.data();
        const int n = conv_list_candidates(a, part ? cap : 0, cands, 640);
        // candidate n = the heuristic choice (the baseline); a candidate has to beat it by > 2 %
        time_candidates(p, s, n + 1, [&](int i) {
            ConvArgs b = a;
            b.ups_y = nullptr; b.ubw_dz = nullptr;   // (candidates are compared without the fused extras only split-K ones write)
            if (i < n) { b.force_variant = cands[i].variant + 1; b.force_ksplit = cands[i].ksplit; }
            return launch_conv(b, part, cap, s);
        }, tms);
        const float base = tms[n];
        float best = base;
        ConvChoice bc{-1, 0};
        int bi = -1;
        for (int i = 0; i < n; ++i)
            if (bi < 0 ? tms[i] < 1e29f : tms[i] < tms[bi]) bi = i;
        if (bi >= 0 && tms[bi] < base * 0.98f) { best = tms[bi]; bc = cands[bi]; }
        vec[idx] = bc;
        if (const char* af = getenv("WUN_TUNE_ALTS")) {
            // near-best candidates of this position (isolated timing) for the whole-step tuner, tools/step_tune.py
            if (FILE* f = fopen(af, "a")) {
                const float lim = std::min(best, base) * alts_tol();
                fprintf(f, "%s %zu %d %d %.4f\n", p->in_bwd ? "cb" : "cf", idx, -1, 0, base);
                std::vector<int> order;
                for (int i = 0; i < n; ++i) if (tms[i] <= lim) order.push_back(i);
                std::sort(order.begin(), order.end(), [&](int x, int y) { return tms[x] < tms[y]; });
                for (size_t k = 0; k < order.size() && k < alts_max(); ++k)
                    fprintf(f, "%s %zu %d %d %.4f\n", p->in_bwd ? "cb" : "cf", idx, cands[order[k]].variant, cands[order[k]].ksplit, tms[order[k]]);
                fclose(f);
            }
        }
        if (getenv("WUN_TUNE_LOG"))
            fprintf(stderr, "[tune conv %s#%zu] C=%d N=%d T=%d K=%d ld=%d ph2=%d cands=%d base %.3f ms -> v=%d ks=%d %.3f ms\n",
                    p->in_bwd ? "bwd" : "fwd", idx, a.C0 + a.C1, a.N, a.Tout, a.KW, a.loader, (a.flags & F_PHASE2) ? 1 : 0, n,
                    base, bc.variant, bc.ksplit, best);
    }
    if (p->tune_mode >= 1 && idx < vec.size() && vec[idx].variant >= 0 &&
        conv_choice_ok(a, part ? cap : 0, vec[idx].variant, vec[idx].ksplit > 0 ? vec[idx].ksplit : 1)) {
        // (an entry that is not a legal choice for this launch -- a stale or edited table -- is ignored)
        a.force_variant = vec[idx].variant + 1; a.force_ksplit = vec[idx].ksplit;
    }
    return launch_conv(a, part, cap, s);
}

// ---------------------------------------------------------------------------------------
// forward: get_output (UnetAudioSeparator.py:85-144)
// ---------------------------------------------------------------------------------------
extern "C" int wun_forward(const wun_plan* p, const float* params, const float* mix_btc, float* ws,
                           float* outputs, int training, void* stream) {
    if (!p || !params || !mix_btc || !ws || !outputs) return fail(WUN_ERR_INVALID, "null argument");
    hipStream_t s = (hipStream_t)stream;
    const int L = p->L, Kd = p->cfg.filter_size, Ku = p->cfg.merge_filter_size;
    const bool same = p->same;
    const int padD = same ? (Kd - 1) / 2 : 0, padU = same ? (Ku - 1) / 2 : 0;
    int rc0;
    if ((rc0 = side_init(p))) return rc0;
    p->ci = 0; p->in_bwd = false;
    hipStream_t s2 = (p->side && !g_profiling && p->tune_mode != 1) ? p->side : s;   // side stream (skip-window convs)
    bool side_used = false;

    p->cur_params = params; p->cur_ws = ws;
    if (p->bf16) {
        if (!p->dev_pack) return fail(WUN_ERR_HIP, "plan was created without a usable HIP device");
        HIP_TRY(launch_pack_bf16(params, ws, p->dev_pack, p->npack_fwd, p->pack_max, s));
    }
    p->wt_ready = false;
    if (training && !p->wt.empty() && p->dev_wt && s2 != s) {
        // the backward pass will need tap-flipped / transposed copies of every kernel: make them now,
        // beside the forward convs (they depend on the parameters only)
        if (!p->wt_ev) HIP_TRY(hipEventCreateWithFlags(&p->wt_ev, event_flags()));
        if ((rc0 = stream_dep(p, s, s2))) return rc0;
        HIP_TRY(launch_make_wt(params, ws, p->dev_wt, (int)p->wt.size(), p->wt_max, s2));
        if (p->bf16)
            HIP_TRY(launch_pack_bf16(params, ws, p->dev_pack + p->npack_fwd, (int)p->pack.size() - p->npack_fwd, p->pack_max, s2));
        HIP_TRY(hipEventRecord(p->wt_ev, s2));
        p->wt_ready = true;
        side_used = true;
    }
    HIP_TRY(launch_btc_to_ncw(mix_btc, ws + p->mix_ncw.off, p->B, p->Tin, p->C, p->mix_ncw.pitch, s));
    if (p->head16 && training)
        HIP_TRY(launch_cast_rows_bf16(ws + p->mix_ncw.off, ws + p->mix16.off, (long long)p->B * p->C, p->Tin, p->mix_ncw.pitch,
                                      p->mix16.pitch, s));

    // Context mode: the skip-window conv of level i is only consumed by up level L-1-i, i.e. the windows of the
    // shallow, FLOP-heavy levels are needed LAST.  The deep levels (few positions per excerpt) form a dependent
    // chain of launch-latency-bound kernels that leaves most CUs idle, so the window convs are deferred: queued on a
    // third stream (deepest-needed first) and awaited per level by the up path.  They fill the idle CUs instead of
    // competing with their own level's decimating conv.
    int defer_below = 0;                                            // levels [0, defer_below) are deferred
    hipStream_t s3 = (p->side2 && s2 != s) ? p->side2 : s2;
    if (!same && s3 != s2) {
        while (defer_below < L && (long long)p->B * p->dsh[defer_below].t_dec >= 16384) ++defer_below;
        if (L - defer_below < 3) defer_below = 0;                   // no deep chain to hide them under
        // ... and then the deep levels' (small) window convs are deferred as well: ONE event on the caller's stream
        // starts all of them instead of one event per level (each event holds the dependent chain for ~6 us); same-box
        // A/B 9.085 -> 9.04 ms.  (Awaiting the deep ones in groups instead of per level stalls the up path: 9.10-9.16.)
        if (defer_below > 0) defer_below = L;
        if (defer_below > 0 && p->skip_ev.size() < (size_t)L) {
            p->skip_ev.resize(L, nullptr);
            for (auto& e : p->skip_ev)
                if (!e) HIP_TRY(hipEventCreateWithFlags(&e, event_flags()));
        }
    }
    std::vector<ConvArgs> deferred((size_t)defer_below);
    std::vector<long long> deferred_pos((size_t)defer_below, -1);
    const long long part_half = p->conv_part_floats / 2, part_q = p->conv_part_floats / 4;

    // The 2x upsampling that opens up level j reads only the producer's output (bottleneck conv for j = 0, up conv
    // j - 1 otherwise).  A producer launch that ends in the split-K epilogue kernel -- 10 of the 12 on the headline
    // configuration -- writes the upsampled copy from there (ConvArgs.ups_*): one launch less on the dependent chain per
    // level; the others still launch upsample_vec_kernel.  WUN_NO_FUSE_UPS=1: always the separate kernel.
    const bool fuse_ups = !p->bf16 && getenv("WUN_NO_FUSE_UPS") == nullptr;
    bool ups_done = false;
    auto want_ups = [&](ConvArgs& a, int j) {
        a.ups_y = ws + p->ups[j].off; a.ups_bs = p->ups[j].bs; a.ups_pitch = p->ups[j].pitch; a.ups_tup = p->ush[j].t_up;
        a.ups_w = p->interp[j] >= 0 ? params + p->interp[j] : nullptr;
    };

    const Buf* x = &p->mix_ncw;
    for (int i = 0; i < L; ++i) {                                   // :97-100
        const DownShape& d = p->dsh[i];
        const ConvLayer& cl = p->down[i];
        if (same) {
            ConvArgs a = conv_base(p);
            set_src0(a, ws, *x, 0, d.cin);
            a.Tin = d.t_in; a.shift = padD; a.W = params + cl.woff; a.bias = params + cl.boff;
            a.KW = Kd; a.N = a.N0 = d.cout; a.Tout = d.t_conv; a.flags = F_LRELU;
            set_dst0(a, ws, p->skip[i], 0, nullptr);
            a.dec = ws + p->dec[i].off; a.decbs = p->dec[i].bs; a.decpitch = p->dec[i].pitch;
            HIP_TRY(conv_dispatch(p, a, ws + p->conv_part_off, p->conv_part_floats / 2, s));
        } else {
            // x (written on `s`) is ready for both convs of this level: the side stream may start.  (Levels whose
            // window conv is deferred queue nothing on s2: no event -- every record / wait on the caller's stream is a
            // barrier packet that holds the dependent chain for ~7 us.)
            if (i >= defer_below && (rc0 = stream_dep(p, s, s2))) return rc0;
            // stride-2 conv straight into the decimated stream (odd outputs are never observed)
            ConvArgs a = conv_base(p);
            set_src0(a, ws, *x, 0, d.cin);
            a.loader = LOADER_DEINT;
            a.Tin = d.t_in; a.shift = 0; a.W = params + cl.woff; a.bias = params + cl.boff;
            a.KW = Kd; a.N = a.N0 = d.cout; a.Tout = d.t_dec; a.flags = F_LRELU;
            set_dst0(a, ws, p->dec[i], 0, nullptr);
            if (p->dedup) {
                // ... and, where 2q lies inside the crop window, into the skip window as well: the decimated stream IS a
                // slice of the encoder output (:98-100), one value, one rounding
                a.dec = ws + p->skip[i].off; a.decbs = p->skip[i].bs; a.decpitch = p->skip[i].pitch;
                a.dec_exp = 1; a.dec_lo = d.cs; a.dec_len = (unsigned)d.tc;
            }
            HIP_TRY(conv_dispatch(p, a, ws + p->conv_part_off, p->conv_part_floats / 2, s));
            // the rest of the window the skip connection crops (Utils.py:104-123) -- dedup plans: its ODD positions, a second
            // stride-2 conv over x shifted by one sample, stored with stride 2; else a full-rate conv over the whole window;
            // independent of the decimating conv -> side stream, own half of the split-K scratch
            ConvArgs b = conv_base(p);
            bool have_b = true;
            if (p->dedup) {
                have_b = d.n_odd > 0;
                set_src0(b, ws, *x, d.t_odd0, d.cin);
                b.loader = LOADER_DEINT;
                b.Tin = d.t_in - d.t_odd0; b.shift = 0; b.W = params + cl.woff; b.bias = params + cl.boff;
                b.KW = Kd; b.N = b.N0 = d.cout; b.Tout = d.n_odd; b.flags = F_LRELU;
                set_dst0(b, ws, p->skip[i], d.t_odd0 - d.cs, nullptr);
                b.ostride = 2;
            } else {
                set_src0(b, ws, *x, d.cs, d.cin);
                b.Tin = d.tc + Kd - 1; b.shift = 0; b.W = params + cl.woff; b.bias = params + cl.boff;
                b.KW = Kd; b.N = b.N0 = d.cout; b.Tout = d.tc; b.flags = F_LRELU;
                set_dst0(b, ws, p->skip[i], 0, nullptr);
            }
            if (i < defer_below) {
                deferred[(size_t)i] = b;
                deferred_pos[(size_t)i] = have_b ? (long long)p->ci++ : -2;   // its position in the canonical launch order
            } else if (have_b) {
                HIP_TRY(conv_dispatch(p, b, ws + p->conv_part_off + part_half, part_q, s2));
                side_used = side_used || (s2 != s);
            }
            if (defer_below > 0 && i == defer_below - 1) {
                // every input the deferred windows read has been issued on `s`: start them on the third stream
                if ((rc0 = stream_dep(p, s, s3))) return rc0;
                for (int k = defer_below - 1; k >= 0; --k) {
                    if (deferred_pos[(size_t)k] != -2)
                        HIP_TRY(conv_dispatch(p, deferred[(size_t)k], ws + p->conv_part_off + part_half + part_q, part_q, s3,
                                              deferred_pos[(size_t)k]));
                    HIP_TRY(hipEventRecord(p->skip_ev[(size_t)k], s3));
                }
            }
        }
        x = &p->dec[i];
    }
    {                                                               // :102
        ConvArgs a = conv_base(p);
        set_src0(a, ws, *x, 0, p->bott.Cin);
        a.Tin = p->t_b_in; a.shift = padD; a.W = params + p->bott.woff; a.bias = params + p->bott.boff;
        a.KW = Kd; a.N = a.N0 = p->c_b; a.Tout = p->t_b; a.flags = F_LRELU;
        set_dst0(a, ws, p->bott_out, 0, nullptr);
        if (fuse_ups) want_ups(a, 0);
        HIP_TRY(conv_dispatch(p, a, ws + p->conv_part_off, p->conv_part_floats / 2, s));
        ups_done = fuse_ups && conv_last_fused_ups() != 0;
    }
    if (side_used && (rc0 = stream_dep(p, s2, s))) return rc0;     // the up path reads the skip windows
    const Buf* cur = &p->bott_out;
    for (int j = 0; j < L; ++j) {                                   // :107-125
        const UpShape& u = p->ush[j];
        if (!ups_done) {
            // (the producer's launch did not end in the split-K epilogue kernel, which writes this copy itself)
            UpsampleArgs ua;
            memset(&ua, 0, sizeof(ua));
            ua.x = ws + cur->off; ua.xbs = cur->bs; ua.xpitch = cur->pitch; ua.n = u.t_cur;
            ua.y = ws + p->ups[j].off; ua.ybs = p->ups[j].bs; ua.ypitch = p->ups[j].pitch; ua.tup = u.t_up;
            ua.w = p->interp[j] >= 0 ? params + p->interp[j] : nullptr;
            ua.C = u.c_cur; ua.B = p->B; ua.context = p->cfg.context; ua.bf = p->bf16 ? 1 : 0;
            HIP_TRY(launch_upsample(ua, s));
        }
        if (L - 1 - j < defer_below) HIP_TRY(hipStreamWaitEvent(s, p->skip_ev[(size_t)(L - 1 - j)], 0));
        ConvArgs a = conv_base(p);
        set_src0(a, ws, p->skip[L - 1 - j], 0, u.c_skip);          // crop already applied when it was written
        set_src1(a, ws, p->ups[j], 0, u.c_cur);
        a.Tin = u.t_up; a.shift = padU; a.W = params + p->up[j].woff; a.bias = params + p->up[j].boff;
        a.KW = Ku; a.N = a.N0 = u.cout; a.Tout = u.t_conv; a.flags = F_LRELU;
        set_dst0(a, ws, p->upo[j], 0, nullptr);
        if (fuse_ups && j + 1 < L) want_ups(a, j + 1);
        HIP_TRY(conv_dispatch(p, a, ws + p->conv_part_off, p->conv_part_floats / 2, s));
        ups_done = fuse_ups && j + 1 < L && conv_last_fused_ups() != 0;
        cur = &p->upo[j];
    }
    HeadArgs h = head_args(p, params, ws, outputs, training);
    long long hoff[4] = {0, 0, 0, 0};
    for (int i = 0; i < p->Sh; ++i) hoff[i] = p->head[i].woff;
    HIP_TRY(launch_head_fwd_off(h, hoff, s));
    return WUN_OK;
}

// ---------------------------------------------------------------------------------------
// loss + backward
// ---------------------------------------------------------------------------------------
// All parts of one layer's weight gradient (a down level has two: the decimated and the window
// positions) use ONE tile geometry, so their splits land in one tile-major partial buffer that a
// single reduction sums.  Returns false if the parts do not resolve to the same geometry.
static bool wgrad_common_geom(WgradArgs* parts, int nparts, int mtw, int nw) {
    int m0 = 0, n0 = 0;
    for (int i = 0; i < nparts; ++i) {
        parts[i].force_mtw = mtw; parts[i].force_nw = nw;
        int m, n;
        wgrad_resolved_geom(parts[i], m, n);
        if (i == 0) { m0 = m; n0 = n; }
        else if (m != m0 || n != n0) return false;
    }
    return true;
}

static int run_wgrad(const wun_plan* p, WgradArgs* parts, int nparts, const ConvLayer& cl, float* ws,
                     float* grads, hipStream_t main, hipStream_t s, bool dep = true) {
    // everything this weight gradient reads (dz, activations) has been issued on `main`
    // (dep == false: the caller already made `s` wait -- one event for a batch of weight gradients)
    if (dep) {
        int rcd = stream_dep(p, main, s);
        if (rcd) return rcd;
    }
    // bf16 speed mode: operands rounded to bf16 in LDS (same tiles, same partial layout); launches with few
    // positions are latency-bound and stay exact fp32
    if (p->bf16) {
        // bf16 mode: inputs and gradients are bf16 tensors, the bf16 kernel is the only reader
        if (!wgrad_bf16_supported(parts[0])) return fail(WUN_ERR_UNSUPPORTED, "bf16 mode: weight-gradient shape not served by the bf16 kernel");
        for (int i = 0; i < nparts; ++i) { parts[i].bf16 = 1; parts[i].sbf = 1; }
    }
    // weight gradients alternate between two side streams; each has its own half of the partial buffer
    const long long pcap = p->partial_floats / 2;
    float* partial = ws + p->partial_off + ((p->side2 && s == p->side2) ? pcap : 0);
    float* out_w = grads + cl.woff;
    float* out_b = out_w + (long long)cl.KW * cl.Cin * cl.Cout;
    const size_t idx = p->wi++;
    // exact fp32: the register-window kernel (wun_wgrad_win.hip) where every part qualifies (15 / 5 taps, channel counts
    // in whole row tiles); its split partials are in the final layout, so the parts need not share a tile geometry
    static const bool no_win = getenv("WUN_NO_WIN") != nullptr;
    bool win_ok = !no_win && !parts[0].bf16;
    for (int i = 0; i < nparts && win_ok; ++i) { WgradArgs t = parts[i]; t.win = 1; win_ok = wgrad_win_supported(t); }
    auto set_win = [&](WgradArgs* q, int cgw, int nw) {
        for (int i = 0; i < nparts; ++i) { q[i].win = 1; q[i].force_mtw = cgw; q[i].force_nw = nw; }
    };
    if (win_ok) {
        set_win(parts, 0, 0);
    } else {
        // default: the heuristic geometry of the largest part, lowered until every part agrees
        int m, n;
        parts[0].force_mtw = parts[0].force_nw = 0;
        wgrad_resolved_geom(parts[0], m, n);
        while (!wgrad_common_geom(parts, nparts, m, n) && m > 1) m = m == 6 ? 4 : m / 2;   // (bf16: 8 -> 4)
    }
    for (int i = 0; i < nparts; ++i) parts[i].nsplit = wgrad_pick_nsplit(parts[i]);

    auto run = [&](WgradArgs* q) -> hipError_t {
        int total = 0;
        for (int i = 0; i < nparts; ++i) total += q[i].nsplit;
        if (total == 1) {
            q[0].out = out_w; q[0].direct = 1; q[0].split_base = 0;
            return launch_wgrad(q[0], s);
        }
        // the arena is sized at plan creation for the heuristic split counts with 2x headroom; a policy that asks for
        // more on some shape gets fewer splits, not a failed step
        for (int guard = 0; (long long)total * wgrad_partial_floats(q[0]) > pcap && total > nparts && guard < 32; ++guard) {
            total = 0;
            for (int i = 0; i < nparts; ++i) { q[i].nsplit = (q[i].nsplit + 1) / 2; total += q[i].nsplit; }
        }
        if ((long long)total * wgrad_partial_floats(q[0]) > pcap) return hipErrorOutOfMemory;
        if (total == 1) {
            q[0].out = out_w; q[0].direct = 1; q[0].split_base = 0;
            return launch_wgrad(q[0], s);
        }
        int done = 0;
        for (int i = 0; i < nparts; ++i) {
            q[i].out = partial; q[i].direct = 0; q[i].split_base = done;
            hipError_t e = launch_wgrad(q[i], s);
            if (e != hipSuccess) return e;
            done += q[i].nsplit;
        }
        return launch_wgrad_reduce(q[0], partial, total, out_w, out_b, s);
    };

    if (p->tune_mode == 1) {
        if (p->wg_bwd.size() <= idx) p->wg_bwd.resize(idx + 1, WgradChoice{0, 0, {0, 0}});
        // candidates: shared geometry x per-part split counts, timed with the split reduction (round-robin, see
        // time_candidates); candidate 0 = the heuristic choice
        struct Cand { WgradArgs g[2]; WgradChoice c; };
        std::vector<Cand> cv;
        { Cand c0; for (int i = 0; i < nparts; ++i) c0.g[i] = parts[i]; c0.c = WgradChoice{0, 0, {0, 0}}; cv.push_back(c0); }
        static const int mtws[] = {8, 6, 4, 2, 1};         // (8: bf16 kernel only; 6, 2, 1: exact-fp32 kernel only)
        WgradArgs g[2];
        if (win_ok) {
            // register-window kernel: column tiles per wave x split counts (choice code: mtw = 16 + column groups per workgroup)
            const int ntile = (parts[0].N + 15) / 16;
            int bestpad = 1 << 30;
            for (int nw = 2; nw <= 6; ++nw) bestpad = std::min(bestpad, (ntile + nw - 1) / nw * nw);
            for (int nw = (parts[0].KW == 15 ? 3 : 2); nw <= (parts[0].KW == 15 ? 5 : 6); ++nw) {
                if ((ntile + nw - 1) / nw * nw > bestpad + (bestpad >= 8 ? 1 : 0) && nw != 3) continue;
                for (int i = 0; i < nparts; ++i) g[i] = parts[i];
                set_win(g, 1, nw);
                int basens[2] = {0, 0}, units[2] = {0, 0};
                for (int i = 0; i < nparts; ++i) { basens[i] = wgrad_pick_nsplit(g[i]); units[i] = wgrad_max_units(g[i]); }
                static const int num[5] = {4, 2, 6, 8, 3};          // split factor / 4: 1, 1/2, 3/2, 2, 3/4
                for (int oi = 0; oi < 5; ++oi) {
                    for (int i = 0; i < nparts; ++i) {
                        int ns = basens[i] * num[oi] / 4;
                        if (ns < 1) ns = 1;
                        if (ns > units[i]) ns = units[i];
                        g[i].nsplit = ns;
                    }
                    Cand c;
                    for (int i = 0; i < nparts; ++i) c.g[i] = g[i];
                    c.c = WgradChoice{17, nw, {g[0].nsplit, nparts > 1 ? g[1].nsplit : 0}};
                    cv.push_back(c);
                }
            }
        }
        for (int mi = 0; mi < 5; ++mi)
            for (int nw = 5; nw >= 1; --nw) {
                if (nw > 3 && (mtws[mi] == 6 || parts[0].N <= 48)) continue;
                for (int i = 0; i < nparts; ++i) { g[i] = parts[i]; g[i].win = 0; }
                if (!wgrad_common_geom(g, nparts, mtws[mi], nw)) continue;
                int m, n;
                wgrad_resolved_geom(g[0], m, n);
                if (m != mtws[mi] || n != nw) continue;          // lowered by the staging limit: duplicate
                int basens[2] = {0, 0}, units[2] = {0, 0};
                for (int i = 0; i < nparts; ++i) { basens[i] = wgrad_pick_nsplit(g[i]); units[i] = wgrad_max_units(g[i]); }
                static const int num[4] = {4, 2, 8, 1};            // split factor / 4: 1, 1/2, 2, 1/4
                for (int oi = 0; oi < 4; ++oi) {
                    bool same = oi > 0;
                    for (int i = 0; i < nparts; ++i) {
                        int ns = basens[i] * num[oi] / 4;
                        if (ns < 1) ns = 1;
                        if (ns > units[i]) ns = units[i];
                        if (ns != basens[i]) same = false;
                        g[i].nsplit = ns;
                    }
                    if (same) continue;
                    Cand c;
                    for (int i = 0; i < nparts; ++i) c.g[i] = g[i];
                    c.c = WgradChoice{mtws[mi], nw, {g[0].nsplit, nparts > 1 ? g[1].nsplit : 0}};
                    cv.push_back(c);
                }
            }
        std::vector<float> tms(cv.size());
        time_candidates(p, s, (int)cv.size(), [&](int i) { return run(cv[(size_t)i].g); }, tms.data());
        const float base = tms[0];
        float best = base;
        WgradChoice bc{0, 0, {0, 0}};
        size_t bi = 0;
        for (size_t i = 1; i < cv.size(); ++i)
            if (tms[i] < tms[bi]) bi = i;
        if (bi > 0 && tms[bi] < base * 0.98f) { best = tms[bi]; bc = cv[bi].c; }
        if (const char* af = getenv("WUN_TUNE_ALTS")) {
            if (FILE* f = fopen(af, "a")) {
                const float lim = std::min(best, base) * alts_tol();
                std::vector<size_t> order;
                for (size_t i = 0; i < cv.size(); ++i) if (tms[i] <= lim) order.push_back(i);
                std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return tms[x] < tms[y]; });
                for (size_t k = 0; k < order.size() && k < alts_max(); ++k) {
                    const WgradChoice& c = cv[order[k]].c;
                    fprintf(f, "wg %zu %d %d %d %d %.4f\n", idx, c.mtw, c.nw, c.nsplit[0], c.nsplit[1], tms[order[k]]);
                }
                fclose(f);
            }
        }
        p->wg_bwd[idx] = bc;
        if (getenv("WUN_TUNE_LOG"))
            fprintf(stderr, "[tune wgrad #%zu] C=%d N=%d T=%d K=%d ld=%d parts=%d base(ns=%d) %.3f ms -> mtw=%d nw=%d ns=%d,%d %.3f ms\n",
                    idx, parts[0].C0 + parts[0].C1, parts[0].N, parts[0].Tq, parts[0].KW, parts[0].loader, nparts,
                    parts[0].nsplit, base, bc.mtw, bc.nw, bc.nsplit[0], bc.nsplit[1], best);
    }
    if (p->tune_mode >= 1 && idx < p->wg_bwd.size() && p->wg_bwd[idx].nsplit[0] > 0) {
        const WgradChoice& c = p->wg_bwd[idx];
        const bool cwin = c.mtw == 17;
        bool ok = cwin ? (win_ok && c.nw >= 1 && c.nw <= 6)
                       : ((c.mtw == 1 || c.mtw == 2 || c.mtw == 4 || c.mtw == 6 || c.mtw == 8) && c.nw >= 1 && c.nw <= 5);
        for (int i = 0; ok && i < nparts; ++i) ok = c.nsplit[i] >= 1;
        WgradArgs g[2];
        for (int i = 0; i < nparts; ++i) { g[i] = parts[i]; g[i].win = 0; }
        if (ok && cwin) {
            set_win(g, 1, c.nw);
            for (int i = 0; ok && i < nparts; ++i) ok = c.nsplit[i] <= wgrad_max_units(g[i]);
            if (ok)
                for (int i = 0; i < nparts; ++i) { parts[i] = g[i]; parts[i].nsplit = c.nsplit[i]; }
        } else if (ok && wgrad_common_geom(g, nparts, c.mtw, c.nw)) {
            for (int i = 0; ok && i < nparts; ++i) ok = c.nsplit[i] <= wgrad_max_units(g[i]);
            if (ok)
                for (int i = 0; i < nparts; ++i) { parts[i] = g[i]; parts[i].nsplit = c.nsplit[i]; }
        }
    }
    hipError_t e = run(parts);
    if (e == hipErrorOutOfMemory) return fail(WUN_ERR_INVALID, "internal: wgrad partial buffer too small");
    HIP_TRY(e);
    return WUN_OK;
}

// Narrow layers (audio-input conv, output head): direct reduction kernel instead of MFMA tiles.  All parts
// (a down level's decimated + window positions) write consecutive splits of one partial list; one reduction.
static int run_narrow_wgrad(const wun_plan* p, NarrowWgradArgs* parts, int nparts, const long long* woff,
                            const long long* boff, float* ws, float* grads, hipStream_t main, hipStream_t s) {
    int rcd = WUN_OK;
    hipStream_t side_of_caller = s;                            // (bucket events of the data-parallel path are recorded there)
    // bf16 mode, history (round 5): built WITH packed fp32 VALU instructions, narrow_wgrad_kernel (the LDS-staged
    // form: the output head, audio-input convs with < 4 taps) returned different accumulators from run to run whenever bf16 MFMA
    // kernels ran beside it; round 5 built the unit without them AND, as a second line, ran this launch alone on the caller's
    // stream.  Round 6: tools/probes/pk_fma_probe.hip reproduces the defect stand-alone (the compiler's packed instruction mix beside a
    // v_mfma_f32_16x16x32_bf16 spinner: 2085 of 10000 launches differ; alone, beside an fp32-MFMA spinner, or built without
    // packed ops: 0), the library with packed ops + overlap differs in 60 of 60 probe steps, the shipped build with overlap in
    // 0 of 600 -- so the launch is back on the side stream (~1 % of the bf16 step).  WUN_BF16_HEAD_SERIAL=1: round 5's placement.
    if (p->bf16 && s != main && getenv("WUN_BF16_HEAD_SERIAL") != nullptr) {
        bool lds_form = false;
        for (int i = 0; i < nparts; ++i) lds_form = lds_form || narrow_wgrad_uses_lds(parts[i]);
        if (lds_form) {
            if (p->side != nullptr && (rcd = stream_dep(p, p->side, main))) return rcd;
            if (p->side2 != nullptr && (rcd = stream_dep(p, p->side2, main))) return rcd;
            s = main;
        }
    }
    if (s != main && (rcd = stream_dep(p, main, s))) return rcd;      // everything this launch reads has been issued on `main`
    const long long pcap = p->partial_floats / 2;
    float* partial = ws + p->partial_off + ((p->side2 && s == p->side2) ? pcap : 0);
    int total = 0;
    for (int i = 0; i < nparts; ++i) { parts[i].nsplit = narrow_wgrad_pick_nsplit(parts[i]); total += parts[i].nsplit; }
    const long long P = narrow_wgrad_partial_floats(parts[0]);
    while (total * P > pcap && total > nparts) {               // (never in practice: P is a few hundred floats)
        total = 0;
        for (int i = 0; i < nparts; ++i) { parts[i].nsplit = (parts[i].nsplit + 1) / 2; total += parts[i].nsplit; }
    }
    int done = 0;
    for (int i = 0; i < nparts; ++i) {
        parts[i].partial = partial; parts[i].split_base = done;
        HIP_TRY(launch_narrow_wgrad(parts[i], s));
        done += parts[i].nsplit;
    }
    HIP_TRY(launch_narrow_wgrad_reduce(parts[0], partial, total, grads, woff, boff, s));
    // (moved to `main`: the side stream the caller named is where it records "gradients complete" -- it follows)
    if (s != side_of_caller && (rcd = stream_dep(p, s, side_of_caller))) return rcd;
    return WUN_OK;
}

struct BucketSignal {
    const int64_t* starts; void* const* events; int n; int next;   // buckets in descending start order
    // every gradient at arena offset >= floor is final with respect to stream `st`
    int ready(long long floor, hipStream_t st) {
        while (next < n && starts[next] >= floor) {
            hipError_t e = hipEventRecord((hipEvent_t)events[next], st);
            if (e != hipSuccess) return fail(WUN_ERR_HIP, std::string("hipEventRecord(bucket): ") + hipGetErrorString(e));
            ++next;
        }
        return WUN_OK;
    }
};

extern "C" int wun_loss_backward(const wun_plan* p, const float* params, const float* mix_btc, float* ws,
                                 const float* outputs, const float* targets, float* grads, float* loss,
                                 void* stream) {
    return wun_loss_backward_ex(p, params, mix_btc, ws, outputs, targets, grads, loss, stream, nullptr, nullptr, 0);
}

extern "C" int wun_loss_backward_ex(const wun_plan* p, const float* params, const float* mix_btc, float* ws,
                                    const float* outputs, const float* targets, float* grads, float* loss,
                                    void* stream, const int64_t* bucket_starts, void* const* bucket_events,
                                    int32_t nbuckets) {
    (void)mix_btc;
    if (nbuckets < 0 || (nbuckets > 0 && (!bucket_starts || !bucket_events))) return fail(WUN_ERR_INVALID, "bad bucket arguments");
    for (int k = 1; k < nbuckets; ++k)
        if (bucket_starts[k] >= bucket_starts[k - 1]) return fail(WUN_ERR_INVALID, "bucket_starts must be strictly descending");
    BucketSignal sig{bucket_starts, bucket_events, nbuckets, 0};
    if (!p || !params || !ws || !outputs || !targets || !grads || !loss) return fail(WUN_ERR_INVALID, "null argument");
    if (!p->wt.empty() && !p->dev_wt) return fail(WUN_ERR_HIP, "plan was created without a usable HIP device");
    hipStream_t s = (hipStream_t)stream;
    const int L = p->L, Kd = p->cfg.filter_size, Ku = p->cfg.merge_filter_size, Ko = p->cfg.output_filter_size;
    const bool same = p->same;
    const int padD = same ? (Kd - 1) / 2 : 0, padU = same ? (Ku - 1) / 2 : 0;
    const int F = p->cfg.num_initial_filters, C = p->C;
    int rc;
    if ((rc = side_init(p))) return rc;
    p->ci = 0; p->wi = 0; p->in_bwd = true;
    // side streams: weight gradients + their reductions, alternating between two streams so the
    // ramp-up / drain of consecutive (independent) weight-gradient kernels overlap
    hipStream_t s2 = (p->side && !g_profiling && p->tune_mode != 1) ? p->side : s;
    hipStream_t s3 = (p->side2 && s2 != s) ? p->side2 : s2;
    int wg_rr = 0;
    auto wstream = [&]() { return (wg_rr++ & 1) ? s3 : s2; };
    // bucket events are recorded on s2 once it has also seen everything queued on s3
    auto ready2 = [&](long long floor) -> int {
        if (s3 != s2 && sig.next < sig.n && sig.starts[sig.next] >= floor) {
            int rcj = stream_dep(p, s3, s2);
            if (rcj) return rcj;
        }
        return sig.ready(floor, s2);
    };
    // Weight gradients are queued and flushed one layer at a time: one event on the caller's stream per layer, both side
    // streams wait on it.  (Batching several deep levels behind one event -- every event is a barrier packet that holds
    // the dependent chain for ~7 us -- was measured in round 2: 41 -> 26 stalls per step, but the delayed weight gradients
    // lengthen the tail after the last input gradient by more: 9.12 ms per step with one layer per event, 9.19 - 9.23 with 2 - 5.)
    struct PendingWgrad { WgradArgs w[2]; int n; const ConvLayer* cl; };
    std::vector<PendingWgrad> pend;
    // Early skip-window input gradients (context mode).  The input gradient of down level i is the transposed stride-2
    // conv of dz_dec[i] over the whole row PLUS the full-rate conv of dz_skip[i] over the crop window.  dz_skip[i] is
    // final as soon as up level L-1-i's input gradient has run -- the shallow, FLOP-heavy levels' at the very start of
    // the backward pass -- while the row-wide part can only run when the dependent chain reaches level i at its very
    // end.  The window part is therefore launched as soon as its input exists, on the side streams (it fills the
    // launch-latency-bound deep part of the chain instead of lengthening the FLOP-bound end of it), stores into the
    // window of dz_dec[i-1], and the row-wide conv later ADDS inside the window (ConvArgs.acc_lo / acc_len) and stores
    // outside it: a + b == b + a, results are bit-identical to the old order.  Queued here, issued by the next flush
    // (whose event already orders the side streams behind the producing kernels: no extra packet on the chain).
    // Only the deep levels (input gradient = separate phase launches on a launch-latency-bound chain): same-box A/B
    // 8.84 -> 8.82 ms; moving the FLOP-heavy levels' window parts too changed nothing (8.98 vs 8.99: the end of the backward
    // pass is throughput-bound, not chain-bound).  WUN_EARLY_WINDOW=0 restores the old order (other launch order: the
    // tuning-table header records it).
    const char* ew_env = getenv("WUN_EARLY_WINDOW");
    const bool early_win = !same && !p->bf16 && !(ew_env != nullptr && ew_env[0] == '0');
    auto level_fused = [&](int i) {                                  // (the rule of the down-path loop below)
        const DownShape& d = p->dsh[i];
        ConvArgs f = conv_base(p);
        f.Tin = d.t_dec; f.KW = p->down[i].J0; f.kw_full = Kd; f.N = f.N0 = d.cin; f.Tout = (d.t_in + 1) / 2; f.Tlim = d.t_in;
        f.flags = F_PHASE2; f.C0 = d.cout; f.B = p->B;
        return (d.cin & 3) == 0 && f.Tout >= 256 && conv_natural_wgs_phase2(f) >= 256;
    };
    // dedup plans: ranges of dz_dec[i - 1] that two more writers touch before / beside the row-wide transposed conv of level i --
    // E = the even half of skip window i - 1's gradient (stored by up level L - i's input gradient), W = the input gradient of
    // level i's odd window positions.  The early form of W (it ADDS inside E and stores elsewhere; the row-wide conv then adds
    // inside W) needs E inside W, which the centred crops of every shipped config give; else W runs after the row-wide conv.
    auto e_range = [&](int i, int& lo, int& len) { lo = p->dsh[i].t_ev0 / 2; len = p->dsh[i].n_even; };
    auto w_range = [&](int i, int& lo, int& len) {
        const DownShape& d = p->dsh[i];
        if (p->dedup) { lo = d.t_odd0; len = d.n_odd > 0 ? 2 * (d.n_odd - 1) + Kd : 0; }
        else { lo = d.cs; len = d.tc + Kd - 1; }
    };
    // Which levels' window input gradients leave the dependent chain.  Dedup plans (round 6): ALL of them -- the odd-window
    // launches are half the size of the old window convs, and for the middle levels (row-wide part fused, window part too
    // small to fuse) the chain otherwise carries two phase launches + their split-K epilogues per level: same-box A/B, each arm
    // autotuned, 8.14 -> 8.03 ms per step, 7.94 together with the lower fuse floor below (profiles/round6_ab_dedup_schedule.txt).
    // Rounds 3 - 5 (full-window convs): only the deep levels, moving the FLOP-heavy ones changed nothing (8.98 vs 8.99).
    // WUN_EARLY_WINDOW=deep | all | 0 overrides (a non-default mode is part of the tuning-table header).
    const bool early_all = ew_env != nullptr ? ew_env[0] == 'a' : p->dedup;
    auto level_early = [&](int i) {
        if (!(early_win && i > 0 && (early_all || !level_fused(i)))) return false;
        if (!p->dedup) return true;
        int elo, elen, wlo, wlen;
        e_range(i - 1, elo, elen); w_range(i, wlo, wlen);
        return wlen > 0 && (elen == 0 || (wlo <= elo && elo + elen <= wlo + wlen));
    };
    if (early_win && p->win_ev.size() < (size_t)L) {
        p->win_ev.resize(L, nullptr);
        for (auto& e : p->win_ev)
            if (!e) HIP_TRY(hipEventCreateWithFlags(&e, event_flags()));
    }
    std::vector<int> pend_win;
    std::vector<UpsampleBwdArgs> pend_interp;
    const long long cpart_half = p->conv_part_floats / 2, cpart_q = p->conv_part_floats / 4;
    auto window_dgrad_args = [&](int i) {
        const DownShape& d = p->dsh[i];
        const ConvLayer& cl = p->down[i];
        ConvArgs a = conv_base(p);
        set_src0(a, ws, p->dz_skip[i], 0, d.cout);
        a.Tin = d.tc; a.shift = Kd - 1; a.W = ws + cl.wt_full; a.KW = Kd;
        a.N = a.N0 = d.cin; a.Tout = d.tc + Kd - 1;
        set_dst0(a, ws, p->dz_dec[i - 1], d.cs, &p->dec[i - 1]);
        return a;
    };
    // Transposed stride-2 conv of down level i into dz_dec[i - 1] (masked with dec[i - 1]'s LeakyReLU branch): of the decimated
    // stream's gradient dz_dec[i] over the whole row (odd = false), or -- dedup plans -- of the odd window positions' gradient
    // dz_odd[i] into [t_odd0, t_odd0 + 2 (n_odd - 1) + Kd) (odd = true).  Both output phases fused in one launch (a lane owns 8
    // consecutive outputs) when the launch fills the chip, else one phase at a time (those launches can use split-K).
    // accum: add to what the row holds inside [acc_lo, acc_lo + acc_len) (acc_len == 0: everywhere), store elsewhere.
    auto tconv2 = [&](int i, bool odd, bool accum, int acc_lo, unsigned acc_len, hipStream_t st, float* part, long long cap) -> int {
        const DownShape& d = p->dsh[i];
        const ConvLayer& cl = p->down[i];
        const Buf& src = odd ? p->dz_odd[i] : p->dz_dec[i];
        const int n_in = odd ? d.n_odd : d.t_dec;
        const int out_off = odd ? d.t_odd0 : 0;
        const int out_len = odd ? 2 * (d.n_odd - 1) + Kd : d.t_in;
        ConvArgs f = conv_base(p);
        set_src0(f, ws, src, 0, d.cout);
        f.Tin = n_in; f.KW = cl.J0; f.kw_full = Kd; f.shift = cl.J0 - 1; f.W = ws + cl.wt_ph2;
        f.N = f.N0 = d.cin; f.Tout = (out_len + 1) / 2; f.Tlim = out_len; f.flags = F_PHASE2;
        set_dst0(f, ws, p->dz_dec[i - 1], out_off, &p->dec[i - 1]);
        if (accum) { f.flags |= F_ACCUM; f.acc_lo = acc_lo; f.acc_len = acc_len; }
        // Odd-window part: its outputs start at the odd row position t_odd0 -- scalar read-modify-write stores.  With the
        // filter shifted by one tap (wt_ph2s: the same sums, one leading zero tap) the launch starts at t_odd0 - 1, and -- one
        // more (zero) input position in front when that is not a multiple of 4 -- at t_odd0 - 3: a 16-byte boundary, the vector
        // epilogue.  The leading outputs it adds are sums over zero taps / positions before the first sample: +0 where it
        // accumulates, 0 where it stores (positions the row-wide conv then stores over: they lie outside its accumulate range).
        const bool no_align = getenv("WUN_NO_ODD_ALIGN") != nullptr;
        if (odd && cl.wt_ph2s >= 0 && !no_align) {
            const int base = d.t_odd0 - 1, extra = (base & 3) ? 2 : 0;
            if (base - extra >= 0) {
                f.KW = cl.J0s; f.shift = cl.J0s - 1 + (extra ? 1 : 0); f.W = ws + cl.wt_ph2s;
                const int len2 = out_len + 1 + extra;
                f.Tout = (len2 + 1) / 2; f.Tlim = len2;
                set_dst0(f, ws, p->dz_dec[i - 1], base - extra, &p->dec[i - 1]);
            }
        }
        // (bf16 mode: always fused when the channel count allows -- one launch, the gradient tile staged once,
        //  contiguous 32-byte stores instead of two stride-2 scatter passes)
        // (the odd-window launches fuse from 64 workgroups / 64 output pairs on: they run on the side streams, where one
        //  launch beats two phase launches + two split-K epilogues; WUN_ODD_FUSE_MIN overrides the floor)
        const char* of_env = getenv("WUN_ODD_FUSE_MIN");
        const int odd_min = of_env ? atoi(of_env) : 64;
        const int tmin = odd ? std::min(256, odd_min) : 256, wmin = odd ? odd_min : 256;
        if ((d.cin & 3) == 0 && (p->bf16 || (f.Tout >= tmin && conv_natural_wgs_phase2(f) >= wmin))) {
            HIP_TRY(conv_dispatch(p, f, part, cap, st));
            return WUN_OK;
        }
        for (int ph = 0; ph < 2; ++ph) {
            ConvArgs a = conv_base(p);
            set_src0(a, ws, src, 0, d.cout);
            a.Tin = n_in; a.KW = cl.Jp[ph]; a.shift = cl.Jp[ph] - 1; a.W = ws + cl.wt_ph[ph];
            a.N = a.N0 = d.cin; a.Tout = (out_len - ph + 1) / 2;
            set_dst0(a, ws, p->dz_dec[i - 1], out_off + ph, &p->dec[i - 1]);
            a.ostride = 2;
            if (accum) { a.flags |= F_ACCUM; a.acc_lo = acc_lo; a.acc_len = acc_len; }
            if (a.Tout > 0) HIP_TRY(conv_dispatch(p, a, part, cap, st));
        }
        return WUN_OK;
    };
    auto flush_wgrads = [&]() -> int {
        if (pend.empty() && pend_win.empty() && pend_interp.empty()) return WUN_OK;
        if (s2 != s) {
            hipEvent_t e = p->events[p->ev_next++ % p->events.size()];
            HIP_TRY(hipEventRecord(e, s));
            HIP_TRY(hipStreamWaitEvent(s2, e, 0));
            if (s3 != s2) HIP_TRY(hipStreamWaitEvent(s3, e, 0));
        }
        for (auto& ub : pend_interp) HIP_TRY(launch_interp_grad(ub, wstream()));
        pend_interp.clear();
        for (auto& q : pend) {
            int rcq = run_wgrad(p, q.w, q.n, *q.cl, ws, grads, s, wstream(), false);
            if (rcq) return rcq;
            if ((rcq = ready2(q.cl->woff))) return rcq;
        }
        pend.clear();
        for (int i : pend_win) {
            // own quarter of the split-K scratch per side stream (the chain on `s` uses the first half)
            hipStream_t sw = wstream();
            float* part = ws + p->conv_part_off + cpart_half + ((sw == s3 && s3 != s2) ? cpart_q : 0);
            if (p->dedup) {
                int elo, elen;
                e_range(i - 1, elo, elen);
                int rcw = tconv2(i, true, elen > 0, elo, (unsigned)elen, sw, sw == s ? ws + p->conv_part_off : part, sw == s ? cpart_half : cpart_q);
                if (rcw) return rcw;
            } else {
                HIP_TRY(conv_dispatch(p, window_dgrad_args(i), sw == s ? ws + p->conv_part_off : part, sw == s ? cpart_half : cpart_q, sw));
            }
            if (sw != s) HIP_TRY(hipEventRecord(p->win_ev[(size_t)i], sw));
        }
        pend_win.clear();
        return WUN_OK;
    };
    auto submit_wgrad = [&](const WgradArgs* w, int n, const ConvLayer& cl) -> int {
        PendingWgrad q;
        for (int k = 0; k < n; ++k) q.w[k] = w[k];
        q.n = n; q.cl = &cl;
        pend.push_back(q);
        return flush_wgrads();
    };

    if (p->wt_ready) {
        HIP_TRY(hipStreamWaitEvent(s, p->wt_ev, 0));       // made during the forward pass
        p->wt_ready = false;
    } else {
        HIP_TRY(launch_make_wt(params, ws, p->dev_wt, (int)p->wt.size(), p->wt_max, s));
        if (p->bf16)
            HIP_TRY(launch_pack_bf16(params, ws, p->dev_pack + p->npack_fwd, (int)p->pack.size() - p->npack_fwd, p->pack_max, s));
    }
    p->cur_params = params; p->cur_ws = ws;

    // ---- head: loss, d(pre-activation), d(feature map) ----
    HeadArgs h = head_args(p, params, ws, const_cast<float*>(outputs), 1);
    h.tgt = targets;
    long long hoff[4] = {0, 0, 0, 0};
    for (int i = 0; i < p->Sh; ++i) hoff[i] = p->head[i].woff;
    HIP_TRY(launch_head_bwd_off(h, hoff, s));
    HIP_TRY(launch_loss_finish(h.loss_partial, head_bwd_blocks(h),
                               1.0f / ((float)p->S * (float)p->B * (float)p->Tout * (float)p->C), loss, s));
    bool head_done = false;
    if (p->head16)
        HIP_TRY(launch_cast_rows_bf16(h.dpre, ws + p->dpre16_off, (long long)p->Sh * p->B * C, p->Tout, h.dppitch, p->dp16_pitch, s));
    if (p->Sh > 0 && !p->head16) {
        // every source's output conv in ONE direct-reduction launch (OutputLayer.py:8,15): dz rows = (source, channel)
        NarrowWgradArgs nw;
        memset(&nw, 0, sizeof(nw));
        nw.src0 = ws + p->mix_ncw.off; nw.bs0 = p->mix_ncw.bs; nw.pitch0 = p->mix_ncw.pitch; nw.off0 = p->in_crop_start; nw.C0 = C;
        nw.src1 = ws + p->upo[L - 1].off; nw.bs1 = p->upo[L - 1].bs; nw.pitch1 = p->upo[L - 1].pitch; nw.off1 = 0; nw.C1 = F;
        nw.Tin = p->t_feat; nw.shift = h.padl; nw.KW = Ko; nw.stride = 1;
        nw.dz = h.dpre; nw.zss = h.dps; nw.dzbs = h.dpbs; nw.dzpitch = h.dppitch;
        nw.N = p->Sh * C; nw.Nper = C; nw.Tq = p->Tout; nw.B = p->B;
        nw.et = p->bf16 ? 2 : 0;                                  // fp32 audio + (bf16) feature map, fp32 d(pre-activation)
        if (narrow_wgrad_supported(nw) && (p->bf16 || getenv("WUN_NO_NARROW") == nullptr)) {
            long long woff[4] = {0, 0, 0, 0}, boff[4] = {0, 0, 0, 0};
            for (int sh = 0; sh < p->Sh; ++sh) { woff[sh] = p->head[sh].woff; boff[sh] = p->head[sh].boff; }
            if ((rc = run_narrow_wgrad(p, &nw, 1, woff, boff, ws, grads, s, wstream()))) return rc;
            head_done = true;
        } else if (p->bf16) {
            // bf16 mode: the head's inputs are the fp32 audio and the bf16 feature map -- only the narrow kernels read
            // that mix.  More (input channel, output row) pairs than one launch holds (the deep variant: 50 x 6): one
            // launch per source
            nw.N = nw.Nper = C;
            if (!narrow_wgrad_supported(nw)) return fail(WUN_ERR_UNSUPPORTED, "bf16 mode: output-layer shape not served by the narrow weight-gradient kernels");
            for (int sh = 0; sh < p->Sh; ++sh) {
                NarrowWgradArgs one = nw;
                one.dz = h.dpre + (long long)sh * h.dps;
                const long long woff[4] = {p->head[sh].woff, 0, 0, 0}, boff[4] = {p->head[sh].boff, 0, 0, 0};
                if ((rc = run_narrow_wgrad(p, &one, 1, woff, boff, ws, grads, s, wstream()))) return rc;
            }
            head_done = true;
        }
    }
    for (int sh = 0; sh < p->Sh && !head_done; ++sh) {
        WgradArgs w = wgrad_base(p);
        wset_src0(w, ws, p->head16 ? p->mix16 : p->mix_ncw, p->in_crop_start, C);
        wset_src1(w, ws, p->upo[L - 1], 0, F);
        w.Tin = p->t_feat; w.shift = h.padl; w.KW = Ko;
        if (p->head16)       // (bf16 rows: element strides; the float* base advances by half as many floats)
            wset_dz(w, ws + p->dpre16_off + ((long long)sh * p->B * C * p->dp16_pitch) / 2, (long long)C * p->dp16_pitch, p->dp16_pitch, C, p->Tout);
        else
            wset_dz(w, h.dpre + (long long)sh * h.dps, h.dpbs, h.dppitch, C, p->Tout);
        if ((rc = run_wgrad(p, &w, 1, p->head[sh], ws, grads, s, wstream()))) return rc;
    }
    if (p->Sh > 0 && (rc = ready2(p->head[0].woff))) return rc;

    // ---- up path, last level first ----
    const bool fuse_ups = !p->bf16 && getenv("WUN_NO_FUSE_UPS") == nullptr;
    bool adj_done = false;
    for (int j = L - 1; j >= 0; --j) {
        const UpShape& u = p->ush[j];
        const int i = L - 1 - j;
        {
            WgradArgs w = wgrad_base(p);
            wset_src0(w, ws, p->skip[i], 0, u.c_skip);
            wset_src1(w, ws, p->ups[j], 0, u.c_cur);
            w.Tin = u.t_up; w.shift = padU; w.KW = Ku;
            wset_dz(w, ws + p->dz_upo[j].off, p->dz_upo[j].bs, p->dz_upo[j].pitch, u.cout, u.t_conv);
            // (interp_j, written on `s` by the previous level's upsample_bwd, sits above up[j] in
            // the arena; the flush makes the side streams wait for everything issued on `s` so far)
            if ((rc = submit_wgrad(&w, 1, p->up[j]))) return rc;
        }
        {
            ConvArgs a = conv_base(p);
            set_src0(a, ws, p->dz_upo[j], 0, u.cout);
            a.Tin = u.t_conv; a.shift = Ku - 1 - padU; a.W = ws + p->up[j].wt_full; a.KW = Ku;
            a.N = u.c_skip + u.c_cur; a.N0 = u.c_skip; a.Tout = u.t_up;
            set_dst0(a, ws, p->dz_skip[i], 0, &p->skip[i]);
            set_dst1(a, ws, p->d_ups[j], 0, nullptr);
            if (p->dedup) {
                // window element q sits at absolute conv position cs + q: the even positions are elements of the decimated
                // stream -- their gradient goes into dz_dec[i] (index (cs + q) / 2), the odd ones compact into dz_odd[i]
                const DownShape& d = p->dsh[i];
                float* ev = ws + p->dz_dec[i].off;
                float* od = ws + p->dz_odd[i].off;
                const bool cs_even = (d.cs & 1) == 0;
                a.dec = cs_even ? ev : od;  a.decbs = cs_even ? p->dz_dec[i].bs : p->dz_odd[i].bs;
                a.decpitch = cs_even ? p->dz_dec[i].pitch : p->dz_odd[i].pitch; a.dec_off = cs_even ? d.t_ev0 / 2 : 0;
                a.dec1 = cs_even ? od : ev; a.dec1bs = cs_even ? p->dz_odd[i].bs : p->dz_dec[i].bs;
                a.dec1pitch = cs_even ? p->dz_odd[i].pitch : p->dz_dec[i].pitch; a.dec1_off = cs_even ? 0 : d.t_ev0 / 2;
            }
            const Buf& prev = (j == 0) ? p->bott_out : p->upo[j - 1];
            const Buf& dzprev = (j == 0) ? p->dz_bott : p->dz_upo[j - 1];
            // linear interpolation: a launch that ends in the split-K epilogue kernel applies the adjoint of the 2x
            // upsampling there (ConvArgs.ubw_*) instead of storing d_ups[j] for upsample_bwd_vec_kernel
            if (fuse_ups && p->interp[j] < 0 && dzprev.bs == prev.bs && dzprev.pitch == prev.pitch) {
                a.ubw_dz = ws + dzprev.off; a.ubw_x = ws + prev.off; a.ubw_bs = prev.bs; a.ubw_pitch = prev.pitch;
                a.ubw_n = u.t_cur;
            }
            HIP_TRY(conv_dispatch(p, a, ws + p->conv_part_off, p->conv_part_floats / 2, s));
            adj_done = a.ubw_dz != nullptr && conv_last_fused_ups() != 0;
            if (level_early(i)) pend_win.push_back(i);         // dz_skip[i] is final: its window input gradient can start
        }
        if (!adj_done) {
            const Buf& prev = (j == 0) ? p->bott_out : p->upo[j - 1];
            const Buf& dzprev = (j == 0) ? p->dz_bott : p->dz_upo[j - 1];
            UpsampleBwdArgs ub;
            memset(&ub, 0, sizeof(ub));
            ub.dy = ws + p->d_ups[j].off; ub.ybs = p->d_ups[j].bs; ub.ypitch = p->d_ups[j].pitch; ub.tup = u.t_up;
            ub.x = ws + prev.off; ub.xbs = prev.bs; ub.xpitch = prev.pitch; ub.n = u.t_cur;
            ub.dz = ws + dzprev.off;
            ub.w = p->interp[j] >= 0 ? params + p->interp[j] : nullptr;
            ub.dw = p->interp[j] >= 0 ? grads + p->interp[j] : nullptr;
            ub.dw_partial = (p->interp[j] >= 0 && !p->interp_partial_off.empty()) ? ws + p->interp_partial_off[(size_t)j] : nullptr;
            ub.C = u.c_cur; ub.B = p->B; ub.context = p->cfg.context; ub.bf = p->bf16 ? 1 : 0;
            HIP_TRY(launch_upsample_bwd(ub, s));
            // the interpolation weights' gradient is nobody's input on the chain: with the next flush, on a side stream
            // (interp_<j> lies just below up[j]'s kernel in the arena: complete before the next layer's bucket signal)
            if (ub.dw != nullptr) pend_interp.push_back(ub);
        }
    }

    // ---- bottleneck ----
    {
        WgradArgs w = wgrad_base(p);
        wset_src0(w, ws, p->dec[L - 1], 0, p->bott.Cin);
        w.Tin = p->t_b_in; w.shift = padD; w.KW = Kd;
        wset_dz(w, ws + p->dz_bott.off, p->dz_bott.bs, p->dz_bott.pitch, p->c_b, p->t_b);
        if ((rc = submit_wgrad(&w, 1, p->bott))) return rc;
        ConvArgs a = conv_base(p);
        set_src0(a, ws, p->dz_bott, 0, p->c_b);
        a.Tin = p->t_b; a.shift = Kd - 1 - padD; a.W = ws + p->bott.wt_full; a.KW = Kd;
        a.N = a.N0 = p->bott.Cin; a.Tout = p->t_b_in;
        if (same) {
            set_dst0(a, ws, p->dz_skip[L - 1], 0, &p->skip[L - 1]);
            a.ostride = 2; a.flags = F_ACCUM;
        } else {
            set_dst0(a, ws, p->dz_dec[L - 1], 0, &p->dec[L - 1]);
            if (p->dedup && p->dsh[L - 1].n_even > 0) {
                // (the even half of skip window L-1's gradient is already there)
                int elo, elen;
                e_range(L - 1, elo, elen);
                a.flags = F_ACCUM; a.acc_lo = elo; a.acc_len = (unsigned)elen;
            }
        }
        HIP_TRY(conv_dispatch(p, a, ws + p->conv_part_off, p->conv_part_floats / 2, s));
    }

    // ---- down path ----
    for (int i = L - 1; i >= 0; --i) {
        const DownShape& d = p->dsh[i];
        const ConvLayer& cl = p->down[i];
        const Buf& x = (i == 0) ? p->mix_ncw : p->dec[i - 1];
        // the audio-input conv (1 or 2 input channels): direct reduction instead of MFMA tiles (10 TFLOP/s of mostly
        // padding); WUN_NO_NARROW_DOWN0=1 keeps the MFMA kernel (A/B: 9.36 -> 9.32 ms per step with the narrow kernel)
        NarrowWgradArgs nw[2];
        bool narrow = false;
        if (i == 0) {
            memset(nw, 0, sizeof(nw));
            for (int k = 0; k < 2; ++k) {
                nw[k].src0 = ws + x.off; nw[k].bs0 = x.bs; nw[k].pitch0 = x.pitch; nw[k].C0 = d.cin;
                nw[k].KW = Kd; nw[k].N = nw[k].Nper = d.cout; nw[k].B = p->B;
                nw[k].et = p->bf16 ? 4 : 0;                       // fp32 audio, (bf16) dz
            }
            if (same) {
                nw[0].Tin = d.t_in; nw[0].shift = padD; nw[0].stride = 1; nw[0].off0 = 0;
                nw[0].dz = ws + p->dz_skip[0].off; nw[0].dzbs = p->dz_skip[0].bs; nw[0].dzpitch = p->dz_skip[0].pitch; nw[0].Tq = d.t_conv;
            } else {
                nw[0].Tin = d.t_in; nw[0].shift = 0; nw[0].stride = 2; nw[0].off0 = 0;
                nw[0].dz = ws + p->dz_dec[0].off; nw[0].dzbs = p->dz_dec[0].bs; nw[0].dzpitch = p->dz_dec[0].pitch; nw[0].Tq = d.t_dec;
                if (p->dedup) {
                    nw[1].Tin = d.t_in - d.t_odd0; nw[1].shift = 0; nw[1].stride = 2; nw[1].off0 = d.t_odd0;
                    nw[1].dz = ws + p->dz_odd[0].off; nw[1].dzbs = p->dz_odd[0].bs; nw[1].dzpitch = p->dz_odd[0].pitch; nw[1].Tq = d.n_odd;
                } else {
                    nw[1].Tin = d.tc + Kd - 1; nw[1].shift = 0; nw[1].stride = 1; nw[1].off0 = d.cs;
                    nw[1].dz = ws + p->dz_skip[0].off; nw[1].dzbs = p->dz_skip[0].bs; nw[1].dzpitch = p->dz_skip[0].pitch; nw[1].Tq = d.tc;
                }
            }
            const int nparts0 = same ? 1 : ((p->dedup && d.n_odd == 0) ? 1 : 2);
            narrow = narrow_wgrad_supported(nw[0]) && (nparts0 == 1 || narrow_wgrad_supported(nw[1])) &&
                     (p->bf16 || (getenv("WUN_NO_NARROW") == nullptr && getenv("WUN_NO_NARROW_DOWN0") == nullptr));
            // (bf16 mode: the narrow kernels are the only ones that read fp32 audio against bf16 gradients)
            if (p->bf16 && !narrow) return fail(WUN_ERR_UNSUPPORTED, "bf16 mode: audio-input conv shape not served by the narrow weight-gradient kernels");
        }
        if (narrow) {
            if ((rc = flush_wgrads())) return rc;
            const long long woff[4] = {cl.woff, 0, 0, 0}, boff[4] = {cl.boff, 0, 0, 0};
            if ((rc = run_narrow_wgrad(p, nw, (same || (p->dedup && d.n_odd == 0)) ? 1 : 2, woff, boff, ws, grads, s, wstream()))) return rc;
            if ((rc = ready2(cl.woff))) return rc;
        } else if (same) {
            WgradArgs w = wgrad_base(p);
            wset_src0(w, ws, x, 0, d.cin);
            w.Tin = d.t_in; w.shift = padD; w.KW = Kd;
            wset_dz(w, ws + p->dz_skip[i].off, p->dz_skip[i].bs, p->dz_skip[i].pitch, d.cout, d.t_conv);
            if ((rc = submit_wgrad(&w, 1, cl))) return rc;
            if (i > 0) {
                ConvArgs a = conv_base(p);
                set_src0(a, ws, p->dz_skip[i], 0, d.cout);
                a.Tin = d.t_conv; a.shift = Kd - 1 - padD; a.W = ws + cl.wt_full; a.KW = Kd;
                a.N = a.N0 = d.cin; a.Tout = d.t_in;
                set_dst0(a, ws, p->dz_skip[i - 1], 0, &p->skip[i - 1]);
                a.ostride = 2; a.flags = F_ACCUM;
                HIP_TRY(conv_dispatch(p, a, ws + p->conv_part_off, p->conv_part_floats / 2, s));
            }
        } else {
            WgradArgs w[2];
            w[0] = wgrad_base(p);
            wset_src0(w[0], ws, x, 0, d.cin);
            w[0].loader = LOADER_DEINT; w[0].Tin = d.t_in; w[0].shift = 0; w[0].KW = Kd;
            wset_dz(w[0], ws + p->dz_dec[i].off, p->dz_dec[i].bs, p->dz_dec[i].pitch, d.cout, d.t_dec);
            w[1] = wgrad_base(p);
            int nparts = 2;
            if (p->dedup) {
                // the odd window positions: the same stride-2 geometry over x shifted by t_odd0 samples
                nparts = d.n_odd > 0 ? 2 : 1;
                wset_src0(w[1], ws, x, d.t_odd0, d.cin);
                w[1].loader = LOADER_DEINT; w[1].Tin = d.t_in - d.t_odd0; w[1].shift = 0; w[1].KW = Kd;
                wset_dz(w[1], ws + p->dz_odd[i].off, p->dz_odd[i].bs, p->dz_odd[i].pitch, d.cout, d.n_odd);
            } else {
                wset_src0(w[1], ws, x, d.cs, d.cin);
                w[1].Tin = d.tc + Kd - 1; w[1].shift = 0; w[1].KW = Kd;
                wset_dz(w[1], ws + p->dz_skip[i].off, p->dz_skip[i].bs, p->dz_skip[i].pitch, d.cout, d.tc);
            }
            if ((rc = submit_wgrad(w, nparts, cl))) return rc;
            if (i > 0) {
                const bool win_early = level_early(i) && !p->win_ev.empty();
                int alo = 0, alen = 0;
                bool acc = false;
                if (win_early) {
                    // the window part is already in dz_dec[i-1] (side stream): wait for it, add inside the window (a
                    // launch still sitting in the queue -- win_ev[i] would be last step's record -- is issued now)
                    if (std::find(pend_win.begin(), pend_win.end(), i) != pend_win.end() && (rc = flush_wgrads())) return rc;
                    if (s2 != s) HIP_TRY(hipStreamWaitEvent(s, p->win_ev[(size_t)i], 0));
                    w_range(i, alo, alen);
                    acc = true;
                } else if (p->dedup) {
                    e_range(i - 1, alo, alen);        // the even half of skip window i-1's gradient is already there
                    acc = alen > 0;
                }
                if ((rc = tconv2(i, false, acc, alo, (unsigned)alen, s, ws + p->conv_part_off, p->conv_part_floats / 2))) return rc;
                if (!win_early) {
                    if (p->dedup) {
                        if (d.n_odd > 0 && (rc = tconv2(i, true, true, 0, 0u, s, ws + p->conv_part_off, p->conv_part_floats / 2))) return rc;
                    } else {
                        ConvArgs a = window_dgrad_args(i);
                        a.flags = F_ACCUM;
                        HIP_TRY(conv_dispatch(p, a, ws + p->conv_part_off, p->conv_part_floats / 2, s));
                    }
                }
            }
        }
    }
    if ((rc = flush_wgrads())) return rc;
    if ((rc = stream_dep(p, s3, s))) return rc;
    if ((rc = stream_dep(p, s2, s))) return rc;      // all gradients are complete w.r.t. `stream`
    if ((rc = sig.ready(0, s))) return rc;           // any bucket not yet signalled (e.g. single-stream mode)
    return WUN_OK;
}

extern "C" int wun_plan_tune(const wun_plan* p, const float* params, const float* mix_btc, float* ws,
                             float* outputs, const float* targets, float* grads, float* loss, void* stream) {
    if (!p) return fail(WUN_ERR_INVALID, "null argument");
    if (!p->tev0) {
        HIP_TRY(hipEventCreate(&p->tev0));
        HIP_TRY(hipEventCreate(&p->tev1));
    }
    p->conv_fwd.clear(); p->conv_bwd.clear(); p->wg_bwd.clear();
    p->tune_mode = 1;
    int rc = wun_forward(p, params, mix_btc, ws, outputs, 1, stream);
    if (rc == WUN_OK) rc = wun_loss_backward(p, params, mix_btc, ws, outputs, targets, grads, loss, stream);
    const hipError_t sync = hipStreamSynchronize((hipStream_t)stream);
    p->tune_mode = (rc == WUN_OK && sync == hipSuccess) ? 2 : 0;     // never left in measuring mode
    if (rc == WUN_OK && sync != hipSuccess)
        return fail(WUN_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(sync));
    return rc;
}

// Tuning-table header: identifies the plan (every config key that changes a launch), the launch
// order of this library build and the number of entries per section, so a table written for
// another plan, another library build or truncated on disk is rejected at import.
#define WUN_TUNE_ORDER "r6a"      /* bump whenever the order / number of conv or wgrad launches changes */
#define WUN_TUNE_ORDER_BF16 "r5b" /* ... of the bf16 mode (round 5: bf16 activations in HBM, other tile menu limits) */
static std::string tune_header(const wun_plan* p, size_t ncf, size_t ncb, size_t nwg) {
    char line[320];
    const wun_config& c = p->cfg;
    snprintf(line, sizeof(line),
             "wun-tune 2 order=%s variants=%d B=%d Tin=%lld L=%d F=%d K=%d,%d,%d ups=%d out=%d ctx=%d S=%d C=%d act=%d "
             "dt=%d arena=%lld cf=%zu cb=%zu wg=%zu",
             p->bf16 ? WUN_TUNE_ORDER_BF16 : WUN_TUNE_ORDER, conv_num_variants(), p->B, (long long)p->Tin, p->L, c.num_initial_filters, c.filter_size,
             c.merge_filter_size, c.output_filter_size, c.upsampling, c.output_type, c.context, c.num_sources,
             c.num_channels, c.output_activation, c.compute_dtype, (long long)p->arena, ncf, ncb, nwg);
    std::string h = line;
    // a non-default early-window mode changes the order of the backward conv launches: such tables only match themselves
    if (const char* ew = getenv("WUN_EARLY_WINDOW")) {
        if (ew[0] == '0') h += " ew=0";
        if (ew[0] == 'a' && !p->dedup) h += " ew=all";
        if (ew[0] == 'd' && p->dedup) h += " ew=deep";
    }
    if (const char* of = getenv("WUN_ODD_FUSE_MIN")) h += std::string(" oddfuse=") + of;
    if (getenv("WUN_NO_ODD_ALIGN") != nullptr && p->dedup) h += " oddalign=0";
    if (!p->same && !p->bf16 && !p->dedup) h += " dedup=0";       // (WUN_NO_DEDUP=1: rounds 1 - 5's launch sequence)
    return h;
}

extern "C" int wun_plan_tune_export(const wun_plan* p, char* buf, int64_t cap) {
    if (!p || !buf) return fail(WUN_ERR_INVALID, "null argument");
    if (p->tune_mode != 2) return fail(WUN_ERR_INVALID, "plan has not been tuned");
    std::string out = tune_header(p, p->conv_fwd.size(), p->conv_bwd.size(), p->wg_bwd.size()) + "\n";
    char line[128];
    auto dump = [&](const char* tag, const std::vector<ConvChoice>& v) {
        for (const ConvChoice& c : v) { snprintf(line, sizeof(line), "%s %d %d\n", tag, c.variant, c.ksplit); out += line; }
    };
    dump("cf", p->conv_fwd);
    dump("cb", p->conv_bwd);
    for (const WgradChoice& c : p->wg_bwd) {
        snprintf(line, sizeof(line), "wg %d %d %d %d\n", c.mtw, c.nw, c.nsplit[0], c.nsplit[1]);
        out += line;
    }
    out += "end\n";
    if ((int64_t)out.size() + 1 > cap) return fail(WUN_ERR_INVALID, "buffer too small for the tuning table");
    memcpy(buf, out.c_str(), out.size() + 1);
    return WUN_OK;
}

extern "C" int wun_plan_tune_import(const wun_plan* p, const char* text) {
    if (!p || !text) return fail(WUN_ERR_INVALID, "null argument");
    const char* nl = strchr(text, '\n');
    if (!nl) return fail(WUN_ERR_INVALID, "malformed tuning table");
    const std::string head(text, (size_t)(nl - text));
    size_t ncf = 0, ncb = 0, nwg = 0;
    {
        const size_t pos = head.rfind(" cf=");
        if (pos == std::string::npos || sscanf(head.c_str() + pos, " cf=%zu cb=%zu wg=%zu", &ncf, &ncb, &nwg) != 3)
            return fail(WUN_ERR_INVALID, "tuning table belongs to a different plan or library build");
    }
    if (head != tune_header(p, ncf, ncb, nwg))
        return fail(WUN_ERR_INVALID, "tuning table belongs to a different plan or library build");
    std::vector<ConvChoice> cf, cb;
    std::vector<WgradChoice> wg;
    const char* q = nl + 1;
    bool ended = false;
    const int nvar = conv_num_variants();
    while (*q) {
        int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        if (strncmp(q, "end", 3) == 0) { ended = true; break; }
        if (sscanf(q, "cf %d %d", &a0, &a1) == 2 && q[1] == 'f') cf.push_back(ConvChoice{a0, a1});
        else if (sscanf(q, "cb %d %d", &a0, &a1) == 2 && q[1] == 'b') cb.push_back(ConvChoice{a0, a1});
        else if (sscanf(q, "wg %d %d %d %d", &a0, &a1, &a2, &a3) == 4) wg.push_back(WgradChoice{a0, a1, {a2, a3}});
        else return fail(WUN_ERR_INVALID, "malformed tuning table");
        const char* e = strchr(q, '\n');
        if (!e) break;
        q = e + 1;
    }
    if (!ended || cf.size() != ncf || cb.size() != ncb || wg.size() != nwg)
        return fail(WUN_ERR_INVALID, "truncated tuning table");
    for (const std::vector<ConvChoice>* v : {&cf, &cb})
        for (const ConvChoice& c : *v)
            if (c.variant < -1 || (c.variant >= nvar && !(c.variant >= kBf16VariantBase && c.variant < kBf16VariantBase + 27)) ||
                c.ksplit < 0 || c.ksplit > 64)
                return fail(WUN_ERR_INVALID, "tuning table entry out of range");
    for (const WgradChoice& c : wg)
        if (c.nsplit[0] < 0 || c.nsplit[1] < 0 || c.mtw < 0 || (c.mtw > 8 && c.mtw != 17) || c.nw < 0 || c.nw > 6)
            return fail(WUN_ERR_INVALID, "tuning table entry out of range");
    // (whether each entry is a legal choice for the launch at its position is checked when it is used)
    p->conv_fwd = cf; p->conv_bwd = cb; p->wg_bwd = wg;
    p->tune_mode = 2;
    return WUN_OK;
}

extern "C" int wun_adam_step(const wun_plan* p, float* params, const float* grads, float* m, float* v,
                             int64_t step, float lr, float beta1, float beta2, float eps, float grad_scale,
                             void* stream) {
    if (!p || !params || !grads || !m || !v) return fail(WUN_ERR_INVALID, "null argument");
    if (step < 1) return fail(WUN_ERR_INVALID, "step is 1-based");
    const double lr_t = (double)lr * std::sqrt(1.0 - std::pow((double)beta2, (double)step)) /
                        (1.0 - std::pow((double)beta1, (double)step));
    HIP_TRY(launch_adam(params, grads, m, v, p->arena, (float)lr_t, beta1, beta2, eps, grad_scale, (hipStream_t)stream));
    return WUN_OK;
}

// ---------------------------------------------------------------------------------------
// single operators
// ---------------------------------------------------------------------------------------
// the single-operator entry points use a lazily allocated split-K scratch of their own
static const long long kOpScratchFloats = 8ll << 20;
static int g_op_variant = -1, g_op_ksplit = 0;          // wun_op_force_conv_variant (test hook)
static int g_op_wg_mtw = 0, g_op_wg_nw = 0, g_op_wg_nsplit = 0;   // wun_op_force_wgrad_variant (test hook)
static int g_op_wg_bf16 = 0;                                       // wun_op_set_wgrad_bf16 (test hook)
static int g_op_wg_narrow = 0;                                     // wun_op_set_wgrad_narrow (test hook)
static int g_op_wg_win = 0;                                        // wun_op_set_wgrad_win (test hook)
static float* g_op_copy0 = nullptr; static float* g_op_copy1 = nullptr;   // wun_op_set_conv_copies (test hook)
static int g_op_copy_t0 = 0, g_op_copy_t1 = 0, g_op_copy_exp = 0, g_op_copy_lo = 0, g_op_copy_len = 0, g_op_acc_lo = 0, g_op_acc_len = 0;
static float* op_scratch() {
    static float* buf = nullptr;
    if (!buf && hipMalloc((void**)&buf, kOpScratchFloats * sizeof(float)) != hipSuccess) {
        buf = nullptr;
        (void)hipGetLastError();
    }
    return buf;
}

// The bf16 kernels read bf16 rows (the plan's activations are born bf16); the single-operator entry points receive fp32
// tensors and convert them first -- rounding to nearest even, exactly what "operands rounded to bf16" means -- into a
// process-wide temporary (slot 0 / 1) that grows on demand.  Rows are re-pitched to 16 bytes.
static void* op_bf16_tmp(int slot, size_t bytes) {
    static void* buf[2] = {nullptr, nullptr};
    static size_t cap[2] = {0, 0};
    if (bytes > cap[slot]) {
        (void)hipDeviceSynchronize();
        if (buf[slot]) (void)hipFree(buf[slot]);
        buf[slot] = nullptr; cap[slot] = 0;
        const size_t want = bytes + (bytes >> 2) + 4096;
        if (hipMalloc(&buf[slot], want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        cap[slot] = want;
    }
    return buf[slot];
}
static inline int pad8(int t) { return (t + 7) / 8 * 8; }
// fp32 [rows][spitch] (T valid) -> bf16 [rows][pad8(T)] in temporary `slot`; returns the bf16 base or null
static const float* op_to_bf16(int slot, const float* src, long long rows, int T, long long spitch, hipStream_t s) {
    void* dst = op_bf16_tmp(slot, (size_t)rows * pad8(T) * 2 + 64);
    if (!dst) return nullptr;
    if (launch_cast_rows_bf16(src, dst, rows, T, spitch, pad8(T), s) != hipSuccess) return nullptr;
    return reinterpret_cast<const float*>(dst);
}

static hipError_t op_launch_conv(ConvArgs a, hipStream_t s) {
    if (g_op_variant >= 0) { a.force_variant = g_op_variant + 1; a.force_ksplit = (a.flags & F_PHASE2) ? 0 : g_op_ksplit; }
    return launch_conv(a, op_scratch(), kOpScratchFloats, s);
}

static void op_src(ConvArgs& a, const float* x, int C, int T) {
    const int pitch = T;
    a.src0 = x; a.bs0 = (long long)C * pitch; a.pitch0 = pitch; a.off0 = 0; a.C0 = C;
}

extern "C" int wun_op_conv1d(const float* x, const float* w, const float* bias, float* y, int batch, int cin,
                             int cout, int k, int t_in, int t_out, int stride, int pad_left, int lrelu,
                             void* stream) {
    if (!x || !w || !y) return fail(WUN_ERR_INVALID, "null argument");
    if (stride != 1 && stride != 2) return fail(WUN_ERR_UNSUPPORTED, "stride must be 1 or 2");
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.B = batch; a.ostride = 1;
    op_src(a, x, cin, t_in);
    a.loader = stride == 2 ? LOADER_DEINT : LOADER_DIRECT;
    a.Tin = t_in; a.shift = pad_left; a.W = w; a.bias = bias; a.KW = k; a.N = a.N0 = cout; a.Tout = t_out;
    a.flags = lrelu ? F_LRELU : 0;
    a.dst0 = y; a.obs0 = (long long)cout * t_out; a.opitch0 = t_out;
    HIP_TRY(op_launch_conv(a, (hipStream_t)stream));
    return WUN_OK;
}

static inline int pad4(int t) { return (t + 3) / 4 * 4; }

static WgradArgs op_wgrad_args(const float* x, const float* dz, int batch, int cin, int cout, int k, int t_in,
                               int t_out, int stride, int pad_left, int xp, int zp) {
    WgradArgs w;
    memset(&w, 0, sizeof(w));
    w.B = batch; w.loader = stride == 2 ? LOADER_DEINT : LOADER_DIRECT;
    w.src0 = x; w.bs0 = (long long)cin * xp; w.pitch0 = xp; w.C0 = cin;
    w.Tin = t_in; w.shift = pad_left; w.KW = k;
    w.dz = dz; w.dzbs = (long long)cout * zp; w.dzpitch = zp; w.N = cout; w.Tq = t_out;
    return w;
}

// split partials of one loader kind under the current (possibly forced) geometry / split count
static long long op_wgrad_part_floats(int batch, int cin, int cout, int k, int t_out, int loader) {
    WgradArgs a = wgrad_shape_only(batch, cin, 0, k, loader, cout, t_out);
    a.bf16 = (g_op_wg_bf16 && wgrad_bf16_supported(a)) ? 1 : 0;
    a.win = (g_op_wg_win && !a.bf16) ? 1 : 0;
    if (a.win && !wgrad_win_supported(a)) a.win = 0;
    if (g_op_wg_mtw > 0) { a.force_mtw = g_op_wg_mtw; a.force_nw = g_op_wg_nw; }
    long long ns = wgrad_pick_nsplit(a);
    if (g_op_wg_nsplit > 0) ns = std::min(g_op_wg_nsplit, wgrad_max_units(a));
    if (g_op_wg_nsplit < 0 && a.win) ns = std::min(std::max(1, -g_op_wg_nsplit / wgrad_win_tiles(a)), wgrad_max_units(a));
    return ns * wgrad_partial_floats(a);
}

extern "C" int64_t wun_op_conv1d_wgrad_scratch(int batch, int cin, int cout, int k, int t_out) {
    // split partials (worst case over both loaders) + repacked copies of x (t_in <= 2*t_out + k) and dz
    long long part = std::max(op_wgrad_part_floats(batch, cin, cout, k, t_out, LOADER_DIRECT),
                              op_wgrad_part_floats(batch, cin, cout, k, t_out, LOADER_DEINT));
    if (g_op_wg_narrow) {
        // wun_op_set_wgrad_narrow(1): the direct-reduction kernels keep one (k * cin + 1) * cout vector per split
        NarrowWgradArgs nw;
        memset(&nw, 0, sizeof(nw));
        nw.C0 = cin; nw.KW = k; nw.N = nw.Nper = cout; nw.Tq = t_out; nw.B = batch;
        for (int stride = 1; stride <= 2; ++stride) {
            nw.stride = stride;
            part = std::max(part, (long long)narrow_wgrad_pick_nsplit(nw) * narrow_wgrad_partial_floats(nw));
        }
    }
    const long long tin_max = 2ll * t_out + k + 8;
    return part + (long long)batch * cin * pad4((int)tin_max) + (long long)batch * cout * pad4(t_out) + 512;
}

extern "C" int wun_op_conv1d_wgrad(const float* x, const float* dz, float* dw, float* db, float* scratch,
                                   int batch, int cin, int cout, int k, int t_in, int t_out, int stride,
                                   int pad_left, void* stream) {
    if (!x || !dz || !dw || !db || !scratch) return fail(WUN_ERR_INVALID, "null argument");
    if (stride != 1 && stride != 2) return fail(WUN_ERR_UNSUPPORTED, "stride must be 1 or 2");
    if (t_in > 2ll * t_out + k + 8) return fail(WUN_ERR_INVALID, "t_in larger than the conv can consume");
    hipStream_t s = (hipStream_t)stream;
    // repack x / dz into the canonical 4-padded row layout the kernels use
    const int xp = pad4(t_in), zp = pad4(t_out);
    float* xs = scratch;                                   // 64-float aligned by construction below
    xs = (float*)(((uintptr_t)xs + 255) & ~(uintptr_t)255);
    float* zs = xs + (long long)batch * cin * xp;
    float* part = zs + (long long)batch * cout * zp;
    HIP_TRY(hipMemcpy2DAsync(xs, (size_t)xp * 4, x, (size_t)t_in * 4, (size_t)t_in * 4, (size_t)batch * cin,
                             hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpy2DAsync(zs, (size_t)zp * 4, dz, (size_t)t_out * 4, (size_t)t_out * 4, (size_t)batch * cout,
                             hipMemcpyDeviceToDevice, s));
    if (g_op_wg_narrow) {
        // the direct-reduction kernels of wun_narrow.hip (what the plan runs for the audio-input conv and the head)
        NarrowWgradArgs nw;
        memset(&nw, 0, sizeof(nw));
        nw.src0 = xs; nw.bs0 = (long long)cin * xp; nw.pitch0 = xp; nw.off0 = 0; nw.C0 = cin;
        nw.Tin = t_in; nw.shift = pad_left; nw.KW = k; nw.stride = stride;
        nw.dz = zs; nw.zss = 0; nw.dzbs = (long long)cout * zp; nw.dzpitch = zp;
        nw.N = nw.Nper = cout; nw.Tq = t_out; nw.B = batch;
        if (!narrow_wgrad_supported(nw)) return fail(WUN_ERR_UNSUPPORTED, "shape not served by the narrow weight-gradient kernels");
        nw.nsplit = narrow_wgrad_pick_nsplit(nw);
        part = (float*)(((uintptr_t)part + 255) & ~(uintptr_t)255);
        nw.partial = part; nw.split_base = 0;
        HIP_TRY(launch_narrow_wgrad(nw, s));
        const long long woff[4] = {0, 0, 0, 0}, boff[4] = {(long long)(db - dw), 0, 0, 0};
        HIP_TRY(launch_narrow_wgrad_reduce(nw, part, nw.nsplit, dw, woff, boff, s));
        return WUN_OK;
    }
    WgradArgs w = op_wgrad_args(xs, zs, batch, cin, cout, k, t_in, t_out, stride, pad_left, xp, zp);
    if (g_op_wg_bf16 && !wgrad_bf16_supported(w)) return fail(WUN_ERR_UNSUPPORTED, "shape not served by the bf16 weight-gradient kernel");
    w.bf16 = g_op_wg_bf16;
    w.win = (g_op_wg_win && !w.bf16) ? 1 : 0;
    if (w.win && !wgrad_win_supported(w)) return fail(WUN_ERR_UNSUPPORTED, "shape not served by the register-window weight-gradient kernel");
    if (g_op_wg_mtw > 0) {
        w.force_mtw = g_op_wg_mtw; w.force_nw = g_op_wg_nw;
        int m, n;
        wgrad_resolved_geom(w, m, n);
        if (m != g_op_wg_mtw || n != g_op_wg_nw)
            return fail(WUN_ERR_UNSUPPORTED, "forced weight-gradient tile geometry is not available for this shape");
    }
    w.nsplit = wgrad_pick_nsplit(w);
    if (g_op_wg_nsplit > 0) {
        w.nsplit = std::min(g_op_wg_nsplit, wgrad_max_units(w));
    }
    if (g_op_wg_nsplit < 0 && w.win)       // (window kernel: a negative count is a target grid size)
        w.nsplit = std::min(std::max(1, -g_op_wg_nsplit / wgrad_win_tiles(w)), wgrad_max_units(w));
    part = (float*)(((uintptr_t)part + 255) & ~(uintptr_t)255);
    w.out = part; w.direct = 0; w.split_base = 0;      // always through the split reduction (dw and db are separate buffers)
    if (w.bf16) {
        // the bf16 kernel reads bf16 rows: convert the repacked copies of x and dz
        w.src0 = op_to_bf16(0, xs, (long long)batch * cin, t_in, xp, s);
        w.dz = op_to_bf16(1, zs, (long long)batch * cout, t_out, zp, s);
        if (!w.src0 || !w.dz) return fail(WUN_ERR_NOMEM, "bf16 temporary");
        w.pitch0 = pad8(t_in); w.bs0 = (long long)cin * w.pitch0;
        w.dzpitch = pad8(t_out); w.dzbs = (long long)cout * w.dzpitch;
        w.sbf = 1;
    }
    HIP_TRY(launch_wgrad(w, s));
    HIP_TRY(launch_wgrad_reduce(w, part, w.nsplit, dw, db, s));
    return WUN_OK;
}

extern "C" int wun_op_conv1d_dgrad(const float* dz, const float* w, float* dx, float* wt_scratch, int batch,
                                   int cin, int cout, int k, int t_in, int t_out, int stride, int pad_left,
                                   void* stream) {
    if (!dz || !w || !dx || !wt_scratch) return fail(WUN_ERR_INVALID, "null argument");
    if (stride != 1 && stride != 2) return fail(WUN_ERR_UNSUPPORTED, "stride must be 1 or 2");
    hipStream_t s = (hipStream_t)stream;
    if (stride == 1) {
        WtDesc d; d.src_off = 0; d.dst_off = 0; d.J = k; d.C = cin; d.N = cout; d.k_last = k - 1; d.k_step = 1; d.mode = 0;
        HIP_TRY(launch_make_wt_one(w, wt_scratch, d, s));
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        a.B = batch; a.ostride = 1;
        op_src(a, dz, cout, t_out);
        a.Tin = t_out; a.shift = k - 1 - pad_left; a.W = wt_scratch; a.KW = k; a.N = a.N0 = cin; a.Tout = t_in;
        a.dst0 = dx; a.obs0 = (long long)cin * t_in; a.opitch0 = t_in;
        HIP_TRY(op_launch_conv(a, s));
    } else {
        if (pad_left != 0) return fail(WUN_ERR_UNSUPPORTED, "stride-2 dgrad supports pad_left == 0 only");
        const int J0 = (k + 1) / 2;
        ConvArgs f;
        memset(&f, 0, sizeof(f));
        f.B = batch; f.ostride = 1;
        op_src(f, dz, cout, t_out);
        f.Tin = t_out; f.KW = J0; f.kw_full = k; f.shift = J0 - 1; f.W = wt_scratch; f.N = f.N0 = cin;
        f.Tout = (t_in + 1) / 2; f.Tlim = t_in; f.flags = F_PHASE2;
        f.dst0 = dx; f.obs0 = (long long)cin * t_in; f.opitch0 = t_in;
        if ((cin & 3) == 0 && conv_natural_wgs_phase2(f) >= 64) {
            WtDesc d; d.src_off = 0; d.dst_off = 0; d.J = J0; d.C = cin; d.N = cout; d.k_last = 2 * (J0 - 1);
            d.k_step = k; d.mode = 1;
            HIP_TRY(launch_make_wt_one(w, wt_scratch, d, s));
            HIP_TRY(op_launch_conv(f, s));
            return WUN_OK;
        }
        for (int ph = 0; ph < 2; ++ph) {
            const int Jp = (k - ph + 1) / 2;
            float* wt = wt_scratch + (long long)ph * k * cin * cout;
            WtDesc d; d.src_off = 0; d.dst_off = 0; d.J = Jp; d.C = cin; d.N = cout;
            d.k_last = 2 * (Jp - 1) + ph; d.k_step = 2; d.mode = 0;
            if (Jp > 0) HIP_TRY(launch_make_wt_one(w, wt, d, s));
            ConvArgs a;
            memset(&a, 0, sizeof(a));
            a.B = batch; a.ostride = 2;
            op_src(a, dz, cout, t_out);
            a.Tin = t_out; a.KW = Jp; a.shift = Jp - 1; a.W = wt; a.N = a.N0 = cin; a.Tout = (t_in - ph + 1) / 2;
            a.dst0 = dx; a.obs0 = (long long)cin * t_in; a.opitch0 = t_in; a.ooff0 = ph;
            HIP_TRY(op_launch_conv(a, s));
        }
    }
    return WUN_OK;
}

extern "C" int wun_op_force_conv_variant(int variant, int ksplit) {
    g_op_variant = variant; g_op_ksplit = ksplit;
    return WUN_OK;
}

extern "C" int wun_op_num_conv_variants(void) { return conv_num_variants(); }

extern "C" int wun_op_set_wgrad_bf16(int on) { g_op_wg_bf16 = on ? 1 : 0; return WUN_OK; }
extern "C" int wun_op_set_wgrad_win(int on) { g_op_wg_win = on ? 1 : 0; return WUN_OK; }
extern "C" int wun_op_set_wgrad_narrow(int on) { g_op_wg_narrow = on ? 1 : 0; return WUN_OK; }

extern "C" int wun_op_force_wgrad_variant(int mtw, int nw, int nsplit) {
    g_op_wg_mtw = mtw; g_op_wg_nw = nw; g_op_wg_nsplit = nsplit;
    return WUN_OK;
}

// General form of the conv launch the plan uses: virtual channel-concat of two sources (crop_and_concat,
// Utils.py:11-24), accumulate into the destination, LeakyReLU-derivative mask, output stride / offset.
extern "C" int wun_op_conv1d_ex(const float* x0, int c0, const float* x1, int c1, const float* w, const float* bias,
                                float* y, const float* mask, int batch, int cout, int k, int t_in, int t_out,
                                int t_y, int stride, int pad_left, int lrelu, int accumulate, int ostride, int ooff,
                                void* stream) {
    if (!x0 || !w || !y) return fail(WUN_ERR_INVALID, "null argument");
    if (stride != 1 && stride != 2) return fail(WUN_ERR_UNSUPPORTED, "stride must be 1 or 2");
    if (c0 < 1 || c1 < 0 || (c1 > 0 && !x1)) return fail(WUN_ERR_INVALID, "bad source channels");
    if (ostride < 1 || ooff < 0 || (long long)(t_out - 1) * ostride + ooff >= t_y) return fail(WUN_ERR_INVALID, "output does not fit t_y");
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.B = batch; a.ostride = ostride;
    a.src0 = x0; a.bs0 = (long long)c0 * t_in; a.pitch0 = t_in; a.C0 = c0;
    if (c1 > 0) { a.src1 = x1; a.bs1 = (long long)c1 * t_in; a.pitch1 = t_in; a.C1 = c1; }
    a.loader = stride == 2 ? LOADER_DEINT : LOADER_DIRECT;
    a.Tin = t_in; a.shift = pad_left; a.W = w; a.bias = bias; a.KW = k; a.N = a.N0 = cout; a.Tout = t_out;
    a.flags = (lrelu ? F_LRELU : 0) | (accumulate ? F_ACCUM : 0);
    a.dst0 = y; a.obs0 = (long long)cout * t_y; a.opitch0 = t_y; a.ooff0 = ooff; a.msk0 = mask;
    if (g_op_copy0 != nullptr) {
        a.dec = g_op_copy0; a.decpitch = g_op_copy_t0; a.decbs = (long long)cout * g_op_copy_t0;
        a.dec_exp = g_op_copy_exp; a.dec_lo = g_op_copy_lo; a.dec_len = (unsigned)g_op_copy_len;
    }
    if (g_op_copy1 != nullptr) { a.dec1 = g_op_copy1; a.dec1pitch = g_op_copy_t1; a.dec1bs = (long long)cout * g_op_copy_t1; }
    if (accumulate && g_op_acc_len > 0) { a.acc_lo = g_op_acc_lo; a.acc_len = (unsigned)g_op_acc_len; }
    HIP_TRY(op_launch_conv(a, (hipStream_t)stream));
    return WUN_OK;
}

extern "C" int wun_op_set_conv_copies(float* copy0, int t0, int expand, int exp_lo, int exp_len, float* copy1, int t1,
                                      int acc_lo, int acc_len) {
    if ((copy0 && t0 < 1) || (copy1 && t1 < 1) || exp_len < 0 || acc_len < 0) return fail(WUN_ERR_INVALID, "bad copy geometry");
    g_op_copy0 = copy0; g_op_copy_t0 = t0; g_op_copy_exp = expand ? 1 : 0; g_op_copy_lo = exp_lo; g_op_copy_len = exp_len;
    g_op_copy1 = copy1; g_op_copy_t1 = t1; g_op_acc_lo = acc_lo; g_op_acc_len = acc_len;
    return WUN_OK;
}

// bf16-MFMA conv as a single operator: packs w (fp32 [K][Cin][Cout]) into the bf16 image in `scratch`
// (>= wun_op_conv1d_bf16_scratch floats), then runs the bf16 kernel.  Same semantics as wun_op_conv1d.
extern "C" int64_t wun_op_conv1d_bf16_scratch(int cin, int cout, int k) {
    return (int64_t)k * bf16_image_groups(cin) * ((cout + 63) / 64 * 64) * 4 + 64;
}

extern "C" int wun_op_conv1d_bf16(const float* x, const float* w, const float* bias, float* y, float* scratch,
                                  int batch, int cin, int cout, int k, int t_in, int t_out, int stride, int pad_left,
                                  int lrelu, void* stream) {
    if (!x || !w || !y || !scratch) return fail(WUN_ERR_INVALID, "null argument");
    if (stride != 1 && stride != 2) return fail(WUN_ERR_UNSUPPORTED, "stride must be 1 or 2");
    hipStream_t s = (hipStream_t)stream;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.B = batch; a.ostride = 1;
    op_src(a, x, cin, t_in);
    a.loader = stride == 2 ? LOADER_DEINT : LOADER_DIRECT;
    a.Tin = t_in; a.shift = pad_left; a.bias = bias; a.KW = k; a.N = a.N0 = cout; a.Tout = t_out;
    a.flags = lrelu ? F_LRELU : 0;
    a.dst0 = y; a.obs0 = (long long)cout * t_out; a.opitch0 = t_out;
    // the kernel reads bf16 rows: convert x (fp32 output, obf = 0, keeps the comparison with float64 sharp)
    a.src0 = op_to_bf16(0, x, (long long)batch * cin, t_in, t_in, s);
    if (!a.src0) return fail(WUN_ERR_NOMEM, "bf16 temporary");
    a.pitch0 = pad8(t_in); a.bs0 = (long long)cin * a.pitch0; a.xbf = 1; a.obf = 0;
    if (!conv_bf16_supported(a)) return fail(WUN_ERR_UNSUPPORTED, "shape not served by the bf16 kernel (cin < 8 or k > 15)");
    float* img = (float*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
    PackDesc d;
    d.src_off = 0; d.src_in_ws = 0; d.dst_off = 0; d.KW = k; d.C = cin; d.N = cout;
    d.C8p = bf16_image_groups(cin); d.Npad = (cout + 63) / 64 * 64;
    PackDesc* dd = nullptr;
    HIP_TRY(hipMalloc((void**)&dd, sizeof(PackDesc)));
    hipError_t e = hipMemcpyAsync(dd, &d, sizeof(d), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = launch_pack_bf16(w, img, dd, 1, (long long)k * d.C8p * d.Npad, s);
    a.W = img; a.wb_c8p = d.C8p; a.wb_npad = d.Npad;
    if (e == hipSuccess) e = launch_conv_bf16(a, s);
    (void)hipStreamSynchronize(s);
    (void)hipFree(dd);
    HIP_TRY(e);
    return WUN_OK;
}

// Input gradient of the bf16 speed mode as a single operator (wun_op_conv1d_dgrad semantics): stride 1 = the
// bf16 conv on tap-flipped / transposed weights, stride 2 = the fused two-phase transposed conv (a lane owns 8
// consecutive outputs).  scratch: >= wun_op_conv1d_dgrad_bf16_scratch floats.  Synchronises the stream.
extern "C" int64_t wun_op_conv1d_dgrad_bf16_scratch(int cin, int cout, int k) {
    const int64_t wt = 2ll * (k + 1) * cin * cout + 64;                                       // transposed fp32 copy
    const int64_t img = (int64_t)(k + 1) * bf16_image_groups(cout) * ((2 * cin + 32 + 63) / 64 * 64) * 4 + 64;
    return wt + img + 128;
}

extern "C" int wun_op_conv1d_dgrad_bf16(const float* dz, const float* w, float* dx, float* scratch, int batch, int cin,
                                        int cout, int k, int t_in, int t_out, int stride, int pad_left, void* stream) {
    if (!dz || !w || !dx || !scratch) return fail(WUN_ERR_INVALID, "null argument");
    if (stride != 1 && stride != 2) return fail(WUN_ERR_UNSUPPORTED, "stride must be 1 or 2");
    if (stride == 2 && (pad_left != 0 || (cin & 3) != 0)) return fail(WUN_ERR_UNSUPPORTED, "stride-2: pad_left 0 and cin % 4 == 0 only");
    hipStream_t s = (hipStream_t)stream;
    float* wt = (float*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
    float* img = (float*)(((uintptr_t)(wt + 2ll * (k + 1) * cin * cout) + 255) & ~(uintptr_t)255);
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.B = batch; a.ostride = 1;
    op_src(a, dz, cout, t_out);
    a.Tin = t_out; a.N = a.N0 = cin;
    a.dst0 = dx; a.obs0 = (long long)cin * t_in; a.opitch0 = t_in;
    WtDesc d; d.src_off = 0; d.dst_off = 0; d.C = cin; d.N = cout;
    PackDesc pd; pd.src_off = 0; pd.src_in_ws = 0; pd.dst_off = 0; pd.C = cout;
    if (stride == 1) {
        d.J = k; d.k_last = k - 1; d.k_step = 1; d.mode = 0;
        a.shift = k - 1 - pad_left; a.KW = k; a.Tout = t_in;
        pd.KW = k; pd.N = cin; pd.Npad = (cin + 63) / 64 * 64;
    } else {
        const int J0 = (k + 1) / 2;
        d.J = J0; d.k_last = 2 * (J0 - 1); d.k_step = k; d.mode = 1;
        a.KW = J0; a.kw_full = k; a.shift = J0 - 1; a.Tout = (t_in + 1) / 2; a.Tlim = t_in; a.flags = F_PHASE2;
        pd.KW = J0; pd.N = 2 * cin; pd.Npad = (2 * cin + 32 + 63) / 64 * 64;
    }
    pd.C8p = bf16_image_groups(cout);
    a.src0 = op_to_bf16(0, dz, (long long)batch * cout, t_out, t_out, s);
    if (!a.src0) return fail(WUN_ERR_NOMEM, "bf16 temporary");
    a.pitch0 = pad8(t_out); a.bs0 = (long long)cout * a.pitch0; a.xbf = 1; a.obf = 0;
    if (!conv_bf16_supported(a)) return fail(WUN_ERR_UNSUPPORTED, "shape not served by the bf16 kernel");
    HIP_TRY(launch_make_wt_one(w, wt, d, s));
    PackDesc* dd = nullptr;
    HIP_TRY(hipMalloc((void**)&dd, sizeof(PackDesc)));
    hipError_t e = hipMemcpyAsync(dd, &pd, sizeof(pd), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = launch_pack_bf16(wt, img, dd, 1, (long long)pd.KW * pd.C8p * pd.Npad, s);
    a.W = img; a.wb_c8p = pd.C8p; a.wb_npad = pd.Npad;
    if (e == hipSuccess) e = launch_conv_bf16(a, s);
    (void)hipStreamSynchronize(s);
    (void)hipFree(dd);
    HIP_TRY(e);
    return WUN_OK;
}

/* Lane layout probe of v_mfma_f32_16x16x32_bf16: d[16][16] = bf16(a[16][32]) * bf16(b[32][16]). */
extern "C" int wun_op_mfma_bf16_probe(const float* a, const float* b, float* d, void* stream) {
    if (!a || !b || !d) return fail(WUN_ERR_INVALID, "null argument");
    HIP_TRY(launch_mfma_bf16_probe(a, b, d, (hipStream_t)stream));
    return WUN_OK;
}

extern "C" int wun_op_mfma_probe(const float* a, const float* b, float* d, void* stream) {
    if (!a || !b || !d) return fail(WUN_ERR_INVALID, "null argument");
    HIP_TRY(launch_mfma_probe(a, b, d, (hipStream_t)stream));
    return WUN_OK;
}

extern "C" int wun_profile_begin(void) { g_profiling = true; prof_begin(); return WUN_OK; }

extern "C" int wun_profile_end(char* json_out, int64_t capacity) {
    const std::string js = prof_end();
    g_profiling = false;
    if (!json_out || capacity < (int64_t)js.size() + 1) return fail(WUN_ERR_INVALID, "profile buffer too small");
    memcpy(json_out, js.c_str(), js.size() + 1);
    return WUN_OK;
}

extern "C" int wun_abi_sizes(int64_t* sizes, int n) {
    const int64_t v[3] = {(int64_t)sizeof(wun_config), (int64_t)sizeof(wun_plan_info), (int64_t)sizeof(wun_tensor_info)};
    for (int i = 0; i < 3 && i < n && sizes != nullptr; ++i) sizes[i] = v[i];
    return 3;
}

extern "C" const char* wun_last_error(void) { return g_err.c_str(); }
extern "C" const char* wun_version(void) { return "wun 0.5 (gfx950, fp32 MFMA 16x16x4 + bf16 MFMA 16x16x32 speed mode)"; }
