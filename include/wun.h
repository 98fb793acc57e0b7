/*
 * wun.h -- C ABI of the MI355X-native Wave-U-Net hot path (libwun.so).
 *
 * Drop-in boundary for the reference's separator "plugin" surface
 *   UnetAudioSeparator(model_config)                 /root/reference/Models/UnetAudioSeparator.py:15-32
 *   .get_padding(shape)                              /root/reference/Models/UnetAudioSeparator.py:34-83
 *   .get_output(input, training, ...)                /root/reference/Models/UnetAudioSeparator.py:85-144
 * and for one `sess.run([separator_solver, ...])` of the training loop
 *   loss + tf.gradients + AdamOptimizer.minimize     /root/reference/Training.py:50-63,70-77,103-109
 *
 * Plain C types only.  All device buffers are owned by the CALLER (torch / hipMalloc);
 * the library owns only an opaque, immutable plan (shape tables + a small device-side
 * descriptor table).  Every call is asynchronous with respect to the hipStream_t passed
 * (as void*), re-entrant across plans, and returns 0 or a negative wun_status; the
 * message is available from wun_last_error() (thread-local).  Nothing aborts.
 *
 * Threading: one host thread drives a plan at a time (the reference is a single-threaded
 * sess.run loop); different plans may be driven from different threads.  A plan also owns
 * its side HIP streams / events, its split-reduction bookkeeping and -- after wun_plan_tune --
 * the tuned launch table, so the plan is immutable in its SHAPES, not in that scheduling
 * state.  wun_profile_begin/end and the wun_op_* entry points (single-operator tests and
 * benchmarks) use process-global state and are not meant for concurrent use.
 *
 * Streams: the side streams (weight gradients, deferred skip-window convs: work that fills the gaps of the
 * dependent chain on the caller's stream) are created at NORMAL queue priority, and at the LOWEST priority only
 * when wun_config.exclusive_streams = 1 (nothing else shares the device; see the field's comment for the hazard).
 * The plan's internal cross-stream events carry no system-scope fence: they order kernels of this device only.
 * Everything a call enqueues is complete with respect to the caller's stream when the call's work on that stream
 * is (the last internal operation of every call is the caller's stream waiting for the side streams); host code
 * synchronises through that stream, never through the plan's events.
 *
 * Concurrency caveat (measured on MI355X, tools/probes/pk_fma_probe.hip): a packed fp32 FMA whose low lane
 * reads the high half of a register pair (v_pk_fma_f32 ... op_sel:[0,1,0], which the compiler emits freely) returns wrong
 * results while ANOTHER kernel's v_mfma_f32_16x16x32_bf16 waves share the CU.  Every kernel of the bf16 mode, and every kernel
 * both modes share, is built without packed fp32 instructions; the exact-fp32 MFMA kernels keep them (0.5 % per step) because
 * inside one plan they only ever run beside fp32 MFMA kernels.  Do not RUN an exact-fp32 plan concurrently (other stream /
 * thread, same device) with a bf16-mode plan or with other bf16-MFMA work (e.g. a bf16 GEMM); back to back is fine.
 * `make -C wave-u-net_amd/csrc nopkall` builds a library without any packed fp32 instruction for callers who must.
 *
 * Tensor layouts at the boundary are the reference's: audio is float32 [B, T, C]
 * (channel-last, exactly what get_output receives/returns); kernels are TF layout
 * [K, Cin, Cout]; variables sit in a flat float32 arena in TF creation order
 * (conv1d, conv1d_1, ... and interp_<i>), each tensor at the offset the plan reports.
 */
#ifndef WUN_H
#define WUN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum wun_status {
    WUN_OK = 0,
    WUN_ERR_INVALID = -1,      /* bad argument / impossible shape (reference: assert)        */
    WUN_ERR_UNSUPPORTED = -2,  /* reference: NotImplementedError                             */
    WUN_ERR_HIP = -3,          /* a HIP runtime call failed                                   */
    WUN_ERR_NOMEM = -4
} wun_status;

/* The model_config keys UnetAudioSeparator.__init__ reads (UnetAudioSeparator.py:20-32). */
typedef struct wun_config {
    int32_t num_layers;
    int32_t num_initial_filters;
    int32_t filter_size;        /* 1..15 (the register-staged loaders hold at most 15 taps;       */
    int32_t merge_filter_size;  /*  larger sizes return WUN_ERR_UNSUPPORTED at plan creation)     */
    int32_t input_filter_size;  /* used by get_padding only (UnetAudioSeparator.py:73); the graph */
                                /* convolves layer 0 with filter_size (UnetAudioSeparator.py:98)   */
    int32_t output_filter_size;
    int32_t upsampling;         /* 0 = "linear", 1 = "learned"                                */
    int32_t output_type;        /* 0 = "direct", 1 = "difference"                             */
    int32_t context;            /* 0 = same padding, 1 = valid convolutions with context      */
    int32_t num_sources;        /* len(source_names)                                          */
    int32_t num_channels;       /* 1 if mono_downmix else 2                                   */
    int32_t output_activation;  /* 0 = "tanh", 1 = "linear"                                   */
    int32_t compute_dtype;      /* 0 = exact fp32 (v_mfma_f32_16x16x4_f32; the reference's arithmetic),                   */
                                /* 1 = bf16 mode: every activation and activation-gradient tensor lives in HBM as bf16   */
                                /*     (rounded once, by the epilogue that writes it), convs / input gradients / weight  */
                                /*     gradients on v_mfma_f32_16x16x32_bf16 with fp32 accumulate; parameters, weight    */
                                /*     gradients, Adam state, the audio and the head's d(pre-activation) stay fp32.      */
                                /*     Needs num_initial_filters % 8 == 0 (else the plan is the exact-fp32 plan:        */
                                /*     wun_plan_info.compute_dtype_effective says which one was built).                  */
                                /*     Not recommended with upsampling = 1 (learned): the gradient of an interpolation    */
                                /*     weight is a sum of DIFFERENCES of adjacent activations that are already rounded   */
                                /*     to bf16 (cancellation amplifies the storage rounding: up to 0.5 of max|g| on the  */
                                /*     312-element interp_0 of M5 against the un-rounded oracle, tests/test_gpu_bf16.py). */
    int32_t exclusive_streams;  /* scheduling hint, no effect on results.  1 = nothing else runs on this device  */
                                /* beside the plan's calls: its side streams get the LOWEST queue priority (they  */
                                /* fill the gaps of the dependent chain on the caller's stream, ~1 % per step).   */
                                /* 0 (default) = normal priority -- REQUIRED when collectives (RCCL) or other     */
                                /* streams share the device: low-priority queues beside a communication stream    */
                                /* were measured 40 % slower (and a process that ever created them stays slow).   */
} wun_config;

typedef struct wun_plan wun_plan;

typedef struct wun_plan_info {
    int64_t batch;
    int64_t input_frames;       /* Tin                                                       */
    int64_t output_frames;      /* Tout                                                      */
    int64_t num_params;         /* trainable scalars (sum of tensor sizes, no padding)       */
    int64_t arena_floats;       /* size of the param / grad / m / v arenas (with padding)    */
    int64_t workspace_floats;   /* activation + gradient + scratch workspace                 */
    int64_t num_tensors;        /* number of TF variables                                    */
    int64_t num_outputs;        /* == num_sources                                            */
    double  fwd_flops;          /* algorithmic conv FLOPs / step as executed (dead work skipped) */
    double  bwd_flops;
    double  fwd_flops_dense;    /* the reference graph's FLOPs (no dead-work skipping)       */
    double  fwd_flops_unique;   /* every OBSERVED conv output computed once.  == fwd_flops for same-padding plans and for  */
    double  bwd_flops_unique;   /* context plans of the exact-fp32 mode (round 6); the bf16 mode's context plans still run */
                                /* a full-rate conv over each skip window, i.e. compute its even positions twice           */
    int64_t compute_dtype_effective; /* the arithmetic the plan really runs: 0 = exact fp32, 1 = bf16 mode.  A config that   */
                                /* asks for compute_dtype = 1 but does not qualify (num_initial_filters % 8 != 0, a      */
                                /* tap-less conv phase, rows beyond the bf16 kernels' 32-bit offsets) gets the exact-fp32 */
                                /* plan and reports 0 here -- callers label their results with THIS value.                */
} wun_plan_info;

typedef struct wun_tensor_info {
    char    name[64];           /* TF variable name, e.g. "separator/conv1d_3/kernel"        */
    int64_t offset;             /* float offset into the arenas                              */
    int32_t ndim;
    int64_t shape[4];
} wun_tensor_info;

/* UnetAudioSeparator.get_padding (UnetAudioSeparator.py:34-83): smallest valid
 * (input_frames, output_frames) whose output covers desired_frames.  Pure host integer work. */
int wun_get_padding(const wun_config* cfg, int64_t desired_frames,
                    int64_t* input_frames, int64_t* output_frames);

/* Build the static plan for (config, batch, input_frames): the equivalent of the reference
 * building its TF graph once (Training.py:47).  Fails with WUN_ERR_INVALID where the
 * reference's asserts would (UnetAudioSeparator.py:55,121; Utils.py:117). */
int wun_plan_create(const wun_config* cfg, int64_t batch, int64_t input_frames, wun_plan** out);
void wun_plan_destroy(wun_plan* plan);
int wun_plan_query(const wun_plan* plan, wun_plan_info* info);
int wun_plan_tensor(const wun_plan* plan, int64_t index, wun_tensor_info* info);

/* Where the forward activations of the plan live in the caller's workspace after wun_forward(training = 1) -- what
 * the backward pass reads its LeakyReLU derivatives from (Utils.py:79-80).  Used by the parity tests to pin the float64
 * oracle's LeakyReLU branch decisions to the ones the kernels took (a pre-activation within fp32 rounding of 0 otherwise
 * decides a 1-vs-0.2 factor differently in the two precisions), and by tools/ws_diff.py.  Every tensor is NCW:
 * element (b, c, j) is element b*batch_stride + c*pitch + j of the array that starts `offset` floats into the workspace
 * (float32 or bfloat16 elements: elem_bytes) and holds the POST-activation output of conv
 * position t0 + j*tstep of its layer (UnetAudioSeparator.py:98-100,102,123):
 *   kind 0, index i : decimated stream of down level i  (t0 = 0, tstep = 2: the [:, ::2, :] of :100)
 *   kind 1, index i : skip window of down level i       (context: the centre crop Utils.crop takes, t0 = crop start;
 *                                                         same padding: the whole conv output)
 *   kind 2          : bottleneck conv output (:102)
 *   kind 3, index j : output of up conv j (:123)
 * and, for the layer-by-layer parity tests of the bf16 mode (each launch's output against a float64 computation from the
 * tensors that launch read; tests/test_gpu_bf16.py), the other tensors a training step leaves in the workspace -- valid
 * after wun_loss_backward, same addressing, no activation implied:
 *   kind 4, index j : the 2x-upsampled input of up conv j (:109-118); not written by plans whose split-K epilogue
 *                     fuses the interpolation (compute_dtype = 0)
 *   kind 5, index j : d loss / d pre-activation of up conv j
 *   kind 6, index j : d loss / d (tensor of kind 4); compute_dtype = 0: only where the adjoint is not fused
 *   kind 7, index i : d loss / d pre-activation of down conv i at the positions of kind 1
 *   kind 8, index i : d loss / d pre-activation of down conv i at the positions of kind 0 (context only; with same
 *                     padding kind 7 holds the whole row)
 *   kind 9          : d loss / d pre-activation of the bottleneck conv
 * All fields are int64.  WUN_ERR_INVALID for an unknown kind / index. */
typedef struct wun_activation_info {
    int64_t offset;                         /* of the tensor, in FLOATS from the workspace base */
    int64_t batch_stride, pitch;            /* in ELEMENTS of the tensor */
    int64_t channels, frames;               /* frames = valid positions j per row */
    int64_t t0, tstep;
    int64_t elem_bytes;                     /* 4 = float32; 2 = bfloat16 (compute_dtype = 1: activations live in HBM as bf16) */
} wun_activation_info;
int wun_plan_activation(const wun_plan* plan, int32_t kind, int32_t index, wun_activation_info* info);

/* get_output (UnetAudioSeparator.py:85-144).
 *   params   : device, arena_floats
 *   mix_btc  : device, [B, Tin, C]
 *   workspace: device, workspace_floats (intermediates are kept for wun_loss_backward)
 *   outputs  : device, [S, B, Tout, C] -- source s in source_names order
 *   training : 0 => AudioClip active (Utils.py:82-92) */
int wun_forward(const wun_plan* plan, const float* params, const float* mix_btc,
                float* workspace, float* outputs, int training, void* stream);

/* Loss (Training.py:50-63) + full backward of the graph (the tf.gradients implied by
 * Training.py:77).  Must follow wun_forward(training=1) on the same workspace/outputs.
 *   targets  : device, [S, B, Tout, C]
 *   grads    : device, arena_floats (overwritten)
 *   loss     : device, 1 float */
int wun_loss_backward(const wun_plan* plan, const float* params, const float* mix_btc,
                      float* workspace, const float* outputs, const float* targets,
                      float* grads, float* loss, void* stream);

/* Same as wun_loss_backward, plus data-parallel overlap hooks: bucket k covers arena floats
 * [bucket_starts[k], end) where `end` is the previous bucket's start (buckets are given from the
 * END of the arena towards 0, i.e. in backward completion order).  bucket_events[k] (hipEvent_t,
 * created by the caller) is recorded -- on an internal stream -- as soon as every gradient at an
 * offset >= bucket_starts[k] is final, so the caller can start that bucket's all-reduce on a
 * communication stream (hipStreamWaitEvent) while the rest of the backward pass still runs.
 *
 * A C caller (no torch.distributed) pairs the events with its own RCCL calls -- the library itself links no collective
 * library -- exactly as wave-u-net_amd/parallel.py does:
 *     hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) for every bucket;  comm = a non-blocking stream, highest priority
 *     wun_loss_backward_ex(..., stream, starts, (void* const*)ev, n);
 *     for k in 0..n-1:  hipStreamWaitEvent(comm, ev[k], 0);
 *                       ncclAllReduce(grads + starts[k], grads + starts[k], end[k] - starts[k], ncclFloat, ncclSum, nccl_comm, comm);
 *     hipEventRecord(done, comm); hipStreamWaitEvent(stream, done, 0);
 *     wun_adam_step(..., grad_scale = 1.0f / world_size, stream);
 * with wun_config.exclusive_streams = 0 (low-priority side streams beside a communication stream cost 40 %). */
int wun_loss_backward_ex(const wun_plan* plan, const float* params, const float* mix_btc,
                         float* workspace, const float* outputs, const float* targets,
                         float* grads, float* loss, void* stream,
                         const int64_t* bucket_starts, void* const* bucket_events, int32_t nbuckets);

/* Optional autotuning pass (no reference counterpart): runs one forward + loss/backward on the
 * given buffers while timing, for every conv / weight-gradient launch of the step, the candidate
 * tile shapes and split factors, and caches the fastest per launch in the plan.  The contents of
 * outputs / loss / grads / workspace after this call are NOT meaningful (accumulating launches are
 * replayed while timing): run the real step afterwards.  Parameters and optimizer state are not
 * touched.  Synchronises the stream. */
int wun_plan_tune(const wun_plan* plan, const float* params, const float* mix_btc, float* workspace,
                  float* outputs, const float* targets, float* grads, float* loss, void* stream);

/* Tuned choices as text (one line per launch position), so a later process can reuse them without
 * re-tuning: export writes a NUL-terminated string into buf (WUN_ERR_INVALID if cap is too small or
 * the plan is untuned); import accepts that string for a plan of the same config / batch / length
 * written by the same library build (WUN_ERR_INVALID otherwise: the header line carries the config,
 * the launch-order version and the entry counts, and the table ends with an "end" line, so stale or
 * truncated tables are refused) and switches the plan to the tuned choices.  An entry that is not a
 * legal choice for the launch at its position is ignored at launch time (heuristic choice instead);
 * a split factor can never exceed the split-K scratch. */
int wun_plan_tune_export(const wun_plan* plan, char* buf, int64_t cap);
int wun_plan_tune_import(const wun_plan* plan, const char* text);

/* tf.train.AdamOptimizer update (Training.py:77), TF rule:
 *   lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v; theta -= lr_t*m/(sqrt(v)+eps); g := grad_scale*grad
 * step is 1-based.  grad_scale = 1/world_size after a sum all-reduce. */
int wun_adam_step(const wun_plan* plan, float* params, const float* grads, float* m, float* v,
                  int64_t step, float lr, float beta1, float beta2, float eps, float grad_scale,
                  void* stream);

/* ---- single operators (used by the parity tests and for per-kernel profiling) ---------- */

/* y[b][co][q] = act(bias[co] + sum_{k,ci} w[k][ci][co] * x[b][ci][q*stride + k - pad_left]),
 * x zero outside [0, t_in).  NCW float32, w in TF layout [K, Cin, Cout].
 * stride in {1, 2}; t_out is given by the caller.  lrelu: 0/1 (alpha = 0.2). */
int wun_op_conv1d(const float* x, const float* w, const float* bias, float* y,
                  int batch, int cin, int cout, int k, int t_in, int t_out,
                  int stride, int pad_left, int lrelu, void* stream);

/* dw[k][ci][co] = sum_{b,q} x[b][ci][q*stride + k - pad_left] * dz[b][co][q]; db[co] = sum dz.
 * scratch: device floats, at least wun_op_conv1d_wgrad_scratch(...) of them. */
int64_t wun_op_conv1d_wgrad_scratch(int batch, int cin, int cout, int k, int t_out);
int wun_op_conv1d_wgrad(const float* x, const float* dz, float* dw, float* db, float* scratch,
                        int batch, int cin, int cout, int k, int t_in, int t_out,
                        int stride, int pad_left, void* stream);

/* dx[b][ci][t] = sum_{k,co} w[k][ci][co] * dz[b][co][(t + pad_left - k)/stride] (when divisible
 * and in range).  wt_scratch: device floats, >= 2*k*cin*cout. */
int wun_op_conv1d_dgrad(const float* dz, const float* w, float* dx, float* wt_scratch,
                        int batch, int cin, int cout, int k, int t_in, int t_out,
                        int stride, int pad_left, void* stream);

/* The plan's general conv launch as a single operator: the input is the virtual channel-concat of
 * x0 [B][c0][t_in] and (optionally) x1 [B][c1][t_in] (Utils.crop_and_concat, Utils.py:11-24, without
 * the copy); output element q of channel n is stored at y[b][n][ooff + q*ostride] (y is [B][cout][t_y]);
 * mask (same geometry as y, or NULL) multiplies by the LeakyReLU derivative of the value stored
 * there (1 if > 0 else 0.2); accumulate adds to what y already holds.  Semantics otherwise as
 * wun_op_conv1d. */
int wun_op_conv1d_ex(const float* x0, int c0, const float* x1, int c1, const float* w, const float* bias,
                     float* y, const float* mask, int batch, int cout, int k, int t_in, int t_out,
                     int t_y, int stride, int pad_left, int lrelu, int accumulate, int ostride, int ooff,
                     void* stream);

/* Test hook for the secondary outputs of the plan's conv launch (round 6: a context plan computes every conv output once --
 * the decimated stream is a slice of the encoder output, UnetAudioSeparator.py:98-100).  Every following wun_op_conv1d_ex
 * launch also writes
 *   copy0 [B][cout][t0]: expand = 0: the compact copy of its EVEN outputs, copy0[b][n][q / 2] (same-padding down levels);
 *                        expand = 1: output q at copy0[b][n][2q - exp_lo] where 0 <= 2q - exp_lo < exp_len (the stride-2
 *                        launch of a down level writing the even positions of the skip window);
 *   copy1 [B][cout][t1]: the compact copy of its ODD outputs, copy1[b][n][q / 2] (an up level's input gradient splitting
 *                        the skip window's gradient by parity);
 * and, when acc_len > 0, `accumulate` applies only to the row positions ooff + q*ostride inside [acc_lo, acc_lo + acc_len)
 * (stored elsewhere).  NULL pointers / acc_len = 0 switch each part off; wun_op_set_conv_copies(0,0,0,0,0,0,0,0,0) resets. */
int wun_op_set_conv_copies(float* copy0, int t0, int expand, int exp_lo, int exp_len, float* copy1, int t1,
                           int acc_lo, int acc_len);

/* Test hook: force the tile variant (index into the kernel's variant table, -1 = automatic) and
 * split-K factor (0 = automatic) of every following wun_op_conv1d / _ex / _dgrad launch, so the parity
 * tests can reach every tiling.  A choice the dispatcher would never make for that launch (tile
 * mostly padding, split past the scratch buffer, loader not supported by the tile) makes the
 * launch fail with WUN_ERR_HIP -- it is rejected, not miscomputed. */
int wun_op_force_conv_variant(int variant, int ksplit);
int wun_op_num_conv_variants(void);

/* Test hook for wun_op_conv1d_wgrad: force the weight-gradient tile geometry (mtw in {1,2,4,6} row
 * tiles per wave, nw in 1..5 column tiles; 0,0 = automatic) and the number of reduction splits
 * (0 = automatic).  Call wun_op_conv1d_wgrad_scratch AFTER forcing: the scratch size depends on it.
 * A geometry the kernel's staging cannot hold for the shape fails with WUN_ERR_UNSUPPORTED. */
int wun_op_force_wgrad_variant(int mtw, int nw, int nsplit);

/* Test hook: run the following wun_op_conv1d_wgrad calls on the bf16 speed-mode kernel (operands rounded
 * to bf16, v_mfma_f32_16x16x32_bf16, fp32 accumulate; same tiles / splits / reduction). */
int wun_op_set_wgrad_bf16(int on);

/* Test hook: run the following (exact-fp32) wun_op_conv1d_wgrad calls on the register-window form of the weight-gradient
 * kernel (wgrad_win_kernel: aligned 16-byte operand reads, DMA staging issued from inside the MFMA stream, split partials
 * in the final [K][Cin][Cout] layout) -- the form the plan uses for every layer it serves.  wun_op_force_wgrad_variant then
 * means (1, column tiles per wave 1..5|6, splits; a negative split count = target grid size).  Shapes the kernel does not
 * serve (taps other than 15 / 5; 15 taps with Cin not a multiple of 8) fail with WUN_ERR_UNSUPPORTED.  Ignored while the
 * bf16 hook is on. */
int wun_op_set_wgrad_win(int on);

/* Test hook: run the following wun_op_conv1d_wgrad calls on the direct-reduction ("narrow") kernels the plan uses for the
 * layers without a dense channel x channel face -- the 1-/2-channel audio-input conv and the output head (wun_narrow.hip:
 * the LDS-staged form for Cin * Cout <= 256, the streaming form for one input channel and <= 24 output channels).  Other
 * shapes fail with WUN_ERR_UNSUPPORTED. */
int wun_op_set_wgrad_narrow(int on);

/* The bf16 mode's conv as a single operator (wun_op_conv1d semantics, Cin >= 8, K <= 15): x and w
 * are rounded to bf16 (nearest-even; x into a temporary bf16 copy -- the kernel reads bf16 rows), products accumulate
 * in fp32, y is stored as fp32.  scratch: device floats, at least
 * wun_op_conv1d_bf16_scratch(cin, cout, k) (packed bf16 weight image).  Synchronises the stream. */
int64_t wun_op_conv1d_bf16_scratch(int cin, int cout, int k);
int wun_op_conv1d_bf16(const float* x, const float* w, const float* bias, float* y, float* scratch,
                       int batch, int cin, int cout, int k, int t_in, int t_out, int stride, int pad_left,
                       int lrelu, void* stream);

/* Input gradient in the bf16 mode (wun_op_conv1d_dgrad semantics; stride 2 = the fused two-phase transposed
 * conv, pad_left 0 and cin % 4 == 0).  scratch: device floats, >= wun_op_conv1d_dgrad_bf16_scratch(cin, cout, k).
 * Synchronises the stream. */
int64_t wun_op_conv1d_dgrad_bf16_scratch(int cin, int cout, int k);
int wun_op_conv1d_dgrad_bf16(const float* dz, const float* w, float* dx, float* scratch, int batch, int cin,
                             int cout, int k, int t_in, int t_out, int stride, int pad_left, void* stream);

/* Lane layout probe of v_mfma_f32_16x16x32_bf16: d[16][16] = bf16(a[16][32]) * bf16(b[32][16]) (row-major). */
int wun_op_mfma_bf16_probe(const float* a, const float* b, float* d, void* stream);

/* Lane layout probe of v_mfma_f32_16x16x4_f32: d[16][16] = a[16][4] * b[4][16] (row-major). */
int wun_op_mfma_probe(const float* a, const float* b, float* d, void* stream);

/* Per-kernel timing with HIP events recorded on the launch stream around every heavy launch
 * between begin and end; end() synchronises and writes a JSON summary
 * {"bracket_overhead_ms", "kernels":[{"name","launches","ms","flops"}]} (used by bench.py for the
 * roofline line).  While active, the library's internal side stream is not used, so launch
 * durations are not distorted by concurrent kernels.  "ms" is the raw sum of the event brackets;
 * "bracket_overhead_ms" is the median duration of an EMPTY bracket recorded after every launch
 * (two event packets cost ~5 us that are not kernel time): subtract it once per launch to compare
 * with a profiler's kernel durations. */
int wun_profile_begin(void);
int wun_profile_end(char* json_out, int64_t capacity);

const char* wun_last_error(void);
const char* wun_version(void);

/* sizeof() of the structs this header declares, as the LIBRARY was compiled: sizes[0] = wun_config,
 * sizes[1] = wun_plan_info, sizes[2] = wun_tensor_info; writes min(n, 3) entries and returns 3.  A binding written
 * in another language (the ctypes stub of INTEGRATION.md, cffi, cgo ...) checks its own struct sizes against these
 * before the first call instead of passing a short struct to a library that reads a longer one. */
int wun_abi_sizes(int64_t* sizes, int n);

#ifdef __cplusplus
}
#endif
#endif /* WUN_H */
