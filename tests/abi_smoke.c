/* A C (non-ctypes) caller of the C ABI of include/wun.h: host-side entry points only, so it runs without a GPU.
 * Built and run by tests/test_abi_host.py:  gcc -std=c99 -Iinclude tests/abi_smoke.c -Lwave-u-net_amd -lwun ...
 * Reference surface exercised: UnetAudioSeparator.__init__ / get_padding (Models/UnetAudioSeparator.py:15-83) and
 * the variable table the graph construction creates (UnetAudioSeparator.py:92-142). */
#include <stdio.h>
#include <string.h>
#include "wun.h"

#define CHECK(cond, msg) do { if (!(cond)) { fprintf(stderr, "abi_smoke: FAILED %s (%s)\n", msg, wun_last_error()); return 1; } } while (0)

int main(void) {
    int64_t sizes[3] = {0, 0, 0};
    CHECK(wun_abi_sizes(sizes, 3) == 3, "wun_abi_sizes");
    CHECK(sizes[0] == (int64_t)sizeof(wun_config) && sizes[1] == (int64_t)sizeof(wun_plan_info) &&
          sizes[2] == (int64_t)sizeof(wun_tensor_info), "struct sizes of header and library agree");

    /* M1 with context: 12 levels, 24 filters, 15/5/15/1 taps, linear upsampling, direct output, mono, 2 sources, tanh */
    wun_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.num_layers = 12; cfg.num_initial_filters = 24; cfg.filter_size = 15; cfg.merge_filter_size = 5;
    cfg.input_filter_size = 15; cfg.output_filter_size = 1; cfg.context = 1; cfg.num_sources = 2; cfg.num_channels = 1;
    int64_t tin = 0, tout = 0;
    CHECK(wun_get_padding(&cfg, 16384, &tin, &tout) == WUN_OK, "wun_get_padding");
    CHECK(tin == 147443 && tout == 16389, "get_padding(16384) == (147443, 16389)");   /* SURVEY.md section 8(c) */

    wun_plan* plan = NULL;
    CHECK(wun_plan_create(&cfg, 16, tin, &plan) == WUN_OK && plan != NULL, "wun_plan_create");
    wun_plan_info info;
    CHECK(wun_plan_query(plan, &info) == WUN_OK, "wun_plan_query");
    CHECK(info.batch == 16 && info.input_frames == 147443 && info.output_frames == 16389, "plan shapes");
    CHECK(info.num_params == 10263028 && info.num_tensors == 54 && info.num_outputs == 2, "M1: 10 263 028 parameters in 54 variables");
    wun_tensor_info ti;
    CHECK(wun_plan_tensor(plan, 0, &ti) == WUN_OK, "wun_plan_tensor(0)");
    CHECK(strcmp(ti.name, "separator/conv1d/kernel") == 0 && ti.ndim == 3 && ti.shape[0] == 15 && ti.shape[1] == 1 &&
          ti.shape[2] == 24 && ti.offset == 0, "first variable = separator/conv1d/kernel [15,1,24]");
    CHECK(wun_plan_tensor(plan, info.num_tensors, &ti) == WUN_ERR_INVALID, "tensor index past the table is refused");
    /* where the activations live (wun_plan_activation): level 0 of M1 + context convolves 147443 samples to 147429, the
     * decimated stream keeps the 73715 even positions, the skip window is the 16393-sample centre crop the last up level takes */
    wun_activation_info ai;
    CHECK(wun_plan_activation(plan, 0, 0, &ai) == WUN_OK && ai.channels == 24 && ai.frames == 73715 && ai.t0 == 0 &&
          ai.tstep == 2 && ai.elem_bytes == 4 && ai.pitch >= ai.frames && ai.batch_stride == ai.channels * ai.pitch, "dec_0 geometry");
    CHECK(wun_plan_activation(plan, 1, 0, &ai) == WUN_OK && ai.channels == 24 && ai.frames == 16393 && ai.tstep == 1 &&
          ai.t0 == (147429 - 16393) / 2, "skip_0 = centre crop of the level-0 conv output (Utils.py:120-121)");
    CHECK(wun_plan_activation(plan, 2, 0, &ai) == WUN_OK && ai.channels == 312, "bottleneck: 312 channels");
    CHECK(wun_plan_activation(plan, 3, 11, &ai) == WUN_OK && ai.channels == 24 && ai.frames == 16389, "up conv 11 = the feature map");
    /* ... and the gradient tensors of a training step (kinds 4 - 9): d loss / d pre-activation lives where the activation's
     * positions are -- the skip window's gradient has the skip window's geometry, the decimated stream's likewise */
    CHECK(wun_plan_activation(plan, 7, 0, &ai) == WUN_OK && ai.channels == 24 && ai.frames == 16393 && ai.t0 == (147429 - 16393) / 2,
          "dz_skip_0 has the geometry of skip_0");
    CHECK(wun_plan_activation(plan, 8, 0, &ai) == WUN_OK && ai.frames == 73715 && ai.tstep == 2, "dz_dec_0 has the geometry of dec_0");
    CHECK(wun_plan_activation(plan, 4, 11, &ai) == WUN_OK && ai.channels == 48 && ai.frames == 16393, "upsampled input of up conv 11: 48 x 16393");
    CHECK(wun_plan_activation(plan, 5, 11, &ai) == WUN_OK && ai.channels == 24 && ai.frames == 16389 &&
          wun_plan_activation(plan, 6, 11, &ai) == WUN_OK && ai.channels == 48 && ai.frames == 16393 &&
          wun_plan_activation(plan, 9, 0, &ai) == WUN_OK && ai.channels == 312, "dz_up_11, d_ups_11, dz_bottleneck");
    CHECK(wun_plan_activation(plan, 3, 12, &ai) == WUN_ERR_INVALID && wun_plan_activation(plan, 10, 0, &ai) == WUN_ERR_INVALID &&
          wun_plan_activation(plan, 9, 1, &ai) == WUN_ERR_INVALID, "unknown activation kind / index is refused");
    wun_plan_destroy(plan);
    /* the bf16 mode keeps its activations in HBM as bfloat16 (compute_dtype = 1, layer widths in groups of 8) */
    cfg.compute_dtype = 1;
    plan = NULL;
    CHECK(wun_plan_create(&cfg, 16, tin, &plan) == WUN_OK && plan != NULL, "wun_plan_create (bf16 mode)");
    CHECK(wun_plan_activation(plan, 0, 0, &ai) == WUN_OK && ai.elem_bytes == 2 && (ai.pitch & 7) == 0, "bf16 mode: dec_0 holds bfloat16, rows of 16 bytes");
    wun_plan_destroy(plan);
    cfg.compute_dtype = 0;

    /* error convention: negative status + message, never an abort (reference: assert, UnetAudioSeparator.py:55) */
    cfg.filter_size = 31;                                   /* > 15 taps: WUN_ERR_UNSUPPORTED (include/wun.h) */
    plan = NULL;
    int rc = wun_plan_create(&cfg, 16, tin, &plan);
    CHECK(rc < 0 && plan == NULL, "bad config is refused");
    CHECK(strlen(wun_last_error()) > 0, "wun_last_error carries the reason");
    cfg.filter_size = 15;
    CHECK(wun_plan_create(&cfg, 16, 1000, &plan) < 0, "an input length the model cannot produce is refused");
    printf("abi_smoke: ok (wun_config %lld bytes, %s)\n", (long long)sizes[0], wun_version());
    return 0;
}
