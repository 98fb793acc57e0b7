// Probe of ds_read_b64_tr_b16 (gfx950): which LDS elements land in which lane.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/tr_probe.hip -o /tmp/tr_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int pitch) {
    __shared__ __attribute__((aligned(16))) short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    // lane i of a 16-lane group reads 4 contiguous elements: row (i >> 2) of the group's 4-row block, columns 4*(i & 3)..
    const int e = (4 * g + (i >> 2)) * pitch + 4 * (i & 3);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + e));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    for (int pitch : {16, 32, 48}) {
        k<<<1, 64>>>(d, pitch);
        short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int want = (4 * (l >> 4) + j) * pitch + (l & 15);     // row 4g+j, column l&15
                if (h[l * 4 + j] != want) ++bad;
            }
        printf("pitch %d: %s (%d mismatches)\n", pitch, bad ? "DIFFERENT" : "lane c gets column c, element j = row 4g+j", bad);
        if (bad) for (int l = 0; l < 64; l += 5) printf("  lane %d: %d %d %d %d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
