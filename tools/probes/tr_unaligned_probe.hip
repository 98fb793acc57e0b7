// Does ds_read_b64_tr_b16 (gfx950) accept lane addresses that are only 2-byte aligned, and at what cost?
// (Round 5: a conv loader that DMAs bf16 NCW rows into LDS as they are and builds its A fragments -- 16 times x 8 channels --
//  with two transposing reads per fragment would need one element offset per TAP, i.e. arbitrary 2-byte alignment.)
// For element offset `off` 0..7: lane i of a 16-lane group reads row (i >> 2) at columns off + 4*(i & 3) .. +3 of a
// [64 rows][pitch] b16 image; expected: lane c receives column off + c of rows 4g .. 4g+3.  Also times 4096 dependent reads.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/tr_unaligned_probe.hip -o /tmp/tr_unaligned_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__global__ void k(short* out, long long* cyc, int pitch, int off) {
    __shared__ __attribute__((aligned(16))) short lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    const int e = (4 * g + (i >> 2)) * pitch + off + 4 * (i & 3);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + e));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
    // throughput: 512 x 8 independent reads at rotating rows
    s16x4 acc = {0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < 512; ++r) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const s16x4 w = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + e + ((r + u) & 7) * 16 * pitch / 16));
            acc += w;
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (l == 0) *cyc = t1 - t0;
    if (acc[0] == 12345 && acc[1] == 777) out[0] = acc[2];
}
int main() {
    short* d; long long* c;
    hipMalloc(&d, 64 * 4 * 2); hipMalloc(&c, 8);
    for (int pitch : {64, 80, 72}) {
        for (int off = 0; off < 8; ++off) {
            hipMemset(d, 0xff, 64 * 4 * 2);
            k<<<1, 64>>>(d, c, pitch, off);
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) { printf("pitch %d off %d: LAUNCH FAILED %s\n", pitch, off, hipGetErrorString(e)); return 1; }
            short h[256]; long long cy;
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 4; ++j) {
                    const int want = (4 * (l >> 4) + j) * pitch + off + (l & 15);
                    if (h[l * 4 + j] != (short)want) ++bad;
                }
            printf("pitch %3d off %d: %s (%3d mismatches)  %5.1f cycles per read\n", pitch, off, bad ? "WRONG" : "ok   ", bad, (double)cy / 4096.0);
            if (bad && off < 4) for (int l = 0; l < 64; l += 21) printf("    lane %2d got %d %d %d %d want %d..\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], (4 * (l >> 4)) * pitch + off + (l & 15));
        }
    }
    return 0;
}
