// Probe: which XCC (XCD) runs workgroup i of a dispatch?  build: hipcc -O3 --offload-arch=gfx950 tools/probes/xcc_probe.hip -o /tmp/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;   // HW_REG_XCC_ID[3:0]
}
__global__ void busy(float* x, int n) { float v = x[threadIdx.x]; for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f; x[threadIdx.x] = v; }
int main() {
    const int G = 2048;
    int* d; hipMalloc(&d, G * 4);
    float* f; hipMalloc(&f, 1024 * 4);
    hipStream_t s2; hipStreamCreate(&s2);
    for (int rep = 0; rep < 4; ++rep) {
        if (rep >= 2) busy<<<777, 256, 0, s2>>>(f, 200000);     // a concurrent dispatch on another queue
        k<<<G, 256>>>(d);
        int h[G]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 8; i < G; ++i) if (h[i] != h[i & 7]) ++bad;
        printf("rep %d first 16:", rep); for (int i = 0; i < 16; ++i) printf(" %d", h[i]); printf("  | residue mismatches: %d\n", bad);
        hipDeviceSynchronize();
    }
    return 0;
}
