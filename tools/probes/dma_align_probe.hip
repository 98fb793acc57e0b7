#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;
__global__ void k(const float* src, float* out, int shift) {
    __shared__ __attribute__((aligned(16))) float lds[1024];
    const int lane = threadIdx.x;
    __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(src + shift + 4 * lane), (lds_void_t*)(lds), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[4 * lane + i] = lds[4 * lane + i];
}
int main() {
    float *src, *out; hipMalloc(&src, 4096 * 4); hipMalloc(&out, 1024 * 4);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        hipMemset(out, 0, 1024 * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out, shift);
        float o[256]; hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < 256; ++i) if (o[i] != (float)(i + shift)) ++bad;
        printf("shift %d: %d mismatches (first values %g %g %g %g %g)\n", shift, bad, o[0], o[1], o[2], o[3], o[4]);
    }
    return 0;
}
