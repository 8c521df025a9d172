// Which lanes feed which output of v_mfma_f32_4x4x4_16B_f16?  A = (lane id, 0, 0, 0) against B = (1, 0, 0, 0) shows the lane whose A
// operand lands in D[i] of every lane; the roles swapped show the B side.  hipcc --offload-arch=gfx950 mfma4_layout.hip -o mfma4_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    half4_t idv = {(_Float16)(float)lane, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f}, one = {(_Float16)1.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    float4_t z = {0.f, 0.f, 0.f, 0.f};
    float4_t da = __builtin_amdgcn_mfma_f32_4x4x4f16(idv, one, z, 0, 0, 0);
    float4_t db = __builtin_amdgcn_mfma_f32_4x4x4f16(one, idv, z, 0, 0, 0);
    for (int i = 0; i < 4; i++) { out[lane * 8 + i] = da[i]; out[lane * 8 + 4 + i] = db[i]; }
}
int main() {
    float* d; hipMalloc(&d, 64 * 8 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[64 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) if (l < 10 || l > 59) printf("lane %2d: A-source lanes of D[0..3] = %g %g %g %g | B-source lanes = %g %g %g %g\n", l, h[l*8], h[l*8+1], h[l*8+2], h[l*8+3], h[l*8+4], h[l*8+5], h[l*8+6], h[l*8+7]);
    return 0;
}
