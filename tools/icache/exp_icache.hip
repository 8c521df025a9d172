// Micro-experiment: how much does a cold instruction cache cost a short kernel on gfx950?
// Kernels of N KB straight-line code (s_nop 0 = 4 bytes, 1 cycle), each wave executes it once.  Back-to-back launches on one
// stream; time per launch vs code size.  Build: hipcc --offload-arch=gfx950 -O2 exp_icache.hip -o exp_icache
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N16 asm volatile("s_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0");
#define N256 N16 N16 N16 N16 N16 N16 N16 N16 N16 N16 N16 N16 N16 N16 N16 N16      // 1 KB
#define K4 N256 N256 N256 N256
#define K16 K4 K4 K4 K4
template <int KB> __global__ void k(int* out) {
    if (KB >= 1) { N256 }
    if (KB >= 2) { N256 }
    if (KB >= 4) { N256 N256 }
    if (KB >= 8) { K4 }
    if (KB >= 16) { K4 K4 }
    if (KB >= 32) { K16 }
    if (KB >= 64) { K16 K16 }
    if (out && threadIdx.x == 9999) out[0] = 1;
}
template <int KB> float run(int blocks, int threads, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k<KB>, dim3(blocks), dim3(threads), 0, 0, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k<KB>, dim3(blocks), dim3(threads), 0, 0, nullptr);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}
int main() {
    const int reps = 500;
    for (int cfg = 0; cfg < 3; cfg++) {
        int blocks = cfg == 0 ? 256 : (cfg == 1 ? 1024 : 4096), threads = 256;
        printf("blocks %d x %d threads: us/launch for 0/1/4/8/16/32/64 KB of code:", blocks, threads);
        printf(" %.2f", run<0>(blocks, threads, reps));
        printf(" %.2f", run<1>(blocks, threads, reps));
        printf(" %.2f", run<4>(blocks, threads, reps));
        printf(" %.2f", run<8>(blocks, threads, reps));
        printf(" %.2f", run<16>(blocks, threads, reps));
        printf(" %.2f", run<32>(blocks, threads, reps));
        printf(" %.2f\n", run<64>(blocks, threads, reps));
    }
    return 0;
}
