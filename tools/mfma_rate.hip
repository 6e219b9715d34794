// Stand-alone calibration (GPU box): issue rate of v_mfma_f32_32x32x16_f16 as a function of the number of accumulators in rotation, in
// s_memtime ticks AND in s_memrealtime (100 MHz) ticks, one wave per SIMD (256-thread workgroup), one workgroup per CU or two.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_mfma_rate tools/mfma_rate.hip && tools/_mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void rate_kernel(unsigned long long* out, float* sink, int reps) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[NACC];
    half8 a[4], b[4];
    for (int i = 0; i < NACC; ++i)
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) {
            a[i][j] = (_Float16)(0.01f * ((lane * 7 + i * 3 + j) % 13 - 6));
            b[i][j] = (_Float16)(0.02f * ((lane * 5 + i * 11 + j) % 17 - 8));
        }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_sched_barrier(0);
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int k = 0; k < 32 / NACC; ++k)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(k + i) & 3], b[(k * 3 + i) & 3], acc[i], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int q = 0; q < 16; ++q) s += acc[i][q];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = t1 - t0;
        out[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

template <int NACC>
void run(int nblocks, const char* what) {
    unsigned long long* out;
    float* sink;
    hipMalloc(&out, nblocks * 16);
    hipMalloc(&sink, nblocks * 256 * 4);
    const int reps = 64;  // 64 x 32 = 2048 MFMAs per wave
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(rate_kernel<NACC>, dim3(nblocks), dim3(256), 0, 0, out, sink, reps);
    hipDeviceSynchronize();
    unsigned long long h[4];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    const double n = reps * 32.0;
    printf("%-28s %d accumulators in rotation: %.1f s_memtime ticks / MFMA, %.2f ns / MFMA (s_memrealtime), tick = %.3f ns\n", what, NACC, h[0] / n,
           h[1] * 10.0 / n, h[1] * 10.0 / h[0]);
    hipFree(out);
    hipFree(sink);
}

int main() {
    run<1>(256, "one workgroup per CU,");
    run<2>(256, "one workgroup per CU,");
    run<4>(256, "one workgroup per CU,");
    run<8>(256, "one workgroup per CU,");
    run<4>(512, "two workgroups per CU,");
    run<8>(512, "two workgroups per CU,");
    run<8>(1, "a single workgroup,");
    return 0;
}
