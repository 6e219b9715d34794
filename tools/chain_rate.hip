// Stand-alone calibration (GPU box): cycles per time step of the SRU cell-state chain  c <- u0 + (c - u0) / (1 + exp2(u1 + vf c))  and of the
// deferred reset-gate / highway / f16 split + store ("write-back") as the sweep kernel issues them, one wave per SIMD, with the kernel's
// register footprint (128 gate registers per lane).
//   hipcc --offload-arch=gfx950 -O3 -o tools/_b_chain_rate tools/chain_rate.hip && tools/_b_chain_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

// MODE 0: the chain alone.  1: chain, exp/rcp replaced by multiplies.  2: two independent chains interleaved.
// 3: write-back, arithmetic only.  4: write-back with the two 2-byte LDS stores at an immediate offset.  5: with the kernel's scalar clamped row
// address (s_add, s_min, s_mul per step).  6: same, incremental scalar address (s_add, s_min).  7: chain on waves 0-1 while waves 2-3 wait at a barrier.
template <int MODE, int G = 4, int VAR = 0>
__global__ __launch_bounds__(256, 2) void chain_kernel(unsigned long long* out, float* sink, int reps, int L, int dirsel) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dir = (wave & 1) ^ dirsel;
    float u[4][32];
    for (int g = 0; g < 4; ++g)
        for (int q = 0; q < 32; ++q) u[g][q] = 0.01f * (g + 1) * ((lane * (7 - g) + q) % 13 - 6);
    const float vf = 0.3f + 0.001f * lane, vr = 0.2f - 0.001f * lane;
    float c = 0.f, c2 = 0.1f;
    char* const Hb = reinterpret_cast<char*>(smem);
    int col = ((lane >> 5) * 66 * 136 + dir * 32 + (lane & 31)) * 2;
    asm volatile("" : "+v"(col));
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_sched_barrier(0);
    for (int rep = 0; rep < reps; ++rep) {
        if (MODE <= 2 || MODE == 7) {
            if (MODE != 7 || wave < 2) {
#pragma unroll
                for (int q = 0; q < 32; ++q) {
                    if (MODE == 1) {
                        float z = fmaf(vf, c, u[1][q]);
                        z = z * 1.0001f;
                        z = 1.0f + z;
                        z = z * 0.9999f;
                        c = fmaf(c - u[0][q], z, u[0][q]);
                    } else {
                        const float f = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(vf, c, u[1][q])));
                        c = fmaf(c - u[0][q], f, u[0][q]);
                    }
                    if (MODE == 2) {
                        const float f2 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(vr, c2, u[2][q])));
                        c2 = fmaf(c2 - u[3][q], f2, u[3][q]);
                    }
                }
            }
            if (MODE == 7) __syncthreads();
        } else {
            const float vrr = vr + rep * 1e-7f;  // (nothing of a step is invariant across repetitions)
            int r0w = dir ? L - 1 : 0;
            asm volatile("" : "+s"(r0w));
            int srow = r0w * 272;
            const int sstep = dir ? -272 : 272;
#pragma unroll
            for (int q4 = 0; q4 < 32; q4 += G) {
                // VAR 0: everything.  1: no f16 split (the f32 value's low half is stored).  2: exp / rcp replaced by multiplies.  3: only the gate (fma, exp, add, rcp).
                float z[G], d[G], hv[G];
                _Float16 hh[G];
                unsigned lo[G];
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    const int q = q4 + i;
                    z[i] = fmaf(vrr, q == 0 ? c : u[0][q - 1], u[2][q]);
                    d[i] = fmaf(-vrr, u[3][q], u[0][q]);
                }
#pragma unroll
                for (int i = 0; i < G; ++i) z[i] = VAR == 2 ? z[i] * 1.0001f : __builtin_amdgcn_exp2f(z[i]);
#pragma unroll
                for (int i = 0; i < G; ++i) z[i] = 1.0f + z[i];
#pragma unroll
                for (int i = 0; i < G; ++i) z[i] = VAR == 2 ? z[i] * 0.9999f : __builtin_amdgcn_rcpf(z[i]);
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    hv[i] = VAR == 3 ? z[i] : fmaf(d[i], z[i], u[3][q4 + i]);
                    hh[i] = (_Float16)hv[i];
                }
                if (VAR == 0 || VAR == 2) {
#pragma unroll
                    for (int i = 0; i < G; ++i) asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lo[i]) : "v"(hv[i]), "v"(hh[i]));
                } else {
#pragma unroll
                    for (int i = 0; i < G; ++i) lo[i] = __float_as_uint(hv[i]);
                }
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    const int k = q4 + i;
                    if (MODE == 3) {
                        c2 += VAR == 3 ? hv[i] : (float)hh[i] + __uint_as_float(lo[i] << 16);
                    } else {
                        int o;
                        if (MODE == 4)
                            o = col + k * 272;
                        else if (MODE == 5)
                            o = col + (int)min((unsigned)(dir ? r0w - k : r0w + k), (unsigned)L) * 272;
                        else {
                            o = col + (int)min((unsigned)srow, (unsigned)(L * 272));
                            srow += sstep;
                        }
                        *reinterpret_cast<_Float16*>(Hb + o) = hh[i];
                        *reinterpret_cast<unsigned short*>(Hb + o + 128) = (unsigned short)lo[i];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            c += c2 * 1e-9f;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = c + c2;
    for (int g = 0; g < 4; ++g)
        for (int q = 0; q < 32; ++q) s += u[g][q];
    sink[blockIdx.x * 256 + threadIdx.x] = s + reinterpret_cast<_Float16*>(smem)[threadIdx.x];
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = t1 - t0;
        out[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

template <int MODE, int G = 4, int VAR = 0>
void run(int nblocks, const char* what) {
    unsigned long long* out;
    float* sink;
    (void)hipMalloc(&out, nblocks * 16);
    (void)hipMalloc(&sink, nblocks * 256 * 4);
    const int reps = 64;  // 2048 steps
    const size_t lds = 2 * 66 * 272 + 32768;
    (void)hipFuncSetAttribute((const void*)(chain_kernel<MODE, G, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((chain_kernel<MODE, G, VAR>), dim3(nblocks), dim3(256), lds, 0, out, sink, reps, 57, 0);
    (void)hipDeviceSynchronize();
    unsigned long long h[2];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    const double n = reps * 32.0;
    printf("%-64s %3d blocks: %.1f cycles / step, %.2f ns / step, clock %.2f GHz\n", what, nblocks, h[0] / n, h[1] * 10.0 / n, h[0] / (h[1] * 10.0));
    (void)hipFree(out);
    (void)hipFree(sink);
}

int main() {
    run<0>(1, "chain alone");
    run<0>(256, "chain alone");
    run<0>(512, "chain alone (two waves per SIMD)");
    run<1>(256, "chain, exp/rcp replaced by multiplies");
    run<2>(256, "two independent chains interleaved");
    run<7>(256, "chain on waves 0-1, waves 2-3 at the barrier");
    run<3>(256, "write-back, arithmetic only, groups of 4");
    run<3, 8>(256, "write-back, arithmetic only, groups of 8");
    run<3, 8, 1>(256, "  ... without the f16 split");
    run<3, 8, 2>(256, "  ... exp / rcp replaced by multiplies");
    run<3, 8, 3>(256, "  ... the gate only (fma, exp, add, rcp)");
    run<4, 4>(256, "write-back + 2 LDS stores, immediate offsets, groups of 4");
    run<4, 8>(256, "write-back + 2 LDS stores, immediate offsets, groups of 8");
    run<4, 8>(512, "  ... two waves per SIMD");
    run<4, 8, 1>(256, "  ... without the f16 split");
    run<5, 4>(256, "write-back + stores, scalar clamped row (add, min, mul)");
    return 0;
}
