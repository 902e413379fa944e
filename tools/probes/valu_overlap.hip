// Probe: do VALU instructions issue in the shadow of v_mfma_f32_32x32x16_f16 on gfx950?
// Per step: 1 MFMA (32 matrix-pipe cycles) + NV VALU instructions of a given kind, all independent of the MFMA chain.
// The MFMA-only step takes 20.4 ns (the chip runs ~1.6 GHz under this load); anything above is what the other instructions cost.  One wave per SIMD (256 threads) and two
// (512 threads), 256 workgroups.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_overlap.hip -o tools/probes/valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

// KIND 0: v_fma_f32   1: v_pk_fma_f32 (2 values per instruction)   2: v_cvt_pkrtz_f16_f32   3: v_exp_f32   4: v_max_f32
//      5: s_add_u32 (SALU)   6: v_mov_b32 via DPP row_shr   7: the f16 split of the kernels (6 VALU per 2 values)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NV, int KIND, int THREADS, int NCHAIN>
__global__ __launch_bounds__(THREADS, 1) void probe(float* out, long long* ticks, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(1e-3f * (lane + e)); b[e] = (_Float16)(2e-3f * (lane - e)); }
    f32x16 acc[NCHAIN];
#pragma unroll
    for (int j = 0; j < NCHAIN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed * (lane + i);
    unsigned sacc = (unsigned)iters;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            acc[s % NCHAIN] = MF(a, b, acc[s % NCHAIN]);
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int i = (s + q) & 15, i2 = (2 * (s + q)) & 15;
                if (KIND == 0) v[i] = __builtin_fmaf(v[i], 1.0001f, seed);
                if (KIND == 1) {
                    f32x2 t = {v[i2], v[i2 + 1]};
                    t = __builtin_elementwise_fma(t, f32x2{1.0001f, 1.0001f}, f32x2{seed, seed});
                    v[i2] = t[0]; v[i2 + 1] = t[1];
                }
                if (KIND == 2) {
                    const f16x2 hh = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(v[i2], v[i2 + 1]));
                    v[i2] = __builtin_bit_cast(float, hh);          // (bits only: keeps a dependency, no extra VALU)
                }
                if (KIND == 3) v[i] = __builtin_amdgcn_exp2f(v[i]);
                if (KIND == 4) v[i] = __builtin_fmaxf(v[i], v[(i + 1) & 15]);
                if (KIND == 5) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc));
                if (KIND == 6) v[i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[i]), 0x111, 0xf, 0xf, false));
            }
            if (KIND == 7) {
#pragma unroll
                for (int q = 0; q < NV / 6; ++q) {
                    const int i = (2 * (s + q)) & 15;
                    const float x0 = v[i] * seed, x1 = v[i + 1] * seed;
                    const f16x2 hh = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x0, x1));
                    const f16x2 ll = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x0 - (float)hh[0], x1 - (float)hh[1]));
                    v[i] = __builtin_bit_cast(float, hh);
                    v[i + 1] = __builtin_bit_cast(float, ll);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float sum = (float)sacc;
#pragma unroll
    for (int j = 0; j < NCHAIN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[j][r];
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += v[i];
    out[blockIdx.x * THREADS + threadIdx.x] = sum;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int NV, int KIND, int THREADS, int NCHAIN>
void run() {
    const int nblk = 256, iters = 300;
    float* out; long long* ticks;
    hipMalloc(&out, nblk * THREADS * 4); hipMalloc(&ticks, nblk * 8);
    auto kern = probe<NV, KIND, THREADS, NCHAIN>;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<nblk, THREADS>>>(out, ticks, 10, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<nblk, THREADS>>>(out, ticks, iters, 1e-3f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nblk);
    hipMemcpy(h.data(), ticks, nblk * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto t : h) avg += t; avg /= nblk;
    const double n = 16.0 * iters;
    static const char* kn[] = {"v_fma_f32", "v_pk_fma_f32", "v_cvt_pkrtz", "v_exp_f32", "v_max_f32", "s_add_u32", "v_mov dpp", "f16 split"};
    printf("%-12s x%2d per MFMA, %d wave(s)/SIMD, %d chain(s): wall-ns/step %6.2f\n", kn[KIND], NV, THREADS / 256, NCHAIN, ms * 1e6 / n);
    hipFree(out); hipFree(ticks);
}

int main() {
    run<0, 0, 256, 1>(); run<0, 0, 256, 2>(); run<0, 0, 512, 1>();
    run<4, 0, 256, 1>(); run<8, 0, 256, 1>(); run<16, 0, 256, 1>(); run<8, 0, 256, 2>(); run<8, 0, 512, 1>();
    run<4, 1, 256, 1>(); run<8, 1, 256, 1>();
    run<4, 2, 256, 1>(); run<8, 2, 256, 1>();
    run<4, 3, 256, 1>(); run<8, 3, 256, 1>();
    run<4, 4, 256, 1>(); run<8, 4, 256, 1>();
    run<8, 5, 256, 1>(); run<16, 5, 256, 1>();
    run<4, 6, 256, 1>(); run<8, 6, 256, 1>();
    run<6, 7, 256, 1>(); run<12, 7, 256, 1>(); run<12, 7, 512, 1>();
    return 0;
}
