// Probe: does the issue rate of v_mfma_f32_32x32x16_f16 depend on WHERE its operands live (arch VGPRs vs the
// accumulator half of the register file) and on the accumulator reuse pattern?  One wave per SIMD, 256 workgroups.
//   hipcc --offload-arch=gfx950 -O3 mfma_operand_rate.hip -o mfma_operand_rate
// MODE 0: B operand in VGPRs,  1 accumulator      MODE 1: B operand in AGPRs, 1 accumulator
// MODE 2: B operand in AGPRs, 3 accumulators round-robin   MODE 3: B in VGPRs, 3 accumulators
// MODE 4: B in AGPRs, steps of 3 MFMAs on one accumulator, 5 accumulators in turn (the P.V pattern)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* ticks, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 q[32], a[4];
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) q[i][e] = (_Float16)(1e-3f * (lane + i + e));
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[i][e] = (_Float16)(1e-3f * (lane - i + e));
    if (MODE == 1 || MODE == 2 || MODE == 4) {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("" : "+a"(q[i]));
    } else {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(q[i]));
    }
    f32x16 acc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 48; ++u) {
            int j = 0;
            if (MODE == 2 || MODE == 3) j = u % 3;
            if (MODE == 4) j = (u / 3) % 5;
            acc[j] = MF(a[u & 3], q[(u * 7) & 31], acc[j]);
        }
        if (MODE == 1 || MODE == 2 || MODE == 4) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("" : "+a"(q[i]));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int nblk) {
    float* out; long long* ticks;
    hipMalloc(&out, nblk * 256 * 4); hipMalloc(&ticks, nblk * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<nblk, 256>>>(out, ticks, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE><<<nblk, 256>>>(out, ticks, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nblk);
    hipMemcpy(h.data(), ticks, nblk * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto t : h) avg += t; avg /= nblk;
    const double n = 48.0 * iters;
    printf("%-60s blocks %4d  %7.3f ms  ticks/MFMA %5.1f  wall-ns/MFMA %5.2f\n", name, nblk, ms, avg / n, ms * 1e6 / n);
    hipFree(out); hipFree(ticks);
}

int main() {
    for (int nblk : {256, 64}) {
        run<0>("B in VGPRs, 1 accumulator", nblk);
        run<1>("B in AGPRs, 1 accumulator", nblk);
        run<2>("B in AGPRs, 3 accumulators round-robin", nblk);
        run<3>("B in VGPRs, 3 accumulators round-robin", nblk);
        run<4>("B in AGPRs, 3 MFMAs per accumulator, 5 accumulators in turn", nblk);
    }
    return 0;
}
