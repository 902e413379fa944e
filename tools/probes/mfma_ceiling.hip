// Round 6: what does v_mfma_f32_32x32x16_f16 SUSTAIN chip-wide on this box, and is the gap to the 2.5 PFLOP/s
// nominal a clock / power effect?  Runs each operand fill (zeros / uniform random [-1, 1) f16) for `secs` seconds of back-to-back
// launches (4 independent accumulator chains, one or two waves per SIMD, no memory traffic inside the loop) and prints, per 0.25 s
// window, issued TFLOP/s and the effective shader clock = shader ticks (s_memtime) / wall time.  tools/mfma_ceiling.sh samples
// rocm-smi (sclk, power) beside it.    hipcc --offload-arch=gfx950 -O3 mfma_ceiling.hip -o mfma_ceiling
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void burn(const _Float16* __restrict__ src, float* out, long long* ticks, int iters) {
    const int tid = threadIdx.x;
    f16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const f16x8*>(src + ((size_t)(blockIdx.x * 256 + tid) * 8 + i) * 8);
        b[i] = *reinterpret_cast<const f16x8*>(src + ((size_t)(blockIdx.x * 256 + tid) * 8 + 4 + i) * 8);
    }
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u)
            acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(u >> 2) & 3], b[u & 3], acc[u & 3], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 3.0;
    const int wgs_per_cu = argc > 2 ? atoi(argv[2]) : 1;
    const int nblk = 256 * wgs_per_cu, iters = 2000;
    const size_t nh = (size_t)nblk * 256 * 8 * 8;
    std::vector<_Float16> h(nh);
    _Float16* src; float* out; long long* ticks;
    hipMalloc(&src, nh * 2); hipMalloc(&out, (size_t)nblk * 256 * 4); hipMalloc(&ticks, nblk * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flop_per_launch = (double)nblk * 4 * iters * 64 * 2.0 * 32 * 32 * 16;
    for (int fill = 0; fill < 3; ++fill) {        // 0: zeros, 1: uniform random [-1, 1), 2: zeros again (is it the data or the warm chip?)
        srand(1);
        for (size_t i = 0; i < nh; ++i) h[i] = (fill == 1) ? (_Float16)(2.0f * rand() / RAND_MAX - 1.0f) : (_Float16)0.f;
        hipMemcpy(src, h.data(), nh * 2, hipMemcpyHostToDevice);
        printf("# fill=%s  %d workgroups of 256 threads (%d per CU), %d x 64 MFMAs per wave and launch\n", fill == 1 ? "uniform[-1,1)" : "zeros", nblk,
               wgs_per_cu, iters);
        const auto T0 = std::chrono::steady_clock::now();
        double tw = 0, fw = 0, cw = 0; int nw = 0, win = 0;
        while (true) {
            hipEventRecord(e0);
            burn<<<nblk, 256>>>(src, out, ticks, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long tk; hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost);
            tw += ms; fw += flop_per_launch; cw += (double)tk; ++nw;
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count();
            if (tw >= 250.0 || el >= secs) {
                // s_memtime / readcyclecounter ticks at a constant 100 MHz on gfx950? print both readings: ticks per MFMA and ticks per us
                printf("t=%.2fs window %d: %.1f TFLOP/s issued (%.3f of 2500), launch %.3f ms, counter ticks/MFMA %.2f, ticks/us %.1f\n", el, win++,
                       fw / (tw * 1e-3) * 1e-12, fw / (tw * 1e-3) * 1e-12 / 2500.0, tw / nw, cw / nw / ((double)iters * 64), cw / (tw * 1e3));
                fflush(stdout);
                tw = fw = cw = 0; nw = 0;
            }
            if (el >= secs) break;
        }
    }
    return 0;
}
