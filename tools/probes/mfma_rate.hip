// Ground-truth probe: how many cycles does a v_mfma_f32_32x32x2_f32 cost on this box in the
// instruction mixes our kernels use?  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

// MODE 0: one dependent chain, operands in VGPRs
// MODE 1: one dependent chain, B operand parked in AGPRs
// MODE 2: 4 independent chains
// MODE 3: one chain + LDS reads pipelined 8 ahead (like the QK loop)
template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* ticks, int iters) {
    __shared__ float lds[256 * 33];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 256 * 33; i += 256) lds[i] = 1e-3f * (i % 17);
    __syncthreads();
    float b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = 1e-3f * (lane + i);
    if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+a"(b[i]));
    }
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = 1e-3f * lane;
    const float* kl = lds + (lane >> 5) * 33 + (lane & 31);
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {
            float av[2][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) av[0][u] = kl[(2 * u) * 33];
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int bt = 0; bt < 16; ++bt) {
                if (bt + 1 < 16) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) av[(bt + 1) & 1][u] = kl[(2 * ((bt + 1) * 8 + u)) * 33];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[0] = MF(av[bt & 1][u], b[u], acc[0]);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 128; ++u) {
                if (MODE == 2) acc[u & 3] = MF(a, b[u & 15], acc[u & 3]);
                else acc[0] = MF(a, b[u & 15], acc[0]);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int nblk) {
    float* out; long long* ticks;
    hipMalloc(&out, nblk * 256 * 4); hipMalloc(&ticks, nblk * 8);
    const int iters = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<nblk, 256>>>(out, ticks, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE><<<nblk, 256>>>(out, ticks, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nblk); hipMemcpy(h.data(), ticks, nblk * 8, hipMemcpyDeviceToHost);
    double nm = 128.0 * iters;
    double tf = 2.0 * 32 * 32 * 2 * nm * nblk * 4 / (ms * 1e-3) / 1e12;
    printf("%-44s blocks %4d  %.3f ms  %6.1f TF  wall-cycles/MFMA@2.4GHz %.1f  counter-ticks/MFMA %.1f (tick rate %.0f MHz)\n",
           name, nblk, ms, tf, ms * 1e-3 * 2.4e9 / nm, h[0] / nm, h[0] / (ms * 1e-3) / 1e6);
    hipFree(out); hipFree(ticks);
}

int main() {
    for (int nblk : {256, 64}) {
        run<0>("dependent chain, VGPR operands", nblk);
        run<1>("dependent chain, B in AGPR", nblk);
        run<2>("4 independent chains", nblk);
        run<3>("dependent chain + pipelined LDS reads", nblk);
    }
    return 0;
}
