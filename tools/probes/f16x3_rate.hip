// Probe for the split-precision plan (fp32 operands as f16 hi + f16 lo, 3 MFMA terms, fp32 accumulate):
// what does v_mfma_f32_32x32x16_f16 sustain chip-wide on this box, alone and in the operand mix the
// correlation loop would use (per k-step: 2 ds_read_b128 of the key tile, 3 MFMAs against a register-
// resident query slice)?   hipcc --offload-arch=gfx950 -O3 f16x3_rate.hip -o f16x3_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

// MODE 0: one dependent chain;  MODE 1: 2 chains;  MODE 2: 4 chains
// MODE 3: QK-like: 16 k-steps x (2 LDS b128 reads + 3 MFMAs on 2 accumulators), reads one step ahead
template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* ticks, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[2 * 32 * 264];   // [hi|lo][key 32][ch 256 + 8 pad]
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * 32 * 264; i += 256) lds[i] = (_Float16)(1e-3f * (i % 17));
    __syncthreads();
    f16x8 qh[16], ql[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { qh[i][e] = (_Float16)(1e-3f * (lane + i + e)); ql[i][e] = (_Float16)(1e-4f * (lane + e)); }
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const _Float16* kh = lds + (lane & 31) * 264 + (lane >> 5) * 8;
    const _Float16* kl = kh + 32 * 264;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {
            f16x8 ah[2], al[2];
            ah[0] = *reinterpret_cast<const f16x8*>(kh);
            al[0] = *reinterpret_cast<const f16x8*>(kl);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int cur = s & 1, nxt = cur ^ 1;
                if (s + 1 < 16) {
                    ah[nxt] = *reinterpret_cast<const f16x8*>(kh + (s + 1) * 16);
                    al[nxt] = *reinterpret_cast<const f16x8*>(kl + (s + 1) * 16);
                }
                acc[0] = MF(ah[cur], qh[s], acc[0]);
                acc[1] = MF(ah[cur], ql[s], acc[1]);
                acc[1] = MF(al[cur], qh[s], acc[1]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 48; ++u) {
                const int j = MODE == 0 ? 0 : MODE == 1 ? (u & 1) : (u & 3);
                acc[j] = MF(qh[u & 15], ql[(u + 3) & 15], acc[j]);
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
    double mf = 48.0 * iters;                       // MFMAs per wave
    double flops = mf * 32 * 32 * 16 * 2 * 4.0 * nblk;
    printf("%-44s blocks %4d  %.3f ms  %7.1f TF issued (%6.1f TF fp32-equivalent /3)  ticks/MFMA %.1f  wall-cycles/MFMA@2.4GHz %.1f\n",
           name, nblk, ms, flops / ms / 1e9, flops / ms / 1e9 / 3, (double)h[0] / mf, ms * 1e-3 * 2.4e9 / mf);
    hipFree(out); hipFree(ticks);
}

int main() {
    for (int nblk : {256, 64}) {
        run<0>("f16 32x32x16: one dependent chain", nblk);
        run<1>("f16 32x32x16: 2 chains", nblk);
        run<2>("f16 32x32x16: 4 chains", nblk);
        run<3>("QK-like: 2 LDS b128 + 3 MFMA per k-step", nblk);
    }
    return 0;
}
