// Which part of a fused tile costs more than its 64 cycles per MFMA?  Pure-LDS, no global traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
#define SGB(m, n) __builtin_amdgcn_sched_group_barrier(m, n, 0)
__device__ __forceinline__ constexpr int rb(int r) { return (r & 3) + 8 * (r >> 2); }
constexpr int LD = 33;

// MODE 0: QK loop only (128 MFMAs, one accumulator)          per "tile"
// MODE 1: PV loop only (16 x NACC MFMAs, NACC accumulators, transposed LDS reads)
// MODE 2: QK + PV
// MODE 3: QK + softmax VALU + PV
// MODE 4: MODE 3 + __syncthreads per tile
// MODE 5: MODE 4 + 52 ds_write_b32 per thread per tile (commit)
template <int MODE, int NACC>
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* ticks, int tiles) {
    extern __shared__ float lds[];   // [2][(256 + 160)][33]
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    for (int i = tid; i < 2 * 416 * LD; i += 256) lds[i] = 1e-3f * (i % 17);
    __syncthreads();
    float q[128];
#pragma unroll
    for (int i = 0; i < 128; ++i) { q[i] = 1e-3f * (lane + i); asm volatile("" : "+a"(q[i])); }
    f32x16 o[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 1e-3f * r;
    float m_run = 0.f, l_run = 0.f;
    long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < tiles; ++t) {
        const float* kt = lds + (t & 1) * 416 * LD;
        const float* vt = kt + 256 * LD;
        if (MODE != 1) {
            f32x16 sn;
#pragma unroll
            for (int r = 0; r < 16; ++r) sn[r] = 0.f;
            const float* kl = kt + h * LD + c;
            float a[2][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[0][u] = kl[(2 * u) * LD];
            SGB(0x100, 4);
#pragma unroll
            for (int bt = 0; bt < 16; ++bt) {
                if (bt + 1 < 16) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) a[(bt + 1) & 1][u] = kl[(2 * ((bt + 1) * 8 + u)) * LD];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) sn = MF(a[bt & 1][u], q[bt * 8 + u], sn);
                if (MODE >= 3 && bt < 8) {   // softmax of the previous tile hidden here
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        const int r = bt * 2 + qq;
                        s[r] = __builtin_amdgcn_exp2f(s[r] * 1.7f - m_run);
                        l_run += s[r];
                    }
                }
                if (MODE >= 5 && bt >= 3) {
                    float* d = lds + ((t + 1) & 1) * 416 * LD + ((bt - 3) * 32 + (tid >> 3)) * LD + (tid & 7) * 4;
                    d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { SGB(0x008, 2); SGB(0x100, 1); }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] += sn[r];
            } else if (MODE == 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = sn[r] * 1e-3f;
            } else {
                m_run += 1e-6f;
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = s[r] * 0.5f + sn[r] * 1e-3f;
            }
        }
        if (MODE >= 1) {
            const float* vl = vt + c * LD + 4 * h;
            float va[2][NACC];
#pragma unroll
            for (int cb = 0; cb < NACC; ++cb) va[0][cb] = vl[cb * 32 * LD + rb(0)];
            SGB(0x100, NACC);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r + 1 < 16) {
#pragma unroll
                    for (int cb = 0; cb < NACC; ++cb) va[(r + 1) & 1][cb] = vl[cb * 32 * LD + rb(r + 1)];
                }
#pragma unroll
                for (int cb = 0; cb < NACC; ++cb) o[cb] = MF(va[r & 1][cb], s[r], o[cb]);
#pragma unroll
                for (int cb = 0; cb < NACC; ++cb) { SGB(0x008, 1); SGB(0x100, 1); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MODE >= 4) __syncthreads();
    }
    long long t1 = __builtin_readcyclecounter();
    float acc = l_run;
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc += o[j][r];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += s[r];
    out[blockIdx.x * 256 + tid] = acc;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE, int NACC>
void run(const char* name) {
    const int nblk = 256, tiles = 128;
    float* out; long long* ticks;
    hipMalloc(&out, nblk * 256 * 4); hipMalloc(&ticks, nblk * 8);
    auto k = probe<MODE, NACC>;
    const int smem = 2 * 416 * LD * 4;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<nblk, 256, smem>>>(out, ticks, 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<nblk, 256, smem>>>(out, ticks, tiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> hh(nblk); hipMemcpy(hh.data(), ticks, nblk * 8, hipMemcpyDeviceToHost);
    double nm = tiles * ((MODE != 1 ? 128.0 : 0.0) + (MODE >= 1 ? 16.0 * NACC : 0.0));
    printf("%-52s %.3f ms  ticks/MFMA %.1f  (clock %.0f MHz)  TF %.1f\n", name, ms, hh[0] / nm,
           hh[0] / (ms * 1e-3) / 1e6, 4096.0 * nm * nblk * 4 / (ms * 1e-3) / 1e12);
}

int main() {
    run<0, 1>("QK loop only");
    run<1, 5>("PV loop only, 5 accumulators");
    run<1, 8>("dX-like loop, 8 accumulators");
    run<1, 1>("PV loop only, 1 accumulator");
    run<2, 5>("QK + PV(5)");
    run<3, 5>("QK + softmax VALU hidden + PV(5)");
    run<4, 5>("... + barrier per tile");
    run<5, 5>("... + 52 ds_write per tile");
    run<5, 1>("same, PV(1)  (Cv = 3 shape)");
    return 0;
}
