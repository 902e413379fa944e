// Probe: what does ONE memory-type instruction cost a wave that is otherwise issuing v_mfma_f32_32x32x16_f16 back to
// back (one wave per SIMD, 4 waves per workgroup, 256 workgroups)?  Per step: 3 MFMAs (96 matrix-pipe cycles) plus the
// fillers of the mode; ticks per step above 96 are what the fillers cost beyond what the MFMAs hide.
//   hipcc --offload-arch=gfx950 -O3 filler_cost.hip -o filler_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

enum { NONE = 0, DS_READ2, DS_WRITE128, DS_WRITE64x2, BUF_LOAD, BUF_STORE, PIECE, PIECE_AND_READS, BUF_LOAD_NT,
       DS_WRITE16x4, BUF_STORE_HOT, PIECE_SPREAD, NMODES };
static const char* kNames[NMODES] = {
    "3 MFMA only", "+ 2 ds_read_b128", "+ 1 ds_write_b128", "+ 2 ds_write_b64", "+ 1 buffer_load_dwordx4 (L2-hot)",
    "+ 1 buffer_store_dwordx4 (streaming)", "+ piece: ds_write_b128 + buffer_load_dwordx4",
    "+ piece + 2 ds_read_b128 (the QK step)", "+ 1 buffer_load_dwordx4 nt (L2-hot)", "+ 4 ds_write_b16",
    "+ 1 buffer_store_dwordx4 (same 1 KB: L2-hot)", "+ piece, write and load in different MFMA gaps"};

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* ticks, const u32x4* src, u32x4* dst, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    _Float16* lds = reinterpret_cast<_Float16*>(smem);
    for (int i = tid; i < 32768; i += 256) lds[i] = (_Float16)(1e-3f * (i % 17));
    __syncthreads();
    f16x8 q[8], a[2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) q[i][e] = (_Float16)(1e-3f * (lane + i + e));
    const _Float16* rp = lds + (lane & 31) * 264 + (lane >> 5) * 8;
    a[0] = *reinterpret_cast<const f16x8*>(rp);
    a[1] = *reinterpret_cast<const f16x8*>(rp + 16);
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(src), 0, 1 << 30, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 1 << 30, 0x00020000);
    u32x4 st = {1u, 2u, 3u, 4u};
    unsigned* wp = reinterpret_cast<unsigned*>(smem) + 16384 + tid * 4;      // 16-byte slot per thread
    const unsigned goff = (unsigned)tid * 16u;
    const unsigned wg_off = (unsigned)blockIdx.x * 65536u;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            acc[0] = MF(a[s & 1], q[s & 7], acc[0]);
            if (MODE == PIECE_SPREAD) { *reinterpret_cast<u32x4*>(wp) = st; __builtin_amdgcn_sched_barrier(0); }
            acc[1] = MF(a[s & 1], q[(s + 3) & 7], acc[1]);
            if (MODE == PIECE_SPREAD) {
                st = __builtin_amdgcn_raw_buffer_load_b128(srs, (int)goff, (int)((s & 15) * 4096u), 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            acc[2] = MF(a[(s + 1) & 1], q[s & 7], acc[2]);
            if (MODE == DS_READ2 || MODE == PIECE_AND_READS) {
                a[s & 1] = *reinterpret_cast<const f16x8*>(rp + ((s + 2) & 15) * 16);
                a[(s + 1) & 1] = *reinterpret_cast<const f16x8*>(rp + 8448 + ((s + 2) & 15) * 16);
            }
            if (MODE == DS_WRITE128 || MODE == PIECE || MODE == PIECE_AND_READS) *reinterpret_cast<u32x4*>(wp) = st;
            if (MODE == DS_WRITE64x2) {
                *reinterpret_cast<u32x2*>(wp) = u32x2{st.x, st.y};
                *reinterpret_cast<u32x2*>(wp + 2) = u32x2{st.z, st.w};
            }
            if (MODE == DS_WRITE16x4) {
                unsigned short* hp = reinterpret_cast<unsigned short*>(smem) + 40000 + lane + (tid >> 6) * 2048;
                hp[0] = (unsigned short)st.x; hp[64] = (unsigned short)st.y; hp[128] = (unsigned short)st.z; hp[192] = (unsigned short)st.w;
            }
            if (MODE == BUF_LOAD || MODE == PIECE || MODE == PIECE_AND_READS)
                st = __builtin_amdgcn_raw_buffer_load_b128(srs, (int)goff, (int)((s & 15) * 4096u), 0);
            if (MODE == BUF_LOAD_NT) st = __builtin_amdgcn_raw_buffer_load_b128(srs, (int)goff, (int)((s & 15) * 4096u), 2);
            if (MODE == BUF_STORE)
                __builtin_amdgcn_raw_buffer_store_b128(st, drs, (int)(goff + wg_off), (int)(((it * 16 + s) & 15) * 4096u), 0);
            if (MODE == BUF_STORE_HOT) __builtin_amdgcn_raw_buffer_store_b128(st, drs, (int)goff, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float sum = (float)st.x;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[j][r];
    out[blockIdx.x * 256 + tid] = sum + (float)a[0][0];
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(int nblk) {
    float* out; long long* ticks; u32x4 *src, *dst;
    hipMalloc(&out, nblk * 256 * 4); hipMalloc(&ticks, nblk * 8);
    hipMalloc(&src, 1 << 20); hipMalloc(&dst, (size_t)nblk * 65536 + (1 << 20));
    hipMemset(src, 0, 1 << 20);
    const int iters = 500;
    auto kern = probe<MODE>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<nblk, 256, 98304>>>(out, ticks, src, dst, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<nblk, 256, 98304>>>(out, ticks, src, dst, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nblk);
    hipMemcpy(h.data(), ticks, nblk * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto t : h) avg += t; avg /= nblk;
    const double n = 16.0 * iters;
    printf("%-58s blocks %4d  %7.3f ms  ticks/step %6.1f (96 = MFMA-bound)  wall-ns/step %6.1f\n", kNames[MODE], nblk, ms,
           avg / n, ms * 1e6 / n);
    hipFree(out); hipFree(ticks); hipFree(src); hipFree(dst);
}

int main() {
    for (int nblk : {256}) {
        run<NONE>(nblk); run<DS_READ2>(nblk); run<DS_WRITE128>(nblk); run<DS_WRITE64x2>(nblk); run<DS_WRITE16x4>(nblk);
        run<BUF_LOAD>(nblk); run<BUF_LOAD_NT>(nblk); run<BUF_STORE>(nblk); run<BUF_STORE_HOT>(nblk); run<PIECE>(nblk);
        run<PIECE_SPREAD>(nblk); run<PIECE_AND_READS>(nblk);
    }
    return 0;
}
