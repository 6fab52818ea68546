// mfma_issue_probe.hip -- what does an instruction cost next to a saturating f32 MFMA chain on one SIMD?
//
// Each MFMA wave runs a dependent chain of v_mfma_f32_32x32x2_f32 (64 cycles each at best) and, after every MFMA, F filler
// instructions of one kind: independent v_fma, ds_read_b128, LDS-DMA (buffer_load ... lds from an L2-resident buffer), s_nop.
// Cycles per loop iteration (s_memtime) tell how many fillers hide in an MFMA's shadow.  Modes:
//   same      : 1 wave per SIMD (256 threads), fillers in the MFMA wave itself
//   same x2   : 2 waves per SIMD (512 threads), both MFMA + fillers (two chains share the pipe: 128 cycles per iteration at best)
//   partner   : 2 waves per SIMD, waves 0-3 pure MFMA, waves 4-7 pure fillers (the round-1 loader-wave arrangement); reports the
//               MFMA wave's cycles per MFMA and the filler wave's cycles per filler group
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_issue_probe.hip -o gpurun_out/mfma_issue_probe   (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { F_NONE = 0, F_VALU = 1, F_DSREAD = 2, F_DMA = 3, F_SNOP = 4 };

template <int KIND, int F>
__device__ __forceinline__ void fillers(float (&v)[16], f32x4 (&r)[4], uint32_t lds_addr, __amdgpu_buffer_rsrc_t rs, int voff, float4* lds)
{
#pragma unroll
    for (int f = 0; f < F; ++f) {
        if (KIND == F_VALU) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[f & 15]));
        else if (KIND == F_DSREAD) asm volatile("ds_read_b128 %0, %1" : "=v"(r[f & 3]) : "v"(lds_addr));
        else if (KIND == F_DMA)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 64 * (f & 3)), 16, voff, 0, 0, 0);
        else if (KIND == F_SNOP) asm volatile("s_nop 0");
    }
}

template <int KIND, int F, int MODE>   // MODE 0: every wave MFMA + fillers; 1: waves >= 4 fillers only, waves < 4 MFMA only
__global__ __launch_bounds__(512) void probe(const float* __restrict__ src, unsigned long long* __restrict__ out, int iters)
{
    extern __shared__ float4 lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float4* my = lds + wave * 256;
    my[lane] = make_float4(1.f, 2.f, 3.f, 4.f);
    __syncthreads();
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 1.0f + lane * 1e-3f;
    f32x4 r[4] = {};
    const uint32_t lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(my + lane);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x7fffffff, 0x00020000);
    const int voff = ((blockIdx.x * 8 + wave) & 255) * 1024 + lane * 16;
    float a = 1.0f + lane, b = 0.5f;
    const bool do_mfma = MODE == 0 || wave < 4, do_fill = MODE == 0 || wave >= 4;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (do_mfma && do_fill) {
        for (int it = 0; it < iters; ++it) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            fillers<KIND, F>(v, r, lds_addr, rs, voff, my);
            __builtin_amdgcn_sched_barrier(0);
            if (KIND == F_DSREAD && (it & 7) == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (KIND == F_DMA && (it & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    } else if (do_mfma) {
        for (int it = 0; it < iters; ++it) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
    } else {
        for (int it = 0; it < iters; ++it) {
            fillers<KIND, F>(v, r, lds_addr, rs, voff, my);
            __builtin_amdgcn_sched_barrier(0);
            if (KIND == F_DSREAD && (it & 7) == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (KIND == F_DMA && (it & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sink = 0.f;
    for (int i = 0; i < 16; ++i) sink += acc[i] + v[i];
    for (int i = 0; i < 4; ++i) sink += r[i].x;
    if (lane == 0) out[(size_t)blockIdx.x * 8 + wave] = t1 - t0;
    if (sink == 123.456f) out[0] = 0;   // keep everything live
}

template <int KIND, int F, int MODE>
void run(const char* kname, int threads, const float* src, unsigned long long* dout, int iters)
{
    const int nblk = 256;
    hipLaunchKernelGGL((probe<KIND, F, MODE>), dim3(nblk), dim3(threads), 32 * 1024, 0, src, dout, iters);
    hipLaunchKernelGGL((probe<KIND, F, MODE>), dim3(nblk), dim3(threads), 32 * 1024, 0, src, dout, iters);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
    std::vector<unsigned long long> h((size_t)nblk * 8);
    (void)hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    double m = 0, f = 0; int nm = 0, nf = 0;
    for (int b = 0; b < nblk; ++b)
        for (int w = 0; w < nw; ++w) {
            const double c = (double)h[(size_t)b * 8 + w] / iters;
            if (MODE == 0 || w < 4) { m += c; ++nm; } else { f += c; ++nf; }
        }
    if (MODE == 0) printf("%-8s F=%2d  %s  cycles/iter %7.1f\n", kname, F, threads == 256 ? "same   (1 wave/SIMD)" : "same x2 (2 waves/SIMD)", m / nm);
    else printf("%-8s F=%2d  partner (MFMA wave | filler wave)  cycles/MFMA %7.1f | cycles/filler-group %7.1f\n", kname, F, m / nm, f / nf);
}

template <int KIND>
void sweep(const char* kname, const float* src, unsigned long long* dout, int iters)
{
    run<KIND, 1, 0>(kname, 256, src, dout, iters);
    run<KIND, 2, 0>(kname, 256, src, dout, iters);
    run<KIND, 4, 0>(kname, 256, src, dout, iters);
    run<KIND, 8, 0>(kname, 256, src, dout, iters);
    run<KIND, 16, 0>(kname, 256, src, dout, iters);
    run<KIND, 1, 0>(kname, 512, src, dout, iters);
    run<KIND, 4, 0>(kname, 512, src, dout, iters);
    run<KIND, 8, 0>(kname, 512, src, dout, iters);
    run<KIND, 1, 1>(kname, 512, src, dout, iters);
    run<KIND, 4, 1>(kname, 512, src, dout, iters);
    run<KIND, 16, 1>(kname, 512, src, dout, iters);
}

int main()
{
    float* src; unsigned long long* dout;
    (void)hipMalloc(&src, 1 << 20); (void)hipMemset(src, 0, 1 << 20);
    (void)hipMalloc(&dout, 256 * 8 * 8);
    const int iters = 4096;
    run<F_NONE, 0, 0>("none", 256, src, dout, iters);
    run<F_NONE, 0, 0>("none", 512, src, dout, iters);
    sweep<F_VALU>("v_fma", src, dout, iters);
    sweep<F_DSREAD>("ds_read", src, dout, iters);
    sweep<F_DMA>("lds_dma", src, dout, iters);
    sweep<F_SNOP>("s_nop", src, dout, iters);
    return 0;
}
