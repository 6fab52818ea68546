// mfma_rate_probe.hip -- f32 MFMA (v_mfma_f32_32x32x2_f32) throughput on one SIMD as a function of how the chains are spread:
// W waves per SIMD x A independent accumulators per wave, accumulators in VGPRs or AGPRs.  Reports in-kernel cycles per MFMA
// per SIMD (s_memtime), the in-kernel clock (s_memtime / s_memrealtime) and wall-clock TFLOP/s (HIP events) over the chip.
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_rate_probe.hip -o tools/bin/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int A, bool AGPR>
__global__ __launch_bounds__(1024) void rate(unsigned long long* __restrict__ out, float* __restrict__ sinkp, int iters)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[A];
    for (int k = 0; k < A; ++k)
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    float a = 1.0f + lane * 1e-3f, b = 0.5f + wave * 1e-3f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < A; ++k) {
            if (AGPR) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[k]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int k = 0; k < A; ++k)
        for (int i = 0; i < 16; ++i) s += acc[k][i];
    if (lane == 0) { out[((size_t)blockIdx.x * 16 + wave) * 2] = t1 - t0; out[((size_t)blockIdx.x * 16 + wave) * 2 + 1] = r1 - r0; }
    if (s == 123.456f) sinkp[0] = s;
}

template <int A, bool AGPR>
void run(int waves_per_simd, unsigned long long* dout, float* sink)
{
    const int threads = 256 * waves_per_simd, nblk = 256, iters = 8192 / A;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((rate<A, AGPR>), dim3(nblk), dim3(threads), 0, 0, dout, sink, iters);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL((rate<A, AGPR>), dim3(nblk), dim3(threads), 0, 0, dout, sink, iters);
    (void)hipEventRecord(e1, 0);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)nblk * 32);
    (void)hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0; int n = 0;
    for (int b = 0; b < nblk; ++b)
        for (int w = 0; w < threads / 64; ++w) { cyc += (double)h[((size_t)b * 16 + w) * 2]; real += (double)h[((size_t)b * 16 + w) * 2 + 1]; ++n; }
    cyc /= n; real /= n;
    const double mfma_per_wave = (double)iters * A;
    const double flops = 4.0 * mfma_per_wave * 4096.0 * (threads / 64) * nblk;
    printf("%d wave(s)/SIMD x %d acc (%s): %6.1f cycles per MFMA per SIMD | clock %.2f GHz | %6.1f TFLOP/s wall (one block per CU assumed)\n",
           waves_per_simd, A, AGPR ? "AGPR" : "VGPR", cyc / (mfma_per_wave * waves_per_simd), cyc / real * 0.1, flops / (ms * 1e-3) / 1e12);
}

int main()
{
    unsigned long long* dout; float* sink;
    (void)hipMalloc(&dout, 256 * 32 * 8); (void)hipMalloc(&sink, 64);
    for (int w = 1; w <= 4; ++w) {
        run<1, false>(w, dout, sink); run<1, true>(w, dout, sink);
        run<2, false>(w, dout, sink); run<2, true>(w, dout, sink);
        run<4, true>(w, dout, sink);
    }
    return 0;
}
