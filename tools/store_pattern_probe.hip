// store_pattern_probe.hip -- why does a conv epilogue write its NHWC output at ~1 TB/s when the prep kernels write HBM at 5-6 TB/s?
// Writes a 402 MB tensor [524288 pixels][192 channels] of floats (the first layer's output at Config 2) in the patterns below and
// reports TB/s.  Variables: bytes per lane (4 / 16), how a wave's lanes map to the tensor (linear; two 128-byte row pieces of two
// pixels as the MFMA C/D layout gives; one pixel's 768-byte channel run), stores per wave back to back (16 / 96), waves per CU.
// build: hipcc -O3 --offload-arch=gfx950 tools/store_pattern_probe.hip -o tools/bin/store_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr long NPIX = 524288, C = 192;

// MODE 0: linear float4 (lane i of the grid writes element 4i..4i+3), grid-stride
// MODE 1: conv epilogue: a wave owns 32 pixels x 32 channels: 16 store instructions, each = two 128-byte pieces (rows r and r+4)
// MODE 2: conv epilogue with all six 32-channel tiles of the 32 pixels in one wave (96 store instructions back to back)
// MODE 3: float4, a wave owns 32 pixels x 192 channels and writes whole pixels (48 lanes x 16 B = one pixel's channels)
template <int MODE>
__global__ __launch_bounds__(256) void wr(float* __restrict__ out, long ntile)
{
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwave = ((long)gridDim.x * blockDim.x) >> 6;
    const float v = (float)lane;
    if (MODE == 0) {
        float4* o = reinterpret_cast<float4*>(out);
        const long n4 = NPIX * C / 4;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) o[i] = make_float4(v, v, v, v);
    } else if (MODE == 1) {
        for (long t = wave; t < ntile * 6; t += nwave) {                 // tile t: pixel block t / 6, channel tile t % 6
            const long pb = t / 6; const int j = (int)(t - pb * 6);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                out[(pb * 32 + row) * C + j * 32 + l31] = v;
            }
        }
    } else if (MODE == 2) {
        for (long pb = wave; pb < ntile; pb += nwave)
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                    out[(pb * 32 + row) * C + j * 32 + l31] = v;
                }
    } else {
        for (long pb = wave; pb < ntile; pb += nwave)
#pragma unroll
            for (int it = 0; it < 24; ++it) {                             // 32 pixels x 48 float4 = 1536 float4 / 64 lanes
                const int e = it * 64 + lane, row = e / 48, q = e - row * 48;
                *reinterpret_cast<float4*>(out + (pb * 32 + row) * C + 4 * q) = make_float4(v, v, v, v);
            }
    }
}

template <int MODE>
void run(float* d, int blocks_per_cu, const char* what)
{
    const long ntile = NPIX / 32;
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(wr<MODE>, dim3(grid), dim3(256), 0, 0, d, ntile);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(wr<MODE>, dim3(grid), dim3(256), 0, 0, d, ntile);
    (void)hipEventRecord(e1, 0);
    if (hipDeviceSynchronize() != hipSuccess) { printf("failed\n"); exit(1); }
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-78s %2d blocks/CU: %7.1f us  %5.2f TB/s\n", what, blocks_per_cu, ms * 1e3 / 5, NPIX * C * 4.0 / (ms * 1e-3 / 5) / 1e12);
    fflush(stdout);
}

int main()
{
    float* d; (void)hipMalloc(&d, NPIX * C * 4);
    for (int b : {2, 8}) {
        run<0>(d, b, "linear float4, grid-stride");
        run<1>(d, b, "MFMA C/D layout: wave = 32 px x 32 ch, 16 stores of two 128-byte row pieces");
        run<2>(d, b, "MFMA C/D layout: wave = 32 px x 192 ch, 96 stores back to back");
        run<3>(d, b, "float4: wave = 32 px x 192 ch, whole 768-byte pixels");
    }
    return 0;
}
