// mfma_1xNb_probe.hip -- the K = 1 multi-block f32 MFMA forms of gfx950:
//   v_mfma_f32_32x32x1_2b_f32 (2 blocks of 32x32), v_mfma_f32_16x16x1_4b_f32 (4 blocks of 16x16), v_mfma_f32_4x4x1_16b_f32 (16 of 4x4).
// One product per accumulator per instruction, so a sequence of them should be a sequential fmaf chain in ISSUE order by construction
// (the numeric contract of include/pc_math.h then holds with any k order we choose to issue, and the tile shape is free).
// (a) bit-compare each form, over a chain of KSTEPS instructions, with fmaf in issue order -- incl. the CBSZ/ABID (A broadcast) and BLGP
//     (B lane-group) modifiers that turn four 16x16 blocks into one 16x64 / 64x16 tile;
// (b) issue rate per SIMD: dependent chains, W waves per SIMD, with and without the ds_read_b128 traffic a 64x16 / 16x64 wave tile needs
//     (two 16-byte operand reads per four K = 1 instructions -- twice the LDS bytes per FLOP of the 32x32x2 form the kernel uses today).
// build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/mfma_1xNb_probe.hip -o tools/bin/mfma_1xNb_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

#define KSTEPS 24

// ---------------------------------------------------------------------------------------------------------------- (a) exactness
// A, B: [KSTEPS][64] per-lane operands; C, D: [64 lanes][NREG].  The host model decodes (block, i, j) of every (lane, reg).
template <int CBSZ, int ABID, int BLGP>
__global__ void k_32x32x1(const float* A, const float* B, const float* C, float* D)
{
    const int l = threadIdx.x;
    f32x32 c;
    for (int r = 0; r < 32; ++r) c[r] = C[l * 32 + r];
    for (int k = 0; k < KSTEPS; ++k) c = __builtin_amdgcn_mfma_f32_32x32x1f32(A[k * 64 + l], B[k * 64 + l], c, CBSZ, ABID, BLGP);
    for (int r = 0; r < 32; ++r) D[l * 32 + r] = c[r];
}
template <int CBSZ, int ABID, int BLGP>
__global__ void k_16x16x1(const float* A, const float* B, const float* C, float* D)
{
    const int l = threadIdx.x;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = C[l * 16 + r];
    for (int k = 0; k < KSTEPS; ++k) c = __builtin_amdgcn_mfma_f32_16x16x1f32(A[k * 64 + l], B[k * 64 + l], c, CBSZ, ABID, BLGP);
    for (int r = 0; r < 16; ++r) D[l * 16 + r] = c[r];
}
template <int CBSZ, int ABID, int BLGP>
__global__ void k_4x4x1(const float* A, const float* B, const float* C, float* D)
{
    const int l = threadIdx.x;
    f32x4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[l * 4 + r];
    for (int k = 0; k < KSTEPS; ++k) c = __builtin_amdgcn_mfma_f32_4x4x1f32(A[k * 64 + l], B[k * 64 + l], c, CBSZ, ABID, BLGP);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}

struct Form { const char* name; int bs, nblk, nreg; };   // block size, blocks, accumulator registers per lane

// (lane, reg) -> (block, i, j) of the C/D layout
static void cd_index(const Form& f, int lane, int r, int& blk, int& i, int& j)
{
    if (f.bs == 32) { blk = r >> 4; j = lane & 31; i = (r & 3) + 8 * ((r & 15) >> 2) + 4 * (lane >> 5); }
    else if (f.bs == 16) { blk = r >> 2; j = lane & 15; i = 4 * (lane >> 4) + (r & 3); }
    else { blk = lane >> 2; j = lane & 3; i = r; }
}
// which lane supplies A[i] of block blk under CBSZ/ABID (A of block ABID of each group of 2^CBSZ blocks is broadcast to the group)
static int a_lane(const Form& f, int blk, int i, int cbsz, int abid)
{
    const int g = 1 << cbsz;
    const int src = cbsz ? (blk / g) * g + (abid % g) : blk;
    return src * f.bs + i;
}
// which lane supplies B[j] of block blk under BLGP (lane-group pattern applied to the 64-lane B register)
static int b_lane(const Form& f, int blk, int j, int blgp)
{
    const int l = blk * f.bs + j;
    switch (blgp) {
    case 0: return l;
    case 1: return l & 31;                 // lanes 0-31 broadcast to both halves
    case 2: return 32 + (l & 31);          // lanes 32-63
    case 3: return (l + 16) & 63;          // rotate by 16 (direction checked by the alternative below)
    case 4: return l & 15;                 // lanes 0-15 to all four groups
    case 5: return 16 + (l & 15);
    case 6: return 32 + (l & 15);
    default: return 48 + (l & 15);
    }
}

template <typename KF>
static int check_form(const Form& f, KF kernel, int cbsz, int abid, int blgp, int trials, const char* tag)
{
    const int nreg = f.nreg;
    std::vector<float> A(KSTEPS * 64), B(KSTEPS * 64), Cc(64 * nreg), D(64 * nreg);
    float *dA, *dB, *dC, *dD;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dC, Cc.size() * 4); (void)hipMalloc(&dD, D.size() * 4);
    unsigned s = 777u + 31u * (unsigned)f.bs + 7u * (unsigned)cbsz + 3u * (unsigned)blgp;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 8) % 20001 - 10000) / 3000.0f * (1.0f + (float)(s & 255) * 1e-3f); };
    long bad = 0, total = 0, bad_unfused = 0;
    for (int t = 0; t < trials; ++t) {
        for (auto& v : A) v = rnd();
        for (auto& v : B) v = rnd();
        for (auto& v : Cc) v = rnd() * (t & 1 ? 1e3f : 1.0f);
        (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dC, Cc.data(), Cc.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        if (hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("%s: launch failed\n", tag); return 1; }
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < nreg; ++r) {
                int blk, i, j;
                cd_index(f, l, r, blk, i, j);
                const int la = a_lane(f, blk, i, cbsz, abid), lb = b_lane(f, blk, j, blgp);
                float acc = Cc[l * nreg + r], acc2 = acc;
                for (int k = 0; k < KSTEPS; ++k) {
                    acc = fmaf(A[k * 64 + la], B[k * 64 + lb], acc);
                    volatile float p = A[k * 64 + la] * B[k * 64 + lb];
                    acc2 = acc2 + p;
                }
                ++total;
                if (memcmp(&acc, &D[l * nreg + r], 4)) ++bad;
                if (memcmp(&acc2, &D[l * nreg + r], 4)) ++bad_unfused;
            }
    }
    printf("(a) %-28s cbsz=%d abid=%d blgp=%d: %ld / %ld accumulators differ from the fmaf chain in issue order over %d steps (%ld differ from the unfused mul+add chain)  %s\n",
           f.name, cbsz, abid, blgp, bad, total, KSTEPS, bad_unfused, bad == 0 ? "EXACT" : "not exact under this operand map");
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dD);
    return bad != 0;
}

// ---------------------------------------------------------------------------------------------------------------- (b) issue rate
// FORM 0: 32x32x2 (the kernel's instruction), 1: 32x32x1_2b, 2: 16x16x1_4b, 3: 4x4x1_16b.  Every wave runs ONE dependent chain (the
// slice-chain situation: one accumulator tile per wave); LDS = 0: operands in registers; LDS = 1: per group of four instructions the
// wave issues the two ds_read_b128 a [row][k-quad] LDS image needs for them (forms 1-3: one 16-byte piece per operand covers four k of
// K = 1 instructions; form 0: the same two reads cover 8 k = four 32x32x2) -- i.e. the same read instructions per MFMA *instruction*,
// twice the bytes per FLOP for forms 2-3.
template <int FORM, int LDS>
__global__ __launch_bounds__(1024) void rate(unsigned long long* __restrict__ out, float* __restrict__ sinkp, int iters)
{
    __shared__ float lds[4096 + 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096 + 64; i += blockDim.x) lds[i] = 1.0f + (float)i * 1e-6f;
    f32x32 acc32;
    f32x16 acc16;
    f32x4 acc4;
    for (int i = 0; i < 32; ++i) acc32[i] = 0.f;
    for (int i = 0; i < 16; ++i) acc16[i] = 0.f;
    for (int i = 0; i < 4; ++i) acc4[i] = 0.f;
    f32x4 a = {1.0f + lane * 1e-3f, 1.1f, 1.2f, 1.3f}, b = {0.5f + wave * 1e-3f, 0.6f, 0.7f, 0.8f};
    const unsigned aoff = (unsigned)(((lane * 4 + wave * 16) & 1023) * 4), boff = (unsigned)(((lane * 4 + 2048 + wave * 16) & 4095) * 4);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(b) : "v"(aoff), "v"(boff) : "memory");
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (FORM == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc16) : "v"(a[q]), "v"(b[q]));
            if (FORM == 1) asm volatile("v_mfma_f32_32x32x1_2b_f32 %0, %1, %2, %0" : "+v"(acc32) : "v"(a[q]), "v"(b[q]));
            if (FORM == 2) asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0" : "+v"(acc16) : "v"(a[q]), "v"(b[q]));
            if (FORM == 3) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc4) : "v"(a[q]), "v"(b[q]));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += acc32[i];
    for (int i = 0; i < 16; ++i) s += acc16[i];
    for (int i = 0; i < 4; ++i) s += acc4[i];
    if (lane == 0) { out[((size_t)blockIdx.x * 16 + wave) * 2] = t1 - t0; out[((size_t)blockIdx.x * 16 + wave) * 2 + 1] = r1 - r0; }
    if (s == 123.456f) sinkp[0] = s;
}

template <int FORM, int LDS>
static void run_rate(int waves_per_simd, unsigned long long* dout, float* sink)
{
    static const char* names[4] = {"32x32x2 (today)", "32x32x1_2b", "16x16x1_4b", "4x4x1_16b"};
    static const double flop_per_instr[4] = {4096, 4096, 2048, 512};
    const int threads = 256 * waves_per_simd, nblk = 256;
    const int iters = FORM == 3 ? 16384 : (FORM == 2 ? 4096 : 2048);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((rate<FORM, LDS>), dim3(nblk), dim3(threads), 0, 0, dout, sink, iters);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL((rate<FORM, LDS>), dim3(nblk), dim3(threads), 0, 0, dout, sink, iters);
    (void)hipEventRecord(e1, 0);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)nblk * 32);
    (void)hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    int n = 0;
    for (int b = 0; b < nblk; ++b)
        for (int w = 0; w < threads / 64; ++w) { cyc += (double)h[((size_t)b * 16 + w) * 2]; real += (double)h[((size_t)b * 16 + w) * 2 + 1]; ++n; }
    cyc /= n; real /= n;
    const double instr_per_wave = (double)iters * 4;
    const double flops = 4.0 * flop_per_instr[FORM] * instr_per_wave * (threads / 64) * nblk;
    printf("(b) %-16s %d wave(s)/SIMD, %s: %6.1f cycles per instruction per SIMD | %6.1f TFLOP/s wall | clock %.2f GHz\n", names[FORM], waves_per_simd,
           LDS ? "2 ds_read_b128 per 4 instr" : "operands in registers   ", cyc / (instr_per_wave * waves_per_simd), flops / (ms * 1e-3) / 1e12, cyc / real * 0.1);
}

int main()
{
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) { printf("no HIP device\n"); return 1; }
    printf("device: %s (%s), %d CUs\n", p.name, p.gcnArchName, p.multiProcessorCount);
    const Form F32 = {"v_mfma_f32_32x32x1_2b_f32", 32, 2, 32}, F16 = {"v_mfma_f32_16x16x1_4b_f32", 16, 4, 16}, F4 = {"v_mfma_f32_4x4x1_16b_f32", 4, 16, 4};
    const int T = 40;
    int fails = 0;
    fails += check_form(F32, k_32x32x1<0, 0, 0>, 0, 0, 0, T, "32");
    fails += check_form(F16, k_16x16x1<0, 0, 0>, 0, 0, 0, T, "16");
    fails += check_form(F4, k_4x4x1<0, 0, 0>, 0, 0, 0, T, "4");
    // A broadcast: block ABID of each group of 2^CBSZ blocks feeds the whole group -> 32x64 / 16x64 tiles from one A block
    check_form(F32, k_32x32x1<1, 0, 0>, 1, 0, 0, T, "32 cbsz1");
    check_form(F32, k_32x32x1<1, 1, 0>, 1, 1, 0, T, "32 cbsz1 abid1");
    check_form(F16, k_16x16x1<2, 0, 0>, 2, 0, 0, T, "16 cbsz2");
    check_form(F16, k_16x16x1<2, 3, 0>, 2, 3, 0, T, "16 cbsz2 abid3");
    check_form(F16, k_16x16x1<1, 1, 0>, 1, 1, 0, T, "16 cbsz1 abid1");
    check_form(F4, k_4x4x1<4, 0, 0>, 4, 0, 0, T, "4 cbsz4");
    // B lane-group patterns: one B block feeds all blocks -> 64x32 / 64x16 tiles from one B block
    check_form(F32, k_32x32x1<0, 0, 1>, 0, 0, 1, T, "32 blgp1");
    check_form(F32, k_32x32x1<0, 0, 2>, 0, 0, 2, T, "32 blgp2");
    check_form(F16, k_16x16x1<0, 0, 4>, 0, 0, 4, T, "16 blgp4");
    check_form(F16, k_16x16x1<0, 0, 7>, 0, 0, 7, T, "16 blgp7");
    check_form(F16, k_16x16x1<0, 0, 3>, 0, 0, 3, T, "16 blgp3");
    check_form(F16, k_16x16x1<0, 0, 1>, 0, 0, 1, T, "16 blgp1");

    unsigned long long* dout;
    float* sink;
    (void)hipMalloc(&dout, 256 * 32 * 8); (void)hipMalloc(&sink, 64);
    for (int w = 1; w <= 4; w += (w == 1 ? 1 : 2)) {
        run_rate<0, 0>(w, dout, sink); run_rate<0, 1>(w, dout, sink);
        run_rate<1, 0>(w, dout, sink); run_rate<1, 1>(w, dout, sink);
        run_rate<2, 0>(w, dout, sink); run_rate<2, 1>(w, dout, sink);
        run_rate<3, 0>(w, dout, sink); run_rate<3, 1>(w, dout, sink);
    }
    printf("plain forms exact: %s\n", fails ? "NO" : "yes");
    return 0;
}
