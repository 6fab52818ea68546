// cu_mask_probe.hip -- which physical CUs (XCC_ID, SE, SH/array, CU of HW_ID) does a stream created with hipExtStreamCreateWithCUMask
// dispatch to?  Round 3 question: can two slice-chain streams of a codec each be given half of every XCD's CUs (so that an M = 8192
// launch has twice the chains per SIMD -- 7 instead of 3.5 at N = 224 -- and the other chain's launches do not interleave with it)?
// build: hipcc -O3 --offload-arch=gfx950 tools/cu_mask_probe.hip -o tools/bin/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

__global__ void where(unsigned* out, int spin)
{
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(0xF804), xcc = __builtin_amdgcn_s_getreg(0xF814);   // HW_REG_HW_ID, HW_REG_XCC_ID
        out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc;
    }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {}          // keep the slot busy so that the grid spreads
}

static void run(const char* name, const std::vector<uint32_t>& mask)
{
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%-34s hipExtStreamCreateWithCUMask failed: %s\n", name, hipGetErrorString(e)); (void)hipGetLastError(); return; }
    const int nb = 4096;
    unsigned* d;
    (void)hipMalloc(&d, nb * 2 * sizeof(unsigned));
    (void)hipMemset(d, 0, nb * 2 * sizeof(unsigned));
    hipLaunchKernelGGL(where, dim3(nb), dim3(256), 0, s, d, 20000);
    (void)hipStreamSynchronize(s);
    std::vector<unsigned> h(nb * 2);
    (void)hipMemcpy(h.data(), d, nb * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<int, std::set<int>> per_xcc;
    for (int b = 0; b < nb; ++b) {
        const unsigned hw = h[b * 2], xcc = h[b * 2 + 1] & 15u;
        const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_xcc[(int)xcc].insert((se * 2 + sh) * 16 + cu);
    }
    int total = 0;
    printf("%-34s", name);
    for (auto& kv : per_xcc) { printf(" xcc%d:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  = %d CUs\n", total);
    (void)hipFree(d);
    (void)hipStreamDestroy(s);
}

int main()
{
    run("all 256 bits", std::vector<uint32_t>(8, 0xffffffffu));
    run("bits 0..127", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0});
    run("bits 128..255", {0, 0, 0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu});
    run("even bits", std::vector<uint32_t>(8, 0x55555555u));
    run("bits with (i % 8) < 4", std::vector<uint32_t>(8, 0x0f0f0f0fu));
    run("bits with (i / 8) even", std::vector<uint32_t>(8, 0x00ff00ffu));
    run("bits with (i / 16) even", std::vector<uint32_t>(8, 0x0000ffffu));
    run("bits 0..31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0});
    run("bits 0..7", {0xffu, 0, 0, 0, 0, 0, 0, 0});
    return 0;
}
