// build: hipcc -O2 -ffp-contract=off --offload-arch=gfx950 tools/mfma16x16x4_probe.hip -o /tmp/mfma16 ; run on an MI355X.  Result (round 1): no
// permutation of the four k reproduces a sequential fmaf chain (a few random coincidences only) -- see DESIGN.md section 7.
// Probe: is v_mfma_f32_16x16x4_f32 an exact sequential fmaf chain over its four k (lane groups 0-15, 16-31, 32-47, 48-63), and in
// which order?  D[i][j] = C[i][j] + sum_k A[i][k] B[k][j].  A: lane l holds A[i = l%16][k = l/16]; B: lane l holds B[k = l/16][j = l%16];
// C/D: 4 VGPRs, lane l, reg r -> row i = 4*(l/16) + r, col j = l%16.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, const float* C, float* D)
{
    const int l = threadIdx.x;
    f32x4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[(4 * (l / 16) + r) * 16 + (l % 16)];
    const float a = A[(l % 16) * 4 + (l / 16)], b = B[(l / 16) * 16 + (l % 16)];
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + (l % 16)] = c[r];
}
int main()
{
    std::vector<float> A(64), B(64), C(256), D(256);
    unsigned s = 12345;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 8) % 20001 - 10000) / 3000.0f * (1.0f + (float)(s & 255) * 1e-3f); };
    int perm[4] = {0, 1, 2, 3}, best_perm[4] = {0, 0, 0, 0};
    int trials = 200, exact_all[24] = {0}, pi = 0;
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    std::vector<std::vector<int>> perms;
    do { perms.push_back({perm[0], perm[1], perm[2], perm[3]}); } while (std::next_permutation(perm, perm + 4));
    for (int t = 0; t < trials; ++t) {
        for (auto& v : A) v = rnd();
        for (auto& v : B) v = rnd();
        for (auto& v : C) v = rnd() * 1e3f;
        hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        pi = 0;
        for (auto& pm : perms) {
            bool ok = true;
            for (int i = 0; i < 16 && ok; ++i)
                for (int j = 0; j < 16; ++j) {
                    float acc = C[i * 16 + j];
                    for (int q = 0; q < 4; ++q) acc = fmaf(A[i * 4 + pm[q]], B[pm[q] * 16 + j], acc);
                    if (memcmp(&acc, &D[i * 16 + j], 4)) { ok = false; break; }
                }
            exact_all[pi++] += ok;
        }
    }
    for (size_t q = 0; q < perms.size(); ++q)
        if (exact_all[q]) printf("order %d%d%d%d: bit-exact sequential fmaf chain in %d of %d random trials\n", perms[q][0], perms[q][1], perms[q][2], perms[q][3], exact_all[q], trials);
    printf("done\n");
    (void)best_perm;
    return 0;
}
