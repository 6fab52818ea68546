// rans_device_probe.hip -- what would an on-device rANS decoder ("rans64_dec, one lane per stream") cost?
// One slice of the codec at Config 2: 32 streams (images) x 8192 symbols, GaussianConditional tables (64 rows, stride 3129), the
// index distribution of the synthetic-weight codec (most symbols on the narrow rows).  The strings are made by the library's own
// encoder; the device decoder is the reference's loop (rans_interface.cpp:206-275) with one lane per stream, (a) binary search in
// the CDF row, (b) a 256-entry start table per row as in the host fast path; its output is checked against the symbols.  Compared
// with pc_rans_decode_batch_u8 on the host threads.  Result: see DESIGN.md section 4 (host rANS row).
// build: hipcc -O3 --offload-arch=gfx950 tools/rans_device_probe.hip -Iinclude -Lprogressivecodec_amd -lpcodec
//        -Wl,-rpath,'$ORIGIN/../../progressivecodec_amd' -o tools/bin/rans_device_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "pcodec.h"

constexpr int NROW = 64, NSTREAM = 32, NSYM = 8192;

template <bool LUT>
__global__ void rans_dec(const uint8_t* __restrict__ bytes, const uint32_t* __restrict__ off, const uint32_t* __restrict__ len,
                         const uint8_t* __restrict__ idx, const int32_t* __restrict__ cdfs, int stride, const int32_t* __restrict__ sizes,
                         const int32_t* __restrict__ offsets, const uint16_t* __restrict__ lut, int32_t* __restrict__ out, int* __restrict__ err)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= NSTREAM) return;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(bytes + off[t]);
    const uint32_t nw = len[t] / 4;
    uint64_t x = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    uint32_t p = 2;
    bool trunc = false;
    auto renorm = [&]() { if (x < (1ull << 31)) { if (p >= nw) { trunc = true; return; } x = (x << 32) | w[p++]; } };
    auto get_bits = [&]() -> int32_t { const int32_t v = (int32_t)(x & 15u); x >>= 4; renorm(); return v; };
    for (int i = 0; i < NSYM; ++i) {
        const int ci = idx[(size_t)t * NSYM + i];
        const int32_t ln = sizes[ci];
        const int32_t* cdf = cdfs + (size_t)ci * stride;
        const uint32_t cf = (uint32_t)(x & 0xFFFFu);
        int32_t lo = 0, hi = ln - 1;
        if (LUT) { lo = lut[ci * 256 + (cf >> 8)]; hi = lut[ci * 256 + (cf >> 8) + 16384]; }       // first / last candidate of this cf bucket
        while (hi - lo > 1) { const int32_t mid = (lo + hi) >> 1; if ((uint32_t)cdf[mid] <= cf) lo = mid; else hi = mid; }
        const uint32_t start = (uint32_t)cdf[lo], freq = (uint32_t)(cdf[lo + 1] - cdf[lo]);
        x = (uint64_t)freq * (x >> 16) + (x & 0xFFFFu) - start;
        renorm();
        int32_t value = lo;
        if (value == ln - 2) {
            int32_t val = get_bits(), nb = val;
            while (val == 15 && !trunc) { val = get_bits(); nb += val; }
            int32_t raw = 0;
            for (int32_t j = 0; j < nb; ++j) { val = get_bits(); if (j < 8) raw |= val << (j * 4); }
            value = raw >> 1;
            if (raw & 1) value = -value - 1; else value += ln - 2;
        }
        out[(size_t)t * NSYM + i] = value + offsets[ci];
    }
    if (trunc) atomicAdd(err, 1);
}

int main()
{
    // GaussianConditional tables as entropy.py builds them (entropy_models.py:599-624), through the library's quantiser
    std::vector<float> st(NROW);
    for (int i = 0; i < NROW; ++i) st[i] = std::exp(std::log(0.11f) + (std::log(256.0f) - std::log(0.11f)) * i / (NROW - 1));
    const double mult = 6.109410204869;                        // -norm.ppf(1e-9 / 2)
    std::vector<int> center(NROW), plen(NROW);
    int maxlen = 0;
    for (int i = 0; i < NROW; ++i) { center[i] = (int)std::ceil(st[i] * mult); plen[i] = 2 * center[i] + 1; maxlen = std::max(maxlen, plen[i]); }
    const int stride = maxlen + 2;
    std::vector<int32_t> cdf((size_t)NROW * stride, 0), sizes(NROW), offsets(NROW);
    for (int i = 0; i < NROW; ++i) {
        std::vector<float> pmf(plen[i] + 1);
        auto cum = [](double v) { return 0.5 * std::erfc(-v / std::sqrt(2.0)); };
        for (int k = 0; k < plen[i]; ++k) { const double s = std::abs(k - center[i]); pmf[k] = (float)(cum((0.5 - s) / st[i]) - cum((-0.5 - s) / st[i])); }
        pmf[plen[i]] = (float)(2 * cum((-0.5 - center[i]) / st[i]));
        std::vector<uint32_t> q(plen[i] + 2);
        if (pc_pmf_to_quantized_cdf(pmf.data(), plen[i] + 1, 16, q.data()) != 0) { printf("cdf failed\n"); return 1; }
        for (int k = 0; k < plen[i] + 2; ++k) cdf[(size_t)i * stride + k] = (int32_t)q[k];
        sizes[i] = plen[i] + 2; offsets[i] = -center[i];
    }
    // symbols: indexes as the synthetic codec produces them (histogram 0..27, mostly small), value ~ N(0, scale)
    std::vector<int32_t> sym((size_t)NSTREAM * NSYM), idx32((size_t)NSTREAM * NSYM);
    std::vector<uint8_t> idx8((size_t)NSTREAM * NSYM);
    unsigned s = 7;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0 / 16777216.0); };
    for (size_t i = 0; i < sym.size(); ++i) {
        const int ci = std::min(27, (int)(-std::log(1.0 - rnd() * 0.999) * 6.0));
        const double g = std::sqrt(-2.0 * std::log(rnd() + 1e-12)) * std::cos(6.283185307 * rnd());
        idx32[i] = ci; idx8[i] = (uint8_t)ci; sym[i] = (int32_t)std::lrint(g * st[ci]);
    }
    const size_t bound = pc_rans_bound(NSYM);
    std::vector<uint8_t> enc((size_t)NSTREAM * bound);
    std::vector<size_t> lens(NSTREAM);
    if (pc_rans_encode_batch(sym.data(), idx32.data(), NSTREAM, NSYM, cdf.data(), NROW, stride, sizes.data(), offsets.data(), enc.data(), bound, lens.data(), 0) != 0) { printf("encode failed\n"); return 1; }
    size_t total = 0;
    std::vector<uint32_t> off(NSTREAM), len32(NSTREAM);
    for (int t = 0; t < NSTREAM; ++t) { off[t] = (uint32_t)(t * bound); len32[t] = (uint32_t)lens[t]; total += lens[t]; }
    // start table of the host fast path: per row and cf >> 8, the last s with cdf[s] <= bucket start, and the first s with cdf[s] > bucket end
    std::vector<uint16_t> lut((size_t)2 * NROW * 256);
    for (int r = 0; r < NROW; ++r)
        for (int b = 0; b < 256; ++b) {
            const int32_t* c = cdf.data() + (size_t)r * stride;
            int lo = 0; while (lo + 1 < sizes[r] - 1 && (uint32_t)c[lo + 1] <= (uint32_t)(b << 8)) ++lo;
            int hi = lo; while (hi < sizes[r] - 1 && (uint32_t)c[hi] <= (uint32_t)((b << 8) | 255)) ++hi;
            lut[r * 256 + b] = (uint16_t)lo; lut[16384 + r * 256 + b] = (uint16_t)std::min(hi, sizes[r] - 1);
        }
    // host reference timing
    std::vector<const uint8_t*> ptrs(NSTREAM);
    for (int t = 0; t < NSTREAM; ++t) ptrs[t] = enc.data() + off[t];
    std::vector<int32_t> dec((size_t)NSTREAM * NSYM);
    auto t0 = std::chrono::steady_clock::now();
    const int HREP = 20;
    for (int r = 0; r < HREP; ++r)
        if (pc_rans_decode_batch_u8(ptrs.data(), lens.data(), NSTREAM, idx8.data(), NSYM, cdf.data(), NROW, stride, sizes.data(), offsets.data(), dec.data(), 0) != 0) { printf("host decode failed\n"); return 1; }
    const double host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / HREP;
    if (dec != sym) { printf("host decode mismatch\n"); return 1; }
    // device
    uint8_t *d_bytes, *d_idx; uint32_t *d_off, *d_len; int32_t *d_cdf, *d_sizes, *d_offsets, *d_out; uint16_t* d_lut; int* d_err;
    (void)hipMalloc(&d_bytes, enc.size()); (void)hipMalloc(&d_idx, idx8.size()); (void)hipMalloc(&d_off, 4 * NSTREAM); (void)hipMalloc(&d_len, 4 * NSTREAM);
    (void)hipMalloc(&d_cdf, cdf.size() * 4); (void)hipMalloc(&d_sizes, 4 * NROW); (void)hipMalloc(&d_offsets, 4 * NROW); (void)hipMalloc(&d_out, sym.size() * 4);
    (void)hipMalloc(&d_lut, lut.size() * 2); (void)hipMalloc(&d_err, 4);
    (void)hipMemcpy(d_bytes, enc.data(), enc.size(), hipMemcpyHostToDevice); (void)hipMemcpy(d_idx, idx8.data(), idx8.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_off, off.data(), 4 * NSTREAM, hipMemcpyHostToDevice); (void)hipMemcpy(d_len, len32.data(), 4 * NSTREAM, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_cdf, cdf.data(), cdf.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(d_sizes, sizes.data(), 4 * NROW, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_offsets, offsets.data(), 4 * NROW, hipMemcpyHostToDevice); (void)hipMemcpy(d_lut, lut.data(), lut.size() * 2, hipMemcpyHostToDevice);
    printf("one slice: %d streams x %d symbols, %.1f KB of strings, tables %d rows x %d\n", NSTREAM, NSYM, total / 1024.0, NROW, stride);
    printf("host  pc_rans_decode_batch_u8 (pool threads): %8.1f us per slice\n", host_us);
    for (int variant = 0; variant < 2; ++variant) {
        (void)hipMemset(d_err, 0, 4); (void)hipMemset(d_out, 0, sym.size() * 4);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        auto launch = [&] {
            if (variant == 0) hipLaunchKernelGGL(rans_dec<false>, dim3(1), dim3(64), 0, 0, d_bytes, d_off, d_len, d_idx, d_cdf, stride, d_sizes, d_offsets, d_lut, d_out, d_err);
            else hipLaunchKernelGGL(rans_dec<true>, dim3(1), dim3(64), 0, 0, d_bytes, d_off, d_len, d_idx, d_cdf, stride, d_sizes, d_offsets, d_lut, d_out, d_err);
        };
        launch();
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < 5; ++r) launch();
        (void)hipEventRecord(e1, 0);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<int32_t> got(sym.size()); int err = 0;
        (void)hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost);
        printf("device one lane per stream, %-28s %8.1f us per slice  (%s, %d truncated)\n", variant == 0 ? "binary search in the row:" : "256-entry start table per row:",
               ms * 1e3 / 5, got == sym ? "symbols identical" : "MISMATCH", err);
    }
    return 0;
}
