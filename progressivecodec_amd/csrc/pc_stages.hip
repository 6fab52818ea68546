// pc_stages.hip -- the non-GEMM device stages of the codec:
//   * shifted-window attention core                         (layers/win_attention.py:84-115,153-207)
//   * per-image variance-quantile mask threshold            (layers/masking.py:205-223, torch.quantile)
//   * GaussianConditional index + quantise + dequantise     (entropy_models.py:126-165,661-666; CHProg_cnn.py:751-755,819-834)
//   * EntropyBottleneck quantise / dequantise               (entropy_models.py:508-522)
// All are HBM/L2-bound byte or element passes: coalesced 128-B rows in, LDS-transposed
// coalesced rows out (the rANS coder consumes symbols in C,H,W raster order while the
// network runs NHWC), wavefront-wide (64-lane) reductions, no atomics on floats.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pc_math.h"
#include "pc_device.h"

namespace {

// ------------------------------------------------------------------------------------------
// window attention: one 64-lane wave handles 64/T (window, head) pairs, lane = query token.
// s_j = fmaf-chain_e (q[e]*scale) * k_j[e];  += bias;  += shift mask;  softmax;  o[e] = fmaf-chain_j p_j v_j[e]
// (same chains as oracle/pc_oracle.c:orc_win_attention)
// ------------------------------------------------------------------------------------------
typedef float f32x2v __attribute__((ext_vector_type(2)));

template <int WS, int D>
__global__ __launch_bounds__(256) void win_attention_kernel(const float* __restrict__ qkv, const float* __restrict__ bias,
                                                            int B, int H, int W, int C, int heads, int shift, float scale,
                                                            float* __restrict__ out, int npairs, int bias_ji)
{
    constexpr int T = WS * WS, G = 64 / T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / T, i = lane % T;
    const int pair = (blockIdx.x * 4 + wave) * G + sub;
    if (pair >= npairs) return;
    const int win = pair / heads, h = pair % heads;
    const int nwx = W / WS, nwy = H / WS;
    const int b = win / (nwx * nwy), wy = (win / nwx) % nwy, wx = win % nwx;
    const int C3 = 3 * C;

    auto token = [&](int t, int& reg) -> int64_t {
        const int ys = wy * WS + t / WS, xs = wx * WS + t % WS;
        const int y = (ys + shift) % H, x = (xs + shift) % W;
        const int ry = ys < H - WS ? 0 : (ys < H - shift ? 1 : 2);
        const int rx = xs < W - WS ? 0 : (xs < W - shift ? 1 : 2);
        reg = ry * 3 + rx;
        return ((int64_t)b * H + y) * W + x;
    };

    // G == 1 (8x8 windows): the wave's 64 lanes are the 64 tokens of ONE (window, head) pair.  Every lane needs every token's k and
    // v: staged once in LDS (each lane writes its own token's two vectors) and read back as broadcasts, instead of 2 x 64 x D/4
    // same-address vector loads per lane from L1.  Same fmaf chains, same results.
    // Round 4: on the G == 1 path the two fmaf loops run on PACKED f32 instructions (v_pk_fma_f32: two independent chains per instruction
    // -- scores of two neighbouring keys j, j+1 in QK^T; outputs of two neighbouring channels e, e+1 in PV).  Per component the same fmaf
    // on the same operands in the same order: the bits do not move (the bit-exact suites are the check).  For the key pairs K sits in LDS
    // TRANSPOSED ([e][j]: a broadcast float4 read gives four consecutive keys of one channel); V stays [j][e].
    constexpr int LDK = D + 4;                                 // padded row: 16-byte aligned, spreads the banks
    __shared__ float kv_lds[G == 1 ? 4 * (T * D + T * LDK) : 1];
    float* k_lds = kv_lds + (G == 1 ? wave * (T * D + T * LDK) : 0);     // G == 1: [D][T]
    float* v_lds = k_lds + (G == 1 ? T * D : 0);                          //         [T][LDK]

    int reg_i;
    const int64_t pix_i = token(i, reg_i);
    if (G == 1) {
        const float4* kp = reinterpret_cast<const float4*>(qkv + pix_i * C3 + C + h * D);
        const float4* vp = reinterpret_cast<const float4*>(qkv + pix_i * C3 + 2 * C + h * D);
#pragma unroll
        for (int e = 0; e < D / 4; ++e) {
            const float4 kk = kp[e];
            k_lds[(4 * e + 0) * T + i] = kk.x; k_lds[(4 * e + 1) * T + i] = kk.y; k_lds[(4 * e + 2) * T + i] = kk.z; k_lds[(4 * e + 3) * T + i] = kk.w;
            *reinterpret_cast<float4*>(v_lds + i * LDK + 4 * e) = vp[e];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float q[D];
    {
        const float4* qp = reinterpret_cast<const float4*>(qkv + pix_i * C3 + h * D);
#pragma unroll
        for (int e = 0; e < D / 4; ++e) {
            const float4 v = qp[e];
            q[4 * e + 0] = v.x * scale; q[4 * e + 1] = v.y * scale; q[4 * e + 2] = v.z * scale; q[4 * e + 3] = v.w * scale;
        }
    }
    float s[T];
    float m = -INFINITY;
    if (G == 1) {
        // all T chains advance together, channel by channel (each chain still visits e = 0, 1, ... in order), two keys per instruction
        f32x2v s2[T / 2];
#pragma unroll
        for (int j = 0; j < T / 2; ++j) s2[j] = f32x2v{0.0f, 0.0f};
#pragma unroll
        for (int e = 0; e < D; ++e) {
            const f32x2v qe = {q[e], q[e]};
#pragma unroll
            for (int jq = 0; jq < T / 4; ++jq) {
                const float4 kk = *reinterpret_cast<const float4*>(k_lds + e * T + 4 * jq);
                s2[2 * jq] = __builtin_elementwise_fma(qe, f32x2v{kk.x, kk.y}, s2[2 * jq]);
                s2[2 * jq + 1] = __builtin_elementwise_fma(qe, f32x2v{kk.z, kk.w}, s2[2 * jq + 1]);
            }
        }
#pragma unroll
        for (int j = 0; j < T / 2; ++j) { s[2 * j] = s2[j].x; s[2 * j + 1] = s2[j].y; }
    }
#pragma unroll
    for (int j = 0; j < T; ++j) {
        int reg_j;
        const int64_t pix_j = token(j, reg_j);
        float acc = 0.0f;
        if (G == 1) acc = s[j];
        else {
            const float4* kp = reinterpret_cast<const float4*>(qkv + pix_j * C3 + C + h * D);
#pragma unroll
            for (int e = 0; e < D / 4; ++e) {
                const float4 k = kp[e];
                acc = fmaf(q[4 * e + 0], k.x, acc);
                acc = fmaf(q[4 * e + 1], k.y, acc);
                acc = fmaf(q[4 * e + 2], k.z, acc);
                acc = fmaf(q[4 * e + 3], k.w, acc);
            }
        }
        // bias_ji: the table is stored [head][j][i] -- for a fixed key j the wave's 64 query lanes read 64 consecutive floats (one 256-byte
        // request); in the module's own [head][i][j] order every lane reads its own row and a load instruction touches 64 cache lines
        acc = acc + (bias_ji ? bias[((int64_t)h * T + j) * T + i] : bias[((int64_t)h * T + i) * T + j]);
        if (shift > 0) acc = acc + (reg_i != reg_j ? -100.0f : 0.0f);
        s[j] = acc;
        m = acc > m ? acc : m;
    }
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < T; ++j) { s[j] = pc_expf(s[j] - m); sum = sum + s[j]; }
#pragma unroll
    for (int j = 0; j < T; ++j) s[j] = s[j] / sum;
    float o[D];
#pragma unroll
    for (int e = 0; e < D; ++e) o[e] = 0.0f;
    if (G == 1) {
        f32x2v o2[D / 2];
#pragma unroll
        for (int e = 0; e < D / 2; ++e) o2[e] = f32x2v{0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const f32x2v pj = {s[j], s[j]};
            const float4* vp = reinterpret_cast<const float4*>(v_lds + j * LDK);
#pragma unroll
            for (int e = 0; e < D / 4; ++e) {
                const float4 v = vp[e];
                o2[2 * e] = __builtin_elementwise_fma(pj, f32x2v{v.x, v.y}, o2[2 * e]);
                o2[2 * e + 1] = __builtin_elementwise_fma(pj, f32x2v{v.z, v.w}, o2[2 * e + 1]);
            }
        }
#pragma unroll
        for (int e = 0; e < D / 2; ++e) { o[2 * e] = o2[e].x; o[2 * e + 1] = o2[e].y; }
    } else {
#pragma unroll
        for (int j = 0; j < T; ++j) {
            int reg_j;
            const int64_t pix_j = token(j, reg_j);
            const float4* vp = reinterpret_cast<const float4*>(qkv + pix_j * C3 + 2 * C + h * D);
#pragma unroll
            for (int e = 0; e < D / 4; ++e) {
                const float4 v = vp[e];
                o[4 * e + 0] = fmaf(s[j], v.x, o[4 * e + 0]);
                o[4 * e + 1] = fmaf(s[j], v.y, o[4 * e + 1]);
                o[4 * e + 2] = fmaf(s[j], v.z, o[4 * e + 2]);
                o[4 * e + 3] = fmaf(s[j], v.w, o[4 * e + 3]);
            }
        }
    }
    float4* op = reinterpret_cast<float4*>(out + pix_i * C + h * D);
#pragma unroll
    for (int e = 0; e < D / 4; ++e) op[e] = make_float4(o[4 * e], o[4 * e + 1], o[4 * e + 2], o[4 * e + 3]);
}

// ------------------------------------------------------------------------------------------
// quantile threshold: exact order statistics by MSB-first radix select (4 x 8-bit digits),
// one 1024-thread workgroup per image, integer LDS histograms only (deterministic).
// thr = ATen lerp(s[lo], s[hi], w) with rank = q*(n-1) evaluated in float32 (see oracle orc_quantile).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fkey(float f)
{
    const uint32_t u = pc_f2bits(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // total order: -inf < ... < -0 < +0 < ... < +inf < NaN(+)
}
__device__ __forceinline__ float fkey_inv(uint32_t k)
{
    return pc_bits2f((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// one histogram increment per DISTINCT bin of a wave: the scales of a slice share their exponent, so in the first radix pass nearly
// all lanes hit the same two or three bins and per-lane LDS atomics serialise (65 us per call at Config 2 before this)
__device__ __forceinline__ void hist_add_wave(uint32_t* hist, uint32_t bin, bool valid)
{
    const int lane = threadIdx.x & 63;
    unsigned long long active = __ballot(valid);
    while (active) {
        const int leader = __ffsll((long long)active) - 1;
        const uint32_t b = (uint32_t)__shfl((int)bin, leader);
        const unsigned long long m = __ballot(valid && bin == b);
        if (lane == leader) atomicAdd(&hist[b], (uint32_t)__popcll(m));
        active &= ~m;
    }
}

// One workgroup selects the order statistics lo = floor(q*(n_total-1)) and lo+1 of one image and writes the interpolated threshold.
// EPT > 0: the n_scan <= 1024*EPT keys are read ONCE and live in registers across the radix passes; EPT == 0: any n, re-read per
// pass.  CAND: the keys come from a candidate list (multi-block path below) that holds every element of the histogram bins
// containing the two ranks; less0 elements of the image lie below those bins, nan0 NaNs were seen by the histogram pass.
template <int EPT, bool CAND>
__device__ __forceinline__ void quantile_block(const float* __restrict__ base, int ld, int C, const uint32_t* __restrict__ cand,
                                               int64_t n_scan, int64_t n_total, uint32_t less0, uint32_t nan0, float q, float* out)
{
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sh_prefix, sh_rank, sh_less, sh_eq, sh_nan, sh_min;
    const int tid = threadIdx.x;
    const int64_t n = n_scan;
    auto key_at = [&](int64_t e, uint32_t& nan_acc) -> uint32_t {
        if (CAND) return cand[e];
        const int64_t p = e / C;
        const float f = base[p * ld + (e - p * C)];
        if (f != f) nan_acc++;
        return fkey(f);
    };
    constexpr int NK = EPT > 0 ? EPT : 1;
    uint32_t keys[NK];
    uint32_t nan_local = 0;
    if (EPT > 0) {
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const int64_t e = tid + 1024 * (int64_t)i;
            keys[i] = e < n ? key_at(e, nan_local) : 0u;
        }
    }

    const float rank = q * (float)(n_total - 1);
    const float lo_f = floorf(rank), hi_f = ceilf(rank);
    const uint32_t lo = (uint32_t)lo_f, hi = (uint32_t)hi_f;
    const float w = rank - lo_f;

    if (tid == 0) { sh_prefix = 0; sh_rank = lo - less0; sh_less = less0; sh_nan = nan0; sh_min = 0xffffffffu; }
    uint32_t prefix_mask = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int k = tid; k < 256; k += 1024) hist[k] = 0;
        __syncthreads();
        const uint32_t prefix = sh_prefix;
        if (EPT > 0) {
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                const bool in = tid + 1024 * (int64_t)i < n;
                const uint32_t bin = (keys[i] >> shift) & 255u;
                if (pass == 0) hist_add_wave(hist, bin, in);                 // clustered digits: aggregate per wave
                else if (in && (keys[i] & prefix_mask) == prefix) atomicAdd(&hist[bin], 1u);   // few survivors, spread digits
            }
        } else {
            uint32_t nan_pass = 0;
            for (int64_t e = tid; e < n; e += 1024) {         // large images: plain LDS atomics measured faster than aggregation here
                const uint32_t k = key_at(e, nan_pass);
                if ((k & prefix_mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
            }
            if (pass == 0) nan_local = nan_pass;
        }
        if (pass == 0 && nan_local) atomicAdd(&sh_nan, nan_local);
        __syncthreads();
        if (tid < 64) {
            // digit d = first bin whose running count exceeds the remaining rank: one wave, 4 bins per lane + a wave prefix sum
            const uint32_t r = sh_rank;
            const uint32_t h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const uint32_t mine = h0 + h1 + h2 + h3;
            uint32_t incl = mine;
            for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, off); if (tid >= off) incl += o; }
            const uint32_t excl = incl - mine;
            const unsigned long long hit = __ballot(r < incl);
            if (hit) {
                const int owner = __ffsll((long long)hit) - 1;
                if (tid == owner) {
                    uint32_t rr = r - excl, d = 4 * tid, hv = h0;
                    if (rr >= h0) { rr -= h0; d++; hv = h1; if (rr >= h1) { rr -= h1; d++; hv = h2; if (rr >= h2) { rr -= h2; d++; hv = h3; } } }
                    sh_prefix = prefix | (d << shift);
                    sh_less += r - rr; sh_rank = rr; sh_eq = hv;
                }
            } else if (tid == 0) {                            // cannot happen for rank < n; keep the serial code's fallback
                sh_prefix = prefix | (256u << shift); sh_less += incl; sh_rank = r - incl; sh_eq = hist[255];
            }
        }
        __syncthreads();
        prefix_mask |= 0xffu << shift;
    }
    const uint32_t key_lo = sh_prefix;
    uint32_t key_hi = key_lo;
    if (hi != lo && hi >= sh_less + sh_eq) {
        // next order statistic: the smallest key above key_lo
        uint32_t mn = 0xffffffffu, dummy = 0;
        if (EPT > 0) {
#pragma unroll
            for (int i = 0; i < NK; ++i) { const uint32_t k = keys[i]; if (tid + 1024 * (int64_t)i < n && k > key_lo && k < mn) mn = k; }
        } else {
            for (int64_t e = tid; e < n; e += 1024) { const uint32_t k = key_at(e, dummy); if (k > key_lo && k < mn) mn = k; }
        }
        for (int off = 32; off; off >>= 1) { const uint32_t o = __shfl_xor(mn, off); mn = o < mn ? o : mn; }
        if ((tid & 63) == 0) atomicMin(&sh_min, mn);
        __syncthreads();
        key_hi = sh_min;
    }
    if (tid == 0) {
        float r;
        if (sh_nan) r = pc_bits2f(0x7fc00000u);
        else {
            const float a = fkey_inv(key_lo), bb = fkey_inv(key_hi);
            const float d = bb - a;
            r = (fabsf(w) < 0.5f) ? fmaf(w, d, a) : fmaf(-d, 1.0f - w, bb);
        }
        *out = r;
    }
}

template <int EPT>
__global__ __launch_bounds__(1024) void quantile_thr_kernel(const float* __restrict__ scale, int ld, int HW, int C, float q,
                                                           float* __restrict__ thr, int64_t sb)
{
    const int b = blockIdx.x;
    const int64_t n = (int64_t)HW * C;
    quantile_block<EPT, false>(scale + (int64_t)b * sb, ld, C, nullptr, n, n, 0u, 0u, q, thr + b);
}

// ---- multi-block path for large images (n > 32768): the single-workgroup kernel above reads an image five times from one CU.
// Here the image is read ONCE, spread over the chip:
//  (1) one workgroup per image reads 4096 sample keys (256 runs of 16 consecutive elements) and picks, by a 12 + 8 bit radix
//      select on the samples, a key bracket [a, b] that holds the two wanted ranks with high probability (sample ranks q*m -/+ a
//      margin of several standard deviations);
//  (2) G workgroups per image stream the image with 16-byte loads, count the keys below a and the NaNs, and append the keys inside
//      [a, b] (a few percent of the image) to a candidate list -- one global atomic per workgroup and 16 K elements;
//  (3) one workgroup runs the exact register-resident select on the candidates.  The sample only steers: (3) checks that both ranks
//      fall inside the candidates (less <= lo, hi < less + count, count <= capacity) and otherwise -- a bad bracket, or more ties at
//      the quantile than the list holds -- falls back to the generic select over the whole image, so the result is always the exact
//      order statistic.
// work (uint32 per image, stride PC_QW_STRIDE): nan count, candidate count, key a, key b, less, pad[3], then PC_QW_CAP candidates.
// Round 4 built the three steps as ONE launch (blocks with ticketed ids, the image's producer block
// publishing the bracket, the last block of an image selecting) and measured it on the Config-4 slice: 0.26-0.43 ms against the 0.055 ms of
// the three launches below, results identical.  On an 8-XCD part every block-to-block hand-over inside a kernel is a device-scope
// release / acquire -- an L2 write-back / invalidate per wave or per block (the XCDs' L2s are not coherent with each other) -- and 2048
// blocks x those cost five times what the two kernel boundaries cost, which do the same once for the whole chip.  Rejected by
// measurement: profiles/r04_d_quantile_one_launch_probe.log (the kernel: pc_stages.hip at commit 8ee4ee6).
#define PC_QW_CAP 32768
#define PC_QW_HDR 8
#define PC_QW_STRIDE (PC_QW_HDR + PC_QW_CAP)
#define PC_QW_M 4096
#define PC_QW_STEP 4096

// rank r (0-based) of the 4096 sample keys held 16 per thread by 256 threads -> the 20-bit key prefix (key >> 12) that holds it
__device__ __forceinline__ uint32_t sample_prefix(const uint32_t (&k)[16], uint32_t r, uint32_t* hist /* 4096 */, uint32_t* sh /* 4 */)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t prefix = 0, mask = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const int bits = pass == 0 ? 12 : 8, shift = pass == 0 ? 20 : 12, nb = 1 << bits, per = nb / 256;
        for (int i = tid; i < nb; i += 256) hist[i] = 0;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) if ((k[i] & mask) == prefix) atomicAdd(&hist[(k[i] >> shift) & (nb - 1)], 1u);
        __syncthreads();
        uint32_t mine = 0;
        for (int i = 0; i < per; ++i) mine += hist[per * tid + i];
        uint32_t incl = mine;
        for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, off); if (lane >= off) incl += o; }
        if (lane == 63) sh[wv] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (int i = 0; i < wv; ++i) before += sh[i];
        const uint32_t excl = before + incl - mine;
        __syncthreads();
        if (r >= excl && r < excl + mine) {                     // exactly one thread: the rank lies in its bins
            uint32_t rr = r - excl, d = per * tid;
            for (int i = 0; i < per; ++i) { const uint32_t h = hist[per * tid + i]; if (rr < h) break; rr -= h; d++; }
            sh[0] = prefix | (d << shift); sh[1] = rr;
        }
        __syncthreads();
        prefix = sh[0]; r = sh[1];
        mask |= (uint32_t)(nb - 1) << shift;
        __syncthreads();
    }
    return prefix >> 12;
}

__global__ __launch_bounds__(256) void quantile_sample_kernel(const float* __restrict__ scale, int ld, int HW, int C, float q, int64_t sb,
                                                             uint32_t* __restrict__ work)
{
    __shared__ uint32_t hist[4096];
    __shared__ uint32_t sh[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t n = (int64_t)HW * C;
    const float* base = scale + (int64_t)b * sb;
    const int64_t start = (int64_t)tid * ((n - 16) / 255);
    uint32_t k[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t e = start + i, p = e / C;
        k[i] = fkey(base[p * ld + (e - p * C)]);
    }
    // sample ranks around q * (m - 1): margin = 7 standard deviations of a binomial sample quantile (the runs of 16 are correlated,
    // so count on half the nominal sample size) + 8
    const float m1 = (float)(PC_QW_M - 1);
    const float c = q * m1, sd = sqrtf(fmaxf(q * (1.0f - q), 0.0f) * (float)PC_QW_M);
    const float mg = 7.0f * sd + 8.0f;
    const float fa = floorf(c - mg), fb = ceilf(c + mg);
    uint32_t ka = 0u, kb = 0xffffffffu;
    if (fa > 0.0f) ka = sample_prefix(k, (uint32_t)fa, hist, sh) << 12;
    if (fb < m1) kb = (sample_prefix(k, (uint32_t)fb, hist, sh) << 12) | 0xfffu;
    if (tid == 0) {
        uint32_t* w = work + (size_t)b * PC_QW_STRIDE;
        w[0] = 0; w[1] = 0; w[2] = ka; w[3] = kb; w[4] = 0;
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void quantile_bracket_kernel(const float* __restrict__ scale, int ld, int HW, int C, int64_t sb,
                                                              uint32_t* __restrict__ work)
{
    __shared__ uint32_t sh_take[4], sh_less[4], sh_nan[4], sh_base;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t* w = work + (size_t)b * PC_QW_STRIDE;
    const uint32_t ka = w[2], kb = w[3];
    uint32_t* cand = w + PC_QW_HDR;
    const int64_t n = (int64_t)HW * C;
    const float* base = scale + (int64_t)b * sb;
    // 4 K elements per step: thread t owns elements e0 + 4 * (t + 256 * i) .. + 3, i < 4; all 16-byte loads in flight before the first use
    for (int64_t e0 = (int64_t)blockIdx.x * PC_QW_STEP; e0 < n; e0 += (int64_t)gridDim.x * PC_QW_STEP) {
        uint32_t k[16];
        bool in[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t e = e0 + 4 * (tid + 256 * (int64_t)i);
            if (VEC) {
                const int64_t p = e / C;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < n) v = *reinterpret_cast<const float4*>(base + p * ld + (e - p * C));    // n % 4 == 0: whole quad or nothing
                k[4 * i] = fkey(v.x); k[4 * i + 1] = fkey(v.y); k[4 * i + 2] = fkey(v.z); k[4 * i + 3] = fkey(v.w);
#pragma unroll
                for (int j = 0; j < 4; ++j) in[4 * i + j] = e < n;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t ee = e + j, p = ee / C;
                    in[4 * i + j] = ee < n;
                    k[4 * i + j] = in[4 * i + j] ? fkey(base[p * ld + (ee - p * C)]) : 0u;
                }
            }
        }
        uint32_t take = 0, less = 0, nan = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            take += in[i] && k[i] >= ka && k[i] <= kb;
            less += in[i] && k[i] < ka;
            nan += in[i] && (k[i] > 0xff800000u || k[i] < 0x007fffffu);      // keys of +NaN lie above +inf's, of -NaN below -inf's
        }
        uint32_t incl = take;
        for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, off); if (lane >= off) incl += o; }
        for (int off = 32; off; off >>= 1) { less += (uint32_t)__shfl_xor((int)less, off); nan += (uint32_t)__shfl_xor((int)nan, off); }
        if (lane == 63) sh_take[wv] = incl;
        if (lane == 0) { sh_less[wv] = less; sh_nan[wv] = nan; }
        __syncthreads();
        if (tid == 0) {
            const uint32_t tt = sh_take[0] + sh_take[1] + sh_take[2] + sh_take[3];
            const uint32_t tl = sh_less[0] + sh_less[1] + sh_less[2] + sh_less[3], tn = sh_nan[0] + sh_nan[1] + sh_nan[2] + sh_nan[3];
            sh_base = tt ? atomicAdd(&w[1], tt) : 0u;
            if (tl) atomicAdd(&w[4], tl);
            if (tn) atomicAdd(&w[0], tn);
        }
        uint32_t before = 0;
        for (int i = 0; i < wv; ++i) before += sh_take[i];
        __syncthreads();
        uint32_t pos = sh_base + before + incl - take;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (in[i] && k[i] >= ka && k[i] <= kb) { if (pos < PC_QW_CAP) cand[pos] = k[i]; pos++; }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void quantile_final_kernel(const float* __restrict__ scale, int ld, int HW, int C, float q,
                                                             float* __restrict__ thr, int64_t sb, const uint32_t* __restrict__ work)
{
    const int b = blockIdx.x;
    const uint32_t* w = work + (size_t)b * PC_QW_STRIDE;
    const int64_t n = (int64_t)HW * C;
    const uint32_t cnt = w[1], less = w[4];
    const float rank = q * (float)(n - 1);
    const uint32_t lo = (uint32_t)floorf(rank), hi = (uint32_t)ceilf(rank);
    const bool ok = less <= lo && hi < less + cnt;
    if (ok && cnt <= 8192) quantile_block<8, true>(nullptr, 0, 1, w + PC_QW_HDR, cnt, n, less, w[0], q, thr + b);
    else if (ok && cnt <= PC_QW_CAP) quantile_block<32, true>(nullptr, 0, 1, w + PC_QW_HDR, cnt, n, less, w[0], q, thr + b);
    else quantile_block<0, false>(scale + (int64_t)b * sb, ld, C, nullptr, n, n, 0u, 0u, q, thr + b);
}

// ------------------------------------------------------------------------------------------
// GaussianConditional stages.  Tile = 64 pixels x 32 channels; NHWC in, [B][C][HW] out via LDS transpose.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int gc_index(float s, const float* table, int nt, float bound)
{
    // entropy_models.py:661-666: idx = (nt-1) - #{k < nt-1 : max(s, bound) <= table[k]}  ==  #{k < nt-1 : table[k] < max(s, bound)}
    if (s != s) return nt - 1;                 // torch.max propagates NaN; every compare is then false
    s = s > bound ? s : bound;
    int lo = 0, hi = nt - 1;                   // first k in [0, nt-1) with table[k] >= s
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (table[mid] < s) lo = mid + 1; else hi = mid; }
    return lo;
}
// the same count for a table held as 64 LDS words, entries from nt-1 on replaced by +inf (pc_fill_table64): six uniform steps, no
// data-dependent loop, so the searches of a thread's elements overlap their LDS latencies.  Ascending tables only (as gc_index).
__device__ __forceinline__ int gc_index64(float s, const float* t64, int nt, float bound)
{
    const bool isnan = s != s;
    s = s > bound ? s : bound;
    int lo = 0;
#pragma unroll
    for (int step = 32; step; step >>= 1) lo += (t64[lo + step - 1] < s) ? step : 0;      // lo + step - 1 <= 62
    return isnan ? nt - 1 : lo;
}
__device__ __forceinline__ void pc_fill_table64(float* t64, const float* table, int nt)
{
    const int tid = threadIdx.x;
    if (tid < 64) t64[tid] = tid < nt - 1 ? table[tid] : __builtin_inff();
}

// GaussianConditional._likelihood of an already rounded value (entropy_models.py:626-643, :578-582), the way the reference's eval
// path evaluates it (forward_single_quality, CHProg_cnn.py:1050,1150): values = |outputs - means| (base slices: the float32
// expression (round(y-mu)+mu)-mu, not exactly the symbol) or |round(.)| (enhancement slices, no means), scales lower-bounded,
// upper/lower = 0.5 * erfc(-(2**-0.5) * ((+-0.5 - values) / scales)) in float32, likelihood = max(upper - lower, 1e-9).
// erfc is evaluated in double on its float32 argument and rounded (correctly rounded float erfc up to double rounding); not part
// of the bitstream, so no cross-device contract is needed here: tolerance-tested against torch.erfc.
__device__ __forceinline__ float gc_likelihood(float values, float s_eff)
{
    const float a = fabsf(values);
    const float cst = -0.70710678118654752440f;
    const float u = (0.5f - a) / s_eff, l = (-0.5f - a) / s_eff;
    const float up = 0.5f * (float)erfc((double)(cst * u));
    const float lo = 0.5f * (float)erfc((double)(cst * l));
    const float lik = up - lo;
    return lik > 1e-9f ? lik : 1e-9f;
}

template <int MODE>   // 0 = encoder (index+quantise+dequantise), 1 = decoder index only
__global__ __launch_bounds__(256) void gc_prep_kernel(const pc_prep_params p)
{
    constexpr int TP = 64, CC = 32;
    __shared__ int32_t t_sym[CC][TP + 1];
    __shared__ int32_t t_idx[CC][TP + 1];
    __shared__ float t_msk[CC][TP + 1];
    __shared__ float t_lik[MODE == 0 ? CC : 1][TP + 1];
    __shared__ float s_table[64];
    const int b = blockIdx.y, p0 = blockIdx.x * TP, tid = threadIdx.x;
    if (tid < p.ntable && tid < 64) s_table[tid] = p.table[tid];
    __syncthreads();
    float thr = 0.0f;
    if (p.mask_mode == 1) thr = p.thr[b];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int e = tid + 256 * k;
        const int px = e >> 5, c = e & 31;
        if (p0 + px >= p.HW) continue;
        const int64_t pix = (int64_t)b * p.HW + p0 + px;
        const float s = p.scale[pix * p.ld_scale + c];
        float m = 1.0f;
        if (p.mask_mode == 1) {
            const float mv = p.mask_src ? p.mask_src[(int64_t)b * p.mask_sb + (int64_t)c * p.HW + p0 + px] : s;
            m = (mv >= thr) ? 1.0f : 0.0f;
        }
        else if (p.mask_mode == 3) m = 0.0f;
        const float sm = (p.mask_mode == 0) ? s : s * m;
        t_idx[c][px] = gc_index(sm, s_table, p.ntable, p.bound);
        t_msk[c][px] = m;
        if (MODE == 0) {
            const float mu = p.mu[pix * p.ld_mu + c];
            float y = p.y[pix * p.ld_y + c];
            if (p.ybase) y = y - p.ybase[pix * p.ld_ybase + c];          // delta_encode, CHProg_cnn.py:780-781
            float v = y - mu;                                              // entropy_models.py:137-139 / CHProg_cnn.py:830
            if (p.mask_mode != 0) v = v * m;
            const int32_t sym = (int32_t)pc_roundevenf(v);
            t_sym[c][px] = sym;
            p.yhat[pix * p.ld_yhat + c] = (float)sym + mu;                 // CHProg_cnn.py:754-755,833-834
            if (p.lik) {                                                   // lower_bound_scale, then _likelihood
                const float values = (p.mask_mode == 0) ? ((float)sym + mu) - mu : (float)sym;
                t_lik[c][px] = gc_likelihood(values, sm > p.bound ? sm : p.bound);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int e = tid + 256 * k;
        const int c = e >> 6, px = e & 63;
        if (p0 + px >= p.HW) continue;
        const int64_t o = ((int64_t)b * CC + c) * p.HW + p0 + px;
        p.idx[o] = t_idx[c][px];
        if (p.idx8) p.idx8[o] = (uint8_t)t_idx[c][px];
        if (MODE == 0) p.sym[o] = t_sym[c][px];
        if (p.mask) p.mask[o] = t_msk[c][px];
        if (MODE == 0 && p.lik) p.lik[(int64_t)b * p.lik_sb + (int64_t)c * p.HW + p0 + px] = t_lik[c][px];
    }
}

// decoder: yhat(NHWC) = float(sym[B][C][HW]) + mu(NHWC)          entropy_models.py:159-165, CHProg_cnn.py:896,971
__global__ __launch_bounds__(256) void gc_dequant_kernel(const pc_prep_params p)
{
    constexpr int TP = 64, CC = 32;
    __shared__ int32_t t_sym[CC][TP + 1];
    const int b = blockIdx.y, p0 = blockIdx.x * TP, tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int e = tid + 256 * k;
        const int c = e >> 6, px = e & 63;
        if (p0 + px < p.HW) t_sym[c][px] = p.sym[((int64_t)b * CC + c) * p.HW + p0 + px];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int e = tid + 256 * k;
        const int px = e >> 5, c = e & 31;
        if (p0 + px >= p.HW) continue;
        const int64_t pix = (int64_t)b * p.HW + p0 + px;
        p.yhat[pix * p.ld_yhat + c] = (float)t_sym[c][px] + p.mu[pix * p.ld_mu + c];
    }
}

// ---- 16-byte forms of the two kernels above (same arithmetic per element, same tile): a lane moves four channels of a pixel on
// the NHWC side and four pixels of a channel on the plane side, so every global access is a dwordx4 and a wave's loads are all in
// flight before the first use.  Needs 16-byte aligned rows (pointers and pixel strides), HW % 4 == 0 and no mask_src; the
// launchers fall back to the scalar kernels otherwise.  LDS rows of 68 words keep the plane-side b128 reads aligned.
#define PC_LS 68
template <int MODE, bool LIK>
__global__ __launch_bounds__(256) void gc_prep_vec_kernel(const pc_prep_params p)
{
    constexpr int TP = 64, CC = 32;
    __shared__ __attribute__((aligned(16))) int32_t t_sym[MODE == 0 ? CC : 1][PC_LS];
    __shared__ __attribute__((aligned(16))) int32_t t_idx[CC][PC_LS];
    __shared__ __attribute__((aligned(16))) float t_msk[CC][PC_LS];
    __shared__ __attribute__((aligned(16))) float t_lik[LIK ? CC : 1][PC_LS];
    __shared__ float s_table[64];
    const int b = blockIdx.y, p0 = blockIdx.x * TP, tid = threadIdx.x;
    pc_fill_table64(s_table, p.table, p.ntable);
    float thr = 0.0f;
    if (p.mask_mode == 1) thr = p.thr[b];
    float4 s4[2], mu4[2], y4[2], yb4[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int u = tid + 256 * k, px = u >> 3, c0 = (u & 7) * 4;
        s4[k] = mu4[k] = y4[k] = yb4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p0 + px < p.HW) {
            const int64_t pix = (int64_t)b * p.HW + p0 + px;
            s4[k] = *reinterpret_cast<const float4*>(p.scale + pix * p.ld_scale + c0);
            if (MODE == 0) {
                mu4[k] = *reinterpret_cast<const float4*>(p.mu + pix * p.ld_mu + c0);
                y4[k] = *reinterpret_cast<const float4*>(p.y + pix * p.ld_y + c0);
                if (p.ybase) yb4[k] = *reinterpret_cast<const float4*>(p.ybase + pix * p.ld_ybase + c0);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int u = tid + 256 * k, px = u >> 3, c0 = (u & 7) * 4;
        if (p0 + px >= p.HW) continue;
        const int64_t pix = (int64_t)b * p.HW + p0 + px;
        const float sv[4] = {s4[k].x, s4[k].y, s4[k].z, s4[k].w};
        const float muv[4] = {mu4[k].x, mu4[k].y, mu4[k].z, mu4[k].w};
        const float yv[4] = {y4[k].x, y4[k].y, y4[k].z, y4[k].w};
        const float ybv[4] = {yb4[k].x, yb4[k].y, yb4[k].z, yb4[k].w};
        float yh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j;
            const float s = sv[j];
            float m = 1.0f;
            if (p.mask_mode == 1) m = (s >= thr) ? 1.0f : 0.0f;
            else if (p.mask_mode == 3) m = 0.0f;
            const float sm = (p.mask_mode == 0) ? s : s * m;
            t_idx[c][px] = gc_index64(sm, s_table, p.ntable, p.bound);
            t_msk[c][px] = m;
            if (MODE == 0) {
                const float mu = muv[j];
                float y = yv[j];
                if (p.ybase) y = y - ybv[j];
                float v = y - mu;
                if (p.mask_mode != 0) v = v * m;
                const int32_t sym = (int32_t)pc_roundevenf(v);
                t_sym[c][px] = sym;
                yh[j] = (float)sym + mu;
                if (LIK) {
                    const float values = (p.mask_mode == 0) ? ((float)sym + mu) - mu : (float)sym;
                    t_lik[c][px] = gc_likelihood(values, sm > p.bound ? sm : p.bound);
                }
            }
        }
        if (MODE == 0) *reinterpret_cast<float4*>(p.yhat + pix * p.ld_yhat + c0) = make_float4(yh[0], yh[1], yh[2], yh[3]);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int u = tid + 256 * k, c = u >> 4, q = (u & 15) * 4;
        if (p0 + q >= p.HW) continue;
        const int64_t o = ((int64_t)b * CC + c) * p.HW + p0 + q;
        const int4 iv = *reinterpret_cast<const int4*>(&t_idx[c][q]);
        *reinterpret_cast<int4*>(p.idx + o) = iv;
        if (p.idx8) *reinterpret_cast<uchar4*>(p.idx8 + o) = make_uchar4((uint8_t)iv.x, (uint8_t)iv.y, (uint8_t)iv.z, (uint8_t)iv.w);
        if (MODE == 0) *reinterpret_cast<int4*>(p.sym + o) = *reinterpret_cast<const int4*>(&t_sym[c][q]);
        if (p.mask) *reinterpret_cast<float4*>(p.mask + o) = *reinterpret_cast<const float4*>(&t_msk[c][q]);
        if (LIK) *reinterpret_cast<float4*>(p.lik + (int64_t)b * p.lik_sb + (int64_t)c * p.HW + p0 + q) = *reinterpret_cast<const float4*>(&t_lik[c][q]);
    }
}

__global__ __launch_bounds__(256) void gc_dequant_vec_kernel(const pc_prep_params p)
{
    constexpr int TP = 64, CC = 32;
    __shared__ __attribute__((aligned(16))) int32_t t_sym[CC][PC_LS];
    const int b = blockIdx.y, p0 = blockIdx.x * TP, tid = threadIdx.x;
    float4 mu4[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int u = tid + 256 * k, c = u >> 4, q = (u & 15) * 4;
        if (p0 + q < p.HW) *reinterpret_cast<int4*>(&t_sym[c][q]) = *reinterpret_cast<const int4*>(p.sym + ((int64_t)b * CC + c) * p.HW + p0 + q);
        const int px = u >> 3, c0 = (u & 7) * 4;
        mu4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p0 + px < p.HW) mu4[k] = *reinterpret_cast<const float4*>(p.mu + ((int64_t)b * p.HW + p0 + px) * p.ld_mu + c0);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int u = tid + 256 * k, px = u >> 3, c0 = (u & 7) * 4;
        if (p0 + px >= p.HW) continue;
        const int64_t pix = (int64_t)b * p.HW + p0 + px;
        *reinterpret_cast<float4*>(p.yhat + pix * p.ld_yhat + c0) =
            make_float4((float)t_sym[c0][px] + mu4[k].x, (float)t_sym[c0 + 1][px] + mu4[k].y, (float)t_sym[c0 + 2][px] + mu4[k].z,
                        (float)t_sym[c0 + 3][px] + mu4[k].w);
    }
}

static inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }
static bool prep_vec_ok(const pc_prep_params& p, int mode)   // mode 0 encoder, 1 decoder index, 2 dequantise
{
    static const bool off = pc_tune("PC_PREP_SCALAR", 0) != 0;
    if (off || (p.HW & 3) || p.mask_src) return false;
    bool ok = true;
    if (mode != 2) ok = ok && al16(p.scale) && !(p.ld_scale & 3) && al16(p.idx) && al16(p.mask) && (!p.idx8 || !(reinterpret_cast<uintptr_t>(p.idx8) & 3u));
    if (mode != 1) ok = ok && al16(p.mu) && !(p.ld_mu & 3) && al16(p.yhat) && !(p.ld_yhat & 3) && al16(p.sym);
    if (mode == 0) ok = ok && al16(p.y) && !(p.ld_y & 3) && al16(p.ybase) && !(p.ld_ybase & 3) && al16(p.lik) && !(p.lik_sb & 3);
    return ok;
}

// EntropyBottleneck: z NHWC [B][HW][C] -> sym [B][C][HW] (= round(z - median_c)), zhat NHWC = float(sym) + median_c
__global__ void eb_quant_kernel(const float* __restrict__ z, int B, int HW, int C, const float* __restrict__ med,
                                int32_t* __restrict__ sym, float* __restrict__ zhat)
{
    const int64_t n = (int64_t)B * HW * C;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const int64_t pix = e / C;
        const int b = (int)(pix / HW), px = (int)(pix % HW);
        const float m = med[c];
        const int32_t s = (int32_t)pc_roundevenf(z[e] - m);
        sym[((int64_t)b * C + c) * HW + px] = s;
        zhat[e] = (float)s + m;
    }
}

__global__ void eb_dequant_kernel(const int32_t* __restrict__ sym, int B, int HW, int C, const float* __restrict__ med,
                                  float* __restrict__ zhat)
{
    const int64_t n = (int64_t)B * HW * C;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const int64_t pix = e / C;
        const int b = (int)(pix / HW), px = (int)(pix % HW);
        zhat[e] = (float)sym[((int64_t)b * C + c) * HW + px] + med[c];
    }
}

// REM: refined scale = ret * att + scale, att = round(star - bar) with both masks taken on the unrefined scale (CHProgREM.py:395-401, :84-86).
// mu != null (mu_std=True, :397-398,414-416): ret carries 2N channels per pixel -- mu <- ret[:N] * att + mu, scale <- ret[N:] * att + scale.
__global__ void rem_combine_kernel(const float* __restrict__ ret, int ld_ret, float* __restrict__ scale, int ld_scale, int B, int HW,
                                   const float* __restrict__ thr_star, int mode_star, const float* __restrict__ thr_bar, int mode_bar,
                                   float* __restrict__ mu, int ld_mu)
{
    const int64_t n = (int64_t)B * HW * 32;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e & 31);
        const int64_t pix = e >> 5;
        const int b = (int)(pix / HW);
        const float s = scale[pix * ld_scale + c];
        const float star = mode_star == 1 ? (s >= thr_star[b] ? 1.0f : 0.0f) : (mode_star == 2 ? 1.0f : 0.0f);
        const float bar = mode_bar == 1 ? (s >= thr_bar[b] ? 1.0f : 0.0f) : (mode_bar == 2 ? 1.0f : 0.0f);
        const float att = pc_roundevenf(star - bar);
        if (mu) {
            mu[pix * ld_mu + c] = ret[pix * ld_ret + c] * att + mu[pix * ld_mu + c];
            scale[pix * ld_scale + c] = ret[pix * ld_ret + 32 + c] * att + s;
        } else scale[pix * ld_scale + c] = ret[pix * ld_ret + c] * att + s;
    }
}

// one 32-channel slice of an NCHW tensor -> NHWC [B][HW][C] (the REM's checkpoint representation arrives NCHW)
__global__ void nchw_slice_to_nhwc_kernel(const float* __restrict__ src, int64_t sb, int B, int HW, int C, float* __restrict__ dst)
{
    const int64_t n = (int64_t)B * HW * C;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const int64_t pix = e / C;
        const int b = (int)(pix / HW), hw = (int)(pix - (int64_t)b * HW);
        dst[e] = src[(int64_t)b * sb + (int64_t)c * HW + hw];
    }
}

}  // namespace

#define PC_LAUNCH_CHECK() (hipGetLastError() == hipSuccess ? PC_OK : PC_ERR_HIP)

int pc_win_attention_launch(const float* qkv, const float* bias, int B, int H, int W, int C, int heads, int ws,
                            int shift, float scale, float* out, hipStream_t stream, int bias_ji)
{
    if (heads <= 0 || C % heads || H % ws || W % ws || shift < 0 || shift >= ws) return PC_ERR_ARG;
    const int d = C / heads, T = ws * ws;
    const int npairs = B * (H / ws) * (W / ws) * heads;
    const int per_block = 4 * (64 / T);
    dim3 grid((npairs + per_block - 1) / per_block), block(256);
    if (ws == 8 && d == 24)
        hipLaunchKernelGGL((win_attention_kernel<8, 24>), grid, block, 0, stream, qkv, bias, B, H, W, C, heads, shift, scale, out, npairs, bias_ji);
    else if (ws == 4 && d == 80)
        hipLaunchKernelGGL((win_attention_kernel<4, 80>), grid, block, 0, stream, qkv, bias, B, H, W, C, heads, shift, scale, out, npairs, bias_ji);
    else if (ws == 4 && d == 40)
        hipLaunchKernelGGL((win_attention_kernel<4, 40>), grid, block, 0, stream, qkv, bias, B, H, W, C, heads, shift, scale, out, npairs, bias_ji);
    else
        return PC_ERR_ARG;
    return PC_LAUNCH_CHECK();
}

// EntropyBottleneck._logits_cumulative / _likelihood (entropy_models.py:400-433) on the dequantised hyper-latent.
// net per channel: [sp0 3][b0 3][tf0 3] then 3 x {[sp 3x3][b 3][tf 3]} then [sp4 3][b4 1]; sp = softplus(matrix), tf = tanh(factor)
__device__ __forceinline__ float eb_logits(const float* n, float x)
{
    float l[3], t[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { l[j] = n[j] * x + n[3 + j]; l[j] = l[j] + n[6 + j] * pc_tanhf(l[j]); }
    n += 9;
#pragma unroll
    for (int layer = 0; layer < 3; ++layer) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float a = n[3 * j] * l[0];
            a = a + n[3 * j + 1] * l[1];
            a = a + n[3 * j + 2] * l[2];
            a = a + n[9 + j];
            t[j] = a + n[12 + j] * pc_tanhf(a);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) l[j] = t[j];
        n += 15;
    }
    float o = n[0] * l[0];
    o = o + n[1] * l[1];
    o = o + n[2] * l[2];
    return o + n[3];
}

__global__ void eb_likelihood_kernel(const int32_t* __restrict__ sym, int B, int HW, int C, const float* __restrict__ med,
                                     const float* __restrict__ net, float* __restrict__ lik)
{
    const int64_t n = (int64_t)B * C * HW;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)((e / HW) % C);
        const float v = (float)sym[e] + med[c];                       // quantize(.., "dequantize", medians), entropy_models.py:470-472
        const float* w = net + (size_t)c * PC_EB_NET_FLOATS;
        const float lower = eb_logits(w, v - 0.5f), upper = eb_logits(w, v + 0.5f);
        const float sum = lower + upper;
        const float sign = sum > 0.0f ? -1.0f : (sum < 0.0f ? 1.0f : 0.0f);          // -torch.sign(lower + upper)
        const float l = fabsf(pc_sigmoidf(sign * upper) - pc_sigmoidf(sign * lower));
        lik[e] = l > 1e-9f ? l : 1e-9f;
    }
}

int pc_eb_likelihood_launch(const int32_t* sym, int B, int HW, int C, const float* med, const float* net, float* lik, hipStream_t stream)
{
    const int64_t n = (int64_t)B * HW * C;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(eb_likelihood_kernel, dim3(blocks), dim3(256), 0, stream, sym, B, HW, C, med, net, lik);
    return PC_LAUNCH_CHECK();
}

int pc_quantile_thr_launch(const float* scale, int ld, int B, int HW, int C, float q, float* thr, uint32_t* work, hipStream_t stream, int64_t sb)
{
    if (B <= 0 || HW <= 0 || C <= 0) return PC_ERR_ARG;
    if (sb == 0) sb = (int64_t)HW * ld;
    const int64_t n = (int64_t)HW * C;
    static const bool single = pc_tune("PC_QUANTILE_SINGLE", 0) != 0;
    if (n <= 1024 * 8) hipLaunchKernelGGL(quantile_thr_kernel<8>, dim3(B), dim3(1024), 0, stream, scale, ld, HW, C, q, thr, sb);
    else if (n <= PC_QUANTILE_SMALL_N) hipLaunchKernelGGL(quantile_thr_kernel<32>, dim3(B), dim3(1024), 0, stream, scale, ld, HW, C, q, thr, sb);
    else if (single) hipLaunchKernelGGL(quantile_thr_kernel<0>, dim3(B), dim3(1024), 0, stream, scale, ld, HW, C, q, thr, sb);
    else {
        uint32_t* w = work;
        if (!w && hipMallocAsync(reinterpret_cast<void**>(&w), pc_quantile_work_bytes(B), stream) != hipSuccess) return PC_ERR_HIP;
        const int G = (int)std::min<int64_t>(256, (n + PC_QW_STEP - 1) / PC_QW_STEP);
        const bool vec = !(C & 3) && !(ld & 3) && !(sb & 3) && !(reinterpret_cast<uintptr_t>(scale) & 15u);
        hipLaunchKernelGGL(quantile_sample_kernel, dim3(B), dim3(256), 0, stream, scale, ld, HW, C, q, sb, w);
        if (vec) hipLaunchKernelGGL(quantile_bracket_kernel<true>, dim3(G, B), dim3(256), 0, stream, scale, ld, HW, C, sb, w);
        else hipLaunchKernelGGL(quantile_bracket_kernel<false>, dim3(G, B), dim3(256), 0, stream, scale, ld, HW, C, sb, w);
        hipLaunchKernelGGL(quantile_final_kernel, dim3(B), dim3(1024), 0, stream, scale, ld, HW, C, q, thr, sb, (const uint32_t*)w);
        if (!work && hipFreeAsync(w, stream) != hipSuccess) return PC_ERR_HIP;
    }
    return PC_LAUNCH_CHECK();
}
size_t pc_quantile_work_bytes(int B) { return (size_t)B * PC_QW_STRIDE * sizeof(uint32_t); }

int pc_prep_enc_launch(const pc_prep_params& p, hipStream_t stream)
{
    if (p.C != 32 || p.ntable > 64 || p.ntable < 2) return PC_ERR_ARG;
    const dim3 grid((p.HW + 63) / 64, p.B);
    if (!prep_vec_ok(p, 0)) hipLaunchKernelGGL((gc_prep_kernel<0>), grid, dim3(256), 0, stream, p);
    else if (p.lik) hipLaunchKernelGGL((gc_prep_vec_kernel<0, true>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((gc_prep_vec_kernel<0, false>), grid, dim3(256), 0, stream, p);
    return PC_LAUNCH_CHECK();
}
int pc_prep_dec_index_launch(const pc_prep_params& p, hipStream_t stream)
{
    if (p.C != 32 || p.ntable > 64 || p.ntable < 2) return PC_ERR_ARG;
    const dim3 grid((p.HW + 63) / 64, p.B);
    if (!prep_vec_ok(p, 1)) hipLaunchKernelGGL((gc_prep_kernel<1>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((gc_prep_vec_kernel<1, false>), grid, dim3(256), 0, stream, p);
    return PC_LAUNCH_CHECK();
}
int pc_prep_dec_dequant_launch(const pc_prep_params& p, hipStream_t stream)
{
    if (p.C != 32) return PC_ERR_ARG;
    const dim3 grid((p.HW + 63) / 64, p.B);
    if (!prep_vec_ok(p, 2)) hipLaunchKernelGGL(gc_dequant_kernel, grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(gc_dequant_vec_kernel, grid, dim3(256), 0, stream, p);
    return PC_LAUNCH_CHECK();
}
int pc_eb_quant_launch(const float* z, int B, int HW, int C, const float* med, int32_t* sym, float* zhat, hipStream_t stream)
{
    const int64_t n = (int64_t)B * HW * C;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(eb_quant_kernel, dim3(blocks), dim3(256), 0, stream, z, B, HW, C, med, sym, zhat);
    return PC_LAUNCH_CHECK();
}
int pc_eb_dequant_launch(const int32_t* sym, int B, int HW, int C, const float* med, float* zhat, hipStream_t stream)
{
    const int64_t n = (int64_t)B * HW * C;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(eb_dequant_kernel, dim3(blocks), dim3(256), 0, stream, sym, B, HW, C, med, zhat);
    return PC_LAUNCH_CHECK();
}

int pc_rem_combine_launch(const float* ret, int ld_ret, float* scale, int ld_scale, int B, int HW, const float* thr_star, int mode_star,
                          const float* thr_bar, int mode_bar, hipStream_t stream, float* mu, int ld_mu)
{
    if (!ret || !scale || B <= 0 || HW <= 0 || (mode_star == 1 && !thr_star) || (mode_bar == 1 && !thr_bar)) return PC_ERR_ARG;
    const int64_t n = (int64_t)B * HW * 32;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(rem_combine_kernel, dim3(blocks), dim3(256), 0, stream, ret, ld_ret, scale, ld_scale, B, HW, thr_star, mode_star, thr_bar, mode_bar, mu, ld_mu);
    return PC_LAUNCH_CHECK();
}

int pc_nchw_slice_to_nhwc_launch(const float* src, int64_t batch_stride, int B, int HW, int C, float* dst, hipStream_t stream)
{
    if (!src || !dst || B <= 0 || HW <= 0 || C <= 0) return PC_ERR_ARG;
    const int64_t n = (int64_t)B * HW * C;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(nchw_slice_to_nhwc_kernel, dim3(blocks), dim3(256), 0, stream, src, batch_stride, B, HW, C, dst);
    return PC_LAUNCH_CHECK();
}
