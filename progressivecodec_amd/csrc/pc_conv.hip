// pc_conv.hip -- tap-list implicit-GEMM convolution on the gfx950 f32 MFMA pipe.
//
// One kernel family serves every dense contraction of the codec (reference modules in
// /root/reference/src/compress): nn.Conv2d 5x5 s2 (models/utils.py:186), conv3x3 s1/s2
// and conv1x1 (layers/layers.py:15,27), nn.Linear qkv/proj (layers/win_attention.py:76,78),
// the GDN/IGDN 1x1 contraction over x^2 (layers/gdn.py:56), sub-pixel conv with
// PixelShuffle folded into the store (layers/layers.py:20-24) and the four output phases
// of ConvTranspose2d(5, s2, p2, op1) (models/utils.py:196).
//
//   out[b, i*osy+ooy, j*osx+oox, n] = epi( bias[n] + sum_{t<T} sum_{c<Cin} f(in[b, i*s+dy_t, j*s+dx_t, c]) * w[wtap_t][c][n] )
//
// GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = T*Cin.  Activations are NHWC so a
// K-chunk (one tap x 16 channels) is 64 contiguous bytes per pixel; the channel axis of the
// input may be a *virtual concatenation* of up to 8 segments (no torch.cat copies in the
// 20-step slice chain).  Numeric contract (include/pc_math.h): each output
// element is ONE fmaf chain starting at +0 over the flattened index k = tap*Cin + channel,
// in aligned groups of 8 visited as 0,4,1,5,2,6,3,7 -- what v_mfma_f32_32x32x2_f32 computes
// when lanes 0-31 / 32-63 carry the two 16-byte quads of a group and nothing is split along
// K -- so results are bit-identical for every tile shape, batch size and device, and equal
// to oracle/pc_oracle.c:orc_conv_nhwc.
//
// Two kernels: conv_igemm_uni_kernel (further down; weight layout 1) serves every layer whose Cin is a multiple of 16 and
// Cout > 4 -- all but two; conv_igemm_kernel (next; weight layout 0) is the plain form kept for the 3-channel input layer
// (element gather from NCHW) and as the generic fallback behind pc_conv2d_nhwc: 256 threads = 4 waves, block tile BM x BN,
// K-chunk 16, each wave owns TM x TN MFMA tiles of 32x32 (16 accumulator VGPRs each); A/B chunks are register-staged
// global->LDS with double buffering (one barrier per chunk); LDS images are k-major ([k][m] / [k][n]) so every ds_read_b32
// of an MFMA operand is bank-conflict-free.  (Round 1's wave-specialised kernel with separate loader waves, the A/B baseline of
// round 2, was removed in round 3: its measurements are in profiles/r01_*, r02_z_*_r01kernel_same_box.*.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/pc_math.h"
#include "pc_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// epilogue arithmetic on values already loaded (a0 / a1 = the aux tensors' elements at this output position)
__device__ __forceinline__ float epilogue_apply(int epi, float v, float a0, float a1)
{
    switch (epi) {
    case PC_EPI_NONE: return v;
    case PC_EPI_GELU: return pc_geluf(v);
    case PC_EPI_RES_GELU: return pc_geluf(v + a0);
    case PC_EPI_RES: return a0 + v;
    case PC_EPI_GATE: return a0 * pc_sigmoidf(v) + a1;
    case PC_EPI_GDN: return a0 * pc_rsqrtf(v);
    case PC_EPI_IGDN: return a0 * sqrtf(v);
    case PC_EPI_CLAMP01: return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    case PC_EPI_LRP: return a0 + 0.5f * pc_tanhf(v);
    case PC_EPI_LRP_ADD: return (a0 + 0.5f * pc_tanhf(v)) + a1;
    case PC_EPI_LEAKY: return v > 0.0f ? v : v * 0.01f;
    case PC_EPI_LEAKY_RES: return (v > 0.0f ? v : v * 0.01f) + a0;
    default: return v;
    }
}
// ---- GELU on TWO elements per lane with packed f32 operations (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32; round 4).
// The epilogue's GELU is VALU issue: ~55 instructions per element once a wave holds arguments on both sides of erf's |x| < 1 split (both
// branches run under divergence), 16 elements per lane and tile, beside the other workgroups' MFMA issue on the same SIMD -- 2.3 ms of a
// 51.7 ms serial step.  Here both branches are evaluated for a PAIR of elements by packed instructions and the
// results selected: per component exactly the operations of pc_geluf in exactly its order (IEEE fma / mul / add per component, the
// same constants), so every bit is pc_geluf's -- checked over ALL 2^32 arguments by pc_selftest_packed_gelu (tests/test_gpu_ops.py).
// pc_expf's range tests are dropped for the erfc branch: its argument -a^2 + q(a) lies in [-18.2, -1.9] for 1 <= a < 4, where they
// never fire; for other a the branch's value is not selected.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, float c) { return __builtin_elementwise_fma(a, b, f32x2{c, c}); }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, float b, f32x2 c) { return __builtin_elementwise_fma(a, f32x2{b, b}, c); }
__device__ __forceinline__ f32x2 pc_geluf2(f32x2 x)
{
    const f32x2 xs = x * 0.70710678118654752440f;                       // pc_geluf: erf's argument
    // |xs| < 1
    const f32x2 t = xs * xs;
    f32x2 r = {1.059527904e-06f, 1.059527904e-06f};
    r = pk_fma(r, t, -1.391170190e-05f);
    r = pk_fma(r, t, 1.195694713e-04f);
    r = pk_fma(r, t, -8.542670403e-04f);
    r = pk_fma(r, t, 5.223785061e-03f);
    r = pk_fma(r, t, -2.686613426e-02f);
    r = pk_fma(r, t, 1.128379107e-01f);
    r = pk_fma(r, t, -3.761263788e-01f);
    r = pk_fma(r, t, 1.283791661e-01f);
    const f32x2 e_small = pk_fma(xs, r, xs);
    // 1 <= |xs| < 4
    const f32x2 a = {fabsf(xs.x), fabsf(xs.y)};
    const f32x2 u = a - 2.5f;
    f32x2 q = {1.777464753e-08f, 1.777464753e-08f};
    q = pk_fma(q, u, -4.104854057e-08f);
    q = pk_fma(q, u, -1.758092054e-07f);
    q = pk_fma(q, u, 1.844959684e-06f);
    q = pk_fma(q, u, -1.272867667e-05f);
    q = pk_fma(q, u, 7.693984662e-05f);
    q = pk_fma(q, u, -4.209505278e-04f);
    q = pk_fma(q, u, 2.165525686e-03f);
    q = pk_fma(q, u, -1.085806731e-02f);
    q = pk_fma(q, u, 5.610625818e-02f);
    q = pk_fma(q, u, -3.526807427e-01f);
    q = pk_fma(q, u, -1.556815267e+00f);
    const f32x2 z = pk_fma(-a, a, q);                                   // erfc(a) = exp(-a^2 + q(a))
    f32x2 nf = pk_fma(z, 1.442695040888963387f, f32x2{0.5f, 0.5f});     // pc_expf(z), range tests dropped (see above)
    nf = f32x2{floorf(nf.x), floorf(nf.y)};
    f32x2 rr = pk_fma(nf, -6.931152344e-01f, z);
    rr = pk_fma(nf, -3.194618495e-05f, rr);
    f32x2 pp = {1.989939192e-04f, 1.989939192e-04f};
    pp = pk_fma(pp, rr, 1.393373357e-03f);
    pp = pk_fma(pp, rr, 8.333298378e-03f);
    pp = pk_fma(pp, rr, 4.166646302e-02f);
    pp = pk_fma(pp, rr, 1.666666716e-01f);
    pp = pk_fma(pp, rr, 5.000000000e-01f);
    const f32x2 r2 = rr * rr;
    const f32x2 ee = pk_fma(pp, r2, rr) + 1.0f;
    const f32x2 sc = {pc_bits2f((uint32_t)((int)nf.x + 127) << 23), pc_bits2f((uint32_t)((int)nf.y + 127) << 23)};
    const f32x2 big = 1.0f - ee * sc;                                   // 1 - erfc(a)
    // select, per component, what pc_erff returns
    f32x2 erf;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float xc = xs[c], ac = a[c];
        const float sat = (xc != xc) ? xc : (xc > 0.0f ? 1.0f : -1.0f);
        const float bg = xc > 0.0f ? big[c] : -big[c];
        erf[c] = ac < 1.0f ? e_small[c] : (!(ac < 4.0f) ? sat : bg);
    }
    return (0.5f * x) * (1.0f + erf);
}

__device__ __forceinline__ bool epilogue_uses_aux0(int epi) { return epi == PC_EPI_RES_GELU || epi == PC_EPI_RES || epi == PC_EPI_GATE || epi == PC_EPI_GDN || epi == PC_EPI_IGDN || epi == PC_EPI_LRP || epi == PC_EPI_LRP_ADD || epi == PC_EPI_LEAKY_RES; }
__device__ __forceinline__ bool epilogue_uses_aux1(int epi) { return epi == PC_EPI_GATE || epi == PC_EPI_LRP_ADD; }

__device__ __forceinline__ float epilogue_value(const pc_conv_params& p, float v, int64_t pix, int n)
{
    const float a0 = epilogue_uses_aux0(p.epi) ? p.aux0[pix * p.ld0 + n] : 0.0f;
    const float a1 = epilogue_uses_aux1(p.epi) ? p.aux1[pix * p.ld1 + n] : 0.0f;
    return epilogue_apply(p.epi, v, a0, a1);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool SMALLC>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const pc_conv_params p)
{
    constexpr int BK = 16;
    constexpr int TM = BM / WAVES_M / 32, TN = BN / WAVES_N / 32;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int AI = (BM * 4 + 255) / 256;          // float4 A loads per thread per chunk
    constexpr int BI = (BN * 4 + 255) / 256;          // float4 B loads per thread per chunk
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
    __shared__ float smem[2 * BK * (LDA + LDB)];
    float* As = smem;
    float* Bs = smem + 2 * BK * LDA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int phase = blockIdx.z;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int T = p.ntap[phase];
    const int HoWo = p.Ho * p.Wo;

    // ---- per-thread A rows (fixed over the K loop)
    int a_row[AI], a_iy0[AI], a_ix0[AI];
    int64_t a_boff[AI];
    bool a_ok[AI];
    const int kq = tid & 3;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int row = (tid >> 2) + i * 64;
        a_row[i] = row;
        const int m = m0 + row;
        a_ok[i] = (row < BM) && (m < p.M);
        const int mm = a_ok[i] ? m : 0;
        const int b = mm / HoWo, r = mm - b * HoWo;
        const int oy = r / p.Wo, ox = r - oy * p.Wo;
        a_iy0[i] = oy * p.stride;
        a_ix0[i] = ox * p.stride;
        a_boff[i] = (int64_t)b * p.H * p.W;
    }
    // ---- per-thread B positions
    constexpr int BQ = BN / 4;                         // float4 per B row
    int b_k[BI], b_n[BI];
    bool b_ok[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int e = tid + i * 256;
        b_k[i] = e / BQ;
        b_n[i] = (e % BQ) * 4;
        b_ok[i] = (b_k[i] < BK) && (n0 + b_n[i] < p.Cout);   // Cout % 4 == 0 except the N=3 layer (handled per element)
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // ---- chunk iterator state (uniform): tap t, segment s, channel c within segment, global channel cg
    int it_t = 0, it_s = 0, it_c = 0, it_cg = 0;
    int nchunks;
    if (SMALLC) nchunks = (T * p.Cin + BK - 1) / BK;
    else nchunks = T * (p.Cin / BK);

    float4 ra[AI], rb[BI];

    auto load_chunk = [&](int chunk) {
        if (SMALLC) {
            // flattened k = t*Cin + c; element-wise gather with generic input strides (NCHW image input)
            const int Ktot = T * p.Cin;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kf = chunk * BK + kq * 4 + j;
                    float x = 0.0f;
                    if (a_ok[i] && kf < Ktot) {
                        const int t = kf / p.Cin, c = kf - t * p.Cin;
                        const int iy = a_iy0[i] + p.dy[phase][t], ix = a_ix0[i] + p.dx[phase][t];
                        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                            const int64_t b = a_boff[i] / ((int64_t)p.H * p.W);
                            x = p.seg[0].ptr[b * p.in_sb + (int64_t)iy * p.in_sy + (int64_t)ix * p.in_sx + (int64_t)c * p.in_sc];
                        }
                    }
                    v[j] = x;
                }
                ra[i] = make_float4(v[0], v[1], v[2], v[3]);
            }
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                const int kf = chunk * BK + b_k[i];
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (b_ok[i] && kf < Ktot) {
                    const float* wp = p.w + (int64_t)kf * p.Cout + n0 + b_n[i];   // w[t][c][n] flattened == w[kf][n] (wtap = t)
                    if (n0 + b_n[i] + 3 < p.Cout) w = *reinterpret_cast<const float4*>(wp);
                    else { w.x = wp[0]; if (n0 + b_n[i] + 1 < p.Cout) w.y = wp[1]; if (n0 + b_n[i] + 2 < p.Cout) w.z = wp[2]; }
                }
                rb[i] = w;
            }
        } else {
            const int dy = p.dy[phase][it_t], dx = p.dx[phase][it_t];
            const float* sp = p.seg[it_s].ptr;
            const int sld = p.seg[it_s].ld;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a_ok[i] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                    v = *reinterpret_cast<const float4*>(sp + (a_boff[i] + (int64_t)iy * p.W + ix) * sld + it_c + kq * 4);
                if (p.square) { v.x *= v.x; v.y *= v.y; v.z *= v.z; v.w *= v.w; }
                ra[i] = v;
            }
            const float* wrow = p.w + ((int64_t)p.wtap[phase][it_t] * p.Cin + it_cg) * p.Cout + n0;
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (b_ok[i]) {
                    const float* wp = wrow + (int64_t)b_k[i] * p.Cout + b_n[i];
                    if (n0 + b_n[i] + 3 < p.Cout) w = *reinterpret_cast<const float4*>(wp);
                    else { w.x = wp[0]; if (n0 + b_n[i] + 1 < p.Cout) w.y = wp[1]; if (n0 + b_n[i] + 2 < p.Cout) w.z = wp[2]; }
                }
                rb[i] = w;
            }
            // advance (t, seg, c)
            it_c += BK; it_cg += BK;
            if (it_c >= p.seg[it_s].nch) { it_c = 0; ++it_s; if (it_s >= p.nseg) { it_s = 0; it_cg = 0; ++it_t; } }
        }
    };

    auto store_chunk = [&](int buf) {
        float* a = As + buf * BK * LDA;
        float* b = Bs + buf * BK * LDB;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            if (a_row[i] < BM) {
                a[(kq * 4 + 0) * LDA + a_row[i]] = ra[i].x;
                a[(kq * 4 + 1) * LDA + a_row[i]] = ra[i].y;
                a[(kq * 4 + 2) * LDA + a_row[i]] = ra[i].z;
                a[(kq * 4 + 3) * LDA + a_row[i]] = ra[i].w;
            }
        }
#pragma unroll
        for (int i = 0; i < BI; ++i)
            if (b_k[i] < BK) *reinterpret_cast<float4*>(b + b_k[i] * LDB + b_n[i]) = rb[i];
    };

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    int cur = 0;
    const int half = lane >> 5, l31 = lane & 31;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        if (chunk + 1 < nchunks) load_chunk(chunk + 1);
        const float* a = As + cur * BK * LDA + wm * (TM * 32) + l31;
        const float* b = Bs + cur * BK * LDB + wn * (TN * 32) + l31;
#pragma unroll
        for (int st = 0; st < BK / 2; ++st) {
            // chain order of the contract: aligned groups of 8 k, in-group order 0,4,1,5,2,6,3,7 -- MFMA step s of a
            // group takes k = s (lanes 0-31) and k = s + 4 (lanes 32-63)
            const int kk = (st >> 2) * 8 + (st & 3) + 4 * half;
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = a[kk * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = b[kk * LDB + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (chunk + 1 < nchunks) store_chunk(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    // Dense NHWC output on the GEMM's own pixel grid without aux tensors (the 3-channel input layer: 402 MB of output at Config 2): the
    // output pixel of a row is the row itself -- no divisions, the 16 rows unrolled, buffer stores relative to the tile's first row with
    // the rows beyond M dropped by the range check (same form as conv_igemm_uni_kernel's direct epilogue).
    if (p.dense_out && p.out_sc == 1 && !epilogue_uses_aux0(p.epi) && (int64_t)32 * p.out_sx * 4 < ((int64_t)1 << 31)) {
        const int epi_v = p.epi;
        const uint32_t sx4 = (uint32_t)p.out_sx * 4u;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * (TM * 32) + i * 32;
            if (mb >= p.M) continue;
            const int rows_live = p.M - mb < 32 ? p.M - mb : 32;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.out + (int64_t)mb * p.out_sx, 0, (int)((int64_t)rows_live * p.out_sx * 4), 0x00020000);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * (TN * 32) + j * 32 + l31;
                if (n >= p.Cout) continue;
                const float bv = p.bias ? p.bias[n] : 0.0f;
                const bool hb = p.bias != nullptr;
                const uint32_t lane_off = (uint32_t)((4 * half * (int)p.out_sx + n) * 4);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r];
                    if (hb) v = v + bv;
                    v = epilogue_apply(epi_v, v, 0.0f, 0.0f);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, (int)((uint32_t)((r & 3) + 8 * (r >> 2)) * sx4 + lane_off), 0, 0);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int m = m0 + wm * (TM * 32) + i * 32 + row;
            if (m >= p.M) continue;
            const int b = m / HoWo, rr = m - b * HoWo;
            const int oy = rr / p.Wo, ox = rr - oy * p.Wo;
            int Y = oy * p.osy + p.ooy[phase], X = ox * p.osx + p.oox[phase];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * (TN * 32) + j * 32 + l31;
                if (n >= p.Cout) continue;
                float v = acc[i][j][r];
                if (p.bias) v = v + p.bias[n];
                int nn = n, YY = Y, XX = X;
                if (p.pixel_shuffle) { nn = n >> 2; YY = 2 * Y + ((n >> 1) & 1); XX = 2 * X + (n & 1); }
                const int64_t pix = ((int64_t)b * p.outH + YY) * p.outW + XX;
                v = epilogue_value(p, v, pix, nn);
                p.out[(int64_t)b * p.out_sb + (int64_t)YY * p.out_sy + (int64_t)XX * p.out_sx + (int64_t)nn * p.out_sc] = v;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// g_a.0 + g_a.1 in ONE kernel (round 4): the 3 -> 192 input convolution (5x5 s2, element gather
// from the NCHW image; models/cnn.py:34-35, models/utils.py:186-193) and the GDN that consumes it (layers/gdn.py:50-63).  As two
// launches the pair wrote the 192-channel tensor (402 MB at Config 2), read it back twice (GEMM operand + epilogue) and wrote the
// normalised tensor: 0.93 ms for 26.9 GFLOP.  Here a workgroup owns 64 pixels x ALL 192 channels (2 x 2 waves, wave tile 32 x 96 =
// three accumulators): phase 1 is conv_igemm_kernel's SMALLC loop (K = 75 in five 16-chunks); x = chain + bias goes to LDS k-major
// ([channel][pixel]) -- exactly the A image phase 2 needs; phase 2 contracts x^2 with gamma (K = 192, gamma chunks double-buffered
// through LDS) and the epilogue reads x back from the same image: out = x * rsqrt(beta + norm).  Per output element the two fmaf
// chains, the bias adds, the square and the rsqrt are those of the two-launch form (same k order, same operations): bit-identical.
// Phase 2 reads its operands the unified kernel's way: 16-byte k-quads ([pixel][channel] image of x, gamma chunks as [column][k-quad] with
// an XOR swizzle), one ds_read_b128 per operand and four MFMA steps.  LDS: 64 x 196 floats of x + 2 x 192 x 16 of weights = 75 392 B
// -> two workgroups per CU.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_igemm_in_gdn_kernel(const pc_conv_params p)
{
    constexpr int BM = 64, CN = 192, BK = 16, TN = 3;
    constexpr int XP = CN + 4, LDA = BM + 4, LDB = CN + 4;
    extern __shared__ float fsm[];
    // x image [pixel][channel], pitch 196 floats: phase 1 writes it conflict-free (lane = channel), phase 2 reads 16-byte k-quads of a
    // pixel row (consecutive rows start 16 B apart modulo 128: eight lanes, eight bank quads), the epilogue reads it as it was written
    float* Xs = fsm;                                   // [BM][XP]
    // phase 2: 2 x [CN][BK] gamma chunks as 16-byte k-quads, quad q of column n stored at quad slot q ^ ((n >> 1) & 3) (conflict-free
    // ds_write_b128 / ds_read_b128); phase 1: [BK][LDB] weights, then [BK][LDA] pixels, k-major (conv_igemm_kernel's image)
    float* Bs = Xs + BM * XP;
    float* As = Bs + BK * LDB;
    static_assert(BK * LDB + BK * LDA <= 2 * CN * BK, "phase 1 fits phase 2's buffers");
    int* ktab = reinterpret_cast<int*>(Bs + 2 * CN * BK);    // [2][80]: per flattened k = tap * Cin + c, the element offset from the pixel's (iy0, ix0) and dy << 16 | dx

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int T = p.ntap[0];
    const int HoWo = p.Ho * p.Wo;
    const int Ktot = T * p.Cin;

    // ---- phase 1: this thread gathers 4 consecutive k of one pixel row per chunk; 3 float4 of weights.  lane = pixel row, wave = k-quad:
    // one load instruction then reads ONE (tap, channel) for 64 neighbouring output pixels -- addresses two floats apart along an image
    // row -- instead of sixteen pixels x four different taps (conv_igemm_kernel's mapping: 876 -> 8xx us for the fused pair)
    const int kq = wave, a_row = lane;
    const int64_t am = m0 + a_row;
    const bool a_ok = am < (int64_t)p.M;
    int a_iy0, a_ix0;
    int64_t a_b;
    {
        const int64_t mm = a_ok ? am : 0;
        a_b = mm / HoWo;
        const int r = (int)(mm - a_b * HoWo);
        const int oy = r / p.Wo, ox = r - oy * p.Wo;
        a_iy0 = oy * p.stride; a_ix0 = ox * p.stride;
    }
    const float* xin = p.seg[0].ptr + a_b * p.in_sb + (int64_t)a_iy0 * p.in_sy + (int64_t)a_ix0 * p.in_sx;
    if (tid < 80) {                                     // the k -> (tap, channel) division once per workgroup, not once per gathered element
        int o = 0, d = 0;
        if (tid < Ktot) {
            const int t = tid / p.Cin, c = tid - t * p.Cin;
            o = (int)((int64_t)p.dy[0][t] * p.in_sy + (int64_t)p.dx[0][t] * p.in_sx + (int64_t)c * p.in_sc);
            d = (p.dy[0][t] << 16) | (p.dx[0][t] & 0xffff);
        }
        ktab[tid] = o; ktab[80 + tid] = d;
    }
    __syncthreads();
    constexpr int BQ = CN / 4;
    int b_k[3], b_n[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int e = tid + i * 256; b_k[i] = e / BQ; b_n[i] = (e % BQ) * 4; }

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    float4 ra, rb[3];
    auto load1 = [&](int chunk) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kf = chunk * BK + kq * 4 + j;
            float x = 0.0f;
            if (a_ok && kf < Ktot) {
                const int d = ktab[80 + kf];
                const int iy = a_iy0 + (d >> 16), ix = a_ix0 + (int)(short)(d & 0xffff);
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) x = xin[ktab[kf]];
            }
            v[j] = x;
        }
        ra = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int kf = chunk * BK + b_k[i];
            rb[i] = kf < Ktot ? *reinterpret_cast<const float4*>(p.w + (int64_t)kf * CN + b_n[i]) : make_float4(0.f, 0.f, 0.f, 0.f);   // w[t][c][n] == w[kf][n]
        }
    };
    auto store1 = [&]() {
        As[(kq * 4 + 0) * LDA + a_row] = ra.x;
        As[(kq * 4 + 1) * LDA + a_row] = ra.y;
        As[(kq * 4 + 2) * LDA + a_row] = ra.z;
        As[(kq * 4 + 3) * LDA + a_row] = ra.w;
#pragma unroll
        for (int i = 0; i < 3; ++i) *reinterpret_cast<float4*>(Bs + b_k[i] * LDB + b_n[i]) = rb[i];
    };
    const int nch1 = (Ktot + BK - 1) / BK;
    load1(0);
    for (int chunk = 0; chunk < nch1; ++chunk) {
        store1();
        __syncthreads();
        if (chunk + 1 < nch1) load1(chunk + 1);                          // in flight under this chunk's MFMAs
        const float* a = As + wm * 32 + l31;
        const float* b = Bs + wn * (TN * 32) + l31;
#pragma unroll
        for (int st = 0; st < BK / 2; ++st) {
            const int kk = (st >> 2) * 8 + (st & 3) + 4 * half;           // the contract's in-group order 0,4,1,5,2,6,3,7
            const float av = a[kk * LDA];
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[kk * LDB + j * 32], acc[j], 0, 0, 0);
        }
        __syncthreads();                                                  // single-buffered: everyone is done reading before the next store
    }
    // x = conv + bias -> LDS, k-major
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = wn * (TN * 32) + j * 32 + l31;
        const float bv = p.bias ? p.bias[n] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float v = acc[j][r];
            if (p.bias) v = v + bv;
            Xs[row * XP + n] = v;
            acc[j][r] = 0.0f;
        }
    }
    // ---- phase 2: norm = gamma . x^2 (K = 192), gamma in layout 1 ([n][k]); chunk c in buffer c & 1
    int g_n[3], g_q[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int e = tid + i * 256; g_n[i] = e >> 2; g_q[i] = e & 3; }
    auto load2 = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < 3; ++i) rb[i] = *reinterpret_cast<const float4*>(p.fg_gamma + (int64_t)g_n[i] * CN + chunk * BK + g_q[i] * 4);
    };
    auto store2 = [&](int buf) {
        float* b = Bs + buf * CN * BK;
#pragma unroll
        for (int i = 0; i < 3; ++i) *reinterpret_cast<float4*>(b + g_n[i] * BK + ((g_q[i] ^ ((g_n[i] >> 1) & 3)) << 2)) = rb[i];
    };
    load2(0);
    store2(0);
    __syncthreads();                                                      // x image and gamma chunk 0 in place
    constexpr int NCH2 = CN / BK;
    for (int chunk = 0; chunk < NCH2; ++chunk) {
        const int cur = chunk & 1;
        if (chunk + 1 < NCH2) load2(chunk + 1);
        // operands as 16-byte k-quads, the unified kernel's way: lanes 0-31 hold quad 2g, lanes 32-63 quad 2g+1 of an aligned group of 8 k --
        // MFMA step s then contracts k = 8g + s and 8g + 4 + s, the contract's 0,4,1,5,2,6,3,7.  One read per operand feeds four MFMAs
        // (the first form read 32 bits per MFMA step: 32 reads + 8 multiplies per chunk against 8 + 4 packed here; at two waves per SIMD
        // every instruction between MFMAs shows)
        const float* arow = Xs + (wm * 32 + l31) * XP + chunk * BK;
        const float* bbase = Bs + cur * CN * BK;
#pragma unroll
        for (int g8 = 0; g8 < BK / 8; ++g8) {
            f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + 8 * g8 + 4 * half);
            a4 = a4 * a4;                                                 // gdn.py:56: the contraction runs over x^2
            f32x4 b4[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = wn * (TN * 32) + j * 32 + l31;
                b4[j] = *reinterpret_cast<const f32x4*>(bbase + col * BK + (((2 * g8 + half) ^ ((col >> 1) & 3)) << 2));
            }
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[st], b4[j][st], acc[j], 0, 0, 0);
        }
        if (chunk + 1 < NCH2) store2(cur ^ 1);
        __syncthreads();
    }
    // ---- epilogue: out = x * rsqrt(beta + norm), NHWC
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = wn * (TN * 32) + j * 32 + l31;
        const float bv = p.fg_beta[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int64_t m = m0 + row;
            if (m >= (int64_t)p.M) continue;
            const float v = acc[j][r] + bv;
            p.out[m * p.out_sx + n] = Xs[row * XP + n] * pc_rsqrtf(v);
        }
    }
}


// ------------------------------------------------------------------------------------------
// LDS-DMA implicit GEMM (weight layout 1): shared pieces.  `buffer_load_dwordx4 ... lds` moves 16 B per lane straight into LDS
// (no VGPR staging, no ds_write); the LDS image is [row][k-quad] 16-byte pieces with an XOR swizzle on the SOURCE address (LDS-DMA
// writes lane-linearly), read back conflict-free with ds_read_b128; weights are pre-packed [tap][Cout][Cin] (K contiguous per
// output channel).  The contract's chain order inside an aligned group of 8 k (0,4,1,5,2,6,3,7) is exactly what the MFMA computes
// when lanes 0-31 hold the group's first 16-byte quad and lanes 32-63 its second: per 4 MFMAs a wave issues 2 ds_read_b128 and no
// VALU (an earlier k-ascending contract needed 3 reads + 4 v_cndmask per 4 MFMAs: 22 % slower, profiles/r01_tune_tune12.log).
// ------------------------------------------------------------------------------------------
// diagnostic build only (PC_CONV_DBG & 64): per-block cycle sums of the K-loop / prologue / epilogue phases, [block][16]
__device__ unsigned long long pc_dbg_stamps[8192][16];

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KQ> __device__ __forceinline__ int pc_swz(int r) { return KQ == 16 ? (r & 15) : (KQ == 4 ? ((r >> 2) & 3) : ((r >> 1) & 7)); }

template <int N> __device__ __forceinline__ void pc_wait_vm()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most `chunks` * NI of this wave's DMA instructions are still in flight (chunks is wave-uniform, small)
template <int NI, int MAXC> __device__ __forceinline__ void pc_wait_chunks(int chunks)
{
    if constexpr (MAXC <= 0) pc_wait_vm<0>();
    else {
        if (chunks >= MAXC) pc_wait_vm<MAXC * NI>();
        else pc_wait_chunks<NI, MAXC - 1>(chunks);
    }
}

struct pc_run { const float* a_base; const float* w_base; int ld, nch, tap, pad; };   // one (tap, input segment) of the K loop

// ------------------------------------------------------------------------------------------
// Unified kernel (round 2): every wave loads AND multiplies; no loader waves.
//
// Why: a `buffer_load ... lds` instruction holds a SIMD's issue port for tens of cycles
// (MI355X_MICROARCH.md: ~60 among bare MFMAs).  Issued by a separate loader wave at raised priority it
// lands wherever the arbiter puts it -- often at the moment the matrix pipe drains, and the MFMA that
// was ready waits behind it.  Issued by the MFMA wave itself, right AFTER one of its own MFMAs, the
// DMA sits in that MFMA's 64-cycle shadow (the pipe is busy anyway, no other MFMA could start).
// So: 256 threads = 4 waves (2 x 2), each wave owns TM x TN accumulator tiles of 32x32 (block tile
// 64*TM x 64*TN) and a quarter of every chunk's LDS-DMA pieces, placed one by one behind MFMAs of the
// chunk it is computing.  Larger wave tiles (TM, TN = 2) halve the L2->LDS bytes and the DMA / ds_read
// instructions per FLOP.  LDS image, swizzle, run table, buffer-form addressing, XCD-aware tile order,
// epilogues and -- above all -- each output element's fmaf chain are those of the kernel above.
//
// Pipeline (S stages, chunk c in stage c % S), one raw barrier per chunk:
//   loop c:  s_waitcnt vmcnt(NI*(S-2))   my pieces of chunk c have landed (c+1 .. c+S-2 may be in flight)
//            s_barrier                   everyone's have; everyone is done reading chunk c-1
//            MFMAs of chunk c, with the NI pieces of chunk c+S-1 (into the stage chunk c-1 occupied) between them
// ------------------------------------------------------------------------------------------
// DBG (tuning builds only, PC_CONV_DBG): 1 no MFMAs, 2 no DMA issue, 4 every DMA piece out of range (zero fill, no L2 traffic), 8 no operand reads
// (register budget: the K-chunk-16 instantiation serves the grids with several workgroups per CU and is held to six waves per SIMD --
// the unrolled epilogue alone would take 96 registers, five waves; K-chunk 32 serves the one-workgroup-per-CU grids)
template <int BK, int S, int TM, int TN, bool SQ, int DBG = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TM * TN == 1 ? (BK == 16 ? 6 : 5) : 1))) void conv_igemm_uni_kernel(const pc_conv_params p)
{
    constexpr int BM = 64 * TM, BN = 64 * TN, KQ = BK / 4;
    constexpr int NTH = 256;
    constexpr int A_PIECES = BM * KQ, B_PIECES = BN * KQ, STAGE = A_PIECES + B_PIECES;
    constexpr int AIN = A_PIECES / NTH, BIN = B_PIECES / NTH, NI = AIN + BIN;   // DMA instructions per thread per chunk
    constexpr int NG = BK / 8;                              // groups of 8 k = 4 MFMA steps per accumulator
    constexpr int NSLOT = NG * 4 * TM * TN;                 // MFMAs per wave per chunk
    static_assert(A_PIECES % NTH == 0 && B_PIECES % NTH == 0 && NI <= NSLOT, "tile / thread mismatch");
    static_assert(S >= 2 && (S - 2) * NI < 64, "vmcnt range");
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ float4 smem[];                        // S stages, then the run table (launch_uni sizes it)

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (p.dbg & 512) {
        // Workgroups that share a CU were dispatched ~256 ids apart (round-robin over the CUs): give them different issue priorities so
        // that one of them runs at full rate and the others fill its stalls, instead of all advancing in lockstep and meeting at
        // their barriers together.  Speed only; nothing depends on the placement guess.
        switch ((blockIdx.x >> 8) & 3) {
        case 0: __builtin_amdgcn_s_setprio(3); break;
        case 1: __builtin_amdgcn_s_setprio(2); break;
        case 2: __builtin_amdgcn_s_setprio(1); break;
        default: __builtin_amdgcn_s_setprio(0); break;
        }
    }
    unsigned long long tl[4] = {0, 0, 0, 0};               // DBG & 64: timeline of this wave in s_memrealtime ticks (100 MHz, chip-wide)
    if (DBG & 64) tl[0] = __builtin_amdgcn_s_memrealtime();
    // XCD-aware tile order (1-D grid).  Workgroups are handed to the 8 XCDs round-robin in dispatch order and every XCD has its own
    // 4 MB L2.  With a plain (x = M tile, y = N tile) grid the N tiles / phases / groups that read the SAME activation tile ran thousands
    // of workgroups apart and neighbouring M tiles (which share input rows through the taps) landed on different XCDs: PMC showed 3.7 GB
    // (x2 by the gfx950 correction) fetched for the 0.4 GB input of the largest layer.  Here XCD x owns the contiguous band of M tiles
    // [x*mpx, (x+1)*mpx) and walks it with (N tile, phase/group) fastest.
    const int MT = (p.M + BM - 1) / BM, NT = (p.Cout + BN - 1) / BN, NZ = p.ngroup == 2 ? 2 : p.nphase;
    const int mpx = (MT + 7) >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int m_local = slot / (NT * NZ), nz = slot - m_local * (NT * NZ);
    // permuted rows: the cheap border tiles are the last tiles of the launch -- deal the tiles round-robin over the XCDs so that every
    // XCD ends on its share of them (contiguous bands would leave them all to the last XCD and the launch would end no earlier)
    int m_tile = p.rowperm ? m_local * 8 + xcd : xcd * mpx + m_local;
    int zsel = nz / NT, n_tile = nz - zsel * NT;
    if (p.group_xcd) {
        // Grouped launch (cc_mean || cc_scale): with both groups on every XCD an L2 streams 2 x NT weight slabs -- 9.4 MB at N = 224 against
        // 4 MB of L2 -- and the heads fetched 12x their operands (profiles/r03_m_traffic_by_shape.json; 4.4x ungrouped).  Here XCDs 0-3
        // run group 0 and XCDs 4-7 group 1: the same workgroups, each L2 sees one group's slabs.
        const int x4 = xcd & 3, per = 2 * mpx;                       // M tiles this XCD may take (>= MT / 4)
        const int ml = slot / NT;
        zsel = xcd >> 2; n_tile = slot - ml * NT;
        m_tile = p.rowperm ? ml * 4 + x4 : x4 * per + ml;
    }
    if (m_tile >= MT) return;
    int phase = zsel;
    const float* seg0_ptr = p.seg[0].ptr;
    const float* wbase = p.w;
    const float* bias = p.bias;
    float* outp = p.out;
    if (p.ngroup == 2 && zsel == 1) { seg0_ptr = p.g1_seg0; wbase = p.g1_w; bias = p.g1_bias; outp = p.g1_out; }
    if (p.ngroup == 2) phase = 0;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const int T = p.ntap[phase];
    const int HoWo = p.Ho * p.Wo;
    int chunks_per_tap = 0;
    for (int s = 0; s < p.nseg; ++s) chunks_per_tap += (p.seg[s].nch + BK - 1) / BK;
    int Tv = T;                                            // taps this tile really visits (rowperm: fewer on border tiles)
    uint32_t tmask = T >= 32 ? 0xffffffffu : ((1u << T) - 1u);

    // ---- this thread's DMA pieces: piece pa of the A image = (row pa / KQ, LDS slot pa % KQ) holds source k-quad slot ^ swz(row)
    constexpr int OOB = (int)0x80000000;
    int a_q[AIN], a_rel[AIN]; uint32_t a_mask[AIN];
    int64_t pix0;
    if (p.rowtab && p.rowperm) {
        // permuted rows: the tile's rows may lie anywhere in the batch.  Offsets are taken from the tile's smallest pixel index
        // (buffer offsets are unsigned), and the taps no row of the tile needs are dropped from the run table below.
        __shared__ int s_minpix;
        __shared__ uint32_t s_tmask;
        if (threadIdx.x == 0) { s_minpix = 0x7fffffff; s_tmask = 0u; }
        __syncthreads();
        int pixv[AIN];
        int mn = 0x7fffffff;
        uint32_t orm = 0u;
#pragma unroll
        for (int i = 0; i < AIN; ++i) {
            const int pa = (wave * AIN + i) * 64 + lane;
            const int row = pa / KQ, sl = pa % KQ;
            a_q[i] = sl ^ pc_swz<KQ>(row);
            const int m = m0 + row;
            const bool ok = m < p.M;
            pixv[i] = ok ? p.rowtab[m] : 0x7fffffff;
            a_mask[i] = ok ? (uint32_t)p.rowtab[(size_t)(1 + phase) * p.M + m] : 0u;
            mn = pixv[i] < mn ? pixv[i] : mn;
            orm |= a_mask[i];
        }
        for (int off = 32; off; off >>= 1) { const int o = __shfl_xor(mn, off); mn = o < mn ? o : mn; orm |= (uint32_t)__shfl_xor((int)orm, off); }
        if (lane == 0) { atomicMin(&s_minpix, mn); atomicOr(&s_tmask, orm); }
        __syncthreads();
        pix0 = s_minpix;
        tmask &= s_tmask;
        if (tmask == 0u) tmask = 1u;                       // (a tile without a live row: keep one tap so that the loops stay well-formed)
        Tv = __popc(tmask);
#pragma unroll
        for (int i = 0; i < AIN; ++i) a_rel[i] = pixv[i] == 0x7fffffff ? 0 : pixv[i] - (int)pix0;
    } else if (p.rowtab) {
        pix0 = p.rowtab[m0];
#pragma unroll
        for (int i = 0; i < AIN; ++i) {
            const int pa = (wave * AIN + i) * 64 + lane;
            const int row = pa / KQ, sl = pa % KQ;
            a_q[i] = sl ^ pc_swz<KQ>(row);
            const int m = m0 + row;
            const bool ok = m < p.M;
            a_rel[i] = ok ? p.rowtab[m] - (int)pix0 : 0;
            a_mask[i] = ok ? (uint32_t)p.rowtab[(size_t)(1 + phase) * p.M + m] : 0u;
        }
    } else if (p.ident_rows) {
        pix0 = m0;
#pragma unroll
        for (int i = 0; i < AIN; ++i) {
            const int pa = (wave * AIN + i) * 64 + lane;
            const int row = pa / KQ, sl = pa % KQ;
            a_q[i] = sl ^ pc_swz<KQ>(row);
            a_rel[i] = row;
            a_mask[i] = (m0 + row < p.M) ? 1u : 0u;
        }
    } else {
        {
            const int b = m0 / HoWo, r = m0 - b * HoWo;
            const int oy = r / p.Wo, ox = r - oy * p.Wo;
            pix0 = ((int64_t)b * p.H + (int64_t)oy * p.stride) * p.W + (int64_t)ox * p.stride;
        }
#pragma unroll
        for (int i = 0; i < AIN; ++i) {
            const int pa = (wave * AIN + i) * 64 + lane;
            const int row = pa / KQ, sl = pa % KQ;
            a_q[i] = sl ^ pc_swz<KQ>(row);
            const int m = m0 + row;
            const bool ok = m < p.M;
            const int mm = ok ? m : m0;
            const int b = mm / HoWo, r = mm - b * HoWo;
            const int oy = r / p.Wo, ox = r - oy * p.Wo;
            const int iy0 = oy * p.stride, ix0 = ox * p.stride;
            a_rel[i] = (int)((((int64_t)b * p.H + iy0) * p.W + ix0) - pix0);
            uint32_t mask = 0;
            for (int t = 0; t < T; ++t) {
                const int iy = iy0 + p.dy[phase][t], ix = ix0 + p.dx[phase][t];
                if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) mask |= 1u << t;
            }
            a_mask[i] = ok ? mask : 0u;
        }
    }
    int b_q[BIN], b_off[BIN];
#pragma unroll
    for (int i = 0; i < BIN; ++i) {
        const int pb = (wave * BIN + i) * 64 + lane;
        const int row = pb / KQ, sl = pb % KQ;
        b_q[i] = sl ^ pc_swz<KQ>(row);
        b_off[i] = (n0 + row < p.Cout) ? (row * p.Cin + 4 * b_q[i]) * 4 : OOB;
    }
    // ---- run table, built once per block: one entry per (tap, input segment), and a second one for the segment's last nch % BK
    // channels when there are any -- a short chunk is then a run of its own whose per-piece offsets (computed at the run boundary)
    // already mask the absent k-quads, and the chunk loop never has to look at the channel count
    pc_run* runs = reinterpret_cast<pc_run*>(smem + S * STAGE);
    int runs_per_tap = 0;
    for (int sg = 0; sg < p.nseg; ++sg) runs_per_tap += (p.seg[sg].nch >= BK ? 1 : 0) + (p.seg[sg].nch % BK ? 1 : 0);
    const int nchunks = Tv * chunks_per_tap;
    const int nruns = Tv * runs_per_tap;
    for (int r = threadIdx.x; r < nruns; r += NTH) {
        const int tv = r / runs_per_tap;                   // the tv-th tap this tile visits = the tv-th set bit of tmask
        int t = 0;
        { uint32_t mk = tmask; for (int q = 0; q < tv; ++q) mk &= mk - 1u; t = __ffs((int)mk) - 1; }
        int rr = r - tv * runs_per_tap, sg = 0, cbase = 0, ch0 = 0, nch = 0;
        for (;; ++sg) {                                    // locate run rr of this tap: (segment, full part or remainder)
            const int full = p.seg[sg].nch / BK * BK, rem = p.seg[sg].nch - full;
            const int here = (full ? 1 : 0) + (rem ? 1 : 0);
            if (rr < here) { const bool is_rem = full ? rr == 1 : true; ch0 = is_rem ? full : 0; nch = is_rem ? rem : full; break; }
            rr -= here; cbase += p.seg[sg].nch;
        }
        pc_run d;
        d.ld = p.seg[sg].ld; d.nch = nch; d.tap = t; d.pad = 0;
        d.a_base = (sg == 0 ? seg0_ptr : p.seg[sg].ptr) + (int64_t)(p.dy[phase][t] * p.W + p.dx[phase][t]) * d.ld + ch0;
        d.w_base = wbase + (int64_t)p.wtap[phase][t] * p.Cout * p.Cin + cbase + ch0;
        runs[r] = d;
    }
    __syncthreads();                                       // no DMA in flight yet: a plain barrier is fine
    const uint32_t runs_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)runs;
    // The steady-state instruction stream of a chunk holds NO VALU instruction: an MFMA chain on one accumulator only keeps its
    // 64-cycle cadence while nothing but MFMAs, LDS reads, LDS-DMAs and scalar instructions sit between two dependent MFMAs -- a
    // single v_add / v_cndmask / v_readfirstlane there costs the wave ~60 cycles (tools/mfma_issue_probe.hip: 124 vs 64 cycles per
    // MFMA with one VALU in between).  Hence: the two buffer descriptors live in SGPRs from run boundary to run boundary, the
    // per-piece offsets are the run's own (a segment's short last chunk is a run of its own: no per-chunk select), and the operand
    // read addresses are fixed registers with the stage as an immediate offset (the chunk body is instantiated per stage).
    auto mk = [](uint64_t a) { return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(a), 0, 0x7fffffff, 0x00020000); };
    __amdgpu_buffer_rsrc_t rsrc_a = mk(0), rsrc_b = mk(0);
    int ea[AIN], eb[BIN];                                  // this run's per-piece offsets: tap validity and (short runs) absent k-quads folded in
    int run = 0, c_left = 0, koff = 0;
    if (DBG & 4) {
#pragma unroll
        for (int i = 0; i < BIN; ++i) b_off[i] = OOB;
    }
    auto enter_run = [&](int r) {                          // hand-written LDS reads: behind a plain LDS load hipcc places s_waitcnt vmcnt(0) (it must assume the read aliases an in-flight LDS-DMA)
        u32x4 lo, hi;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(lo), "=&v"(hi) : "v"(runs_lds + (uint32_t)r * 32u) : "memory");
        const uint64_t pa = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)lo.y) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)lo.x);
        const uint64_t pw = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)lo.w) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)lo.z);
        const int d_ld = __builtin_amdgcn_readfirstlane((int)hi.x), d_nch = __builtin_amdgcn_readfirstlane((int)hi.y);
        const int d_tap = __builtin_amdgcn_readfirstlane((int)hi.z);
        // (readfirstlane once more on the finished addresses: hipcc must SEE that the descriptors are wave-uniform, or it keeps them in
        // VGPRs and wraps every DMA in a waterfall loop)
        auto uni64 = [](uint64_t v) {
            return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
        };
        rsrc_a = mk(uni64(pa + (uint64_t)(pix0 * d_ld) * 4u));
        rsrc_b = mk(uni64(pw + (uint64_t)((int64_t)n0 * p.Cin) * 4u));
        c_left = d_nch; koff = 0;
#pragma unroll
        for (int i = 0; i < AIN; ++i) ea[i] = (((a_mask[i] >> d_tap) & 1u) && 4 * a_q[i] < d_nch && !(DBG & 4)) ? (a_rel[i] * d_ld + 4 * a_q[i]) * 4 : OOB;
#pragma unroll
        for (int i = 0; i < BIN; ++i) eb[i] = (4 * b_q[i] < d_nch) ? b_off[i] : OOB;
    };
    enter_run(0);
    // piece q of the chunk being fetched, into stage `stage`
    auto piece = [&](int q, int stage) {
        if (DBG & 2) return;
        float4* base = smem + stage * STAGE;
        if (q < AIN)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void*)(base + (wave * AIN + q) * 64), 16, ea[q < AIN ? q : 0], koff, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (__attribute__((address_space(3))) void*)(base + A_PIECES + (wave * BIN + (q - AIN)) * 64), 16,
                                                     eb[q >= AIN ? q - AIN : 0], koff, 0, 0);
    };
    auto advance = [&]() {
        koff += BK * 4;
        c_left -= BK;
        if (c_left <= 0 && ++run < nruns) enter_run(run);
    };

    // ---- MFMA side
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    const int am = wm * 32 * TM + l31, bn = wn * 32 * TN + l31;     // first A / B row of this lane (tile i / j adds 32 i / 32 j)
    const int a_swz = pc_swz<KQ>(am), b_swz = pc_swz<KQ>(bn);       // rows 32 apart share the swizzle
    int nlive = 0;                                                    // column tiles of this wave inside Cout
#pragma unroll
    for (int j = 0; j < TN; ++j) nlive += (n0 + (wn * TN + j) * 32 < p.Cout) ? 1 : 0;
    nlive = __builtin_amdgcn_readfirstlane(nlive);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // Operand reads are hand-written ds_read_b128: behind a plain LDS load hipcc places `s_waitcnt vmcnt(0)` whenever the wave has an
    // LDS-DMA in flight (it must assume the read aliases the DMA's destination), which would drain the prefetch pipeline at every
    // read.  The stage being read was retired by the counted vmcnt + barrier at the top of the chunk; the DMAs in flight target
    // another stage.  Per-lane byte addresses of the NG k-groups' pieces (stage 0); stage and tile are immediate offsets.
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    uint32_t a_addr[NG], b_addr[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        a_addr[g] = smem_lds + (uint32_t)(am * KQ + ((2 * g + half) ^ a_swz)) * 16u;
        b_addr[g] = smem_lds + (uint32_t)(A_PIECES + bn * KQ + ((2 * g + half) ^ b_swz)) * 16u;
    }
    constexpr bool IMM_STAGE = (S - 1) * STAGE * 16 + (TM > TN ? TM : TN) * 32 * KQ * 16 < 65536;   // ds_read's offset field is 16 bits
    // The K loop is software-pipelined across the chunk boundary: the barrier that opens chunk c+1 (its pieces have landed, every wave
    // is done reading chunk c) sits in front of the LAST k-group of chunk c, whose operands are already in registers, and the first
    // operand reads of chunk c+1 are issued right behind it -- their LDS latency runs under that group's MFMAs instead of draining
    // the wave's MFMA chain at every chunk start (a workgroup alone on its CU -- the N <= 64 layers, the last round of every launch --
    // ran at 62 % of its chain's pace with the barrier at the chunk start; tools/conv_timeline.py).  All NI pieces of the chunk being
    // fetched are placed behind the MFMAs of the first NG-1 groups, so the vmcnt in front of that barrier is the same constant as
    // before.  NEXT: 0 = last chunk, 1 = another main step follows, 2 = a tail chunk follows (`allowed` chunks may stay in flight).
    f32x4 va[2][TM], vb[2][TN];
    if (DBG & 8) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
            for (int i = 0; i < TM; ++i) va[h2][i] = f32x4{1.f, 2.f, 3.f, 4.f};
#pragma unroll
            for (int j = 0; j < TN; ++j) vb[h2][j] = f32x4{1.f, 2.f, 3.f, 4.f};
        }
    }
    auto reads = [&](auto nl_tag, auto st_tag, auto g_tag) {
        constexpr int NL = decltype(nl_tag)::value, ST = decltype(st_tag)::value, g = decltype(g_tag)::value;
        if ((DBG & 8) || NL == 0) return;
        const uint32_t st_off = IMM_STAGE ? 0u : (uint32_t)(ST * STAGE * 16);
        constexpr int IMM = IMM_STAGE ? ST * STAGE * 16 : 0;
        const uint32_t aa = a_addr[g] + st_off, ba = b_addr[g] + st_off;
#pragma unroll
        for (int i = 0; i < TM; ++i) { f32x4 t; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(aa), "n"(IMM + i * 32 * KQ * 16)); va[g & 1][i] = t; }
#pragma unroll
        for (int j = 0; j < NL; ++j) { f32x4 t; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(ba), "n"(IMM + j * 32 * KQ * 16)); vb[g & 1][j] = t; }
    };
    // The operands of a chunk's first k-group are in registers before control leaves the straight-line code that issued their reads:
    // hipcc may copy or re-allocate the buffers at a block boundary (loop back-edge, run boundary), which must not happen to a
    // register an LDS read is still writing.
    auto land0 = [&]() {
        if (DBG & 8) return;
#pragma unroll
        for (int i = 0; i < TM; ++i) { f32x4 t = va[0][i]; asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t)); va[0][i] = t; }
#pragma unroll
        for (int j = 0; j < TN; ++j) { f32x4 t = vb[0][j]; asm volatile("" : "+v"(t)); vb[0][j] = t; }
        __builtin_amdgcn_sched_barrier(0);
    };
    // open chunk `next`: its pieces landed (this wave's; the barrier makes that true of every wave's), every wave done with the chunk before
    auto open_main = [&]() { pc_wait_vm<NI * (S - 2)>(); __builtin_amdgcn_s_barrier(); };
    auto open_tail = [&](int allowed) { pc_wait_chunks<NI, S - 2>(allowed); __builtin_amdgcn_s_barrier(); };
    auto chunk = [&](auto nl_tag, auto issue_tag, auto st_tag, auto next_tag, int allowed) {
        constexpr int NL = decltype(nl_tag)::value;
        constexpr bool ISSUE = decltype(issue_tag)::value;
        constexpr int NEXT = decltype(next_tag)::value;
        constexpr int ST = decltype(st_tag)::value, ST_I = (ST + S - 1) % S, ST_N = (ST + 1) % S;   // chunk c in stage ST; chunk c + S - 1 goes where c - 1 was
        constexpr int NSE = (NG > 1 ? NG - 1 : 1) * 4 * TM * (NL > 0 ? NL : 1);   // MFMA slots that may carry a piece: those of the first NG-1 groups
        auto handover = [&]() {                                           // in front of the last group's MFMAs
            if (NEXT == 0) return;
            if (NEXT == 1) open_main(); else open_tail(allowed);
            reads(nl_tag, std::integral_constant<int, ST_N>{}, std::integral_constant<int, 0>{});
        };
        if (ISSUE && (NL == 0 || (DBG & 16))) {                       // no MFMAs to hide behind: all pieces now
#pragma unroll
            for (int q = 0; q < NI; ++q) piece(q, ST_I);
        }
        if (NL == 0) { handover(); return; }                          // a wave with no live column only loads its share
        constexpr bool interleave = ISSUE && !(DBG & 16);
        auto group = [&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            if constexpr (g + 1 < NG) {
                reads(nl_tag, st_tag, std::integral_constant<int, g + 1>{});
                // wait for group g's operands only (the TM + NL reads of group g+1 just issued stay in flight); the operands are tied
                // to the wait so that no MFMA using them can be scheduled above it
#pragma unroll
                for (int i = 0; i < TM; ++i) { if (!(DBG & 8)) { f32x4 t = va[g & 1][i]; asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(t) : "n"(TM + NL)); va[g & 1][i] = t; } }
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i) { if (!(DBG & 8)) { f32x4 t = va[g & 1][i]; asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t)); va[g & 1][i] = t; } }
            }
#pragma unroll
            for (int j = 0; j < NL; ++j) { f32x4 t = vb[g & 1][j]; asm volatile("" : "+v"(t)); vb[g & 1][j] = t; }
            __builtin_amdgcn_sched_barrier(0);
            float xa[TM][4], xb[TN][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                f32x4 x = va[g & 1][i];
                if (SQ) x = x * x;                                    // GDN feeds x^2 (gdn.py:56)
                xa[i][0] = x.x; xa[i][1] = x.y; xa[i][2] = x.z; xa[i][3] = x.w;
            }
#pragma unroll
            for (int j = 0; j < NL; ++j) { const f32x4 y = vb[g & 1][j]; xb[j][0] = y.x; xb[j][1] = y.y; xb[j][2] = y.z; xb[j][3] = y.w; }
            if constexpr (g + 1 == NG) {                              // this wave's reads of the chunk are complete (lgkmcnt(0) above)
                __builtin_amdgcn_sched_barrier(0);
                handover();
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < NL; ++j) {
                        if (!(DBG & 1)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[i][s], xb[j][s], acc[i][j], 0, 0, 0);
                        if (interleave && (g + 1 < NG || NG == 1)) {
                            const int slot_ix = ((g * 4 + s) * TM + i) * NL + j;      // compile-time after unrolling
#pragma unroll
                            for (int q = 0; q < NI; ++q)
                                if ((q * NSE) / NI == slot_ix) { __builtin_amdgcn_sched_barrier(0); piece(q, ST_I); __builtin_amdgcn_sched_barrier(0); }
                        }
                    }
            __builtin_amdgcn_sched_barrier(0);
        };
        group(std::integral_constant<int, 0>{});
        if constexpr (NG > 1) group(std::integral_constant<int, 1>{});
        if constexpr (NG > 2) group(std::integral_constant<int, 2>{});
        if constexpr (NG > 3) group(std::integral_constant<int, 3>{});
        static_assert(NG <= 4, "k-groups per chunk");
        if (NEXT != 0) land0();                                       // the next chunk's first operands, read under the last group's MFMAs
    };
    // The chunk loop is unrolled by S so that the stage is a compile-time constant of each body (immediate LDS offsets, no address
    // VALU) while the accumulators stay in ONE register set along a straight-line path (a per-chunk switch over the stage made hipcc
    // ping-pong them between two AGPR sets with 16 v_accvgpr_mov per chunk).
    auto kloop = [&](auto nl_tag) {
        int st_i = 0, issued = 0;
        for (; issued < S - 1 && issued < nchunks; ++issued) {       // prologue: chunks 0 .. S-2 in flight
#pragma unroll
            for (int q = 0; q < NI; ++q) piece(q, st_i);
            advance();
            st_i = st_i + 1 == S ? 0 : st_i + 1;
        }
        const int n_main = nchunks - (S - 1) > 0 ? nchunks - (S - 1) : 0;
        // open chunk 0 and start its first operand reads
        if (n_main > 0) open_main(); else open_tail(nchunks - 1);
        reads(nl_tag, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        land0();
        // chunk c; what follows it decides how the next one is opened from inside its last group
        auto run_chunk = [&](auto st_tag, int c) {
            const int after = nchunks - 1 - c;                            // chunks behind this one
            if (c < n_main) {                                            // fetch chunk c + S - 1 behind its MFMAs
                if (c + 1 < n_main) chunk(nl_tag, std::true_type{}, st_tag, std::integral_constant<int, 1>{}, 0);
                else if (after > 0) chunk(nl_tag, std::true_type{}, st_tag, std::integral_constant<int, 2>{}, after - 1);
                else chunk(nl_tag, std::true_type{}, st_tag, std::integral_constant<int, 0>{}, 0);
                advance();
            } else {                                                     // one of the last S-1 chunks: nothing left to fetch
                if (after > 0) chunk(nl_tag, std::false_type{}, st_tag, std::integral_constant<int, 2>{}, after - 1);
                else chunk(nl_tag, std::false_type{}, st_tag, std::integral_constant<int, 0>{}, 0);
            }
        };
        int c = 0;
        for (; c + S <= n_main - 1; c += S) {                            // full rounds of main steps, each followed by another main step
            auto step = [&](auto st_tag) { chunk(nl_tag, std::true_type{}, st_tag, std::integral_constant<int, 1>{}, 0); advance(); };
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1 % S>{});
            if constexpr (S >= 3) step(std::integral_constant<int, 2 % S>{});
            if constexpr (S >= 4) step(std::integral_constant<int, 3 % S>{});
            if constexpr (S >= 5) step(std::integral_constant<int, 4 % S>{});
            if constexpr (S >= 6) step(std::integral_constant<int, 5 % S>{});
        }
        // what is left (at most S main chunks, then the S-1 tail chunks): c is a multiple of S, so leftover t sits in stage t % S
        auto rest = [&](auto t_tag) {
            constexpr int t = decltype(t_tag)::value;
            if (c + t < nchunks) run_chunk(std::integral_constant<int, t % S>{}, c + t);
        };
        rest(std::integral_constant<int, 0>{});
        rest(std::integral_constant<int, 1>{});
        rest(std::integral_constant<int, 2>{});
        if constexpr (2 * S - 1 > 3) rest(std::integral_constant<int, 3>{});
        if constexpr (2 * S - 1 > 4) rest(std::integral_constant<int, 4>{});
        if constexpr (2 * S - 1 > 5) rest(std::integral_constant<int, 5>{});
        if constexpr (2 * S - 1 > 6) rest(std::integral_constant<int, 6>{});
        if constexpr (2 * S - 1 > 7) rest(std::integral_constant<int, 7>{});
        if constexpr (2 * S - 1 > 8) rest(std::integral_constant<int, 8>{});
        if constexpr (2 * S - 1 > 9) rest(std::integral_constant<int, 9>{});
        if constexpr (2 * S - 1 > 10) rest(std::integral_constant<int, 10>{});
        static_assert(S <= 6, "stages");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    if (DBG & 64) tl[1] = __builtin_amdgcn_s_memrealtime();
    if (nlive == TN) kloop(std::integral_constant<int, TN>{});
    else if (nlive == 0) kloop(std::integral_constant<int, 0>{});
    else kloop(std::integral_constant<int, 1>{});                     // TN == 2 only
    if (DBG & 64) tl[2] = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_barrier();                                     // every wave is done with the stages: the epilogue reuses them
    auto stamp_out = [&]() {
        if (!(DBG & 64) || lane != 0 || blockIdx.x >= 8192 / 4) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* d = pc_dbg_stamps[blockIdx.x * 4 + wave];
        d[0] = tl[0]; d[1] = tl[1]; d[2] = tl[2]; d[3] = __builtin_amdgcn_s_memrealtime();
        d[4] = (unsigned long long)__builtin_amdgcn_s_getreg(0xF804);      // HW_REG_HW_ID
        d[5] = (unsigned long long)__builtin_amdgcn_s_getreg(0xF814);      // HW_REG_XCC_ID
        d[6] = (unsigned long long)nlive; d[7] = (unsigned long long)blockIdx.x;
    };

    // ---- epilogue, tile by tile
    // Round 3.  The first form of this epilogue was a ROLLED loop over the 16 accumulator rows of a lane (hipcc kept it rolled: dynamic
    // VGPR indexing through s_set_gpr_idx, the switch over the epilogue kind evaluated per element, two 64-bit multiplies of address
    // arithmetic and -- on the strided / PixelShuffle outputs -- two integer divisions per element): ~740 instructions per element, 6-10 us
    // per workgroup on the short-K layers where the K loop itself takes 3-8 us (tools/conv_timeline.py, profiles/r03_d_*).  Now:
    //  * the epilogue kind is a compile-time constant of the code that runs (one dispatch per workgroup);
    //  * lanes 0-31 compute the 32 rows' output offsets ONCE per tile (row table of the layer, or one division pair per ROW for the
    //    strided / PixelShuffle outputs) and park them in LDS; every lane fetches its 16 rows' offsets with four ds_read_b128;
    //  * the 16 rows are unrolled (accumulators addressed directly) and leave through buffer stores relative to the tile's first
    //    row: 32-bit offsets, rows beyond M dropped by the hardware's range check.
    // Same arithmetic on the same values in the same order: the bits do not move.
    if (nlive == 0) { stamp_out(); return; }
    // (raising the wave's priority for the epilogue -- VALU work beside the other workgroups' MFMA issue -- changed nothing: profiles/r03_g_*)
    const bool u0 = epilogue_uses_aux0(p.epi), u1 = epilogue_uses_aux1(p.epi);
    // LDS: the stages are free now.  Aux tiles of the tile being finished, 4 KB each: aux0 alone -> 4 KB per wave, aux0 + aux1 -> 8 KB per
    // wave at the start of the allocation; behind them (and behind the stages, whichever is larger: launch_uni sizes it) 64 words per
    // wave of row tables: [0..31] byte offset of the row's output pixel relative to the tile's first, [32..63] its aux pixel index.
    const uint32_t epi_bytes = !p.dense_out ? 0u : (u1 ? 32768u : (u0 ? 16384u : 0u));
    const uint32_t tab_byte0 = (uint32_t)(S * STAGE * 16) > epi_bytes ? (uint32_t)(S * STAGE * 16) : epi_bytes;
    uint32_t* rowtab_lds = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(smem) + tab_byte0) + wave * 64;
    const uint32_t rowtab_lds_addr = smem_lds + tab_byte0 + (uint32_t)wave * 256u;
    // (the unrolled form is instantiated for the one-tile waves only -- every default instantiation; the two- and four-tile waves, reachable
    // through PC_CONV_TM / PC_CONV_TN for the bit-invariance runs, keep the generic loop: 24 more epilogue bodies per tile otherwise)
    const bool fast = TM * TN == 1 && p.out_sc == 1 && (p.dense_out || !u0);
    // (dynamic indexing of the kernel-argument struct from inside a lambda makes hipcc copy the whole struct to scratch: hoisted)
    const int ooy_ph = p.ooy[phase], oox_ph = p.oox[phase];
    const float* const aux0_p = p.aux0; const float* const aux1_p = p.aux1;
    const int ld0_v = p.ld0, ld1_v = p.ld1, epi_v = p.epi;
    // every kernel argument the tile code below needs, as plain locals: the lambdas must not touch `p` (with the 24 inlined epilogue bodies
    // referring to it hipcc gave up promoting the argument struct and copied all 1.6 KB of it to scratch at kernel entry)
    const int e_Cout = p.Cout, e_M = p.M, e_dense = p.dense_out, e_perm = p.rowperm, e_ps = p.pixel_shuffle, e_osy = p.osy, e_osx = p.osx, e_Wo = p.Wo,
              e_outH = p.outH, e_outW = p.outW;
    const int64_t e_sx = p.out_sx, e_sy = p.out_sy, e_sb = p.out_sb, e_sc = p.out_sc;
    const int* const e_rowtab = p.rowtab;
    auto slow_tile = [&](auto i_tag, auto j_tag) __attribute__((always_inline)) {   // generic form: any output strides, aux tensors read from global memory
        constexpr int i = decltype(i_tag)::value, j = decltype(j_tag)::value;
        const int mb = m0 + wm * 32 * TM + i * 32, nb = n0 + (wn * TN + j) * 32;
        const int n = nb + l31;
        const float bv = (bias && n < e_Cout) ? bias[n] : 0.0f;
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int m = mb + row;
            if (m >= e_M || n >= e_Cout) continue;
            int64_t opix;
            float v = acc[i][j][r];
            if (bias) v = v + bv;
            if (e_dense) {
                opix = e_perm ? (int64_t)e_rowtab[m] : (int64_t)m;
                v = epilogue_apply(epi_v, v, u0 ? aux0_p[opix * ld0_v + n] : 0.0f, u1 ? aux1_p[opix * ld1_v + n] : 0.0f);
                outp[opix * e_sx + (int64_t)n * e_sc] = v;
                continue;
            }
            const int b = m / HoWo, rr = m - b * HoWo;
            const int oy = rr / e_Wo, ox = rr - oy * e_Wo;
            const int Y = oy * e_osy + ooy_ph, X = ox * e_osx + oox_ph;
            int nn = n, YY = Y, XX = X;
            if (e_ps) { nn = n >> 2; YY = 2 * Y + ((n >> 1) & 1); XX = 2 * X + (n & 1); }
            const int64_t pix = ((int64_t)b * e_outH + YY) * e_outW + XX;
            v = epilogue_apply(epi_v, v, u0 ? aux0_p[pix * ld0_v + nn] : 0.0f, u1 ? aux1_p[pix * ld1_v + nn] : 0.0f);
            outp[(int64_t)b * e_sb + (int64_t)YY * e_sy + (int64_t)XX * e_sx + (int64_t)nn * e_sc] = v;
        }
    };
    // DIRECT (compile-time in each copy): dense output on the GEMM's own pixel grid with rows in pixel order -- the output pixel of tile row
    // `row` is mb + row, no table needed: the row part of a store's offset is a scalar multiple of the pixel stride.  Otherwise (permuted
    // rows; strided / PixelShuffle outputs) lanes 0-31 compute the 32 rows' offsets once per tile, park them in LDS, and every lane picks
    // up the four rows of an accumulator quad with one ds_read_b128.
    auto fast_tile = [&](auto epi_tag, auto direct_tag, auto i_tag, auto j_tag) __attribute__((always_inline)) {
        constexpr int EPI = decltype(epi_tag)::value, i = decltype(i_tag)::value, j = decltype(j_tag)::value;
        constexpr bool DIRECT = decltype(direct_tag)::value;
        constexpr bool U0 = EPI == PC_EPI_RES_GELU || EPI == PC_EPI_RES || EPI == PC_EPI_GATE || EPI == PC_EPI_GDN || EPI == PC_EPI_IGDN || EPI == PC_EPI_LRP ||
                            EPI == PC_EPI_LRP_ADD || EPI == PC_EPI_LEAKY_RES;
        constexpr bool U1 = EPI == PC_EPI_GATE || EPI == PC_EPI_LRP_ADD;
        float* tile = reinterpret_cast<float*>(smem) + wave * (U1 ? 2048 : 1024);
        const int mb = m0 + wm * 32 * TM + i * 32, nb = n0 + (wn * TN + j) * 32;
        const int n = nb + l31;
        const float bv = (bias && n < e_Cout) ? bias[n] : 0.0f;
        int64_t off0 = 0;                                             // element offset of the tile's first output pixel (wave-uniform)
        if (DIRECT) off0 = (int64_t)mb * e_sx;
        else {
            // row table of this tile: every lane computes row l31's entry (both halves alike), lanes 0-31 write it
            const int m = mb + l31;
            const bool ok = m < e_M;
            int64_t off;                                              // element offset of the row's output pixel, channel 0
            int pixrel = l31;                                         // aux pixel index (permuted rows: the pixel itself, taken from the tensor start)
            if (e_dense) { pixrel = e_rowtab[ok ? m : mb]; off = (int64_t)pixrel * e_sx; }
            else {
                const int mm = ok ? m : mb;
                const int b = mm / HoWo, rr_ = mm - b * HoWo;
                const int oy = rr_ / e_Wo, ox = rr_ - oy * e_Wo;
                const int Y = oy * e_osy + ooy_ph, X = ox * e_osx + oox_ph;
                off = e_ps ? (int64_t)b * e_sb + (int64_t)(2 * Y) * e_sy + (int64_t)(2 * X) * e_sx
                                      : (int64_t)b * e_sb + (int64_t)Y * e_sy + (int64_t)X * e_sx;
                // the tile's first row is a live row (mb < M) and the rows of a tile ascend: offsets are taken relative to it
                off0 = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)off >> 32)) << 32) |
                                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)off));
            }
            const uint32_t rel = ok ? (uint32_t)((off - off0) * 4) : 0x80000000u;      // bytes; >= num_records: the store is dropped
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the previous tile's reads of the table are complete
            if (lane < 32) { rowtab_lds[lane] = rel; rowtab_lds[32 + lane] = (uint32_t)pixrel; }
        }
        // rows beyond M: DIRECT -- the descriptor ends behind the tile's last live row (a lane's own offset n*4 is smaller than a pixel
        // stride, and the range check looks at the vector offset, which holds the row part); table form -- their entry is out of range
        const int rows_live = e_M - mb < 32 ? e_M - mb : 32;
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(outp + off0, 0, DIRECT ? (int)((int64_t)rows_live * e_sx * 4) : 0x7fffffff, 0x00020000);
        // aux tiles: four LDS-DMA loads per tile and tensor into the idle stage buffers -- one memory latency per wave instead of 16
        // dependent global loads, no extra VGPRs
        if (U0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // row table written; the previous tile's reads of the aux region complete
            // the four pieces' pixel indices BEFORE the first DMA goes out: behind a plain LDS load hipcc drains vmcnt whenever an LDS-DMA
            // is in flight
            int prow4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = (k * 64 + lane) >> 3;
                prow4[k] = DIRECT ? row : (int)rowtab_lds[32 + row];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if (a == 1 && !U1) break;
                const int ld = a == 0 ? ld0_v : ld1_v;
                const float* src = (a == 0 ? aux0_p : aux1_p) + (DIRECT ? (int64_t)mb * ld : (int64_t)0) + nb;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x7fffffff, 0x00020000);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int pc = k * 64 + lane, row = pc >> 3, cq = pc & 7;
                    const bool okp = mb + row < e_M && nb + 4 * cq < e_Cout;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(tile + a * 1024 + k * 256), 16,
                                                             okp ? (prow4[k] * ld + 4 * cq) * 4 : OOB, 0, 0, 0);
                }
            }
        }
        uint32_t lane_off;                                            // bytes: this lane's part of a store's offset
        if (DIRECT) lane_off = (uint32_t)((4 * half * (int)e_sx + n) * 4);
        else if (e_ps) lane_off = (uint32_t)((((n >> 1) & 1) * (int)e_sy + (n & 1) * (int)e_sx + (n >> 2)) * 4);
        else lane_off = (uint32_t)n * 4u;
        const uint32_t sx4 = (uint32_t)e_sx * 4u;
        // this lane's rows 8q + 4*half + {0..3} are accumulator registers 4q + {0..3}; their table entries are one ds_read_b128
        u32x4 rq = {0u, 0u, 0u, 0u};
        auto read_rows = [&](int q) __attribute__((always_inline)) {
            if (DIRECT) return;
            u32x4 t;
            asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(rowtab_lds_addr + (uint32_t)half * 16u + (uint32_t)q * 32u) : "memory");
            rq = t;
        };
        read_rows(0);
        // aux tiles landed, first row offsets here (tied to the wait: nothing that uses them may be scheduled above it)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(rq) :: "memory");
        if (n < e_Cout) {
            const bool hb = bias != nullptr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float a0v[4], a1v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = e + 8 * q + 4 * half;
                    a0v[e] = U0 ? tile[row * 32 + l31] : 0.0f;
                    a1v[e] = U1 ? tile[1024 + row * 32 + l31] : 0.0f;
                }
                const u32x4 rcur = rq;
                if (q < 3) read_rows(q + 1);
                float o4[4];
                if (EPI == PC_EPI_GELU || EPI == PC_EPI_RES_GELU) {           // two elements per packed GELU (bit-identical to pc_geluf)
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        f32x2 v2 = {acc[i][j][4 * q + e], acc[i][j][4 * q + e + 1]};
                        if (hb) v2 = v2 + bv;
                        if (EPI == PC_EPI_RES_GELU) v2 = v2 + f32x2{a0v[e], a0v[e + 1]};
                        const f32x2 g2 = pc_geluf2(v2);
                        o4[e] = g2.x; o4[e + 1] = g2.y;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float o;
                    if (EPI == PC_EPI_GELU || EPI == PC_EPI_RES_GELU) o = o4[e];
                    else {
                        float v = acc[i][j][4 * q + e];
                        if (hb) v = v + bv;
                        o = epilogue_apply(EPI, v, a0v[e], a1v[e]);
                    }
                    uint32_t ro;
                    if (DIRECT) ro = (uint32_t)(8 * q + e) * sx4;
                    else ro = e == 0 ? rcur.x : (e == 1 ? rcur.y : (e == 2 ? rcur.z : rcur.w));
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rs_out, (int)(ro + lane_off), 0, 0);
                    __builtin_amdgcn_sched_barrier(0);                   // one element at a time: interleaved, the 16 GELUs cost 20 VGPRs (one wave per SIMD less); letting the
                                                                         // one-workgroup-per-CU instantiation interleave them changed nothing (profiles/r03_o_*)
                }
                if (!DIRECT && q < 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rq) :: "memory");
            }
        }
    };
    // NCHW output with PixelShuffle(2) folded into the store (the 192 -> 3 output layer in 12-column sub-pixel form: x_hat [B][3][2H][2W]).
    // With lane = column every store instruction wrote 12 scattered dwords -- a 64-byte write request per 4-byte element, 403 MB of
    // WRITE_SIZE for the 25 MB tensor (profiles/r02_z_hbm_traffic.json).  A tile's 32 rows are 32 consecutive x of one image row
    // (Wo % 32 == 0), so for a fixed (colour, row parity) the tile owns 64 CONSECUTIVE output floats: the tile goes through LDS
    // ([row][col], 33-word pitch) and each (colour, parity) pair leaves as one 256-byte store, lane = output x.
    auto nchw_ps_tile = [&](auto i_tag, auto j_tag) __attribute__((always_inline)) {
        constexpr int i = decltype(i_tag)::value, j = decltype(j_tag)::value;
        const int mb = m0 + wm * 32 * TM + i * 32, nb = n0 + (wn * TN + j) * 32;
        const int n = nb + l31;
        const float bv = (bias && n < e_Cout) ? bias[n] : 0.0f;
        float* t = reinterpret_cast<float*>(smem) + wave * 1056;      // 32 x 33 floats per wave (the stages are free)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the previous tile's reads are complete
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            float v = acc[i][j][r];
            if (bias) v = v + bv;
            t[row * 33 + l31] = epilogue_apply(epi_v, v, 0.0f, 0.0f);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // a wave's LDS operations execute in order; the tile is wave-private
        const int b = mb / HoWo, rr_ = mb - b * HoWo;
        const int oy = rr_ / e_Wo, ox0 = rr_ - oy * e_Wo;             // the tile: pixels (oy, ox0 .. ox0 + 31) of image b
        const int Y = oy * e_osy + ooy_ph, X0 = ox0 * e_osx + oox_ph;
        const int ncp = (e_Cout - nb < 32 ? e_Cout - nb : 32) >> 1;   // (colour, row parity) pairs among this tile's columns
        for (int cp = 0; cp < ncp; ++cp) {
            const int nn = nb + 2 * cp;                               // column of px = 0
            const float v = t[(lane >> 1) * 33 + 2 * cp + (lane & 1)];
            outp[(int64_t)b * e_sb + (int64_t)(nn >> 2) * e_sc + (int64_t)(2 * Y + ((nn >> 1) & 1)) * e_sy + (int64_t)(2 * X0 + lane) * e_sx] = v;
        }
    };
    auto finish = [&](auto epi_tag, auto direct_tag) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (j >= nlive) continue;
                if (m0 + wm * 32 * TM + i * 32 >= e_M) continue;
                // (compile-time tile indices: the accumulators are addressed directly)
                if (i == 0 && j == 0) fast_tile(epi_tag, direct_tag, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                if constexpr (TN > 1) { if (i == 0 && j == 1) fast_tile(epi_tag, direct_tag, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}); }
                if constexpr (TM > 1) { if (i == 1 && j == 0) fast_tile(epi_tag, direct_tag, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}); }
                if constexpr (TM > 1 && TN > 1) { if (i == 1 && j == 1) fast_tile(epi_tag, direct_tag, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}); }
            }
    };
    static_assert(TM <= 2 && TN <= 2, "tile indices of the epilogue dispatch");
    const bool nchw_ps = TM * TN == 1 && e_ps && !e_dense && !u0 && e_sx == 1 && e_osx == 1 && e_osy == 1 && e_Wo % 32 == 0 && e_M % 32 == 0 && (e_Cout & 3) == 0 &&
                         (size_t)S * STAGE * 16 >= (size_t)4 * 1056 * 4;
    if (nchw_ps) {
        if constexpr (TM * TN == 1) { if (m0 + wm * 32 < e_M) nchw_ps_tile(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); }
    } else if (!fast) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (j >= nlive) continue;
                if (m0 + wm * 32 * TM + i * 32 >= e_M) continue;
                if (i == 0 && j == 0) slow_tile(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                if constexpr (TN > 1) { if (i == 0 && j == 1) slow_tile(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}); }
                if constexpr (TM > 1) { if (i == 1 && j == 0) slow_tile(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}); }
                if constexpr (TM > 1 && TN > 1) { if (i == 1 && j == 1) slow_tile(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}); }
            }
    } else if constexpr (TM * TN == 1) {
        const bool direct = e_dense && !e_perm && (int64_t)32 * e_sx * 4 < ((int64_t)1 << 31);
        switch (epi_v) {
#define PC_EPI_CASE(E) case E: if (direct) finish(std::integral_constant<int, E>{}, std::true_type{}); else finish(std::integral_constant<int, E>{}, std::false_type{}); break;
        PC_EPI_CASE(PC_EPI_GELU) PC_EPI_CASE(PC_EPI_RES_GELU) PC_EPI_CASE(PC_EPI_RES) PC_EPI_CASE(PC_EPI_GATE) PC_EPI_CASE(PC_EPI_GDN)
        PC_EPI_CASE(PC_EPI_IGDN) PC_EPI_CASE(PC_EPI_CLAMP01) PC_EPI_CASE(PC_EPI_LRP) PC_EPI_CASE(PC_EPI_LRP_ADD) PC_EPI_CASE(PC_EPI_LEAKY)
        PC_EPI_CASE(PC_EPI_LEAKY_RES)
#undef PC_EPI_CASE
        default: if (direct) finish(std::integral_constant<int, PC_EPI_NONE>{}, std::true_type{}); else finish(std::integral_constant<int, PC_EPI_NONE>{}, std::false_type{}); break;
        }
    }
    stamp_out();
#endif
}

// ---- per-geometry row tables of the LDS-DMA kernel (see its prologue) ----
__global__ void conv_rowtab_kernel(const pc_conv_params p, int* __restrict__ tab)
{
    const int HoWo = p.Ho * p.Wo;
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < p.M; m += gridDim.x * blockDim.x) {
        const int b = m / HoWo, r = m - b * HoWo;
        const int oy = r / p.Wo, ox = r - oy * p.Wo;
        const int iy0 = oy * p.stride, ix0 = ox * p.stride;
        tab[m] = (b * p.H + iy0) * p.W + ix0;
        for (int ph = 0; ph < p.nphase; ++ph) {
            uint32_t mask = 0;
            for (int t = 0; t < p.ntap[ph]; ++t) {
                const int iy = iy0 + p.dy[ph][t], ix = ix0 + p.dx[ph][t];
                if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) mask |= 1u << t;
            }
            tab[(size_t)(1 + ph) * p.M + m] = (int)mask;
        }
    }
}

struct RowTabKey {
    int dev, B, H, W, stride, nphase, Ho, Wo, M, perm, ntap[4];
    int dy[4][PC_MAX_TAP], dx[4][PC_MAX_TAP];
    bool operator==(const RowTabKey& o) const { return std::memcmp(this, &o, sizeof(*this)) == 0; }
};

}  // namespace

// Cache of the per-geometry row tables.  One per codec object (pc_codec owns it and frees it in pc_codec_destroy) plus one
// process-wide instance for the stand-alone entry points (pc_conv2d_nhwc ...).  Hash lookup, LRU order, a byte cap: a service that
// sees thousands of distinct image sizes keeps a bounded amount of HBM.  A table is built asynchronously on the stream
// that first needs it; other streams order themselves behind its `ready` event -- no host synchronisation on the launch path.
struct pc_rowtab_cache {
    // `pins` counts the launches that were handed this table and have not been enqueued yet (conv_rowtab pins, pc_conv_launch unpins
    // right behind hipLaunchKernelGGL): a pinned table is never evicted, so another host thread's make_room cannot free it between
    // the lookup and the launch; once the launch is enqueued the device drain in front of the hipFree covers it.
    struct Entry { RowTabKey key; int* tab; size_t bytes; hipEvent_t ready; bool done; uint64_t tick; int pins; };
    std::mutex mu;
    std::unordered_multimap<uint64_t, Entry> map;
    size_t bytes = 0, cap;
    uint64_t tick = 0;
    explicit pc_rowtab_cache(size_t cap_bytes) : cap(cap_bytes) {}
    static uint64_t hash(const RowTabKey& k)
    {
        uint64_t h = 1469598103934665603ull;
        const unsigned char* p = reinterpret_cast<const unsigned char*>(&k);
        for (size_t i = 0; i < sizeof(k); ++i) { h ^= p[i]; h *= 1099511628211ull; }
        return h;
    }
    void release_all()
    {
        std::lock_guard<std::mutex> lk(mu);
        for (auto& kv : map) { (void)hipFree(kv.second.tab); (void)hipEventDestroy(kv.second.ready); }
        map.clear();
        bytes = 0;
    }
    // Called with `mu` held: unlink least-recently-used, unpinned tables until `need` more bytes fit.  The victims are only taken out of
    // the map here; the caller drains the device and frees them AFTER dropping the lock (a launch already enqueued may still read
    // them, and a device drain under the lock would stall every other lane's lookups).
    void unlink_victims(size_t need, std::vector<Entry>& victims)
    {
        while (bytes + need > cap) {
            auto victim = map.end();
            for (auto it = map.begin(); it != map.end(); ++it)
                if (it->second.pins == 0 && (victim == map.end() || it->second.tick < victim->second.tick)) victim = it;
            if (victim == map.end()) break;                // everything left is pinned: go over the cap rather than free a table in use
            bytes -= victim->second.bytes;
            victims.push_back(victim->second);
            map.erase(victim);
        }
    }
    // Never called with `mu` held.  A table belongs to the device in its key (the process-wide default cache serves every device): that
    // device is drained -- a launch already enqueued there may still read the table -- before the table is freed.
    static void free_victims(std::vector<Entry>& victims)
    {
        if (victims.empty()) return;
        int cur = 0, drained = -1;
        (void)hipGetDevice(&cur);
        for (auto& v : victims) {
            if (v.key.dev != drained) {
                (void)hipSetDevice(v.key.dev);
                (void)hipDeviceSynchronize();
                drained = v.key.dev;
            }
            (void)hipFree(v.tab);
            (void)hipEventDestroy(v.ready);
        }
        (void)hipSetDevice(cur);
        victims.clear();
    }
    void unpin(const int* tab)
    {
        std::lock_guard<std::mutex> lk(mu);
        for (auto& kv : map) if (kv.second.tab == tab) { if (kv.second.pins > 0) --kv.second.pins; return; }
    }
};

namespace {
__global__ __launch_bounds__(256) void gelu_selftest_kernel(unsigned long long* out)
{
    unsigned long long bad = 0, nanp = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 31); i += (uint64_t)gridDim.x * blockDim.x) {
        const f32x2 x = {pc_bits2f((uint32_t)(2 * i)), pc_bits2f((uint32_t)(2 * i + 1))};
        const f32x2 g = pc_geluf2(x);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float s = pc_geluf(x[c]);
            if (pc_f2bits(s) != pc_f2bits(g[c])) { if (s != s && g[c] != g[c]) ++nanp; else ++bad; }
        }
    }
    for (int off = 32; off; off >>= 1) { bad += __shfl_xor(bad, off); nanp += __shfl_xor(nanp, off); }
    if ((threadIdx.x & 63) == 0) { if (bad) atomicAdd(&out[0], bad); if (nanp) atomicAdd(&out[1], nanp); }
}
}  // namespace

// The packed GELU of the epilogue against pc_geluf over all 2^32 float arguments: *n_mismatch = arguments whose results differ in any bit
// (NaN results of both forms counted apart in *n_nan_payload: the payload a NaN operation keeps is not part of the contract).  Synchronous.
extern "C" int pc_selftest_packed_gelu(uint64_t* n_mismatch, uint64_t* n_nan_payload)
{
    if (!n_mismatch || !n_nan_payload) return PC_ERR_ARG;
    unsigned long long* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), 16) != hipSuccess || hipMemset(d, 0, 16) != hipSuccess) return PC_ERR_HIP;
    hipLaunchKernelGGL(gelu_selftest_kernel, dim3(8192), dim3(256), 0, nullptr, d);
    unsigned long long h[2] = {0, 0};
    const hipError_t e = hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return PC_ERR_HIP;
    *n_mismatch = h[0]; *n_nan_payload = h[1];
    return PC_OK;
}

pc_rowtab_cache* pc_rowtab_cache_create(size_t cap_bytes) { return new (std::nothrow) pc_rowtab_cache(cap_bytes); }
void pc_rowtab_cache_destroy(pc_rowtab_cache* c)
{
    if (!c) return;
    c->release_all();
    delete c;
}
size_t pc_rowtab_cache_bytes(pc_rowtab_cache* c) { if (!c) return 0; std::lock_guard<std::mutex> lk(c->mu); return c->bytes; }

namespace {

pc_rowtab_cache* default_rowtab_cache()
{
    static pc_rowtab_cache cache((size_t)64 << 20);       // stand-alone entry points: at most 64 MB of tables per process
    return &cache;
}

// the cached table for this layer geometry, built on first use on `stream`; nullptr = fall back to in-kernel row arithmetic.
// The table comes back PINNED: the caller unpins it (conv_rowtab_unpin) once its launch is enqueued.
const int* conv_rowtab(const pc_conv_params& p, hipStream_t stream)
{
    if ((int64_t)p.B * p.H * p.W >= (int64_t)1 << 31) return nullptr;          // pixel indices are int32 in the table
    pc_rowtab_cache* rc = p.rowtab_cache ? reinterpret_cast<pc_rowtab_cache*>(p.rowtab_cache) : default_rowtab_cache();
    RowTabKey k;
    std::memset(&k, 0, sizeof(k));
    (void)hipGetDevice(&k.dev);
    k.B = p.B; k.H = p.H; k.W = p.W; k.stride = p.stride; k.nphase = p.nphase; k.Ho = p.Ho; k.Wo = p.Wo; k.M = p.M; k.perm = p.rowperm;
    for (int ph = 0; ph < p.nphase; ++ph) {
        k.ntap[ph] = p.ntap[ph];
        for (int t = 0; t < p.ntap[ph]; ++t) { k.dy[ph][t] = p.dy[ph][t]; k.dx[ph][t] = p.dx[ph][t]; }
    }
    const uint64_t h = pc_rowtab_cache::hash(k);
    const size_t bytes = (size_t)(1 + p.nphase) * p.M * sizeof(int);
    // look the geometry up; on a hit order this stream behind the table's build and pin it
    auto lookup = [&](const int** out) -> bool {             // with rc->mu held
        auto range = rc->map.equal_range(h);
        for (auto it = range.first; it != range.second; ++it) {
            pc_rowtab_cache::Entry& e = it->second;
            if (!(e.key == k)) continue;
            e.tick = ++rc->tick;
            if (!e.done) {                                 // built on another stream a moment ago: order this stream behind it
                if (hipEventQuery(e.ready) == hipSuccess) e.done = true;
                else if (hipStreamWaitEvent(stream, e.ready, 0) != hipSuccess) { *out = nullptr; return true; }
            }
            ++e.pins;
            *out = e.tab;
            return true;
        }
        return false;
    };
    std::vector<pc_rowtab_cache::Entry> victims;
    {
        std::lock_guard<std::mutex> lk(rc->mu);
        const int* hit = nullptr;
        if (lookup(&hit)) return hit;
        if (bytes > rc->cap) return nullptr;
        rc->unlink_victims(bytes, victims);
    }
    pc_rowtab_cache::free_victims(victims);                // device drain + hipFree outside the lock
    pc_rowtab_cache::Entry e;
    e.key = k; e.bytes = bytes; e.done = false; e.tick = 0; e.tab = nullptr; e.pins = 1;
    if (hipMalloc(reinterpret_cast<void**>(&e.tab), bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipEventCreateWithFlags(&e.ready, hipEventDisableTiming) != hipSuccess) { (void)hipFree(e.tab); return nullptr; }
    if (p.rowperm) {
        // rows grouped by tap-validity pattern (interior first, then one group per border pattern): built on the host once per geometry
        // (stride 1, one phase, small images only: M <= a few 10^4), copied synchronously -- it is ready for every stream at once
        std::vector<int> host((size_t)2 * p.M);
        std::vector<std::pair<uint32_t, int>> order((size_t)p.M);
        const uint32_t full = p.ntap[0] >= 32 ? 0xffffffffu : ((1u << p.ntap[0]) - 1u);
        for (int m = 0; m < p.M; ++m) {
            const int b = m / (p.H * p.W), r = m - b * p.H * p.W, y = r / p.W, x = r - y * p.W;
            uint32_t mask = 0;
            for (int t = 0; t < p.ntap[0]; ++t) {
                const int iy = y + p.dy[0][t], ix = x + p.dx[0][t];
                if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) mask |= 1u << t;
            }
            order[(size_t)m] = {mask == full ? 0u : mask, m};          // sort key: interior rows first
        }
        std::stable_sort(order.begin(), order.end(), [](const std::pair<uint32_t, int>& a, const std::pair<uint32_t, int>& b2) { return a.first < b2.first; });
        for (int m = 0; m < p.M; ++m) {
            host[(size_t)m] = order[(size_t)m].second;
            host[(size_t)p.M + m] = (int)(order[(size_t)m].first == 0u ? full : order[(size_t)m].first);
        }
        if (hipMemcpy(e.tab, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(e.tab); (void)hipEventDestroy(e.ready); return nullptr; }
        e.done = true;
    } else {
        hipLaunchKernelGGL(conv_rowtab_kernel, dim3(std::min(2048, (p.M + 255) / 256)), dim3(256), 0, stream, p, e.tab);
        if (hipGetLastError() != hipSuccess || hipEventRecord(e.ready, stream) != hipSuccess) { (void)hipFree(e.tab); (void)hipEventDestroy(e.ready); return nullptr; }
    }
    const int* other = nullptr;
    {
        std::lock_guard<std::mutex> lk(rc->mu);
        if (!(lookup(&other) && other)) {
            e.tick = ++rc->tick;
            rc->bytes += bytes;
            rc->map.emplace(h, e);
            return e.tab;
        }
    }
    // another lane built the same geometry while this one was building outside the lock: use (and pin) that table.  Ours is read by
    // nothing but its own build kernel: dropped after that kernel has run -- the device drain happens here, with the lock released.
    std::vector<pc_rowtab_cache::Entry> mine{e};
    pc_rowtab_cache::free_victims(mine);
    return other;
}

void conv_rowtab_unpin(const pc_conv_params& p)
{
    if (!p.rowtab) return;
    pc_rowtab_cache* rc = p.rowtab_cache ? reinterpret_cast<pc_rowtab_cache*>(p.rowtab_cache) : default_rowtab_cache();
    rc->unpin(p.rowtab);
}

template <int BK, int S, int TM, int TN, bool SQ = false, int DBG = 0>
hipError_t launch_uni(const pc_conv_params& p, hipStream_t stream)
{
    constexpr int BM = 64 * TM, BN = 64 * TN;
    int tmax = 0;
    for (int ph = 0; ph < p.nphase; ++ph) tmax = std::max(tmax, p.ntap[ph]);
    const bool u1 = p.epi == PC_EPI_GATE || p.epi == PC_EPI_LRP_ADD;                         // epilogue_uses_aux1
    const bool u0 = u1 || p.epi == PC_EPI_RES_GELU || p.epi == PC_EPI_RES || p.epi == PC_EPI_GDN || p.epi == PC_EPI_IGDN || p.epi == PC_EPI_LRP ||
                    p.epi == PC_EPI_LEAKY_RES;                                                 // epilogue_uses_aux0
    const size_t stage_bytes = (size_t)S * (BM + BN) * (BK / 4) * 16, epi_bytes = !p.dense_out ? 0 : (u1 ? 32768 : (u0 ? 16384 : 0));
    // stages (or the epilogue's aux tiles, whichever is larger), then the run table of the K loop / the epilogue's row tables (256 B per wave)
    const size_t lds = std::max(stage_bytes, epi_bytes) + std::max((size_t)tmax * p.nseg * 2 * sizeof(pc_run), (size_t)1024);
    const int MT = (p.M + BM - 1) / BM, NT = (p.Cout + BN - 1) / BN, NZ = p.ngroup == 2 ? 2 : p.nphase;
    dim3 grid(8 * ((MT + 7) / 8) * NT * NZ, 1, 1);
    auto kern = conv_igemm_uni_kernel<BK, S, TM, TN, SQ, DBG>;
    static std::atomic<uint32_t> attr_set{0};             // bit per device: more than 64 KB of dynamic LDS allowed for this instantiation
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (lds > 48 * 1024 && !(attr_set.load(std::memory_order_acquire) & (1u << (dev & 31)))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess)
            return hipGetLastError();
        attr_set.fetch_or(1u << (dev & 31), std::memory_order_release);
    }
    static std::atomic<int> printed{0};
    if ((p.dbg & 256) && printed.fetch_add(1) == 0) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds);
        fprintf(stderr, "[pc_conv] uni<%d,%d,%d,%d> grid %u, lds %zu B, max active blocks per CU %d\n", BK, S, TM, TN, grid.x, lds, nb);
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p);
    return hipGetLastError();
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool SMALLC>
hipError_t launch_cfg(const pc_conv_params& p, hipStream_t stream)
{
    dim3 grid((p.M + BM - 1) / BM, (p.Cout + BN - 1) / BN, p.nphase);
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N, SMALLC>), grid, dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace

extern "C" __attribute__((visibility("default"))) int pc_debug_read_stamps(unsigned long long* dst, int nblocks)
{
    if (!dst || nblocks <= 0 || nblocks > 8192) return PC_ERR_ARG;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(pc_dbg_stamps), sizeof(unsigned long long) * 16 * (size_t)nblocks) == hipSuccess ? PC_OK : PC_ERR_HIP;
}

int pc_conv_weight_layout(int kind, int Cin, int Cout, int k)
{
    const long ktot = (long)(kind == 0 ? k * k : 9) * Cin;     // largest K of any output phase
    return (Cin % 16 == 0 && Cout > 4 && ktot >= 64) ? 1 : 0;
}

// Host-side validation + tile selection.  Returns a pc status code.
int pc_conv_launch(const pc_conv_params& p_in, hipStream_t stream)
{
    pc_conv_params p = p_in;                               // derived fields (row table, fast-path flags, debug bits) are set on a private copy
    p.rowtab = nullptr;
    // the row table comes back pinned in its cache; it is unpinned when this function returns, i.e. once the launch is enqueued (from then
    // on the device drain in front of an eviction's hipFree protects it) or on any early error return
    struct Unpin { const pc_conv_params& q; ~Unpin() { conv_rowtab_unpin(q); } } unpin_guard{p};
    if (p.nphase < 1 || p.nphase > 4 || p.M <= 0 || p.Cout <= 0 || p.Cin <= 0) return PC_ERR_ARG;
    static const int dbg_env = (int)pc_tune("PC_CONV_DBG", 0);
    if (dbg_env) p.dbg = dbg_env;
    {   // row tables only pay for layers with several taps (1x1 layers: one tap, always valid)
        int tmax = 0;
        for (int ph = 0; ph < p.nphase; ++ph) tmax = std::max(tmax, p.ntap[ph]);
        static const bool no_tab = pc_tune("PC_CONV_NO_ROWTAB", 0) != 0;
        // Padding taps (PC_CONV_ROWPERM, default on): on small images a 3x3 window multiplies zero padding for 8 % (16x16), 16 % (8x8),
        // 30 % (4x4) of its MACs.  A tile of consecutive pixels always mixes border and interior rows; with the rows grouped by
        // tap-validity pattern a border tile skips its padding taps as whole runs.  A chain never holds -0 (it starts at +0, exact
        // cancellation rounds to +0), so dropping fmaf(0, w, acc) terms keeps the bits.  Stride 1, "same" output grid, one phase,
        // unified kernel, dense NHWC output, 32-bit piece offsets.
        static const bool perm_on = pc_tune("PC_CONV_ROWPERM", 1) != 0;
        static const bool uni_on = pc_tune("PC_CONV_KERN", 1) == 1;
        static const long perm_min_blocks = pc_tune("PC_CONV_ROWPERM_MIN", 256L);
        int maxld = 0;
        for (int sg = 0; sg < p.nseg; ++sg) maxld = std::max(maxld, p.seg[sg].ld);
        maxld = std::max(maxld, std::max(p.ld0, p.ld1));
        maxld = std::max(maxld, (int)std::min<int64_t>(p.out_sx, 0x7fffffff));   // the epilogue's output offsets of permuted rows are 32-bit too
        const bool dense = p.nphase == 1 && p.osy == 1 && p.osx == 1 && p.ooy[0] == 0 && p.oox[0] == 0 && p.outH == p.Ho && p.outW == p.Wo &&
                           !p.pixel_shuffle && p.out_sy == (int64_t)p.outW * p.out_sx && p.out_sb == (int64_t)p.outH * p.out_sy;
        p.rowperm = (perm_on && uni_on && !no_tab && p.wlayout == 1 && !p.square && !p.smallc && tmax > 1 && tmax <= 25 && p.nphase == 1 && p.stride == 1 &&
                     p.Ho == p.H && p.Wo == p.W && p.H <= 32 && p.W <= 32 && dense && !(p.dbg & 64) &&
                     (long)((p.M + 63) / 64) * ((p.Cout + 63) / 64) * (p.ngroup == 2 ? 2 : 1) > perm_min_blocks &&   // one block per CU: the launch lasts as long as its interior tiles
                     (int64_t)p.B * p.H * p.W * (int64_t)std::max(maxld, 1) * 4 < ((int64_t)1 << 31)) ? 1 : 0;
        p.rowtab = (p.wlayout == 1 && tmax > 1 && !no_tab) ? conv_rowtab(p, stream) : nullptr;
        if (!p.rowtab) p.rowperm = 0;
    }
    static const bool group_xcd_on = pc_tune("PC_CONV_GROUP_XCD", 1) != 0;
    p.group_xcd = (group_xcd_on && p.ngroup == 2 && p.wlayout == 1) ? 1 : 0;
    p.ident_rows = p.nphase == 1 && p.ntap[0] == 1 && p.dy[0][0] == 0 && p.dx[0][0] == 0 && p.stride == 1 &&
                                                p.Ho == p.H && p.Wo == p.W;
    p.dense_out = p.nphase == 1 && p.osy == 1 && p.osx == 1 && p.ooy[0] == 0 && p.oox[0] == 0 && p.outH == p.Ho &&
                                               p.outW == p.Wo && !p.pixel_shuffle && p.out_sy == (int64_t)p.outW * p.out_sx &&
                                               p.out_sb == (int64_t)p.outH * p.out_sy;
    if (!p.smallc) {
        int c = 0;
        if (p.nseg < 1 || p.nseg > PC_MAX_SEG) return PC_ERR_ARG;
        for (int s = 0; s < p.nseg; ++s) {
            if (p.seg[s].nch <= 0 || p.seg[s].nch % 16 || p.seg[s].ld % 4 || ((uintptr_t)p.seg[s].ptr & 15)) return PC_ERR_ARG;
            c += p.seg[s].nch;
        }
        if (c != p.Cin) return PC_ERR_ARG;
        if (p.Cout % 4 && p.Cout > 4) return PC_ERR_ARG;
    } else if (p.nseg != 1) return PC_ERR_ARG;
    for (int ph = 0; ph < p.nphase; ++ph)
        if (p.ntap[ph] < 1 || p.ntap[ph] > PC_MAX_TAP) return PC_ERR_ARG;
    if (p.pixel_shuffle && (p.Cout % 4)) return PC_ERR_ARG;

    // Kernel selection (measured on MI355X with tools/conv_tune.py; profiles/r01_*):
    //  * weight layout 1 (chosen at pack time by pc_conv_weight_layout: Cin % 16 == 0, Cout > 4; GDN's gamma is always layout 1):
    //    wave-specialised 64x64 LDS-DMA kernel, K-chunk 32, three LDS stages;
    //  * weight layout 0: the plain BK=16 kernel -- the 3-channel output layer and the 3-channel input layer (element-gather
    //    loader).  (1x1 convs and GDN moved to the LDS-DMA kernel: 17-25 % faster, profiles/r01_tune_tune27.log.)
    hipError_t e;
    if (p.fg_gamma) {
        // the input layer with its GDN fused (conv_igemm_in_gdn_kernel): 3 -> 192, one phase, dense NHWC output of 192 channels
        if (!p.smallc || p.wlayout != 0 || p.Cout != 192 || p.nphase != 1 || !p.dense_out || p.out_sc != 1 || !p.fg_beta || p.epi != PC_EPI_NONE ||
            p.ngroup == 2 || p.ntap[0] * p.Cin > 80)
            return PC_ERR_ARG;
        constexpr size_t lds = (size_t)(64 * 196 + 2 * 192 * 16 + 2 * 80) * sizeof(float);
        if ((int64_t)3 * p.H * p.W >= ((int64_t)1 << 31)) return PC_ERR_ARG;             // (element offsets inside one image are 32-bit)
        static std::atomic<uint32_t> attr_set{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!(attr_set.load(std::memory_order_acquire) & (1u << (dev & 31)))) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_in_gdn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return PC_ERR_HIP;
            attr_set.fetch_or(1u << (dev & 31), std::memory_order_release);
        }
        hipLaunchKernelGGL(conv_igemm_in_gdn_kernel, dim3((unsigned)((p.M + 63) / 64)), dim3(256), lds, stream, p);
        return hipGetLastError() == hipSuccess ? PC_OK : PC_ERR_HIP;
    }
    if (p.wlayout == 1) {
        if (p.smallc || (p.Cin % 16)) return PC_ERR_ARG;
        if (p.ngroup == 2 && (p.nphase != 1 || !p.g1_seg0 || !p.g1_w || !p.g1_out)) return PC_ERR_ARG;
        // Measured on MI355X (tools/conv_tune.py; profiles/r01_tune_tune18.log, tune25, tune26): K-chunk 32 with three LDS stages
        // (two chunks in flight) wins or ties on every layer shape of the codec, small and large grids alike -- 64-channel
        // chunks, 2 or 4 stages and the 32x64 / 64x32 / 32x32 block tiles (WM, WN < 2) were all slower or equal.  The other
        // instantiations stay reachable for tuning through PC_CONV_BK / PC_CONV_S.
        static const int bk_env = (int)pc_tune("PC_CONV_BK", 0);
        static const int s_env = (int)pc_tune("PC_CONV_S", 0);
        int chunks = 0;                                    // K-loop length of the longest phase, in 32-channel chunks
        for (int ph = 0; ph < p.nphase; ++ph) {
            int c = 0;
            for (int sg = 0; sg < p.nseg; ++sg) c += (p.seg[sg].nch + 31) / 32;
            chunks = std::max(chunks, c * p.ntap[ph]);
        }
        const int bk = bk_env ? bk_env : 32;
        const int S = s_env ? s_env : (chunks <= 8 ? 2 : 3);   // 1x1 layers with K <= 256: two stages (nothing to keep in flight)
        // kernel family: 0 = wave-specialised (round 1), 1 = unified (every wave loads and multiplies)
        static const int kern_env = (int)pc_tune("PC_CONV_KERN", 1);
        static const int tm_env = (int)pc_tune("PC_CONV_TM", 0);
        static const int tn_env = (int)pc_tune("PC_CONV_TN", 0);
        e = hipErrorInvalidValue;
        if (kern_env == 1) {
            // Kernel choice (PC_CONV_POLICY; tools/conv_tune.py, bench A/Bs: profiles/r02_l_*, r02_n_*, r02_o_*).  3 (default): K-chunk 16,
            // three stages, 64x64 blocks -- 24 KB of LDS and 80 registers, up to six workgroups per CU -- for every layer with more than
            // PC_CONV_SMALL_THR (256) blocks, i.e. more than one per CU; K-chunk 32 / three stages (two for K <= 256) for the small grids,
            // where a workgroup is alone on its CU and the deeper prefetch per barrier counts.  1: K-chunk 16 everywhere.  2: K-chunk 16
            // except 128x64 two-accumulator blocks on the large layers.  0: the first half of round 2 (K-chunk 32; 128x64 blocks with two
            // stages once there are >= 1536 of the 64x64 blocks).  All of them produce identical bits.
            const long nb64 = (long)((p.M + 63) / 64) * ((p.Cout + 63) / 64) * (p.ngroup == 2 ? 2 : p.nphase);
            static const int policy = (int)pc_tune("PC_CONV_POLICY", 3);
            static const long small_thr = pc_tune("PC_CONV_SMALL_THR", 256L);
            static const long tm_thr = pc_tune("PC_CONV_TM_THR", 1536L);
            int tm = tm_env ? tm_env : (nb64 >= tm_thr ? 2 : 1), tn = tn_env ? tn_env : 1;
            const int Su = s_env ? s_env : ((chunks <= 8 || tm * tn > 1) ? 2 : 3);
            const int ab = (p.dbg & 64) ? 0 : (p.dbg & 15);                   // ablation builds of the 64x64 three-stage instantiation
            if (p.square) e = launch_uni<32, 2, 1, 1, true>(p, stream);
#ifdef PC_CONV_TUNING   // ablation builds of the 64x64 three-stage instantiation (make FLAGS+=-DPC_CONV_TUNING): no MFMAs / no DMA issue / zero-fill DMAs / no operand reads ...
            else if (ab == 1) e = launch_uni<32, 3, 1, 1, false, 1>(p, stream);
            else if (ab == 2) e = launch_uni<32, 3, 1, 1, false, 2>(p, stream);
            else if (ab == 4) e = launch_uni<32, 3, 1, 1, false, 4>(p, stream);
            else if (ab == 8) e = launch_uni<32, 3, 1, 1, false, 8>(p, stream);
            else if (ab == 10) e = launch_uni<32, 3, 1, 1, false, 10>(p, stream);
            else if (ab == 3) e = launch_uni<32, 3, 1, 1, false, 3>(p, stream);
            else if (ab == 0 && (p.dbg & 16) && !(p.dbg & 64)) e = launch_uni<32, 3, 1, 1, false, 16>(p, stream);
            else if ((p.dbg & 64) && (p.dbg & 8)) e = launch_uni<32, 3, 1, 1, false, 74>(p, stream);
#endif
            else if ((p.dbg & 64) && bk == 16) e = launch_uni<16, 3, 1, 1, false, 64>(p, stream);   // timeline of the default large-grid instantiation (PC_CONV_BK=16)
            else if (p.dbg & 64) e = launch_uni<32, 3, 1, 1, false, 64>(p, stream);
            else if (bk == 16 && tm == 1 && tn == 1 && s_env == 2) e = launch_uni<16, 2, 1, 1>(p, stream);
            else if (bk == 16 && tm == 1 && tn == 1 && s_env == 3) e = launch_uni<16, 3, 1, 1>(p, stream);
            else if (bk == 16 && tm == 1 && tn == 1) e = launch_uni<16, 4, 1, 1>(p, stream);
            else if (policy == 1 && !s_env && !tm_env && !tn_env) e = launch_uni<16, 3, 1, 1>(p, stream);
            else if (policy == 2 && !s_env && !tm_env && !tn_env && tm == 1) e = launch_uni<16, 3, 1, 1>(p, stream);
            else if (policy == 3 && !s_env && !tm_env && !tn_env && nb64 > small_thr) e = launch_uni<16, 3, 1, 1>(p, stream);
            else if (policy == 3 && !s_env && !tm_env && !tn_env) e = chunks <= 8 ? launch_uni<32, 2, 1, 1>(p, stream) : launch_uni<32, 3, 1, 1>(p, stream);
#define PC_UNI_CASE(S_, TM_, TN_) else if (Su == S_ && tm == TM_ && tn == TN_) e = launch_uni<32, S_, TM_, TN_>(p, stream);
            PC_UNI_CASE(3, 1, 1) PC_UNI_CASE(2, 1, 1) PC_UNI_CASE(3, 2, 1) PC_UNI_CASE(2, 2, 1) PC_UNI_CASE(2, 1, 2)
            PC_UNI_CASE(2, 2, 2)
#undef PC_UNI_CASE
        }
        if (e == hipErrorInvalidValue) return PC_ERR_ARG;
    } else {
        if (p.ngroup == 2) return PC_ERR_ARG;
        int cfg = p.tile_cfg;
        if (cfg == PC_TILE_AUTO) cfg = (p.Cout <= 4) ? PC_TILE_128x32 : PC_TILE_64x64;
        switch (cfg) {
        case PC_TILE_128x128: e = p.smallc ? launch_cfg<128, 128, 2, 2, true>(p, stream) : launch_cfg<128, 128, 2, 2, false>(p, stream); break;
        case PC_TILE_64x64: e = p.smallc ? launch_cfg<64, 64, 2, 2, true>(p, stream) : launch_cfg<64, 64, 2, 2, false>(p, stream); break;
        case PC_TILE_128x32: e = p.smallc ? launch_cfg<128, 32, 4, 1, true>(p, stream) : launch_cfg<128, 32, 4, 1, false>(p, stream); break;
        default: return PC_ERR_ARG;
        }
    }
    return e == hipSuccess ? PC_OK : PC_ERR_HIP;
}
