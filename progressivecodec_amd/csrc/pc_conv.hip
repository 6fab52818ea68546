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
// element is ONE fmaf chain over (tap, channel) ascending starting at +0 -- exactly what
// v_mfma_f32_32x32x2_f32 computes when the K loop feeds it in order and nothing is split
// along K -- so results are bit-identical for every tile shape, batch size and device, and
// equal to oracle/pc_oracle.c:orc_conv_nhwc.
//
// Tiling: 256 threads = 4 waves (64 lanes); block tile BM x BN; K-chunk 16; each wave owns
// TM x TN MFMA tiles of 32x32 (16 accumulator VGPRs each); A/B chunks are register-staged
// global->LDS with double buffering (one barrier per chunk); LDS images are k-major
// ([k][m] / [k][n]) so every ds_read_b32 of an MFMA operand is bank-conflict-free.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cstdlib>

#include "../../include/pc_math.h"
#include "pc_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ float epilogue_value(const pc_conv_params& p, float v, int64_t pix, int n)
{
    switch (p.epi) {
    case PC_EPI_NONE: return v;
    case PC_EPI_GELU: return pc_geluf(v);
    case PC_EPI_RES_GELU: return pc_geluf(v + p.aux0[pix * p.ld0 + n]);
    case PC_EPI_RES: return p.aux0[pix * p.ld0 + n] + v;
    case PC_EPI_GATE: return p.aux0[pix * p.ld0 + n] * pc_sigmoidf(v) + p.aux1[pix * p.ld1 + n];
    case PC_EPI_GDN: return p.aux0[pix * p.ld0 + n] * pc_rsqrtf(v);
    case PC_EPI_IGDN: return p.aux0[pix * p.ld0 + n] * sqrtf(v);
    case PC_EPI_CLAMP01: return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    case PC_EPI_LRP: return p.aux0[pix * p.ld0 + n] + 0.5f * pc_tanhf(v);
    case PC_EPI_LRP_ADD: return (p.aux0[pix * p.ld0 + n] + 0.5f * pc_tanhf(v)) + p.aux1[pix * p.ld1 + n];
    default: return v;
    }
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
        for (int kk = 0; kk < BK; kk += 2) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = a[(kk + half) * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = b[(kk + half) * LDB + j * 32];
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
// Main kernel (all layers with Cin % 16 == 0).  Differences from the element-gather kernel above:
//  * K-chunk BK = 16 or 32 channels of ONE tap and ONE segment (wave-uniform iterator: tap,
//    segment, channel live in SGPRs, parameters come through scalar loads); a segment tail shorter
//    than BK is zero-filled (zeros leave the fmaf chain unchanged);
//  * two chunks in flight in registers (prefetch distance 2) on top of the LDS double buffer: a
//    global-load round trip (~1.1 us measured per chunk on the M = 8192 slice-chain GEMMs, which
//    have <= 2 blocks per CU) is then covered by two MFMA phases instead of one;
//  * register budget kept low (<= 128 VGPRs) because occupancy is what hides the per-chunk barrier.
// The fmaf chain per output element is unchanged: chunks, and k inside a chunk, ascend.
// ------------------------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N, int BK>
__global__ __launch_bounds__(256) void conv_igemm2_kernel(const pc_conv_params p)
{
    constexpr int TM = BM / WAVES_M / 32, TN = BN / WAVES_N / 32;
    constexpr int LDA = BM + 1, LDB = BN + 4;
    constexpr int KQ = BK / 4;                        // 16-byte k-quads per A row
    constexpr int ROWS_PER_PASS = 256 / KQ;
    constexpr int AI = (BM + ROWS_PER_PASS - 1) / ROWS_PER_PASS;   // A float4 per thread per chunk
    constexpr int BQ = BN / 4;                        // float4 per B row
    constexpr int BROWS_PER_PASS = 256 / BQ;
    constexpr int BI = (BK + BROWS_PER_PASS - 1) / BROWS_PER_PASS;
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

    int chunks_per_tap = 0;
    for (int s = 0; s < p.nseg; ++s) chunks_per_tap += (p.seg[s].nch + BK - 1) / BK;
    const int nchunks = T * chunks_per_tap;

    // ---- A: fixed rows, k-quad column kq
    const int kq = tid % KQ;
    const int arow0 = tid / KQ;
    int a_iy0[AI], a_ix0[AI];
    int64_t a_boff[AI];
    bool a_ok[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int row = arow0 + i * ROWS_PER_PASS;
        const int m = m0 + row;
        a_ok[i] = (row < BM) && (m < p.M);
        const int mm = a_ok[i] ? m : 0;
        const int b = mm / HoWo, r = mm - b * HoWo;
        const int oy = r / p.Wo, ox = r - oy * p.Wo;
        a_iy0[i] = oy * p.stride;
        a_ix0[i] = ox * p.stride;
        a_boff[i] = (int64_t)b * p.H * p.W;
    }
    // ---- B: rows bk0 + i * BROWS_PER_PASS, columns b_n
    const int b_n = (tid % BQ) * 4;
    const int bk0 = tid / BQ;
    const bool b_nok = n0 + b_n < p.Cout;
    const bool b_full = n0 + b_n + 3 < p.Cout;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // wave-uniform chunk iterator
    int it_t = 0, it_s = 0, it_c = 0, it_cg = 0, it_sbase = 0;

    auto load_chunk = [&](float4 (&ra)[AI], float4 (&rb)[BI]) {
        const int dy = p.dy[phase][it_t], dx = p.dx[phase][it_t];
        const int nch = p.seg[it_s].nch;
        const float* sp = p.seg[it_s].ptr + it_c + 4 * kq;
        const int sld = p.seg[it_s].ld;
        const bool kq_ok = it_c + 4 * kq < nch;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kq_ok && a_ok[i] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                v = *reinterpret_cast<const float4*>(sp + (a_boff[i] + (int64_t)iy * p.W + ix) * sld);
            if (p.square) { v.x *= v.x; v.y *= v.y; v.z *= v.z; v.w *= v.w; }
            ra[i] = v;
        }
        const float* wrow = p.w + ((int64_t)p.wtap[phase][it_t] * p.Cin + it_cg) * p.Cout + n0 + b_n;
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int k = bk0 + i * BROWS_PER_PASS;
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < BK && it_c + k < nch && b_nok) {
                const float* wp = wrow + (int64_t)k * p.Cout;
                if (b_full) w = *reinterpret_cast<const float4*>(wp);
                else { w.x = wp[0]; if (n0 + b_n + 1 < p.Cout) w.y = wp[1]; if (n0 + b_n + 2 < p.Cout) w.z = wp[2]; }
            }
            rb[i] = w;
        }
        it_c += BK; it_cg += BK;
        if (it_c >= nch) {
            it_sbase += nch; it_c = 0; it_cg = it_sbase; ++it_s;
            if (it_s >= p.nseg) { it_s = 0; it_sbase = 0; it_cg = 0; ++it_t; }
        }
    };

    auto store_chunk = [&](int buf, const float4 (&ra)[AI], const float4 (&rb)[BI]) {
        float* a = As + buf * BK * LDA;
        float* b = Bs + buf * BK * LDB;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int row = arow0 + i * ROWS_PER_PASS;
            if (row < BM) {
                a[(kq * 4 + 0) * LDA + row] = ra[i].x;
                a[(kq * 4 + 1) * LDA + row] = ra[i].y;
                a[(kq * 4 + 2) * LDA + row] = ra[i].z;
                a[(kq * 4 + 3) * LDA + row] = ra[i].w;
            }
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int k = bk0 + i * BROWS_PER_PASS;
            if (k < BK) *reinterpret_cast<float4*>(b + k * LDB + b_n) = rb[i];
        }
    };

    const int half = lane >> 5, l31 = lane & 31;
    auto compute = [&](int buf) {
        // operands of G k-steps are read into registers one group ahead of the MFMAs that consume them, so the
        // LDS latency of group g+1 hides behind the 64-cycle MFMAs of group g
        constexpr int G = 8, NG = BK / (2 * G);
        const float* a = As + buf * BK * LDA + wm * (TM * 32) + l31 + half * LDA;
        const float* b = Bs + buf * BK * LDB + wn * (TN * 32) + l31 + half * LDB;
        float av[2][G][TM], bv[2][G][TN];
        auto rd = [&](int g, int slot) {
#pragma unroll
            for (int s = 0; s < G; ++s) {
                const int kk = 2 * (g * G + s);
#pragma unroll
                for (int i = 0; i < TM; ++i) av[slot][s][i] = a[kk * LDA + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[slot][s][j] = b[kk * LDB + j * 32];
            }
        };
        rd(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) rd(g + 1, (g + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);     // keep the next group's LDS reads ahead of this group's MFMAs
#pragma unroll
            for (int s = 0; s < G; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][s][i], bv[g & 1][s][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    float4 ra0[AI], rb0[BI], ra1[AI], rb1[BI];
    load_chunk(ra0, rb0);
    if (nchunks > 1) load_chunk(ra1, rb1);
    store_chunk(0, ra0, rb0);
    __syncthreads();
    for (int c = 0;;) {
        if (c + 2 < nchunks) load_chunk(ra0, rb0);
        compute(0);
        if (c + 1 < nchunks) store_chunk(1, ra1, rb1);
        __syncthreads();
        if (++c >= nchunks) break;
        if (c + 2 < nchunks) load_chunk(ra1, rb1);
        compute(1);
        if (c + 1 < nchunks) store_chunk(0, ra0, rb0);
        __syncthreads();
        if (++c >= nchunks) break;
    }

    // ---- epilogue
#pragma unroll
    for (int i = 0; i < TM; ++i) {
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

template <int BM, int BN, int WAVES_M, int WAVES_N, int BK>
hipError_t launch_cfg2(const pc_conv_params& p, hipStream_t stream)
{
    dim3 grid((p.M + BM - 1) / BM, (p.Cout + BN - 1) / BN, p.nphase);
    hipLaunchKernelGGL((conv_igemm2_kernel<BM, BN, WAVES_M, WAVES_N, BK>), grid, dim3(256), 0, stream, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Wave-specialised kernel: 512 threads = 4 MFMA waves + 4 loader waves (one of each per SIMD).
// PMC profile of the kernels above (profiles/r01_c_pmc_*.txt): per K-chunk a wave spends ~1000 cycles
// in its 16 dependent MFMAs and ~2500 in ~225 VALU/SALU/LDS/VMEM instructions and their waits, all
// in phase with its SIMD partner, so the matrix pipe idles ~55 % of the time.  Here the MFMA waves
// only read operands from LDS and issue MFMAs; the loader waves do the im2col address arithmetic,
// global loads (prefetch distance 2, registers) and LDS stores for the next chunk on the VALU/VMEM/LDS
// pipes, which run beside the matrix pipe.  One block-wide barrier per chunk hands stage (c+1)%2 to the
// MFMA waves.  Same fmaf chains as every other variant.
// ------------------------------------------------------------------------------------------
template <int TM, int TN, int BK>
__global__ __launch_bounds__(512) void conv_igemm_ws_kernel(const pc_conv_params p)
{
    constexpr int WAVES_M = 2, WAVES_N = 2;
    constexpr int BM = 32 * TM * WAVES_M, BN = 32 * TN * WAVES_N;
    constexpr int LDA = BM + 1, LDB = BN + 4;
    constexpr int KQ = BK / 4;
    constexpr int ROWS_PER_PASS = 256 / KQ;
    constexpr int AI = (BM + ROWS_PER_PASS - 1) / ROWS_PER_PASS;
    constexpr int BQ = BN / 4;
    constexpr int BROWS_PER_PASS = 256 / BQ;
    constexpr int BI = (BK + BROWS_PER_PASS - 1) / BROWS_PER_PASS;
    __shared__ float smem[2 * BK * (LDA + LDB)];
    float* As = smem;
    float* Bs = smem + 2 * BK * LDA;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool loader = wave >= 4;
    const int phase = blockIdx.z;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int T = p.ntap[phase];
    const int HoWo = p.Ho * p.Wo;
    int chunks_per_tap = 0;
    for (int s = 0; s < p.nseg; ++s) chunks_per_tap += (p.seg[s].nch + BK - 1) / BK;
    const int nchunks = T * chunks_per_tap;

    if (loader) {
        const int tid = threadIdx.x - 256;
        const int kq = tid % KQ, arow0 = tid / KQ;
        int a_iy0[AI], a_ix0[AI];
        int64_t a_pix[AI];
        bool a_ok[AI];
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int row = arow0 + i * ROWS_PER_PASS;
            const int m = m0 + row;
            a_ok[i] = (row < BM) && (m < p.M);
            const int mm = a_ok[i] ? m : 0;
            const int b = mm / HoWo, r = mm - b * HoWo;
            const int oy = r / p.Wo, ox = r - oy * p.Wo;
            a_iy0[i] = oy * p.stride;
            a_ix0[i] = ox * p.stride;
            a_pix[i] = ((int64_t)b * p.H + a_iy0[i]) * p.W + a_ix0[i];      // pixel index of tap (0,0)
        }
        const int b_n = (tid % BQ) * 4, bk0 = tid / BQ;
        const bool b_nok = n0 + b_n < p.Cout, b_full = n0 + b_n + 3 < p.Cout;
        int it_t = 0, it_s = 0, it_c = 0, it_cg = 0, it_sbase = 0;

        auto load_chunk = [&](float4 (&ra)[AI], float4 (&rb)[BI]) {
            const int dy = p.dy[phase][it_t], dx = p.dx[phase][it_t];
            const int nch = p.seg[it_s].nch;
            const int sld = p.seg[it_s].ld;
            const float* sp = p.seg[it_s].ptr + (int64_t)(dy * p.W + dx) * sld + it_c + 4 * kq;
            const bool kq_ok = it_c + 4 * kq < nch;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kq_ok && a_ok[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                    v = *reinterpret_cast<const float4*>(sp + a_pix[i] * sld);
                if (p.square) { v.x *= v.x; v.y *= v.y; v.z *= v.z; v.w *= v.w; }
                ra[i] = v;
            }
            const float* wrow = p.w + ((int64_t)p.wtap[phase][it_t] * p.Cin + it_cg) * p.Cout + n0 + b_n;
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                const int k = bk0 + i * BROWS_PER_PASS;
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < BK && it_c + k < nch && b_nok) {
                    const float* wp = wrow + (int64_t)k * p.Cout;
                    if (b_full) w = *reinterpret_cast<const float4*>(wp);
                    else { w.x = wp[0]; if (n0 + b_n + 1 < p.Cout) w.y = wp[1]; if (n0 + b_n + 2 < p.Cout) w.z = wp[2]; }
                }
                rb[i] = w;
            }
            it_c += BK; it_cg += BK;
            if (it_c >= nch) {
                it_sbase += nch; it_c = 0; it_cg = it_sbase; ++it_s;
                if (it_s >= p.nseg) { it_s = 0; it_sbase = 0; it_cg = 0; ++it_t; }
            }
        };
        auto store_chunk = [&](int buf, const float4 (&ra)[AI], const float4 (&rb)[BI]) {
            float* a = As + buf * BK * LDA;
            float* b = Bs + buf * BK * LDB;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const int row = arow0 + i * ROWS_PER_PASS;
                if (row < BM) {
                    a[(kq * 4 + 0) * LDA + row] = ra[i].x;
                    a[(kq * 4 + 1) * LDA + row] = ra[i].y;
                    a[(kq * 4 + 2) * LDA + row] = ra[i].z;
                    a[(kq * 4 + 3) * LDA + row] = ra[i].w;
                }
            }
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                const int k = bk0 + i * BROWS_PER_PASS;
                if (k < BK) *reinterpret_cast<float4*>(b + k * LDB + b_n) = rb[i];
            }
        };
        float4 ra0[AI], rb0[BI], ra1[AI], rb1[BI];
        if (p.dbg & 2) {                                   // ablation: barriers only
            __syncthreads();
            for (int c = 0; c < nchunks; ++c) __syncthreads();
            return;
        }
        load_chunk(ra0, rb0);
        if (nchunks > 1) load_chunk(ra1, rb1);
        store_chunk(0, ra0, rb0);
        __syncthreads();                                   // stage 0 ready
        for (int c = 0;;) {
            if (c + 2 < nchunks) load_chunk(ra0, rb0);
            if (c + 1 < nchunks) store_chunk(1, ra1, rb1);
            __syncthreads();
            if (++c >= nchunks) break;
            if (c + 2 < nchunks) load_chunk(ra1, rb1);
            if (c + 1 < nchunks) store_chunk(0, ra0, rb0);
            __syncthreads();
            if (++c >= nchunks) break;
        }
        return;
    }

    // ---- MFMA waves
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int half = lane >> 5, l31 = lane & 31;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    auto compute = [&](int buf) {
        // operands of G k-steps are read into registers one group ahead of the MFMAs that consume them, so the
        // LDS latency of group g+1 hides behind the 64-cycle MFMAs of group g
        constexpr int G = 8, NG = BK / (2 * G);
        const float* a = As + buf * BK * LDA + wm * (TM * 32) + l31 + half * LDA;
        const float* b = Bs + buf * BK * LDB + wn * (TN * 32) + l31 + half * LDB;
        float av[2][G][TM], bv[2][G][TN];
        auto rd = [&](int g, int slot) {
#pragma unroll
            for (int s = 0; s < G; ++s) {
                const int kk = 2 * (g * G + s);
#pragma unroll
                for (int i = 0; i < TM; ++i) av[slot][s][i] = a[kk * LDA + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[slot][s][j] = b[kk * LDB + j * 32];
            }
        };
        rd(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) rd(g + 1, (g + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);     // keep the next group's LDS reads ahead of this group's MFMAs
#pragma unroll
            for (int s = 0; s < G; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][s][i], bv[g & 1][s][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    __syncthreads();                                       // stage 0 ready
    if (p.dbg & 1) {                                       // ablation: barriers only
        for (int c = 0; c < nchunks; ++c) __syncthreads();
    } else
    for (int c = 0;;) {
        compute(0);
        __syncthreads();
        if (++c >= nchunks) break;
        compute(1);
        __syncthreads();
        if (++c >= nchunks) break;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
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

template <int TM, int TN, int BK>
hipError_t launch_ws(const pc_conv_params& p, hipStream_t stream)
{
    constexpr int BM = 64 * TM, BN = 64 * TN;
    dim3 grid((p.M + BM - 1) / BM, (p.Cout + BN - 1) / BN, p.nphase);
    hipLaunchKernelGGL((conv_igemm_ws_kernel<TM, TN, BK>), grid, dim3(512), 0, stream, p);
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

// Host-side validation + tile selection.  Returns a pc status code.
int pc_conv_launch(const pc_conv_params& p, hipStream_t stream)
{
    if (p.nphase < 1 || p.nphase > 4 || p.M <= 0 || p.Cout <= 0 || p.Cin <= 0) return PC_ERR_ARG;
    static const int dbg_env = [] { const char* v = std::getenv("PC_CONV_DBG"); return v ? std::atoi(v) : 0; }();
    if (dbg_env) const_cast<pc_conv_params&>(p).dbg = dbg_env;
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

    // Tile / K-loop selection (measured on MI355X, tools/conv_tune.py, profiles/r01_*):
    //  * 64x64 block tiles win everywhere except the 3-channel output layer;
    //  * with >= 4 blocks per CU the low-register BK=16 loop is fastest (occupancy hides the per-chunk
    //    barrier); with fewer blocks (the M = 8192 slice-chain GEMMs) the BK=32 / prefetch-distance-2 loop is.
    int cfg = p.tile_cfg;
    static const int impl_env = [] { const char* v = std::getenv("PC_CONV_IMPL"); return v ? std::atoi(v) : 0; }();
    if (cfg == PC_TILE_AUTO) cfg = (p.Cout <= 4) ? PC_TILE_128x32 : PC_TILE_64x64;
    const int bm = cfg == PC_TILE_64x64 ? 64 : 128, bn = cfg == PC_TILE_128x128 ? 128 : (cfg == PC_TILE_64x64 ? 64 : 32);
    const long blocks = (long)((p.M + bm - 1) / bm) * ((p.Cout + bn - 1) / bn) * p.nphase;
    long ktot = 0;
    for (int ph = 0; ph < p.nphase; ++ph) ktot = std::max<long>(ktot, (long)p.ntap[ph] * p.Cin);
    //  * wave-specialised 64x64 kernel (4 MFMA + 4 loader waves) for every layer with a real K loop; K-chunk 64 when
    //    the grid is small (the loaders then need the longer MFMA phase to cover a global-load round trip);
    //  * the plain BK=16 kernel for K <= 256 (1x1 convs / GDN: epilogue-dominated) and for the 3-channel output layer.
    int impl = impl_env ? impl_env : ((ktot <= 256 || cfg != PC_TILE_64x64) ? 1 : 4);
    hipError_t e;
    if (!p.smallc && impl == 4) {
        static const int bk_env = [] { const char* v = std::getenv("PC_CONV_BK"); return v ? std::atoi(v) : 0; }();
        const int bk = bk_env ? bk_env : ((blocks < 1024 && ktot >= 1024) ? 64 : 32);
        switch (cfg) {
        case PC_TILE_128x128: e = launch_ws<2, 2, 32>(p, stream); break;
        case PC_TILE_64x64: e = bk == 64 ? launch_ws<1, 1, 64>(p, stream) : launch_ws<1, 1, 32>(p, stream); break;
        default: return PC_ERR_ARG;
        }
    } else if (p.smallc || impl == 1) {
        switch (cfg) {
        case PC_TILE_128x128: e = p.smallc ? launch_cfg<128, 128, 2, 2, true>(p, stream) : launch_cfg<128, 128, 2, 2, false>(p, stream); break;
        case PC_TILE_64x64: e = p.smallc ? launch_cfg<64, 64, 2, 2, true>(p, stream) : launch_cfg<64, 64, 2, 2, false>(p, stream); break;
        case PC_TILE_128x32: e = p.smallc ? launch_cfg<128, 32, 4, 1, true>(p, stream) : launch_cfg<128, 32, 4, 1, false>(p, stream); break;
        default: return PC_ERR_ARG;
        }
    } else {
        switch (cfg) {
        case PC_TILE_128x128: e = launch_cfg2<128, 128, 2, 2, 32>(p, stream); break;
        case PC_TILE_64x64: e = launch_cfg2<64, 64, 2, 2, 32>(p, stream); break;
        case PC_TILE_128x32: e = launch_cfg2<128, 32, 4, 1, 32>(p, stream); break;
        default: return PC_ERR_ARG;
        }
    }
    return e == hipSuccess ? PC_OK : PC_ERR_HIP;
}
