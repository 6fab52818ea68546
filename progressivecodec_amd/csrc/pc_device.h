// pc_device.h -- internal declarations shared by the HIP translation units of libpcodec.
#ifndef PC_DEVICE_H
#define PC_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pcodec.h"

// Tuning scaffolding.  The A/B builds of the profiling rounds (`make tuning` -> libpcodec_tuning.so, -DPC_TUNING; tools/env_matrix.sh,
// tools/gpu_ab.sh select it with PC_LIB) read their switches from the environment; the PRODUCT library reads none of them -- pc_tune()
// is the compiled-in default there.  What the product does read: PC_HOST_THREADS / PC_HOST_NO_PIN (host entropy-coding pool),
// LOCAL_RANK / LOCAL_WORLD_SIZE (the launcher's), PC_TIMING, PC_PROFILE_CSV (diagnostics); the schedule is set per object with
// pc_codec_set_option (include/pcodec.h).
#ifdef PC_TUNING
#include <cstdlib>
static inline long pc_tune(const char* name, long dflt) { const char* v = std::getenv(name); return v ? std::atol(v) : dflt; }
#else
static inline constexpr long pc_tune(const char*, long dflt) { return dflt; }
#endif

#define PC_MAX_SEG 8
#define PC_MAX_TAP 25

enum {
    PC_EPI_NONE = 0,
    PC_EPI_GELU = 1,      // nn.GELU (erf form)
    PC_EPI_RES_GELU = 2,  // ResidualUnit tail: gelu(conv + identity)            layers.py:51-57
    PC_EPI_RES = 3,       // window attention: shortcut + proj(...)               win_attention.py:205
    PC_EPI_GATE = 4,      // WAM: a * sigmoid(b) + identity                       layers.py:72-75
    PC_EPI_GDN = 5,       // x * rsqrt(beta + gamma . x^2)                        gdn.py:56-63
    PC_EPI_IGDN = 6,      // x * sqrt(...)
    PC_EPI_CLAMP01 = 7,   // x_hat.clamp_(0, 1)                                   CHProg_cnn.py:909,988
    PC_EPI_LRP = 8,       // y_hat + 0.5 * tanh(lrp)                              CHProg_cnn.py:759-762
    PC_EPI_LRP_ADD = 9,   // (y_hat + 0.5 * tanh(lrp)) + base  (merge "res")      CHProg_cnn.py:837-843
    PC_EPI_LEAKY = 10,    // nn.LeakyReLU() (slope 0.01)                          models/utils.py:65,79
    PC_EPI_LEAKY_RES = 11,// ResidualBlock tail: leaky(conv2) + identity          models/utils.py:80-86
};

enum { PC_TILE_AUTO = 0, PC_TILE_128x128 = 1, PC_TILE_64x64 = 2, PC_TILE_128x32 = 3 };

struct pc_seg {
    const float* ptr;  // first channel of this segment at pixel 0
    int ld;            // floats between consecutive pixels
    int nch;           // channels in this segment (multiple of 16)
};

struct pc_conv_params {
    // input: NHWC, channel axis = concatenation of segments (all segments share B, H, W)
    pc_seg seg[PC_MAX_SEG];
    int nseg, Cin;
    int B, H, W;
    int smallc;                              // element-wise gather path (Cin not a multiple of 16); uses in_s*
    int64_t in_sb, in_sy, in_sx, in_sc;      // generic input strides (smallc only)
    int square;                              // feed x*x (GDN)
    // taps per output phase
    int nphase;
    int ntap[4];
    int dy[4][PC_MAX_TAP], dx[4][PC_MAX_TAP];   // int32: wave-uniform reads become s_load_dword (int8 made hipcc emit vector byte loads)
    int wtap[4][PC_MAX_TAP];                 // weight tap index
    int stride;
    // weights: layout 0 = [ntaps][Cin][Cout]; layout 1 = [ntaps][Cout][Cin] (pc_conv.hip)
    const float* w;
    int wlayout;
    const float* bias;
    int Cout;
    // output grid per phase and mapping into the output tensor
    int Ho, Wo, M;                           // M = B*Ho*Wo
    int osy, osx, ooy[4], oox[4];
    int outH, outW;
    int64_t out_sb, out_sy, out_sx, out_sc;  // generic output strides (NHWC slice or NCHW)
    float* out;
    int pixel_shuffle;                       // PixelShuffle(2) folded into the store
    // epilogue
    int epi;
    const float* aux0; int ld0;
    const float* aux1; int ld1;
    int tile_cfg;
    // fused GDN behind the 3-channel input layer (conv_igemm_in_gdn_kernel): gamma [192][192] in layout 1 (re-parametrised), beta [192];
    // out = x * rsqrt(beta + gamma . x^2) with x = this layer's conv + bias.  Null: plain layer.
    const float* fg_gamma; const float* fg_beta;
    // grouped launch: a second GEMM of identical shape (cc_mean || cc_scale of one chain step) rides on blockIdx.z == 1;
    // it differs only in its first input segment, weights, bias and output
    int ngroup;
    const float* g1_seg0; const float* g1_w; const float* g1_bias; float* g1_out;
    const int* rowtab;                       // set by pc_conv_launch (layers with more than one tap): cached per layer geometry,
                                             // [M] input-pixel index of each GEMM row, then [nphase][M] tap-validity masks
    void* rowtab_cache;                      // pc_rowtab_cache* of the calling codec (null: the process-wide cache of the stand-alone entry points)
    int ident_rows;                          // set by pc_conv_launch: 1x1 stride-1 layer, GEMM row m reads input pixel m (no row table)
    int dense_out;                           // set by pc_conv_launch: output pixel index == GEMM row (plain NHWC-strided store)
    int rowperm;                             // set by pc_conv_launch: the row table is a PERMUTATION of the pixels (rows grouped by tap-validity
                                             // pattern, so that a tile's padding taps can be skipped as whole runs); output pixel of row m = rowtab[m]
    int group_xcd;                           // set by pc_conv_launch (grouped launches): XCDs 0-3 run group 0, XCDs 4-7 group 1 -- each L2 streams one group's weight slabs
    int dbg;                                 // tuning only (PC_CONV_DBG bits): 1 skip MFMAs, 2 skip DMA issue, 4 DMAs read the zero page, 64 stamps, 256 print occupancy
};

int pc_conv_launch(const pc_conv_params& p, hipStream_t stream);
// row-table cache (pc_conv.hip): owned by a codec, freed with it; cap_bytes bounds the HBM it may hold (LRU eviction)
struct pc_rowtab_cache;
pc_rowtab_cache* pc_rowtab_cache_create(size_t cap_bytes);
void pc_rowtab_cache_destroy(pc_rowtab_cache* c);
size_t pc_rowtab_cache_bytes(pc_rowtab_cache* c);
// weight layout the launcher expects for a layer (kind 0 conv / 1 transposed conv k5 s2)
int pc_conv_weight_layout(int kind, int Cin, int Cout, int k);

// window attention core
// bias: dense relative-position bias, [heads][T][T] as bias[h][i][j] (bias_ji = 0: the module's order, win_attention.py:97-100) or
// transposed bias[h][j][i] (bias_ji = 1: what the codec stores -- coalesced across the query lanes)
int pc_win_attention_launch(const float* qkv, const float* bias, int B, int H, int W, int C, int heads, int ws,
                            int shift, float scale, float* out, hipStream_t stream, int bias_ji = 0);

// entropy-parameter stages
struct pc_prep_params {
    int B, HW, C;             // one slice: B images x HW pixels x C (=32) channels, NHWC inputs
    const float* scale; int ld_scale;
    const float* mu; int ld_mu;
    const float* y; int ld_y;          // encoder: latent slice
    const float* ybase; int ld_ybase;  // encoder, delta_encode: base slice to subtract (or null)
    const float* thr;                  // per-image mask threshold (null: no mask; enhancement only)
    const float* mask_src; int64_t mask_sb;   // optional: threshold this map instead of `scale` (cust_map, masking.py:171-194):
                                       // element (b, c, p) at mask_src[b * mask_sb + c * HW + p]
    int mask_mode;                     // 0 none, 1 threshold compare, 2 all ones, 3 all zeros
    const float* table; int ntable; float bound;
    int32_t* sym;                      // [B][C][HW]  (C,H,W raster order = rANS order)
    int32_t* idx;                      // [B][C][HW]
    uint8_t* idx8;                     // optional (decoder): the same indexes as bytes, [B][C][HW] -- what the host coder reads back
    float* mask;                       // [B][C][HW] float 0/1 or null
    float* yhat; int ld_yhat;          // NHWC: float(sym) + mu
    float* lik; int64_t lik_sb;        // optional (encoder): Gaussian likelihood of the coded symbol, element (b, c, p) at
                                       // lik[b * lik_sb + c * HW + p]   (entropy_models.py:626-659)
};
int pc_prep_enc_launch(const pc_prep_params& p, hipStream_t stream);
int pc_prep_dec_index_launch(const pc_prep_params& p, hipStream_t stream);   // scale(+mask) -> idx (+mask)
int pc_prep_dec_dequant_launch(const pc_prep_params& p, hipStream_t stream); // sym + mu -> yhat

int pc_quantile_thr_launch(const float* scale, int ld, int B, int HW, int C, float q, float* thr, uint32_t* work,
                           hipStream_t stream, int64_t batch_stride = 0);   // batch_stride 0: HW * ld
size_t pc_quantile_work_bytes(int B);
#define PC_QUANTILE_SMALL_N 32768   // up to here one workgroup per image holds the keys in registers and `work` is not used

// REM (models/CHProgREM.py:84-86,395-401): scale <- ret * round(star - bar) + scale, the two masks thresholding the UNREFINED scale
// (mode 1: value >= thr[b]; 2: ones; 3: zeros).  ret / scale NHWC [B][HW][32] with pixel strides.
// mu != null: the mu_std form -- ret has 2N channels per pixel, mu <- ret[:N] * att + mu as well (:397-398,414-416)
int pc_rem_combine_launch(const float* ret, int ld_ret, float* scale, int ld_scale, int B, int HW, const float* thr_star, int mode_star,
                          const float* thr_bar, int mode_bar, hipStream_t stream, float* mu = nullptr, int ld_mu = 0);
// one C-channel slice of an NCHW tensor (element (b, c, p) at src[b * batch_stride + c * HW + p]) -> NHWC [B][HW][C]
int pc_nchw_slice_to_nhwc_launch(const float* src, int64_t batch_stride, int B, int HW, int C, float* dst, hipStream_t stream);

int pc_eb_quant_launch(const float* z, int B, int HW, int C, const float* medians, int32_t* sym, float* zhat,
                       hipStream_t stream);
int pc_eb_dequant_launch(const int32_t* sym, int B, int HW, int C, const float* medians, float* zhat,
                         hipStream_t stream);
// EntropyBottleneck._likelihood of the dequantised hyper-latent (entropy_models.py:400-433): sym [B][C][HW] -> lik [B][C][HW];
// net = PC_EB_NET_FLOATS floats per channel (softplus(matrix), bias, tanh(factor) of the 1-3-3-3-3-1 density network)
#define PC_EB_NET_FLOATS 58
int pc_eb_likelihood_launch(const int32_t* sym, int B, int HW, int C, const float* medians, const float* net, float* lik,
                            hipStream_t stream);

#endif
