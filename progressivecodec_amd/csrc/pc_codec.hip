// pc_codec.hip -- native runtime of ChannelProgresssiveWACNN.compress()/decompress()
// (reference: /root/reference/src/compress/models/CHProg_cnn.py:686-999) for the canonical
// configuration, as HIP launch sequences over HBM-resident weights.
//
// Data layout in HBM
//   * activations NHWC float32; the 640-channel latent y, the hyper-synthesis outputs
//     latent_means / latent_scales ([M][640]) and the decoded slices (y_hat base / enhancement,
//     [M][320] each) are single buffers; the reference's torch.cat() supports are expressed as
//     channel *segments* of these buffers handed to the conv kernel (no copies).
//   * conv weights tap-major [tap][Cin][Cout] (N contiguous = GEMM B operand rows).
//   * symbols / indexes [slice][B][32][h*w] int32 = the rANS coder's C,H,W raster order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <new>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/pcodec.h"
#include "pc_device.h"
#include "pc_host.h"

// last HIP error seen by ANY thread of the library (the decoder's chains run on their own host threads: a thread-local would hide
// their errors from the caller of pc_last_hip_error)
static std::atomic<int> g_last_hip{0};
struct pc_codec;
static thread_local pc_codec* g_prof = nullptr;   // codec whose conv launches are being timed (profile mode)
static thread_local pc_rowtab_cache* g_rowtabs = nullptr;   // row-table cache of the codec this thread is working for
#define HIPCHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { g_last_hip = (int)_e; return PC_ERR_HIP; } } while (0)
#define PCCHK(expr) do { int _r = (expr); if (_r != PC_OK) return _r; } while (0)

extern "C" int pc_last_hip_error(void) { return g_last_hip.load(); }

namespace {

constexpr int NS0 = 10, D0 = 320, MLAT = 640, NCH = 192, HEADS = 8, SLICE = 32;
const int CC_W[5] = {224, 176, 128, 64, 32};

struct HostTensor { std::vector<uint8_t> data; std::vector<int64_t> shape; int dtype; };

struct ConvW { float* w = nullptr; float* b = nullptr; int Cin = 0, Cout = 0, k = 0, kind = 0, layout = 0; };
struct GdnW { float* beta = nullptr; float* gamma = nullptr; int C = 0; };   // gamma [C_out][C_in], re-parametrised
struct RuW { ConvW c0, c2, c4; };
struct WamW { RuW a[3], b[3]; ConvW qkv, proj, out; float* bias = nullptr; int C = 0, ws = 0, shift = 0; };
struct Stack5W { ConvW c[5]; };
struct RbW { ConvW c1, c2, skip; bool has_skip = false; };                        // ResidualBlock (models/utils.py:59-87)
struct LrrW { RbW ent_base[3], ent_enh[3], base_rep[3], enc[4]; int n_sub = 0, n_enc = 0; };   // LatentRateReduction (CHProgREM.py:12-72)
struct HsW { ConvW c0, c2, c4, c6, c8; };
struct GsW { WamW w0, w5; ConvW d1, d3, d6, d8; GdnW g2, g4, g7; };   // d8: the 192 -> 3 deconv in sub-pixel form (load_deconv3_subpixel)

struct Tables {
    std::vector<int32_t> cdf, len, off;
    std::vector<uint64_t> lut;                   // decoder start table per row (pc_host.h: DecTables)
    int n = 0, stride = 0;
    bool ok() const { return n > 0; }
    pc::DecTables dec() const { return pc::DecTables{cdf.data(), n, stride, len.data(), off.data(), lut.data()}; }
};

struct DevBuf { void* p = nullptr; size_t bytes = 0; };

}  // namespace

struct pc_codec {
    int device = 0;
    bool finalized = false;
    std::atomic<bool> busy{false};               // a compress / decompress / forward call is inside: the object is not re-entrant
    hipEvent_t call_done = nullptr;              // recorded on the caller's stream when a call returns; the next call's stream waits for it
    hipEvent_t staging_done = nullptr;           // decoder: recorded behind the last chain of a decompress -- every H2D copy out of the pinned staging has run by then
    bool staging_pending = false;
    pc_rowtab_cache* rowtabs = nullptr;          // per-geometry row tables of the conv kernel: owned here, freed in pc_codec_destroy
    std::map<std::string, HostTensor> sd;
    std::vector<void*> weight_allocs;
    std::map<std::string, DevBuf> bufs;          // grow-only named workspace
    // network
    struct GaW { ConvW c0, c2, c5, c7; GdnW g1, g3, g6; WamW w4, w8; };
    GaW ga[2];                                   // ga[1] only with multiple_encoder (CHProg_cnn.py:131-144)
    // REM: PostRateProcessedNetwork.post_latent[level][slice] (CHProgREM.py:227-234), loaded when the state dict carries post_latent.* keys
    std::vector<LrrW> rem;                       // [level * NS0 + slice]
    int rem_levels_loaded = 0;                   // check levels with weights
    int rem_n = 0;                               // active check levels (0: REM off) -- pc_codec_set_rem
    double rem_check[3] = {0, 0, 0};
    bool multi_enc = false;                      // two 3 -> 320 encoders, y = cat(g_a[0](x), g_a[1](x))  (:691-697)
    GsW gs[2];
    ConvW ha[5];
    HsW hms[2], hss[2];
    Stack5W cc_mean[NS0], cc_scale[NS0], lrp[NS0], cc_mean_p[NS0], cc_scale_p[NS0], lrp_p[NS0];
    float* medians = nullptr;                    // [192] device
    const float* cust_map = nullptr;             // pc_codec_set_cust_map: consumed by the next compress / decompress call
    const float* rem_ckpt = nullptr;             // pc_codec_set_rem_checkpoint: NCHW [B][320][HW] representation for the REM nets' x_base input (next call)
    int opt_serial = 0, opt_lanes_enc = 0, opt_lanes_dec = 0;   // pc_codec_set_option
    int rem_mu_std = 0;                          // the loaded post_latent nets are the mu_std=True form (2N-channel enhancement branch and output)
    float* eb_net = nullptr;                     // [192][PC_EB_NET_FLOATS] device: density network of the EntropyBottleneck (forward path)
    float* scale_table = nullptr;                // [64] device
    int n_table = 0;
    float scale_bound = 0.11f;
    Tables gc, eb;
    // pinned host staging
    int32_t* h_sym = nullptr; int32_t* h_idx = nullptr; size_t h_cap = 0;
    // results of the last compress
    int res_slices = 0, res_B = 0;               // res_slices = string slots: 10 base + 10 per coded level
    std::vector<char> res_level_coded;            // per level of the last compress: enhancement strings present (quality > 0)
    std::vector<hipEvent_t> lvl_events;           // D2H completion of the base pass / of each level
    hipStream_t copy_stream = nullptr;            // per-slice D2H of the last pass (streamed entropy coding)
    hipStream_t pipe_stream = nullptr;            // enhancement chain when it is pipelined against the base chain
    hipStream_t hyper_streams[3] = {nullptr, nullptr, nullptr};   // the four hyper-synthesis nets run side by side
    hipEvent_t hyper_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> pipe_ev;              // [NS0] base slice i complete, [NS0] pipeline fork, [NS0+1] pipeline join
    std::vector<hipEvent_t> slice_ev;             // [2*NS0]: prep done / copied, per slice of that pass
    std::vector<std::vector<uint8_t>> y_strings;  // [slot*B + b]; slot = slice (base) or 10 + 10*level + (slice - 10)
    std::vector<std::vector<uint8_t>> z_strings;  // [b]
    int n_threads = 0;
    // optional per-launch profiling of the MFMA conv family (bench.py roofline leg)
    bool profile = false;
    bool profile_in_schedule = false;             // pc_codec_set_option("profile_in_schedule"): brackets only -- the schedule stays what it is
    bool serial_profile() const { return profile && !profile_in_schedule; }   // profiling that forces the chain onto the caller's stream
    std::mutex prof_mu;                           // in-schedule profiling records from every host thread of the object
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    double prof_flops = 0.0, prof_bytes = 0.0;
    struct ProfRec { int M, N, K, nphase, epi; double flops, bytes; double t0_ms, t1_ms; };   // t0 / t1: against the process epoch (pc_profile_set_epoch)
    std::vector<ProfRec> prof_rec;
    // slice-chain lanes: one pair of non-blocking streams per sub-batch
    struct Lane { hipStream_t sA = nullptr, sB = nullptr; hipEvent_t eA = nullptr, eB = nullptr, eDone = nullptr; };
    std::vector<Lane> lanes;
    hipEvent_t eFork = nullptr;
    std::mutex buf_mu;
    double t_host_decode_ms = 0.0;
    // host entropy-coding figures of the last compress / decompress call (pc_codec_host_stats)
    double t_compress_ms = 0.0, t_host_encode_ms = 0.0, t_host_encode_exposed_ms = 0.0, t_decompress_ms = 0.0;
    double n_sym_encoded = 0.0, n_sym_decoded = 0.0;
    // last-call geometry for taps
    int last_B = 0, last_h16 = 0, last_w16 = 0;

    template <typename T> int buf(const std::string& name, size_t count, T** out, bool zero_when_allocated = false)
    {
        std::lock_guard<std::mutex> lk(buf_mu);
        DevBuf& d = bufs[name];
        const size_t need = count * sizeof(T);
        if (d.bytes < need) {
            if (d.p) (void)hipFree(d.p);
            d.p = nullptr; d.bytes = 0;
            HIPCHK(hipMalloc(&d.p, need));
            d.bytes = need;
            if (zero_when_allocated) { HIPCHK(hipMemset(d.p, 0, need)); HIPCHK(hipStreamSynchronize(nullptr)); }   // once per size: scratch whose kernels keep it zero between launches (the lanes' streams are non-blocking: the fill must have landed before one of them uses the buffer)
        }
        *out = reinterpret_cast<T*>(d.p);
        return PC_OK;
    }
};

namespace {

// conv launch, optionally bracketed by HIP events on the launch stream (profile mode)
int launch_conv(const pc_conv_params& q_in, hipStream_t st)
{
    pc_conv_params q = q_in;
    q.rowtab_cache = g_rowtabs;
    pc_codec* c = g_prof;
    if (!c) return pc_conv_launch(q, st);
    long taps = 0;
    for (int ph = 0; ph < q.nphase; ++ph) taps += q.ntap[ph];
    const double ng = q.ngroup == 2 ? 2.0 : 1.0;
    double fl = 2.0 * (double)q.M * (double)q.Cout * (double)taps * (double)q.Cin * ng;
    if (q.fg_gamma) fl += 2.0 * (double)q.M * (double)q.Cout * (double)q.Cout;          // the fused GDN's 1x1 contraction
    // algorithmic HBM bytes of the launch (SURVEY.md section 8d style: every operand once): the input tensor once (all segments; a
    // grouped launch reads a second first segment), the weights and bias of every phase / group, the output once, each aux tensor once
    double by = 0.0;
    for (int sg = 0; sg < q.nseg; ++sg) by += 4.0 * (double)q.B * q.H * q.W * q.seg[sg].nch * (sg == 0 ? ng : 1.0);
    by += ng * 4.0 * ((double)taps * q.Cin * q.Cout + q.Cout);
    if (q.fg_gamma) by += 4.0 * ((double)q.Cout * q.Cout + q.Cout);
    by += ng * 4.0 * (double)q.nphase * q.M * q.Cout;
    if (q.aux0 && q.epi != PC_EPI_NONE && q.epi != PC_EPI_GELU && q.epi != PC_EPI_CLAMP01) by += 4.0 * (double)q.nphase * q.M * q.Cout;
    if (q.aux1 && (q.epi == PC_EPI_GATE || q.epi == PC_EPI_LRP_ADD)) by += 4.0 * (double)q.nphase * q.M * q.Cout;
    hipEvent_t e0, e1;
    {
        std::lock_guard<std::mutex> lk(c->prof_mu);                      // (the object's chains may launch from two host threads)
        if (c->ev_used + 2 > c->ev.size()) {
            const size_t old = c->ev.size();
            c->ev.resize(old + 512);
            for (size_t i = old; i < c->ev.size(); ++i) HIPCHK(hipEventCreate(&c->ev[i]));
        }
        c->prof_flops += fl;
        c->prof_bytes += by;
        c->prof_rec.push_back({q.M, q.Cout, (int)(taps * q.Cin / q.nphase), q.nphase, q.epi, fl, by, 0.0, 0.0});
        e0 = c->ev[c->ev_used++]; e1 = c->ev[c->ev_used++];
    }
    HIPCHK(hipEventRecord(e0, st));
    const int r = pc_conv_launch(q, st);
    HIPCHK(hipEventRecord(e1, st));
    return r;
}

// ------------------------------------------------------------------------------------------ weights
const HostTensor* find(const pc_codec* c, const std::string& k, int dtype, std::initializer_list<int64_t> shape)
{
    auto it = c->sd.find(k);
    if (it == c->sd.end() || it->second.dtype != dtype) return nullptr;
    if (it->second.shape.size() != shape.size()) return nullptr;
    size_t i = 0;
    for (int64_t s : shape) if (it->second.shape[i++] != s) return nullptr;
    return &it->second;
}

int upload(pc_codec* c, const std::vector<float>& h, float** dev)
{
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, std::max<size_t>(h.size(), 4) * sizeof(float)));
    HIPCHK(hipMemcpy(p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    c->weight_allocs.push_back(p);
    *dev = reinterpret_cast<float*>(p);
    return PC_OK;
}

int load_conv(pc_codec* c, const std::string& p, int Cin, int Cout, int k, int kind, ConvW* out)
{
    const HostTensor* w = kind == 0 ? find(c, p + ".weight", PC_F32, {Cout, Cin, k, k}) : find(c, p + ".weight", PC_F32, {Cin, Cout, k, k});
    const HostTensor* b = find(c, p + ".bias", PC_F32, {Cout});
    if (!w || !b) { std::fprintf(stderr, "[pcodec] missing/mis-shaped tensor %s\n", p.c_str()); return PC_ERR_MISSING; }
    std::vector<float> packed((size_t)k * k * Cin * Cout);
    PCCHK(pc_pack_conv_weight(reinterpret_cast<const float*>(w->data.data()), kind, Cout, Cin, k, packed.data()));
    PCCHK(upload(c, packed, &out->w));
    std::vector<float> bias(reinterpret_cast<const float*>(b->data.data()), reinterpret_cast<const float*>(b->data.data()) + Cout);
    PCCHK(upload(c, bias, &out->b));
    out->Cin = Cin; out->Cout = Cout; out->k = k; out->kind = kind; out->layout = pc_conv_weight_layout(kind, Cin, Cout, k);
    return PC_OK;
}

// The last layer of g_s, ConvTranspose2d(192, 3, 5, s2, p2, op1) (models/utils.py:196, CHProg_cnn.py:160): with three output
// channels a GEMM over output phases wastes 29 of every 32 MFMA columns (2.0 ms per Config-2 batch).  Sub-pixel form instead: ONE
// stride-1 conv over the 3x3 input neighbourhood with 12 virtual channels n = c*4 + py*2 + px, stored through the PixelShuffle
// epilogue; weight (tap (dy,dx), n) = W[ci][c][py+2-2dy][px+2-2dx] where that kernel index exists, else 0.  Taps are visited
// dy = +1,0,-1 (outer), dx = +1,0,-1 (inner), so for every phase the taps it really has appear in ascending (ky,kx) -- the
// contract's order for a transposed-conv phase -- and a zero-weight tap leaves the fmaf chain's value unchanged: bit-identical
// to the four-phase form (tests/test_gpu_codec.py vs the oracle's four-phase restatement).
int load_deconv3_subpixel(pc_codec* c, const std::string& p, int Cin, ConvW* out)
{
    const HostTensor* w = find(c, p + ".weight", PC_F32, {Cin, 3, 5, 5});
    const HostTensor* b = find(c, p + ".bias", PC_F32, {3});
    if (!w || !b || Cin % 16) { std::fprintf(stderr, "[pcodec] missing/mis-shaped tensor %s\n", p.c_str()); return PC_ERR_MISSING; }
    const float* W = reinterpret_cast<const float*>(w->data.data());
    std::vector<float> packed((size_t)9 * 12 * Cin, 0.0f);                    // layout 1: [tap][n][ci]
    for (int t = 0; t < 9; ++t) {
        const int dy = 1 - t / 3, dx = 1 - t % 3;
        for (int n = 0; n < 12; ++n) {
            const int cch = n >> 2, py = (n >> 1) & 1, px = n & 1;
            const int ky = py + 2 - 2 * dy, kx = px + 2 - 2 * dx;
            if (ky < 0 || ky >= 5 || kx < 0 || kx >= 5) continue;
            for (int ci = 0; ci < Cin; ++ci) packed[((size_t)t * 12 + n) * Cin + ci] = W[(((size_t)ci * 3 + cch) * 5 + ky) * 5 + kx];
        }
    }
    PCCHK(upload(c, packed, &out->w));
    std::vector<float> bias(12);
    for (int n = 0; n < 12; ++n) bias[n] = reinterpret_cast<const float*>(b->data.data())[n >> 2];
    PCCHK(upload(c, bias, &out->b));
    out->Cin = Cin; out->Cout = 12; out->k = 3; out->kind = 0; out->layout = 1;
    return PC_OK;
}

int load_linear(pc_codec* c, const std::string& p, int Cin, int Cout, ConvW* out)     // nn.Linear weight [out][in]
{
    const HostTensor* w = find(c, p + ".weight", PC_F32, {Cout, Cin});
    const HostTensor* b = find(c, p + ".bias", PC_F32, {Cout});
    if (!w || !b) { std::fprintf(stderr, "[pcodec] missing/mis-shaped tensor %s\n", p.c_str()); return PC_ERR_MISSING; }
    std::vector<float> packed((size_t)Cin * Cout);
    PCCHK(pc_pack_conv_weight(reinterpret_cast<const float*>(w->data.data()), 0, Cout, Cin, 1, packed.data()));
    PCCHK(upload(c, packed, &out->w));
    std::vector<float> bias(reinterpret_cast<const float*>(b->data.data()), reinterpret_cast<const float*>(b->data.data()) + Cout);
    PCCHK(upload(c, bias, &out->b));
    out->Cin = Cin; out->Cout = Cout; out->k = 1; out->kind = 0; out->layout = pc_conv_weight_layout(0, Cin, Cout, 1);
    return PC_OK;
}

// NonNegativeParametrizer.forward (ops/parametrizers.py:46-49): max(x, bound)^2 - pedestal, in float32
int load_gdn(pc_codec* c, const std::string& p, int C, GdnW* out)
{
    const HostTensor* beta = find(c, p + ".beta", PC_F32, {C});
    const HostTensor* gamma = find(c, p + ".gamma", PC_F32, {C, C});
    const HostTensor* bb = find(c, p + ".beta_reparam.lower_bound.bound", PC_F32, {1});
    const HostTensor* bp = find(c, p + ".beta_reparam.pedestal", PC_F32, {1});
    const HostTensor* gb = find(c, p + ".gamma_reparam.lower_bound.bound", PC_F32, {1});
    const HostTensor* gp = find(c, p + ".gamma_reparam.pedestal", PC_F32, {1});
    if (!beta || !gamma || !bb || !bp || !gb || !gp) { std::fprintf(stderr, "[pcodec] missing GDN tensors %s\n", p.c_str()); return PC_ERR_MISSING; }
    auto f = [](const HostTensor* t) { return reinterpret_cast<const float*>(t->data.data()); };
    std::vector<float> hb(C), hg((size_t)C * C);
    for (int i = 0; i < C; ++i) { const float v = std::max(f(beta)[i], f(bb)[0]); hb[i] = v * v - f(bp)[0]; }
    for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) { const float v = std::max(f(gamma)[(size_t)i * C + j], f(gb)[0]); hg[(size_t)i * C + j] = v * v - f(gp)[0]; }
    PCCHK(upload(c, hb, &out->beta));
    PCCHK(upload(c, hg, &out->gamma));
    out->C = C;
    return PC_OK;
}

int load_ru(pc_codec* c, const std::string& p, int C, RuW* r)
{
    PCCHK(load_conv(c, p + ".conv.0", C, C / 2, 1, 0, &r->c0));
    PCCHK(load_conv(c, p + ".conv.2", C / 2, C / 2, 3, 0, &r->c2));
    PCCHK(load_conv(c, p + ".conv.4", C / 2, C, 1, 0, &r->c4));
    return PC_OK;
}

int load_wam(pc_codec* c, const std::string& p, int C, int ws, int shift, WamW* w)
{
    w->C = C; w->ws = ws; w->shift = shift;
    for (int i = 0; i < 3; ++i) PCCHK(load_ru(c, p + ".conv_a." + std::to_string(i), C, &w->a[i]));
    for (int i = 0; i < 3; ++i) PCCHK(load_ru(c, p + ".conv_b." + std::to_string(i + 1), C, &w->b[i]));
    PCCHK(load_linear(c, p + ".conv_b.0.attn.qkv", C, 3 * C, &w->qkv));
    PCCHK(load_linear(c, p + ".conv_b.0.attn.proj", C, C, &w->proj));
    PCCHK(load_conv(c, p + ".conv_b.4", C, C, 1, 0, &w->out));
    const int T = ws * ws, R = (2 * ws - 1) * (2 * ws - 1);
    const HostTensor* tab = find(c, p + ".conv_b.0.attn.relative_position_bias_table", PC_F32, {R, HEADS});
    const HostTensor* idx = find(c, p + ".conv_b.0.attn.relative_position_index", PC_I64, {T, T});
    if (!tab || !idx) { std::fprintf(stderr, "[pcodec] missing attention tables %s\n", p.c_str()); return PC_ERR_MISSING; }
    // dense bias, stored TRANSPOSED: bias[h][j][i] = table[index[i][j]][h]   (win_attention.py:97-100; [j][i]: coalesced across the query lanes)
    std::vector<float> dense((size_t)HEADS * T * T);
    const float* t = reinterpret_cast<const float*>(tab->data.data());
    const int64_t* ix = reinterpret_cast<const int64_t*>(idx->data.data());
    for (int i = 0; i < T; ++i)
        for (int j = 0; j < T; ++j) {
            const int64_t r = ix[(size_t)i * T + j];
            if (r < 0 || r >= R) return PC_ERR_MISSING;
            for (int h = 0; h < HEADS; ++h) dense[((size_t)h * T + j) * T + i] = t[r * HEADS + h];
        }
    PCCHK(upload(c, dense, &w->bias));
    return PC_OK;
}

int load_rb(pc_codec* c, const std::string& p, int ci, int co, RbW* r)
{
    PCCHK(load_conv(c, p + ".conv1", ci, co, 3, 0, &r->c1));
    PCCHK(load_conv(c, p + ".conv2", co, co, 3, 0, &r->c2));
    r->has_skip = ci != co;
    if (r->has_skip) PCCHK(load_conv(c, p + ".skip", ci, co, 1, 0, &r->skip));
    return PC_OK;
}

int load_stack5(pc_codec* c, const std::string& p, int Cin, Stack5W* s)
{
    int ci = Cin;
    for (int j = 0; j < 5; ++j) { PCCHK(load_conv(c, p + "." + std::to_string(2 * j), ci, CC_W[j], 3, 0, &s->c[j])); ci = CC_W[j]; }
    return PC_OK;
}

int load_hs(pc_codec* c, const std::string& p, HsW* h)
{
    PCCHK(load_conv(c, p + ".0", NCH, 192, 3, 0, &h->c0));
    PCCHK(load_conv(c, p + ".2.0", 192, 224 * 4, 3, 0, &h->c2));
    PCCHK(load_conv(c, p + ".4", 224, 256, 3, 0, &h->c4));
    PCCHK(load_conv(c, p + ".6.0", 256, 288 * 4, 3, 0, &h->c6));
    PCCHK(load_conv(c, p + ".8", 288, D0, 3, 0, &h->c8));
    return PC_OK;
}

// ------------------------------------------------------------------------------------------ launch helpers
struct Seg { const float* p; int ld; int nch; };

void fill_conv_taps(pc_conv_params& q, int k, int stride)
{
    const int pad = k / 2;
    q.nphase = 1; q.ntap[0] = k * k; q.stride = stride;
    for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx) {
            const int t = ky * k + kx;
            q.dy[0][t] = (ky - pad); q.dx[0][t] = (kx - pad); q.wtap[0][t] = t;
        }
    q.osy = q.osx = 1; q.ooy[0] = q.oox[0] = 0;
}

void fill_deconv_taps(pc_conv_params& q)   // ConvTranspose2d(5, s2, p2, op1) as 4 output phases
{
    q.nphase = 4; q.stride = 1; q.osy = q.osx = 2;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            const int ph = py * 2 + px;
            int t = 0;
            for (int ky = py; ky < 5; ky += 2)
                for (int kx = px; kx < 5; kx += 2) {
                    q.dy[ph][t] = ((py + 2 - ky) / 2); q.dx[ph][t] = ((px + 2 - kx) / 2);
                    q.wtap[ph][t] = ky * 5 + kx;
                    ++t;
                }
            q.ntap[ph] = t; q.ooy[ph] = py; q.oox[ph] = px;
        }
}

// generic conv over NHWC segments -> NHWC output slice (channel offset folded into `out`, pixel stride ldo)
struct Group1 { const ConvW* w; const float* seg0; float* out; };   // second GEMM of a grouped launch

int conv(hipStream_t st, const ConvW& w, std::initializer_list<Seg> segs, int B, int H, int W, int stride,
         float* out, int ldo, int epi, const float* aux0 = nullptr, int ld0 = 0, const float* aux1 = nullptr, int ld1 = 0,
         bool pixel_shuffle = false, const Group1* g1 = nullptr)
{
    pc_conv_params q;
    std::memset(&q, 0, sizeof(q));
    int cin = 0;
    for (const Seg& s : segs) {
        if (s.nch == 0) continue;
        if (q.nseg >= PC_MAX_SEG) return PC_ERR_ARG;
        q.seg[q.nseg].ptr = s.p; q.seg[q.nseg].ld = s.ld; q.seg[q.nseg].nch = s.nch; ++q.nseg; cin += s.nch;
    }
    if (cin != w.Cin) return PC_ERR_ARG;
    q.Cin = cin; q.B = B; q.H = H; q.W = W;
    q.w = w.w; q.wlayout = w.layout; q.bias = w.b; q.Cout = w.Cout;
    q.epi = epi; q.aux0 = aux0; q.ld0 = ld0; q.aux1 = aux1; q.ld1 = ld1;
    q.out = out;
    if (w.kind == 0) {
        fill_conv_taps(q, w.k, stride);
        q.Ho = (H + 2 * (w.k / 2) - w.k) / stride + 1;
        q.Wo = (W + 2 * (w.k / 2) - w.k) / stride + 1;
        q.outH = q.Ho; q.outW = q.Wo;
    } else {
        fill_deconv_taps(q);
        q.Ho = H; q.Wo = W; q.outH = 2 * H; q.outW = 2 * W;
    }
    q.M = B * q.Ho * q.Wo;
    q.pixel_shuffle = pixel_shuffle ? 1 : 0;
    if (pixel_shuffle) { q.outH *= 2; q.outW *= 2; }
    q.out_sc = 1; q.out_sx = ldo; q.out_sy = (int64_t)q.outW * ldo; q.out_sb = (int64_t)q.outH * q.outW * ldo;
    if (g1) {
        if (g1->w->Cin != w.Cin || g1->w->Cout != w.Cout || g1->w->k != w.k || g1->w->layout != 1 || w.layout != 1) return PC_ERR_ARG;
        q.ngroup = 2; q.g1_seg0 = g1->seg0; q.g1_w = g1->w->w; q.g1_bias = g1->w->b; q.g1_out = g1->out;
    }
    return launch_conv(q, st);
}

int gdn(hipStream_t st, const GdnW& g, const float* x, int B, int H, int W, bool inverse, float* out)
{
    pc_conv_params q;
    std::memset(&q, 0, sizeof(q));
    q.nseg = 1; q.seg[0].ptr = x; q.seg[0].ld = g.C; q.seg[0].nch = g.C; q.Cin = g.C;
    q.B = B; q.H = H; q.W = W; q.square = 1;
    fill_conv_taps(q, 1, 1);
    q.w = g.gamma; q.wlayout = 1; q.bias = g.beta; q.Cout = g.C;
    q.Ho = H; q.Wo = W; q.outH = H; q.outW = W; q.M = B * H * W;
    q.out = out; q.out_sc = 1; q.out_sx = g.C; q.out_sy = (int64_t)W * g.C; q.out_sb = (int64_t)H * W * g.C;
    q.epi = inverse ? PC_EPI_IGDN : PC_EPI_GDN; q.aux0 = x; q.ld0 = g.C;
    return launch_conv(q, st);
}

// ResidualUnit (layers/layers.py:38-57): x -> gelu(conv1x1) -> gelu(conv3x3) -> conv1x1 + x -> gelu
int ru(hipStream_t st, const RuW& r, const float* x, int C, int B, int H, int W, float* t1, float* t2, float* out)
{
    PCCHK(conv(st, r.c0, {{x, C, C}}, B, H, W, 1, t1, C / 2, PC_EPI_GELU));
    PCCHK(conv(st, r.c2, {{t1, C / 2, C / 2}}, B, H, W, 1, t2, C / 2, PC_EPI_GELU));
    PCCHK(conv(st, r.c4, {{t2, C / 2, C / 2}}, B, H, W, 1, out, C, PC_EPI_RES_GELU, x, C));
    return PC_OK;
}

// Win_noShift_Attention (layers/layers.py:59-75)
int wam(pc_codec* c, hipStream_t st, const WamW& w, const float* x, int B, int H, int W, float* out, int ldo = 0)
{
    const int C = w.C;
    if (ldo == 0) ldo = C;                                              // output pixel stride (a channel slice of a wider tensor)
    const size_t M = (size_t)B * H * W;
    float *t1, *t2, *a0, *a1, *qkv, *o;
    PCCHK(c->buf("wam_t1", M * (C / 2), &t1));
    PCCHK(c->buf("wam_t2", M * (C / 2), &t2));
    PCCHK(c->buf("wam_a0", M * C, &a0));
    PCCHK(c->buf("wam_a1", M * C, &a1));
    PCCHK(c->buf("wam_b0", M * C, &o));
    PCCHK(c->buf("wam_qkv", M * 3 * C, &qkv));
    float* b1;
    PCCHK(c->buf("wam_b1", M * C, &b1));
    // branch a: three residual units
    PCCHK(ru(st, w.a[0], x, C, B, H, W, t1, t2, a0));
    PCCHK(ru(st, w.a[1], a0, C, B, H, W, t1, t2, a1));
    PCCHK(ru(st, w.a[2], a1, C, B, H, W, t1, t2, a0));                  // a in a0
    // branch b: window attention, three residual units, 1x1
    PCCHK(conv(st, w.qkv, {{x, C, C}}, B, H, W, 1, qkv, 3 * C, PC_EPI_NONE));
    PCCHK(pc_win_attention_launch(qkv, w.bias, B, H, W, C, HEADS, w.ws, w.shift, (float)std::pow((double)(C / HEADS), -0.5), a1, st, 1));
    PCCHK(conv(st, w.proj, {{a1, C, C}}, B, H, W, 1, o, C, PC_EPI_RES, x, C));     // shortcut + proj(attn)
    PCCHK(ru(st, w.b[0], o, C, B, H, W, t1, t2, b1));
    PCCHK(ru(st, w.b[1], b1, C, B, H, W, t1, t2, o));
    PCCHK(ru(st, w.b[2], o, C, B, H, W, t1, t2, b1));
    PCCHK(conv(st, w.out, {{b1, C, C}}, B, H, W, 1, out, ldo, PC_EPI_GATE, a0, C, x, C));   // a * sigmoid(b) + x
    return PC_OK;
}

int stack5(pc_codec* c, hipStream_t st, const Stack5W& s, std::initializer_list<Seg> segs, int B, int h, int w,
           float* out, int ldo, int epi, const float* aux0 = nullptr, int ld0 = 0, const float* aux1 = nullptr, int ld1 = 0,
           const char* tag = "s5")
{
    const size_t M = (size_t)B * h * w;
    float *t0, *t1;
    PCCHK(c->buf(std::string(tag) + "_t0", M * 224, &t0));
    PCCHK(c->buf(std::string(tag) + "_t1", M * 176, &t1));
    PCCHK(conv(st, s.c[0], segs, B, h, w, 1, t0, 224, PC_EPI_GELU));
    PCCHK(conv(st, s.c[1], {{t0, 224, 224}}, B, h, w, 1, t1, 176, PC_EPI_GELU));
    PCCHK(conv(st, s.c[2], {{t1, 176, 176}}, B, h, w, 1, t0, 128, PC_EPI_GELU));
    PCCHK(conv(st, s.c[3], {{t0, 128, 128}}, B, h, w, 1, t1, 64, PC_EPI_GELU));
    PCCHK(conv(st, s.c[4], {{t1, 64, 64}}, B, h, w, 1, out, ldo, epi, aux0, ld0, aux1, ld1));
    return PC_OK;
}

// cc_mean || cc_scale of one chain step as five grouped launches (identical shapes; the supports differ only in
// their first segment: latent_means vs latent_scales)
int stack5_pair(pc_codec* c, hipStream_t st, const Stack5W& sm, const Stack5W& ss, std::initializer_list<Seg> segs_mean,
                const float* seg0_scale, int B, int h, int w, float* out_mean, float* out_scale, const char* tagm, const char* tags)
{
    const size_t M = (size_t)B * h * w;
    float *m0, *m1, *s0, *s1;
    PCCHK(c->buf(std::string(tagm) + "_t0", M * 224, &m0));
    PCCHK(c->buf(std::string(tagm) + "_t1", M * 176, &m1));
    PCCHK(c->buf(std::string(tags) + "_t0", M * 224, &s0));
    PCCHK(c->buf(std::string(tags) + "_t1", M * 176, &s1));
    Group1 g{&ss.c[0], seg0_scale, s0};
    PCCHK(conv(st, sm.c[0], segs_mean, B, h, w, 1, m0, 224, PC_EPI_GELU, nullptr, 0, nullptr, 0, false, &g));
    g = Group1{&ss.c[1], s0, s1};
    PCCHK(conv(st, sm.c[1], {{m0, 224, 224}}, B, h, w, 1, m1, 176, PC_EPI_GELU, nullptr, 0, nullptr, 0, false, &g));
    g = Group1{&ss.c[2], s1, s0};
    PCCHK(conv(st, sm.c[2], {{m1, 176, 176}}, B, h, w, 1, m0, 128, PC_EPI_GELU, nullptr, 0, nullptr, 0, false, &g));
    g = Group1{&ss.c[3], s0, s1};
    PCCHK(conv(st, sm.c[3], {{m0, 128, 128}}, B, h, w, 1, m1, 64, PC_EPI_GELU, nullptr, 0, nullptr, 0, false, &g));
    g = Group1{&ss.c[4], s1, out_scale};
    PCCHK(conv(st, sm.c[4], {{m1, 64, 64}}, B, h, w, 1, out_mean, SLICE, PC_EPI_NONE, nullptr, 0, nullptr, 0, false, &g));
    return PC_OK;
}

// hyper-synthesis net (CHProg_cnn.py:208-232): z_hat [B][zh][zw][192] -> out slice [B][4zh][4zw][320] (ld 640)
int hs(pc_codec* c, hipStream_t st, const HsW& h, const float* z, int B, int zh, int zw, float* out, int ldo, const char* tag = "")
{
    const size_t M = (size_t)B * zh * zw;
    float *t0, *t1, *t2, *t3;
    PCCHK(c->buf(std::string("hs_t0") + tag, M * 192, &t0));
    PCCHK(c->buf(std::string("hs_t1") + tag, M * 4 * 224, &t1));
    PCCHK(c->buf(std::string("hs_t2") + tag, M * 4 * 256, &t2));
    PCCHK(c->buf(std::string("hs_t3") + tag, M * 16 * 288, &t3));
    PCCHK(conv(st, h.c0, {{z, 192, 192}}, B, zh, zw, 1, t0, 192, PC_EPI_GELU));
    PCCHK(conv(st, h.c2, {{t0, 192, 192}}, B, zh, zw, 1, t1, 224, PC_EPI_GELU, nullptr, 0, nullptr, 0, true));
    PCCHK(conv(st, h.c4, {{t1, 224, 224}}, B, 2 * zh, 2 * zw, 1, t2, 256, PC_EPI_GELU));
    PCCHK(conv(st, h.c6, {{t2, 256, 256}}, B, 2 * zh, 2 * zw, 1, t3, 288, PC_EPI_GELU, nullptr, 0, nullptr, 0, true));
    PCCHK(conv(st, h.c8, {{t3, 288, 288}}, B, 4 * zh, 4 * zw, 1, out, ldo, PC_EPI_NONE));
    return PC_OK;
}

// one analysis transform (models/cnn.py:34-44): x NCHW [B][3][H][W] -> out [B][H/16][W/16][Cy] written with pixel stride ldy
int g_a_net(pc_codec* c, hipStream_t st, const pc_codec::GaW& g, const float* x, int B, int H, int W, float* y, int ldy)
{
    const int Cy = g.c7.Cout;
    float *t0, *t1, *t2, *t3, *t4;
    PCCHK(c->buf("ga_t0", (size_t)B * (H / 2) * (W / 2) * NCH, &t0));
    PCCHK(c->buf("ga_t1", (size_t)B * (H / 2) * (W / 2) * NCH, &t1));
    PCCHK(c->buf("ga_t2", (size_t)B * (H / 4) * (W / 4) * NCH, &t2));
    PCCHK(c->buf("ga_t3", (size_t)B * (H / 4) * (W / 4) * NCH, &t3));
    PCCHK(c->buf("ga_t4", (size_t)B * (H / 16) * (W / 16) * MLAT, &t4));
    {   // conv 3 -> 192, 5x5 s2, reading the NCHW image directly (element-wise gather path)
        pc_conv_params q;
        std::memset(&q, 0, sizeof(q));
        q.nseg = 1; q.seg[0].ptr = x; q.seg[0].ld = 0; q.seg[0].nch = 3; q.Cin = 3; q.smallc = 1;
        q.B = B; q.H = H; q.W = W;
        q.in_sb = (int64_t)3 * H * W; q.in_sc = (int64_t)H * W; q.in_sy = W; q.in_sx = 1;
        fill_conv_taps(q, 5, 2);
        q.w = g.c0.w; q.bias = g.c0.b; q.Cout = NCH;
        q.Ho = H / 2; q.Wo = W / 2; q.outH = q.Ho; q.outW = q.Wo; q.M = B * q.Ho * q.Wo;
        // g_a.1 (GDN) rides in the same kernel: the 192-channel conv output never leaves the workgroup's LDS (round 4)
        q.fg_gamma = g.g1.gamma; q.fg_beta = g.g1.beta;
        q.out = t1; q.out_sc = 1; q.out_sx = NCH; q.out_sy = (int64_t)q.Wo * NCH; q.out_sb = (int64_t)q.Ho * q.Wo * NCH;
        PCCHK(launch_conv(q, st));
    }
    PCCHK(conv(st, g.c2, {{t1, NCH, NCH}}, B, H / 2, W / 2, 2, t2, NCH, PC_EPI_NONE));
    PCCHK(gdn(st, g.g3, t2, B, H / 4, W / 4, false, t3));
    PCCHK(wam(c, st, g.w4, t3, B, H / 4, W / 4, t2));
    PCCHK(conv(st, g.c5, {{t2, NCH, NCH}}, B, H / 4, W / 4, 2, t0, NCH, PC_EPI_NONE));
    PCCHK(gdn(st, g.g6, t0, B, H / 8, W / 8, false, t1));
    PCCHK(conv(st, g.c7, {{t1, NCH, NCH}}, B, H / 8, W / 8, 2, t4, Cy, PC_EPI_NONE));
    PCCHK(wam(c, st, g.w8, t4, B, H / 16, W / 16, y, ldy));
    return PC_OK;
}

// y = g_a(x), or cat(g_a[0](x), g_a[1](x)) over channels with multiple_encoder (CHProg_cnn.py:691-697): the second net writes
// channels 320.. of the same [M][640] buffer -- a virtual concatenation, no copy
int g_a(pc_codec* c, hipStream_t st, const float* x, int B, int H, int W, float* y)
{
    if (!c->multi_enc) return g_a_net(c, st, c->ga[0], x, B, H, W, y, MLAT);
    PCCHK(g_a_net(c, st, c->ga[0], x, B, H, W, y, MLAT));
    return g_a_net(c, st, c->ga[1], x, B, H, W, y + D0, MLAT);
}

// g_s[k] (CHProg_cnn.py:149-161): y_hat [B][h][w][320] -> x_hat NCHW [B][3][16h][16w], clamped to [0,1]
int g_s(pc_codec* c, hipStream_t st, const GsW& g, const float* yhat, int B, int h, int w, float* x_hat)
{
    float *t0, *t1, *t2;
    PCCHK(c->buf("gs_t0", (size_t)B * h * w * D0, &t0));
    PCCHK(c->buf("gs_t1", (size_t)B * (8 * h) * (8 * w) * NCH, &t1));
    PCCHK(c->buf("gs_t2", (size_t)B * (8 * h) * (8 * w) * NCH, &t2));
    PCCHK(wam(c, st, g.w0, yhat, B, h, w, t0));
    PCCHK(conv(st, g.d1, {{t0, D0, D0}}, B, h, w, 1, t1, NCH, PC_EPI_NONE));
    PCCHK(gdn(st, g.g2, t1, B, 2 * h, 2 * w, true, t2));
    PCCHK(conv(st, g.d3, {{t2, NCH, NCH}}, B, 2 * h, 2 * w, 1, t1, NCH, PC_EPI_NONE));
    PCCHK(gdn(st, g.g4, t1, B, 4 * h, 4 * w, true, t2));
    PCCHK(wam(c, st, g.w5, t2, B, 4 * h, 4 * w, t1));
    PCCHK(conv(st, g.d6, {{t1, NCH, NCH}}, B, 4 * h, 4 * w, 1, t2, NCH, PC_EPI_NONE));
    PCCHK(gdn(st, g.g7, t2, B, 8 * h, 8 * w, true, t1));
    {   // deconv 192 -> 3 in sub-pixel form (load_deconv3_subpixel), output written NCHW with clamp
        pc_conv_params q;
        std::memset(&q, 0, sizeof(q));
        const int H = 8 * h, W = 8 * w;
        q.nseg = 1; q.seg[0].ptr = t1; q.seg[0].ld = NCH; q.seg[0].nch = NCH; q.Cin = NCH;
        q.B = B; q.H = H; q.W = W;
        q.nphase = 1; q.ntap[0] = 9; q.stride = 1; q.osy = q.osx = 1;
        for (int t = 0; t < 9; ++t) { q.dy[0][t] = 1 - t / 3; q.dx[0][t] = 1 - t % 3; q.wtap[0][t] = t; }
        q.w = g.d8.w; q.wlayout = 1; q.bias = g.d8.b; q.Cout = 12;
        q.Ho = H; q.Wo = W; q.outH = 2 * H; q.outW = 2 * W; q.M = B * H * W;
        q.pixel_shuffle = 1;
        q.out = x_hat; q.out_sx = 1; q.out_sy = q.outW; q.out_sc = (int64_t)q.outH * q.outW; q.out_sb = 3 * q.out_sc;
        q.epi = PC_EPI_CLAMP01;
        PCCHK(launch_conv(q, st));
    }
    return PC_OK;
}

int h_a(pc_codec* c, hipStream_t st, const float* y, int B, int h, int w, float* z)
{
    float *t0, *t1;
    PCCHK(c->buf("ha_t0", (size_t)B * h * w * 320, &t0));
    PCCHK(c->buf("ha_t1", (size_t)B * h * w * 288, &t1));
    PCCHK(conv(st, c->ha[0], {{y, MLAT, MLAT}}, B, h, w, 1, t0, 320, PC_EPI_GELU));
    PCCHK(conv(st, c->ha[1], {{t0, 320, 320}}, B, h, w, 1, t1, 288, PC_EPI_GELU));
    PCCHK(conv(st, c->ha[2], {{t1, 288, 288}}, B, h, w, 2, t0, 256, PC_EPI_GELU));
    PCCHK(conv(st, c->ha[3], {{t0, 256, 256}}, B, h / 2, w / 2, 1, t1, 224, PC_EPI_GELU));
    PCCHK(conv(st, c->ha[4], {{t1, 224, 224}}, B, h / 2, w / 2, 2, z, NCH, PC_EPI_NONE));
    return PC_OK;
}

int ensure_host_staging(pc_codec* c, size_t n_int32)
{
    if (c->h_cap >= n_int32) return PC_OK;
    if (c->h_sym) (void)hipHostFree(c->h_sym);
    if (c->h_idx) (void)hipHostFree(c->h_idx);
    c->h_sym = c->h_idx = nullptr; c->h_cap = 0;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&c->h_sym), n_int32 * 4, hipHostMallocDefault));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&c->h_idx), n_int32 * 4, hipHostMallocDefault));
    c->h_cap = n_int32;
    return PC_OK;
}

int mask_mode_for(int mask_pol, double quality, float* q_out)
{
    // layers/masking.py:205-228
    if (mask_pol == PC_MASK_TWO_LEVELS) return quality == 0 ? 3 : 2;
    if (mask_pol == PC_MASK_THREE_LEVELS_STD) {       // :229-247: 0 -> zeros, 2 -> ones, anything else -> the top 20 % by scale
        if (quality == 0) return 3;
        if (quality == 2) return 2;
        *q_out = (float)0.8;
        return 1;
    }
    if (quality >= 10) return 2;
    if (quality == 0) return 3;
    const double pr = quality * 0.1;          // :212
    *q_out = (float)(1.0 - pr);               // :213, converted to the tensor dtype by torch.quantile
    return 1;
}

int hyper(pc_codec* c, hipStream_t st, const float* z_hat, int B, int zh, int zw, double quality, float* lm, float* ls)
{
    // h_scale_s[k] / h_mean_s[k] (CHProg_cnn.py:705-715) all read z_hat and nothing else: their first layers have ~110 workgroups
    // each on 256 CUs, so the two (base only) or four nets CAN run side by side on their own streams with their own workspaces
    // (PC_HYPER_PARALLEL=1).  That was worth 1 % in round 1; with the encoder / decoder objects side by side and the chains pipelined it costs
    // 3 % of the overlapped bench and 2-5 % of the sequential one (profiles/r03_t_hyper_parallel_ab.log): four more streams' launches in a
    // chip that is already shared by four chains.  Default off since round 3.
    static const bool par = pc_tune("PC_HYPER_PARALLEL", 0) != 0;
    const int n = quality != 0 ? 4 : 2;
    if (!par || c->serial_profile() || c->opt_serial) {
        PCCHK(hs(c, st, c->hss[0], z_hat, B, zh, zw, ls, MLAT));
        PCCHK(hs(c, st, c->hms[0], z_hat, B, zh, zw, lm, MLAT));
        if (quality != 0) {                       // CHProg_cnn.py:708-715
            PCCHK(hs(c, st, c->hss[1], z_hat, B, zh, zw, ls + D0, MLAT));
            PCCHK(hs(c, st, c->hms[1], z_hat, B, zh, zw, lm + D0, MLAT));
        }
        return PC_OK;
    }
    if (!c->hyper_ev[0]) {
        for (auto& sx : c->hyper_streams) HIPCHK(hipStreamCreateWithFlags(&sx, hipStreamNonBlocking));
        for (auto& e : c->hyper_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const HsW* nets[4] = {&c->hss[0], &c->hms[0], &c->hss[1], &c->hms[1]};
    float* outs[4] = {ls, lm, ls + D0, lm + D0};
    const char* tags[4] = {"", "_h1", "_h2", "_h3"};
    HIPCHK(hipEventRecord(c->hyper_ev[0], st));                                         // fork: z_hat ready
    for (int i = 0; i < n; ++i) {
        hipStream_t sx = i == 0 ? st : c->hyper_streams[i - 1];
        if (i > 0) HIPCHK(hipStreamWaitEvent(sx, c->hyper_ev[0], 0));
        PCCHK(hs(c, sx, *nets[i], z_hat, B, zh, zw, outs[i], MLAT, tags[i]));
        if (i > 0) { HIPCHK(hipEventRecord(c->hyper_ev[i], sx)); HIPCHK(hipStreamWaitEvent(st, c->hyper_ev[i], 0)); }   // join
    }
    return PC_OK;
}

}  // namespace

// ============================================================================================== C ABI
extern "C" int pc_pack_conv_weight(const float* w, int kind, int Cout, int Cin, int k, float* out)
{
    if (!w || !out || Cout <= 0 || Cin <= 0 || k <= 0) return PC_ERR_ARG;
    auto src = [&](int co, int ci, int ky, int kx) -> float {
        return kind == 0 ? w[(((size_t)co * Cin + ci) * k + ky) * k + kx] : w[(((size_t)ci * Cout + co) * k + ky) * k + kx];
    };
    if (pc_conv_weight_layout(kind, Cin, Cout, k) == 0) {
        // layout 0: [ky*k+kx][Cin][Cout]
        for (int ky = 0; ky < k; ++ky)
            for (int kx = 0; kx < k; ++kx)
                for (int ci = 0; ci < Cin; ++ci) {
                    float* dst = out + (((size_t)(ky * k + kx) * Cin) + ci) * Cout;
                    for (int co = 0; co < Cout; ++co) dst[co] = src(co, ci, ky, kx);
                }
    } else {
        // layout 1: [ky*k+kx][Cout][Cin]  (K contiguous per output channel: 16-byte k-quads for the LDS-DMA kernel)
        for (int ky = 0; ky < k; ++ky)
            for (int kx = 0; kx < k; ++kx)
                for (int co = 0; co < Cout; ++co) {
                    float* dst = out + (((size_t)(ky * k + kx) * Cout) + co) * Cin;
                    for (int ci = 0; ci < Cin; ++ci) dst[ci] = src(co, ci, ky, kx);
                }
    }
    return PC_OK;
}

extern "C" int pc_conv2d_nhwc(const float* x, int B, int H, int W, int Cin, const float* w_packed, const float* bias,
                              int kind, int Cout, int k, int stride, int act, int tile_cfg, float* out, void* stream)
{
    if (!x || !w_packed || !out) return PC_ERR_ARG;
    pc_conv_params q;
    std::memset(&q, 0, sizeof(q));
    q.nseg = 1; q.seg[0].ptr = x; q.seg[0].ld = Cin; q.seg[0].nch = Cin; q.Cin = Cin;
    q.B = B; q.H = H; q.W = W;
    if (Cin % 16) { q.smallc = 1; q.in_sc = 1; q.in_sx = Cin; q.in_sy = (int64_t)W * Cin; q.in_sb = (int64_t)H * W * Cin; if (kind != 0) return PC_ERR_ARG; }
    q.w = w_packed; q.bias = bias; q.Cout = Cout; q.epi = act ? PC_EPI_GELU : PC_EPI_NONE; q.tile_cfg = tile_cfg;
    q.wlayout = pc_conv_weight_layout(kind, Cin, Cout, k);
    if (kind == 0) {
        fill_conv_taps(q, k, stride);
        q.Ho = (H + 2 * (k / 2) - k) / stride + 1; q.Wo = (W + 2 * (k / 2) - k) / stride + 1; q.outH = q.Ho; q.outW = q.Wo;
    } else {
        if (k != 5) return PC_ERR_ARG;
        fill_deconv_taps(q);
        q.Ho = H; q.Wo = W; q.outH = 2 * H; q.outW = 2 * W;
    }
    q.M = B * q.Ho * q.Wo;
    q.out = out; q.out_sc = 1; q.out_sx = Cout; q.out_sy = (int64_t)q.outW * Cout; q.out_sb = (int64_t)q.outH * q.outW * Cout;
    return pc_conv_launch(q, (hipStream_t)stream);
}

extern "C" int pc_gdn_nhwc(const float* x, int B, int H, int W, int C, const float* beta, const float* gamma, int inverse,
                           float* out, void* stream)
{
    if (!x || !beta || !gamma || !out || C % 16) return PC_ERR_ARG;
    GdnW g; g.beta = const_cast<float*>(beta); g.gamma = const_cast<float*>(gamma); g.C = C;
    return gdn((hipStream_t)stream, g, x, B, H, W, inverse != 0, out);
}

extern "C" int pc_win_attention_nhwc(const float* qkv, const float* bias, int B, int H, int W, int C, int heads, int window,
                                     int shift, float* out, void* stream)
{
    if (!qkv || !bias || !out || heads <= 0) return PC_ERR_ARG;
    return pc_win_attention_launch(qkv, bias, B, H, W, C, heads, window, shift, (float)std::pow((double)(C / heads), -0.5) /* win_attention.py:57: head_dim ** -0.5 in Python double, applied as a float32 scalar */, out,
                                   (hipStream_t)stream);
}

extern "C" int pc_mask_quantile_threshold(const float* scale, int ld, int B, int HW, int C, float q, float* thr, void* stream)
{
    if (!scale || !thr) return PC_ERR_ARG;
    return pc_quantile_thr_launch(scale, ld, B, HW, C, q, thr, nullptr, (hipStream_t)stream);
}

namespace {
int prep_encode(const float* scale, int ld_scale, const float* mu, int ld_mu, const float* y, int ld_y, const float* ybase, int ld_ybase,
                const float* thr, int mask_mode, int B, int HW, const float* scale_table, int n_table, float scale_bound, int32_t* sym,
                int32_t* idx, float* mask, float* yhat, int ld_yhat, float* lik, int64_t lik_sb, hipStream_t stream,
                const float* mask_src = nullptr, int64_t mask_sb = 0)
{
    if (!scale || !mu || !y || !sym || !idx || !yhat || !scale_table || (mask_mode == 1 && !thr)) return PC_ERR_ARG;
    pc_prep_params p;
    std::memset(&p, 0, sizeof(p));
    p.B = B; p.HW = HW; p.C = 32;
    p.scale = scale; p.ld_scale = ld_scale; p.mu = mu; p.ld_mu = ld_mu; p.y = y; p.ld_y = ld_y;
    p.ybase = ybase; p.ld_ybase = ld_ybase; p.thr = thr; p.mask_mode = mask_mode;
    p.table = scale_table; p.ntable = n_table; p.bound = scale_bound;
    p.sym = sym; p.idx = idx; p.mask = mask; p.yhat = yhat; p.ld_yhat = ld_yhat;
    p.lik = lik; p.lik_sb = lik_sb; p.mask_src = mask_src; p.mask_sb = mask_sb;
    return pc_prep_enc_launch(p, stream);
}
}  // namespace

extern "C" int pc_gc_prep_encode(const float* scale, int ld_scale, const float* mu, int ld_mu, const float* y, int ld_y,
                                 const float* ybase, int ld_ybase, const float* thr, int mask_mode, int B, int HW,
                                 const float* scale_table, int n_table, float scale_bound,
                                 int32_t* sym, int32_t* idx, float* mask, float* yhat, int ld_yhat, void* stream)
{
    return prep_encode(scale, ld_scale, mu, ld_mu, y, ld_y, ybase, ld_ybase, thr, mask_mode, B, HW, scale_table, n_table, scale_bound, sym,
                       idx, mask, yhat, ld_yhat, nullptr, 0, (hipStream_t)stream);
}

namespace {
int prep_decode_index(const float* scale, int ld_scale, const float* thr, int mask_mode, int B, int HW, const float* scale_table,
                      int n_table, float scale_bound, int32_t* idx, float* mask, hipStream_t stream, const float* mask_src = nullptr,
                      int64_t mask_sb = 0, uint8_t* idx8 = nullptr)
{
    if (!scale || !idx || !scale_table || (mask_mode == 1 && !thr)) return PC_ERR_ARG;
    pc_prep_params p;
    std::memset(&p, 0, sizeof(p));
    p.B = B; p.HW = HW; p.C = 32; p.scale = scale; p.ld_scale = ld_scale; p.thr = thr; p.mask_mode = mask_mode;
    p.table = scale_table; p.ntable = n_table; p.bound = scale_bound; p.idx = idx; p.mask = mask;
    p.mask_src = mask_src; p.mask_sb = mask_sb; p.idx8 = idx8;
    return pc_prep_dec_index_launch(p, stream);
}
}  // namespace

extern "C" int pc_gc_prep_decode_index(const float* scale, int ld_scale, const float* thr, int mask_mode, int B, int HW,
                                       const float* scale_table, int n_table, float scale_bound, int32_t* idx, float* mask,
                                       void* stream)
{
    return prep_decode_index(scale, ld_scale, thr, mask_mode, B, HW, scale_table, n_table, scale_bound, idx, mask, (hipStream_t)stream);
}

extern "C" int pc_gc_dequantize(const int32_t* sym, const float* mu, int ld_mu, int B, int HW, float* yhat, int ld_yhat, void* stream)
{
    if (!sym || !mu || !yhat) return PC_ERR_ARG;
    pc_prep_params p;
    std::memset(&p, 0, sizeof(p));
    p.B = B; p.HW = HW; p.C = 32; p.sym = const_cast<int32_t*>(sym); p.mu = mu; p.ld_mu = ld_mu; p.yhat = yhat; p.ld_yhat = ld_yhat;
    return pc_prep_dec_dequant_launch(p, (hipStream_t)stream);
}

extern "C" int pc_codec_create(pc_codec** out, int device)
{
    if (!out) return PC_ERR_ARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { g_last_hip = (int)e; return PC_ERR_HIP; }   // no GPU: fail loudly, there is no CPU fallback
    if (device < 0 || device >= n) return PC_ERR_ARG;
    HIPCHK(hipSetDevice(device));
    pc_codec* c = new (std::nothrow) pc_codec;
    if (!c) return PC_ERR_NOMEM;
    c->device = device;
    c->rowtabs = pc_rowtab_cache_create((size_t)512 << 20);          // <= 512 MB of row tables per codec, least recently used evicted
    if (!c->rowtabs) { delete c; return PC_ERR_NOMEM; }
    *out = c;
    return PC_OK;
}

extern "C" void pc_codec_destroy(pc_codec* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();                                    // nothing of this codec is in flight when its memory goes
    pc_rowtab_cache_destroy(c->rowtabs);
    for (void* p : c->weight_allocs) (void)hipFree(p);
    for (auto& kv : c->bufs) if (kv.second.p) (void)hipFree(kv.second.p);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    for (auto& L : c->lanes) { (void)hipStreamDestroy(L.sA); (void)hipStreamDestroy(L.sB); (void)hipEventDestroy(L.eA); (void)hipEventDestroy(L.eB); (void)hipEventDestroy(L.eDone); }
    if (c->eFork) (void)hipEventDestroy(c->eFork);
    if (c->call_done) (void)hipEventDestroy(c->call_done);
    if (c->staging_done) (void)hipEventDestroy(c->staging_done);
    for (hipEvent_t e : c->lvl_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->slice_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->pipe_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->hyper_ev) if (e) (void)hipEventDestroy(e);
    for (hipStream_t sx : c->hyper_streams) if (sx) (void)hipStreamDestroy(sx);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->pipe_stream) (void)hipStreamDestroy(c->pipe_stream);
    if (c->h_sym) (void)hipHostFree(c->h_sym);
    if (c->h_idx) (void)hipHostFree(c->h_idx);
    delete c;
}

extern "C" int pc_codec_set_tensor(pc_codec* c, const char* name, const void* data, int dtype, const int64_t* shape, int ndim)
{
    if (!c || !name || (!data && ndim) || ndim < 0 || ndim > 8) return PC_ERR_ARG;
    if (c->finalized) return PC_ERR_STATE;
    size_t n = 1;
    HostTensor t;
    for (int i = 0; i < ndim; ++i) { if (shape[i] < 0) return PC_ERR_ARG; n *= (size_t)shape[i]; t.shape.push_back(shape[i]); }
    const size_t es = dtype == PC_I64 ? 8 : 4;
    t.dtype = dtype;
    t.data.assign(reinterpret_cast<const uint8_t*>(data), reinterpret_cast<const uint8_t*>(data) + n * es);
    c->sd[name] = std::move(t);
    return PC_OK;
}

extern "C" int pc_codec_set_tables(pc_codec* c, int which, const int32_t* cdf, int n_cdf, int cdf_stride, const int32_t* sizes,
                                   const int32_t* offsets)
{
    if (!c || !cdf || !sizes || !offsets || n_cdf <= 0 || cdf_stride < 2 || which < 0 || which > 1) return PC_ERR_ARG;
    for (int i = 0; i < n_cdf; ++i) {          // rans_interface.cpp:48-57 (assert_cdfs), enforced here
        const int32_t* row = cdf + (size_t)i * cdf_stride;
        if (sizes[i] < 2 || sizes[i] > cdf_stride || row[0] != 0 || row[sizes[i] - 1] != (1 << 16)) return PC_ERR_CDF;
        for (int j = 0; j + 1 < sizes[i]; ++j) if (row[j + 1] <= row[j]) return PC_ERR_CDF;
    }
    Tables& t = which == 0 ? c->gc : c->eb;
    t.cdf.assign(cdf, cdf + (size_t)n_cdf * cdf_stride);
    t.len.assign(sizes, sizes + n_cdf);
    t.off.assign(offsets, offsets + n_cdf);
    t.n = n_cdf; t.stride = cdf_stride;
    t.lut.assign((size_t)n_cdf * 256, 0);
    pc::build_decode_lut(t.cdf.data(), t.n, t.stride, t.len.data(), t.lut.data());
    return PC_OK;
}

extern "C" int pc_codec_set_threads(pc_codec* c, int n) { if (!c) return PC_ERR_ARG; c->n_threads = n; return PC_OK; }

// The object's schedule options (the product's replacement for the tuning builds' environment switches; results never depend on them:
// tests/test_gpu_codec.py::test_schedule_options_do_not_change_results).
extern "C" int pc_codec_set_option(pc_codec* c, const char* name, int value)
{
    if (!c || !name || value < 0) return PC_ERR_ARG;
    const std::string n(name);
    if (n == "serial_schedule") c->opt_serial = value ? 1 : 0;
    else if (n == "lanes_enc") { if (value > 8) return PC_ERR_ARG; c->opt_lanes_enc = value; }
    else if (n == "lanes_dec") { if (value > 8) return PC_ERR_ARG; c->opt_lanes_dec = value; }
    else if (n == "host_threads") c->n_threads = value;
    else if (n == "profile_in_schedule") c->profile_in_schedule = value != 0;
    else return PC_ERR_ARG;
    return PC_OK;
}

extern "C" int pc_codec_finalize(pc_codec* c)
{
    if (!c) return PC_ERR_ARG;
    if (c->finalized) return PC_OK;
    HIPCHK(hipSetDevice(c->device));
    c->multi_enc = c->sd.count("g_a.0.0.weight") != 0;                  // ModuleList of two encoders: keys g_a.<k>.<layer>...
    for (int k = 0; k < (c->multi_enc ? 2 : 1); ++k) {
        const std::string p = c->multi_enc ? "g_a." + std::to_string(k) : std::string("g_a");
        const int Cy = c->multi_enc ? D0 : MLAT;
        pc_codec::GaW& g = c->ga[k];
        PCCHK(load_conv(c, p + ".0", 3, NCH, 5, 0, &g.c0));
        PCCHK(load_gdn(c, p + ".1", NCH, &g.g1));
        PCCHK(load_conv(c, p + ".2", NCH, NCH, 5, 0, &g.c2));
        PCCHK(load_gdn(c, p + ".3", NCH, &g.g3));
        PCCHK(load_wam(c, p + ".4", NCH, 8, 4, &g.w4));
        PCCHK(load_conv(c, p + ".5", NCH, NCH, 5, 0, &g.c5));
        PCCHK(load_gdn(c, p + ".6", NCH, &g.g6));
        PCCHK(load_conv(c, p + ".7", NCH, Cy, 5, 0, &g.c7));
        PCCHK(load_wam(c, p + ".8", Cy, 4, 2, &g.w8));
    }
    for (int k = 0; k < 2; ++k) {
        const std::string p = "g_s." + std::to_string(k);
        GsW& g = c->gs[k];
        PCCHK(load_wam(c, p + ".0", D0, 4, 2, &g.w0));
        PCCHK(load_conv(c, p + ".1", D0, NCH, 5, 1, &g.d1));
        PCCHK(load_gdn(c, p + ".2", NCH, &g.g2));
        PCCHK(load_conv(c, p + ".3", NCH, NCH, 5, 1, &g.d3));
        PCCHK(load_gdn(c, p + ".4", NCH, &g.g4));
        PCCHK(load_wam(c, p + ".5", NCH, 8, 4, &g.w5));
        PCCHK(load_conv(c, p + ".6", NCH, NCH, 5, 1, &g.d6));
        PCCHK(load_gdn(c, p + ".7", NCH, &g.g7));
        PCCHK(load_deconv3_subpixel(c, p + ".8", NCH, &g.d8));
    }
    const int ha_c[6] = {MLAT, 320, 288, 256, 224, NCH};
    for (int j = 0; j < 5; ++j) PCCHK(load_conv(c, "h_a." + std::to_string(2 * j), ha_c[j], ha_c[j + 1], 3, 0, &c->ha[j]));
    for (int k = 0; k < 2; ++k) {
        PCCHK(load_hs(c, "h_mean_s." + std::to_string(k), &c->hms[k]));
        PCCHK(load_hs(c, "h_scale_s." + std::to_string(k), &c->hss[k]));
    }
    for (int i = 0; i < NS0; ++i) {
        const std::string s = "." + std::to_string(i);
        PCCHK(load_stack5(c, "cc_mean_transforms" + s, D0 + 32 * std::min(i, 5), &c->cc_mean[i]));
        PCCHK(load_stack5(c, "cc_scale_transforms" + s, D0 + 32 * std::min(i, 5), &c->cc_scale[i]));
        PCCHK(load_stack5(c, "lrp_transforms" + s, D0 + 32 * std::min(i + 1, 6), &c->lrp[i]));
        PCCHK(load_stack5(c, "cc_mean_transforms_prog" + s, D0 + 32 * std::min(i + 1, 6), &c->cc_mean_p[i]));
        PCCHK(load_stack5(c, "cc_scale_transforms_prog" + s, D0 + 32 * std::min(i + 1, 6), &c->cc_scale_p[i]));
        PCCHK(load_stack5(c, "lrp_transforms_prog" + s, D0 + 32 * std::min(i + 2, 7), &c->lrp_p[i]));
    }
    {   // EntropyBottleneck medians = quantiles[:, 0, 1]  (entropy_models.py:350)
        const HostTensor* q = find(c, "entropy_bottleneck.quantiles", PC_F32, {NCH, 1, 3});
        if (!q) return PC_ERR_MISSING;
        std::vector<float> med(NCH);
        for (int i = 0; i < NCH; ++i) med[i] = reinterpret_cast<const float*>(q->data.data())[3 * i + 1];
        PCCHK(upload(c, med, &c->medians));
    }
    {   // EntropyBottleneck density network for the likelihood path: softplus(_matrix_i), _bias_i, tanh(_factor_i)
        // (entropy_models.py:400-418), filters (3,3,3,3); layout of pc_stages.hip: eb_logits
        static const int F[6] = {1, 3, 3, 3, 3, 1};
        std::vector<float> net((size_t)NCH * PC_EB_NET_FLOATS, 0.0f);
        bool ok = true;
        for (int i = 0; i < 5 && ok; ++i) {
            const int fo = F[i + 1], fi = F[i];
            const HostTensor* m = find(c, "entropy_bottleneck._matrix" + std::to_string(i), PC_F32, {NCH, fo, fi});
            const HostTensor* b = find(c, "entropy_bottleneck._bias" + std::to_string(i), PC_F32, {NCH, fo, 1});
            const HostTensor* f = i < 4 ? find(c, "entropy_bottleneck._factor" + std::to_string(i), PC_F32, {NCH, fo, 1}) : nullptr;
            if (!m || !b || (i < 4 && !f)) { ok = false; break; }
            const int base = i == 0 ? 0 : 9 + 15 * (i - 1);
            for (int ch = 0; ch < NCH; ++ch) {
                float* d = net.data() + (size_t)ch * PC_EB_NET_FLOATS + base;
                const float* mm = reinterpret_cast<const float*>(m->data.data()) + (size_t)ch * fo * fi;
                for (int e = 0; e < fo * fi; ++e) d[e] = mm[e] > 20.0f ? mm[e] : log1pf(expf(mm[e]));          // F.softplus
                for (int e = 0; e < fo; ++e) d[fo * fi + e] = reinterpret_cast<const float*>(b->data.data())[(size_t)ch * fo + e];
                if (f) for (int e = 0; e < fo; ++e) d[fo * fi + fo + e] = tanhf(reinterpret_cast<const float*>(f->data.data())[(size_t)ch * fo + e]);
            }
        }
        if (ok) PCCHK(upload(c, net, &c->eb_net));          // absent in a state dict: pc_codec_forward returns PC_ERR_STATE
    }
    {   // scale table from the module buffer (never recomputed), LowerBound from its buffer
        auto it = c->sd.find("gaussian_conditional.scale_table");
        if (it == c->sd.end() || it->second.dtype != PC_F32 || it->second.shape.size() != 1 || it->second.shape[0] < 2 || it->second.shape[0] > 64) return PC_ERR_MISSING;
        c->n_table = (int)it->second.shape[0];
        std::vector<float> tab(reinterpret_cast<const float*>(it->second.data.data()), reinterpret_cast<const float*>(it->second.data.data()) + c->n_table);
        PCCHK(upload(c, tab, &c->scale_table));
        const HostTensor* b = find(c, "gaussian_conditional.lower_bound_scale.bound", PC_F32, {1});
        if (!b) return PC_ERR_MISSING;
        c->scale_bound = reinterpret_cast<const float*>(b->data.data())[0];
    }
    {   // REM weights, if the caller set them (keys post_latent.<level>.<slice>.<subnet>.<block>...): 2 or 3 blocks per sub-net
        // ("middle" / "big"), enc one more
        int levels = 0;
        while (levels < 3 && c->sd.count("post_latent." + std::to_string(levels) + ".0.enc.0.conv1.weight")) ++levels;
        if (levels) {
            const int n_sub = c->sd.count("post_latent.0.0.enc_base_rep.2.conv1.weight") ? 3 : 2;
            {   // mu_std=True nets take cat(mu, scale) in the enhancement-parameter branch: conv1 weight [N][2N][3][3] (CHProgREM.py:30,46)
                auto it = c->sd.find("post_latent.0.0.enc_enh_entropy_params.0.conv1.weight");
                c->rem_mu_std = (it != c->sd.end() && it->second.shape.size() == 4 && it->second.shape[1] == 2 * SLICE) ? 1 : 0;
            }
            const int e_in = c->rem_mu_std ? 2 * SLICE : SLICE, e_out = c->rem_mu_std ? 2 * SLICE : SLICE;
            c->rem.assign((size_t)levels * NS0, LrrW{});
            for (int k = 0; k < levels; ++k)
                for (int i = 0; i < NS0; ++i) {
                    LrrW& L = c->rem[(size_t)k * NS0 + i];
                    L.n_sub = n_sub; L.n_enc = n_sub + 1;
                    const std::string p = "post_latent." + std::to_string(k) + "." + std::to_string(i) + ".";
                    for (int j = 0; j < n_sub; ++j) {
                        const std::string js = "." + std::to_string(j);
                        PCCHK(load_rb(c, p + "enc_base_entropy_params" + js, j == 0 ? 2 * SLICE : SLICE, SLICE, &L.ent_base[j]));
                        PCCHK(load_rb(c, p + "enc_enh_entropy_params" + js, j == 0 ? e_in : SLICE, SLICE, &L.ent_enh[j]));
                        PCCHK(load_rb(c, p + "enc_base_rep" + js, SLICE, SLICE, &L.base_rep[j]));
                    }
                    for (int j = 0; j < L.n_enc; ++j)
                        PCCHK(load_rb(c, p + "enc." + std::to_string(j), j == 0 ? 3 * SLICE : 2 * SLICE, j == L.n_enc - 1 ? e_out : 2 * SLICE, &L.enc[j]));
                }
            c->rem_levels_loaded = levels;
        }
    }
    c->sd.clear();
    c->finalized = true;
    return PC_OK;
}

extern "C" int pc_codec_set_scale_table(pc_codec* c, const float* table, int n)
{
    // GaussianConditional.update_scale_table (entropy_models.py:588-597): a new table of scales for build_indexes; the caller sets the
    // matching CDF tables with pc_codec_set_tables
    if (!c || !table || n < 2 || n > 64) return PC_ERR_ARG;
    if (!c->finalized) return PC_ERR_STATE;
    for (int i = 0; i + 1 < n; ++i) if (!(table[i] < table[i + 1])) return PC_ERR_ARG;     // the index search needs an ascending table
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    std::vector<float> t(table, table + n);
    float* dev = nullptr;
    PCCHK(upload(c, t, &dev));
    c->scale_table = dev; c->n_table = n;
    return PC_OK;
}

extern "C" int pc_codec_set_rem(pc_codec* c, const double* check_levels, int n_levels)
{
    // PostRateProcessedNetwork(base_net, check_levels) (CHProgREM.py:207-235): from now on compress / decompress refine the predicted
    // scale of every enhancement slice with the LatentRateReduction net of the quality's range; n_levels = 0 switches it off again
    if (!c || n_levels < 0 || n_levels > 3 || (n_levels && !check_levels)) return PC_ERR_ARG;
    if (n_levels && (!c->finalized || c->rem_levels_loaded < n_levels)) return PC_ERR_STATE;
    for (int i = 0; i + 1 < n_levels; ++i) if (!(check_levels[i] < check_levels[i + 1])) return PC_ERR_ARG;
    c->rem_n = n_levels;
    for (int i = 0; i < n_levels; ++i) c->rem_check[i] = check_levels[i];
    return PC_OK;
}

extern "C" int pc_codec_set_cust_map(pc_codec* c, const float* cust_map)
{
    if (!c) return PC_ERR_ARG;
    c->cust_map = cust_map;
    return PC_OK;
}

extern "C" int pc_codec_set_rem_checkpoint(pc_codec* c, const float* rep)
{
    // checkpoint_rep of PostRateProcessedNetwork.compress / decompress (CHProgREM.py:676,773 / :901,989): consumed by the next call
    if (!c) return PC_ERR_ARG;
    c->rem_ckpt = rep;
    return PC_OK;
}

extern "C" int pc_codec_num_slices(const pc_codec* c) { return c ? c->res_slices : 0; }

extern "C" int pc_codec_get_string(const pc_codec* c, int slice, int b, const uint8_t** data, size_t* len)
{
    if (!c || !data || !len || b < 0 || b >= c->res_B) return PC_ERR_ARG;
    const std::vector<uint8_t>* s;
    if (slice == -1) s = &c->z_strings[b];
    else if (slice >= 0 && slice < c->res_slices) s = &c->y_strings[(size_t)slice * c->res_B + b];
    else return PC_ERR_ARG;
    *data = s->data(); *len = s->size();
    return PC_OK;
}

// ---------------------------------------------------------------------------------------------- slice chain
// The 20-step chain is serial per image but independent across images, and inside a step the mean and
// scale stacks are independent.  Its GEMMs are small (M = B*h*w rows), so the chain is run as `n_lanes`
// sub-batches, each on its own pair of non-blocking HIP streams (mean/prep/LRP on sA, scale (+quantile) on
// sB, joined with events): up to 2*n_lanes kernels are resident at once and fill the MFMA pipes that one
// such kernel leaves ~50 % idle; in the decoder each lane is driven by its own host thread, so one lane's
// host rANS round trip overlaps the other lanes' GPU work.  Results do not depend on the split
// (numeric contract: no cross-image arithmetic), which tests/test_gpu_codec.py checks.
namespace {

#define PC_DEFAULT_LANES_ENC 1
#define PC_DEFAULT_LANES_DEC 2

// hand-off between the two pipelined chains' host threads: slice i's event has been recorded (or the producer failed)
struct SliceSignal {
    std::mutex m;
    std::condition_variable cv;
    int count = 0;
    bool failed = false;
    void publish(int n) { { std::lock_guard<std::mutex> lk(m); count = n; } cv.notify_all(); }
    void fail() { { std::lock_guard<std::mutex> lk(m); failed = true; } cv.notify_all(); }
    bool wait_for(int i) { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return failed || count > i; }); return !failed; }   // false: producer failed
};

struct ChainCtx {
    pc_codec* c;
    int B, h, w, HW;
    size_t M;                                   // B * HW
    float *y, *lm, *ls, *yb, *ye, *mu, *scale, *thr, *masks;
    int32_t *sym, *idx;
    uint8_t* idx8;                              // decoder: byte copy of idx for the host coder
    int mode; float q; bool enh;
    double quality; int mask_pol;               // REM: the level being coded and the caller's mask policy (range / attention-mask selection)
    int step0, step1;                           // chain steps to run: [0,10) base, [10,20) enhancement
    int level;                                  // enhancement strings of this level sit at slot 10 + 10*level + i
    float* lik; int lik_nch;                    // forward path: y likelihoods, NCHW [B][lik_nch][HW] (null otherwise)
    const float* cust_map;                      // NCHW [B][320][HW]: enhancement masks threshold this map instead of the scale
    const float* rem_ckpt;                      // NCHW [B][320][HW] or null: x_base input of the REM nets instead of the decoded base slices
    hipEvent_t* sig;                            // if set: record sig[i] on the lane's stream once slice i (of this pass) is complete
    SliceSignal* sig_count;                     //         ... and publish the number of recorded events to other host threads
    hipEvent_t* waitv;                          // if set: slice i of this pass starts only after waitv[i] (recorded by the other chain)
    SliceSignal* wait_count;                    //         host side: do not look at waitv[i] before it has been recorded
    size_t h_off;                               // decoder: offset (int32 units) of this chain's region in the pinned staging buffers
    int32_t *so_sym, *so_idx;                   // streamed pass (single lane): pinned destinations; slice i of the pass is copied to
                                                // so_sym + i*M*SLICE on c->copy_stream as soon as its prep kernel is done
};

template <typename T> inline T* img(T* p, int b0, size_t per_image) { return p ? p + (size_t)b0 * per_image : nullptr; }

int ensure_lanes(pc_codec* c, int n)
{
    while ((int)c->lanes.size() < n) {
        pc_codec::Lane L;
        HIPCHK(hipStreamCreateWithFlags(&L.sA, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&L.sB, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&L.eA, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&L.eB, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&L.eDone, hipEventDisableTiming));
        c->lanes.push_back(L);
    }
    if (!c->eFork) HIPCHK(hipEventCreateWithFlags(&c->eFork, hipEventDisableTiming));
    return PC_OK;
}

int lane_count(const pc_codec* c, int B, bool decode)
{
    static const int env = (int)pc_tune("PC_LANES", 0);
    static const int env_e = (int)pc_tune("PC_LANES_ENC", 0);
    static const int env_d = (int)pc_tune("PC_LANES_DEC", 0);
    // Config 2, round-1 final kernels (enc / dec ms per batch): lanes 1/1 35.9 / 36.4, 1/2 35.7 / 35.2, 2/2 37.1 / 36.0 -- the
    // encoder's GEMMs fill the chip best undivided; in the decoder a second lane hides the other lane's host rANS round trips
    int n = decode ? (env_d > 0 ? env_d : (env > 0 ? env : PC_DEFAULT_LANES_DEC)) : (env_e > 0 ? env_e : (env > 0 ? env : PC_DEFAULT_LANES_ENC));
    if (decode ? c->opt_lanes_dec > 0 : c->opt_lanes_enc > 0) n = decode ? c->opt_lanes_dec : c->opt_lanes_enc;      // pc_codec_set_option
    if (c->serial_profile() || c->opt_serial) n = 1;
    return std::max(1, std::min(n, std::min(B, 8)));
}

// per-image mask threshold of enhancement slice i (layers/masking.py:205-223): the (1 - pr/10) quantile of the slice's scale, or of
// the caller's custom map when one was given (:171-194; its slice is a flat [32*HW] run per image, NCHW)
// scratch of the large-image quantile path (null for slices the single-workgroup kernel handles): per lane, reused across steps
int quantile_work(const ChainCtx& k, int nb, const std::string& tag, uint32_t** w)
{
    *w = nullptr;
    if ((int64_t)k.HW * SLICE <= PC_QUANTILE_SMALL_N) return PC_OK;
    return k.c->buf("qwork" + tag, pc_quantile_work_bytes(nb) / sizeof(uint32_t), w);
}

int mask_threshold(const ChainCtx& k, int i, int b0, int nb, const float* sc_i, hipStream_t st, const std::string& tag)
{
    float* thr = k.thr + (size_t)i * k.B + b0;
    uint32_t* w;
    PCCHK(quantile_work(k, nb, tag, &w));
    if (!k.cust_map) return pc_quantile_thr_launch(sc_i, SLICE, nb, k.HW, SLICE, k.q, thr, w, st);
    const float* m = k.cust_map + ((size_t)b0 * D0 + (size_t)SLICE * i) * k.HW;
    return pc_quantile_thr_launch(m, SLICE, nb, k.HW, SLICE, k.q, thr, w, st, (int64_t)D0 * k.HW);
}

// ResidualBlock (models/utils.py:59-87): leaky(conv2(leaky(conv1(x)))) + (skip(x) or x)
int rem_rb(hipStream_t st, const RbW& r, std::initializer_list<Seg> segs, const float* x_single, int ld_x, int B, int h, int w, float* t1, float* ts,
           float* out)
{
    const int co = r.c1.Cout;
    PCCHK(conv(st, r.c1, segs, B, h, w, 1, t1, co, PC_EPI_LEAKY));
    const float* idn = x_single;
    int ld_i = ld_x;
    if (r.has_skip) { PCCHK(conv(st, r.skip, segs, B, h, w, 1, ts, co, PC_EPI_NONE)); idn = ts; ld_i = co; }
    return conv(st, r.c2, {{t1, co, co}}, B, h, w, 1, out, co, PC_EPI_LEAKY_RES, idn, ld_i);
}

// apply_latent_enhancement (CHProgREM.py:375-428) + LatentRateReduction.forward (:74-86) for enhancement slice i, images [b0, b0+nb):
// the predicted scale sc_i [nb*HW][32] is refined in place.  mu_std = False: the means are left alone.
int rem_refine(const ChainCtx& k, int i, int b0, int nb, float* mu_i, float* sc_i, hipStream_t st, const std::string& tag)
{
    pc_codec* c = k.c;
    const double q = k.quality;
    if (!c->rem_n || q <= c->rem_check[0]) return PC_OK;                         // :399-400: nothing below the first check level
    int lvl; double q_bar;                                                        // find_check_quality :446-466, index choice :402-417
    if (c->rem_n == 1) { lvl = 0; q_bar = c->rem_check[0]; }
    else if (q <= c->rem_check[1]) { lvl = 0; q_bar = c->rem_check[0]; }
    else if (c->rem_n == 2) { lvl = 1; q_bar = c->rem_check[1]; }
    else if (q <= c->rem_check[2]) { lvl = 1; q_bar = c->rem_check[1]; }
    else { lvl = 2; q_bar = c->rem_check[2]; }
    const LrrW& L = c->rem[(size_t)lvl * NS0 + i];
    const size_t pi = (size_t)k.HW, m = (size_t)nb * pi;
    const int B = nb, h = k.h, w = k.w;
    float *t1, *ts, *pa, *pb, *fp, *fl, *fb, *thr2;
    PCCHK(c->buf("rem_t1" + tag, m * 64, &t1));
    PCCHK(c->buf("rem_ts" + tag, m * 64, &ts));
    PCCHK(c->buf("rem_pa" + tag, m * 64, &pa));
    PCCHK(c->buf("rem_pb" + tag, m * 64, &pb));
    PCCHK(c->buf("rem_fp" + tag, m * 32, &fp));
    PCCHK(c->buf("rem_fl" + tag, m * 32, &fl));
    PCCHK(c->buf("rem_fb" + tag, m * 32, &fb));
    PCCHK(c->buf("rem_thr" + tag, (size_t)2 * k.B, &thr2));
    // attention mask = round(star - bar), both thresholds on the UNREFINED scale (:386-396)
    float qs = 0, qb = 0;
    // The reference never forwards the caller's mask_pol here (CHProgREM.py:620,832,1060 pass `training` only; the parameter's default
    // at :385 is "point-based-std"): the attention mask is quantile-based whatever policy the block mask uses.
    const int mode_star = mask_mode_for(PC_MASK_POINT_BASED_STD, q, &qs), mode_bar = mask_mode_for(PC_MASK_POINT_BASED_STD, q_bar, &qb);
    uint32_t* qw;
    PCCHK(quantile_work(k, nb, tag, &qw));
    if (mode_star == 1) PCCHK(pc_quantile_thr_launch(sc_i, SLICE, nb, k.HW, SLICE, qs, thr2, qw, st));
    if (mode_bar == 1) PCCHK(pc_quantile_thr_launch(sc_i, SLICE, nb, k.HW, SLICE, qb, thr2 + k.B, qw, st));
    const float* yb_i = img(k.yb, b0, pi * D0) + 32 * i;
    const float* mu_b = k.mu + (size_t)i * k.M * SLICE + (size_t)b0 * pi * SLICE;       // base step i: mu / scale kept per slice
    const float* sd_b = k.scale + (size_t)i * k.M * SLICE + (size_t)b0 * pi * SLICE;
    // f_ent_prog = enc_enh_entropy_params(scale), or of cat(mu, scale) in the mu_std form (:789)
    const float* x = sc_i; int ldx = SLICE;
    for (int j = 0; j < L.n_sub; ++j) {
        float* o = j == L.n_sub - 1 ? fp : (j & 1 ? pb : pa);
        if (j == 0 && c->rem_mu_std) PCCHK(rem_rb(st, L.ent_enh[0], {{mu_i, SLICE, SLICE}, {sc_i, SLICE, SLICE}}, nullptr, 0, B, h, w, t1, ts, o));
        else PCCHK(rem_rb(st, L.ent_enh[j], {{x, ldx, SLICE}}, x, ldx, B, h, w, t1, ts, o));
        x = o; ldx = SLICE;
    }
    // f_latent = enc_base_rep(y_b_hat); y_b_hat = the decoded base slice, or the caller's checkpoint representation (:773,989), which
    // arrives NCHW and is transposed to the chain's NHWC here
    x = yb_i; ldx = D0;
    if (k.rem_ckpt) {
        float* ck;
        PCCHK(c->buf("rem_ck" + tag, m * 32, &ck));
        PCCHK(pc_nchw_slice_to_nhwc_launch(k.rem_ckpt + ((size_t)b0 * D0 + (size_t)SLICE * i) * pi, (int64_t)D0 * (int64_t)pi, nb, (int)pi, SLICE, ck, st));
        x = ck; ldx = SLICE;
    }
    for (int j = 0; j < L.n_sub; ++j) {
        float* o = j == L.n_sub - 1 ? fl : (j & 1 ? pb : pa);
        PCCHK(rem_rb(st, L.base_rep[j], {{x, ldx, SLICE}}, x, ldx, B, h, w, t1, ts, o));
        x = o; ldx = SLICE;
    }
    // f_ent_base = enc_base_entropy_params(cat(mu_base, std_base))
    PCCHK(rem_rb(st, L.ent_base[0], {{mu_b, SLICE, SLICE}, {sd_b, SLICE, SLICE}}, nullptr, 0, B, h, w, t1, ts, L.n_sub == 1 ? fb : pa));
    x = pa; ldx = SLICE;
    for (int j = 1; j < L.n_sub; ++j) {
        float* o = j == L.n_sub - 1 ? fb : (j & 1 ? pb : pa);
        PCCHK(rem_rb(st, L.ent_base[j], {{x, ldx, SLICE}}, x, ldx, B, h, w, t1, ts, o));
        x = o;
    }
    // ret = enc(cat(f_latent, f_ent_base, f_ent_prog))   (:80)
    PCCHK(rem_rb(st, L.enc[0], {{fl, SLICE, SLICE}, {fb, SLICE, SLICE}, {fp, SLICE, SLICE}}, nullptr, 0, B, h, w, t1, ts, pa));
    x = pa;
    for (int j = 1; j < L.n_enc; ++j) {
        float* o = (j & 1) ? pb : pa;
        const int ci = L.enc[j].c1.Cin;
        PCCHK(rem_rb(st, L.enc[j], {{x, ci, ci}}, x, ci, B, h, w, t1, ts, o));
        x = o;
    }
    // scale <- ret * att + scale   (:81-85); mu_std: ret has 2N channels, (mu, scale) <- ret * cat(att, att) + cat(mu, scale)   (:397-398,414-416)
    if (c->rem_mu_std) return pc_rem_combine_launch(x, 2 * SLICE, sc_i, SLICE, nb, k.HW, thr2, mode_star, thr2 + k.B, mode_bar, st, mu_i, SLICE);
    return pc_rem_combine_launch(x, SLICE, sc_i, SLICE, nb, k.HW, thr2, mode_star, thr2 + k.B, mode_bar, st);
}

// mean / scale stacks (+ quantile threshold) of chain step `step` (0..9 base, 10..19 enhancement) for images [b0, b0+nb)
int chain_params(const ChainCtx& k, int step, int b0, int nb, hipStream_t sA, hipStream_t sB, hipEvent_t eA, hipEvent_t eB,
                 const std::string& tag)
{
    pc_codec* c = k.c;
    const size_t pi = (size_t)k.HW;             // pixels per image
    float* lm = img(k.lm, b0, pi * MLAT); float* ls = img(k.ls, b0, pi * MLAT);
    float* yb = img(k.yb, b0, pi * D0); float* ye = img(k.ye, b0, pi * D0);
    float* mu_i = k.mu + (size_t)step * k.M * SLICE + (size_t)b0 * pi * SLICE;
    float* sc_i = k.scale + (size_t)step * k.M * SLICE + (size_t)b0 * pi * SLICE;
    static const bool grouped = pc_tune("PC_GROUPED", 1) != 0;
    const std::string tm = "s5m" + tag, ts = "s5s" + tag;
    if (grouped) {   // one grouped launch per layer: mean (z = 0) and scale (z = 1)
        if (step < NS0) {
            const int i = step, ns = std::min(5, i);
            PCCHK(stack5_pair(c, sA, c->cc_mean[i], c->cc_scale[i], {{lm, MLAT, D0}, {yb, D0, 32 * ns}}, ls, nb, k.h, k.w, mu_i, sc_i, tm.c_str(), ts.c_str()));
        } else {
            const int i = step - NS0, s = std::min(5, i);
            PCCHK(stack5_pair(c, sA, c->cc_mean_p[i], c->cc_scale_p[i], {{lm + D0, MLAT, D0}, {yb + 32 * i, D0, 32}, {ye + 32 * (i - s), D0, 32 * s}},
                              ls + D0, nb, k.h, k.w, mu_i, sc_i, tm.c_str(), ts.c_str()));
            PCCHK(rem_refine(k, i, b0, nb, mu_i, sc_i, sA, tag));                           // REM: refined scale before the mask (CHProgREM.py:812-826)
            if (k.mode == 1) PCCHK(mask_threshold(k, i, b0, nb, sc_i, sA, tag));
        }
        return PC_OK;
    }
    const bool two = sB != sA;
    if (two) { HIPCHK(hipEventRecord(eA, sA)); HIPCHK(hipStreamWaitEvent(sB, eA, 0)); }
    if (step < NS0) {
        const int i = step, ns = std::min(5, i);
        PCCHK(stack5(c, sA, c->cc_mean[i], {{lm, MLAT, D0}, {yb, D0, 32 * ns}}, nb, k.h, k.w, mu_i, SLICE, PC_EPI_NONE, nullptr, 0, nullptr, 0, tm.c_str()));
        PCCHK(stack5(c, sB, c->cc_scale[i], {{ls, MLAT, D0}, {yb, D0, 32 * ns}}, nb, k.h, k.w, sc_i, SLICE, PC_EPI_NONE, nullptr, 0, nullptr, 0, ts.c_str()));
    } else {
        const int i = step - NS0, s = std::min(5, i);
        PCCHK(stack5(c, sA, c->cc_mean_p[i], {{lm + D0, MLAT, D0}, {yb + 32 * i, D0, 32}, {ye + 32 * (i - s), D0, 32 * s}}, nb, k.h, k.w, mu_i, SLICE, PC_EPI_NONE, nullptr, 0, nullptr, 0, tm.c_str()));
        PCCHK(stack5(c, sB, c->cc_scale_p[i], {{ls + D0, MLAT, D0}, {yb + 32 * i, D0, 32}, {ye + 32 * (i - s), D0, 32 * s}}, nb, k.h, k.w, sc_i, SLICE, PC_EPI_NONE, nullptr, 0, nullptr, 0, ts.c_str()));
        if (two && c->rem_n && c->rem_mu_std) { HIPCHK(hipEventRecord(eA, sA)); HIPCHK(hipStreamWaitEvent(sB, eA, 0)); }   // the refinement reads (and rewrites) mu, which stream A has just produced
        PCCHK(rem_refine(k, i, b0, nb, mu_i, sc_i, sB, tag));
        if (k.mode == 1) PCCHK(mask_threshold(k, i, b0, nb, sc_i, sB, tag));   // :819-824
    }
    if (two) { HIPCHK(hipEventRecord(eB, sB)); HIPCHK(hipStreamWaitEvent(sA, eB, 0)); }
    return PC_OK;
}

// LRP stack of chain step `step` (in place on the decoded slice), images [b0, b0+nb)
int chain_lrp(const ChainCtx& k, int step, int b0, int nb, hipStream_t sA, const std::string& tag)
{
    pc_codec* c = k.c;
    const size_t pi = (size_t)k.HW;
    float* lm = img(k.lm, b0, pi * MLAT);
    float* yb = img(k.yb, b0, pi * D0); float* ye = img(k.ye, b0, pi * D0);
    const std::string tm = "s5m" + tag;
    if (step < NS0) {
        const int i = step;
        if (i < 5)
            return stack5(c, sA, c->lrp[i], {{lm, MLAT, D0}, {yb, D0, 32 * (i + 1)}}, nb, k.h, k.w, yb + 32 * i, D0, PC_EPI_LRP, yb + 32 * i, D0, nullptr, 0, tm.c_str());
        return stack5(c, sA, c->lrp[i], {{lm, MLAT, D0}, {yb, D0, 160}, {yb + 32 * i, D0, 32}}, nb, k.h, k.w, yb + 32 * i, D0, PC_EPI_LRP, yb + 32 * i, D0, nullptr, 0, tm.c_str());
    }
    const int i = step - NS0, s = std::min(5, i);
    return stack5(c, sA, c->lrp_p[i], {{lm + D0, MLAT, D0}, {yb + 32 * i, D0, 32}, {ye + 32 * (i - s), D0, 32 * (s + 1)}}, nb, k.h, k.w,
                  ye + 32 * i, D0, PC_EPI_LRP_ADD, ye + 32 * i, D0, yb + 32 * i, D0, tm.c_str());
}

int encode_lane(const ChainCtx& k, int b0, int nb, hipStream_t sA, hipStream_t sB, hipEvent_t eA, hipEvent_t eB, const std::string& tag)
{
    pc_codec* c = k.c;
    const size_t pi = (size_t)k.HW;
    for (int step = k.step0; step < k.step1; ++step) {
        if (k.waitv) {                                                                   // pipelined against the other chain
            const int i = step >= NS0 ? step - NS0 : step;
            if (k.wait_count && !k.wait_count->wait_for(i)) return PC_ERR_STATE;             // the other chain failed
            HIPCHK(hipStreamWaitEvent(sA, k.waitv[i], 0));
        }
        PCCHK(chain_params(k, step, b0, nb, sA, sB, eA, eB, tag));
        const size_t so = (size_t)step * k.M * SLICE + (size_t)b0 * pi * SLICE;
        // forward path: likelihood of slice `step` of image b at lik[((b * lik_nch) + 32 * step + c) * HW + p]
        float* lik = k.lik ? k.lik + ((size_t)b0 * k.lik_nch + (size_t)32 * step) * pi : nullptr;
        const int64_t lik_sb = (int64_t)k.lik_nch * (int64_t)pi;
        if (step < NS0) {                                                                // base slices, :729-764
            PCCHK(prep_encode(k.scale + so, SLICE, k.mu + so, SLICE, img(k.y, b0, pi * MLAT) + 32 * step, MLAT, nullptr, 0, nullptr, 0,
                              nb, k.HW, c->scale_table, c->n_table, c->scale_bound, k.sym + so, k.idx + so, nullptr,
                              img(k.yb, b0, pi * D0) + 32 * step, D0, lik, lik_sb, sA));
        } else {                                                                         // enhancement slices, :775-845
            const int i = step - NS0;
            float* m = k.masks ? k.masks + (size_t)i * k.M * SLICE + (size_t)b0 * pi * SLICE : nullptr;
            PCCHK(prep_encode(k.scale + so, SLICE, k.mu + so, SLICE, img(k.y, b0, pi * MLAT) + 32 * step, MLAT,
                              img(k.y, b0, pi * MLAT) + 32 * i, MLAT, k.thr + (size_t)i * k.B + b0, k.mode, nb, k.HW,
                              c->scale_table, c->n_table, c->scale_bound, k.sym + so, k.idx + so, m,
                              img(k.ye, b0, pi * D0) + 32 * i, D0, lik, lik_sb, sA,
                              k.cust_map ? k.cust_map + ((size_t)b0 * D0 + (size_t)SLICE * i) * pi : nullptr, (int64_t)D0 * (int64_t)pi));
        }
        if (k.so_sym && b0 == 0 && nb == k.B) {                                          // hand this slice's symbols to the host now
            const int i = step >= NS0 ? step - NS0 : step;
            const size_t ns = k.M * SLICE;
            HIPCHK(hipEventRecord(c->slice_ev[i], sA));
            HIPCHK(hipStreamWaitEvent(c->copy_stream, c->slice_ev[i], 0));
            HIPCHK(hipMemcpyAsync(k.so_sym + (size_t)i * ns, k.sym + so, ns * 4, hipMemcpyDeviceToHost, c->copy_stream));
            HIPCHK(hipMemcpyAsync(k.so_idx + (size_t)i * ns, k.idx + so, ns * 4, hipMemcpyDeviceToHost, c->copy_stream));
            HIPCHK(hipEventRecord(c->slice_ev[NS0 + i], c->copy_stream));
        }
        PCCHK(chain_lrp(k, step, b0, nb, sA, tag));
        if (k.sig) {
            const int i = step >= NS0 ? step - NS0 : step;
            HIPCHK(hipEventRecord(k.sig[i], sA));
            if (k.sig_count) k.sig_count->publish(i + 1);
        }
    }
    return PC_OK;
}

int decode_lane(const ChainCtx& k, int b0, int nb, hipStream_t sA, hipStream_t sB, hipEvent_t eA, hipEvent_t eB, const std::string& tag,
                const uint8_t* const* y_strings, const size_t* y_lens, int nt)
{
    pc_codec* c = k.c;
    HIPCHK(hipSetDevice(c->device));
    g_rowtabs = c->rowtabs;                                                             // (this may be a lane's own host thread)
    g_prof = c->profile ? c : nullptr;
    const size_t pi = (size_t)k.HW, per = (size_t)SLICE * k.HW;
    int32_t* h_idx = c->h_idx + k.h_off + (size_t)b0 * per;
    int32_t* h_sym = c->h_sym + k.h_off + (size_t)b0 * per;
    for (int step = k.step0; step < k.step1; ++step) {
        if (k.waitv) {                                                                   // pipelined against the other chain
            const int i = step >= NS0 ? step - NS0 : step;
            if (k.wait_count && !k.wait_count->wait_for(i)) return PC_ERR_STATE;             // the other chain failed
            HIPCHK(hipStreamWaitEvent(sA, k.waitv[i], 0));
        }
        PCCHK(chain_params(k, step, b0, nb, sA, sB, eA, eB, tag));
        const size_t so = (size_t)step * k.M * SLICE + (size_t)b0 * pi * SLICE;
        const bool e = step >= NS0;
        const int i = e ? step - NS0 : step;
        // the host coder reads the indexes back as bytes (a quarter of the int32 traffic on the per-slice critical path)
        PCCHK(prep_decode_index(k.scale + so, SLICE, e ? k.thr + (size_t)i * k.B + b0 : nullptr, e ? k.mode : 0, nb, k.HW,
                                c->scale_table, c->n_table, c->scale_bound, k.idx + so, nullptr, sA,
                                (e && k.cust_map) ? k.cust_map + ((size_t)b0 * D0 + (size_t)SLICE * i) * pi : nullptr, (int64_t)D0 * (int64_t)pi,
                                k.idx8 + so));
        static const bool slow_dec = pc_tune("PC_DEC_FAST", 1) == 0;   // A/B switch
        uint8_t* h_idx8 = reinterpret_cast<uint8_t*>(h_idx);
        if (slow_dec) HIPCHK(hipMemcpyAsync(h_idx, k.idx + so, per * nb * 4, hipMemcpyDeviceToHost, sA));
        else HIPCHK(hipMemcpyAsync(h_idx8, k.idx8 + so, per * nb, hipMemcpyDeviceToHost, sA));
        HIPCHK(hipStreamSynchronize(sA));
        const auto td0 = std::chrono::steady_clock::now();
        const size_t slot = e ? (size_t)NS0 + (size_t)NS0 * k.level + i : (size_t)step;
        if (slow_dec)
            PCCHK(pc_rans_decode_batch(y_strings + slot * k.B + b0, y_lens + slot * k.B + b0, nb, h_idx, per, c->gc.cdf.data(), c->gc.n, c->gc.stride,
                                       c->gc.len.data(), c->gc.off.data(), h_sym, nt));
        else
            PCCHK(pc::rans_decode_u8_batch(y_strings + slot * k.B + b0, y_lens + slot * k.B + b0, nb, h_idx8, per, c->gc.dec(), h_sym, nt));   // :894,969
        { std::lock_guard<std::mutex> lk(c->buf_mu); c->t_host_decode_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - td0).count(); }
        HIPCHK(hipMemcpyAsync(k.sym + so, h_sym, per * nb * 4, hipMemcpyHostToDevice, sA));
        float* dst = e ? img(k.ye, b0, pi * D0) + 32 * i : img(k.yb, b0, pi * D0) + 32 * i;
        PCCHK(pc_gc_dequantize(k.sym + so, k.mu + so, SLICE, nb, k.HW, dst, D0, sA));                                          // :896,971
        PCCHK(chain_lrp(k, step, b0, nb, sA, tag));
        // the H2D copy out of h_sym is ordered before the next step's host decode by that step's hipStreamSynchronize(sA)
        if (k.sig) {
            const int i = step >= NS0 ? step - NS0 : step;
            HIPCHK(hipEventRecord(k.sig[i], sA));
            if (k.sig_count) k.sig_count->publish(i + 1);
        }
    }
    return PC_OK;
}

// run the chain over all images: fork lanes off `st`, join back
int run_chain(const ChainCtx& k, hipStream_t st, bool decode, const uint8_t* const* y_strings, const size_t* y_lens)
{
    pc_codec* c = k.c;
    const int nl = lane_count(c, k.B, decode);
    if (nl == 1) {   // sequential on the caller's stream (also the profiling configuration)
        static const bool two = pc_tune("PC_DUAL_STREAM", 1) != 0;
        if (!two || c->serial_profile() || c->opt_serial) {
            return decode ? decode_lane(k, 0, k.B, st, st, nullptr, nullptr, "", y_strings, y_lens, c->n_threads == 1 ? 1 : 0)
                          : encode_lane(k, 0, k.B, st, st, nullptr, nullptr, "");
        }
    }
    PCCHK(ensure_lanes(c, nl));
    // pre-create every lane's workspace (no allocation inside lanes / threads)
    for (int g = 0; g < nl; ++g) {
        const int b0 = (int)((long)k.B * g / nl), nb = (int)((long)k.B * (g + 1) / nl) - b0;
        const size_t m = (size_t)nb * k.HW;
        float* dummy;
        for (const char* base : {"s5m", "s5s"}) {
            PCCHK(c->buf(std::string(base) + std::to_string(g) + "_t0", m * 224, &dummy));
            PCCHK(c->buf(std::string(base) + std::to_string(g) + "_t1", m * 176, &dummy));
        }
    }
    HIPCHK(hipEventRecord(c->eFork, st));
    for (int g = 0; g < nl; ++g) HIPCHK(hipStreamWaitEvent(c->lanes[g].sA, c->eFork, 0));     // every fallible fork step BEFORE a thread exists
    std::vector<int> rcs(nl, PC_OK);
    std::vector<std::thread> threads;
    for (int g = 0; g < nl; ++g) {
        const int b0 = (int)((long)k.B * g / nl), nb = (int)((long)k.B * (g + 1) / nl) - b0;
        pc_codec::Lane& L = c->lanes[g];
        const std::string tag = std::to_string(g);
        if (decode) {
            threads.emplace_back([&, g, b0, nb, tag] {
                pc_codec::Lane& LL = c->lanes[g];
                rcs[g] = decode_lane(k, b0, nb, LL.sA, LL.sB, LL.eA, LL.eB, tag, y_strings, y_lens, c->n_threads == 1 ? 1 : 0);
            });
        } else {
            rcs[g] = encode_lane(k, b0, nb, L.sA, L.sB, L.eA, L.eB, tag);
        }
    }
    for (auto& t : threads) t.join();                                                          // nothing between creation and join can return
    int rc = PC_OK;
    for (int g = 0; g < nl; ++g) {                                                             // join every lane, remember the first error
        pc_codec::Lane& L = c->lanes[g];
        hipError_t e = hipEventRecord(L.eDone, L.sA);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, L.eDone, 0);
        if (e != hipSuccess && rc == PC_OK) { g_last_hip = (int)e; rc = PC_ERR_HIP; }
    }
    for (int g = 0; g < nl; ++g) if (rcs[g] != PC_OK) return rcs[g];
    return rc;
}

}  // namespace

// ---------------------------------------------------------------------------------------------- compress
namespace {

// host rANS of `n_slots` x B y-streams (symbols / indexes at hs / hi, [slot][B][32*HW]) into string slots first_slot..., plus the B
// z-streams when z_sym != nullptr   (entropy_models.py:226-235 -> compressai.ans)
int encode_streams(pc_codec* c, const int32_t* hs, const int32_t* hi, int first_slot, int n_slots, int B, size_t per,
                   const int32_t* z_sym, const int32_t* z_idx, size_t per_z)
{
    std::atomic<int> rc{PC_OK};
    const size_t n_y_jobs = (size_t)n_slots * B;
    auto job = [&](size_t j) {
        const bool is_z = j >= n_y_jobs;
        const size_t n = is_z ? per_z : per;
        const int32_t* sym = is_z ? z_sym + (j - n_y_jobs) * per_z : hs + j * per;
        const int32_t* idx = is_z ? z_idx + (j - n_y_jobs) * per_z : hi + j * per;
        const Tables& t = is_z ? c->eb : c->gc;
        std::vector<uint8_t>& dst = is_z ? c->z_strings[j - n_y_jobs] : c->y_strings[(size_t)first_slot * B + j];
        dst.resize(pc_rans_bound(n));
        size_t len = 0;
        const int r = pc_rans_encode_with_indexes(sym, idx, n, t.cdf.data(), t.n, t.stride, t.len.data(), t.off.data(), dst.data(),
                                                  dst.size(), &len);
        if (r != PC_OK) rc = r;
        dst.resize(len);
    };
    const size_t n_jobs = n_y_jobs + (z_sym ? (size_t)B : 0);
    if (c->n_threads == 1) for (size_t j = 0; j < n_jobs; ++j) job(j);
    else pc::default_pool().parallel_for(n_jobs, job);
    return rc;
}

// Base chain || enhancement chain.  Enhancement slice i needs base slice i and the enhancement slices before it, nothing else
// (CHProg_cnn.py:775-845: support = y_hat_slices[i] + y_hat_slices_quality[i-5:i]), so the two ten-step chains can run one slice
// apart on two streams over the WHOLE batch: the N = 64 / 32 tail layers of one chain fill the CUs the other leaves idle, and in
// the decoder one chain's host rANS round trip hides behind the other chain's kernels -- without halving M as batch lanes do.
bool pipeline_enabled(const pc_codec* c)
{
    static const bool on = pc_tune("PC_PIPELINE", 1) != 0;
    return on && !c->serial_profile() && !c->opt_serial;
}

// Second set of the level-specific buffers of a chain (decoded enhancement slices, per-slice mu / scale / symbols / indexes, mask
// thresholds): with it the enhancement chains of TWO levels of a multi-level call run side by side -- they depend on the base slices
// only (CHProg_cnn.py:775-845 / :930-983), never on each other.  Set 0 is the object's ordinary buffers (the taps tests read).
int second_level_set(pc_codec* c, const ChainCtx& k, bool decoder, ChainCtx* k2)
{
    *k2 = k;
    PCCHK(c->buf("yhat_enh_L2", k.M * D0, &k2->ye));
    PCCHK(c->buf("mu_L2", k.M * SLICE * 2 * NS0, &k2->mu));
    PCCHK(c->buf("scale_L2", k.M * SLICE * 2 * NS0, &k2->scale));
    PCCHK(c->buf("thr_L2", (size_t)k.B * NS0, &k2->thr));
    PCCHK(c->buf("sym_L2", k.M * SLICE * 2 * NS0, &k2->sym));
    PCCHK(c->buf("idx_L2", k.M * SLICE * 2 * NS0, &k2->idx));
    if (decoder) PCCHK(c->buf("idx8_L2", k.M * SLICE * 2 * NS0, &k2->idx8));
    return PC_OK;
}

int ensure_pipeline(pc_codec* c, size_t M)
{
    if (!c->pipe_stream) {
        HIPCHK(hipStreamCreateWithFlags(&c->pipe_stream, hipStreamNonBlocking));
        c->pipe_ev.resize(NS0 + 2);
        for (auto& e : c->pipe_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    float* dummy;                                     // both chains' stack workspaces, allocated up front
    for (const char* tag : {"s5mPA", "s5sPA", "s5mPB", "s5sPB"}) {
        PCCHK(c->buf(std::string(tag) + "_t0", M * 224, &dummy));
        PCCHK(c->buf(std::string(tag) + "_t1", M * 176, &dummy));
    }
    return PC_OK;
}

// One call at a time per object: its workspaces, streams and staging buffers are the call's.  A second host thread entering the same
// object gets PC_ERR_STATE instead of silently sharing them (two objects side by side are fine: bench.py).
struct BusyGuard {
    pc_codec* c;
    bool ok;
    explicit BusyGuard(pc_codec* cc) : c(cc), ok(false) { bool f = false; ok = c->busy.compare_exchange_strong(f, true, std::memory_order_acquire); }
    ~BusyGuard() { if (ok) c->busy.store(false, std::memory_order_release); }
};
// Calls into one object may arrive on different caller streams (decompress() returns with x_hat still in flight on its stream): the
// object's workspaces must not be reused before the previous call's work is done.  Every call records `call_done` on its stream when
// it returns, by whichever path, and the next call's stream waits for it first -- a no-op when it is the same stream.
struct CallOrder {
    pc_codec* c;
    hipStream_t st;
    int rc;
    CallOrder(pc_codec* cc, hipStream_t s) : c(cc), st(s), rc(PC_OK)
    {
        if (!c->call_done) { if (hipEventCreateWithFlags(&c->call_done, hipEventDisableTiming) != hipSuccess) { c->call_done = nullptr; rc = PC_ERR_HIP; } }
        else if (hipStreamWaitEvent(st, c->call_done, 0) != hipSuccess) rc = PC_ERR_HIP;
    }
    ~CallOrder()
    {
        // whatever path the call left by: work it queued on the object's second chain stream is joined into the caller's stream first
        // (a failed call must not leave a chain of its own running beside the next call)
        if (c->pipe_stream && c->pipe_ev.size() > (size_t)NS0 + 1 && hipEventRecord(c->pipe_ev[NS0 + 1], c->pipe_stream) == hipSuccess)
            (void)hipStreamWaitEvent(st, c->pipe_ev[NS0 + 1], 0);
        if (c->call_done) (void)hipEventRecord(c->call_done, st);
    }
};

// compress() for a list of mask levels.  Everything that does not depend on the level -- g_a, h_a, the hyper-latent
// strings, h_s and the ten base slices (CHProg_cnn.py:692-767) -- runs once; the enhancement chain (:775-845) runs once per
// level with quality > 0.  The GPU always has the next pass queued while the host entropy-codes the previous one:
//   GPU : g_a h_a h_s | base chain | D2H | enh chain L0 | D2H | enh chain L1 | D2H | ...
//   host:                              wait, rANS base + z    wait, rANS L0        wait, rANS L1
// (two pinned staging buffers for the enhancement symbols, used alternately).
int compress_impl(pc_codec* c, const float* x, int B, int H, int W, const double* qualities, int n_levels, int mask_pol,
                  float* const* masks_out, hipStream_t st)
{
    if (!c || !x || !qualities || n_levels < 1 || B <= 0 || H <= 0 || W <= 0 || (H % 64) || (W % 64)) return PC_ERR_ARG;
    if (mask_pol < PC_MASK_POINT_BASED_STD || mask_pol > PC_MASK_THREE_LEVELS_STD) return PC_ERR_ARG;
    if (!c->finalized || !c->gc.ok() || !c->eb.ok()) return PC_ERR_STATE;
    if (c->eb.n != NCH) return PC_ERR_STATE;
    BusyGuard busy(c);
    if (!busy.ok) return PC_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    CallOrder order(c, st);
    if (order.rc != PC_OK) return order.rc;
    g_prof = c->profile ? c : nullptr;
    g_rowtabs = c->rowtabs;
    const int h = H / 16, w = W / 16, zh = H / 64, zw = W / 64, HW = h * w, ZHW = zh * zw;
    const size_t M = (size_t)B * HW;
    bool any_enh = false;
    for (int l = 0; l < n_levels; ++l) any_enh = any_enh || !(qualities[l] <= 0);
    c->last_B = B; c->last_h16 = h; c->last_w16 = w;

    ChainCtx k;
    std::memset(&k, 0, sizeof(k));
    float *z, *z_hat;
    int32_t* z_sym;
    PCCHK(c->buf("y", M * MLAT, &k.y));
    PCCHK(c->buf("z", (size_t)B * ZHW * NCH, &z));
    PCCHK(c->buf("z_hat", (size_t)B * ZHW * NCH, &z_hat));
    PCCHK(c->buf("z_sym", (size_t)B * ZHW * NCH, &z_sym));
    PCCHK(c->buf("latent_means", M * MLAT, &k.lm));
    PCCHK(c->buf("latent_scales", M * MLAT, &k.ls));
    PCCHK(c->buf("yhat_base", M * D0, &k.yb));
    PCCHK(c->buf("yhat_enh", M * D0, &k.ye));
    PCCHK(c->buf("mu", M * SLICE * 2 * NS0, &k.mu));            // per-slice mu / scale kept for taps
    PCCHK(c->buf("scale", M * SLICE * 2 * NS0, &k.scale));
    PCCHK(c->buf("thr", (size_t)B * NS0, &k.thr));
    PCCHK(c->buf("sym", M * SLICE * 2 * NS0, &k.sym));
    PCCHK(c->buf("idx", M * SLICE * 2 * NS0, &k.idx));
    k.c = c; k.B = B; k.h = h; k.w = w; k.HW = HW; k.M = M;
    k.cust_map = c->cust_map; c->cust_map = nullptr;
    k.rem_ckpt = c->rem_ckpt; c->rem_ckpt = nullptr;

    const size_t n_half = (size_t)NS0 * M * SLICE, n_z = (size_t)B * ZHW * NCH;      // symbols of one pass / of z
    const size_t per = (size_t)SLICE * HW, per_z = (size_t)NCH * ZHW;
    PCCHK(ensure_host_staging(c, 3 * n_half + n_z));                                // [base][enh A][enh B][z]
    while ((int)c->lvl_events.size() < n_levels + 1) {
        hipEvent_t e;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->lvl_events.push_back(e);
    }
    c->res_slices = NS0 + NS0 * n_levels; c->res_B = B;
    c->res_level_coded.assign(n_levels, 0);
    c->y_strings.assign((size_t)c->res_slices * B, {});
    c->z_strings.assign(B, {});
    static const bool timing = std::getenv("PC_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    double t_host = 0.0, t_enc_last = 0.0;

    PCCHK(g_a(c, st, x, B, H, W, k.y));                                                  // :692
    PCCHK(h_a(c, st, k.y, B, h, w, z));                                                  // :700
    PCCHK(pc_eb_quant_launch(z, B, ZHW, NCH, c->medians, z_sym, z_hat, st));             // :702-704
    PCCHK(hyper(c, st, z_hat, B, zh, zw, any_enh ? 1.0 : 0.0, k.lm, k.ls));              // :705-715
    // The LAST coded level has no later GPU pass to hide its entropy coding behind: with a single encoder lane its slices are copied
    // out one by one (side stream) and coded while the rest of the chain still runs; only the last slice's coding is exposed.
    int last_coded = -1, first_coded = -1;
    for (int l = 0; l < n_levels; ++l) if (!(qualities[l] <= 0)) { last_coded = l; if (first_coded < 0) first_coded = l; }
    // Default OFF since round 3 (PC_NO_STREAMED_ENCODE=0 switches it on): the per-slice copies and events of the side stream cost the
    // overlapped bench 3.4 % (46.7 -> 48.2 MP/s) and buy a sequential caller 0.6 ms of host coding per call -- nothing measurable
    // (profiles/r03_t_streamed_encode_ab.log)
    static const bool no_stream = pc_tune("PC_NO_STREAMED_ENCODE", 1) != 0;
    const bool can_stream = !no_stream && lane_count(c, B, false) == 1;
    if (can_stream && !c->copy_stream) {
        HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        c->slice_ev.resize(2 * NS0);
        for (auto& e : c->slice_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    auto base_d2h = [&]() -> int {
        HIPCHK(hipMemcpyAsync(c->h_sym, k.sym, n_half * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(c->h_idx, k.idx, n_half * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(c->h_sym + 3 * n_half, z_sym, n_z * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(c->lvl_events[0], st));
        return PC_OK;
    };
    const bool piped = pipeline_enabled(c) && first_coded >= 0 && lane_count(c, B, false) == 1;
    // Two levels in flight (round 4): with more than one coded level the enhancement chains alternate between two buffer sets and two
    // streams -- level j of the coded ones on pipe_stream / set 0 (j even) or on `st` / set 1 (j odd) -- so that level j+1's chain runs
    // beside level j's instead of behind it (the chains are independent; each alone leaves the MFMA pipes as idle as a sequential
    // Config-2 step does).  The host codes level j-1 while both run, as before.
    int n_coded_total = 0;
    for (int l = 0; l < n_levels; ++l) n_coded_total += !(qualities[l] <= 0) ? 1 : 0;
    const bool two_levels = piped && !can_stream && n_coded_total >= 2;
    ChainCtx k2;
    if (two_levels) PCCHK(second_level_set(c, k, false, &k2));
    if (piped) {
        // base slice t on `st`, enhancement slice t-1 of the first coded level on pipe_stream, enqueued alternately
        PCCHK(ensure_pipeline(c, M));
        HIPCHK(hipEventRecord(c->pipe_ev[NS0], st));
        HIPCHK(hipStreamWaitEvent(c->pipe_stream, c->pipe_ev[NS0], 0));
        ChainCtx kb = k, ke = k;
        kb.enh = false; kb.mode = 0; kb.sig = c->pipe_ev.data();
        ke.enh = true; ke.level = first_coded; ke.waitv = c->pipe_ev.data();
        ke.mode = mask_mode_for(k.cust_map ? PC_MASK_POINT_BASED_STD : mask_pol, qualities[first_coded], &ke.q);
        ke.quality = qualities[first_coded]; ke.mask_pol = mask_pol;
        ke.masks = masks_out ? masks_out[first_coded] : nullptr;
        if (can_stream && first_coded == last_coded) { ke.so_sym = c->h_sym + n_half; ke.so_idx = c->h_idx + n_half; }
        for (int t = 0; t <= NS0; ++t) {
            if (t < NS0) {
                kb.step0 = t; kb.step1 = t + 1;
                PCCHK(encode_lane(kb, 0, B, st, st, nullptr, nullptr, "PA"));                // :729-767
                if (t == NS0 - 1) PCCHK(base_d2h());
            }
            if (t >= 1) {
                ke.step0 = NS0 + t - 1; ke.step1 = NS0 + t;
                PCCHK(encode_lane(ke, 0, B, c->pipe_stream, c->pipe_stream, nullptr, nullptr, "PB"));   // :775-845
            }
        }
        if (!two_levels) {                                                                // (two levels in flight: `st` goes on with the next level; joined at the end)
            HIPCHK(hipEventRecord(c->pipe_ev[NS0 + 1], c->pipe_stream));
            HIPCHK(hipStreamWaitEvent(st, c->pipe_ev[NS0 + 1], 0));
        }
    } else {
        k.step0 = 0; k.step1 = NS0; k.enh = false; k.mode = 0;
        PCCHK(run_chain(k, st, false, nullptr, nullptr));                                // :729-767
        PCCHK(base_d2h());
    }

    // what the host still has to entropy-code once its event has fired: -1 = base + z, else the level index
    int pending = -1, pending_buf = 0, n_coded = 0;
    int rc = PC_OK;
    auto drain = [&](int what, int bufsel) -> int {
        const double th = now();
        double te = th;
        int r;
        if (what < 0) {
            HIPCHK(hipEventSynchronize(c->lvl_events[0]));
            te = now();
            for (size_t e = 0; e < n_z; ++e) c->h_idx[3 * n_half + e] = (int32_t)((e / ZHW) % NCH);   // EntropyBottleneck._build_indexes :492-502
            r = encode_streams(c, c->h_sym, c->h_idx, 0, NS0, B, per, c->h_sym + 3 * n_half, c->h_idx + 3 * n_half, per_z);
        } else {
            HIPCHK(hipEventSynchronize(c->lvl_events[1 + what]));
            te = now();
            const size_t off = (size_t)(1 + bufsel) * n_half;
            r = encode_streams(c, c->h_sym + off, c->h_idx + off, NS0 + NS0 * what, NS0, B, per, nullptr, nullptr, per_z);
        }
        t_host += now() - th;
        t_enc_last = now() - te;                             // pure host rANS time of this pass (the last one is not overlapped)
        return r;
    };
    for (int l = 0; l < n_levels; ++l) {
        if (qualities[l] <= 0) continue;                                                 // base only: nothing level-specific to code
        k.step0 = NS0; k.step1 = 2 * NS0; k.enh = true; k.level = l;
        k.mode = mask_mode_for(k.cust_map ? PC_MASK_POINT_BASED_STD : mask_pol, qualities[l], &k.q);   // a custom map overrides the policy
        k.quality = qualities[l]; k.mask_pol = mask_pol;
        k.masks = masks_out ? masks_out[l] : nullptr;
        const int bufsel = n_coded & 1;
        const size_t off = (size_t)(1 + bufsel) * n_half;
        const bool pre = piped && l == first_coded;                                      // already enqueued, pipelined with the base chain
        if (can_stream && l == last_coded) {
            if (!pre) {
                k.so_sym = c->h_sym + off; k.so_idx = c->h_idx + off;
                PCCHK(run_chain(k, st, false, nullptr, nullptr));                        // :775-845, slices streamed out
                k.so_sym = k.so_idx = nullptr;
            }
            c->res_level_coded[l] = 1;
            ++n_coded;
            int r = drain(pending, pending_buf);                                         // the pass before, meanwhile
            if (r != PC_OK) rc = r;
            pending = -2;                                                                // nothing left to drain afterwards
            const double te = now();
            for (int i = 0; i < NS0; ++i) {
                HIPCHK(hipEventSynchronize(c->slice_ev[NS0 + i]));
                const double ts = now();
                r = encode_streams(c, c->h_sym + off + (size_t)i * M * SLICE, c->h_idx + off + (size_t)i * M * SLICE, NS0 + NS0 * l + i, 1, B, per,
                                   nullptr, nullptr, per_z);
                if (r != PC_OK) rc = r;
                if (i == NS0 - 1) t_enc_last = now() - ts;
            }
            t_host += now() - te;
            continue;
        }
        if (two_levels) {
            // coded level number n_coded: even -> set 0 on pipe_stream (the first one is already enqueued there, pipelined with the base
            // chain), odd -> set 1 on `st` (behind the base chain, which it needs and which ran there)
            const bool odd = n_coded & 1;
            ChainCtx kl = odd ? k2 : k;
            kl.step0 = NS0; kl.step1 = 2 * NS0; kl.enh = true; kl.level = l; kl.mode = k.mode; kl.q = k.q; kl.quality = k.quality; kl.mask_pol = k.mask_pol;
            kl.masks = k.masks;
            hipStream_t sx = odd ? st : c->pipe_stream;
            if (!pre) PCCHK(encode_lane(kl, 0, B, sx, sx, nullptr, nullptr, odd ? "PA" : "PB"));   // :775-845
            HIPCHK(hipMemcpyAsync(c->h_sym + off, kl.sym + n_half, n_half * 4, hipMemcpyDeviceToHost, sx));
            HIPCHK(hipMemcpyAsync(c->h_idx + off, kl.idx + n_half, n_half * 4, hipMemcpyDeviceToHost, sx));
            HIPCHK(hipEventRecord(c->lvl_events[1 + l], sx));
            c->res_level_coded[l] = 1;
            ++n_coded;
            const int r = drain(pending, pending_buf);                                   // the level before last, meanwhile
            if (r != PC_OK) rc = r;
            pending = l; pending_buf = bufsel;
            continue;
        }
        if (!pre) PCCHK(run_chain(k, st, false, nullptr, nullptr));                      // :775-845
        HIPCHK(hipMemcpyAsync(c->h_sym + off, k.sym + n_half, n_half * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(c->h_idx + off, k.idx + n_half, n_half * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(c->lvl_events[1 + l], st));
        c->res_level_coded[l] = 1;
        ++n_coded;
        // this level is queued on the GPU: entropy-code the pass before it meanwhile (its staging buffer is the other one)
        const int r = drain(pending, pending_buf);
        if (r != PC_OK) rc = r;
        pending = l; pending_buf = bufsel;
    }
    if (pending != -2) {
        const int r = drain(pending, pending_buf);
        if (r != PC_OK) rc = r;
    }
    if (two_levels) {                                                                    // pipe_stream back into the caller's stream
        HIPCHK(hipEventRecord(c->pipe_ev[NS0 + 1], c->pipe_stream));
        HIPCHK(hipStreamWaitEvent(st, c->pipe_ev[NS0 + 1], 0));
    }
    HIPCHK(hipStreamSynchronize(st));                                                    // masks_out complete for the caller
    c->t_compress_ms = now() - t0; c->t_host_encode_ms = t_host; c->t_host_encode_exposed_ms = t_enc_last;
    c->n_sym_encoded = (double)B * ZHW * NCH + (double)n_half * (1 + n_coded);
    if (timing) std::fprintf(stderr, "[pcodec] compress: %d level(s), %d coded; total %.2f ms; host rANS of the last pass (exposed) %.2f ms\n",
                             n_levels, n_coded, now() - t0, t_enc_last);
    return rc;
}

}  // namespace

extern "C" int pc_codec_compress(pc_codec* c, const float* x, int B, int H, int W, double quality, int mask_pol,
                                 float* masks_out, void* stream)
{
    float* const m[1] = {masks_out};
    const int r = compress_impl(c, x, B, H, W, &quality, 1, mask_pol, masks_out ? m : nullptr, (hipStream_t)stream);
    if (r == PC_OK && quality <= 0) {                           // compress() at quality 0 returns the ten base slices only (:766-767):
        c->res_slices = NS0;                                    // pc_codec_num_slices, _strings_size and _copy_strings all say 10*B + B
        c->y_strings.resize((size_t)NS0 * B);
    }
    return r;
}

extern "C" int pc_codec_compress_levels(pc_codec* c, const float* x, int B, int H, int W, const double* qualities, int n_levels,
                                        int mask_pol, float* const* masks_out, void* stream)
{
    return compress_impl(c, x, B, H, W, qualities, n_levels, mask_pol, masks_out, (hipStream_t)stream);
}

// Bulk transfer of the last compress()'s strings: one copy instead of one call per string (672 strings at Config 2; the per-call
// overhead of a ctypes / cgo / JNI binding is what this avoids).  Order: slot 0..res_slices-1, images 0..B-1 inside a slot, then z.
extern "C" int pc_codec_strings_size(const pc_codec* c, size_t* total_bytes, int* n_strings)
{
    if (!c || !total_bytes || !n_strings) return PC_ERR_ARG;
    size_t t = 0;
    for (const auto& s : c->y_strings) t += s.size();
    for (const auto& s : c->z_strings) t += s.size();
    *total_bytes = t; *n_strings = (int)(c->y_strings.size() + c->z_strings.size());
    return PC_OK;
}

extern "C" int pc_codec_copy_strings(const pc_codec* c, uint8_t* dst, size_t cap, size_t* lens, size_t lens_cap)
{
    if (!c || !dst || !lens) return PC_ERR_ARG;
    if (lens_cap < c->y_strings.size() + c->z_strings.size()) return PC_ERR_BUFFER;
    size_t off = 0, k = 0;
    for (const std::vector<std::vector<uint8_t>>* v : {&c->y_strings, &c->z_strings})
        for (const auto& s : *v) {
            if (off + s.size() > cap) return PC_ERR_BUFFER;
            if (!s.empty()) std::memcpy(dst + off, s.data(), s.size());
            off += s.size(); lens[k++] = s.size();
        }
    return PC_OK;
}

extern "C" int pc_codec_get_level_string(const pc_codec* c, int level, int slice, int b, const uint8_t** data, size_t* len)
{
    if (!c || !data || !len || b < 0 || b >= c->res_B || level < 0 || level >= (int)c->res_level_coded.size()) return PC_ERR_ARG;
    const std::vector<uint8_t>* s;
    if (slice == -1) s = &c->z_strings[b];
    else if (slice >= 0 && slice < NS0) s = &c->y_strings[(size_t)slice * c->res_B + b];
    else if (slice < 2 * NS0 && c->res_level_coded[level]) s = &c->y_strings[((size_t)NS0 + (size_t)NS0 * level + (slice - NS0)) * c->res_B + b];
    else return PC_ERR_ARG;
    *data = s->data(); *len = s->size();
    return PC_OK;
}

// ---------------------------------------------------------------------------------------------- forward (likelihood path)
// forward_single_quality in eval mode (CHProg_cnn.py:1002-1198): the encoder chain without entropy
// coding -- the likelihood of every quantised latent element under its Gaussian (entropy_models.py:626-659) comes out of the same
// fused mask / index / quantise kernel, the hyper-latent's from the EntropyBottleneck density network (:400-433) -- followed by the
// synthesis transform.  x_hat equals decompress(compress(x)) bit for bit (same y_hat); estimated bits = -sum(log2(likelihood)).
extern "C" int pc_codec_forward(pc_codec* c, const float* x, int B, int H, int W, double quality, int mask_pol, float* x_hat,
                                float* y_lik, float* z_lik, float* masks_out, int force_enhanced, void* stream)
{
    if (!c || !x || !x_hat || !y_lik || !z_lik || B <= 0 || H <= 0 || W <= 0 || (H % 64) || (W % 64)) return PC_ERR_ARG;
    if (mask_pol < PC_MASK_POINT_BASED_STD || mask_pol > PC_MASK_THREE_LEVELS_STD) return PC_ERR_ARG;
    if (!c->finalized || !c->eb_net) return PC_ERR_STATE;
    BusyGuard busy(c);
    if (!busy.ok) return PC_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    CallOrder order(c, st);
    if (order.rc != PC_OK) return order.rc;
    g_prof = c->profile ? c : nullptr;
    g_rowtabs = c->rowtabs;
    const int h = H / 16, w = W / 16, zh = H / 64, zw = W / 64, HW = h * w, ZHW = zh * zw;
    const size_t M = (size_t)B * HW;
    const bool enh = quality != 0 || force_enhanced != 0;                                // :1063 "if quality == 0 and force_enhanced is False"
    c->last_B = B; c->last_h16 = h; c->last_w16 = w;

    ChainCtx k;
    std::memset(&k, 0, sizeof(k));
    float *z, *z_hat;
    int32_t* z_sym;
    PCCHK(c->buf("y", M * MLAT, &k.y));
    PCCHK(c->buf("z", (size_t)B * ZHW * NCH, &z));
    PCCHK(c->buf("z_hat", (size_t)B * ZHW * NCH, &z_hat));
    PCCHK(c->buf("z_sym", (size_t)B * ZHW * NCH, &z_sym));
    PCCHK(c->buf("latent_means", M * MLAT, &k.lm));
    PCCHK(c->buf("latent_scales", M * MLAT, &k.ls));
    PCCHK(c->buf("yhat_base", M * D0, &k.yb));
    PCCHK(c->buf("yhat_enh", M * D0, &k.ye));
    PCCHK(c->buf("mu", M * SLICE * 2 * NS0, &k.mu));
    PCCHK(c->buf("scale", M * SLICE * 2 * NS0, &k.scale));
    PCCHK(c->buf("thr", (size_t)B * NS0, &k.thr));
    PCCHK(c->buf("sym", M * SLICE * 2 * NS0, &k.sym));
    PCCHK(c->buf("idx", M * SLICE * 2 * NS0, &k.idx));
    k.c = c; k.B = B; k.h = h; k.w = w; k.HW = HW; k.M = M;
    k.lik = y_lik; k.lik_nch = enh ? 2 * D0 : D0;

    PCCHK(g_a(c, st, x, B, H, W, k.y));                                                  // :1013
    PCCHK(h_a(c, st, k.y, B, h, w, z));                                                  // compute_hyperprior :399
    PCCHK(pc_eb_quant_launch(z, B, ZHW, NCH, c->medians, z_sym, z_hat, st));             // :401-403
    PCCHK(pc_eb_likelihood_launch(z_sym, B, ZHW, NCH, c->medians, c->eb_net, z_lik, st));   // :400
    PCCHK(hyper(c, st, z_hat, B, zh, zw, enh ? 1.0 : 0.0, k.lm, k.ls));                  // :404-417
    if (enh && pipeline_enabled(c) && lane_count(c, B, false) == 1) {
        // base slice t || enhancement slice t-1, as in compress_impl
        PCCHK(ensure_pipeline(c, M));
        HIPCHK(hipEventRecord(c->pipe_ev[NS0], st));
        HIPCHK(hipStreamWaitEvent(c->pipe_stream, c->pipe_ev[NS0], 0));
        ChainCtx kb = k, ke = k;
        kb.enh = false; kb.mode = 0; kb.sig = c->pipe_ev.data();
        ke.enh = true; ke.level = 0; ke.waitv = c->pipe_ev.data(); ke.masks = masks_out;
        ke.mode = mask_mode_for(mask_pol, quality, &ke.q);
        ke.quality = quality; ke.mask_pol = mask_pol;
        for (int t = 0; t <= NS0; ++t) {
            if (t < NS0) { kb.step0 = t; kb.step1 = t + 1; PCCHK(encode_lane(kb, 0, B, st, st, nullptr, nullptr, "PA")); }                       // :1033-1061
            if (t >= 1) { ke.step0 = NS0 + t - 1; ke.step1 = NS0 + t; PCCHK(encode_lane(ke, 0, B, c->pipe_stream, c->pipe_stream, nullptr, nullptr, "PB")); }   // :1089-1160
        }
        HIPCHK(hipEventRecord(c->pipe_ev[NS0 + 1], c->pipe_stream));
        HIPCHK(hipStreamWaitEvent(st, c->pipe_ev[NS0 + 1], 0));
    } else {
        k.step0 = 0; k.step1 = NS0; k.enh = false; k.mode = 0;
        PCCHK(run_chain(k, st, false, nullptr, nullptr));                                // :1033-1061
        if (enh) {
            k.step0 = NS0; k.step1 = 2 * NS0; k.enh = true; k.level = 0;
            k.mode = mask_mode_for(mask_pol, quality, &k.q);
            k.quality = quality; k.mask_pol = mask_pol;
            k.masks = masks_out;
            PCCHK(run_chain(k, st, false, nullptr, nullptr));                            // :1089-1160
        }
    }
    PCCHK(g_s(c, st, c->gs[enh ? 1 : 0], enh ? k.ye : k.yb, B, h, w, x_hat));            // :1065 / :1166-1170
    return PC_OK;
}

// ---------------------------------------------------------------------------------------------- decompress
namespace {

// decompress() for a list of levels: z, h_s and the ten base slices are decoded once (:855-904); every level with quality > 0
// decodes its own enhancement chain (:930-983) and runs the enhancement synthesis; a level with quality 0 runs the base synthesis
// (:907-916).  y_strings / y_lens: [(10 + 10 * n_levels) * B] in the slot order of compress (slots of quality-0 levels are ignored).
int decompress_impl(pc_codec* c, const uint8_t* const* y_strings, const size_t* y_lens, const uint8_t* const* z_strings,
                    const size_t* z_lens, int B, int zh, int zw, const double* qualities, int n_levels, int mask_pol, float* x_hat,
                    hipStream_t st)
{
    if (!c || !y_strings || !y_lens || !z_strings || !z_lens || !x_hat || !qualities || n_levels < 1 || B <= 0 || zh <= 0 || zw <= 0) return PC_ERR_ARG;
    if (mask_pol < PC_MASK_POINT_BASED_STD || mask_pol > PC_MASK_THREE_LEVELS_STD) return PC_ERR_ARG;
    if (!c->finalized || !c->gc.ok() || !c->eb.ok() || c->eb.n != NCH) return PC_ERR_STATE;
    BusyGuard busy(c);
    if (!busy.ok) return PC_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    CallOrder order(c, st);
    if (order.rc != PC_OK) return order.rc;
    g_prof = c->profile ? c : nullptr;
    g_rowtabs = c->rowtabs;
    const int h = 4 * zh, w = 4 * zw, HW = h * w, ZHW = zh * zw;
    const size_t M = (size_t)B * HW;
    bool any_enh = false;
    for (int l = 0; l < n_levels; ++l) any_enh = any_enh || qualities[l] != 0;
    c->last_B = B; c->last_h16 = h; c->last_w16 = w;

    ChainCtx k;
    std::memset(&k, 0, sizeof(k));
    float* z_hat;
    int32_t* z_sym;
    PCCHK(c->buf("z_hat", (size_t)B * ZHW * NCH, &z_hat));
    PCCHK(c->buf("z_sym", (size_t)B * ZHW * NCH, &z_sym));
    PCCHK(c->buf("latent_means", M * MLAT, &k.lm));
    PCCHK(c->buf("latent_scales", M * MLAT, &k.ls));
    PCCHK(c->buf("yhat_base", M * D0, &k.yb));
    PCCHK(c->buf("yhat_enh", M * D0, &k.ye));
    PCCHK(c->buf("mu", M * SLICE * 2 * NS0, &k.mu));
    PCCHK(c->buf("scale", M * SLICE * 2 * NS0, &k.scale));
    PCCHK(c->buf("thr", (size_t)B * NS0, &k.thr));
    PCCHK(c->buf("sym", M * SLICE * 2 * NS0, &k.sym));
    PCCHK(c->buf("idx", M * SLICE * 2 * NS0, &k.idx));
    PCCHK(c->buf("idx8", M * SLICE * 2 * NS0, &k.idx8));
    k.c = c; k.B = B; k.h = h; k.w = w; k.HW = HW; k.M = M;
    k.cust_map = c->cust_map; c->cust_map = nullptr;
    k.rem_ckpt = c->rem_ckpt; c->rem_ckpt = nullptr;
    const size_t per = (size_t)SLICE * HW, per_z = (size_t)NCH * ZHW;
    // decompress() returns with work still in flight -- x_hat, and the chains' LAST host-to-device symbol copies, which read the pinned
    // staging asynchronously.  The host is about to write that staging again (the z symbols below, then every slice's) and may re-allocate it:
    // wait until the previous call's chains have run (an event behind its last chain, not behind its synthesis transform).  Round 4: found by
    // the CodecPipeline test under the env matrix -- the next call's z decode overwrote lane 0's pending symbols of the last slice, 1 run in 8.
    if (c->staging_pending) { HIPCHK(hipEventSynchronize(c->staging_done)); c->staging_pending = false; }
    if (!c->staging_done) HIPCHK(hipEventCreateWithFlags(&c->staging_done, hipEventDisableTiming));
    struct StagingMark {                         // recorded behind every chain section (`st` has joined the lanes / the second chain stream by then), i.e. in
        pc_codec* c; hipStream_t st; bool marked; // front of the synthesis transforms, whose tail the next call's host work may overlap; and, on whatever
        void mark() { if (hipEventRecord(c->staging_done, st) == hipSuccess) { c->staging_pending = true; marked = true; } }   // path the call leaves by, at the exit
        ~StagingMark() { if (!marked) mark(); }
    } staging_mark{c, st, false};
    PCCHK(ensure_host_staging(c, std::max(per * B, per_z * B)));
    const int nt = c->n_threads == 1 ? 1 : 0;

    // z: host rANS decode -> device dequantise   (:855)
    for (size_t e = 0; e < per_z * B; ++e) c->h_idx[e] = (int32_t)((e / ZHW) % NCH);
    PCCHK(pc_rans_decode_batch(z_strings, z_lens, B, c->h_idx, per_z, c->eb.cdf.data(), c->eb.n, c->eb.stride, c->eb.len.data(),
                               c->eb.off.data(), c->h_sym, nt));
    HIPCHK(hipMemcpyAsync(z_sym, c->h_sym, per_z * B * 4, hipMemcpyHostToDevice, st));
    PCCHK(pc_eb_dequant_launch(z_sym, B, ZHW, NCH, c->medians, z_hat, st));
    HIPCHK(hipStreamSynchronize(st));   // h_sym is reused by the lanes
    PCCHK(hyper(c, st, z_hat, B, zh, zw, any_enh ? 1.0 : 0.0, k.lm, k.ls));              // :856-867
    c->t_host_decode_ms = 0.0;
    int first_enh = -1;
    for (int l = 0; l < n_levels && first_enh < 0; ++l) if (qualities[l] != 0) first_enh = l;
    static const bool pipe_dec = pc_tune("PC_PIPELINE_DEC", 1) != 0;
    const bool piped = pipeline_enabled(c) && pipe_dec && first_enh >= 0;
    if (piped) {
        // base chain on `st` (own host thread) || enhancement chain of the first coded level on pipe_stream (this thread), one slice
        // apart: each chain's host rANS round trip hides behind the other chain's kernels, at full M
        PCCHK(ensure_pipeline(c, M));
        PCCHK(ensure_host_staging(c, std::max(2 * per * B, per_z * B)));
        HIPCHK(hipEventRecord(c->pipe_ev[NS0], st));
        HIPCHK(hipStreamWaitEvent(c->pipe_stream, c->pipe_ev[NS0], 0));
        SliceSignal recorded;
        ChainCtx kb = k, ke = k;
        kb.step0 = 0; kb.step1 = NS0; kb.enh = false; kb.mode = 0; kb.sig = c->pipe_ev.data(); kb.sig_count = &recorded; kb.h_off = 0;
        ke.step0 = NS0; ke.step1 = 2 * NS0; ke.enh = true; ke.level = first_enh; ke.waitv = c->pipe_ev.data(); ke.wait_count = &recorded;
        ke.h_off = per * B;
        ke.mode = mask_mode_for(k.cust_map ? PC_MASK_POINT_BASED_STD : mask_pol, qualities[first_enh], &ke.q);
        ke.quality = qualities[first_enh]; ke.mask_pol = mask_pol;
        int rb = PC_OK;
        std::thread tb([&] {
            rb = decode_lane(kb, 0, B, st, st, nullptr, nullptr, "PA", y_strings, y_lens, nt);                  // :874-904
            if (rb != PC_OK) recorded.fail();                                                                  // release the other chain
        });
        const int re = decode_lane(ke, 0, B, c->pipe_stream, c->pipe_stream, nullptr, nullptr, "PB", y_strings, y_lens, nt);   // :930-983
        tb.join();
        // join pipe_stream back into `st` whatever happened: a failed call must not leave work of its own running beside the next one
        const hipError_t ej = hipEventRecord(c->pipe_ev[NS0 + 1], c->pipe_stream);
        const hipError_t ew = ej == hipSuccess ? hipStreamWaitEvent(st, c->pipe_ev[NS0 + 1], 0) : ej;
        if (rb != PC_OK || re != PC_OK) { (void)hipStreamSynchronize(c->pipe_stream); (void)hipStreamSynchronize(st); }
        if (rb != PC_OK) return rb;
        if (re != PC_OK) return re;
        HIPCHK(ew);
    } else {
        k.step0 = 0; k.step1 = NS0; k.enh = false; k.mode = 0;
        PCCHK(run_chain(k, st, true, y_strings, y_lens));                                // :874-904
    }
    staging_mark.mark();
    const size_t img_elems = (size_t)B * 3 * (16 * h) * (16 * w);
    // Two levels in flight (round 4): the enhancement chains of the levels that are still to decode are independent of each other, and a
    // chain alone leaves the GPU idle through every slice's host entropy decode (at 4K: 93 of 148 ms).  They are decoded in PAIRS -- one
    // level on `st` (own host thread, buffer set 1), the other on pipe_stream (this thread, set 0) -- so that each chain's host round trip
    // hides behind the other's kernels, as the base || first-level pipeline above does; the synthesis transforms follow on `st`.
    std::vector<int> rest;
    for (int l = 0; l < n_levels; ++l) if (qualities[l] != 0 && !(piped && l == first_enh)) rest.push_back(l);
    if (piped && !rest.empty()) {
        ChainCtx k2;
        PCCHK(second_level_set(c, k, true, &k2));
        for (int l = 0; l < n_levels; ++l) if (qualities[l] == 0) PCCHK(g_s(c, st, c->gs[0], k.yb, B, h, w, x_hat + (size_t)l * img_elems));   // :907-916
        PCCHK(g_s(c, st, c->gs[1], k.ye, B, h, w, x_hat + (size_t)first_enh * img_elems));                                               // :986-990
        auto level_ctx = [&](const ChainCtx& base, int l, size_t h_off) {
            ChainCtx q = base;
            q.step0 = NS0; q.step1 = 2 * NS0; q.enh = true; q.level = l; q.waitv = nullptr; q.wait_count = nullptr; q.sig = nullptr; q.sig_count = nullptr;
            q.mode = mask_mode_for(k.cust_map ? PC_MASK_POINT_BASED_STD : mask_pol, qualities[l], &q.q);
            q.quality = qualities[l]; q.mask_pol = mask_pol; q.h_off = h_off;
            return q;
        };
        for (size_t p = 0; p < rest.size(); p += 2) {
            const int la = rest[p], lb = p + 1 < rest.size() ? rest[p + 1] : -1;
            const ChainCtx ka = level_ctx(k2, la, 0);
            int ra = PC_OK, rb = PC_OK;
            if (lb >= 0) {
                const ChainCtx kb2 = level_ctx(k, lb, per * B);
                HIPCHK(hipEventRecord(c->pipe_ev[NS0], st));                             // set 0 is free once the g_s that read it has run
                HIPCHK(hipStreamWaitEvent(c->pipe_stream, c->pipe_ev[NS0], 0));
                std::thread ta([&] { ra = decode_lane(ka, 0, B, st, st, nullptr, nullptr, "PA", y_strings, y_lens, nt); });
                rb = decode_lane(kb2, 0, B, c->pipe_stream, c->pipe_stream, nullptr, nullptr, "PB", y_strings, y_lens, nt);
                ta.join();
                const hipError_t ej = hipEventRecord(c->pipe_ev[NS0 + 1], c->pipe_stream);
                const hipError_t ew = ej == hipSuccess ? hipStreamWaitEvent(st, c->pipe_ev[NS0 + 1], 0) : ej;
                if (ra != PC_OK || rb != PC_OK) { (void)hipStreamSynchronize(c->pipe_stream); (void)hipStreamSynchronize(st); }
                if (ra != PC_OK) return ra;
                if (rb != PC_OK) return rb;
                HIPCHK(ew);
            } else {
                PCCHK(decode_lane(ka, 0, B, st, st, nullptr, nullptr, "PA", y_strings, y_lens, nt));
            }
            staging_mark.mark();
            PCCHK(g_s(c, st, c->gs[1], ka.ye, B, h, w, x_hat + (size_t)la * img_elems));
            if (lb >= 0) PCCHK(g_s(c, st, c->gs[1], k.ye, B, h, w, x_hat + (size_t)lb * img_elems));
        }
    } else
    for (int l = 0; l < n_levels; ++l) {
        float* out = x_hat + (size_t)l * img_elems;
        if (qualities[l] == 0) {
            PCCHK(g_s(c, st, c->gs[0], k.yb, B, h, w, out));                             // :907-916
            continue;
        }
        if (!(piped && l == first_enh)) {
            k.step0 = NS0; k.step1 = 2 * NS0; k.enh = true; k.level = l;
            k.mode = mask_mode_for(k.cust_map ? PC_MASK_POINT_BASED_STD : mask_pol, qualities[l], &k.q);
            k.quality = qualities[l]; k.mask_pol = mask_pol;
            PCCHK(run_chain(k, st, true, y_strings, y_lens));                            // :930-983
            staging_mark.mark();
        }
        PCCHK(g_s(c, st, c->gs[1], k.ye, B, h, w, out));                                 // :986-990
    }
    {
        int n_enh = 0;
        for (int l = 0; l < n_levels; ++l) n_enh += qualities[l] != 0 ? 1 : 0;
        c->n_sym_decoded = (double)per_z * B + (double)per * B * NS0 * (1 + n_enh);
    }
    if (std::getenv("PC_TIMING")) std::fprintf(stderr, "[pcodec] decompress: %d level(s), host rANS decode (summed over lanes) %.2f ms\n", n_levels, c->t_host_decode_ms);
    return PC_OK;
}

}  // namespace

extern "C" int pc_codec_decompress(pc_codec* c, const uint8_t* const* y_strings, const size_t* y_lens, int n_slices,
                                   const uint8_t* const* z_strings, const size_t* z_lens, int B, int zh, int zw,
                                   double quality, int mask_pol, float* x_hat, void* stream)
{
    if (n_slices < (quality == 0 ? NS0 : 2 * NS0)) return PC_ERR_ARG;
    return decompress_impl(c, y_strings, y_lens, z_strings, z_lens, B, zh, zw, &quality, 1, mask_pol, x_hat, (hipStream_t)stream);
}

// decompress_levels from ONE buffer: data = the y strings in slot order ([(10 + 10*n_levels)][B], empty for quality-0 levels) followed
// by the B z strings, lens = their lengths in that order (the layout pc_codec_copy_strings produces)
extern "C" int pc_codec_decompress_packed(pc_codec* c, const uint8_t* data, const size_t* lens, int B, int zh, int zw,
                                          const double* qualities, int n_levels, int mask_pol, float* x_hat, void* stream)
{
    if (!c || !data || !lens || B <= 0 || n_levels < 1) return PC_ERR_ARG;
    const size_t ny = (size_t)(NS0 + NS0 * n_levels) * B;
    std::vector<const uint8_t*> ptr(ny + B);
    size_t off = 0;
    for (size_t i = 0; i < ny + B; ++i) { ptr[i] = data + off; off += lens[i]; }
    return decompress_impl(c, ptr.data(), lens, ptr.data() + ny, lens + ny, B, zh, zw, qualities, n_levels, mask_pol, x_hat, (hipStream_t)stream);
}

extern "C" int pc_codec_decompress_levels(pc_codec* c, const uint8_t* const* y_strings, const size_t* y_lens,
                                          const uint8_t* const* z_strings, const size_t* z_lens, int B, int zh, int zw,
                                          const double* qualities, int n_levels, int mask_pol, float* x_hat, void* stream)
{
    return decompress_impl(c, y_strings, y_lens, z_strings, z_lens, B, zh, zw, qualities, n_levels, mask_pol, x_hat, (hipStream_t)stream);
}

extern "C" int pc_codec_read_tap(pc_codec* c, const char* name, float* host_out, size_t cap, size_t* n)
{
    if (!c || !name || !n) return PC_ERR_ARG;
    auto it = c->bufs.find(name);
    if (it == c->bufs.end()) return PC_ERR_ARG;
    const size_t cnt = it->second.bytes / 4;
    *n = cnt;
    if (!host_out) return PC_OK;
    const size_t m = std::min(cnt, cap);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(host_out, it->second.p, m * 4, hipMemcpyDeviceToHost));
    return PC_OK;
}

extern "C" int pc_codec_read_tap_i32(pc_codec* c, const char* name, int32_t* host_out, size_t cap, size_t* n)
{
    return pc_codec_read_tap(c, name, reinterpret_cast<float*>(host_out), cap, n);
}

extern "C" int pc_codec_profile_begin(pc_codec* c)
{
    if (!c) return PC_ERR_ARG;
    c->profile = true; c->ev_used = 0; c->prof_flops = 0.0; c->prof_bytes = 0.0; c->prof_rec.clear();
    return PC_OK;
}

extern "C" int pc_codec_profile_bytes(const pc_codec* c, double* total_algorithmic_bytes)
{
    if (!c || !total_algorithmic_bytes) return PC_ERR_ARG;
    *total_algorithmic_bytes = c->prof_bytes;                 // of the launches recorded since the last pc_codec_profile_begin
    return PC_OK;
}

extern "C" int pc_codec_host_stats(const pc_codec* c, double* out, int n)
{
    if (!c || !out || n < 6) return PC_ERR_ARG;
    out[0] = c->t_compress_ms; out[1] = c->t_host_encode_ms; out[2] = c->t_host_encode_exposed_ms;
    out[3] = c->t_host_decode_ms; out[4] = c->n_sym_encoded; out[5] = c->n_sym_decoded;
    return PC_OK;
}

// One epoch event per device: what the in-schedule profile of several codec objects (an encoder and a decoder side by side) measures its
// launch intervals against, so that they can be merged into one timeline.
static std::mutex g_epoch_mu;
static hipEvent_t g_epoch[16] = {};
extern "C" int pc_profile_set_epoch(int device)
{
    if (device < 0 || device >= 16) return PC_ERR_ARG;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_epoch_mu);
    if (!g_epoch[device]) HIPCHK(hipEventCreate(&g_epoch[device]));
    HIPCHK(hipEventRecord(g_epoch[device], nullptr));
    HIPCHK(hipEventSynchronize(g_epoch[device]));
    return PC_OK;
}

extern "C" int pc_codec_profile_end(pc_codec* c, int64_t* n_launches, double* total_ms, double* total_flops)
{
    if (!c || !n_launches || !total_ms || !total_flops) return PC_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    hipEvent_t epoch = nullptr;
    { std::lock_guard<std::mutex> lk(g_epoch_mu); if (c->device < 16) epoch = g_epoch[c->device]; }
    double ms = 0.0;
    for (size_t i = 0; i + 1 < c->ev_used; i += 2) {
        float t = 0.0f;
        HIPCHK(hipEventElapsedTime(&t, c->ev[i], c->ev[i + 1]));
        ms += t;
        if (i / 2 < c->prof_rec.size()) {
            float s0 = 0.0f;
            if (epoch && hipEventElapsedTime(&s0, epoch, c->ev[i]) != hipSuccess) { (void)hipGetLastError(); s0 = 0.0f; }
            c->prof_rec[i / 2].t0_ms = s0;
            c->prof_rec[i / 2].t1_ms = (double)s0 + t;
        }
    }
    if (const char* path = c->profile_in_schedule ? nullptr : std::getenv("PC_PROFILE_CSV")) {      // per-launch shapes and times of the SERIAL form (in the schedule a launch's duration is not its own)
        if (FILE* f = std::fopen(path, "w")) {
            std::fprintf(f, "i,M,N,K,nphase,epi,gflop,us,tflops,alg_mbytes\n");
            for (size_t i = 0; i < c->prof_rec.size(); ++i) {
                const auto& r = c->prof_rec[i];
                const double t = r.t1_ms - r.t0_ms;
                std::fprintf(f, "%zu,%d,%d,%d,%d,%d,%.3f,%.1f,%.2f,%.3f\n", i, r.M, r.N, r.K, r.nphase, r.epi, r.flops / 1e9, t * 1e3, r.flops / (t * 1e-3) / 1e12, r.bytes / 1e6);
            }
            std::fclose(f);
        }
    }
    *n_launches = (int64_t)(c->ev_used / 2); *total_ms = ms; *total_flops = c->prof_flops;
    c->profile = false; c->ev_used = 0; g_prof = nullptr;
    return PC_OK;
}

// the launch intervals of the profile that pc_codec_profile_end closed: start / end in ms against the device's epoch, FLOPs per launch
extern "C" int pc_codec_profile_intervals(const pc_codec* c, double* t0_ms, double* t1_ms, double* flops, size_t cap, size_t* n)
{
    if (!c || !n) return PC_ERR_ARG;
    *n = c->prof_rec.size();
    if (!t0_ms && !t1_ms && !flops) return PC_OK;                 // size query
    if (!t0_ms || !t1_ms || !flops || cap < c->prof_rec.size()) return PC_ERR_BUFFER;
    for (size_t i = 0; i < c->prof_rec.size(); ++i) { t0_ms[i] = c->prof_rec[i].t0_ms; t1_ms[i] = c->prof_rec[i].t1_ms; flops[i] = c->prof_rec[i].flops; }
    return PC_OK;
}
