// C ABI of libairpose_hip.so: handle management, weight folding/packing, kernel sequencing.
// Declarations and the reference interfaces each entry point replaces: include/airpose_hip.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <map>
#include <string>
#include <vector>

#include "../../include/airpose_hip.h"
#define AP_API_TU   // kernels.h: declare the launchers of BOTH 16-bit storage flavours (k_bf16:: / k_f16::)
#include "ap_common.h"
#include "kernels.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
// launcher of the 16-bit kernel set the precision selects: fp16 storage (AP_PREC_F16) or bf16 storage / fp32 / split-bf16
#define H16(prec, fn) ((prec) == AP_PREC_F16 ? k_f16::fn : k_bf16::fn)
inline bool prec_half(int prec) { return prec == AP_PREC_BF16 || prec == AP_PREC_F16; }   // the throughput kernels
inline int prec_kind(int prec) { return prec == AP_PREC_F16 ? K_BF16 : prec; }             // storage kind inside a kernel set
inline bool prec_valid(int prec) { return prec == AP_PREC_FP32 || prec == AP_PREC_BF16 || prec == AP_PREC_BF16X2 || prec == AP_PREC_F16; }
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail((int)_e, std::string(#expr) + ": " + hipGetErrorString(_e));               \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t reserve(size_t n) {
        if (n <= bytes) return hipSuccess;
        if (p) {
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) return e;
            (void)hipFree(p);
            p = nullptr;
            bytes = 0;
        }
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <typename T> T* as() const { return (T*)p; }
};

hipError_t upload(DevBuf& b, const void* src, size_t n) {
    hipError_t e = b.reserve(n);
    if (e != hipSuccess) return e;
    return hipMemcpy(b.p, src, n, hipMemcpyHostToDevice);
}

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
    size_t numel() const { return data.size(); }
};

struct Layer {                  // a conv or linear layer in packed device form
    int cin = 0, cout = 0, k = 1, stride = 1, pad = 0;
    int wld = 0, cout_pad = 0;
    int cin2 = 0, stride2 = 1;      // second K segment (downsample folded into conv3)
    DevBuf w, scale, shift;
    DevBuf pw;                      // pointwise layers of layer3 / layer4: the weights as conv_pw.hip's fragment streams
};

struct Timing {
    int on = 0;                     // 0 off, 1 every stage, 2 conv stack only (two events per trunk pass)
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    std::vector<size_t> marks[4];   // pairs of event indices per stage
    std::vector<char> qfree[4];     // per quad: the two passes ran free of each other (own durations instead of the span)
    std::vector<size_t> quads[4];   // two-stream passes: (start A, end A, start B, end B): the stage's span over both streams
    int64_t passes = 0;
    hipError_t rec(hipStream_t st, size_t* idx) {
        if (used == pool.size()) {
            hipEvent_t e;
            hipError_t r = hipEventCreate(&e);
            if (r != hipSuccess) return r;
            pool.push_back(e);
        }
        *idx = used++;
        return hipEventRecord(pool[*idx], st);
    }
    hipError_t collect(double ms[4], int nstage, int64_t* n, bool reset) {
        for (int s = 0; s < nstage; ++s) {
            ms[s] = 0.0;
            for (size_t i = 0; i + 1 < marks[s].size(); i += 2) {
                hipError_t r = hipEventSynchronize(pool[marks[s][i + 1]]);
                if (r != hipSuccess) return r;
                float t = 0.f;
                r = hipEventElapsedTime(&t, pool[marks[s][i]], pool[marks[s][i + 1]]);
                if (r != hipSuccess) return r;
                ms[s] += t;
            }
            for (size_t i = 0; i + 3 < quads[s].size(); i += 4) {
                float span = 0.f;
                if (i / 4 < qfree[s].size() && qfree[s][i / 4]) {
                    // free-running passes (ap_trunk_fwd_twoview_async): the two streams drift apart by up to a step, so the span from
                    // the first start to the last end also counts time in which one of the two was already / still in a
                    // neighbouring step; every pass shares the chip with exactly one other pass for its whole duration, so the
                    // stage's time is the mean of the two passes' OWN durations (equal to the span when they run in lock step)
                    float own = 0.f;
                    for (int a = 0; a < 2; ++a) {
                        hipError_t r = hipEventSynchronize(pool[quads[s][i + 1 + 2 * a]]);
                        if (r != hipSuccess) return r;
                        float t = 0.f;
                        r = hipEventElapsedTime(&t, pool[quads[s][i + 2 * a]], pool[quads[s][i + 1 + 2 * a]]);
                        if (r != hipSuccess) return r;
                        own += 0.5f * t;
                    }
                    ms[s] += own;
                    continue;
                }
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 2; ++b) {            // latest end minus earliest start (negative pairs lose)
                        hipError_t r = hipEventSynchronize(pool[quads[s][i + 1 + 2 * b]]);
                        if (r != hipSuccess) return r;
                        float t = 0.f;
                        r = hipEventElapsedTime(&t, pool[quads[s][i + 2 * a]], pool[quads[s][i + 1 + 2 * b]]);
                        if (r != hipSuccess) return r;
                        span = std::max(span, t);
                    }
                ms[s] += span;
            }
        }
        *n = passes;
        if (reset) {
            for (auto& m : marks) m.clear();
            for (auto& q : quads) q.clear();
            for (auto& q : qfree) q.clear();
            used = 0;
            passes = 0;
        }
        return hipSuccess;
    }
    void destroy() {
        for (auto e : pool) (void)hipEventDestroy(e);
        pool.clear();
    }
};

constexpr double BN_EPS = 1e-5;

unsigned long long* g_conv_dbg = nullptr;   // phase-stamp buffer (ap_debug_set_trace)
// Tuning knob of ap_set_conv_config, process-wide: ONE atomic word holding the raw value (-1 automatic, -4 automatic
// without the slab / lean kernels, -5 automatic without the lean kernel, 0..14 / 17 / 100 one explicit configuration);
// dispatch_conv reads it once per launch and decodes it, so handles on different threads never see a torn setting.
std::atomic<int> g_conv_mode{-1};
void* g_zero[16] = {nullptr};   // per-device 256-byte zero line

hipError_t zero_line(const void** out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    if (!g_zero[dev]) {
        e = hipMalloc(&g_zero[dev], 256);
        if (e != hipSuccess) return e;
        e = hipMemset(g_zero[dev], 0, 256);
        if (e != hipSuccess) return e;
    }
    *out = g_zero[dev];
    return hipSuccess;
}

hipError_t device_cus(int* n) {
    static int cus[AP_MAX_DEVICES] = {};
    int dev = 0;
    hipError_t e = ap_current_device(&dev);
    if (e != hipSuccess) return e;
    if (!cus[dev]) {
        e = hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
    }
    *n = cus[dev];
    return hipSuccess;
}

// choose the tile configuration: large tiles need enough tiles to fill 256 CUs (1 workgroup per CU)
hipError_t dispatch_conv(ConvArgs& a, int prec /* AP_PREC_* */, hipStream_t st) {
    const int is_bf16 = prec_kind(prec);                     // storage kind inside the kernel set: K_F32 / K_BF16 (16-bit) / K_SPLIT
    const int mode = g_conv_mode.load(std::memory_order_relaxed);
    const bool use_slab = mode == -1 || mode == -5, use_lean = mode == -1;
    int cfg = mode < 0 ? -1 : mode;
    if (cfg < 0) {
        // Measured on MI355X at 512 images (tools/conv_bench.py, profiles/r01_d_conv_configs.txt): the 2-stage
        // LDS-DMA ring with 8 waves per 128-row tile wins on every trunk layer -- two workgroups (16 waves) per CU
        // run out of phase, so one tile's HBM-bound prologue/epilogue overlaps the other's MFMA loop:
        //   11  128x128 tile, waves 2(M) x 4(N)   C_out >= 128
        //   12  128x64  tile, waves 4(M) x 2(N)   C_out <= 64 (layer1 conv1/conv2)
        //   100 register-staged kernel, 64x64 tiles: problems too small to fill the chip with 128-row tiles
        //   grids below one workgroup per CU with 128x128 tiles (e.g. layer4 conv1/conv2 at 128 images: 196 tiles) take
        //   the 128x64 tiles (twice the workgroups; 46-48 us against 72-83 us for the register-staged kernel there)
        const long mt128 = (a.M + 127) / 128, nt128 = (a.Cout + 127) / 128, nt64 = (a.Cout + 63) / 64;
        //   (a folded downsample = second K segment needs the 128-wide tiles or the register-staged kernel)
        //   14  the same 128x128 tile with the nine taps of a stride-1 3x3 read from one LDS slab per channel chunk: every
        //       shape it can run, at EVERY size (it is the one kernel whose fp32 summation order differs from the others',
        //       so a size-dependent choice would make a pair's result depend on the batch it arrives in; measured equal or
        //       faster than the small-problem configurations from 2 to 512 images)
        //   17  pointwise layers with a short contraction and several channel tiles (conv3 of layer2-4, layer3.0 conv1: K <= 512,
        //       C_out >= 256)
        //       on three lean workgroups per CU (conv_lean.hip, bit-identical to 11): -4..6 % there, +20 % on K >= 1024
        if (use_slab && k_bf16::ap_conv_slab_supported(a, is_bf16)) cfg = 14;
        else if (use_lean && a.Cin <= 512 && a.Cout >= 256 && mt128 * nt128 >= 768 && k_bf16::ap_conv_lean_supported(a, is_bf16)) cfg = 17;
        else if (mt128 * nt128 >= 256) cfg = a.Cout <= 64 ? 12 : 11;
        else if (mt128 * nt64 >= 128) cfg = a.x2 ? (mt128 * nt128 >= 64 ? 11 : 100) : 12;
        else cfg = 100;
    }
    // the fragment-tiled output exists in the LDS-staged 16-bit epilogues only (trunk_chunk asks for it in the automatic modes)
    if (a.y_tiled && (is_bf16 != 1 || cfg == 17 || (cfg >= 4 && cfg <= 7) || (a.Cout & 7))) return hipErrorInvalidValue;
    if (cfg == 100) return H16(prec, ap_launch_conv)(a, is_bf16, st);
    hipError_t e = zero_line(&a.zero);
    if (e != hipSuccess) return e;
    a.dbg = g_conv_dbg;
    if (cfg == 17) {                                         // lean pointwise kernel; other shapes: the ring kernel's tile
        if (!k_bf16::ap_conv_lean_supported(a, is_bf16)) return H16(prec, ap_launch_conv_pipe)(a, is_bf16, 11, st);
        return H16(prec, ap_launch_conv_lean)(a, st);
    }
    if (cfg == 14) {
        // explicit 14 on a shape the slab kernel cannot run: the ring kernel's tile of the same shape
        if (!k_bf16::ap_conv_slab_supported(a, is_bf16)) return H16(prec, ap_launch_conv_pipe)(a, is_bf16, 11, st);
        return H16(prec, ap_launch_conv_slab)(a, st);
    }
    return H16(prec, ap_launch_conv_pipe)(a, is_bf16, cfg, st);
}
constexpr int ST = 148, SLD = 288, DLD = 148;

}  // namespace

struct ap_net {
    int device = 0, prec = AP_PREC_BF16, variant = 0;
    bool finalized = false;
    std::map<std::string, HostTensor> tensors;
    const ap_net* tensors_of = nullptr;                      // the fp32 reference handle of ap_net_parity_probe packs its owner's host tensors
    ap_net* probe_ref = nullptr;                             // ... that handle (trunk only), valid for the current packing
    DevBuf probe_x, probe_bb, probe_pos, probe_feat, probe_out;
    // trunk
    DevBuf stem_w, stem_wpk, stem_wpk_lo, stem_scale, stem_shift;   // stem_wpk_lo: low plane of the split-bf16 stem weights
    struct Block { Layer c1, c2, c3, down, c3ds; bool has_down = false;
                   DevBuf pair; int pair_p = 0, pair_p2 = 0, pair_c3 = 0, pair_n1 = 0;
                   DevBuf imgw;
                   DevBuf c2img, c2s2; };            // layer2 identity blocks: conv2's weights as the fragment streams of conv_img3.hip             // layer3 identity blocks: the three weight matrices as the fragment streams of block_img.hip   // pair: conv3 of this block + conv1 of the next as one weight stream (conv_pair.hip)
    std::vector<Block> blocks;
    // regressor (fp32)
    Layer fc1_feat, fc1_state, fc2, dec;
    Layer fold_feat, fold_state;   // dec o fc2 o fc1 folded into one 145 x 2332 map (no activation between them)
    DevBuf foldT_feat, foldT_state, fold_bias;   // the same map k-major ([k][148]) for the fused IEF kernel (copenet head)
    bool fuse_ief = true;          // folded map: one split-K feature kernel + one kernel for all IEF iterations
    bool fold = true;
    double fold_check_err = 0.0;   // ap_net_finalize: folded vs literal chain on the probe batch (max |diff| / max |literal|)
    bool fold_rejected = false;    // ... above fold_bar: fold forced off for this checkpoint
    double fold_bar = 1e-5;        // (ap_net_set_fold_bar: test aid)
    bool fuse_ds = true;           // first block of a stage: downsample conv folded into conv3 as a second K segment
    bool fuse_block = true;        // 16-bit modes: each layer1 bottleneck as one kernel (bottleneck2.hip); off: separate convs
    bool tiled = true;             // 16-bit modes: tensors only the fused pair kernel reads (t2, identity) in its fragment-tiled layout
    bool fuse_pair = true;         // bf16: conv3 of an identity block + conv1 of the next block as one pixel-local kernel (conv_pair.hip)
    bool fuse_tail = true;         // 16-bit modes: conv1 of layer2.0 inside the kernel of layer1's last block (bottleneck2.hip, tail variant);
                                   // the block output is then stored at the even pixels only (layer2.0's stride-2 downsample reads nothing else)
    int pw_conv = 1;               // 16-bit modes: conv1 of the layer3 / layer4 bottlenecks that no fused kernel covers on the one-wave-per-SIMD pointwise
                                   // kernel (conv_pw.hip): 0 never; 1 (default) when its tiles fill half the chip or whole rounds of it; 2 whenever
                                   // supported, and conv3 + identity too; 3 conv1 whenever supported
    int s2p = 0;                   // 16-bit modes: conv2 of layer2.0 (3x3 / stride 2 at 56 x 56) on the polyphase kernel (conv_s2p.hip), at every batch size
                                   // (its K order is its own).  OFF by default: 12-27 % faster than the ring kernel alone, but it owns its CUs (8 waves x 240
                                   // registers) and the two free-running passes of the default lose more concurrency than the layer gains: -1.7 % in the bench
    int img3 = 1;                  // 16-bit modes: conv2 of the layer2 identity blocks on the half-image-resident kernel (conv_img3.hip): 0 never, 1 when the
                                   // pass fills whole rounds of the chip with half images (same bits either way), 2 always
    int img_block = 1;             // 16-bit modes: each layer3 identity bottleneck as ONE image-resident kernel (block_img.hip): 0 never,
                                   // 1 when the pass fills whole rounds of the chip (an image per CU; same bits either way), 2 always
    bool even_out = true;          // 16-bit modes: a pair block whose output is read by a stride-2 downsample branch ONLY stores the even pixels
    int fuse_stem = 1;             // bf16 / bf16x2: conv1+bn1+relu+maxpool in one kernel (bit-identical to the two-kernel path); 16-bit
                                   // modes: 1 = the persistent form of stem.hip (default), 2 = a workgroup per strip (the round-4 form)
    bool fuse_pool = false;        // 16-bit modes: AvgPool2d(7) in the epilogue of layer4.2 conv3 (conv_lean.hip POOL variant; bit-identical).
                                   // Off by default: measured neutral (fp16) to -0.5 % (bf16) in the two-stream trunk, profiles/r04_fuse_pool_ab.txt
    DevBuf mean_pose, mean_shape, mean_cam;
    // workspace
    int chunk = 0;
    struct TrunkWs { DevBuf ws_stem, ws_a, ws_b, ws_t1, ws_t2, ws_ds;
                     int* rflag = nullptr; };   // the range word the kernels of THIS pass stream set (AP_PREC_F16; ap_net::range_flag + q)
    TrunkWs tw[4];                 // [1..]: the other concurrent passes when the views run on several streams
    DevBuf ws_feat;
    // two-view forward: view 0 and view 1 as two concurrent trunk passes on two internal streams (an HBM-bound layer of
    // one pass overlaps an MFMA-bound layer of the other: -4 % trunk time at 2 x 256 images); 0 = one pass over both views
    bool dual_stream = true;
    int dual_skew = 0;             // experiment: the second pass starts after the first has finished its stem (1) / its block k-2 (k >= 2)
    hipEvent_t ev_skew = nullptr;
    hipStream_t aux[4] = {nullptr, nullptr, nullptr, nullptr};
    bool unjoined = false;         // ap_trunk_fwd_twoview_async: the last two-pass call joined into another stream than its inputs'
    hipEvent_t ev_fork = nullptr, ev_in = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
    int passes_per_view = 1;       // experiment: 2 = each view as two concurrent half passes (four streams)
    DevBuf ws_H, ws_S, ws_T1, ws_T2, ws_D, ws_state;
    Timing tm;
    bool half() const { return prec_half(prec); }            // the throughput kernels (bf16 or fp16 storage)
    int kind() const { return prec_kind(prec); }
    size_t esize() const { return half() ? 2 : 4; }          // fp32 and split-bf16 pairs: 4 bytes
    // AP_PREC_F16 range sentinel: host-mapped word the pooling stage sets when a trunk feature is not finite (NULL otherwise).
    // range_mode 1 (default): sticky, reported by the NEXT call on the handle and by ap_net_range_status (no sync on the hot path);
    // 2: every trunk-running call synchronises its stream and reports its own pass
    // One word per pass stream (tw[q].rflag = range_flag + q): a snapshot taken on pass stream q behind a batch's last kernel there sees
    // exactly the kernels of this and earlier batches, whatever the sibling stream is already running (ap_net_range_mark_next)
    int* range_flag = nullptr;                               // [4]
    int* range_slots = nullptr;                              // [AP_RANGE_SLOTS][4] host-mapped words: stream-ordered snapshots of the pass words
    int conv_launches = 0;                                   // kernel launches of the conv stack in the most recent trunk call, all passes (ap_net_last_conv_launches)
    int mark_slot = -1;                                      // ap_net_range_mark_next: the next trunk-running call snapshots into this slot
    int range_mode = 1;
    bool range_any() const {
        if (!range_flag) return false;
        int v = 0;
        for (int q = 0; q < 4; ++q) v |= __atomic_load_n(range_flag + q, __ATOMIC_RELAXED);
        return v != 0;
    }
    bool f16_overflow = false;                               // AP_PREC_F16: a packed weight left the fp16 range (ap_net_finalize refuses)
    uint16_t h16(float f) { return prec == AP_PREC_F16 ? host_f32_to_f16(f, &f16_overflow) : host_f32_to_bf16(f); }
};

struct ap_smplx {
    int device = 0;
    SmplxModelDev m{};
    Layer dirs;                 // blend-shape GEMM operand: rows = 3V, K = 512 (fp32: exact fp32 MFMA chain)
    DevBuf dirs_split;          // the same operand as split-bf16 pairs: four-term products on the bf16 matrix pipe (default)
    DevBuf dirs_frag, jv_slot, skin_idx8, skin_w4, skin_idx8b, skin_w4b, jt_pack, ws_side;   // fused contraction + skinning: directions in MFMA fragment order, joint-vertex slots / buffer
    DevBuf ws_cnt;              // ... arrival counters of the body groups (joints by the group's last workgroup); zero between launches
    bool fold_post = true;      // ... and with the post transform composed into those 22 transforms by the prep kernel (A22); ap_smplx_set_fused(h, 7): off (A/B)
    int merge_bones = 1;        // (0: off, 1: on, 2: with 64 bodies per workgroup -- A/B, slower) body-only calls: the fused kernel skins over the 22 posed transforms (merged skin table); ap_smplx_set_fused(h, 6): all 55 (A/B)
    bool fuse_joints = false;   // ap_smplx_set_fused(h, 4): joints / landmarks / projection inside the fused kernel (measured 7 us SLOWER than their own launch)
    bool blend_split = true;
    bool fused = true;          // body-only pose feature, 4 bones per vertex, split-bf16 blend: one kernel for contraction + skinning
    DevBuf j_template, j_shapedirs, parents, depth, skin_idx, skin_w, extra_verts, lmk_tri, lmk_bary;
    DevBuf ws_coef, ws_A, ws_A22, ws_jposed, ws_post, ws_vposed, ws_cc;
    int n_out_joints = 0;
    Timing tm;
};

struct ap_fit {                   // AirPose+ fitting loop state (fitting.hip)
    int device = 0;
    const ap_smplx* body = nullptr;
    DevBuf w1t, w2t, w3t, w1, w2, w3, b1, b2, b3;     // VPoser decoder: k-major transposes (forward) / as stored (backward)
    DevBuf H1, H2, O, dO, dH2, dH1, dz, aa, dphi, dtau, dbeta, loss, adam_m, adam_v, robust;
};

namespace {

// ---------------------------------------------------------------------------------- packing
const HostTensor* find(const ap_net* h, const std::string& name) {
    const ap_net* src = h->tensors_of ? h->tensors_of : h;
    auto it = src->tensors.find(name);
    return it == src->tensors.end() ? nullptr : &it->second;
}

int bn_fold(const ap_net* h, const std::string& p, int c, std::vector<float>& scale, std::vector<float>& shift) {
    const HostTensor *g = find(h, p + ".weight"), *b = find(h, p + ".bias"), *m = find(h, p + ".running_mean"),
                     *v = find(h, p + ".running_var");
    if (!g || !b || !m || !v) return fail(AP_ESTATE, "missing BatchNorm tensors for " + p);
    if ((int)g->numel() != c || (int)b->numel() != c || (int)m->numel() != c || (int)v->numel() != c)
        return fail(AP_ESHAPE, "BatchNorm size mismatch for " + p);
    const int cp = ((c + 127) / 128) * 128;
    scale.assign(cp, 1.f);
    shift.assign(cp, 0.f);
    for (int i = 0; i < c; ++i) {
        const double s = (double)g->data[i] / std::sqrt((double)v->data[i] + BN_EPS);
        scale[i] = (float)s;
        shift[i] = (float)((double)b->data[i] - (double)m->data[i] * s);
    }
    return AP_OK;
}

// OIHW fp32 -> [cout_pad][kh][kw][cin] in the handle's storage type
int pack_conv(ap_net* h, const std::string& wname, const std::string& bnname, int cin, int cout, int k, int stride,
              int pad, Layer& L) {
    const HostTensor* w = find(h, wname);
    if (!w) return fail(AP_ESTATE, "missing tensor " + wname);
    if (w->shape.size() != 4 || w->shape[0] != cout || w->shape[1] != cin || w->shape[2] != k || w->shape[3] != k)
        return fail(AP_ESHAPE, "shape mismatch for " + wname);
    L.cin = cin; L.cout = cout; L.k = k; L.stride = stride; L.pad = pad;
    L.wld = k * k * cin;
    L.cout_pad = ((cout + 127) / 128) * 128;
    std::vector<float> scale, shift;
    int rc = bn_fold(h, bnname, cout, scale, shift);
    if (rc) return rc;
    const size_t n = (size_t)L.cout_pad * L.wld;
    if (h->half()) {
        std::vector<uint16_t> pk(n, 0);
        for (int o = 0; o < cout; ++o)
            for (int c = 0; c < cin; ++c)
                for (int r = 0; r < k; ++r)
                    for (int s = 0; s < k; ++s)
                        pk[(size_t)o * L.wld + (r * k + s) * cin + c] =
                            h->h16(w->data[(((size_t)o * cin + c) * k + r) * k + s]);
        HIP_TRY(upload(L.w, pk.data(), n * 2));
    } else {
        std::vector<float> pk(n, 0.f);
        for (int o = 0; o < cout; ++o)
            for (int c = 0; c < cin; ++c)
                for (int r = 0; r < k; ++r)
                    for (int s = 0; s < k; ++s)
                        pk[(size_t)o * L.wld + (r * k + s) * cin + c] = w->data[(((size_t)o * cin + c) * k + r) * k + s];
        if (h->prec == AP_PREC_BF16X2) {                    // rows of k*k*cin elements, cin a multiple of 8: planar groups of 8
            std::vector<uint16_t> ps(2 * n);
            host_split_pack_planar(pk.data(), n, ps.data());
            HIP_TRY(upload(L.w, ps.data(), n * 4));
        } else {
            HIP_TRY(upload(L.w, pk.data(), n * 4));
        }
    }
    HIP_TRY(upload(L.scale, scale.data(), scale.size() * 4));
    HIP_TRY(upload(L.shift, shift.data(), shift.size() * 4));
    return AP_OK;
}

// conv3 (1x1, planes -> cout) and the downsample conv (1x1 stride s, inplanes -> cout) of a stage's first block
// share the output: relu(bn3(conv3(t)) + bn_ds(conv_ds(x))).  Fold each BN scale into its weights (fp64) and
// concatenate along K: one GEMM over [t | x(strided)] with shift = shift3 + shift_ds, no residual tensor.
int pack_c3_ds(ap_net* h, const std::string& P, int planes, int inplanes, int stride, Layer& L) {
    const HostTensor *w3 = find(h, P + ".conv3.weight"), *wd = find(h, P + ".downsample.0.weight");
    if (!w3 || !wd) return fail(AP_ESTATE, "missing conv3/downsample weights for " + P);
    const int cout = planes * 4, K1 = planes, K2 = inplanes;
    if ((int)w3->numel() != cout * K1 || (int)wd->numel() != cout * K2) return fail(AP_ESHAPE, "shape mismatch in " + P);
    std::vector<float> s3, h3, sd, hd;
    int rc = bn_fold(h, P + ".bn3", cout, s3, h3);
    if (rc) return rc;
    if ((rc = bn_fold(h, P + ".downsample.1", cout, sd, hd))) return rc;
    L.cin = K1; L.cout = cout; L.k = 1; L.stride = 1; L.pad = 0;
    L.cin2 = K2; L.stride2 = stride;
    L.wld = K1 + K2;
    L.cout_pad = ((cout + 127) / 128) * 128;
    const size_t n = (size_t)L.cout_pad * L.wld;
    std::vector<float> pk(n, 0.f), scale(L.cout_pad, 1.f), shift(L.cout_pad, 0.f);
    for (int o = 0; o < cout; ++o) {
        for (int c = 0; c < K1; ++c) pk[(size_t)o * L.wld + c] = (float)((double)w3->data[(size_t)o * K1 + c] * (double)s3[o]);
        for (int c = 0; c < K2; ++c) pk[(size_t)o * L.wld + K1 + c] = (float)((double)wd->data[(size_t)o * K2 + c] * (double)sd[o]);
        shift[o] = h3[o] + hd[o];
    }
    if (h->half()) {
        std::vector<uint16_t> pb(n);
        for (size_t i = 0; i < n; ++i) pb[i] = h->h16(pk[i]);
        HIP_TRY(upload(L.w, pb.data(), n * 2));
    } else if (h->prec == AP_PREC_BF16X2) {
        std::vector<uint16_t> ps(2 * n);
        host_split_pack_planar(pk.data(), n, ps.data());
        HIP_TRY(upload(L.w, ps.data(), n * 4));
    } else {
        HIP_TRY(upload(L.w, pk.data(), n * 4));
    }
    HIP_TRY(upload(L.scale, scale.data(), scale.size() * 4));
    HIP_TRY(upload(L.shift, shift.data(), shift.size() * 4));
    return AP_OK;
}

// fp32 GEMM operand from rows [out][ld_src] taking columns [col0, col0+ncols); K padded to 32
int pack_linear(const float* W, int ld_src, int col0, int ncols, int nout, const float* bias, Layer& L) {
    L.cin = ((ncols + 31) / 32) * 32;
    L.cout = ((nout + 3) / 4) * 4;
    L.k = 1; L.stride = 1; L.pad = 0;
    L.wld = L.cin;
    L.cout_pad = ((nout + 127) / 128) * 128;
    std::vector<float> pk((size_t)L.cout_pad * L.wld, 0.f), scale(L.cout_pad, 1.f), shift(L.cout_pad, 0.f);
    for (int o = 0; o < nout; ++o) {
        memcpy(&pk[(size_t)o * L.wld], W + (size_t)o * ld_src + col0, (size_t)ncols * 4);
        if (bias) shift[o] = bias[o];
    }
    HIP_TRY(upload(L.w, pk.data(), pk.size() * 4));
    HIP_TRY(upload(L.scale, scale.data(), scale.size() * 4));
    HIP_TRY(upload(L.shift, shift.data(), shift.size() * 4));
    return AP_OK;
}

// pw: 0 = the generic kernels; 1 / 2 / 3 = conv_pw.hip where the layer has a stream and the shape fits (1: only when its tiles fill
// half the chip, or whole rounds of it to 80 %) -- same bits either way, so the choice may depend on the problem size
// the size rule of conv_pw.hip's automatic choice: its 196-pixel x 256-channel tiles fill half a round of the chip at least, or whole rounds to 80 %
bool pw_fills(long M, int cout, int pw, int* err) {
    int cus = 0;
    if ((*err = device_cus(&cus)) != hipSuccess) return false;
    const int NN = cout >> 8, gmax = k_bf16::ap_conv_pw_grid(1L << 40, cout, cus);
    const long T = ((M / 196 + 7) & ~7L) * NN, rounds = (T + gmax - 1) / gmax;
    return pw == 2 || pw == 3 || (T <= gmax ? T * 2 >= gmax : T * 5 >= rounds * gmax * 4);
}
// the 3 x 3 / stride-2 convolution of a stage's first block (model_copenet.py:32-34, :18) as nine pointwise taps of conv_pw.hip?
bool pw_k3_args(const Layer& L, int N, int H, int W, int prec, PwArgs* p) {
    if (!L.pw.p || !prec_half(prec) || L.k != 3 || L.stride != 2 || L.pad != 1 || (H & 1) || (W & 1)) return false;
    *p = PwArgs{};
    p->k3 = 1; p->Ho = H / 2; p->Wo = W / 2; p->H2 = H; p->W2 = W; p->stride2 = 2;
    p->M = N * p->Ho * p->Wo; p->Cin = L.cin; p->Cout = L.cout; p->relu = 1;
    p->wfrag = L.pw.p; p->scale = L.scale.as<float>(); p->shift = L.shift.as<float>();
    return k_bf16::ap_conv_pw_k3_supported(*p);
}

int run_conv(const Layer& L, const void* x, int N, int H, int W, void* y, const void* res, int relu, int prec,
             hipStream_t st, int* rflag = nullptr, int y_tiled = 0, int pw = 0) {
    if (pw && L.pw.p && prec_half(prec) && relu && !y_tiled && L.k == 1 && L.stride == 1 &&
        g_conv_mode.load(std::memory_order_relaxed) == -1 && k_bf16::ap_conv_pw_supported((long)N * H * W, L.cin, L.cout)) {
        const long M = (long)N * H * W;
        int err = 0;
        const bool fills = pw_fills(M, L.cout, pw, &err);
        HIP_TRY((hipError_t)err);
        if (fills) {
            PwArgs p{};
            p.x = x; p.y = y; p.res = res; p.wfrag = L.pw.p; p.scale = L.scale.as<float>(); p.shift = L.shift.as<float>();
            p.M = (int)M; p.Cin = L.cin; p.Cout = L.cout; p.relu = 1; p.range_flag = rflag;
            HIP_TRY(H16(prec, ap_launch_conv_pw)(p, st));
            return AP_OK;
        }
    }
    PwArgs p3;
    if (pw && relu && !res && !y_tiled && g_conv_mode.load(std::memory_order_relaxed) == -1 && pw_k3_args(L, N, H, W, prec, &p3)) {
        int err = 0;
        const bool fills = pw_fills(p3.M, L.cout, pw, &err);
        HIP_TRY((hipError_t)err);
        if (fills) {
            p3.x = x; p3.y = y; p3.range_flag = rflag;
            HIP_TRY(H16(prec, ap_launch_conv_pw)(p3, st));
            return AP_OK;
        }
    }
    ConvArgs a{};
    a.range_flag = rflag;
    a.y_tiled = y_tiled;
    a.x = x; a.w = L.w.p; a.scale = L.scale.as<float>(); a.shift = L.shift.as<float>(); a.res = res; a.y = y;
    a.N = N; a.H = H; a.W = W; a.Cin = L.cin;
    a.Ho = (H + 2 * L.pad - L.k) / L.stride + 1;
    a.Wo = (W + 2 * L.pad - L.k) / L.stride + 1;
    a.Cout = L.cout;
    a.KH = a.KW = L.k; a.stride = L.stride; a.pad = L.pad;
    a.M = N * a.Ho * a.Wo;
    a.ldx = L.cin; a.ldy = L.cout; a.ldr = L.cout; a.wld = L.wld;
    a.relu = relu;
    HIP_TRY(dispatch_conv(a, prec, st));
    return AP_OK;
}

// fused conv3 + downsample of a stage's first block: t [N][Ho][Ho][cin] (pointwise) and x [N][Hin][Hin][cin2]
// sampled with stride2, concatenated along K
int run_c3_ds(const Layer& L, const void* t, const void* x, int N, int Ho, int Hin, void* y, int prec,
              hipStream_t st, int* rflag = nullptr, int pw = 0) {
    if (pw && L.pw.p && prec_half(prec) && g_conv_mode.load(std::memory_order_relaxed) == -1) {       // conv_pw.hip, as in run_conv
        PwArgs p{};
        p.x = t; p.y = y; p.wfrag = L.pw.p; p.scale = L.scale.as<float>(); p.shift = L.shift.as<float>();
        p.M = N * Ho * Ho; p.Cin = L.cin; p.Cout = L.cout; p.relu = 1; p.range_flag = rflag;
        p.x2 = x; p.Cin2 = L.cin2; p.Ho = p.Wo = Ho; p.H2 = p.W2 = Hin; p.stride2 = L.stride2;
        if (k_bf16::ap_conv_pw_ds_supported(p)) {
            int cus = 0;
            HIP_TRY(device_cus(&cus));
            const int NN = L.cout >> 8, gmax = k_bf16::ap_conv_pw_grid(1L << 40, L.cout, cus);
            const long T = (((long)p.M / 196 + 7) & ~7L) * NN, rounds = (T + gmax - 1) / gmax;
            // (whole rounds only: at half a round -- 64 pairs -- the generic kernel beside the other pass is faster: -0.8 % of that bench)
            if (pw == 2 || pw == 3 || (T >= gmax && T * 5 >= rounds * gmax * 4)) {
                HIP_TRY(H16(prec, ap_launch_conv_pw)(p, st));
                return AP_OK;
            }
        }
    }
    ConvArgs a{};
    a.range_flag = rflag;
    a.x = t; a.w = L.w.p; a.scale = L.scale.as<float>(); a.shift = L.shift.as<float>(); a.res = nullptr; a.y = y;
    a.N = N; a.H = Ho; a.W = Ho; a.Cin = L.cin; a.Ho = Ho; a.Wo = Ho; a.Cout = L.cout;
    a.KH = a.KW = 1; a.stride = 1; a.pad = 0;
    a.M = N * Ho * Ho;
    a.ldx = L.cin; a.ldy = L.cout; a.ldr = L.cout; a.wld = L.wld; a.relu = 1;
    a.x2 = x; a.H2 = Hin; a.W2 = Hin; a.Cin2 = L.cin2; a.stride2 = L.stride2; a.ldx2 = L.cin2;
    HIP_TRY(dispatch_conv(a, prec, st));
    return AP_OK;
}

// fused layer1 bottleneck (bf16): x [N][H][H][c1.cin] -> y [N][H][H][256]
// c1n (or NULL): conv1 of the NEXT block computed on the block output in the same kernel -> t1n [N][H][H][128]; y_even: the block
// output is stored at the even pixels only (its one remaining reader is a stride-2 downsample branch)
int run_bneck64(const Layer& c1, const Layer& c2, const Layer& c3, bool ds, const void* x, int N, int H, void* y,
                int prec, hipStream_t st, int* rflag = nullptr, const Layer* c1n = nullptr, void* t1n = nullptr, int y_even = 0) {
    BneckArgs a{};
    a.range_flag = rflag;
    if (c1n) {
        if (ds || c1n->cin != 256 || c1n->cout != 128 || c1n->k != 1 || c1n->stride != 1 || !t1n)
            return fail(AP_ESHAPE, "fused layer1 bottleneck + next conv1: identity block, 1x1, 256 -> 128");
        a.w1n = c1n->w.p; a.s1n = c1n->scale.as<float>(); a.h1n = c1n->shift.as<float>(); a.t1n = t1n; a.y_even = y_even;
    }
    a.x = x; a.y = y;
    a.w1 = c1.w.p; a.w2 = c2.w.p; a.w3 = c3.w.p;
    a.s1 = c1.scale.as<float>(); a.h1 = c1.shift.as<float>();
    a.s2 = c2.scale.as<float>(); a.h2 = c2.shift.as<float>();
    a.s3 = c3.scale.as<float>(); a.h3 = c3.shift.as<float>();
    a.N = N; a.H = H; a.W = H;
    HIP_TRY(zero_line(&a.zero));
    a.dbg = g_conv_dbg;
    if (!((!ds && c1.cin == 256) || (ds && c1.cin == 64))) return fail(AP_ESHAPE, "fused layer1 bottleneck: C_in 256 (identity) or 64 (first block)");
    HIP_TRY(H16(prec, ap_launch_bneck2)(a, ds ? 1 : 0, st));
    return AP_OK;
}

// y[M][ldy] = x[M][ldx(:K)] * W^T * scale + shift (+ res)
int run_gemm(const Layer& L, const float* x, int ldx, int K, int M, float* y, int ldy, const float* res, int ldr,
             hipStream_t st) {
    ConvArgs a{};
    a.x = x; a.w = L.w.p; a.scale = L.scale.as<float>(); a.shift = L.shift.as<float>(); a.res = res; a.y = y;
    a.N = M; a.H = a.W = a.Ho = a.Wo = 1;
    a.Cin = K; a.Cout = L.cout;
    a.KH = a.KW = 1; a.stride = 1; a.pad = 0;
    a.M = M;
    a.ldx = ldx; a.ldy = ldy; a.ldr = ldr; a.wld = L.wld;
    a.relu = 0;
    HIP_TRY(dispatch_conv(a, 0, st));
    return AP_OK;
}

// every device buffer a trunk block owns (packed rows, BatchNorm vectors, the weight streams of conv_pair / block_img / conv_pw):
// DevBuf has no destructor, so whoever drops a Block releases it first (re-finalize and ap_net_destroy)
void release_layer(Layer& L) { L.w.release(); L.scale.release(); L.shift.release(); L.pw.release(); }
void release_blocks(ap_net* h) {
    for (auto& B : h->blocks) {
        for (Layer* L : {&B.c1, &B.c2, &B.c3, &B.down, &B.c3ds}) release_layer(*L);
        B.pair.release();
        B.imgw.release();
        B.c2img.release();
        B.c2s2.release();
    }
    h->blocks.clear();
}

int finalize_trunk(ap_net* h) {
    // stem: [64][3][7][7] -> [k = (r,s,c)][64] fp32 for the direct kernel
    const HostTensor* w = find(h, "conv1.weight");
    if (!w) return fail(AP_ESTATE, "missing tensor conv1.weight");
    if (w->numel() != 64 * 3 * 49) return fail(AP_ESHAPE, "shape mismatch for conv1.weight");
    std::vector<float> sw(147 * 64);
    for (int o = 0; o < 64; ++o)
        for (int c = 0; c < 3; ++c)
            for (int r = 0; r < 7; ++r)
                for (int s = 0; s < 7; ++s) sw[((r * 7 + s) * 3 + c) * 64 + o] = w->data[((o * 3 + c) * 7 + r) * 7 + s];
    HIP_TRY(upload(h->stem_w, sw.data(), sw.size() * 4));
    {   // MFMA stem operand: [64][AP_STEM_WLD] bf16, k' = r*32 + s*4 + c (zero elsewhere: 4th channel slot, 8th tap, row pad)
        std::vector<uint16_t> pk(64 * AP_STEM_WLD, 0);
        for (int o = 0; o < 64; ++o)
            for (int c = 0; c < 3; ++c)
                for (int r = 0; r < 7; ++r)
                    for (int s2 = 0; s2 < 7; ++s2)
                        pk[o * AP_STEM_WLD + r * 32 + s2 * 4 + c] = h->h16(w->data[((o * 3 + c) * 7 + r) * 7 + s2]);
        HIP_TRY(upload(h->stem_wpk, pk.data(), pk.size() * 2));
        if (h->prec == AP_PREC_BF16X2) {                    // low plane: bf16(w - hi) at the same positions
            std::vector<uint16_t> pl(64 * AP_STEM_WLD, 0);
            for (int o = 0; o < 64; ++o)
                for (int c = 0; c < 3; ++c)
                    for (int r = 0; r < 7; ++r)
                        for (int s2 = 0; s2 < 7; ++s2) {
                            const float wv = w->data[((o * 3 + c) * 7 + r) * 7 + s2];
                            uint16_t hi, lo;
                            host_split_parts(wv, &hi, &lo);
                            pl[o * AP_STEM_WLD + r * 32 + s2 * 4 + c] = lo;
                        }
            HIP_TRY(upload(h->stem_wpk_lo, pl.data(), pl.size() * 2));
        }
    }
    std::vector<float> sc, sh;
    int rc = bn_fold(h, "bn1", 64, sc, sh);
    if (rc) return rc;
    HIP_TRY(upload(h->stem_scale, sc.data(), sc.size() * 4));
    HIP_TRY(upload(h->stem_shift, sh.data(), sh.size() * 4));

    static const int layers[4] = {3, 4, 6, 3}, planes[4] = {64, 128, 256, 512};
    HIP_TRY(hipDeviceSynchronize());                        // (a re-finalize: no pass of the previous packing may still be in flight)
    release_blocks(h);
    h->blocks.resize(16);
    int inpl = 64, bi_all = 0;
    for (int li = 0; li < 4; ++li)
        for (int bi = 0; bi < layers[li]; ++bi, ++bi_all) {
            char p[64];
            snprintf(p, sizeof p, "layer%d.%d", li + 1, bi);
            const std::string P(p);
            const int pl = planes[li], stride = (bi == 0 && li > 0) ? 2 : 1;
            ap_net::Block& B = h->blocks[bi_all];
            if ((rc = pack_conv(h, P + ".conv1.weight", P + ".bn1", inpl, pl, 1, 1, 0, B.c1))) return rc;
            if ((rc = pack_conv(h, P + ".conv2.weight", P + ".bn2", pl, pl, 3, stride, 1, B.c2))) return rc;
            if ((rc = pack_conv(h, P + ".conv3.weight", P + ".bn3", pl, pl * 4, 1, 1, 0, B.c3))) return rc;
            B.has_down = bi == 0;
            if (B.has_down) {
                if ((rc = pack_conv(h, P + ".downsample.0.weight", P + ".downsample.1", inpl, pl * 4, 1, stride, 0,
                                    B.down)))
                    return rc;
                if ((rc = pack_c3_ds(h, P, pl, inpl, stride, B.c3ds))) return rc;
            }
            inpl = pl * 4;
        }
    // conv3 of a block + conv1 of the next block (conv_pair.hip): the two weight matrices as one stream of 16-KiB tiles in
    // the order the fused kernel consumes them, built on the device from the rows packed above.  Identity blocks: conv3 +
    // identity; stage-first blocks: conv3 with the downsample branch folded in as a second K segment (pack_c3_ds), with the
    // next conv1 where the registers allow, alone otherwise
    if (h->half())
        for (size_t b = 0; b + 1 < h->blocks.size(); ++b) {
            ap_net::Block &A = h->blocks[b], &N = h->blocks[b + 1];
            const Layer& L3 = A.has_down ? A.c3ds : A.c3;
            const int P = L3.cin, P2 = A.has_down ? L3.cin2 : 0, C3 = L3.cout;
            int N1 = N.c1.cout;
            if (N.c1.cin != C3) continue;
            if (!k_bf16::ap_conv_pair_supported(P, P2, C3, N1)) N1 = 0;
            if (!k_bf16::ap_conv_pair_supported(P, P2, C3, N1)) continue;
            HIP_TRY(A.pair.reserve(k_bf16::ap_conv_pair_stream_bytes(P, P2, C3, N1)));
            HIP_TRY(H16(h->prec, ap_launch_pair_pack)(L3.w.p, N1 ? N.c1.w.p : nullptr, A.pair.p, P, P2, C3, N1, nullptr));
            A.pair_p = P; A.pair_p2 = P2; A.pair_c3 = C3; A.pair_n1 = N1;
        }
    // layer3 identity blocks (1024 -> 256 -> 256 -> 1024 at 14 x 14): weight streams of the image-resident kernel
    if (h->half())
        for (auto& B : h->blocks) {
            if (B.has_down || B.c1.cin != 1024 || B.c1.cout != 256 || B.c2.cout != 256 || B.c2.stride != 1 || B.c3.cout != 1024) continue;
            HIP_TRY(B.imgw.reserve(k_bf16::ap_block_img_stream_bytes()));
            HIP_TRY(H16(h->prec, ap_launch_block_img_pack)(B.c1.w.p, B.c2.w.p, B.c3.w.p, B.imgw.p, nullptr));
        }
    // stride-1 3 x 3 of the 28 x 28 stage (layer2.1 - 2.3 conv2): weight streams of the half-image-resident kernel
    if (h->half())
        for (auto& B : h->blocks) {
            const Layer& L = B.c2;
            if (!k_bf16::ap_conv_img3_supported(28, 28, L.cin, L.cout, L.k, L.stride, L.pad) || L.wld != 9 * L.cin) continue;
            HIP_TRY(B.c2img.reserve(k_bf16::ap_conv_img3_stream_bytes()));
            HIP_TRY(H16(h->prec, ap_launch_conv_img3_pack)(L.w.p, B.c2img.p, nullptr));
        }
    // stride-2 3 x 3 of layer2.0 (56 x 56 -> 28 x 28, 128 channels): weight streams of the polyphase kernel
    if (h->half())
        for (auto& B : h->blocks) {
            const Layer& L = B.c2;
            if (!k_bf16::ap_conv_s2p_supported(56, 56, L.cin, L.cout, L.k, L.stride, L.pad) || L.wld != 9 * L.cin) continue;
            HIP_TRY(B.c2s2.reserve(k_bf16::ap_conv_s2p_stream_bytes()));
            HIP_TRY(H16(h->prec, ap_launch_conv_s2p_pack)(L.w.p, B.c2s2.p, nullptr));
        }
    // pointwise layers of the 14 x 14 and 7 x 7 stages: weight streams of conv_pw.hip (conv1, and conv3 of the identity blocks)
    if (h->half())
        for (auto& B : h->blocks)
            for (Layer* L : {&B.c1, &B.c3}) {
                if (L->k != 1 || L->stride != 1 || L->cin2 || L->cin < 256 || L->cin % 128 || L->cout % 256 || (L == &B.c3 && B.has_down)) continue;
                if (L->cin * L->cout < 1024 * 256) continue;             // (layer1 / layer2: HBM-bound, and covered by the fused kernels)
                HIP_TRY(L->pw.reserve(k_bf16::ap_conv_pw_stream_bytes(L->cin, L->cout)));
                HIP_TRY(H16(h->prec, ap_launch_conv_pw_pack)(L->w.p, L->pw.p, L->cin, L->cout, L->wld, nullptr));
            }
    // ... conv2 of layer3.0 / layer4.0 (3 x 3, stride 2: K = [tap][Cin], consumed as nine pointwise taps)
    if (h->half())
        for (auto& B : h->blocks) {
            Layer& L = B.c2;
            const int cc = L.cin >> 6;
            if (L.k != 3 || L.stride != 2 || L.pad != 1 || L.cin % 64 || cc < 2 || (cc & (cc - 1)) || L.cout % 256) continue;
            HIP_TRY(L.pw.reserve(k_bf16::ap_conv_pw_stream_bytes(9 * L.cin, L.cout)));
            HIP_TRY(H16(h->prec, ap_launch_conv_pw_pack)(L.w.p, L.pw.p, 9 * L.cin, L.cout, L.wld, nullptr));
        }
    // ... and conv3 + folded downsample of layer4.0 (K = [t2: 512 | x sampled with stride 2: 1024]; layer3.0's rides in a pair kernel)
    if (h->half() && h->fuse_ds)
        for (auto& B : h->blocks) {
            Layer& L = B.c3ds;
            if (!B.has_down || B.pair_p || !L.w.p || L.cin % 64 || L.cin2 % 64 || (L.cin + L.cin2) % 128 || L.cout % 256 || L.cin + L.cin2 < 1024) continue;
            HIP_TRY(L.pw.reserve(k_bf16::ap_conv_pw_stream_bytes(L.cin + L.cin2, L.cout)));
            HIP_TRY(H16(h->prec, ap_launch_conv_pw_pack)(L.w.p, L.pw.p, L.cin + L.cin2, L.cout, L.wld, nullptr));
        }
    HIP_TRY(hipDeviceSynchronize());
    return AP_OK;
}

int finalize_regressor(ap_net* h) {
    const HostTensor *w1 = find(h, "fc1.weight"), *b1 = find(h, "fc1.bias"), *w2 = find(h, "fc2.weight"),
                     *b2 = find(h, "fc2.bias"), *wp = find(h, "decpose.weight"), *bp = find(h, "decpose.bias"),
                     *wsh = find(h, "decshape.weight"), *bsh = find(h, "decshape.bias"), *ip = find(h, "init_pose"),
                     *is = find(h, "init_shape");
    if (!w1 || !b1 || !w2 || !b2 || !wp || !bp || !wsh || !bsh || !ip || !is)
        return fail(AP_ESTATE, "missing regressor tensors (fc1/fc2/decpose/decshape/init_pose/init_shape)");
    if (h->variant == 1) {
        // single-view HMR head (model_hmr.py:160-172): xc = [xf | pose132 | shape10 | cam3] -> fc1 -> fc2 ->
        // decpose/decshape/deccam, all affine: folded like the copenet head into Wf (145 x 2193), bf
        const HostTensor *wc = find(h, "deccam.weight"), *bc = find(h, "deccam.bias"), *ic = find(h, "init_cam");
        if (!wc || !bc || !ic) return fail(AP_ESTATE, "missing deccam / init_cam tensors");
        if (w1->numel() != (size_t)1024 * 2193 || w2->numel() != (size_t)1024 * 1024 || wp->numel() != (size_t)132 * 1024 ||
            wsh->numel() != (size_t)10 * 1024 || wc->numel() != (size_t)3 * 1024 || ip->numel() < 132 || is->numel() != 10)
            return fail(AP_ESHAPE, "hmr regressor tensor shape mismatch");
        std::vector<float> wd((size_t)145 * 1024), bd(145);
        memcpy(wd.data(), wp->data.data(), (size_t)132 * 1024 * 4);
        memcpy(wd.data() + (size_t)132 * 1024, wsh->data.data(), (size_t)10 * 1024 * 4);
        memcpy(wd.data() + (size_t)142 * 1024, wc->data.data(), (size_t)3 * 1024 * 4);
        memcpy(bd.data(), bp->data.data(), 132 * 4);
        memcpy(bd.data() + 132, bsh->data.data(), 10 * 4);
        memcpy(bd.data() + 142, bc->data.data(), 3 * 4);
        std::vector<double> A((size_t)145 * 1024, 0.0);
        for (int o = 0; o < 145; ++o)
            for (int k = 0; k < 1024; ++k) {
                const double wv = wd[(size_t)o * 1024 + k];
                const float* w2r = &w2->data[(size_t)k * 1024];
                double* ar = &A[(size_t)o * 1024];
                for (int j = 0; j < 1024; ++j) ar[j] += wv * w2r[j];
            }
        std::vector<float> wf((size_t)145 * 2193), bfv(145);
        std::vector<double> row(2193);
        for (int o = 0; o < 145; ++o) {
            std::fill(row.begin(), row.end(), 0.0);
            double bacc = bd[o];
            for (int k = 0; k < 1024; ++k) {
                const double av = A[(size_t)o * 1024 + k];
                const float* w1r = &w1->data[(size_t)k * 2193];
                for (int j = 0; j < 2193; ++j) row[j] += av * w1r[j];
                bacc += av * b1->data[k] + (double)wd[(size_t)o * 1024 + k] * b2->data[k];
            }
            for (int j = 0; j < 2193; ++j) wf[(size_t)o * 2193 + j] = (float)row[j];
            bfv[o] = (float)bacc;
        }
        int rc2;
        if ((rc2 = pack_linear(wf.data(), 2193, 0, 2048, 145, bfv.data(), h->fold_feat))) return rc2;
        if ((rc2 = pack_linear(wf.data(), 2193, 2048, 145, 145, nullptr, h->fold_state))) return rc2;
        std::vector<float> mp(144, 0.f);
        memcpy(mp.data(), ip->data.data(), std::min<size_t>(144, ip->numel()) * 4);
        HIP_TRY(upload(h->mean_pose, mp.data(), 144 * 4));
        HIP_TRY(upload(h->mean_shape, is->data.data(), 10 * 4));
        HIP_TRY(upload(h->mean_cam, ic->data.data(), 3 * 4));
        return AP_OK;
    }
    // copenet_singleview (model_copenet_singleview.py:67,156-168): xc = [xf | bb | pose135 | shape10], i.e. the two-view
    // layout without the partner's 136 columns -> the same code with those fc1 columns zero
    HostTensor w1_padded;
    if (h->variant == 2) {
        if (w1->numel() != (size_t)1024 * 2196) return fail(AP_ESHAPE, "copenet_singleview: fc1.weight must be 1024 x 2196");
        w1_padded.shape = {1024, 2332};
        w1_padded.data.assign((size_t)1024 * 2332, 0.f);
        for (int o = 0; o < 1024; ++o)
            memcpy(&w1_padded.data[(size_t)o * 2332], &w1->data[(size_t)o * 2196], (size_t)2196 * 4);
        w1 = &w1_padded;
    }
    // muhmr (model_muhmr.py:67-72,163-197): xc = [xf | cam3 | orient6 | art126 | shape10 | partner 136], decoders
    // decpose (132) / decshape / deccam.  The weak-perspective camera takes the place of the two-view model's
    // translation: cam goes into the `pos` slot (fc1 columns of bb are zero) and deccam's rows are stacked in front of
    // decpose's, so the state row is [cam3 | pose132 | shape10] and the two-view code runs unchanged
    HostTensor wp_stacked, bp_stacked;
    if (h->variant == 3) {
        const HostTensor *wc = find(h, "deccam.weight"), *bc = find(h, "deccam.bias");
        if (!wc || !bc) return fail(AP_ESTATE, "muhmr: missing deccam tensors");
        if (w1->numel() != (size_t)1024 * 2329 || wp->numel() != (size_t)132 * 1024 || wc->numel() != (size_t)3 * 1024 ||
            bp->numel() != 132 || bc->numel() != 3)
            return fail(AP_ESHAPE, "muhmr: fc1.weight must be 1024 x 2329, decpose 132 x 1024, deccam 3 x 1024");
        w1_padded.shape = {1024, 2332};
        w1_padded.data.assign((size_t)1024 * 2332, 0.f);
        for (int o = 0; o < 1024; ++o) {
            const float* src = &w1->data[(size_t)o * 2329];
            float* dst = &w1_padded.data[(size_t)o * 2332];
            memcpy(dst, src, (size_t)2048 * 4);                       // trunk features
            memcpy(dst + 2051, src + 2048, (size_t)(2329 - 2048) * 4);   // cam -> pos slot, then orient .. partner
        }
        w1 = &w1_padded;
        wp_stacked.shape = {135, 1024};
        wp_stacked.data.resize((size_t)135 * 1024);
        memcpy(wp_stacked.data.data(), wc->data.data(), (size_t)3 * 1024 * 4);
        memcpy(wp_stacked.data.data() + (size_t)3 * 1024, wp->data.data(), (size_t)132 * 1024 * 4);
        bp_stacked.shape = {135};
        bp_stacked.data.resize(135);
        memcpy(bp_stacked.data.data(), bc->data.data(), 3 * 4);
        memcpy(bp_stacked.data.data() + 3, bp->data.data(), 132 * 4);
        wp = &wp_stacked;
        bp = &bp_stacked;
    }
    if (w1->numel() != (size_t)1024 * 2332 || w2->numel() != (size_t)1024 * 1024 || wp->numel() != (size_t)135 * 1024 ||
        wsh->numel() != (size_t)10 * 1024 || ip->numel() < 132 || is->numel() != 10)
        return fail(AP_ESHAPE, "regressor tensor shape mismatch");
    int rc;
    if ((rc = pack_linear(w1->data.data(), 2332, 0, 2048, 1024, b1->data.data(), h->fc1_feat))) return rc;
    if ((rc = pack_linear(w1->data.data(), 2332, 2048, 284, 1024, nullptr, h->fc1_state))) return rc;
    if ((rc = pack_linear(w2->data.data(), 1024, 0, 1024, 1024, b2->data.data(), h->fc2))) return rc;
    std::vector<float> wd((size_t)145 * 1024), bd(145);
    memcpy(wd.data(), wp->data.data(), (size_t)135 * 1024 * 4);
    memcpy(wd.data() + (size_t)135 * 1024, wsh->data.data(), (size_t)10 * 1024 * 4);
    memcpy(bd.data(), bp->data.data(), 135 * 4);
    memcpy(bd.data() + 135, bsh->data.data(), 10 * 4);
    if ((rc = pack_linear(wd.data(), 1024, 0, 1024, 145, bd.data(), h->dec))) return rc;
    {
        // forward_reg is fc1 -> dropout(identity in eval) -> fc2 -> dropout -> decpose/decshape with NO activation
        // (model_copenet.py:186-202), i.e. one affine map.  Fold it once in fp64:
        //   Wf = Wd W2 W1 (145 x 2332),  bf = Wd (W2 b1 + b2) + bd
        std::vector<double> A((size_t)145 * 1024, 0.0);                       // Wd W2
        for (int o = 0; o < 145; ++o)
            for (int k = 0; k < 1024; ++k) {
                const double wv = wd[(size_t)o * 1024 + k];
                const float* w2r = &w2->data[(size_t)k * 1024];
                double* ar = &A[(size_t)o * 1024];
                for (int j = 0; j < 1024; ++j) ar[j] += wv * w2r[j];
            }
        std::vector<float> wf((size_t)145 * 2332), bfv(145);
        std::vector<double> row(2332);
        for (int o = 0; o < 145; ++o) {
            std::fill(row.begin(), row.end(), 0.0);
            double bacc = bd[o];
            for (int k = 0; k < 1024; ++k) {
                const double av = A[(size_t)o * 1024 + k];
                const float* w1r = &w1->data[(size_t)k * 2332];
                for (int j = 0; j < 2332; ++j) row[j] += av * w1r[j];
                bacc += av * b1->data[k] + (double)wd[(size_t)o * 1024 + k] * b2->data[k];
            }
            for (int j = 0; j < 2332; ++j) wf[(size_t)o * 2332 + j] = (float)row[j];
            bfv[o] = (float)bacc;
        }
        if ((rc = pack_linear(wf.data(), 2332, 0, 2048, 145, bfv.data(), h->fold_feat))) return rc;
        if ((rc = pack_linear(wf.data(), 2332, 2048, 284, 145, nullptr, h->fold_state))) return rc;
        {   // Guard of the fold on THIS checkpoint: the fp32-rounded folded map against the literal fc1 -> fc2 -> dec chain, both
            // evaluated in fp64 on a fixed probe batch (post-pooling-like features, states around the mean parameters).  The
            // fold is exact algebra; what can go wrong is cancellation -- folded rows whose fp32 rounding error, summed over the
            // 2332 inputs, is visible in the small decoder outputs.  Above 1e-5 of the output scale the handle evaluates the
            // literal chain instead (ap_net_fold_status reports which and why).
            const int NP = 8;
            uint64_t lcg = 0x9E3779B97F4A7C15ull;
            auto unif = [&]() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (double)(lcg >> 11) * (1.0 / 9007199254740992.0); };
            auto gauss = [&]() { const double u = std::max(unif(), 1e-300), v = unif(); return std::sqrt(-2.0 * std::log(u)) * std::cos(6.283185307179586 * v); };
            double worst = 0.0, scale = 0.0;
            std::vector<double> x(2332), t1(1024), t2(1024);
            std::vector<double> ylit((size_t)NP * 145), yfold((size_t)NP * 145);
            for (int r = 0; r < NP; ++r) {
                for (int j = 0; j < 2048; ++j) x[j] = std::fabs(gauss()) * 0.8;               // pooled post-ReLU features
                for (int j = 2048; j < 2332; ++j) x[j] = 0.3 * gauss();                        // bb / position / 6-D pose / shape
                for (int j = 0; j < 132 && 2054 + j < 2332; ++j) x[2054 + j] += ip->data[j];  // around the mean pose
                for (int o = 0; o < 1024; ++o) {
                    double acc = b1->data[o];
                    const float* wr = &w1->data[(size_t)o * 2332];
                    for (int j = 0; j < 2332; ++j) acc += (double)wr[j] * x[j];
                    t1[o] = acc;
                }
                for (int o = 0; o < 1024; ++o) {
                    double acc = b2->data[o];
                    const float* wr = &w2->data[(size_t)o * 1024];
                    for (int j = 0; j < 1024; ++j) acc += (double)wr[j] * t1[j];
                    t2[o] = acc;
                }
                for (int o = 0; o < 145; ++o) {
                    double acc = bd[o], accf = bfv[o];
                    const float* wr = &wd[(size_t)o * 1024];
                    for (int j = 0; j < 1024; ++j) acc += (double)wr[j] * t2[j];
                    const float* fr = &wf[(size_t)o * 2332];
                    for (int j = 0; j < 2332; ++j) accf += (double)fr[j] * x[j];
                    ylit[(size_t)r * 145 + o] = acc;
                    yfold[(size_t)r * 145 + o] = accf;
                    scale = std::max(scale, std::fabs(acc));
                    worst = std::max(worst, std::fabs(acc - accf));
                }
            }
            h->fold_check_err = scale > 0.0 ? worst / scale : 0.0;
            const bool was_rejected = h->fold_rejected;
            h->fold_rejected = !(h->fold_check_err <= h->fold_bar);
            if (was_rejected && !h->fold_rejected) h->fold = true;      // re-packed weights pass: back to the default
            if (h->fold_rejected) {
                h->fold = false;
                fprintf(stderr, "airpose_hip: the folded regressor map differs from the literal fc1 -> fc2 -> dec chain by %.3e of the "
                                "output scale on the probe batch (bar %.1e): this handle evaluates the literal chain\n", h->fold_check_err, h->fold_bar);
            }
        }
        {   // k-major copies for the fused IEF kernel
            std::vector<float> tf((size_t)2048 * 148, 0.f), ts((size_t)288 * 148, 0.f), tb(148, 0.f);   // (k padded to 288 with zero rows)
            for (int o = 0; o < 145; ++o) {
                for (int k = 0; k < 2048; ++k) tf[(size_t)k * 148 + o] = wf[(size_t)o * 2332 + k];
                for (int k = 0; k < 284; ++k) ts[(size_t)k * 148 + o] = wf[(size_t)o * 2332 + 2048 + k];
                tb[o] = bfv[o];
            }
            HIP_TRY(upload(h->foldT_feat, tf.data(), tf.size() * 4));
            HIP_TRY(upload(h->foldT_state, ts.data(), ts.size() * 4));
            HIP_TRY(upload(h->fold_bias, tb.data(), tb.size() * 4));
        }
    }
    std::vector<float> mp(144, 0.f);
    memcpy(mp.data(), ip->data.data(), std::min<size_t>(144, ip->numel()) * 4);
    HIP_TRY(upload(h->mean_pose, mp.data(), 144 * 4));
    HIP_TRY(upload(h->mean_shape, is->data.data(), 10 * 4));
    if (h->variant == 3) {
        const HostTensor* ic = find(h, "init_cam");
        if (!ic || ic->numel() != 3) return fail(AP_ESTATE, "muhmr: missing init_cam");
        HIP_TRY(upload(h->mean_cam, ic->data.data(), 3 * 4));
    }
    return AP_OK;
}

// workspace of one pass over n images (a grow may synchronise the device and free: never while a sibling pass is in flight)
int reserve_trunk_ws(ap_net* h, ap_net::TrunkWs& w, int n) {
    const bool bf = h->half();
    const size_t es = h->esize();
    if (!(h->fuse_stem && (bf || h->prec == AP_PREC_BF16X2))) HIP_TRY(w.ws_stem.reserve((size_t)n * 112 * 112 * 64 * es));
    HIP_TRY(w.ws_a.reserve((size_t)n * 802816 * es));
    HIP_TRY(w.ws_b.reserve((size_t)n * 802816 * es));
    HIP_TRY(w.ws_ds.reserve((size_t)n * 802816 * es));
    HIP_TRY(w.ws_t1.reserve((size_t)n * 401408 * es));
    HIP_TRY(w.ws_t2.reserve((size_t)n * 200704 * es));
    return AP_OK;
}

// one depth-first pass over n = n0 + n1 images: the first n0 from x0, the rest from x1 (two views, one pass)
int trunk_chunk(ap_net* h, ap_net::TrunkWs& w, const float* x0, int n0, const float* x1, int n1, float* feat, hipStream_t st,
                size_t* ev_out = nullptr, int signal_at = 0) {
    const int bf = h->half();                                // gates the fused kernels of the 16-bit throughput modes
    const int prec = h->prec;                                // selects the kernel set (H16) and, as prec_kind, the storage kind
    const int kind = h->kind();
    const size_t es = h->esize();
    const int n = n0 + n1;
    { int rc0 = reserve_trunk_ws(h, w, n); if (rc0) return rc0; }
    size_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;
    if (h->tm.on == 1) HIP_TRY(h->tm.rec(st, &e0));
    if (bf && h->fuse_stem) {
        HIP_TRY(H16(prec, ap_launch_stem_pool)(x0, x1, n0, h->stem_wpk.p, h->stem_scale.as<float>(), h->stem_shift.as<float>(),
                                               w.ws_a.p, n, w.rflag, h->fuse_stem == 2 ? 1 : 2, st, g_conv_dbg));
    } else if (bf) {
        HIP_TRY(H16(prec, ap_launch_stem_conv_mfma)(x0, x1, n0, h->stem_wpk.p, h->stem_scale.as<float>(),
                                                    h->stem_shift.as<float>(), w.ws_stem.p, n, st));
    } else if (kind == AP_PREC_BF16X2 && h->fuse_stem) {
        HIP_TRY(k_bf16::ap_launch_stem_pool_split(x0, x1, n0, h->stem_wpk.p, h->stem_wpk_lo.p, h->stem_scale.as<float>(),
                                          h->stem_shift.as<float>(), w.ws_a.p, n, st));
    } else if (kind == AP_PREC_BF16X2) {
        HIP_TRY(k_bf16::ap_launch_stem_conv_mfma_split(x0, x1, n0, h->stem_wpk.p, h->stem_wpk_lo.p, h->stem_scale.as<float>(),
                                               h->stem_shift.as<float>(), w.ws_stem.p, n, st));
    } else {
        if (n0)
            HIP_TRY(k_bf16::ap_launch_stem_conv(x0, h->stem_w.as<float>(), h->stem_scale.as<float>(), h->stem_shift.as<float>(),
                                        w.ws_stem.p, n0, kind, st));
        if (n1)
            HIP_TRY(k_bf16::ap_launch_stem_conv(x1, h->stem_w.as<float>(), h->stem_scale.as<float>(), h->stem_shift.as<float>(),
                                        (char*)w.ws_stem.p + (size_t)n0 * 112 * 112 * 64 * es, n1, kind, st));
    }
    if (!(h->fuse_stem && (bf || kind == AP_PREC_BF16X2))) HIP_TRY(H16(prec, ap_launch_maxpool)(w.ws_stem.p, w.ws_a.p, n, kind, w.rflag, st));
    if (h->tm.on) HIP_TRY(h->tm.rec(st, &e1));
    if (signal_at == 1) HIP_TRY(hipEventRecord(h->ev_skew, st));
    void *cur = w.ws_a.p, *nxt = w.ws_b.p;
    int H = 56;
    int rc;
    int blk = 0;
    bool t1_ready = false;                                   // ws_t1 already holds this block's conv1 output (fused pair)
    bool pooled = false;                                     // the last convolution wrote the pooled features itself
    // Fragment-tiled intermediates (ap_common.h: ap_tiled_off): a tensor whose ONLY reader is the fused pair kernel is stored
    // as [M/16][C/8][16 pixels][8 channels], the order the pair kernel's lanes fetch it in -- t2 of every pair block, and a
    // pair block's output when the next block is an identity pair block that also got its conv1 from this kernel.  Same
    // values, same arithmetic: the features are bit-identical with the layout off (ap_net_set_tiled).
    const bool tiling = bf && h->tiled && g_conv_mode.load(std::memory_order_relaxed) < 0;
    auto is_pair = [&](const ap_net::Block& X) {
        return bf && h->fuse_pair && X.pair_p && &X != &h->blocks.back() && (!X.has_down || h->fuse_ds);
    };
    bool cur_tiled = false;                                  // layout of `cur`
    // conv_pw.hip: automatic rule = conv1 only (layer4 at 512 images: 156 / 70 / 71 -> 136 / 60 / 61 us; whole bench +1.0 % single
    // pass, +0.6 % with two concurrent passes).  conv3 + identity (2: forced) is 3-9 us slower than the lean kernel and, beside a
    // concurrent pass, turns the gain into -0.5 %: a one-wave-per-SIMD kernel keeps the other pass's workgroups off its CUs
    const int pw_conv = h->pw_conv;
    for (auto& B : h->blocks) {
        if (signal_at >= 2 && blk++ == signal_at - 2) HIP_TRY(hipEventRecord(h->ev_skew, st));
        const int Ho = (H + 2 - 3) / B.c2.stride + 1;
        if (bf && h->fuse_block && B.c2.cout == 64 && B.c2.stride == 1 && H % 14 == 0 && (!B.has_down || B.c1.cin == 64)) {
            // layer1: conv1 -> conv2 -> conv3 (+identity | folded downsample) in one kernel, intermediates in LDS
            const Layer& L3 = B.has_down ? B.c3ds : B.c3;
            // last block of layer1: conv1 of layer2.0 (model_copenet.py:29-31) on the block output while it is in registers; what
            // is left to read of that output is layer2.0's stride-2 downsample branch (:41-42, :97-102) -> even pixels only
            const ap_net::Block* Nx = &B != &h->blocks.back() ? &B + 1 : nullptr;
            const bool tail = h->fuse_tail && !B.has_down && Nx && Nx->has_down && Nx->c1.cin == 256 && Nx->c1.cout == 128 &&
                              Nx->c2.stride == 2 && Nx->down.stride == 2 && g_conv_mode.load(std::memory_order_relaxed) < 0;
            if ((rc = run_bneck64(B.c1, B.c2, L3, B.has_down, cur, n, H, nxt, prec, st, w.rflag, tail ? &Nx->c1 : nullptr,
                                  w.ws_t1.p, tail && h->even_out)))
                return rc;
            ++h->conv_launches;
            t1_ready = tail;
            std::swap(cur, nxt);
            continue;
        }
        bool img_fit = h->img_block == 2;
        if (h->img_block == 1) {                             // an image per CU: 256 (512) images = one (two) full rounds; 300 = two rounds 59 % full
            int cus = 0;
            HIP_TRY(device_cus(&cus));
            const long rounds = (n + cus - 1) / cus;
            img_fit = (long)n * 8 >= rounds * cus * 7;
        }
        if (bf && img_fit && B.imgw.p && H == 14 && !t1_ready && !cur_tiled && g_conv_mode.load(std::memory_order_relaxed) == -1) {
            // layer3 identity block: conv1 -> conv2 -> conv3 + identity in one kernel, an image per workgroup, t1 / t2 in LDS
            BlkImgArgs a{};
            a.x = cur; a.y = nxt; a.wfrag = B.imgw.p; a.N = n; a.range_flag = w.rflag; a.dbg = g_conv_dbg;
            a.s1 = B.c1.scale.as<float>(); a.h1 = B.c1.shift.as<float>();
            a.s2 = B.c2.scale.as<float>(); a.h2 = B.c2.shift.as<float>();
            a.s3 = B.c3.scale.as<float>(); a.h3 = B.c3.shift.as<float>();
            HIP_TRY(H16(prec, ap_launch_block_img)(a, st));
            ++h->conv_launches;
            std::swap(cur, nxt);
            continue;
        }
        if (!t1_ready && cur_tiled) return fail(AP_ESTATE, "trunk: conv1 of a block would read a tiled block output");
        if (!t1_ready && (rc = run_conv(B.c1, cur, n, H, H, w.ws_t1.p, nullptr, 1, prec, st, w.rflag, 0, pw_conv))) return rc;
        h->conv_launches += (t1_ready ? 0 : 1) + 2 + ((!is_pair(B) && B.has_down && !h->fuse_ds) ? 1 : 0);   // conv1, conv2, conv3 (+ an unfused downsample)
        t1_ready = false;
        const bool pair = is_pair(B);
        // conv2 of a stage's first block on conv_pw.hip (nine taps): it writes NHWC rows, so t2 stays untiled for that block's pair kernel.
        // Automatic rule: only for a pass that has the chip to itself (l4.0.c2 164 -> 149 us, l3.0.c2 168 -> 157: +0.55 % of the whole
        // bench there, -0.45 % beside a concurrent pass, whose workgroups a one-wave-per-SIMD kernel keeps off its CUs; ev_out marks it)
        // conv2 of layer2.0 on the polyphase kernel (conv_s2p.hip; ap_net_set_s2p, off by default): its K order is its own, so when on it
        // takes the layer at EVERY batch size
        const bool c2_s2p = bf && h->s2p && B.c2s2.p && H == 56 && B.c2.stride == 2 && g_conv_mode.load(std::memory_order_relaxed) == -1;
        bool c2_pw = false;
        if (!c2_s2p && pw_conv && pw_conv != 4 && !(pw_conv == 1 && ev_out) && B.c2.stride == 2 && g_conv_mode.load(std::memory_order_relaxed) == -1) {
            PwArgs p3;
            int err = 0;
            c2_pw = pw_k3_args(B.c2, n, H, H, prec, &p3) && pw_fills(p3.M, B.c2.cout, pw_conv, &err);
            HIP_TRY((hipError_t)err);
        }
        const int t2_tiled = pair && tiling && !c2_pw;
        bool c2_img = false;
        if (bf && h->img3 && B.c2img.p && H == 28 && B.c2.stride == 1 && g_conv_mode.load(std::memory_order_relaxed) == -1) {
            c2_img = h->img3 == 2;
            if (h->img3 == 1) {                              // half an image per CU: whole rounds of the chip
                int cus = 0;
                HIP_TRY(device_cus(&cus));
                const long units = 2L * n, rounds = (units + cus - 1) / cus;
                c2_img = units * 8 >= rounds * cus * 7;
            }
        }
        if (c2_s2p) {
            ConvS2pArgs ca{};
            ca.x = w.ws_t1.p; ca.y = w.ws_t2.p; ca.wfrag = B.c2s2.p; ca.scale = B.c2.scale.as<float>(); ca.shift = B.c2.shift.as<float>();
            ca.N = n; ca.y_tiled = t2_tiled; ca.range_flag = w.rflag;
            HIP_TRY(zero_line(&ca.zero));
            HIP_TRY(H16(prec, ap_launch_conv_s2p)(ca, st));
        } else if (c2_img) {
            ConvImg3Args ca{};
            ca.x = w.ws_t1.p; ca.y = w.ws_t2.p; ca.wfrag = B.c2img.p; ca.scale = B.c2.scale.as<float>(); ca.shift = B.c2.shift.as<float>();
            ca.N = n; ca.y_tiled = t2_tiled; ca.range_flag = w.rflag;
            HIP_TRY(zero_line(&ca.zero));
            HIP_TRY(H16(prec, ap_launch_conv_img3)(ca, st));
        } else if ((rc = run_conv(B.c2, w.ws_t1.p, n, H, H, w.ws_t2.p, nullptr, 1, prec, st, w.rflag, t2_tiled, c2_pw ? pw_conv : 0))) return rc;
        if (pair) {
            // conv3 (+ identity | + folded downsample, ReLU) AND -- where the pair carries it -- the next block's conv1 in one
            // kernel: the block output is written once and not read back for conv1 (model_copenet.py:38-45 of this block,
            // :29-31 of the next)
            const ap_net::Block& Nx = *(&B + 1);
            const Layer& L3 = B.has_down ? B.c3ds : B.c3;
            PairArgs a{};
            a.t2 = w.ws_t2.p; a.wstream = B.pair.p;
            a.s3 = L3.scale.as<float>(); a.h3 = L3.shift.as<float>();
            a.s1 = Nx.c1.scale.as<float>(); a.h1 = Nx.c1.shift.as<float>();
            a.out = nxt; a.t1n = w.ws_t1.p; a.M = n * Ho * Ho; a.dbg = g_conv_dbg; a.range_flag = w.rflag;
            a.t2_tiled = t2_tiled; a.res_tiled = cur_tiled;
            a.out_tiled = t2_tiled && B.pair_n1 > 0 && is_pair(Nx) && !Nx.has_down;
            cur_tiled = a.out_tiled != 0;
            a.Ho = a.Wo = Ho;
            if (B.has_down) { a.x2 = cur; a.H2 = a.W2 = H; a.stride2 = L3.stride2; }
            else a.res = cur;
            // the next block is a stage's first one and got its conv1 from this kernel: all that is read of `out` is that block's
            // stride-2 downsample branch (model_copenet.py:41-42, :97-102) -- the even pixels
            a.out_even = h->even_out && !a.out_tiled && B.pair_n1 > 0 && Nx.has_down && Nx.down.stride == 2 && Nx.c2.stride == 2;
            HIP_TRY(H16(prec, ap_launch_conv_pair)(a, B.pair_p, B.pair_p2, B.pair_c3, B.pair_n1, st));
            t1_ready = B.pair_n1 > 0;
        } else if (cur_tiled) {                               // (cannot happen: out_tiled is only set when the next block is a pair block)
            return fail(AP_ESTATE, "trunk: a tiled block output reached a kernel that reads NHWC");
        } else if (B.has_down && h->fuse_ds) {
            if ((rc = run_c3_ds(B.c3ds, w.ws_t2.p, cur, n, Ho, H, nxt, prec, st, w.rflag, pw_conv))) return rc;
        } else if (bf && h->fuse_pool && &B == &h->blocks.back() && !B.has_down && Ho == 7 &&
                   g_conv_mode.load(std::memory_order_relaxed) == -1) {
            // last convolution of the trunk: conv3 + bn3 + identity + ReLU AND AvgPool2d(7) + view in one kernel
            // (model_copenet.py:38-47 of layer4.2, then :173-175); the block output is never written
            ConvArgs a{};
            a.x = w.ws_t2.p; a.w = B.c3.w.p; a.scale = B.c3.scale.as<float>(); a.shift = B.c3.shift.as<float>(); a.res = cur; a.y = nullptr;
            a.N = n; a.H = a.W = a.Ho = a.Wo = Ho; a.Cin = B.c3.cin; a.Cout = B.c3.cout;
            a.KH = a.KW = 1; a.stride = 1; a.pad = 0; a.M = n * Ho * Ho;
            a.ldx = B.c3.cin; a.ldy = B.c3.cout; a.ldr = B.c3.cout; a.wld = B.c3.wld; a.relu = 1;
            a.pool_out = feat; a.range_flag = w.rflag;
            HIP_TRY(zero_line(&a.zero));
            if (k_bf16::ap_conv_lean_supported(a, kind)) {
                HIP_TRY(H16(prec, ap_launch_conv_lean)(a, st));
                pooled = true;
            } else if ((rc = run_conv(B.c3, w.ws_t2.p, n, Ho, Ho, nxt, cur, 1, prec, st, w.rflag))) return rc;
        } else {
            const void* res = cur;
            if (B.has_down) {
                if ((rc = run_conv(B.down, cur, n, H, H, w.ws_ds.p, nullptr, 0, prec, st, w.rflag))) return rc;
                res = w.ws_ds.p;
            }
            if ((rc = run_conv(B.c3, w.ws_t2.p, n, Ho, Ho, nxt, res, 1, prec, st, w.rflag, 0, (B.has_down || pw_conv != 2) ? 0 : pw_conv))) return rc;   // (conv3 + identity: 89-95 us against the lean kernel's 86: only when forced)
        }
        std::swap(cur, nxt);
        H = Ho;
    }
    if (h->tm.on) HIP_TRY(h->tm.rec(st, &e2));
    if (!pooled) HIP_TRY(H16(prec, ap_launch_avgpool)(cur, feat, n, 2048, kind, w.rflag, st));
    if (h->tm.on == 1) HIP_TRY(h->tm.rec(st, &e3));
    if (ev_out) {                                            // the caller combines the events of two concurrent passes
        ev_out[0] = e0; ev_out[1] = e1; ev_out[2] = e2; ev_out[3] = e3;
        return AP_OK;
    }
    if (h->tm.on) { h->tm.marks[1].push_back(e1); h->tm.marks[1].push_back(e2); }
    if (h->tm.on == 1) {
        h->tm.marks[0].push_back(e0); h->tm.marks[0].push_back(e1);
        h->tm.marks[2].push_back(e2); h->tm.marks[2].push_back(e3);
    }
    return AP_OK;
}

// trunk over the concatenation [x0 (n0 images) | x1 (n1 images)]; feat rows follow the same order
int trunk_passes(ap_net* h, const float* x0, int n0, const float* x1, int n1, float* feat, hipStream_t st, hipStream_t st_out);
// st: the stream the inputs are ordered on; st_out (default: st): the stream the features are ordered on.  With two streams
// (ap_trunk_fwd_twoview_async) st is never made to wait for the passes: the next call's passes queue behind this call's on the
// internal streams and the caller's stream stays free
int trunk_fwd(ap_net* h, const float* x0, int n0, const float* x1, int n1, float* feat, hipStream_t st, hipStream_t st_out = nullptr) {
    if (!st_out) st_out = st;
    if (!h->finalized) return fail(AP_ESTATE, "ap_net_finalize has not been called");
    if (n0 <= 0 || n1 < 0 || !x0 || (n1 && !x1) || !feat) return fail(AP_EINVAL, "ap_trunk_fwd: bad arguments");
    if (h->range_mode && h->range_any())
        return fail(AP_ERANGE, "AP_PREC_F16: an earlier trunk pass of this handle produced non-finite features (a stored activation left "
                               "the fp16 range); clear with ap_net_range_status(h, stream, 1) and use AP_PREC_BF16 for this checkpoint");
    int rc_pass = trunk_passes(h, x0, n0, x1, n1, feat, st, st_out);
    if (rc_pass) return rc_pass;
    if (h->range_flag && h->range_mode == 2) {
        HIP_TRY(hipStreamSynchronize(st_out));
        if (h->range_any())
            return fail(AP_ERANGE, "AP_PREC_F16: non-finite trunk features (a stored activation left the fp16 range); use AP_PREC_BF16 "
                                   "for this checkpoint");
    }
    return AP_OK;
}

int trunk_passes(ap_net* h, const float* x0, int n0, const float* x1, int n1, float* feat, hipStream_t st, hipStream_t st_out) {
    const int n_img = n0 + n1;
    h->conv_launches = 0;
    const int chunk = h->chunk > 0 ? h->chunk : 512;
    const size_t IMG_ELEMS = (size_t)3 * 224 * 224;
    if (h->dual_stream && !n1 && n0 >= 128) {                // one list of images (forward_feat_ext, the single-view heads): its two
        n1 = n0 - n0 / 2;                                    // halves as the two concurrent passes (feature rows stay in list order)
        n0 = n0 / 2;
        x1 = x0 + (size_t)n0 * IMG_ELEMS;
    }
    // (measured: +4..5 % at 64 images per view, -4 % at 32, where the launches no longer fill the chip)
    if (h->dual_stream && n0 >= 64 && n1 >= 64 && chunk >= 128) {
        // two views = two concurrent passes: fork from the caller's stream, one pass per internal stream, join.  A view of more
        // than chunk / 2 images goes through its stream in slices of chunk / 2 (same workspace, stream order), so 2 x 256 images
        // are in flight whatever the batch
        if (!h->aux[0]) {
            for (int i = 0; i < 4; ++i) {
                HIP_TRY(hipStreamCreateWithFlags(&h->aux[i], hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
            }
            HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&h->ev_skew, hipEventDisableTiming));
        }
        const int ppv = h->passes_per_view == 2 ? 2 : 1, np = 2 * ppv;
        const int per = chunk / 2 / ppv;                     // images of one slice of one pass
        auto part_of = [&](int q, int* lo, int* cnt) {       // pass q's share of its view
            const int v = q / ppv, part = q % ppv, nv = v ? n1 : n0;
            *lo = part * (nv / ppv);
            *cnt = part == ppv - 1 ? nv - *lo : nv / ppv;
        };
        // every pass's workspace is sized BEFORE the fork: a grow inside a pass would synchronise the device and free
        // buffers while the sibling pass is in flight
        int rounds = 1;
        for (int q = 0; q < np; ++q) {
            int lo, cnt;
            part_of(q, &lo, &cnt);
            int rc = reserve_trunk_ws(h, h->tw[q], std::min(cnt, per));
            if (rc) return rc;
            rounds = std::max(rounds, (cnt + per - 1) / per);
        }
        HIP_TRY(hipEventRecord(h->ev_fork, st));
        std::vector<size_t> ev((size_t)np * rounds * 4, 0);
        int rc = AP_OK, forked = 0;
        for (int q = 0; q < np && !rc; ++q) {
            const int v = q / ppv;
            int lo, cnt;
            part_of(q, &lo, &cnt);
            HIP_TRY(hipStreamWaitEvent(h->aux[q], h->ev_fork, 0));
            forked = q + 1;
            if (q == 1 && np == 2 && h->dual_skew) HIP_TRY(hipStreamWaitEvent(h->aux[1], h->ev_skew, 0));
            const int nr = (cnt + per - 1) / per;            // slices of equal size (+-1): 261 images = 131 + 130, not 256 + 5
            for (int r = 0, s0 = 0, c = 0; r < nr && !rc; ++r, s0 += c) {
                c = cnt / nr + (r < cnt % nr ? 1 : 0);
                const float* xv = (v ? x1 : x0) + (size_t)(lo + s0) * IMG_ELEMS;
                rc = trunk_chunk(h, h->tw[q], xv, c, nullptr, 0, feat + ((v ? (size_t)n0 : 0) + lo + s0) * 2048, h->aux[q],
                                 &ev[((size_t)q * rounds + r) * 4], (q == 0 && np == 2 && r == 0) ? h->dual_skew : 0);
            }
        }
        // ap_net_range_mark_next: each pass stream snapshots ITS range word behind its last kernel of this call
        const int mslot = h->mark_slot;
        h->mark_slot = -1;
        if (mslot >= 0 && h->range_flag)
            for (int q = 0; q < forked; ++q) HIP_TRY(ap_launch_word_copy(h->range_flag + q, h->range_slots + 4 * mslot + q, h->aux[q]));
        // join every stream that forked, also after a failed launch: later calls reuse tw[q] on the caller's stream order
        for (int q = 0; q < forked; ++q) {
            HIP_TRY(hipEventRecord(h->ev_join[q], h->aux[q]));
            HIP_TRY(hipStreamWaitEvent(st_out, h->ev_join[q], 0));
            if (st_out != st) h->unjoined = true;            // st itself is not behind these passes
        }
        if (rc) return rc;
        if (h->tm.on) {
            // per slice round: span over the first and the last pass issued (two streams: exact); a pass without a slice in this
            // round (views of different sizes) lends the other pass's events
            auto quad = [&](int stage, int a, int b) {
                for (int r = 0; r < rounds; ++r)
                    for (int q : {0, np - 1}) {
                        const size_t* e = &ev[((size_t)q * rounds + r) * 4];
                        if (!e[b]) e = &ev[((size_t)(np - 1 - q) * rounds + r) * 4];
                        h->tm.quads[stage].push_back(e[a]);
                        h->tm.quads[stage].push_back(e[b]);
                        if (q == 0) h->tm.qfree[stage].push_back(st_out != st);
                    }
            };
            quad(1, 1, 2);
            if (h->tm.on == 1) { quad(0, 0, 1); quad(2, 2, 3); }
            h->tm.passes++;
        }
        return AP_OK;
    }
    if (h->unjoined) {                                       // an asynchronous two-pass call came before: its passes may still use tw[0]
        for (int q = 0; q < 4; ++q) HIP_TRY(hipStreamWaitEvent(st_out, h->ev_join[q], 0));
        h->unjoined = false;
    }
    if (st_out != st) {                                      // one pass: it runs on st_out, behind the inputs
        if (!h->ev_in) HIP_TRY(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(h->ev_in, st));
        HIP_TRY(hipStreamWaitEvent(st_out, h->ev_in, 0));
        st = st_out;
    }
    for (int i0 = 0; i0 < n_img; i0 += chunk) {
        const int i1 = std::min(n_img, i0 + chunk);
        const int a0 = std::min(i0, n0), a1 = std::min(i1, n0);          // part taken from x0
        const int b0 = std::max(i0, n0) - n0, b1 = std::max(i1, n0) - n0; // part taken from x1
        int rc = trunk_chunk(h, h->tw[0], x0 + a0 * IMG_ELEMS, a1 - a0, x1 ? x1 + b0 * IMG_ELEMS : nullptr, b1 - b0,
                             feat + (size_t)i0 * 2048, st);
        if (rc) return rc;
    }
    if (h->mark_slot >= 0 && h->range_flag) HIP_TRY(ap_launch_word_copy(h->range_flag, h->range_slots + 4 * h->mark_slot, st));
    h->mark_slot = -1;
    if (h->tm.on) h->tm.passes++;
    return AP_OK;
}

struct RegInputs {
    const float *xf0, *xf1, *bb0, *bb1, *pos0, *pos1, *th0, *th1, *sh0, *sh1;
    int th0_bs, th1_bs, sh0_bs, sh1_bs;
};

// xf rows: view 0 then view 1 (two_view) laid out by the caller as two pointers; H rows follow the same order
int regressor_run(ap_net* h, const RegInputs& in, int B, int iters, int two_view, const float* partner,
                  int partner_ld, int pos_bs, float* pose0, float* betas0, float* pose1, float* betas1,
                  hipStream_t st) {
    if (!h->finalized) return fail(AP_ESTATE, "ap_net_finalize has not been called");
    if (h->variant == 1)
        return fail(AP_ESTATE, "regressor entry points need a copenet-layout handle (variants 0, 2, 3), not hmr");
    if (h->variant == 2 && (two_view || partner))
        return fail(AP_ESTATE, "a copenet_singleview handle has no cross-view inputs");
    if (h->variant != 2 && !two_view && !partner) return fail(AP_EINVAL, "regressor step: partner state missing");
    if (B <= 0 || iters < 1) return fail(AP_EINVAL, "regressor: bad B / iters");
    const int rows = two_view ? 2 * B : B;
    HIP_TRY(h->ws_H.reserve((size_t)rows * 1024 * 4));
    HIP_TRY(h->ws_T1.reserve((size_t)rows * 1024 * 4));
    HIP_TRY(h->ws_T2.reserve((size_t)rows * 1024 * 4));
    HIP_TRY(h->ws_S.reserve((size_t)rows * SLD * 4));
    HIP_TRY(h->ws_D.reserve((size_t)rows * DLD * 4));
    HIP_TRY(h->ws_state.reserve((size_t)rows * ST * 4));
    size_t e0 = 0, e1 = 0;
    if (h->tm.on == 1) HIP_TRY(h->tm.rec(st, &e0));
    if (h->fold && h->fuse_ief) {
        HIP_TRY(h->ws_H.reserve((size_t)ap_reg_fold_part_floats(rows) * 4));
        RegInitArgs ia{};
        ia.pos0 = in.pos0; ia.pos1 = in.pos1; ia.theta0 = in.th0; ia.theta1 = in.th1; ia.shape0 = in.sh0; ia.shape1 = in.sh1;
        ia.theta0_bs = in.th0_bs; ia.theta1_bs = in.th1_bs; ia.shape0_bs = in.sh0_bs; ia.shape1_bs = in.sh1_bs;
        ia.pos_bs = pos_bs; ia.rows = rows;
        ia.mean_pose = h->mean_pose.as<float>(); ia.mean_shape = h->mean_shape.as<float>();
        ia.state = nullptr; ia.B = B;
        HIP_TRY(ap_launch_reg_fold_ief(ia, in.xf0, two_view ? in.xf1 : in.xf0, in.bb0, in.bb1, partner, partner_ld,
                                       h->foldT_feat.as<float>(), h->foldT_state.as<float>(), h->fold_bias.as<float>(),
                                       h->ws_H.as<float>(), iters, two_view, pose0, betas0, pose1, betas1, st));
        if (h->tm.on == 1) {
            HIP_TRY(h->tm.rec(st, &e1));
            h->tm.marks[3].push_back(e0); h->tm.marks[3].push_back(e1);
        }
        return AP_OK;
    }
    float *Hb = h->ws_H.as<float>(), *T1 = h->ws_T1.as<float>(), *T2 = h->ws_T2.as<float>(), *S = h->ws_S.as<float>(),
          *D = h->ws_D.as<float>(), *state = h->ws_state.as<float>();
    int rc;
    // trunk-feature part (+ bias), constant over the iterations: of fc1 (literal chain) or of the folded map
    const Layer& Lf = h->fold ? h->fold_feat : h->fc1_feat;
    const int hld = h->fold ? DLD : 1024;
    if (two_view && in.xf1 == in.xf0 + (size_t)B * 2048) {          // both views contiguous: one launch
        if ((rc = run_gemm(Lf, in.xf0, 2048, 2048, 2 * B, Hb, hld, nullptr, 0, st))) return rc;
    } else {
        if ((rc = run_gemm(Lf, in.xf0, 2048, 2048, B, Hb, hld, nullptr, 0, st))) return rc;
        if (two_view)
            if ((rc = run_gemm(Lf, in.xf1, 2048, 2048, B, Hb + (size_t)B * hld, hld, nullptr, 0, st))) return rc;
    }
    RegInitArgs ia{};
    ia.pos0 = in.pos0; ia.pos1 = in.pos1; ia.theta0 = in.th0; ia.theta1 = in.th1; ia.shape0 = in.sh0; ia.shape1 = in.sh1;
    ia.theta0_bs = in.th0_bs; ia.theta1_bs = in.th1_bs; ia.shape0_bs = in.sh0_bs; ia.shape1_bs = in.sh1_bs;
    ia.pos_bs = pos_bs; ia.rows = rows;
    ia.mean_pose = h->mean_pose.as<float>(); ia.mean_shape = h->mean_shape.as<float>();
    ia.state = state; ia.B = B;
    HIP_TRY(ap_launch_reg_init(ia, st));
    for (int it = 0; it < iters; ++it) {
        HIP_TRY(ap_launch_reg_update_assemble(state, it ? D : nullptr, DLD, in.bb0, in.bb1, partner, partner_ld, S, B,
                                              two_view, st));
        if (h->fold) {
            if ((rc = run_gemm(h->fold_state, S, SLD, SLD, rows, D, DLD, Hb, DLD, st))) return rc;
        } else {
            if ((rc = run_gemm(h->fc1_state, S, SLD, SLD, rows, T1, 1024, Hb, 1024, st))) return rc;
            if ((rc = run_gemm(h->fc2, T1, 1024, 1024, rows, T2, 1024, nullptr, 0, st))) return rc;
            if ((rc = run_gemm(h->dec, T2, 1024, 1024, rows, D, DLD, nullptr, 0, st))) return rc;
        }
    }
    // fold the last delta into the state and emit
    HIP_TRY(ap_launch_reg_update_assemble(state, D, DLD, in.bb0, in.bb1, partner, partner_ld, S, B, two_view, st));
    HIP_TRY(ap_launch_reg_output(state, pose0, betas0, pose1, betas1, B, two_view, st));
    if (h->tm.on == 1) {
        HIP_TRY(h->tm.rec(st, &e1));
        h->tm.marks[3].push_back(e0); h->tm.marks[3].push_back(e1);
    }
    return AP_OK;
}

}  // namespace

// ================================================================================== C ABI
extern "C" {

const char* ap_version(void) { return "airpose_hip 0.6 (gfx950; abi 9)"; }
int ap_abi_version(void) { return AP_ABI_VERSION; }
const char* ap_last_error(void) { return g_err.c_str(); }

int ap_net_create(ap_net** out, int device, int precision, int variant) {
    if (!out || !prec_valid(precision) || (variant < 0 || variant > 3))
        return fail(AP_EINVAL, "ap_net_create: bad arguments");
    HIP_TRY(hipSetDevice(device));
    ap_net* h = new ap_net();
    h->device = device; h->prec = precision; h->variant = variant;
    if (precision == AP_PREC_F16) {
        hipError_t e = hipHostMalloc((void**)&h->range_flag, 4 * sizeof(int), hipHostMallocMapped);
        if (e != hipSuccess) { delete h; return fail((int)e, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
        for (int q = 0; q < 4; ++q) h->range_flag[q] = 0;
        e = hipHostMalloc((void**)&h->range_slots, AP_RANGE_SLOTS * 4 * sizeof(int), hipHostMallocMapped);
        if (e != hipSuccess) { (void)hipHostFree(h->range_flag); delete h; return fail((int)e, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
        for (int i = 0; i < AP_RANGE_SLOTS * 4; ++i) h->range_slots[i] = 0;
        for (int q = 0; q < 4; ++q) h->tw[q].rflag = h->range_flag + q;
    }
    *out = h;
    return AP_OK;
}

void ap_net_destroy(ap_net* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for (DevBuf* b : {&h->stem_w, &h->stem_wpk, &h->stem_wpk_lo, &h->stem_scale, &h->stem_shift, &h->mean_pose, &h->mean_shape, &h->mean_cam, &h->tw[0].ws_stem, &h->tw[0].ws_a,
                      &h->tw[0].ws_b, &h->tw[0].ws_t1, &h->tw[0].ws_t2, &h->tw[0].ws_ds, &h->tw[1].ws_stem, &h->tw[1].ws_a, &h->tw[1].ws_b,
                      &h->tw[1].ws_t1, &h->tw[1].ws_t2, &h->tw[1].ws_ds, &h->tw[2].ws_stem, &h->tw[2].ws_a, &h->tw[2].ws_b, &h->tw[2].ws_t1,
                      &h->tw[2].ws_t2, &h->tw[2].ws_ds, &h->tw[3].ws_stem, &h->tw[3].ws_a, &h->tw[3].ws_b, &h->tw[3].ws_t1, &h->tw[3].ws_t2,
                      &h->tw[3].ws_ds, &h->ws_feat, &h->ws_H, &h->ws_S, &h->ws_T1, &h->ws_T2,
                      &h->ws_D, &h->ws_state})
        b->release();
    auto rel = [](Layer& L) { release_layer(L); };
    release_blocks(h);
    rel(h->fc1_feat); rel(h->fc1_state); rel(h->fc2); rel(h->dec); rel(h->fold_feat); rel(h->fold_state);
    h->foldT_feat.release(); h->foldT_state.release(); h->fold_bias.release();
    h->tm.destroy();
    for (int i = 0; i < 4; ++i) {
        if (h->aux[i]) (void)hipStreamDestroy(h->aux[i]);
        if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_in) (void)hipEventDestroy(h->ev_in);
    if (h->ev_skew) (void)hipEventDestroy(h->ev_skew);
    if (h->range_flag) (void)hipHostFree(h->range_flag);
    if (h->range_slots) (void)hipHostFree(h->range_slots);
    if (h->probe_ref) { ap_net* r = h->probe_ref; h->probe_ref = nullptr; ap_net_destroy(r); }
    for (DevBuf* b : {&h->probe_x, &h->probe_bb, &h->probe_pos, &h->probe_feat, &h->probe_out}) b->release();
    delete h;
}

int ap_net_set_tensor(ap_net* h, const char* name, const float* host_data, const int64_t* shape, int ndim) {
    if (!h || !name || !host_data || ndim < 0 || ndim > 8) return fail(AP_EINVAL, "ap_net_set_tensor: bad arguments");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] < 0) return fail(AP_ESHAPE, "negative dimension");
        t.shape.push_back(shape[i]);
        n *= (size_t)shape[i];
    }
    t.data.assign(host_data, host_data + n);
    h->tensors[name] = std::move(t);
    h->finalized = false;
    return AP_OK;
}

int ap_net_finalize(ap_net* h) {
    if (!h) return fail(AP_EINVAL, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    h->f16_overflow = false;
    if (h->probe_ref) { ap_net* r = h->probe_ref; h->probe_ref = nullptr; ap_net_destroy(r); }   // (packed from the previous tensors)
    int rc = finalize_trunk(h);
    if (rc) return rc;
    if ((rc = finalize_regressor(h))) return rc;
    // fp16 storage: a BatchNorm-folded weight above 65 504 would become inf on the device
    if (h->f16_overflow)
        return fail(AP_ESHAPE, "ap_net_finalize: a (BatchNorm-folded) weight exceeds the fp16 range of AP_PREC_F16; "
                               "use AP_PREC_BF16 (precision='bf16': fp32's exponent range)");
    h->finalized = true;
    return AP_OK;
}

int ap_net_precision(const ap_net* h) { return h ? h->prec : AP_EINVAL; }

int ap_net_set_range_check(ap_net* h, int mode) {
    if (!h || mode < 0 || mode > 2) return fail(AP_EINVAL, "ap_net_set_range_check: handle, mode in {0, 1, 2}");
    h->range_mode = mode;
    return AP_OK;
}

int ap_net_range_status(ap_net* h, void* stream, int reset) {
    if (!h) return fail(AP_EINVAL, "null handle");
    if (!h->range_flag) return AP_OK;                        // only fp16 storage has a range to leave
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    const int bad = h->range_any();
    if (reset) for (int q = 0; q < 4; ++q) __atomic_store_n(h->range_flag + q, 0, __ATOMIC_RELAXED);
    if (bad) return fail(AP_ERANGE, "AP_PREC_F16: a trunk pass produced non-finite features (a stored activation left the fp16 range)");
    return AP_OK;
}

int ap_net_last_conv_launches(const ap_net* h) { return h ? h->conv_launches : AP_EINVAL; }

int ap_net_range_peek(const ap_net* h) {
    if (!h) return fail(AP_EINVAL, "null handle");
    if (h->range_any())
        return fail(AP_ERANGE, "AP_PREC_F16: a trunk pass of this handle left the fp16 range (flag read without a stream sync)");
    return AP_OK;
}

int ap_net_range_mark_next(ap_net* h, int slot) {
    if (!h || slot < -1 || slot >= AP_RANGE_SLOTS) return fail(AP_EINVAL, "ap_net_range_mark_next: handle, slot in [-1, AP_RANGE_SLOTS)");
    h->mark_slot = h->range_flag ? slot : -1;
    // (the slot's previous batch is done -- the caller waited for it before reusing the slot -- so the host may clear its words)
    if (h->mark_slot >= 0) for (int q = 0; q < 4; ++q) __atomic_store_n(h->range_slots + 4 * slot + q, 0, __ATOMIC_RELAXED);
    return AP_OK;
}

int ap_net_range_slot(const ap_net* h, int slot) {
    if (!h || slot < 0 || slot >= AP_RANGE_SLOTS) return fail(AP_EINVAL, "ap_net_range_slot: handle, slot in [0, AP_RANGE_SLOTS)");
    int v = 0;
    if (h->range_slots) for (int q = 0; q < 4; ++q) v |= __atomic_load_n(h->range_slots + 4 * slot + q, __ATOMIC_RELAXED);
    if (v)
        return fail(AP_ERANGE, "AP_PREC_F16: a stored activation of this batch's trunk passes (or of an earlier batch's) left the fp16 "
                               "range; use precision bf16 / bf16x2 for this checkpoint");
    return AP_OK;
}

int ap_trunk_fwd(ap_net* h, const float* x_nchw, int n_img, float* feat, void* stream) {
    if (!h) return fail(AP_EINVAL, "null handle");
    return trunk_fwd(h, x_nchw, n_img, nullptr, 0, feat, (hipStream_t)stream);
}

int ap_trunk_fwd_twoview(ap_net* h, const float* x0, const float* x1, int B, float* feat, void* stream) {
    if (!h || !x0 || !x1 || !feat) return fail(AP_EINVAL, "ap_trunk_fwd_twoview: null argument");
    if (B <= 0) return fail(AP_EINVAL, "ap_trunk_fwd_twoview: bad batch");
    return trunk_fwd(h, x0, B, x1, B, feat, (hipStream_t)stream);
}

int ap_trunk_fwd_twoview_async(ap_net* h, const float* x0, const float* x1, int B, float* feat, void* in_stream, void* out_stream) {
    if (!h || !x0 || !x1 || !feat) return fail(AP_EINVAL, "ap_trunk_fwd_twoview_async: null argument");
    if (B <= 0) return fail(AP_EINVAL, "ap_trunk_fwd_twoview_async: bad batch");
    return trunk_fwd(h, x0, B, x1, B, feat, (hipStream_t)in_stream, (hipStream_t)out_stream);
}

int ap_regressor_fwd(ap_net* h, const float* xf0, const float* xf1, const float* bb0, const float* bb1,
                     const float* pos0, const float* pos1, const float* init_theta0, int theta0_bs,
                     const float* init_theta1, int theta1_bs, const float* init_shape0, int shape0_bs,
                     const float* init_shape1, int shape1_bs, int B, int iters, float* pose0, float* betas0,
                     float* pose1, float* betas1, void* stream) {
    if (!h || !xf0 || !xf1 || !bb0 || !bb1 || !pos0 || !pos1 || !pose0 || !betas0 || !pose1 || !betas1)
        return fail(AP_EINVAL, "ap_regressor_fwd: null argument");
    RegInputs in{xf0, xf1, bb0, bb1, pos0, pos1, init_theta0, init_theta1, init_shape0, init_shape1,
                 theta0_bs, theta1_bs, shape0_bs, shape1_bs};
    return regressor_run(h, in, B, iters, 1, nullptr, 0, 3, pose0, betas0, pose1, betas1, (hipStream_t)stream);
}

int ap_regressor_step(ap_net* h, const float* xf, const float* bb, const float* pose_in, const float* betas_in,
                      const float* partner, int partner_ld, int B, float* pose_out, float* betas_out, void* stream) {
    if (!h || !xf || !bb || !pose_in || !betas_in || !partner || !pose_out || !betas_out || partner_ld < 136)
        return fail(AP_EINVAL, "ap_regressor_step: bad argument");
    RegInputs in{xf, nullptr, bb, nullptr, pose_in, nullptr, pose_in + 3, nullptr, betas_in, nullptr, 135, 0, 10, 0};
    return regressor_run(h, in, B, 1, 0, partner, partner_ld, 135, pose_out, betas_out, nullptr, nullptr,
                         (hipStream_t)stream);
}

int ap_regressor_feat_part(ap_net* h, const float* xf, int B, float* hfeat, void* stream) {
    if (!h || !xf || !hfeat || B <= 0) return fail(AP_EINVAL, "ap_regressor_feat_part: bad argument");
    if (!h->finalized) return fail(AP_ESTATE, "ap_net_finalize has not been called");
    if (h->variant != 0 || !h->fold)
        return fail(AP_ESTATE, "ap_regressor_feat_part / _step_local / _step_finish evaluate the folded two-view map (ap_net_fold_status == 1); "
                               "this handle runs the literal chain: use ap_regressor_step");
    HIP_TRY(h->ws_H.reserve((size_t)ap_reg_fold_part_floats(B) * 4));
    HIP_TRY(ap_launch_reg_feat_part(xf, B, h->foldT_feat.as<float>(), h->fold_bias.as<float>(), h->ws_H.as<float>(), hfeat, (hipStream_t)stream));
    return AP_OK;
}

int ap_regressor_step_local(ap_net* h, const float* hfeat, const float* bb, const float* pose_in, const float* betas_in, int B,
                            float* partial, void* stream) {
    if (!h || !hfeat || !bb || !pose_in || !betas_in || !partial || B <= 0) return fail(AP_EINVAL, "ap_regressor_step_local: bad argument");
    if (!h->finalized) return fail(AP_ESTATE, "ap_net_finalize has not been called");
    if (h->variant != 0 || !h->fold) return fail(AP_ESTATE, "ap_regressor_step_local needs the folded two-view map (ap_net_fold_status == 1)");
    HIP_TRY(ap_launch_reg_step_local(hfeat, bb, pose_in, betas_in, B, h->foldT_state.as<float>(), partial, (hipStream_t)stream));
    return AP_OK;
}

int ap_regressor_step_finish(ap_net* h, const float* partial, const float* pose_in, const float* betas_in, const float* partner,
                             int partner_ld, int B, float* pose_out, float* betas_out, void* stream) {
    if (!h || !partial || !pose_in || !betas_in || !partner || !pose_out || !betas_out || partner_ld < 136 || B <= 0)
        return fail(AP_EINVAL, "ap_regressor_step_finish: bad argument");
    if (!h->finalized) return fail(AP_ESTATE, "ap_net_finalize has not been called");
    if (h->variant != 0 || !h->fold) return fail(AP_ESTATE, "ap_regressor_step_finish needs the folded two-view map (ap_net_fold_status == 1)");
    HIP_TRY(ap_launch_reg_step_finish(partial, pose_in, betas_in, partner, partner_ld, B, h->foldT_state.as<float>(), pose_out, betas_out,
                                      (hipStream_t)stream));
    return AP_OK;
}

// The handle's arithmetic against the exact-fp32 mode of the SAME weights on a seeded probe batch, on the GPU: what the 1e-4 claim of a
// 16-bit mode is worth on THIS checkpoint.  Only the trunk differs between the modes (the regressor is fp32 everywhere), so the
// reference is an fp32 trunk packed from the handle's own host tensors (kept until the next ap_net_finalize) and both feature sets
// go through the handle's regressor.
int ap_net_parity_probe(ap_net* h, int n_pairs, uint64_t seed, double* err8, void* stream) {
    if (!h || !err8 || n_pairs < 1 || n_pairs > 64) return fail(AP_EINVAL, "ap_net_parity_probe: handle, 1 <= n_pairs <= 64, err8");
    if (!h->finalized) return fail(AP_ESTATE, "ap_net_finalize has not been called");
    if (h->variant != 0) return fail(AP_ESTATE, "ap_net_parity_probe: two-view copenet handles only");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(h->device));
    for (int i = 0; i < 8; ++i) err8[i] = 0.0;
    if (h->prec == AP_PREC_FP32) return AP_OK;               // the reference itself
    if (!h->probe_ref) {
        ap_net* r = new ap_net();
        r->device = h->device; r->prec = AP_PREC_FP32; r->variant = h->variant; r->tensors_of = h;
        const int rc = finalize_trunk(r);
        if (rc) { ap_net_destroy(r); return rc; }
        r->finalized = true;                                 // (trunk only: its regressor is never called)
        h->probe_ref = r;
    }
    const int B = n_pairs;
    const size_t IMG = (size_t)3 * 224 * 224;
    HIP_TRY(h->probe_x.reserve(2 * B * IMG * 4));
    HIP_TRY(h->probe_bb.reserve((size_t)2 * B * 3 * 4));
    HIP_TRY(h->probe_pos.reserve((size_t)B * 3 * 4));
    HIP_TRY(h->probe_feat.reserve((size_t)2 * 2 * B * 2048 * 4));
    HIP_TRY(h->probe_out.reserve((size_t)2 * 2 * B * 145 * 4));
    float *x = h->probe_x.as<float>(), *bb = h->probe_bb.as<float>(), *pos = h->probe_pos.as<float>();
    HIP_TRY(ap_launch_probe_inputs(x, 2 * B * IMG, bb, 2 * B, seed, st));
    std::vector<float> hp((size_t)B * 3);
    for (int b = 0; b < B; ++b) { hp[3 * b] = 0.f; hp[3 * b + 1] = 0.f; hp[3 * b + 2] = 10.f * 0.05f; }   // copenet_twoview.py:184,201-203
    HIP_TRY(hipMemcpyAsync(pos, hp.data(), hp.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));                       // (hp is a local)
    std::vector<float> out[2];
    for (int m = 0; m < 2; ++m) {
        ap_net* net = m ? h->probe_ref : h;
        float* feat = h->probe_feat.as<float>() + (size_t)m * 2 * B * 2048;
        float* o = h->probe_out.as<float>() + (size_t)m * 2 * B * 145;
        int rc = trunk_fwd(net, x, B, x + B * IMG, B, feat, st);
        if (rc) return rc;
        RegInputs in{feat, feat + (size_t)B * 2048, bb, bb + (size_t)B * 3, pos, pos, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
        float *p0 = o, *p1 = o + (size_t)B * 135, *b0 = o + (size_t)2 * B * 135, *b1 = b0 + (size_t)B * 10;
        if ((rc = regressor_run(h, in, B, 3, 1, nullptr, 0, 3, p0, b0, p1, b1, st))) return rc;
        out[m].resize((size_t)2 * B * 145);
        HIP_TRY(hipMemcpyAsync(out[m].data(), o, out[m].size() * 4, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    if (h->range_any())
        return fail(AP_ERANGE, "ap_net_parity_probe: the probe batch left the fp16 range");
    // slices: translation (3), 6-D rotations (132), betas (10), projected root u = f tx / tz + cx, v likewise (the 2-D error is the
    // translation error seen through the camera: f = 1475, centre (960, 540), constants.py:7-11); norm-wise max|a-b| / max|b| in
    // err8[0..3], element-wise max |a-b| / (1e-2 + |b|) in err8[4..7]
    double num[4] = {0, 0, 0, 0}, den[4] = {0, 0, 0, 0}, el[4] = {0, 0, 0, 0};
    auto acc = [&](int sl, double a, double b) {
        num[sl] = std::max(num[sl], std::fabs(a - b));
        den[sl] = std::max(den[sl], std::fabs(b));
        el[sl] = std::max(el[sl], std::fabs(a - b) / (1e-2 + std::fabs(b)));
    };
    for (int r = 0; r < 2 * B; ++r) {
        const float *pa = &out[0][(size_t)r * 135], *pb = &out[1][(size_t)r * 135];
        for (int e = 0; e < 135; ++e) acc(e < 3 ? 0 : 1, pa[e], pb[e]);
        const float *ba = &out[0][(size_t)2 * B * 135 + (size_t)r * 10], *bbv = &out[1][(size_t)2 * B * 135 + (size_t)r * 10];
        for (int e = 0; e < 10; ++e) acc(2, ba[e], bbv[e]);
        if (std::fabs(pb[2]) > 1e-6 && std::fabs(pa[2]) > 1e-6) {
            acc(3, 1475.0 * pa[0] / pa[2] + 960.0, 1475.0 * pb[0] / pb[2] + 960.0);
            acc(3, 1475.0 * pa[1] / pa[2] + 540.0, 1475.0 * pb[1] / pb[2] + 540.0);
        }
    }
    for (int sl = 0; sl < 4; ++sl) { err8[sl] = den[sl] > 0 ? num[sl] / den[sl] : 0.0; err8[4 + sl] = el[sl]; }
    return AP_OK;
}

int ap_singleview_fwd(ap_net* h, const float* x, const float* bb, const float* pos, const float* init_theta,
                      int theta_bs, const float* init_shape, int shape_bs, int B, int iters, float* pose, float* betas,
                      void* stream) {
    if (!h || !x || !bb || !pos || !pose || !betas) return fail(AP_EINVAL, "ap_singleview_fwd: null argument");
    if (h->variant != 2) return fail(AP_ESTATE, "ap_singleview_fwd needs a copenet_singleview (variant 2) handle");
    if (B <= 0) return fail(AP_EINVAL, "ap_singleview_fwd: bad batch");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(h->ws_feat.reserve((size_t)B * 2048 * 4));
    float* f = h->ws_feat.as<float>();
    int rc = trunk_fwd(h, x, B, nullptr, 0, f, st);
    if (rc) return rc;
    RegInputs in{f, nullptr, bb, nullptr, pos, nullptr, init_theta, nullptr, init_shape, nullptr, theta_bs, 0, shape_bs, 0};
    return regressor_run(h, in, B, iters, 0, nullptr, 0, 3, pose, betas, nullptr, nullptr, st);
}

// feature-level evaluation of the single-view head: `iters` regressor evaluations from pre-computed trunk features
// (model_copenet_singleview.py:159-170 for iters = 1)
int ap_singleview_reg(ap_net* h, const float* xf, const float* bb, const float* pos, const float* init_theta, int theta_bs,
                      const float* init_shape, int shape_bs, int B, int iters, float* pose, float* betas, void* stream) {
    if (!h || !xf || !bb || !pos || !pose || !betas) return fail(AP_EINVAL, "ap_singleview_reg: null argument");
    if (h->variant != 2) return fail(AP_ESTATE, "ap_singleview_reg needs a copenet_singleview (variant 2) handle");
    if (B <= 0) return fail(AP_EINVAL, "ap_singleview_reg: bad batch");
    RegInputs in{xf, nullptr, bb, nullptr, pos, nullptr, init_theta, nullptr, init_shape, nullptr, theta_bs, 0, shape_bs, 0};
    return regressor_run(h, in, B, iters, 0, nullptr, 0, 3, pose, betas, nullptr, nullptr, (hipStream_t)stream);
}

int ap_muhmr_fwd(ap_net* h, const float* x0, const float* x1, const float* init_cam0, int cam0_bs, const float* init_cam1,
                 int cam1_bs, const float* init_theta0, int theta0_bs, const float* init_theta1, int theta1_bs,
                 const float* init_shape0, int shape0_bs, const float* init_shape1, int shape1_bs, int B, int iters,
                 float* campose0, float* betas0, float* campose1, float* betas1, void* stream) {
    if (!h || !x0 || !x1 || !campose0 || !betas0 || !campose1 || !betas1) return fail(AP_EINVAL, "ap_muhmr_fwd: null argument");
    if (h->variant != 3) return fail(AP_ESTATE, "ap_muhmr_fwd needs a muhmr (variant 3) handle");
    if (B <= 0) return fail(AP_EINVAL, "ap_muhmr_fwd: bad batch");
    if (cam0_bs != cam1_bs) return fail(AP_EINVAL, "ap_muhmr_fwd: init_cam0 / init_cam1 must share their batch stride");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(h->ws_feat.reserve((size_t)2 * B * 2048 * 4));
    float* f0 = h->ws_feat.as<float>();
    float* f1 = f0 + (size_t)B * 2048;
    int rc = trunk_fwd(h, x0, B, x1, B, f0, st);
    if (rc) return rc;
    const float* c0 = init_cam0 ? init_cam0 : h->mean_cam.as<float>();
    const float* c1 = init_cam1 ? init_cam1 : h->mean_cam.as<float>();
    const int cbs = init_cam0 ? cam0_bs : 0;
    if (!!init_cam0 != !!init_cam1) return fail(AP_EINVAL, "ap_muhmr_fwd: give both initial cameras or neither");
    // bb has zero weight in the re-mapped fc1: any finite [B][3] floats do (the head of the feature rows is at hand)
    RegInputs in{f0, f1, f0, f1, c0, c1, init_theta0, init_theta1, init_shape0, init_shape1, theta0_bs, theta1_bs, shape0_bs, shape1_bs};
    return regressor_run(h, in, B, iters, 1, nullptr, 0, cbs, campose0, betas0, campose1, betas1, st);
}

int ap_copenet_fwd(ap_net* h, const float* x0, const float* x1, const float* bb0, const float* bb1,
                   const float* pos0, const float* pos1, const float* init_theta0, int theta0_bs,
                   const float* init_theta1, int theta1_bs, const float* init_shape0, int shape0_bs,
                   const float* init_shape1, int shape1_bs, int B, int iters, float* pose0, float* betas0,
                   float* pose1, float* betas1, void* stream) {
    if (!h || !x0 || !x1) return fail(AP_EINVAL, "ap_copenet_fwd: null argument");
    if (B <= 0) return fail(AP_EINVAL, "ap_copenet_fwd: bad batch");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(h->ws_feat.reserve((size_t)2 * B * 2048 * 4));
    float* f0 = h->ws_feat.as<float>();
    float* f1 = f0 + (size_t)B * 2048;
    int rc = trunk_fwd(h, x0, B, x1, B, f0, st);      // both views in one pass (shared weights)
    if (rc) return rc;
    return ap_regressor_fwd(h, f0, f1, bb0, bb1, pos0, pos1, init_theta0, theta0_bs, init_theta1, theta1_bs,
                            init_shape0, shape0_bs, init_shape1, shape1_bs, B, iters, pose0, betas0, pose1, betas1,
                            stream);
}

int ap_conv2d_nhwc(int precision, const void* x, const void* w, const float* scale, const float* shift,
                   const void* res, void* y, int N, int H, int W, int Cin, int Cout, int ksize, int stride, int pad,
                   int relu, void* stream) {
    const int bf = prec_half(precision);
    if (!prec_valid(precision) || !x || !w || !scale ||
        !shift || !y || N <= 0 ||
        H <= 0 || W <= 0 || ksize <= 0 || stride <= 0 || pad < 0)
        return fail(AP_EINVAL, "ap_conv2d_nhwc: bad argument");
    if (Cin % (bf ? 64 : 32) || Cout % (precision == AP_PREC_FP32 ? 4 : 8) || Cin <= 0 || Cout <= 0)
        return fail(AP_ESHAPE, "ap_conv2d_nhwc: Cin must be a multiple of 64 (bf16) / 32 (fp32), Cout of 8 / 4");
    ConvArgs a{};
    a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.res = res; a.y = y;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin;
    a.Ho = (H + 2 * pad - ksize) / stride + 1;
    a.Wo = (W + 2 * pad - ksize) / stride + 1;
    if (a.Ho <= 0 || a.Wo <= 0) return fail(AP_ESHAPE, "ap_conv2d_nhwc: empty output");
    a.Cout = Cout; a.KH = a.KW = ksize; a.stride = stride; a.pad = pad;
    a.M = N * a.Ho * a.Wo;
    a.ldx = Cin; a.ldy = Cout; a.ldr = Cout; a.wld = ksize * ksize * Cin; a.relu = relu;
    HIP_TRY(dispatch_conv(a, precision, (hipStream_t)stream));
    return AP_OK;
}

int ap_bottleneck64_nhwc(int precision, const void* x, const void* w1, const float* s1, const float* h1, const void* w2,
                         const float* s2, const float* h2, const void* w3, const float* s3, const float* h3, void* y,
                         int N, int H, int W, int Cin, int downsample, void* stream) {
    if (!prec_half(precision) || !x || !w1 || !s1 || !h1 || !w2 || !s2 || !h2 || !w3 || !s3 || !h3 || !y || N <= 0)
        return fail(AP_EINVAL, "ap_bottleneck64_nhwc: bad argument (precision: AP_PREC_BF16 or AP_PREC_F16)");
    if (H <= 0 || W <= 0 || H % 14 || W % 14 || !((Cin == 256 && !downsample) || (Cin == 64 && downsample)))
        return fail(AP_ESHAPE, "ap_bottleneck64_nhwc: H, W multiples of 14; Cin 256 (identity) or 64 (downsample)");
    BneckArgs a{};
    a.x = x; a.y = y; a.w1 = w1; a.w2 = w2; a.w3 = w3;
    a.s1 = s1; a.h1 = h1; a.s2 = s2; a.h2 = h2; a.s3 = s3; a.h3 = h3;
    a.N = N; a.H = H; a.W = W;
    HIP_TRY(zero_line(&a.zero));
    a.dbg = g_conv_dbg;
    HIP_TRY(H16(precision, ap_launch_bneck2)(a, downsample ? 1 : 0, (hipStream_t)stream));
    return AP_OK;
}

int ap_bottleneck64_tail_nhwc(int precision, const void* x, const void* w1, const float* s1, const float* h1, const void* w2,
                              const float* s2, const float* h2, const void* w3, const float* s3, const float* h3, void* y,
                              const void* w1n, const float* s1n, const float* h1n, void* t1n, int y_even, int N, int H, int W,
                              void* stream) {
    if (!prec_half(precision) || !x || !w1 || !s1 || !h1 || !w2 || !s2 || !h2 || !w3 || !s3 || !h3 || !y || !w1n || !s1n || !h1n ||
        !t1n || N <= 0)
        return fail(AP_EINVAL, "ap_bottleneck64_tail_nhwc: bad argument (precision: AP_PREC_BF16 or AP_PREC_F16)");
    if (H <= 0 || W <= 0 || H % 14 || W % 14) return fail(AP_ESHAPE, "ap_bottleneck64_tail_nhwc: H, W multiples of 14");
    BneckArgs a{};
    a.x = x; a.y = y; a.w1 = w1; a.w2 = w2; a.w3 = w3;
    a.s1 = s1; a.h1 = h1; a.s2 = s2; a.h2 = h2; a.s3 = s3; a.h3 = h3;
    a.w1n = w1n; a.s1n = s1n; a.h1n = h1n; a.t1n = t1n; a.y_even = y_even != 0;
    a.N = N; a.H = H; a.W = W;
    HIP_TRY(zero_line(&a.zero));
    a.dbg = g_conv_dbg;
    HIP_TRY(H16(precision, ap_launch_bneck2)(a, 0, (hipStream_t)stream));
    return AP_OK;
}

int64_t ap_conv_pw_stream_bytes(int Cin, int Cout) {
    return (Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 256 == 0) ? (int64_t)k_bf16::ap_conv_pw_stream_bytes(Cin, Cout) : -1;
}

int ap_conv_pw_pack(int precision, const void* w, int Cin, int Cout, void* wstream, void* stream) {
    if (!prec_half(precision) || !w || !wstream || Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 256)
        return fail(AP_EINVAL, "ap_conv_pw_pack: bad argument (precision: AP_PREC_BF16 or AP_PREC_F16; Cin % 32 == 0, Cout % 256 == 0)");
    HIP_TRY(H16(precision, ap_launch_conv_pw_pack)(w, wstream, Cin, Cout, Cin, (hipStream_t)stream));
    return AP_OK;
}

int ap_conv_pw_nhwc(int precision, const void* x, const void* wstream, const float* scale, const float* shift, const void* res,
                    void* y, int M, int Cin, int Cout, void* stream) {
    if (!prec_half(precision) || !x || !wstream || !scale || !shift || !y)
        return fail(AP_EINVAL, "ap_conv_pw_nhwc: bad argument (precision: AP_PREC_BF16 or AP_PREC_F16)");
    if (!k_bf16::ap_conv_pw_supported(M, Cin, Cout))
        return fail(AP_ESHAPE, "ap_conv_pw_nhwc: M must be a multiple of 196, Cin of 128 (>= 256), Cout of 256");
    PwArgs p{};
    p.x = x; p.y = y; p.res = res; p.wfrag = wstream; p.scale = scale; p.shift = shift; p.M = M; p.Cin = Cin; p.Cout = Cout; p.relu = 1;
    HIP_TRY(H16(precision, ap_launch_conv_pw)(p, (hipStream_t)stream));
    return AP_OK;
}

int ap_conv_pw_ds_nhwc(int precision, const void* t2, const void* x, const void* wstream, const float* scale, const float* shift,
                       void* y, int N, int Ho, int Cin, int Cin2, int Cout, int stride, void* stream) {
    if (!prec_half(precision) || !t2 || !x || !wstream || !scale || !shift || !y || N <= 0 || Ho <= 0 || stride < 1 || stride > 2)
        return fail(AP_EINVAL, "ap_conv_pw_ds_nhwc: bad argument (precision: AP_PREC_BF16 or AP_PREC_F16; stride 1 or 2)");
    PwArgs p{};
    p.x = t2; p.y = y; p.wfrag = wstream; p.scale = scale; p.shift = shift; p.M = N * Ho * Ho; p.Cin = Cin; p.Cout = Cout; p.relu = 1;
    p.x2 = x; p.Cin2 = Cin2; p.Ho = p.Wo = Ho; p.H2 = p.W2 = Ho * stride; p.stride2 = stride;
    if (!k_bf16::ap_conv_pw_ds_supported(p))
        return fail(AP_ESHAPE, "ap_conv_pw_ds_nhwc: N Ho Ho a multiple of 196 with Ho Ho | 196, Cin and Cin2 multiples of 64 (sum: of 128), Cout of 256");
    HIP_TRY(H16(precision, ap_launch_conv_pw)(p, (hipStream_t)stream));
    return AP_OK;
}

int ap_conv_pw_k3s2_nhwc(int precision, const void* x, const void* wstream, const float* scale, const float* shift, void* y, int N,
                         int H, int Cin, int Cout, void* stream) {
    if (!prec_half(precision) || !x || !wstream || !scale || !shift || !y || N <= 0 || H <= 0)
        return fail(AP_EINVAL, "ap_conv_pw_k3s2_nhwc: bad argument (precision: AP_PREC_BF16 or AP_PREC_F16)");
    PwArgs p{};
    p.k3 = 1; p.Ho = p.Wo = H / 2; p.H2 = p.W2 = H; p.stride2 = 2; p.M = N * p.Ho * p.Wo; p.Cin = Cin; p.Cout = Cout; p.relu = 1;
    p.x = x; p.y = y; p.wfrag = wstream; p.scale = scale; p.shift = shift;
    if ((H & 1) || !k_bf16::ap_conv_pw_k3_supported(p))
        return fail(AP_ESHAPE, "ap_conv_pw_k3s2_nhwc: H even with (H / 2)^2 | 196 (14 or 28), N (H / 2)^2 a multiple of 196, Cin / 64 a power of two >= 2, Cout a multiple of 256");
    HIP_TRY(H16(precision, ap_launch_conv_pw)(p, (hipStream_t)stream));
    return AP_OK;
}

int64_t ap_block_img_stream_bytes(void) { return (int64_t)k_bf16::ap_block_img_stream_bytes(); }

int ap_block_img_pack(int precision, const void* w1, const void* w2, const void* w3, void* wstream, void* stream) {
    if (!prec_half(precision) || !w1 || !w2 || !w3 || !wstream)
        return fail(AP_EINVAL, "ap_block_img_pack: bad argument (precision: AP_PREC_BF16 or AP_PREC_F16)");
    HIP_TRY(H16(precision, ap_launch_block_img_pack)(w1, w2, w3, wstream, (hipStream_t)stream));
    return AP_OK;
}

int ap_block_img_nhwc(int precision, const void* x, const void* wstream, const float* s1, const float* h1, const float* s2,
                      const float* h2, const float* s3, const float* h3, void* y, int N, void* stream) {
    if (!prec_half(precision) || !x || !wstream || !s1 || !h1 || !s2 || !h2 || !s3 || !h3 || !y || N <= 0)
        return fail(AP_EINVAL, "ap_block_img_nhwc: bad argument (precision: AP_PREC_BF16 or AP_PREC_F16)");
    BlkImgArgs a{};
    a.x = x; a.y = y; a.wfrag = wstream; a.s1 = s1; a.h1 = h1; a.s2 = s2; a.h2 = h2; a.s3 = s3; a.h3 = h3; a.N = N;
    a.dbg = g_conv_dbg;
    HIP_TRY(H16(precision, ap_launch_block_img)(a, (hipStream_t)stream));
    return AP_OK;
}

int64_t ap_conv_img3_stream_bytes(void) { return (int64_t)k_bf16::ap_conv_img3_stream_bytes(); }

int64_t ap_conv_s2p_stream_bytes(void) { return (int64_t)k_bf16::ap_conv_s2p_stream_bytes(); }

int ap_conv_s2p_pack(int precision, const void* w2, void* wstream, void* stream) {
    if (!prec_half(precision) || !w2 || !wstream) return fail(AP_EINVAL, "ap_conv_s2p_pack: 16-bit precision, w2 [128][3][3][128], stream buffer");
    HIP_TRY(H16(precision, ap_launch_conv_s2p_pack)(w2, wstream, (hipStream_t)stream));
    return AP_OK;
}

int ap_conv_s2p_nhwc(int precision, const void* x, const void* wstream, const float* scale, const float* shift, void* y, int N,
                     int y_tiled, void* stream) {
    if (!prec_half(precision) || !x || !wstream || !scale || !shift || !y || N <= 0)
        return fail(AP_EINVAL, "ap_conv_s2p_nhwc: bad argument");
    ConvS2pArgs a{};
    a.x = x; a.y = y; a.wfrag = wstream; a.scale = scale; a.shift = shift; a.N = N; a.y_tiled = y_tiled != 0;
    HIP_TRY(zero_line(&a.zero));
    HIP_TRY(H16(precision, ap_launch_conv_s2p)(a, (hipStream_t)stream));
    return AP_OK;
}

int ap_conv_img3_pack(int precision, const void* w2, void* wstream, void* stream) {
    if (!prec_half(precision) || !w2 || !wstream) return fail(AP_EINVAL, "ap_conv_img3_pack: 16-bit precision, w2 [128][3][3][128], stream buffer");
    HIP_TRY(H16(precision, ap_launch_conv_img3_pack)(w2, wstream, (hipStream_t)stream));
    return AP_OK;
}

int ap_conv_img3_nhwc(int precision, const void* x, const void* wstream, const float* scale, const float* shift, void* y, int N,
                      int y_tiled, void* stream) {
    if (!prec_half(precision) || !x || !wstream || !scale || !shift || !y || N <= 0)
        return fail(AP_EINVAL, "ap_conv_img3_nhwc: bad argument");
    ConvImg3Args a{};
    a.x = x; a.y = y; a.wfrag = wstream; a.scale = scale; a.shift = shift; a.N = N; a.y_tiled = y_tiled != 0;
    HIP_TRY(zero_line(&a.zero));
    HIP_TRY(H16(precision, ap_launch_conv_img3)(a, (hipStream_t)stream));
    return AP_OK;
}

// The fused pair kernel consumes its two weight matrices as ONE stream of 16-KiB tiles in consumption order.  The stream is
// CALLER-OWNED: packed once by ap_conv_pair_pack into a buffer of ap_conv_pair_stream_bytes, handed to every launch -- the
// library keeps no hidden copy keyed by weight addresses (an allocator may reuse an address for new contents).
int64_t ap_conv_pair_stream_bytes(int P, int P2, int N1) {
    if (!k_bf16::ap_conv_pair_supported(P, P2, 4 * P, N1)) return AP_ESHAPE;
    return (int64_t)k_bf16::ap_conv_pair_stream_bytes(P, P2, 4 * P, N1);
}

int ap_conv_pair_pack(int precision, const void* w3, const void* w1, int P, int P2, int N1, void* wstream, void* stream) {
    if (!prec_half(precision) || !w3 || !wstream || (N1 > 0 && !w1))
        return fail(AP_EINVAL, "ap_conv_pair_pack: bad argument (precision: AP_PREC_BF16 or AP_PREC_F16)");
    if (!k_bf16::ap_conv_pair_supported(P, P2, 4 * P, N1))
        return fail(AP_ESHAPE, "ap_conv_pair_pack: (P, P2, N1) must be (128,0,128), (128,0,256), (256,0,256), (128,256,128) or (256,512,0)");
    HIP_TRY(H16(precision, ap_launch_pair_pack)(w3, N1 ? w1 : nullptr, wstream, P, P2, 4 * P, N1, (hipStream_t)stream));
    return AP_OK;
}

int ap_conv_pair_nhwc(int precision, const void* t2, const void* wstream, const float* s3, const float* h3, const void* res,
                      const float* s1, const float* h1, void* out, void* t1n, int M, int P, int N1, void* stream) {
    if (!prec_half(precision) || !t2 || !wstream || !s3 || !h3 || !res || !s1 || !h1 || !out || !t1n || M <= 0)
        return fail(AP_EINVAL, "ap_conv_pair_nhwc: bad argument (precision: AP_PREC_BF16 or AP_PREC_F16)");
    if (!k_bf16::ap_conv_pair_supported(P, 0, 4 * P, N1)) return fail(AP_ESHAPE, "ap_conv_pair_nhwc: (P, N1) must be (128,128), (128,256) or (256,256)");
    PairArgs a{};
    a.t2 = t2; a.res = res; a.wstream = wstream; a.s3 = s3; a.h3 = h3; a.s1 = s1; a.h1 = h1; a.out = out; a.t1n = t1n; a.M = M;
    a.dbg = g_conv_dbg;
    HIP_TRY(H16(precision, ap_launch_conv_pair)(a, P, 0, 4 * P, N1, (hipStream_t)stream));
    return AP_OK;
}

int ap_conv_pair_ds_nhwc(int precision, const void* t2, const void* x, const void* wstream, const float* s3, const float* h3,
                         const float* s1, const float* h1, void* out, void* t1n, int N, int Ho, int P, int P2, int stride, int N1,
                         void* stream) {
    if (!prec_half(precision) || !t2 || !x || !wstream || !s3 || !h3 || !out || N <= 0 || Ho <= 0 || (N1 > 0 && (!s1 || !h1 || !t1n)))
        return fail(AP_EINVAL, "ap_conv_pair_ds_nhwc: bad argument (precision: AP_PREC_BF16 or AP_PREC_F16)");
    const int C3 = 4 * P;
    if (!k_bf16::ap_conv_pair_supported(P, P2, C3, N1) || stride < 1 || stride > 2)
        return fail(AP_ESHAPE, "ap_conv_pair_ds_nhwc: (P, P2, N1) must be (128,256,128) or (256,512,0); stride 1 or 2");
    PairArgs a{};
    a.t2 = t2; a.x2 = x; a.wstream = wstream; a.s3 = s3; a.h3 = h3; a.s1 = s1; a.h1 = h1; a.out = out; a.t1n = t1n;
    a.M = N * Ho * Ho; a.Ho = a.Wo = Ho; a.H2 = a.W2 = Ho * stride; a.stride2 = stride;
    a.dbg = g_conv_dbg;
    HIP_TRY(H16(precision, ap_launch_conv_pair)(a, P, P2, C3, N1, (hipStream_t)stream));
    return AP_OK;
}

int ap_debug_set_trace(void* device_buf_160_u64) {
    g_conv_dbg = (unsigned long long*)device_buf_160_u64;
    return AP_OK;
}

int ap_set_conv_config(int cfg) {
    if (cfg != -1 && cfg != -4 && cfg != -5 && cfg != 100 && (cfg < 0 || cfg > 14) && cfg != 17)
        return fail(AP_EINVAL, "ap_set_conv_config: -1, -4, -5, 0..14, 17 or 100");
    g_conv_mode.store(cfg, std::memory_order_relaxed);
    return AP_OK;
}

}  // extern "C"

namespace {
// IEF of the HMR head from trunk features: state rows of 160 floats = pose132 | shape10 | cam3 | pad, left in ws_state
int hmr_ief(ap_net* h, const float* feat, int B, int iters, const float* init_theta, int theta_bs, const float* init_shape,
            int shape_bs, const float* init_cam, int cam_bs, hipStream_t st) {
    HIP_TRY(h->ws_H.reserve((size_t)B * DLD * 4));
    HIP_TRY(h->ws_D.reserve((size_t)B * DLD * 4));
    HIP_TRY(h->ws_state.reserve((size_t)B * 160 * 4));
    float *Hb = h->ws_H.as<float>(), *D = h->ws_D.as<float>(), *state = h->ws_state.as<float>();
    int rc;
    if ((rc = run_gemm(h->fold_feat, feat, 2048, 2048, B, Hb, DLD, nullptr, 0, st))) return rc;
    HIP_TRY(ap_launch_hmr_init(init_theta, theta_bs, init_shape, shape_bs, init_cam, cam_bs, h->mean_pose.as<float>(),
                               h->mean_shape.as<float>(), h->mean_cam.as<float>(), state, B, st));
    for (int it = 0; it < iters; ++it) {
        if ((rc = run_gemm(h->fold_state, state, 160, 160, B, D, DLD, Hb, DLD, st))) return rc;
        HIP_TRY(ap_launch_hmr_update(state, D, DLD, B, st));
    }
    return AP_OK;
}
}  // namespace

extern "C" {

// model_hmr.copenet.forward_reg (:160-172) from pre-computed features: `iters` evaluations, the raw 6-D pose / shape /
// camera state out (no rot6d conversion)
int ap_hmr_reg(ap_net* h, const float* xf, int B, int iters, const float* pose_in, int pose_bs, const float* shape_in,
               int shape_bs, const float* cam_in, int cam_bs, float* pose_out, float* shape_out, float* cam_out,
               void* stream) {
    if (!h || !xf || B <= 0 || iters < 1 || !pose_out || !shape_out || !cam_out)
        return fail(AP_EINVAL, "ap_hmr_reg: bad argument");
    if (h->variant != 1) return fail(AP_ESTATE, "ap_hmr_reg needs an hmr (variant 1) handle");
    hipStream_t st = (hipStream_t)stream;
    int rc = hmr_ief(h, xf, B, iters, pose_in, pose_bs, shape_in, shape_bs, cam_in, cam_bs, st);
    if (rc) return rc;
    const float* state = h->ws_state.as<float>();
    HIP_TRY(hipMemcpy2DAsync(pose_out, 132 * 4, state, 160 * 4, 132 * 4, B, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpy2DAsync(shape_out, 10 * 4, state + 132, 160 * 4, 10 * 4, B, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpy2DAsync(cam_out, 3 * 4, state + 142, 160 * 4, 3 * 4, B, hipMemcpyDeviceToDevice, st));
    return AP_OK;
}

int ap_hmr_fwd(ap_net* h, const float* x, int B, int iters, const float* init_theta, int theta_bs,
               const float* init_shape, int shape_bs, const float* init_cam, int cam_bs, float* rotmat, float* betas,
               float* cam, void* stream) {
    if (!h || !x || B <= 0 || iters < 1 || !rotmat || !betas || !cam) return fail(AP_EINVAL, "ap_hmr_fwd: bad argument");
    if (h->variant != 1) return fail(AP_ESTATE, "ap_hmr_fwd needs an hmr (variant 1) handle");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(h->ws_feat.reserve((size_t)B * 2048 * 4));
    float* feat = h->ws_feat.as<float>();
    int rc = trunk_fwd(h, x, B, nullptr, 0, feat, st);
    if (rc) return rc;
    if ((rc = hmr_ief(h, feat, B, iters, init_theta, theta_bs, init_shape, shape_bs, init_cam, cam_bs, st))) return rc;
    float* state = h->ws_state.as<float>();
    HIP_TRY(ap_launch_hmr_output(state, rotmat, betas, cam, B, st));
    return AP_OK;
}
int ap_net_enable_timing(ap_net* h, int on) {
    if (!h || on < 0 || on > 2) return fail(AP_EINVAL, "ap_net_enable_timing: handle, on in {0, 1, 2}");
    h->tm.on = on;
    return AP_OK;
}

int ap_net_timing(ap_net* h, double ms[4], int64_t* passes, int reset) {
    if (!h || !ms || !passes) return fail(AP_EINVAL, "ap_net_timing: null argument");
    HIP_TRY(h->tm.collect(ms, 4, passes, reset != 0));
    return AP_OK;
}

int ap_net_set_fuse_ds(ap_net* h, int on) {
    if (!h) return fail(AP_EINVAL, "null handle");
    h->fuse_ds = on != 0;
    return AP_OK;
}

int ap_net_set_fuse_block(ap_net* h, int on) {
    if (!h) return fail(AP_EINVAL, "null handle");
    h->fuse_block = on != 0;
    return AP_OK;
}

int ap_net_set_fuse_pair(ap_net* h, int on) {
    if (!h) return fail(AP_EINVAL, "null handle");
    h->fuse_pair = on != 0;
    return AP_OK;
}

int ap_net_set_fuse_tail(ap_net* h, int on) {
    if (!h) return fail(AP_EINVAL, "null handle");
    h->fuse_tail = on != 0;
    return AP_OK;
}

int ap_net_set_pw_conv(ap_net* h, int on) {
    if (!h) return fail(AP_EINVAL, "null handle");
    h->pw_conv = on < 0 ? 0 : (on > 4 ? 4 : on);               // (3: 1 without the size rule; 4: 1 without the 3 x 3 / stride-2 layers -- A/B aids)
    return AP_OK;
}

int ap_net_set_s2p(ap_net* h, int on) {
    if (!h) return fail(AP_EINVAL, "null handle");
    h->s2p = on != 0;
    return AP_OK;
}

int ap_net_set_img3(ap_net* h, int on) {
    if (!h) return fail(AP_EINVAL, "null handle");
    h->img3 = on < 0 ? 0 : (on > 2 ? 2 : on);
    return AP_OK;
}

int ap_net_set_img_block(ap_net* h, int on) {
    if (!h) return fail(AP_EINVAL, "null handle");
    h->img_block = on < 0 ? 0 : (on > 2 ? 2 : on);
    return AP_OK;
}

int ap_net_set_even_out(ap_net* h, int on) {
    if (!h) return fail(AP_EINVAL, "null handle");
    h->even_out = on != 0;
    return AP_OK;
}

int ap_net_set_fuse_ief(ap_net* h, int on) {
    if (!h) return fail(AP_EINVAL, "null handle");
    h->fuse_ief = on != 0;
    return AP_OK;
}

int ap_net_set_fuse_pool(ap_net* h, int on) {
    if (!h) return fail(AP_EINVAL, "null handle");
    h->fuse_pool = on != 0;
    return AP_OK;
}

int ap_net_set_tiled(ap_net* h, int on) {
    if (!h) return fail(AP_EINVAL, "null handle");
    h->tiled = on != 0;
    return AP_OK;
}

int ap_net_set_fuse_stem(ap_net* h, int on) {
    if (!h) return fail(AP_EINVAL, "null handle");
    h->fuse_stem = on == 2 ? 2 : on != 0;
    return AP_OK;
}

int ap_net_set_fold(ap_net* h, int on) {
    if (!h) return fail(AP_EINVAL, "null handle");
    if (on && h->fold_rejected) {
        // a remembered knob re-applied to a checkpoint the probe rejects (copenet._set_knob) must not turn every later call
        // into an error: the literal chain stays, ap_net_fold_status says why
        fprintf(stderr, "airpose_hip: ap_net_set_fold(1) ignored -- ap_net_finalize rejected the fold for this checkpoint "
                        "(ap_net_fold_status); the handle keeps evaluating the literal fc1 -> fc2 -> dec chain\n");
        return AP_OK;
    }
    h->fold = on != 0;
    return AP_OK;
}

int ap_net_set_fold_bar(ap_net* h, double bar) {
    if (!h || !(bar >= 0.0)) return fail(AP_EINVAL, "ap_net_set_fold_bar: handle, bar >= 0");
    h->fold_bar = bar;
    // the bar is applied by the regressor's finalize step (the fold probe): re-run THAT step when the handle holds a packed
    // checkpoint -- behind a device sync (passes in flight read the maps it rebuilds), and with the handle marked un-finalized
    // while it runs, so a failure leaves it in a state every later call reports
    if (h->finalized) {
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipDeviceSynchronize());
        h->finalized = false;
        const int rc = finalize_regressor(h);
        if (rc) return rc;
        h->finalized = true;
    }
    return AP_OK;
}

int ap_net_fold_status(const ap_net* h, double* probe_rel_err) {
    if (!h) return fail(AP_EINVAL, "null handle");
    if (probe_rel_err) *probe_rel_err = h->fold_check_err;
    return h->fold_rejected ? 0 : (h->fold ? 1 : 2);
}

int ap_net_set_dual_stream(ap_net* h, int on) {
    if (!h) return fail(AP_EINVAL, "null handle");
    h->dual_stream = on != 0;
    h->passes_per_view = on == 100 ? 2 : 1;                  // (tuning: 100 = four half passes)
    h->dual_skew = (on > 1 && on < 100) ? on - 1 : 0;        // (tuning: on = 1 + skew point)
    return AP_OK;
}

int ap_net_set_chunk(ap_net* h, int images_per_chunk) {
    if (!h || images_per_chunk < 0) return fail(AP_EINVAL, "ap_net_set_chunk: bad argument");
    // kernels address an activation tensor of one chunk with 32-bit byte offsets in places (folded downsample segment,
    // fused layer1 bottleneck): 1024 images keep every tensor of the pass below 4 GiB in both storage types
    if (images_per_chunk > 1024) return fail(AP_EINVAL, "ap_net_set_chunk: at most 1024 images per chunk");
    h->chunk = images_per_chunk;
    return AP_OK;
}

// ---------------------------------------------------------------------------------- SMPL-X
int ap_smplx_create(ap_smplx** out, const ap_smplx_model* md, int device) {
    if (!out || !md) return fail(AP_EINVAL, "ap_smplx_create: null argument");
    const int V = md->num_verts, J = md->num_joints, NS = md->num_shape_coeffs;
    if (V <= 0 || J <= 1 || J > 64 || NS != 20 || md->num_extra < 0 || md->num_landmarks < 0 ||
        J + md->num_extra + md->num_landmarks > 128)
        return fail(AP_ESHAPE, "ap_smplx_create: unsupported model dimensions");
    if (!md->v_template || !md->shapedirs || !md->posedirs || !md->J_regressor || !md->parents || !md->lbs_weights ||
        (md->num_landmarks && (!md->faces || !md->lmk_faces_idx || !md->lmk_bary_coords)) ||
        (md->num_extra && !md->extra_joint_verts))
        return fail(AP_EINVAL, "ap_smplx_create: null model array");
    HIP_TRY(hipSetDevice(device));
    ap_smplx* h = new ap_smplx();
    h->device = device;
    SmplxModelDev& m = h->m;
    m.V = V; m.J = J; m.ncoef = 512;
    const int NP = (J - 1) * 9;
    if (20 + NP > m.ncoef) { delete h; return fail(AP_ESHAPE, "too many pose features"); }
    // rest joints as an affine function of the shape coefficients (exact algebra, done in fp64):
    //   J = J_regressor (v_template + shapedirs c) = J_template + J_shapedirs c
    std::vector<float> jt((size_t)J * 3), jsd((size_t)J * 3 * 20);
    for (int j = 0; j < J; ++j) {
        double t[3] = {0, 0, 0};
        std::vector<double> sd(60, 0.0);
        for (int v = 0; v < V; ++v) {
            const double r = md->J_regressor[(size_t)j * V + v];
            if (r == 0.0) continue;
            for (int c = 0; c < 3; ++c) {
                t[c] += r * md->v_template[(size_t)v * 3 + c];
                for (int l = 0; l < 20; ++l) sd[c * 20 + l] += r * md->shapedirs[((size_t)v * 3 + c) * 20 + l];
            }
        }
        for (int c = 0; c < 3; ++c) {
            jt[j * 3 + c] = (float)t[c];
            for (int l = 0; l < 20; ++l) jsd[((size_t)j * 3 + c) * 20 + l] = (float)sd[c * 20 + l];
        }
    }
    // kinematic tree
    std::vector<int> par(J), dep(J, 0);
    int maxd = 0;
    for (int j = 0; j < J; ++j) {
        par[j] = j == 0 ? -1 : (int)md->parents[j];
        if (j > 0 && (par[j] < 0 || par[j] >= j)) { delete h; return fail(AP_ESHAPE, "parents must satisfy 0 <= parent < child"); }
        dep[j] = j == 0 ? 0 : dep[par[j]] + 1;
        maxd = std::max(maxd, dep[j]);
    }
    m.max_depth = maxd;
    // sparse skinning weights, ascending bone index, K = max non-zeros per vertex (4 / 8 / exact)
    int K = 1;
    for (int v = 0; v < V; ++v) {
        int nz = 0;
        for (int j = 0; j < J; ++j) nz += md->lbs_weights[(size_t)v * J + j] != 0.f;
        K = std::max(K, nz);
    }
    K = K <= 4 ? 4 : (K <= 8 ? 8 : K);
    m.K = K;
    std::vector<int> sidx((size_t)V * K, 0);
    std::vector<float> sw((size_t)V * K, 0.f);
    for (int v = 0; v < V; ++v) {
        int k = 0;
        for (int j = 0; j < J; ++j) {
            const float w = md->lbs_weights[(size_t)v * J + j];
            if (w != 0.f) { sidx[(size_t)v * K + k] = j; sw[(size_t)v * K + k] = w; ++k; }
        }
    }
    // blend-shape operand: row n = 3v+c, columns [20 shape/expr | (J-1)*9 pose | 0 pad]; shift = v_template
    Layer& L = h->dirs;
    const int rows = 3 * V;
    L.cin = m.ncoef; L.k = 1; L.stride = 1; L.pad = 0; L.wld = m.ncoef;
    L.cout = ((rows + 3) / 4) * 4;
    L.cout_pad = ((rows + 127) / 128) * 128;
    m.ldv = L.cout_pad;
    {
        std::vector<float> pk((size_t)L.cout_pad * L.wld, 0.f), scale(L.cout_pad, 1.f), shift(L.cout_pad, 0.f);
        for (int n = 0; n < rows; ++n) {
            float* dst = &pk[(size_t)n * L.wld];
            memcpy(dst, md->shapedirs + (size_t)n * 20, 20 * 4);
            shift[n] = md->v_template[n];
        }
        for (int p = 0; p < NP; ++p) {
            const float* src = md->posedirs + (size_t)p * rows;
            for (int n = 0; n < rows; ++n) pk[(size_t)n * L.wld + 20 + p] = src[n];
        }
        hipError_t e = upload(L.w, pk.data(), pk.size() * 4);
        if (e == hipSuccess) {
            std::vector<uint16_t> ps(2 * pk.size());          // rows of 512 coefficients: planar groups of 8
            host_split_pack_planar(pk.data(), pk.size(), ps.data());
            e = upload(h->dirs_split, ps.data(), pk.size() * 4);
        }
        if (e == hipSuccess) {
            // the same directions for the fused kernel: split-bf16 MFMA A fragments in register order, K = 224 (20 shape /
            // expression + the 21 body joints' 189 pose features; the 8th K step holds jaw / eye features, zero on that path):
            // block ((g*8 + ks)*3 + c)*2 + plane = 64 lanes x 8 bf16, lane (lr, g4) = row 3*(16 g + lr) + c,
            // coefficients 32 ks + 8 g4 .. + 7
            const int ng = (V + 15) / 16;
            std::vector<uint16_t> fr(ap_smplx_dirs_frag_bytes(V) / 2, 0);
            for (int g = 0; g < ng; ++g)
                for (int ks = 0; ks < 8; ++ks)
                    for (int c = 0; c < 3; ++c)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int v = g * 16 + (lane & 15), k0 = ks * 32 + (lane >> 4) * 8;
                            if (v >= V) continue;
                            uint16_t* hi = &fr[((((size_t)g * 8 + ks) * 3 + c) * 2) * 512 + lane * 8];
                            uint16_t* lo = hi + 512;
                            for (int i = 0; i < 8; ++i) host_split_parts(pk[(size_t)(3 * v + c) * L.wld + k0 + i], &hi[i], &lo[i]);
                        }
            e = upload(h->dirs_frag, fr.data(), fr.size() * 2);
        }
        if (e == hipSuccess) e = upload(L.scale, scale.data(), scale.size() * 4);
        if (e == hipSuccess) e = upload(L.shift, shift.data(), shift.size() * 4);
        if (e != hipSuccess) { ap_smplx_destroy(h); return fail((int)e, std::string("upload: ") + hipGetErrorString(e)); }
    }
    std::vector<int> ev(std::max(1, md->num_extra)), tri(std::max(1, md->num_landmarks * 3));
    for (int i = 0; i < md->num_extra; ++i) {
        ev[i] = (int)md->extra_joint_verts[i];
        if (ev[i] < 0 || ev[i] >= V) { ap_smplx_destroy(h); return fail(AP_ESHAPE, "extra joint vertex id out of range"); }
    }
    for (int l = 0; l < md->num_landmarks; ++l) {
        const int64_t f = md->lmk_faces_idx[l];
        if (f < 0 || f >= md->num_faces) { ap_smplx_destroy(h); return fail(AP_ESHAPE, "landmark face id out of range"); }
        for (int c = 0; c < 3; ++c) {
            tri[l * 3 + c] = (int)md->faces[f * 3 + c];
            if (tri[l * 3 + c] < 0 || tri[l * 3 + c] >= V) { ap_smplx_destroy(h); return fail(AP_ESHAPE, "face vertex id out of range"); }
        }
    }
    // distinct vertices the joints kernel skins (vertex picks + landmark corners): their v_posed goes to a compact side buffer
    std::vector<int> slot(V, -1);
    int n_jv = 0;
    for (int i = 0; i < md->num_extra; ++i) if (slot[ev[i]] < 0) slot[ev[i]] = n_jv++;
    for (int i = 0; i < md->num_landmarks * 3; ++i) if (slot[tri[i]] < 0) slot[tri[i]] = n_jv++;
    m.n_jv = std::max(n_jv, 1);
    hipError_t e = upload(h->j_template, jt.data(), jt.size() * 4);
    if (e == hipSuccess) e = upload(h->jv_slot, slot.data(), slot.size() * 4);
    if (e == hipSuccess && K == 4 && n_jv < 255) {           // fused kernel: bone indices as 6-bit fields + joint-vertex slot, weights padded to whole groups
        const int vp = (V + 15) / 16 * 16;
        std::vector<uint32_t> i8(vp, 0);
        std::vector<float> w4((size_t)vp * 4, 0.f);
        for (int v = 0; v < V; ++v) {
            for (int k = 0; k < 4; ++k) i8[v] |= (uint32_t)(sidx[(size_t)v * 4 + k] & 0x3f) << (6 * k);
            if (slot[v] >= 0 && slot[v] < 255) i8[v] |= (uint32_t)(slot[v] + 1) << 24;
            memcpy(&w4[(size_t)v * 4], &sw[(size_t)v * 4], 16);
        }
        e = upload(h->skin_idx8, i8.data(), i8.size() * 4);
        if (e == hipSuccess) e = upload(h->skin_w4, w4.data(), w4.size() * 4);
        if (e == hipSuccess) {
            // the joints kernel's record per output joint beyond the chain (21 vertex picks, 51 landmarks): its three corner
            // vertices' side-buffer slots, packed bone ids, weights and barycentric weights in 6 x 16 bytes -- one load level
            const int nj2 = md->num_extra + md->num_landmarks;
            std::vector<float> pk((size_t)std::max(nj2, 1) * 24, 0.f);
            for (int t = 0; t < nj2; ++t) {
                float* r = &pk[(size_t)t * 24];
                for (int f = 0; f < 3; ++f) {
                    const bool lm = t >= md->num_extra;
                    const int v = lm ? tri[(t - md->num_extra) * 3 + f] : ev[t];
                    const int sl = slot[v];
                    const uint32_t id = i8[v] & 0x00ffffffu;
                    memcpy(&r[f], &sl, 4);
                    memcpy(&r[4 + f], &id, 4);
                    memcpy(&r[8 + 4 * f], &w4[(size_t)v * 4], 16);
                    r[20 + f] = lm ? md->lmk_bary_coords[(t - md->num_extra) * 3 + f] : (f == 0 ? 1.f : 0.f);
                }
            }
            e = upload(h->jt_pack, pk.data(), pk.size() * 4);
        }
        // body-only table: every joint >= 22 (jaw, eyes, fingers: identity rotation without a hand / face pose) skins exactly like its
        // nearest ancestor < 22, so a vertex needs at most as many DISTINCT transforms as it has bones -- usually fewer (a finger
        // vertex: one).  Bones merged by representative, weights summed in double, heaviest first; unused slots repeat slot 0's bone
        // with weight 0 (the kernel skips zero weights).
        const int NBODY = 22;
        if (e == hipSuccess && J >= NBODY) {
            std::vector<int> rep(J);
            for (int j = 0; j < J; ++j) { int r = j; while (r >= NBODY) r = par[r]; rep[j] = r; }
            std::vector<uint32_t> i8b(vp, 0);
            std::vector<float> w4b((size_t)vp * 4, 0.f);
            for (int v = 0; v < V; ++v) {
                int bj[4]; double bw[4]; int nbn = 0;
                for (int k = 0; k < 4; ++k) {
                    const float w = sw[(size_t)v * 4 + k];
                    if (w == 0.f) continue;
                    const int r = rep[sidx[(size_t)v * 4 + k]];
                    int q = 0;
                    while (q < nbn && bj[q] != r) ++q;
                    if (q == nbn) { bj[nbn] = r; bw[nbn] = 0.0; ++nbn; }
                    bw[q] += (double)w;
                }
                for (int a2 = 0; a2 < nbn; ++a2)             // heaviest first (stable)
                    for (int b2 = a2 + 1; b2 < nbn; ++b2)
                        if (bw[b2] > bw[a2]) { std::swap(bw[a2], bw[b2]); std::swap(bj[a2], bj[b2]); }
                if (nbn == 0) { bj[0] = 0; bw[0] = 0.0; nbn = 1; }
                for (int k = 0; k < 4; ++k) {
                    i8b[v] |= (uint32_t)((k < nbn ? bj[k] : bj[0]) & 0x3f) << (6 * k);
                    w4b[(size_t)v * 4 + k] = k < nbn ? (float)bw[k] : 0.f;
                }
                i8b[v] |= i8[v] & 0xff000000u;
            }
            e = upload(h->skin_idx8b, i8b.data(), i8b.size() * 4);
            if (e == hipSuccess) e = upload(h->skin_w4b, w4b.data(), w4b.size() * 4);
            if (e == hipSuccess) m.nb = NBODY;
        }
    }
    if (e == hipSuccess) e = upload(h->j_shapedirs, jsd.data(), jsd.size() * 4);
    if (e == hipSuccess) e = upload(h->parents, par.data(), par.size() * 4);
    if (e == hipSuccess) e = upload(h->depth, dep.data(), dep.size() * 4);
    if (e == hipSuccess) e = upload(h->skin_idx, sidx.data(), sidx.size() * 4);
    if (e == hipSuccess) e = upload(h->skin_w, sw.data(), sw.size() * 4);
    if (e == hipSuccess) e = upload(h->extra_verts, ev.data(), ev.size() * 4);
    if (e == hipSuccess) e = upload(h->lmk_tri, tri.data(), tri.size() * 4);
    if (e == hipSuccess)
        e = upload(h->lmk_bary, md->num_landmarks ? (const void*)md->lmk_bary_coords : (const void*)tri.data(),
                   std::max(1, md->num_landmarks * 3) * 4);
    if (e != hipSuccess) { ap_smplx_destroy(h); return fail((int)e, std::string("upload: ") + hipGetErrorString(e)); }
    m.j_template = h->j_template.as<float>(); m.j_shapedirs = h->j_shapedirs.as<float>();
    m.parents = h->parents.as<int>(); m.depth = h->depth.as<int>();
    m.skin_idx = h->skin_idx.as<int>(); m.skin_w = h->skin_w.as<float>();
    m.extra_verts = h->extra_verts.as<int>(); m.lmk_tri = h->lmk_tri.as<int>(); m.lmk_bary = h->lmk_bary.as<float>();
    m.n_extra = md->num_extra; m.n_lmk = md->num_landmarks;
    m.dirs_frag = h->dirs_frag.p; m.v_template = h->dirs.shift.as<float>(); m.jv_slot = h->jv_slot.as<int>();
    m.skin_idx8 = h->skin_idx8.as<uint32_t>(); m.skin_w4 = h->skin_w4.as<float>();
    m.skin_idx8b = h->skin_idx8b.as<uint32_t>(); m.skin_w4b = h->skin_w4b.as<float>();
    m.jt_pack = h->jt_pack.as<float4>();
    h->n_out_joints = J + md->num_extra + md->num_landmarks;
    *out = h;
    return AP_OK;
}

void ap_smplx_destroy(ap_smplx* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for (DevBuf* b : {&h->dirs.w, &h->dirs_split, &h->dirs.scale, &h->dirs.shift, &h->j_template, &h->j_shapedirs, &h->parents, &h->depth,
                      &h->skin_idx, &h->skin_w, &h->extra_verts, &h->lmk_tri, &h->lmk_bary, &h->ws_coef, &h->ws_A, &h->ws_A22, &h->dirs_frag, &h->jv_slot, &h->skin_idx8, &h->skin_w4, &h->skin_idx8b, &h->skin_w4b, &h->jt_pack, &h->ws_side,
                      &h->ws_jposed, &h->ws_post, &h->ws_vposed, &h->ws_cc, &h->ws_cnt})
        b->release();
    h->tm.destroy();
    delete h;
}

int ap_smplx_num_joints_out(const ap_smplx* h) { return h ? h->n_out_joints : AP_EINVAL; }

}  // extern "C"

namespace {
int smplx_run(ap_smplx* h, SmplxFwdArgs a, bool body_only, hipStream_t st) {
    h->m.coef_split = h->blend_split ? 1 : 0;
    const SmplxModelDev& m = h->m;
    const int n = a.n;
    HIP_TRY(h->ws_coef.reserve((size_t)n * m.ncoef * 4));
    HIP_TRY(h->ws_A.reserve((size_t)n * m.J * 12 * 4));
    HIP_TRY(h->ws_jposed.reserve((size_t)n * m.J * 3 * 4));
    HIP_TRY(h->ws_post.reserve((size_t)n * 12 * 4));
    // body-only pose feature, 4 bones per vertex, split-bf16 coefficients: contraction + skinning in ONE kernel (v_posed stays
    // on the chip); anything else (hand / face poses: K = 512, more bones per vertex, the fp32 contraction) takes the two kernels
    const bool fused = h->fused && h->blend_split && body_only && ap_smplx_lbs_fused_supported(m);
    if (fused) HIP_TRY(h->ws_side.reserve((size_t)n * m.n_jv * 3 * 4));
    else HIP_TRY(h->ws_vposed.reserve((size_t)n * m.ldv * 4));
    a.dbg = g_conv_dbg;
    a.grp_cnt = nullptr;
    if (fused && h->fuse_joints) {
        const size_t need = (size_t)((n + 31) / 32) * 4;
        if (need > h->ws_cnt.bytes) {                        // (re)allocated: zero once; every launch leaves the counters at zero
            HIP_TRY(h->ws_cnt.reserve(need < 4096 ? 4096 : need));
            HIP_TRY(hipMemsetAsync(h->ws_cnt.p, 0, h->ws_cnt.bytes, st));
        }
        a.grp_cnt = h->ws_cnt.as<int>();
    }
    a.coef = h->ws_coef.as<float>(); a.A = h->ws_A.as<float>(); a.jposed = h->ws_jposed.as<float>();
    a.post = (a.pose6d || a.post_rt) ? h->ws_post.as<float>() : nullptr;
    a.A22 = nullptr;
    if (fused && h->fold_post && h->merge_bones && a.post && !a.transl && !a.grp_cnt && m.nb == 22) {
        HIP_TRY(h->ws_A22.reserve((size_t)n * 22 * 12 * 4));
        a.A22 = h->ws_A22.as<float>();
    }
    a.vposed = h->ws_vposed.as<float>();
    a.vp_side = fused ? h->ws_side.as<float>() : nullptr;
    if (a.n_main > 0 && a.intr0) {                           // camera centres resolved by the prep kernel
        HIP_TRY(h->ws_cc.reserve((size_t)a.n_main * 2 * 4));
        a.cc_ws = h->ws_cc.as<float>();
        a.cam_center = a.cc_ws;
    }
    size_t ev[5] = {0, 0, 0, 0, 0};
    // tm.on == 1: an event between every two kernels (per-stage times; each record costs a bubble of several microseconds on the
    // stream); 2: one event in front of the first kernel and one behind the last (the tail's span, no bubbles inside: reported in slot 0)
    const bool stages = h->tm.on == 1;
    if (h->tm.on) HIP_TRY(h->tm.rec(st, &ev[0]));
    HIP_TRY(ap_launch_smplx_prep(m, a, st));
    if (stages) HIP_TRY(h->tm.rec(st, &ev[1]));
    if (fused) {
        int n_cu = 0;
        HIP_TRY(device_cus(&n_cu));
        HIP_TRY(ap_launch_smplx_lbs_fused(m, a, n_cu, h->merge_bones, st));
        if (stages) { HIP_TRY(h->tm.rec(st, &ev[2])); ev[3] = ev[2]; }        // stage 1 = the fused kernel, stage 2 empty
    } else {
        // v_posed = v_template + [betas | expr | pose_feature] . dirs^T; hand/face rows of the pose feature are
        // identically zero when no extra pose is supplied, so the contraction stops after the 21 body joints
        int K = body_only ? 20 + 21 * 9 : 20 + (m.J - 1) * 9;
        K = ((K + 31) / 32) * 32;
        ConvArgs g{};
        g.x = a.coef; g.w = h->blend_split ? h->dirs_split.p : h->dirs.w.p;
        g.scale = h->dirs.scale.as<float>(); g.shift = h->dirs.shift.as<float>();
        g.out_f32 = 1;                                       // v_posed stays fp32 (only the split kind reads the flag)
        g.res = nullptr; g.y = h->ws_vposed.p;
        g.N = n; g.H = g.W = g.Ho = g.Wo = 1; g.Cin = K; g.Cout = h->dirs.cout; g.KH = g.KW = 1; g.stride = 1; g.pad = 0;
        g.M = n; g.ldx = m.ncoef; g.ldy = m.ldv; g.ldr = 0; g.wld = h->dirs.wld; g.relu = 0;
        HIP_TRY(dispatch_conv(g, h->blend_split ? AP_PREC_BF16X2 : AP_PREC_FP32, st));
        if (stages) HIP_TRY(h->tm.rec(st, &ev[2]));
        HIP_TRY(ap_launch_smplx_skin(m, a, st));
        if (stages) HIP_TRY(h->tm.rec(st, &ev[3]));
    }
    if (!a.grp_cnt) HIP_TRY(ap_launch_smplx_joints(m, a, st));
    if (h->tm.on) {
        HIP_TRY(h->tm.rec(st, &ev[4]));
        if (stages) for (int s = 0; s < 4; ++s) { h->tm.marks[s].push_back(ev[s]); h->tm.marks[s].push_back(ev[s + 1]); }
        else { h->tm.marks[0].push_back(ev[0]); h->tm.marks[0].push_back(ev[4]); }
        h->tm.passes++;
    }
    return AP_OK;
}
}  // namespace

extern "C" {

int ap_smplx_fwd(ap_smplx* h, int n, const float* betas, const float* expression, const float* global_orient,
                 const float* body_pose, const float* extra_pose, const float* transl, float* vertices,
                 float* joints, void* stream) {
    if (!h || n <= 0 || !betas || !body_pose || !vertices || !joints) return fail(AP_EINVAL, "ap_smplx_fwd: bad argument");
    SmplxFwdArgs a{};
    a.n = n; a.betas = betas; a.expression = expression; a.global_orient = global_orient; a.body_pose = body_pose;
    a.extra_pose = extra_pose; a.transl = transl; a.vertices = vertices; a.joints = joints;
    return smplx_run(h, a, extra_pose == nullptr, (hipStream_t)stream);
}

int ap_smplx_fwd_fused(ap_smplx* h, int n, const float* pred_pose, int pose_ld, const float* betas,
                       const float* cam_center, float fx, float fy, float* vertices_cam, float* joints_cam,
                       float* joints2d, float* rotmat, void* stream) {
    if (!h || n <= 0 || !pred_pose || pose_ld < 135 || !betas || !vertices_cam || !joints_cam)
        return fail(AP_EINVAL, "ap_smplx_fwd_fused: bad argument");
    SmplxFwdArgs a{};
    a.n = n; a.betas = betas;
    a.pose6d = pred_pose + 3; a.pose6d_ld = pose_ld;
    a.post_t = pred_pose; a.post_t_ld = pose_ld;
    a.cam_center = cam_center; a.fx = fx; a.fy = fy;
    a.vertices = vertices_cam; a.joints = joints_cam; a.joints2d = cam_center ? joints2d : nullptr;
    a.rotmat_out = rotmat;
    return smplx_run(h, a, true, (hipStream_t)stream);
}

int ap_smplx_fwd_twoview(ap_smplx* h, int B, float* pred_pose, int pose_ld, float trans_scale, const float* betas,
                         const float* intr0, const float* intr1, float fx, float fy, const float* in_smpltrans,
                         float* vertices, float* joints_cam, float* joints2d, float* rotmat, void* stream) {
    if (!h || B <= 0 || !pred_pose || pose_ld < 135 || !betas || !vertices || !joints_cam || (!intr0) != (!intr1) ||
        trans_scale < 0.f)
        return fail(AP_EINVAL, "ap_smplx_fwd_twoview: bad argument");
    SmplxFwdArgs a{};
    a.n_main = 2 * B;
    a.n = in_smpltrans ? 4 * B : 2 * B;
    a.in_trans = in_smpltrans;
    a.betas = betas;
    a.pose6d = pred_pose + 3; a.pose6d_ld = pose_ld;
    a.post_t = pred_pose; a.post_t_ld = pose_ld;
    if (trans_scale > 0.f) { a.pose_rw = pred_pose; a.trans_scale = trans_scale; }
    a.intr0 = intr0; a.intr1 = intr1; a.fx = fx; a.fy = fy;
    a.vertices = vertices; a.joints = joints_cam; a.joints2d = intr0 ? joints2d : nullptr;
    a.rotmat_out = rotmat;
    return smplx_run(h, a, true, (hipStream_t)stream);
}

int ap_smplx_set_blend_precision(ap_smplx* h, int precision) {
    if (!h || (precision != AP_PREC_FP32 && precision != AP_PREC_BF16X2))
        return fail(AP_EINVAL, "ap_smplx_set_blend_precision: AP_PREC_FP32 or AP_PREC_BF16X2");
    h->blend_split = precision == AP_PREC_BF16X2;
    return AP_OK;
}

int ap_smplx_debug_poison_workspace(ap_smplx* h, int n) {
    if (!h || n <= 0) return fail(AP_EINVAL, "ap_smplx_debug_poison_workspace: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(h->ws_coef.reserve((size_t)n * h->m.ncoef * 4));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(h->ws_coef.p, 0xFF, h->ws_coef.bytes));           // NaN bit patterns in both storage forms
    return AP_OK;
}

int ap_smplx_set_fused(ap_smplx* h, int on) {
    if (!h) return fail(AP_EINVAL, "null handle");
    h->fused = on != 0;
    h->fuse_joints = on == 4;                                // 4: joints stage inside the kernel, done by each group's last workgroup (A/B: slower)
    h->merge_bones = on == 6 ? 0 : (on == 8 ? 2 : 1);        // 6: every joint's transform in LDS (round 5's form); 8: merged table, 64 bodies per workgroup (A/B: slower)
    h->fold_post = on != 7;                                  // 7: merged table, post transform applied per vertex instead of composed into the bones (A/B)
    return AP_OK;
}

int ap_smplx_enable_timing(ap_smplx* h, int on) {
    if (!h) return fail(AP_EINVAL, "null handle");
    h->tm.on = on < 0 ? 0 : (on > 2 ? 2 : on);               // 1: per-stage events; 2: the span of the whole tail (two events)
    return AP_OK;
}

int ap_smplx_timing(ap_smplx* h, double ms[4], int64_t* passes, int reset) {
    if (!h || !ms || !passes) return fail(AP_EINVAL, "ap_smplx_timing: null argument");
    HIP_TRY(h->tm.collect(ms, 4, passes, reset != 0));
    return AP_OK;
}

int ap_rotmat_to_angle_axis(const float* rotmat, int n, int cols, float* angle_axis, void* stream) {
    if (!rotmat || !angle_axis || n <= 0 || (cols != 3 && cols != 4))
        return fail(AP_EINVAL, "ap_rotmat_to_angle_axis: bad argument (cols must be 3 or 4)");
    HIP_TRY(ap_launch_rotmat_to_angle_axis(rotmat, n, cols, angle_axis, (hipStream_t)stream));
    return AP_OK;
}

int ap_batch_rodrigues(const float* angle_axis, int n, int variant, float* rotmat, void* stream) {
    if (!angle_axis || !rotmat || n <= 0 || (variant != 0 && variant != 1))
        return fail(AP_EINVAL, "ap_batch_rodrigues: bad argument (variant 0 = smplx lbs, 1 = copenet geometry)");
    HIP_TRY(ap_launch_batch_rodrigues(angle_axis, n, variant, rotmat, (hipStream_t)stream));
    return AP_OK;
}

int ap_rot6d_to_rotmat(const float* x6, int n, float* rotmat, void* stream) {
    if (!x6 || !rotmat || n <= 0) return fail(AP_EINVAL, "ap_rot6d_to_rotmat: bad argument");
    HIP_TRY(ap_launch_rot6d(x6, n, rotmat, (hipStream_t)stream));
    return AP_OK;
}

int ap_transform_points(const float* rt, const float* pts, int B, int P, float* out, void* stream) {
    if (!rt || !pts || !out || B <= 0 || P <= 0) return fail(AP_EINVAL, "ap_transform_points: bad argument");
    HIP_TRY(ap_launch_transform_points(rt, pts, B, P, out, (hipStream_t)stream));
    return AP_OK;
}

int ap_preprocess_crops(const unsigned char* frames, int64_t frame_stride_bytes, int n, int H, int W, int bgr,
                        const int* crop_y0y1x0x1, float* out_nchw, float* scale_out, int* pad_left_top_out, void* stream) {
    if (!frames || !crop_y0y1x0x1 || !out_nchw || !scale_out || !pad_left_top_out || n <= 0 || H <= 0 || W <= 0 ||
        frame_stride_bytes < 0)
        return fail(AP_EINVAL, "ap_preprocess_crops: bad argument");
    HIP_TRY(k_bf16::ap_launch_preprocess(frames, (size_t)frame_stride_bytes, n, H, W, bgr, crop_y0y1x0x1, out_nchw, scale_out,
                                 pad_left_top_out, (hipStream_t)stream));
    return AP_OK;
}

// ---------------------------------------------------------------------------------- AirPose+ fitting loop
int ap_fit_create(ap_fit** out, const ap_smplx* body, const float* w1, const float* b1, const float* w2, const float* b2,
                  const float* w3, const float* b3, int device) {
    if (!out || !body || !w1 || !b1 || !w2 || !b2 || !w3 || !b3) return fail(AP_EINVAL, "ap_fit_create: null argument");
    HIP_TRY(hipSetDevice(device));
    ap_fit* h = new ap_fit();
    h->device = device; h->body = body;
    auto tr = [](const float* w, int o, int i, int ldo) {      // [o][i] -> [i][ldo] (k-major, zero padded columns)
        std::vector<float> t((size_t)i * ldo, 0.f);
        for (int a = 0; a < o; ++a) for (int b = 0; b < i; ++b) t[(size_t)b * ldo + a] = w[(size_t)a * i + b];
        return t;
    };
    hipError_t e = hipSuccess;
    auto up = [&](DevBuf& b, const std::vector<float>& v) { if (e == hipSuccess) e = upload(b, v.data(), v.size() * 4); };
    up(h->w1t, tr(w1, 512, 32, 512)); up(h->w2t, tr(w2, 512, 512, 512)); up(h->w3t, tr(w3, 126, 512, 128));
    up(h->w1, std::vector<float>(w1, w1 + 512 * 32)); up(h->w2, std::vector<float>(w2, w2 + 512 * 512));
    up(h->w3, std::vector<float>(w3, w3 + 126 * 512));
    std::vector<float> b3p(128, 0.f);
    memcpy(b3p.data(), b3, 126 * 4);
    up(h->b1, std::vector<float>(b1, b1 + 512)); up(h->b2, std::vector<float>(b2, b2 + 512)); up(h->b3, b3p);
    if (e != hipSuccess) { delete h; return fail((int)e, std::string("ap_fit_create: ") + hipGetErrorString(e)); }
    *out = h;
    return AP_OK;
}

void ap_fit_destroy(ap_fit* h) {
    if (!h) return;
    for (DevBuf* b : {&h->w1t, &h->w2t, &h->w3t, &h->w1, &h->w2, &h->w3, &h->b1, &h->b2, &h->b3, &h->H1, &h->H2, &h->O, &h->dO,
                      &h->dH2, &h->dH1, &h->dz, &h->aa, &h->dphi, &h->dtau, &h->dbeta, &h->loss, &h->adam_m, &h->adam_v, &h->robust})
        b->release();
    delete h;
}

int ap_fit_run(ap_fit* h, int L, float* z, float* phi, float* tau, float* beta, const float* j2d, const int* robust_host,
               const float* intr, const float* extr, int first_iter, int n_iters, int switch_iter, float lr, float sigma,
               float w_vposer, float w_temporal, float* loss_hist, float* grad_out, void* stream) {
    if (!h || !z || !phi || !tau || !beta || !j2d || !robust_host || !intr || !extr || L <= 0 || n_iters < 0 || first_iter < 0)
        return fail(AP_EINVAL, "ap_fit_run: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int nprm = L * 32 + 2 * L * 9 + 10;
    HIP_TRY(h->H1.reserve((size_t)L * 512 * 4)); HIP_TRY(h->H2.reserve((size_t)L * 512 * 4));
    HIP_TRY(h->O.reserve((size_t)L * 128 * 4)); HIP_TRY(h->dO.reserve((size_t)L * 128 * 4));
    HIP_TRY(h->dH2.reserve((size_t)L * 512 * 4)); HIP_TRY(h->dH1.reserve((size_t)L * 512 * 4));
    HIP_TRY(h->dz.reserve((size_t)L * 32 * 4)); HIP_TRY(h->aa.reserve((size_t)L * 63 * 4));
    HIP_TRY(h->dphi.reserve((size_t)2 * L * 6 * 4)); HIP_TRY(h->dtau.reserve((size_t)2 * L * 3 * 4));
    HIP_TRY(h->dbeta.reserve((size_t)L * 10 * 4)); HIP_TRY(h->loss.reserve((size_t)L * 4 * 4));
    HIP_TRY(h->adam_m.reserve((size_t)nprm * 4)); HIP_TRY(h->adam_v.reserve((size_t)nprm * 4));
    HIP_TRY(h->robust.reserve((size_t)L * 4));
    HIP_TRY(hipMemcpyAsync(h->robust.p, robust_host, (size_t)L * 4, hipMemcpyHostToDevice, st));
    FitArgs a{};
    a.L = L;
    a.O = h->O.as<float>(); a.dO = h->dO.as<float>(); a.ldo = 128; a.aa_all = h->aa.as<float>();
    a.z = z; a.phi = phi; a.tau = tau; a.beta = beta; a.dz = h->dz.as<float>(); a.ldz = 32;
    a.dphi = h->dphi.as<float>(); a.dtau = h->dtau.as<float>(); a.dbeta_part = h->dbeta.as<float>();
    a.loss_part = h->loss.as<float>();
    a.j_template = h->body->m.j_template; a.j_shapedirs = h->body->m.j_shapedirs; a.jsd_ld = 20;
    a.j2d = j2d; a.robust = h->robust.as<int>(); a.intr = intr; a.extr = extr;
    for (int f = 0; f < L; ++f) {
        a.n_robust += robust_host[f] != 0;
        if (f + 1 < L) a.n_pairs += robust_host[f] != 0 && robust_host[f + 1] != 0;
    }
    a.sigma = sigma; a.w_temporal = w_temporal; a.w_vposer = w_vposer; a.lr = lr;
    a.adam_m = h->adam_m.as<float>(); a.adam_v = h->adam_v.as<float>(); a.grad_out = grad_out;
    auto decode = [&]() -> hipError_t {                     // VPoser decoder MLP + matrot2aa, one launch
        return ap_launch_fit_decode(z, L, h->w1t.as<float>(), h->b1.as<float>(), h->w2t.as<float>(), h->b2.as<float>(),
                                    h->w3t.as<float>(), h->b3.as<float>(), h->H1.as<float>(), h->H2.as<float>(),
                                    h->O.as<float>(), h->aa.as<float>(), st);
    };
    bool decoded = false;
    for (int j = first_iter; j < first_iter + n_iters; ++j) {
        const bool with_z = j >= switch_iter;
        if (j == first_iter || j == switch_iter) {          // a new torch.optim.Adam starts with empty state (:279-295)
            HIP_TRY(hipMemsetAsync(h->adam_m.p, 0, (size_t)nprm * 4, st));
            HIP_TRY(hipMemsetAsync(h->adam_v.p, 0, (size_t)nprm * 4, st));
        }
        if (with_z || !decoded) { HIP_TRY(decode()); decoded = true; }      // z is constant before the switch
        if (loss_hist) a.loss_part = loss_hist + (size_t)(j - first_iter) * L * 4;      // the frame kernel writes the history row
        HIP_TRY(ap_launch_fit_frame(a, j, st));
        const int step = j < switch_iter ? j - first_iter + 1 : j - std::max(switch_iter, first_iter) + 1;
        if (with_z)                                          // decoder backward dO -> dz and Adam on everything, one launch
            HIP_TRY(ap_launch_fit_backprop_adam(a, h->w3.as<float>(), h->w2.as<float>(), h->w1.as<float>(), h->H1.as<float>(),
                                                h->H2.as<float>(), step, st));
        else
            HIP_TRY(ap_launch_fit_adam(a, step, 0, st));
    }
    return AP_OK;
}

int ap_perspective_projection(const float* pts, int B, int P, const float* rotation, const float* translation,
                              float fx, float fy, const float* center, float* out, void* stream) {
    if (!pts || !center || !out || B <= 0 || P <= 0) return fail(AP_EINVAL, "ap_perspective_projection: bad argument");
    HIP_TRY(ap_launch_projection(pts, B, P, rotation, translation, fx, fy, center, out, (hipStream_t)stream));
    return AP_OK;
}

}  // extern "C"
