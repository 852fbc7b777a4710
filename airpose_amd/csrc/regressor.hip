// Glue kernels of the iterative (IEF) regressor with cross-view fusion, fp32.
// Reference: copenet/src/copenet/models/model_copenet.py:112-159 (forward), :178-204 (forward_reg).
//
// The three Linear layers run on the fp32 MFMA GEMM (conv_igemm.hip); fc1 is split by linearity into
// the 2048 trunk-feature columns (constant over the IEF iterations: computed once) and the 284
// state columns [bb | pos | orient | art | shape | art_other | shape_other] (model_copenet.py:185,192),
// which these kernels assemble.  Rows 0..B-1 are view 0, rows B..2B-1 view 1; the cross-view swap is
// the partner-row read below (or, in view-split mode, an exchanged `partner` buffer).
#include "ap_common.h"
#include "kernels.h"

namespace {

constexpr int ST = 148;    // state row: pos3 | orient6 | art126 | shape10 | pad3
constexpr int SLD = 288;   // assembled fc1 state-input row (284 + 4 zero pad)

__global__ void reg_init_kernel(const RegInitArgs a) {
    const int row = blockIdx.x, v = row >= a.B, b = v ? row - a.B : row;
    const float* pos = (v ? a.pos1 : a.pos0) + (size_t)b * a.pos_bs;
    const float* theta = v ? a.theta1 : a.theta0;
    const float* shape = v ? a.shape1 : a.shape0;
    const float* th = theta ? theta + (size_t)b * (v ? a.theta1_bs : a.theta0_bs) : a.mean_pose;
    const float* sh = shape ? shape + (size_t)b * (v ? a.shape1_bs : a.shape0_bs) : a.mean_shape;
    float* s = a.state + (size_t)row * ST;
    for (int i = threadIdx.x; i < ST; i += blockDim.x) {
        float val = 0.f;
        if (i < 3) val = pos[i];
        else if (i < 135) val = th[i - 3];          // orient = theta[:6], art = theta[6:132]
        else if (i < 145) val = sh[i - 135];
        s[i] = val;
    }
}

__global__ void reg_update_assemble_kernel(float* __restrict__ state, const float* __restrict__ delta, int ldd,
                                           const float* __restrict__ bb0, const float* __restrict__ bb1,
                                           const float* __restrict__ partner, int partner_ld,
                                           float* __restrict__ S, int B, int two_view) {
    // one block per sample; handles both views of the pair so the swap needs no second pass
    const int b = blockIdx.x, nv = two_view ? 2 : 1;
    __shared__ float st[2][ST];
    for (int i = threadIdx.x; i < nv * ST; i += blockDim.x) {
        const int v = i / ST, e = i - v * ST;
        const size_t row = (size_t)v * B + b;
        float val = state[row * ST + e];
        if (delta && e < 145) {
            val += delta[row * ldd + e];
            state[row * ST + e] = val;
        }
        st[v][e] = val;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nv * SLD; i += blockDim.x) {
        const int v = i / SLD, e = i - v * SLD;
        const size_t row = (size_t)v * B + b;
        float val = 0.f;
        if (e < 3) val = (v ? bb1 : bb0)[(size_t)b * 3 + e];
        else if (e < 148) val = st[v][e - 3];                      // pos, orient, art, shape (state[0:145])
        else if (e < 284) {
            const int k = e - 148;                                 // partner art (126) | shape (10)
            if (two_view) val = st[1 - v][k < 126 ? 9 + k : 135 + (k - 126)];
            else val = partner ? partner[(size_t)b * partner_ld + k] : 0.f;   // (no partner: single-view model)
        }
        S[row * SLD + e] = val;
    }
}

__global__ void reg_output_kernel(const float* __restrict__ state, float* __restrict__ pose0,
                                  float* __restrict__ betas0, float* __restrict__ pose1, float* __restrict__ betas1,
                                  int B, int two_view) {
    const int row = blockIdx.x, v = row >= B, b = v ? row - B : row;
    const float* s = state + (size_t)row * ST;
    float* pose = (v ? pose1 : pose0) + (size_t)b * 135;
    float* betas = (v ? betas1 : betas0) + (size_t)b * 10;
    for (int i = threadIdx.x; i < 145; i += blockDim.x) {
        if (i < 135) pose[i] = s[i];
        else betas[i - 135] = s[i];
    }
}

// ---- folded IEF fast path ------------------------------------------------------------------------------------------
// forward_reg is one affine map (api.hip: Wf = Wd W2 W1, 145 x 2332), so a whole IEF forward is
//   H[row]   = bf + Wf[:, :2048] xf[row]                                  (once; split-K, reg_feat_splitk_kernel)
//   state   += H + Wf[:, 2048:] [bb | state | partner's art, shape]       (iters times; reg_fold_ief_kernel)
// and the cross-view swap only couples the two views of ONE pair: a workgroup owns a pair and runs all the iterations,
// initialisation (model_copenet.py:119-137) and the pose / betas split without leaving the kernel.  Weights are
// k-major ([k][148], coalesced over the output index) and stay in L2.
constexpr int OLD = 148;   // padded output count (145)
constexpr int KSPLIT = 8, KCH = 2048 / KSPLIT, FROWS = 8;

constexpr int KQ = 4;      // k-quarters of a thread block: thread = (output o, quarter q), partial sums meet in LDS

__global__ void __launch_bounds__(OLD * KQ) reg_feat_splitk_kernel(const float* __restrict__ xf0, const float* __restrict__ xf1,
                                                                   int B, int rows, const float* __restrict__ wt,
                                                                   float* __restrict__ part) {
    __shared__ float xs[FROWS][KCH];
    __shared__ float red[KQ][FROWS][OLD];
    const int r0 = blockIdx.x * FROWS, ks = blockIdx.y, o = threadIdx.x % OLD, q = threadIdx.x / OLD;
    for (int i = threadIdx.x; i < FROWS * KCH; i += blockDim.x) {
        const int r = i / KCH, k = i - r * KCH, row = r0 + r;
        float v = 0.f;
        if (row < rows) v = (row < B ? xf0 + (size_t)row * 2048 : xf1 + (size_t)(row - B) * 2048)[ks * KCH + k];
        xs[r][k] = v;
    }
    __syncthreads();
    float acc[FROWS];
#pragma unroll
    for (int r = 0; r < FROWS; ++r) acc[r] = 0.f;
    constexpr int KPT = KCH / KQ;                            // 64 k per thread
    const float* w = wt + ((size_t)ks * KCH + q * KPT) * OLD + o;
    for (int k0 = 0; k0 < KPT; k0 += 8) {
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = w[(size_t)(k0 + u) * OLD];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int r = 0; r < FROWS; ++r) acc[r] = fmaf(wv[u], xs[r][q * KPT + k0 + u], acc[r]);
    }
#pragma unroll
    for (int r = 0; r < FROWS; ++r) red[q][r][o] = acc[r];
    __syncthreads();
    for (int i = threadIdx.x; i < FROWS * OLD; i += blockDim.x) {
        const int r = i / OLD, oo = i - r * OLD;
        if (r0 + r < rows)
            part[((size_t)ks * rows + r0 + r) * OLD + oo] = (red[0][r][oo] + red[1][r][oo]) + (red[2][r][oo] + red[3][r][oo]);
    }
}

__global__ void __launch_bounds__(OLD * KQ) reg_fold_ief_kernel(const RegInitArgs a, const float* __restrict__ bb0,
                                                                const float* __restrict__ bb1, const float* __restrict__ partner,
                                                                int partner_ld, const float* __restrict__ part, int rows,
                                                                const float* __restrict__ bias, const float* __restrict__ wst,
                                                                int iters, int two_view, float* __restrict__ pose0,
                                                                float* __restrict__ betas0, float* __restrict__ pose1,
                                                                float* __restrict__ betas1) {
    __shared__ float st[2][ST];        // state of the pair's views
    __shared__ float S[2][SLD];        // assembled state input rows
    __shared__ float H[2][OLD];        // feature part + bias
    __shared__ float red[KQ][2][OLD];
    const int b = blockIdx.x, nv = two_view ? 2 : 1, o = threadIdx.x % OLD, q = threadIdx.x / OLD;
    for (int i = threadIdx.x; i < nv * ST; i += blockDim.x) {           // reg_init_kernel
        const int v = i / ST, e = i - v * ST;
        const float* pos = (v ? a.pos1 : a.pos0) + (size_t)b * a.pos_bs;
        const float* theta = v ? a.theta1 : a.theta0;
        const float* shape = v ? a.shape1 : a.shape0;
        const float* th = theta ? theta + (size_t)b * (v ? a.theta1_bs : a.theta0_bs) : a.mean_pose;
        const float* sh = shape ? shape + (size_t)b * (v ? a.shape1_bs : a.shape0_bs) : a.mean_shape;
        st[v][e] = e < 3 ? pos[e] : e < 135 ? th[e - 3] : e < 145 ? sh[e - 135] : 0.f;
    }
    for (int i = threadIdx.x; i < nv * OLD; i += blockDim.x) {
        const int v = i / OLD, oo = i - v * OLD;
        const size_t row = (size_t)v * a.B + b;
        float h = oo < 145 ? bias[oo] : 0.f;
        for (int ks = 0; ks < KSPLIT; ++ks) h += part[((size_t)ks * rows + row) * OLD + oo];
        H[v][oo] = h;
    }
    constexpr int KPT = 72;                                  // 4 x 72 = 288 >= 284 (the k-major weights are zero-padded)
    const float* w = wst + (size_t)q * KPT * OLD + o;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        for (int i = threadIdx.x; i < nv * SLD; i += blockDim.x) {      // reg_update_assemble_kernel
            const int v = i / SLD, e = i - v * SLD;
            float val = 0.f;
            if (e < 3) val = (v ? bb1 : bb0)[(size_t)b * 3 + e];
            else if (e < 148) val = st[v][e - 3];
            else if (e < 284) {
                const int k = e - 148;
                if (two_view) val = st[1 - v][k < 126 ? 9 + k : 135 + (k - 126)];
                else val = partner ? partner[(size_t)b * partner_ld + k] : 0.f;   // (no partner: single-view model)
            }
            S[v][e] = val;
        }
        if (!two_view) for (int i = threadIdx.x; i < SLD; i += blockDim.x) S[1][i] = 0.f;
        __syncthreads();
        float d0 = 0.f, d1 = 0.f;
        for (int k0 = 0; k0 < KPT; k0 += 8) {
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = w[(size_t)(k0 + u) * OLD];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                d0 = fmaf(wv[u], S[0][q * KPT + k0 + u], d0);
                d1 = fmaf(wv[u], S[1][q * KPT + k0 + u], d1);
            }
        }
        red[q][0][o] = d0;
        red[q][1][o] = d1;
        __syncthreads();
        for (int i = threadIdx.x; i < nv * 145; i += blockDim.x) {
            const int v = i / 145, oo = i - v * 145;
            st[v][oo] += H[v][oo] + ((red[0][v][oo] + red[1][v][oo]) + (red[2][v][oo] + red[3][v][oo]));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nv * 145; i += blockDim.x) {          // reg_output_kernel
        const int v = i / 145, e = i - v * 145;
        float* pose = (v ? pose1 : pose0) + (size_t)b * 135;
        float* betas = (v ? betas1 : betas0) + (size_t)b * 10;
        if (e < 135) pose[e] = st[v][e];
        else betas[e - 135] = st[v][e];
    }
}

// ---- view-split step in two halves (fc1 is linear, so the step splits by columns of the folded map) --------------
//   hfeat[b]   = bf + Wf[:, :2048] xf[b]                                   (once per forward: ap_regressor_feat_part)
//   partial[b] = hfeat[b] + Wf[:, 2048:2196] [bb | pos | orient | art | shape]   (partner-independent: runs while the exchange is in flight)
//   out[b]     = state[b] + partial[b] + Wf[:, 2196:2332] partner[b]             (the 136 partner columns + the residual add)
__global__ void __launch_bounds__(OLD * KQ) reg_feat_sum_kernel(const float* __restrict__ part, int rows, const float* __restrict__ bias,
                                                                float* __restrict__ hfeat) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * OLD) return;
    const int row = i / OLD, oo = i - row * OLD;
    float h = oo < 145 ? bias[oo] : 0.f;
    for (int ks = 0; ks < KSPLIT; ++ks) h += part[((size_t)ks * rows + row) * OLD + oo];
    hfeat[i] = h;
}

// K0 .. K0 + KN of the k-major state map against one assembled row per sample; thread = (output o, k-quarter q)
template <int K0, int KN, bool FINISH>
__global__ void __launch_bounds__(OLD * KQ) reg_step_half_kernel(const float* __restrict__ base /* hfeat | partial: [B][OLD] */,
                                                                 const float* __restrict__ bb, const float* __restrict__ pose_in,
                                                                 const float* __restrict__ betas_in, const float* __restrict__ partner,
                                                                 int partner_ld, const float* __restrict__ wst,
                                                                 float* __restrict__ out /* partial [B][OLD] */, float* __restrict__ pose_out,
                                                                 float* __restrict__ betas_out) {
    constexpr int KPT = (KN + KQ - 1) / KQ;
    __shared__ float S[KQ * KPT];
    __shared__ float red[KQ][OLD];
    const int b = blockIdx.x, o = threadIdx.x % OLD, q = threadIdx.x / OLD;
    for (int i = threadIdx.x; i < KQ * KPT; i += blockDim.x) {
        float val = 0.f;
        if (i < KN) {
            const int e = K0 + i;                            // column of the assembled row [bb3 | pose135 | betas10 | partner136]
            if (e < 3) val = bb[(size_t)b * 3 + e];
            else if (e < 138) val = pose_in[(size_t)b * 135 + (e - 3)];
            else if (e < 148) val = betas_in[(size_t)b * 10 + (e - 138)];
            else val = partner[(size_t)b * partner_ld + (e - 148)];
        }
        S[i] = val;
    }
    __syncthreads();
    float d = 0.f;
    const float* w = wst + (size_t)(K0 + q * KPT) * OLD + o;
    for (int k = 0; k < KPT; ++k)
        if (q * KPT + k < KN) d = fmaf(w[(size_t)k * OLD], S[q * KPT + k], d);
    red[q][o] = d;
    __syncthreads();
    if (threadIdx.x < 145) {
        const int oo = threadIdx.x;
        const float v = base[(size_t)b * OLD + oo] + ((red[0][oo] + red[1][oo]) + (red[2][oo] + red[3][oo]));
        if (!FINISH) out[(size_t)b * OLD + oo] = v;
        else if (oo < 135) pose_out[(size_t)b * 135 + oo] = pose_in[(size_t)b * 135 + oo] + v;
        else betas_out[(size_t)b * 10 + (oo - 135)] = betas_in[(size_t)b * 10 + (oo - 135)] + v;
    }
}

// ---- single-view HMR head (model_hmr.py:112-172): state = pose132 | shape10 | cam3 (145), row stride 160
__global__ void hmr_init_kernel(const float* __restrict__ theta, int theta_bs, const float* __restrict__ shape,
                                int shape_bs, const float* __restrict__ cam, int cam_bs,
                                const float* __restrict__ mean_pose, const float* __restrict__ mean_shape,
                                const float* __restrict__ mean_cam, float* __restrict__ state) {
    const int b = blockIdx.x;
    const float* th = theta ? theta + (size_t)b * theta_bs : mean_pose;
    const float* sh = shape ? shape + (size_t)b * shape_bs : mean_shape;
    const float* cm = cam ? cam + (size_t)b * cam_bs : mean_cam;
    for (int i = threadIdx.x; i < 160; i += blockDim.x)
        state[(size_t)b * 160 + i] = i < 132 ? th[i] : i < 142 ? sh[i - 132] : i < 145 ? cm[i - 142] : 0.f;
}

__global__ void hmr_update_kernel(float* __restrict__ state, const float* __restrict__ delta, int ldd) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < 145; i += blockDim.x) state[(size_t)b * 160 + i] += delta[(size_t)b * ldd + i];
}

__device__ __forceinline__ void rot6d_rows(const float* __restrict__ x, float* R) {   // geometry.py:47-61
    const float a1x = x[0], a1y = x[2], a1z = x[4], a2x = x[1], a2y = x[3], a2z = x[5];
    const float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-12f);
    const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
    const float d = b1x * a2x + b1y * a2y + b1z * a2z;
    const float ux = a2x - d * b1x, uy = a2y - d * b1y, uz = a2z - d * b1z;
    const float n2 = fmaxf(sqrtf(ux * ux + uy * uy + uz * uz), 1e-12f);
    const float b2x = ux / n2, b2y = uy / n2, b2z = uz / n2;
    R[0] = b1x; R[1] = b2x; R[2] = b1y * b2z - b1z * b2y;
    R[3] = b1y; R[4] = b2y; R[5] = b1z * b2x - b1x * b2z;
    R[6] = b1z; R[7] = b2z; R[8] = b1x * b2y - b1y * b2x;
}

__global__ void hmr_output_kernel(const float* __restrict__ state, float* __restrict__ rotmat,
                                  float* __restrict__ betas, float* __restrict__ cam) {
    const int b = blockIdx.x, t = threadIdx.x;
    const float* s = state + (size_t)b * 160;
    if (t < 22) {
        float R[9];
        rot6d_rows(s + 6 * t, R);
        for (int e = 0; e < 9; ++e) rotmat[((size_t)b * 22 + t) * 9 + e] = R[e];
    } else if (t < 32) {
        betas[(size_t)b * 10 + (t - 22)] = s[132 + (t - 22)];
    } else if (t < 35) {
        cam[(size_t)b * 3 + (t - 32)] = s[142 + (t - 32)];
    }
}

// ---- small helpers of api.hip ---------------------------------------------------------------------------------------
// snapshot of the fp16 range flag (host-mapped word) into a per-slot host-mapped word, in stream order (ap_net_range_mark)
__global__ void word_copy_kernel(const int* __restrict__ src, int* __restrict__ dst) {
    if (threadIdx.x == 0) __hip_atomic_store(dst, __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_SYSTEM);
}
// probe batch of ap_net_parity_probe: crops ~ N(0, 1) (post-normalisation statistics), counter-based (splitmix64 +
// Box-Muller: element i depends on (seed, i) only)
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void probe_normal_kernel(float* __restrict__ x, size_t n, uint64_t seed) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t r = splitmix64(seed * 0x100000001B3ull + i);
    const float u = ((float)(uint32_t)(r >> 40) + 0.5f) * (1.0f / 16777216.0f);          // (0, 1)
    const float v = (float)(uint32_t)((r >> 16) & 0xffffffu) * (1.0f / 16777216.0f);
    x[i] = sqrtf(-2.0f * logf(u)) * cosf(6.2831853071795865f * v);
}
// bb = [U(-.5, .5), U(-.5, .5), U(.2, 1)] per row
__global__ void probe_bb_kernel(float* __restrict__ bb, int rows, uint64_t seed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * 3) return;
    const uint64_t r = splitmix64(seed * 0x9E3779B1ull + 0x5bd1e995ull + (uint64_t)i);
    const float u = (float)(uint32_t)(r >> 40) * (1.0f / 16777216.0f);
    bb[i] = (i % 3) < 2 ? u - 0.5f : 0.2f + 0.8f * u;
}

}  // namespace

hipError_t ap_launch_hmr_init(const float* theta, int theta_bs, const float* shape, int shape_bs, const float* cam,
                              int cam_bs, const float* mean_pose, const float* mean_shape, const float* mean_cam,
                              float* state, int B, hipStream_t st) {
    hipLaunchKernelGGL(hmr_init_kernel, dim3(B), dim3(64), 0, st, theta, theta_bs, shape, shape_bs, cam, cam_bs,
                       mean_pose, mean_shape, mean_cam, state);
    return hipGetLastError();
}
hipError_t ap_launch_hmr_update(float* state, const float* delta, int ldd, int B, hipStream_t st) {
    hipLaunchKernelGGL(hmr_update_kernel, dim3(B), dim3(64), 0, st, state, delta, ldd);
    return hipGetLastError();
}
hipError_t ap_launch_hmr_output(const float* state, float* rotmat, float* betas, float* cam, int B, hipStream_t st) {
    hipLaunchKernelGGL(hmr_output_kernel, dim3(B), dim3(64), 0, st, state, rotmat, betas, cam);
    return hipGetLastError();
}

hipError_t ap_launch_reg_fold_ief(const RegInitArgs& a, const float* xf0, const float* xf1, const float* bb0,
                                  const float* bb1, const float* partner, int partner_ld, const float* wt_feat,
                                  const float* wt_state, const float* bias, float* part, int iters, int two_view,
                                  float* pose0, float* betas0, float* pose1, float* betas1, hipStream_t st) {
    const int rows = a.rows;
    hipLaunchKernelGGL(reg_feat_splitk_kernel, dim3((rows + FROWS - 1) / FROWS, KSPLIT), dim3(OLD * KQ), 0, st, xf0, xf1, a.B,
                       rows, wt_feat, part);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(reg_fold_ief_kernel, dim3(a.B), dim3(OLD * KQ), 0, st, a, bb0, bb1, partner, partner_ld, part, rows, bias,
                       wt_state, iters, two_view, pose0, betas0, pose1, betas1);
    return hipGetLastError();
}
int ap_reg_fold_part_floats(int rows) { return KSPLIT * rows * OLD; }

hipError_t ap_launch_reg_feat_part(const float* xf, int B, const float* wt_feat, const float* bias, float* part, float* hfeat,
                                   hipStream_t st) {
    hipLaunchKernelGGL(reg_feat_splitk_kernel, dim3((B + FROWS - 1) / FROWS, KSPLIT), dim3(OLD * KQ), 0, st, xf, xf, B, B, wt_feat, part);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(reg_feat_sum_kernel, dim3((B * OLD + OLD * KQ - 1) / (OLD * KQ)), dim3(OLD * KQ), 0, st, part, B, bias, hfeat);
    return hipGetLastError();
}
hipError_t ap_launch_reg_step_local(const float* hfeat, const float* bb, const float* pose_in, const float* betas_in, int B,
                                    const float* wt_state, float* partial, hipStream_t st) {
    hipLaunchKernelGGL((reg_step_half_kernel<0, 148, false>), dim3(B), dim3(OLD * KQ), 0, st, hfeat, bb, pose_in, betas_in,
                       (const float*)nullptr, 0, wt_state, partial, (float*)nullptr, (float*)nullptr);
    return hipGetLastError();
}
hipError_t ap_launch_reg_step_finish(const float* partial, const float* pose_in, const float* betas_in, const float* partner,
                                     int partner_ld, int B, const float* wt_state, float* pose_out, float* betas_out, hipStream_t st) {
    hipLaunchKernelGGL((reg_step_half_kernel<148, 136, true>), dim3(B), dim3(OLD * KQ), 0, st, partial, (const float*)nullptr, pose_in,
                       betas_in, partner, partner_ld, wt_state, (float*)nullptr, pose_out, betas_out);
    return hipGetLastError();
}


hipError_t ap_launch_reg_init(const RegInitArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(reg_init_kernel, dim3(a.rows), dim3(64), 0, st, a);
    return hipGetLastError();
}

hipError_t ap_launch_reg_update_assemble(float* state, const float* delta, int ldd, const float* bb0,
                                         const float* bb1, const float* partner, int partner_ld, float* S, int B,
                                         int two_view, hipStream_t st) {
    hipLaunchKernelGGL(reg_update_assemble_kernel, dim3(B), dim3(128), 0, st, state, delta, ldd, bb0, bb1, partner,
                       partner_ld, S, B, two_view);
    return hipGetLastError();
}

hipError_t ap_launch_reg_output(const float* state, float* pose0, float* betas0, float* pose1, float* betas1, int B,
                                int two_view, hipStream_t st) {
    hipLaunchKernelGGL(reg_output_kernel, dim3(two_view ? 2 * B : B), dim3(64), 0, st, state, pose0, betas0, pose1,
                       betas1, B, two_view);
    return hipGetLastError();
}

hipError_t ap_launch_word_copy(const int* src, int* dst, hipStream_t st) {
    hipLaunchKernelGGL(word_copy_kernel, dim3(1), dim3(64), 0, st, src, dst);
    return hipGetLastError();
}
hipError_t ap_launch_probe_inputs(float* x, size_t n, float* bb, int rows, uint64_t seed, hipStream_t st) {
    hipLaunchKernelGGL(probe_normal_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n, seed);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(probe_bb_kernel, dim3((rows * 3 + 63) / 64), dim3(64), 0, st, bb, rows, seed);
    return hipGetLastError();
}
