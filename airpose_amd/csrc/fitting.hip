// AirPose+ fitting loop (BASELINE config 5): copenet_real_data/scripts/bundle_adj.py:262-401 on the GPU, fp32.
//
// 300 Adam steps over a sequence of L frames on [z (VPoser latent, L x 32) | per-view root 6-D rotation and translation
// | shared beta]; objective = Geman-McClure 2-D reprojection of the first 24 SMPL-X chain joints in both views against
// two detectors (:344-347) + VPoser prior (:355) + temporal smoothness (:360-365).  Only the posed JOINTS enter the
// loss (smplx_out.Jtr, :325-334), so the backward pass runs through the kinematic chain and the joint regressor's shape
// directions -- not through skinning.  Every adjoint below is written by hand (no autograd on the product path):
//   VPoser decoder MLP (three small GEMMs, fit_linear_kernel) -> Gram-Schmidt 6-D -> rotation matrix -> axis-angle
//   (tgm 0.1.2 rotation_matrix_to_angle_axis, the decoder's matrot2aa) -> lbs.batch_rodrigues -> kinematic chain ->
//   rigid transform (pytorch3d rotation_6d_to_matrix) -> camera -> projection -> Geman-McClure.
// Script quirks reproduced (see oracle/fitting_ref.py): sigma = 30, hip confidences halved on every iteration,
// loss_beta without gradient, two Adam instances switching at iteration 100.
// One wave per frame runs the whole geometric forward + backward (fit_frame_kernel); per iteration:
//   fit_decode_kernel (decoder + matrot2aa), fit_frame_kernel, fit_backprop_adam_kernel (decoder backward + Adam);
// before the switch to the second optimiser only fit_frame_kernel + fit_adam_kernel.
#include "ap_common.h"
#include "kernels.h"

namespace {

constexpr int NJ = 24, NB = 21;
__constant__ int c_parent[NJ] = {-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15};
__constant__ int c_depth[NJ] = {0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 4, 4, 5, 5, 5, 6, 6, 7, 7, 6, 6};

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

// Gram-Schmidt of both 6-D conventions: b1 = a1/|a1|, b2 = (a2 - (b1.a2) b1)/|..|, b3 = b1 x b2
struct GS { V3 b1, b2, b3; float n1, nu, s; };
__device__ __forceinline__ GS gs_fwd(V3 a1, V3 a2) {
    GS g;
    g.n1 = fmaxf(sqrtf(dot(a1, a1)), 1e-12f);
    g.b1 = (1.f / g.n1) * a1;
    g.s = dot(g.b1, a2);
    const V3 u = a2 - g.s * g.b1;
    g.nu = fmaxf(sqrtf(dot(u, u)), 1e-12f);
    g.b2 = (1.f / g.nu) * u;
    g.b3 = cross(g.b1, g.b2);
    return g;
}
__device__ __forceinline__ void gs_bwd(const GS& g, V3 a2, V3 db1, V3 db2, V3 db3, V3& da1, V3& da2) {
    db1 = db1 + cross(g.b2, db3);                            // b3 = b1 x b2
    db2 = db2 + cross(db3, g.b1);
    const V3 du = (1.f / g.nu) * (db2 - dot(g.b2, db2) * g.b2);
    da2 = du;
    const float ds = -dot(g.b1, du);                         // u = a2 - s b1
    db1 = db1 - g.s * du;
    db1 = db1 + ds * a2;                                     // s = b1 . a2
    da2 = da2 + ds * g.b1;
    da1 = (1.f / g.n1) * (db1 - dot(g.b1, db1) * g.b1);
}

// rotation matrix (row-major r[9]) -> axis-angle, tgm 0.1.2 (see smplx.hip rotmat_to_angle_axis_kernel), keeping what
// the adjoint needs
struct AA { V3 aa; float q[4], t, s, w, x, y, z; int br; };
__device__ __forceinline__ AA aa_fwd(const float* r) {
    AA a;
    const float t00 = r[0], t10 = r[1], t20 = r[2], t01 = r[3], t11 = r[4], t21 = r[5], t02 = r[6], t12 = r[7], t22 = r[8];
    if (t22 < 1e-6f) {
        if (t00 > t11) { a.br = 0; a.t = 1 + t00 - t11 - t22; a.q[0] = t12 - t21; a.q[1] = a.t; a.q[2] = t01 + t10; a.q[3] = t20 + t02; }
        else           { a.br = 1; a.t = 1 - t00 + t11 - t22; a.q[0] = t20 - t02; a.q[1] = t01 + t10; a.q[2] = a.t; a.q[3] = t12 + t21; }
    } else {
        if (t00 < -t11) { a.br = 2; a.t = 1 - t00 - t11 + t22; a.q[0] = t01 - t10; a.q[1] = t20 + t02; a.q[2] = t12 + t21; a.q[3] = a.t; }
        else            { a.br = 3; a.t = 1 + t00 + t11 + t22; a.q[0] = a.t; a.q[1] = t12 - t21; a.q[2] = t20 - t02; a.q[3] = t01 - t10; }
    }
    a.s = 0.5f / sqrtf(a.t);
    a.w = a.q[0] * a.s; a.x = a.q[1] * a.s; a.y = a.q[2] * a.s; a.z = a.q[3] * a.s;
    const float ss = a.x * a.x + a.y * a.y + a.z * a.z, sn = sqrtf(ss);
    const float two_theta = 2.0f * (a.w < 0.f ? atan2f(-sn, -a.w) : atan2f(sn, a.w));
    const float k = ss > 0.f ? two_theta / sn : 2.0f;
    a.aa = v3(a.x * k, a.y * k, a.z * k);
    return a;
}
// d(loss)/d(aa) -> d(loss)/d(R) (row-major dr[9], overwritten)
__device__ __forceinline__ void aa_bwd(const AA& a, V3 daa, float* dr) {
    const float ss = a.x * a.x + a.y * a.y + a.z * a.z, sn = sqrtf(ss);
    float dw = 0.f;
    V3 dxyz;
    if (ss > 0.f) {
        const float T = 2.0f * (a.w < 0.f ? atan2f(-sn, -a.w) : atan2f(sn, a.w)), k = T / sn;
        const float dk = daa.x * a.x + daa.y * a.y + daa.z * a.z;
        dxyz = k * daa;
        const float dT = dk / sn;
        float dsn = -dk * T / ss;
        const float den = ss + a.w * a.w;
        dsn += 2.f * dT * a.w / den;
        dw = -2.f * dT * sn / den;
        const float dss = dsn / (2.f * sn);
        dxyz = dxyz + (2.f * dss) * v3(a.x, a.y, a.z);
    } else {
        dxyz = 2.f * daa;
    }
    const float dq[4] = {dw * a.s, dxyz.x * a.s, dxyz.y * a.s, dxyz.z * a.s};
    const float ds = dw * a.q[0] + dxyz.x * a.q[1] + dxyz.y * a.q[2] + dxyz.z * a.q[3];
    float dt = -ds * a.s / (2.f * a.t);
    float d00 = 0, d10 = 0, d20 = 0, d01 = 0, d11 = 0, d21 = 0, d02 = 0, d12 = 0, d22 = 0;   // d/d t_ab (t_ab = r[b][a])
    switch (a.br) {
        case 0: dt += dq[1]; d12 += dq[0]; d21 -= dq[0]; d01 += dq[2]; d10 += dq[2]; d20 += dq[3]; d02 += dq[3];
                d00 += dt; d11 -= dt; d22 -= dt; break;
        case 1: dt += dq[2]; d20 += dq[0]; d02 -= dq[0]; d01 += dq[1]; d10 += dq[1]; d12 += dq[3]; d21 += dq[3];
                d00 -= dt; d11 += dt; d22 -= dt; break;
        case 2: dt += dq[3]; d01 += dq[0]; d10 -= dq[0]; d20 += dq[1]; d02 += dq[1]; d12 += dq[2]; d21 += dq[2];
                d00 -= dt; d11 -= dt; d22 += dt; break;
        default: dt += dq[0]; d12 += dq[1]; d21 -= dq[1]; d20 += dq[2]; d02 -= dq[2]; d01 += dq[3]; d10 -= dq[3];
                d00 += dt; d11 += dt; d22 += dt; break;
    }
    // t00 = r[0], t10 = r[1], t20 = r[2], t01 = r[3], t11 = r[4], t21 = r[5], t02 = r[6], t12 = r[7], t22 = r[8]
    dr[0] = d00; dr[1] = d10; dr[2] = d20; dr[3] = d01; dr[4] = d11; dr[5] = d21; dr[6] = d02; dr[7] = d12; dr[8] = d22;
}

// lbs.batch_rodrigues: r -> R (row-major), angle = |r + 1e-8|
struct RD { float th, sn, cs; V3 d; };
__device__ __forceinline__ RD rod_fwd(V3 r, float* R) {
    RD o;
    const V3 re = v3(r.x + 1e-8f, r.y + 1e-8f, r.z + 1e-8f);
    o.th = sqrtf(dot(re, re));
    o.d = (1.f / o.th) * r;
    o.sn = sinf(o.th); o.cs = cosf(o.th);
    const float x = o.d.x, y = o.d.y, z = o.d.z, c1 = 1.f - o.cs;
    // K = [[0,-z,y],[z,0,-x],[-y,x,0]];  K^2 = d d^T - |d|^2 I
    const float dd = x * x + y * y + z * z;
    R[0] = 1.f + c1 * (x * x - dd); R[1] = -o.sn * z + c1 * x * y;  R[2] = o.sn * y + c1 * x * z;
    R[3] = o.sn * z + c1 * x * y;   R[4] = 1.f + c1 * (y * y - dd); R[5] = -o.sn * x + c1 * y * z;
    R[6] = -o.sn * y + c1 * x * z;  R[7] = o.sn * x + c1 * y * z;   R[8] = 1.f + c1 * (z * z - dd);
    return o;
}
__device__ __forceinline__ V3 rod_bwd(const RD& o, V3 r, const float* dR) {
    const float x = o.d.x, y = o.d.y, z = o.d.z, c1 = 1.f - o.cs, dd = x * x + y * y + z * z;
    const float K[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    const float K2[9] = {x * x - dd, x * y, x * z, x * y, y * y - dd, y * z, x * z, y * z, z * z - dd};
    float dsin = 0.f, dc1 = 0.f;
#pragma unroll
    for (int e = 0; e < 9; ++e) { dsin += dR[e] * K[e]; dc1 += dR[e] * K2[e]; }
    float dth = o.cs * dsin + o.sn * dc1;
    // dK = sin dR + c1 (dR K^T + K^T dR)
    float dK[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int m = 0; m < 3; ++m) acc += dR[i * 3 + m] * K[j * 3 + m] + K[m * 3 + i] * dR[m * 3 + j];
            dK[i * 3 + j] = o.sn * dR[i * 3 + j] + c1 * acc;
        }
    const V3 dD = v3(dK[7] - dK[5], dK[2] - dK[6], dK[3] - dK[1]);
    V3 dr = (1.f / o.th) * dD;
    dth += -dot(dD, r) / (o.th * o.th);
    const V3 re = v3(r.x + 1e-8f, r.y + 1e-8f, r.z + 1e-8f);
    dr = dr + (dth / o.th) * re;
    return dr;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// Y[L][N] = act(X[L][K] Wt[K][ldw(:N)] + b) (* (G > 0 ? 1 : 0.01) if G): the decoder's layers and their transposes.
// ACT: 0 none, 1 LeakyReLU(0.01).  16 rows x 64 columns per workgroup, weights k-major (coalesced over the output).
template <int ACT>
__global__ void __launch_bounds__(256) fit_linear_kernel(const float* __restrict__ X, int ldx, int K, const float* __restrict__ Wt,
                                                         int ldw, const float* __restrict__ bias, const float* __restrict__ G, int ldg,
                                                         float* __restrict__ Y, int ldy, int L, int N) {
    __shared__ float xs[16][65];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rq = threadIdx.x >> 6, r0 = blockIdx.y * 16;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 64) {
        __syncthreads();
        for (int i = threadIdx.x; i < 16 * 64; i += 256) {
            const int r = i >> 6, k = i & 63;
            xs[r][k] = (r0 + r < L && k0 + k < K) ? X[(size_t)(r0 + r) * ldx + k0 + k] : 0.f;
        }
        __syncthreads();
        if (c < N) {
            const int kn = K - k0 < 64 ? K - k0 : 64;
            for (int k = 0; k < kn; ++k) {
                const float w = Wt[(size_t)(k0 + k) * ldw + c];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaf(w, xs[rq * 4 + j][k], acc[j]);
            }
        }
    }
    if (c >= N) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + rq * 4 + j;
        if (r >= L) continue;
        float v = acc[j] + (bias ? bias[c] : 0.f);
        if (ACT == 1) v = v > 0.f ? v : 0.01f * v;
        if (G) v *= G[(size_t)r * ldg + c] > 0.f ? 1.f : 0.01f;
        Y[(size_t)r * ldy + c] = v;
    }
}

// decoder output -> pose_body axis-angle (needed from the NEIGHBOUR frames by the temporal term)
__global__ void fit_aa_kernel(const float* __restrict__ O, int ldo, float* __restrict__ aa_out, int L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L * NB) return;
    const int f = i / NB, j = i - f * NB;
    const float* o = O + (size_t)f * ldo + 6 * j;
    const GS g = gs_fwd(v3(o[0], o[2], o[4]), v3(o[1], o[3], o[5]));
    const float R[9] = {g.b1.x, g.b2.x, g.b3.x, g.b1.y, g.b2.y, g.b3.y, g.b1.z, g.b2.z, g.b3.z};
    const AA a = aa_fwd(R);
    aa_out[(size_t)i * 3 + 0] = a.aa.x; aa_out[(size_t)i * 3 + 1] = a.aa.y; aa_out[(size_t)i * 3 + 2] = a.aa.z;
}

__global__ void __launch_bounds__(64) fit_frame_kernel(const FitArgs a, int it) {
    const int f = blockIdx.x, lane = threadIdx.x, L = a.L;
    __shared__ float Rl[NJ][9], Jr[NJ][3], G[NJ][12], dG[NJ][12], dRl[NJ][9], dJ[NJ][3], red[32];
    __shared__ float Sd[NJ * 3][10], PT[3][2][9], CAM[32];
    // ---- every global input of the frame is requested up front (one memory latency instead of one per use: the
    //      kernel is a chain of short dependent phases, and it runs 300 times back to back)
    //      Pointers are clamped / selected and the loads unconditional: hipcc waits for a load at the end of the branch it sits
    //      in, and with one branch per input (lane < 54, lane < 32, lane < NJ, pair_prev ...) the "up front" block was ten
    //      dependent round trips in the ISA (24 x s_waitcnt vmcnt(0)) of a kernel that is nothing but latency.
    const int fp = f > 0 ? f - 1 : f, fn = f + 1 < L ? f + 1 : f;
    // (the lane's depth and parent once, in registers: read inside the level loops they were a memory round trip per level)
    const int my_depth = c_depth[min(lane, NJ - 1)], my_parent = c_parent[min(lane, NJ - 1)];
    const int rob_i = a.robust[f], robp_i = a.robust[fp], robn_i = a.robust[fn];
    constexpr int SIT = (NJ * 3 * 10 + 63) / 64;             // 12 loads per lane, all issued before the first LDS write
    float sv[SIT];
#pragma unroll
    for (int k = 0; k < SIT; ++k) {
        const int i = min(lane + k * 64, NJ * 3 * 10 - 1);
        sv[k] = a.j_shapedirs[(i / 10) * a.jsd_ld + i % 10];
    }
    float ptv;
    {                                                        // phi | tau of this frame and of its neighbours, both views (lanes 0..53)
        const int l54 = min(lane, 53), which = l54 / 18, v = (l54 % 18) / 9, e = l54 % 9;
        const int ff = which == 0 ? f : which == 1 ? fp : fn;
        ptv = *(e < 6 ? a.phi + ((size_t)v * L + ff) * 6 + e : a.tau + ((size_t)v * L + ff) * 3 + (e - 6));
    }
    const float camv = *(lane < 24 ? a.extr + lane : a.intr + (min(lane, 31) - 24));
    float gtv[2][2][3], tmpl[3], aprev[3], anext[3], betav[10], ov[6];
#pragma unroll
    for (int k = 0; k < 10; ++k) betav[k] = a.beta[k];
    {
        const int lj = min(lane, NJ - 1), lb = min(lane, NB - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) tmpl[c] = a.j_template[lj * 3 + c];
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int det = 0; det < 2; ++det) {
                const float* gt = a.j2d + ((((size_t)v * L + f) * 2 + det) * NJ + lj) * 3;
                gtv[v][det][0] = gt[0]; gtv[v][det][1] = gt[1]; gtv[v][det][2] = gt[2];
            }
        const float* Ap = a.aa_all + ((size_t)fp * NB + lb) * 3;
        const float* An = a.aa_all + ((size_t)fn * NB + lb) * 3;
        const float* o = a.O + (size_t)f * a.ldo + 6 * lb;
#pragma unroll
        for (int c = 0; c < 3; ++c) { aprev[c] = Ap[c]; anext[c] = An[c]; }
#pragma unroll
        for (int c = 0; c < 6; ++c) ov[c] = o[c];
    }
    const bool rob = rob_i != 0;
    const bool pair_prev = a.n_pairs > 0 && f > 0 && robp_i && rob;
    const bool pair_next = a.n_pairs > 0 && f + 1 < L && rob && robn_i;
#pragma unroll
    for (int k = 0; k < SIT; ++k) {
        const int i = lane + k * 64;
        if (i < NJ * 3 * 10) Sd[i / 10][i % 10] = sv[k];
    }
    if (lane < 54) PT[lane / 18][(lane % 18) / 9][lane % 9] = ptv;
    if (lane < 32) CAM[lane] = camv;
    if (!pair_prev) { aprev[0] = aprev[1] = aprev[2] = 0.f; }
    if (!pair_next) { anext[0] = anext[1] = anext[2] = 0.f; }
    // ---- decoder tail: 6-D -> R -> axis-angle -> R' (lbs.batch_rodrigues), body joints 1..21
    GS g; AA q; RD rd;
    V3 a2 = v3(0, 0, 0), aav = v3(0, 0, 0);
    if (lane < NB) {
        a2 = v3(ov[1], ov[3], ov[5]);
        g = gs_fwd(v3(ov[0], ov[2], ov[4]), a2);
        const float R[9] = {g.b1.x, g.b2.x, g.b3.x, g.b1.y, g.b2.y, g.b3.y, g.b1.z, g.b2.z, g.b3.z};
        q = aa_fwd(R);
        aav = q.aa;
        rd = rod_fwd(aav, Rl[lane + 1]);
    } else if (lane < NJ) {                                  // joints 0, 22, 23: zero axis-angle (root_orient = 0, jaw, eye)
        const int j = lane == NB ? 0 : lane;
        RD t = rod_fwd(v3(0, 0, 0), Rl[j]);
        (void)t;
    }
    if (lane < NJ) {
#pragma unroll
        for (int c = 0; c < 3; ++c) dJ[lane][c] = 0.f;
#pragma unroll
        for (int e = 0; e < 12; ++e) dG[lane][e] = 0.f;
    }
    __syncthreads();                                         // Sd, PT, CAM staged
    if (lane < NJ) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = tmpl[c];
#pragma unroll
            for (int k = 0; k < 10; ++k) v += Sd[lane * 3 + c][k] * betav[k];
            Jr[lane][c] = v;
        }
    }
    __syncthreads();
    // ---- kinematic chain (batch_rigid_transform), level by level
    for (int d = 0; d <= 7; ++d) {
        if (lane < NJ && my_depth == d) {
            const int p = my_parent;
            if (p < 0) {
#pragma unroll
                for (int e = 0; e < 9; ++e) G[lane][e] = Rl[lane][e];
                G[lane][9] = Jr[lane][0]; G[lane][10] = Jr[lane][1]; G[lane][11] = Jr[lane][2];
            } else {
                const float rel[3] = {Jr[lane][0] - Jr[p][0], Jr[lane][1] - Jr[p][1], Jr[lane][2] - Jr[p][2]};
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        G[lane][i * 3 + j] = G[p][i * 3] * Rl[lane][j] + G[p][i * 3 + 1] * Rl[lane][3 + j] + G[p][i * 3 + 2] * Rl[lane][6 + j];
                    G[lane][9 + i] = G[p][i * 3] * rel[0] + G[p][i * 3 + 1] * rel[1] + G[p][i * 3 + 2] * rel[2] + G[p][9 + i];
                }
            }
        }
        __syncthreads();
    }
    // ---- both views: rigid transform, camera, projection, Geman-McClure; gradients to the joints and the rigid pose
    float loss2d = 0.f;
    const float hipw = (lane == 1 || lane == 2) ? exp2f(-(float)(it + 1)) : 1.f;        // halved in place every iteration
    const float inv_n = a.n_robust > 0 ? 1.f / (48.f * a.n_robust) : 0.f;
    V3 pj = v3(0, 0, 0), dpj = v3(0, 0, 0);
    if (lane < NJ) pj = v3(G[lane][9], G[lane][10], G[lane][11]);
    for (int v = 0; v < 2; ++v) {
        const float* ph = &PT[0][v][0];
        const float* ta = &PT[0][v][6];
        const V3 pa2 = v3(ph[3], ph[4], ph[5]);
        const GS gv = gs_fwd(v3(ph[0], ph[1], ph[2]), pa2);  // rows of R_v
        V3 dX = v3(0, 0, 0);
        if (lane < NJ && rob) {
            const V3 X = v3(dot(gv.b1, pj) + ta[0], dot(gv.b2, pj) + ta[1], dot(gv.b3, pj) + ta[2]);
            const float* E = &CAM[v * 12];
            const V3 Xc = v3(E[0] * X.x + E[1] * X.y + E[2] * X.z + E[3], E[4] * X.x + E[5] * X.y + E[6] * X.z + E[7],
                             E[8] * X.x + E[9] * X.y + E[10] * X.z + E[11]);
            const float fx = CAM[24 + v * 4], fy = CAM[25 + v * 4], cx = CAM[26 + v * 4], cy = CAM[27 + v * 4];
            const float u = fx * Xc.x / Xc.z + cx, w = fy * Xc.y / Xc.z + cy;
            float du = 0.f, dw = 0.f;
            for (int det = 0; det < 2; ++det) {
                const float* gt = gtv[v][det];
                const float cf = gt[2] * hipw * inv_n, s2 = a.sigma * a.sigma;
                const float ex = u - gt[0], ey = w - gt[1];
                loss2d += cf * (ex * ex / (ex * ex + s2) + ey * ey / (ey * ey + s2));
                du += cf * 2.f * ex * s2 / ((ex * ex + s2) * (ex * ex + s2));
                dw += cf * 2.f * ey * s2 / ((ey * ey + s2) * (ey * ey + s2));
            }
            const V3 dXc = v3(du * fx / Xc.z, dw * fy / Xc.z, -(du * fx * Xc.x + dw * fy * Xc.y) / (Xc.z * Xc.z));
            dX = v3(E[0] * dXc.x + E[4] * dXc.y + E[8] * dXc.z, E[1] * dXc.x + E[5] * dXc.y + E[9] * dXc.z,
                    E[2] * dXc.x + E[6] * dXc.y + E[10] * dXc.z);
            dpj = dpj + dX.x * gv.b1 + dX.y * gv.b2 + dX.z * gv.b3;       // R_v^T dX
        }
        // reductions over the joints: d tau = sum dX, d R_v[r][:] = sum dX_r p
        const float dt0 = wave_sum(dX.x), dt1 = wave_sum(dX.y), dt2 = wave_sum(dX.z);
        const V3 db1 = v3(wave_sum(dX.x * pj.x), wave_sum(dX.x * pj.y), wave_sum(dX.x * pj.z));
        const V3 db2 = v3(wave_sum(dX.y * pj.x), wave_sum(dX.y * pj.y), wave_sum(dX.y * pj.z));
        const V3 db3 = v3(wave_sum(dX.z * pj.x), wave_sum(dX.z * pj.y), wave_sum(dX.z * pj.z));
        if (lane == 0) {
            V3 da1, da2;
            gs_bwd(gv, pa2, db1, db2, db3, da1, da2);
            float dph[6] = {da1.x, da1.y, da1.z, da2.x, da2.y, da2.z}, dta[3] = {dt0, dt1, dt2};
            // temporal terms on phi and tau: 100 * mean over (selected pairs x dim) of squared differences
            if (a.n_pairs > 0) {
                const float cp = 2.f * 100.f * a.w_temporal / (6.f * a.n_pairs), ct = 2.f * 100.f * a.w_temporal / (3.f * a.n_pairs);
                if (pair_prev) {
                    for (int e = 0; e < 6; ++e) dph[e] += cp * (ph[e] - PT[1][v][e]);
                    for (int e = 0; e < 3; ++e) dta[e] += ct * (ta[e] - PT[1][v][6 + e]);
                }
                if (pair_next) {
                    for (int e = 0; e < 6; ++e) dph[e] -= cp * (PT[2][v][e] - ph[e]);
                    for (int e = 0; e < 3; ++e) dta[e] -= ct * (PT[2][v][6 + e] - ta[e]);
                }
            }
            for (int e = 0; e < 6; ++e) a.dphi[((size_t)v * L + f) * 6 + e] = dph[e];
            for (int e = 0; e < 3; ++e) a.dtau[((size_t)v * L + f) * 3 + e] = dta[e];
        }
    }
    if (lane < NJ) { dG[lane][9] = dpj.x; dG[lane][10] = dpj.y; dG[lane][11] = dpj.z; }
    __syncthreads();
    // ---- chain backward, deepest level first: G_i = G_p [R_i | rel_i]
    for (int d = 7; d >= 1; --d) {
        if (lane < NJ && my_depth == d) {
            const int p = my_parent;
            const float rel[3] = {Jr[lane][0] - Jr[p][0], Jr[lane][1] - Jr[p][1], Jr[lane][2] - Jr[p][2]};
            float drel[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                atomicAdd(&dG[p][9 + i], dG[lane][9 + i]);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    // dG_p.R[i][j] += dt_i rel_j + sum_m dG_i.R[i][m] R_i[j][m]
                    float acc = dG[lane][9 + i] * rel[j];
#pragma unroll
                    for (int m = 0; m < 3; ++m) acc += dG[lane][i * 3 + m] * Rl[lane][j * 3 + m];
                    atomicAdd(&dG[p][i * 3 + j], acc);
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                drel[i] = G[p][i] * dG[lane][9] + G[p][3 + i] * dG[lane][10] + G[p][6 + i] * dG[lane][11];   // G_p.R^T dt
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    dRl[lane][i * 3 + j] = G[p][i] * dG[lane][j] + G[p][3 + i] * dG[lane][3 + j] + G[p][6 + i] * dG[lane][6 + j];
                atomicAdd(&dJ[lane][i], drel[i]);
                atomicAdd(&dJ[p][i], -drel[i]);
            }
        }
        __syncthreads();
    }
    if (lane == 0) { dJ[0][0] += dG[0][9]; dJ[0][1] += dG[0][10]; dJ[0][2] += dG[0][11]; }
    __syncthreads();
    // ---- back through batch_rodrigues, the temporal term on pose_body, matrot2aa and the decoder's Gram-Schmidt
    float ltemp = 0.f;
    if (lane < NB) {
        V3 daa = rod_bwd(rd, aav, dRl[lane + 1]);
        if (a.n_pairs > 0) {
            const float ca = 2.f * 10.f * a.w_temporal / (63.f * a.n_pairs);
            if (pair_prev) {
                const V3 dprev = v3(aav.x - aprev[0], aav.y - aprev[1], aav.z - aprev[2]);
                daa = daa + ca * dprev;
                ltemp += 10.f * a.w_temporal / (63.f * a.n_pairs) * dot(dprev, dprev);     // pair (f-1, f) accounted to frame f
            }
            if (pair_next)
                daa = daa - ca * v3(anext[0] - aav.x, anext[1] - aav.y, anext[2] - aav.z);
        }
        float dR[9];
        aa_bwd(q, daa, dR);
        V3 da1, da2o;
        gs_bwd(g, a2, v3(dR[0], dR[3], dR[6]), v3(dR[1], dR[4], dR[7]), v3(dR[2], dR[5], dR[8]), da1, da2o);
        float* dO = a.dO + (size_t)f * a.ldo + 6 * lane;
        dO[0] = da1.x; dO[1] = da2o.x; dO[2] = da1.y; dO[3] = da2o.y; dO[4] = da1.z; dO[5] = da2o.z;
    } else if (lane < NB + 2) {
        a.dO[(size_t)f * a.ldo + 6 * NB + (lane - NB)] = 0.f;                               // padding columns 126, 127
    }
    // ---- beta: J = J_template + J_shapedirs beta
    if (lane < 10) {
        float s = 0.f;
        for (int i = 0; i < NJ; ++i)
            for (int c = 0; c < 3; ++c) s += Sd[i * 3 + c][lane] * dJ[i][c];
        a.dbeta_part[(size_t)f * 10 + lane] = s;
    }
    // ---- loss bookkeeping (reporting only): 2-D | temporal on pose_body | on phi, tau (pairs accounted to frame f)
    float lrig = 0.f;
    if (lane < 18 && pair_prev) {
        const int v = lane / 9, e = lane % 9;
        const float d = PT[0][v][e] - PT[1][v][e];
        lrig = 100.f * a.w_temporal / ((e < 6 ? 6.f : 3.f) * a.n_pairs) * d * d;
    }
    const float l2 = wave_sum(loss2d), lt = wave_sum(ltemp), lr = wave_sum(lrig);
    if (lane == 0) { a.loss_part[(size_t)f * 4 + 0] = l2; a.loss_part[(size_t)f * 4 + 1] = lt; a.loss_part[(size_t)f * 4 + 2] = lr; a.loss_part[(size_t)f * 4 + 3] = 0.f; }
    (void)red;
}

// Adam (torch.optim.Adam, lr 0.01, betas 0.9 / 0.999, eps 1e-8) over [z | phi | tau | beta]; z joins at the switch.
__global__ void fit_adam_kernel(const FitArgs a, int step, int with_z) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int L = a.L, nz = L * 32, nphi = 2 * L * 6, ntau = 2 * L * 3, total = nz + nphi + ntau + 10;
    if (i >= total) return;
    float* p; float gr;
    if (i < nz) {
        if (!with_z) return;
        p = a.z + i;
        gr = a.dz[(size_t)(i / 32) * a.ldz + (i & 31)] + a.w_vposer * 2.f * a.z[i] / (float)nz;     // + VPoser prior mean(z^2)
    } else if (i < nz + nphi) { p = a.phi + (i - nz); gr = a.dphi[i - nz]; }
    else if (i < nz + nphi + ntau) { p = a.tau + (i - nz - nphi); gr = a.dtau[i - nz - nphi]; }
    else {
        const int k = i - nz - nphi - ntau;
        p = a.beta + k;
        gr = 0.f;
        for (int f0 = 0; f0 < L; f0 += 16) {                 // loads batched (independent), adds in frame order
            float pv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) pv[u] = f0 + u < L ? a.dbeta_part[(size_t)(f0 + u) * 10 + k] : 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) gr += pv[u];
        }
    }
    float m = a.adam_m[i], v = a.adam_v[i];
    m = 0.9f * m + 0.1f * gr;
    v = 0.999f * v + 0.001f * gr * gr;
    a.adam_m[i] = m; a.adam_v[i] = v;
    const float bc1 = 1.f - powf(0.9f, (float)step), bc2 = 1.f - powf(0.999f, (float)step);
    *p -= (a.lr / bc1) * m / (sqrtf(v) / sqrtf(bc2) + 1e-8f);
    if (a.grad_out) a.grad_out[i] = gr;
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused decoder passes (latency: one launch instead of three GEMM launches + glue; the sequence is only L = 64 rows, so
// the per-iteration cost is launch-to-launch latency, not work).  A workgroup of 1024 threads owns DR rows and runs the
// whole 32 -> 512 -> 512 -> 126 chain (resp. its transpose).  Every layer has the same shape of work: thread =
// (group of 4 adjacent output columns, K range), weights k-major so that a wave reads whole 16-byte-per-lane rows
// (dwordx4: the texture-address path takes the same time per instruction for 4 and for 16 bytes per lane), partial sums
// reduced over the K ranges through a 64 KiB LDS array.  Weights stream from L2; the bound is one CU's L2 port.
constexpr int DR = 4;
constexpr int FIT_LDS_FLOATS = 16384 + DR * (32 + 512 + 512 + 128);          // red + the activations of the four rows
__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : 0.01f * v; }

// partial[kq][r][c] = sum over this thread's K range of w[k][c] * x[r][k];  C columns (k-major rows of C floats), K
// rows of w of which the first kmax are real; x rows in LDS with stride XLD
template <int C, int K, int XLD>
__device__ __forceinline__ void layer_partial(const float* __restrict__ w, const float* x, float* red, int kmax) {
    constexpr int CG = C / 4, NKQ = 1024 / CG, KP = K / NKQ, U = KP < 8 ? KP : 8;
    static_assert(K % NKQ == 0 && KP % U == 0, "K range split");
    const int t = threadIdx.x, cg = t % CG, kq = t / CG;
    float acc[DR][4] = {};
    for (int k0 = kq * KP; k0 < kq * KP + KP; k0 += U) {
        float4 wv[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            wv[u] = k0 + u < kmax ? *reinterpret_cast<const float4*>(w + (size_t)(k0 + u) * C + 4 * cg) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < DR; ++r) {
                const float xv = x[r * XLD + k0 + u];
                acc[r][0] = fmaf(wv[u].x, xv, acc[r][0]); acc[r][1] = fmaf(wv[u].y, xv, acc[r][1]);
                acc[r][2] = fmaf(wv[u].z, xv, acc[r][2]); acc[r][3] = fmaf(wv[u].w, xv, acc[r][3]);
            }
    }
#pragma unroll
    for (int r = 0; r < DR; ++r)
        *reinterpret_cast<float4*>(red + (size_t)(kq * DR + r) * C + 4 * cg) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
}
template <int C>
__device__ __forceinline__ float layer_sum(const float* red, int r, int o) {
    constexpr int NKQ = 4096 / C;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
#pragma unroll 4
    for (int kq = 0; kq < NKQ; kq += 4) {
        v0 += red[((kq + 0) * DR + r) * C + o]; v1 += red[((kq + 1) * DR + r) * C + o];
        v2 += red[((kq + 2) * DR + r) * C + o]; v3 += red[((kq + 3) * DR + r) * C + o];
    }
    return (v0 + v1) + (v2 + v3);
}

// z -> H1, H2 (kept for the backward pass), O, and pose_body axis-angle
__global__ void __launch_bounds__(1024) fit_decode_kernel(const float* __restrict__ z, int L, const float* __restrict__ w1t,
                                                          const float* __restrict__ b1, const float* __restrict__ w2t,
                                                          const float* __restrict__ b2, const float* __restrict__ w3t,
                                                          const float* __restrict__ b3, float* __restrict__ H1,
                                                          float* __restrict__ H2, float* __restrict__ O, float* __restrict__ aa_out) {
    extern __shared__ float lds[];
    float* red = lds; float* xs = red + 16384; float* h1 = xs + DR * 32; float* h2 = h1 + DR * 512; float* os = h2 + DR * 512;
    const int t = threadIdx.x, r0 = blockIdx.x * DR;
    if (t < DR * 32) { const int r = t >> 5, c = t & 31; xs[r * 32 + c] = r0 + r < L ? z[(size_t)(r0 + r) * 32 + c] : 0.f; }
    __syncthreads();
    layer_partial<512, 32, 32>(w1t, xs, red, 32);
    __syncthreads();
    for (int i = t; i < DR * 512; i += 1024) {
        const int r = i >> 9, o = i & 511;
        const float v = lrelu(layer_sum<512>(red, r, o) + b1[o]);
        h1[r * 512 + o] = v;
        if (r0 + r < L) H1[(size_t)(r0 + r) * 512 + o] = v;
    }
    __syncthreads();
    layer_partial<512, 512, 512>(w2t, h1, red, 512);
    __syncthreads();
    for (int i = t; i < DR * 512; i += 1024) {
        const int r = i >> 9, o = i & 511;
        const float v = lrelu(layer_sum<512>(red, r, o) + b2[o]);
        h2[r * 512 + o] = v;
        if (r0 + r < L) H2[(size_t)(r0 + r) * 512 + o] = v;
    }
    __syncthreads();
    layer_partial<128, 512, 512>(w3t, h2, red, 512);
    __syncthreads();
    if (t < DR * 128) {
        const int r = t >> 7, o = t & 127;
        const float v = layer_sum<128>(red, r, o) + b3[o];
        os[r * 128 + o] = v;
        if (r0 + r < L) O[(size_t)(r0 + r) * 128 + o] = v;
    }
    __syncthreads();
    if (t < DR * NB) {                                       // matrot2aa of the Gram-Schmidt rotations (fit_aa_kernel)
        const int r = t / NB, j = t - r * NB;
        if (r0 + r < L) {
            const float* o = os + r * 128 + 6 * j;
            const GS g = gs_fwd(v3(o[0], o[2], o[4]), v3(o[1], o[3], o[5]));
            const float R[9] = {g.b1.x, g.b2.x, g.b3.x, g.b1.y, g.b2.y, g.b3.y, g.b1.z, g.b2.z, g.b3.z};
            const AA a = aa_fwd(R);
            float* dst = aa_out + ((size_t)(r0 + r) * NB + j) * 3;
            dst[0] = a.aa.x; dst[1] = a.aa.y; dst[2] = a.aa.z;
        }
    }
}

// dO -> dH2 -> dH1 -> dz, then Adam on every optimised quantity of the workgroup's rows (and beta in workgroup 0).
// Weights as torch stores them ([out][in]) ARE k-major for the transposed products.
__global__ void __launch_bounds__(1024) fit_backprop_adam_kernel(const FitArgs a, const float* __restrict__ w3, const float* __restrict__ w2,
                                                                 const float* __restrict__ w1, const float* __restrict__ H1,
                                                                 const float* __restrict__ H2, int step) {
    extern __shared__ float lds[];
    float* red = lds; float* dzs = red + 16384; float* d2 = dzs + DR * 32; float* d1 = d2 + DR * 512; float* dos = d1 + DR * 512;
    const int t = threadIdx.x, r0 = blockIdx.x * DR, L = a.L;
    if (t < DR * 128) { const int r = t >> 7, c = t & 127; dos[r * 128 + c] = (r0 + r < L && c < 126) ? a.dO[(size_t)(r0 + r) * a.ldo + c] : 0.f; }
    __syncthreads();
    layer_partial<512, 128, 128>(w3, dos, red, 126);         // dH2 = (dO W3) * lrelu'(H2)
    __syncthreads();
    for (int i = t; i < DR * 512; i += 1024) {
        const int r = i >> 9, o = i & 511;
        const float hact = r0 + r < L ? H2[(size_t)(r0 + r) * 512 + o] : 0.f;
        d2[r * 512 + o] = layer_sum<512>(red, r, o) * (hact > 0.f ? 1.f : 0.01f);
    }
    __syncthreads();
    layer_partial<512, 512, 512>(w2, d2, red, 512);          // dH1 = (dH2 W2) * lrelu'(H1)
    __syncthreads();
    for (int i = t; i < DR * 512; i += 1024) {
        const int r = i >> 9, o = i & 511;
        const float hact = r0 + r < L ? H1[(size_t)(r0 + r) * 512 + o] : 0.f;
        d1[r * 512 + o] = layer_sum<512>(red, r, o) * (hact > 0.f ? 1.f : 0.01f);
    }
    __syncthreads();
    layer_partial<32, 512, 512>(w1, d1, red, 512);           // dz = dH1 W1
    __syncthreads();
    if (t < DR * 32) { const int r = t >> 5, c = t & 31; dzs[r * 32 + c] = layer_sum<32>(red, r, c); }
    __syncthreads();
    // ---- Adam (as fit_adam_kernel) on this workgroup's share: z, phi, tau of its rows; beta in workgroup 0
    const int nz = L * 32, nphi = 2 * L * 6, ntau = 2 * L * 3;
    int i = -1; float* p = nullptr; float gr = 0.f;
    if (t < DR * 32) {
        const int r = t >> 5, c = t & 31;
        if (r0 + r < L) { i = (r0 + r) * 32 + c; p = a.z + i; gr = dzs[r * 32 + c] + a.w_vposer * 2.f * a.z[i] / (float)nz; }
    } else if (t < DR * 32 + DR * 18) {
        const int e = t - DR * 32, r = e / 18, q = e - r * 18, v = q / 9, d = q - v * 9;
        if (r0 + r < L) {
            if (d < 6) { const int j = (v * L + r0 + r) * 6 + d; i = nz + j; p = a.phi + j; gr = a.dphi[j]; }
            else { const int j = (v * L + r0 + r) * 3 + d - 6; i = nz + nphi + j; p = a.tau + j; gr = a.dtau[j]; }
        }
    } else if (t < DR * 32 + DR * 18 + 10 && blockIdx.x == 0) {
        const int k = t - DR * 32 - DR * 18;
        i = nz + nphi + ntau + k; p = a.beta + k;
        for (int f0 = 0; f0 < L; f0 += 16) {                 // loads batched (independent), adds in frame order
            float pv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) pv[u] = f0 + u < L ? a.dbeta_part[(size_t)(f0 + u) * 10 + k] : 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) gr += pv[u];
        }
    }
    if (i >= 0) {
        float m = a.adam_m[i], v = a.adam_v[i];
        m = 0.9f * m + 0.1f * gr;
        v = 0.999f * v + 0.001f * gr * gr;
        a.adam_m[i] = m; a.adam_v[i] = v;
        const float bc1 = 1.f - powf(0.9f, (float)step), bc2 = 1.f - powf(0.999f, (float)step);
        *p -= (a.lr / bc1) * m / (sqrtf(v) / sqrtf(bc2) + 1e-8f);
        if (a.grad_out) a.grad_out[i] = gr;
    }
}

}  // namespace

hipError_t ap_launch_fit_linear(const float* X, int ldx, int K, const float* Wt, int ldw, const float* bias, const float* G,
                                int ldg, float* Y, int ldy, int L, int N, int act, hipStream_t st) {
    const dim3 grid((N + 63) / 64, (L + 15) / 16);
    if (act) hipLaunchKernelGGL(fit_linear_kernel<1>, grid, dim3(256), 0, st, X, ldx, K, Wt, ldw, bias, G, ldg, Y, ldy, L, N);
    else hipLaunchKernelGGL(fit_linear_kernel<0>, grid, dim3(256), 0, st, X, ldx, K, Wt, ldw, bias, G, ldg, Y, ldy, L, N);
    return hipGetLastError();
}
hipError_t ap_launch_fit_aa(const float* O, int ldo, float* aa, int L, hipStream_t st) {
    hipLaunchKernelGGL(fit_aa_kernel, dim3((L * NB + 127) / 128), dim3(128), 0, st, O, ldo, aa, L);
    return hipGetLastError();
}
hipError_t ap_launch_fit_frame(const FitArgs& a, int it, hipStream_t st) {
    hipLaunchKernelGGL(fit_frame_kernel, dim3(a.L), dim3(64), 0, st, a, it);
    return hipGetLastError();
}
hipError_t ap_launch_fit_adam(const FitArgs& a, int step, int with_z, hipStream_t st) {
    const int total = a.L * 32 + 2 * a.L * 9 + 10;
    hipLaunchKernelGGL(fit_adam_kernel, dim3((total + 255) / 256), dim3(256), 0, st, a, step, with_z);
    return hipGetLastError();
}

hipError_t ap_launch_fit_decode(const float* z, int L, const float* w1t, const float* b1, const float* w2t, const float* b2,
                                const float* w3t, const float* b3, float* H1, float* H2, float* O, float* aa, hipStream_t st) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(fit_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FIT_LDS_FLOATS * 4);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(fit_decode_kernel, dim3((L + DR - 1) / DR), dim3(1024), FIT_LDS_FLOATS * 4, st, z, L, w1t, b1, w2t, b2, w3t, b3, H1, H2, O, aa);
    return hipGetLastError();
}
hipError_t ap_launch_fit_backprop_adam(const FitArgs& a, const float* w3, const float* w2, const float* w1, const float* H1,
                                       const float* H2, int step, hipStream_t st) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(fit_backprop_adam_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FIT_LDS_FLOATS * 4);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(fit_backprop_adam_kernel, dim3((a.L + DR - 1) / DR), dim3(1024), FIT_LDS_FLOATS * 4, st, a, w3, w2, w1, H1, H2, step);
    return hipGetLastError();
}
