// SMPL-X forward kernels for gfx950: pose prep + kinematic chain, sparse 4-bone skinning with the
// root transform fused, joint/landmark gather + pin-hole projection; plus the stand-alone geometry
// helpers.  The dense blend-shape contraction runs on the fp32 MFMA GEMM (conv_igemm.hip).
//
// Semantics restated from upstream smplx 0.1.28 (lbs.lbs / batch_rigid_transform /
// vertices2landmarks, SMPLX.forward, VertexJointSelector) as called by the reference at
// copenet/src/copenet/copenet_twoview.py:237-246 (SMPLX.forward + transform_smpl) and :307-311
// (perspective_projection); rot6d_to_rotmat: copenet/src/copenet/utils/geometry.py:47-61;
// transform_smpl: copenet/src/copenet/utils/utils.py:237-256.
#include "ap_common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ void rot6d_dev(const float* __restrict__ x, float* R) {
    // six numbers = row-major 3x2: a1 = x[0],x[2],x[4]; a2 = x[1],x[3],x[5]   (geometry.py:55-57)
    const float a1x = x[0], a1y = x[2], a1z = x[4], a2x = x[1], a2y = x[3], a2z = x[5];
    const float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-12f);
    const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
    const float d = b1x * a2x + b1y * a2y + b1z * a2z;
    const float ux = a2x - d * b1x, uy = a2y - d * b1y, uz = a2z - d * b1z;
    const float n2 = fmaxf(sqrtf(ux * ux + uy * uy + uz * uz), 1e-12f);
    const float b2x = ux / n2, b2y = uy / n2, b2z = uz / n2;
    const float b3x = b1y * b2z - b1z * b2y, b3y = b1z * b2x - b1x * b2z, b3z = b1x * b2y - b1y * b2x;
    R[0] = b1x; R[1] = b2x; R[2] = b3x;       // columns b1 b2 b3
    R[3] = b1y; R[4] = b2y; R[5] = b3y;
    R[6] = b1z; R[7] = b2z; R[8] = b3z;
}

// ------------------------------------------------------------------------------------------------
// One wave per body, lane = joint.  Builds the blend-shape coefficient row, the rest joints, runs the
// kinematic chain level by level through LDS and writes the rest-pose-removed bone transforms.
__global__ void __launch_bounds__(64) smplx_prep_kernel(const SmplxModelDev m, const SmplxFwdArgs a) {
    const int b = blockIdx.x, j = threadIdx.x;
    __shared__ float G[64][12];
    __shared__ float Jr[64][3];
    __shared__ float cf[20];
    float* coef = a.coef + (size_t)b * m.ncoef;
    // test-mode input mesh (copenet_twoview.py:258-279): body sb's rotations, zero betas, [I | in_smpltrans]
    const bool inmesh = a.n_main > 0 && b >= a.n_main;
    const int sb = inmesh ? b - a.n_main : b;
    // the blend-shape contraction runs on the bf16 matrix pipe in split-bf16 form (m.coef_split): the coefficient row is
    // written as (hi, lo) pairs, 4 bytes per coefficient like the fp32 it replaces
    auto put = [&](int i, float v) {
        if (m.coef_split) {                                 // planar: group of 8 coefficients = 8 hi | 8 lo (bf16)
            bf16_t* c16 = (bf16_t*)coef;
            const bf16_t hi = f32_to_bf16(v);
            c16[split_hi_pos(i)] = hi;
            c16[split_lo_pos(i)] = f32_to_bf16(v - bf16_to_f32(hi));
        } else {
            coef[i] = v;
        }
    };
    if (j < 20) {
        float c = j < 10 ? a.betas[(size_t)sb * 10 + j] : (a.expression ? a.expression[(size_t)sb * 10 + j - 10] : 0.f);
        if (inmesh) c = 0.f;
        cf[j] = c;
        put(j, c);
    }
    // pad coefficients: through put(), so that in split mode both bf16 halves of every pad slot are written (the
    // workspace comes from a raw hipMalloc and 0 x NaN would poison every vertex of the body) and no live slot is touched
    for (int i = 20 + (m.J - 1) * 9 + j; i < m.ncoef; i += 64) put(i, 0.f);

    float R[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
    if (a.pose6d) {
        if (j < 22) {
            float Rj[9];
            rot6d_dev(a.pose6d + (size_t)sb * a.pose6d_ld + 6 * j, Rj);
            if (a.rotmat_out && !inmesh)
                for (int e = 0; e < 9; ++e) a.rotmat_out[((size_t)b * 22 + j) * 9 + e] = Rj[e];
            if (j == 0) {
                // root 6D is the transform_smpl rotation; the chain root stays identity (copenet_twoview.py:237-243)
                float* P = a.post + (size_t)b * 12;
                for (int rr = 0; rr < 3; ++rr) {
                    float t = 0.f;
                    if (inmesh) {
                        t = a.in_trans[(size_t)sb * 3 + rr];
                    } else if (a.post_t) {
                        t = a.post_t[(size_t)b * a.post_t_ld + rr];
                        if (a.pose_rw) {                    // pred_smpltrans /= trans_scale, in place on pred_pose (:214-218)
                            t = t / a.trans_scale;
                            a.pose_rw[(size_t)b * a.post_t_ld + rr] = t;
                        }
                    }
                    P[rr * 4 + 0] = inmesh ? (rr == 0 ? 1.f : 0.f) : Rj[rr * 3 + 0];
                    P[rr * 4 + 1] = inmesh ? (rr == 1 ? 1.f : 0.f) : Rj[rr * 3 + 1];
                    P[rr * 4 + 2] = inmesh ? (rr == 2 ? 1.f : 0.f) : Rj[rr * 3 + 2];
                    P[rr * 4 + 3] = t;
                }
                if (a.cc_ws && !inmesh) {                   // camera_center = intr[:, :2, 2] of this body's view (:311,317)
                    const int half = a.n_main / 2;
                    const float* K = (b < half ? a.intr0 + (size_t)b * 9 : a.intr1 + (size_t)(b - half) * 9);
                    a.cc_ws[(size_t)b * 2 + 0] = K[2];
                    a.cc_ws[(size_t)b * 2 + 1] = K[5];
                }
            } else {
                for (int e = 0; e < 9; ++e) R[e] = Rj[e];
            }
        }
    } else {
        const float* src = nullptr;
        if (j == 0) src = a.global_orient ? a.global_orient + (size_t)b * 9 : nullptr;
        else if (j < 22) src = a.body_pose + ((size_t)b * 21 + (j - 1)) * 9;
        else if (j < m.J && a.extra_pose) src = a.extra_pose + ((size_t)b * (m.J - 22) + (j - 22)) * 9;
        if (src)
            for (int e = 0; e < 9; ++e) R[e] = src[e];
        if (j == 0 && a.post) {
            float* P = a.post + (size_t)b * 12;
            for (int e = 0; e < 12; ++e)
                P[e] = a.post_rt ? a.post_rt[(size_t)b * 12 + e] : ((e == 0 || e == 5 || e == 10) ? 1.f : 0.f);
        }
    }
    if (j >= 1 && j < m.J) {
        for (int e = 0; e < 9; ++e) put(20 + (j - 1) * 9 + e, R[e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f));
    }
    __syncthreads();
    if (j < m.J) {
        for (int c = 0; c < 3; ++c) {
            const float* sd = m.j_shapedirs + ((size_t)j * 3 + c) * 20;
            float acc = 0.f;
            for (int l = 0; l < 20; ++l) acc = fmaf(sd[l], cf[l], acc);
            Jr[j][c] = m.j_template[j * 3 + c] + acc;
        }
    }
    __syncthreads();
    const int par = (j < m.J && j > 0) ? m.parents[j] : 0;
    const int dep = j < m.J ? m.depth[j] : -1;
    float rel[3] = {0.f, 0.f, 0.f};
    if (j < m.J) {
        for (int c = 0; c < 3; ++c) rel[c] = j == 0 ? Jr[0][c] : Jr[j][c] - Jr[par][c];
        if (j == 0) {
            for (int rr = 0; rr < 3; ++rr) {
                G[0][rr * 4 + 0] = R[rr * 3 + 0]; G[0][rr * 4 + 1] = R[rr * 3 + 1]; G[0][rr * 4 + 2] = R[rr * 3 + 2];
                G[0][rr * 4 + 3] = rel[rr];
            }
        }
    }
    __syncthreads();
    for (int d = 1; d <= m.max_depth; ++d) {
        if (dep == d) {
            float P[12];
            for (int e = 0; e < 12; ++e) P[e] = G[par][e];
            for (int rr = 0; rr < 3; ++rr) {
                const float p0 = P[rr * 4 + 0], p1 = P[rr * 4 + 1], p2 = P[rr * 4 + 2], p3 = P[rr * 4 + 3];
                G[j][rr * 4 + 0] = p0 * R[0] + p1 * R[3] + p2 * R[6];
                G[j][rr * 4 + 1] = p0 * R[1] + p1 * R[4] + p2 * R[7];
                G[j][rr * 4 + 2] = p0 * R[2] + p1 * R[5] + p2 * R[8];
                G[j][rr * 4 + 3] = p0 * rel[0] + p1 * rel[1] + p2 * rel[2] + p3;
            }
        }
        __syncthreads();
    }
    if (j < m.J) {
        float* Aj = a.A + ((size_t)b * m.J + j) * 12;
        float* jp = a.jposed + ((size_t)b * m.J + j) * 3;
        for (int rr = 0; rr < 3; ++rr) {
            const float g0 = G[j][rr * 4 + 0], g1 = G[j][rr * 4 + 1], g2 = G[j][rr * 4 + 2], g3 = G[j][rr * 4 + 3];
            Aj[rr * 4 + 0] = g0; Aj[rr * 4 + 1] = g1; Aj[rr * 4 + 2] = g2;
            Aj[rr * 4 + 3] = g3 - (g0 * Jr[j][0] + g1 * Jr[j][1] + g2 * Jr[j][2]);
            jp[rr] = g3;
        }
    }
}

// ------------------------------------------------------------------------------------------------
template <int KB>
__device__ __forceinline__ void skin_point(const float* __restrict__ A, const int* __restrict__ idx,
                                           const float* __restrict__ w, float x, float y, float z, float* out) {
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        const float wk = w[k];
        const float* Ak = A + idx[k] * 12;
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = fmaf(wk, Ak[e], T[e]);
    }
    out[0] = T[0] * x + T[1] * y + T[2] * z + T[3];
    out[1] = T[4] * x + T[5] * y + T[6] * z + T[7];
    out[2] = T[8] * x + T[9] * y + T[10] * z + T[11];
}

__device__ __forceinline__ void skin_point_dyn(const float* __restrict__ A, const int* __restrict__ idx,
                                               const float* __restrict__ w, int K, float x, float y, float z,
                                               float* out) {
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) T[e] = 0.f;
    for (int k = 0; k < K; ++k) {
        const float wk = w[k];
        const float* Ak = A + idx[k] * 12;
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = fmaf(wk, Ak[e], T[e]);
    }
    out[0] = T[0] * x + T[1] * y + T[2] * z + T[3];
    out[1] = T[4] * x + T[5] * y + T[6] * z + T[7];
    out[2] = T[8] * x + T[9] * y + T[10] * z + T[11];
}

__device__ __forceinline__ void apply_post(const float* __restrict__ P, float* v) {
    const float x = v[0], y = v[1], z = v[2];
    v[0] = P[0] * x + P[1] * y + P[2] * z + P[3];
    v[1] = P[4] * x + P[5] * y + P[6] * z + P[7];
    v[2] = P[8] * x + P[9] * y + P[10] * z + P[11];
}

constexpr int SKIN_BPB = 8;     // bodies per block: per-vertex weights stay in registers across them
constexpr int SKIN_MAXJ = 64;

// thread = vertex, loops over SKIN_BPB bodies whose bone transforms sit in LDS
template <int KB>
__global__ void __launch_bounds__(256) smplx_skin_kernel(const SmplxModelDev m, const SmplxFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float As[SKIN_BPB][SKIN_MAXJ * 12];
    __shared__ float Ps[SKIN_BPB][16];
    const int b0 = blockIdx.y * SKIN_BPB, nb = min(SKIN_BPB, a.n - b0);
    {   // the block's bone transforms are one contiguous run of nb * J * 12 floats (48 J bytes per body: 16-byte aligned
        // for every b0); all of a thread's loads are issued before the first LDS write (a rolled load -> wait -> store
        // loop is ~20 dependent L2 round trips here)
        const int n4 = nb * m.J * 3, J12 = m.J * 12;
        const float4* src = (const float4*)(a.A + (size_t)b0 * J12);
        constexpr int AIT = (SKIN_BPB * SKIN_MAXJ * 3 + 255) / 256;
        float4 t4[AIT];
#pragma unroll
        for (int k = 0; k < AIT; ++k) { const int i = threadIdx.x + k * 256; t4[k] = src[i < n4 ? i : n4 - 1]; }
#pragma unroll
        for (int k = 0; k < AIT; ++k) {
            const int i = threadIdx.x + k * 256;
            if (i < n4) {
                const int f = 4 * i, bb = f / J12, e = f - bb * J12;       // J12 is a multiple of 4: no straddling
                *(float4*)&As[bb][e] = t4[k];
            }
        }
    }
    for (int i = threadIdx.x; i < nb * 16; i += 256) {
        const int bb = i >> 4, e = i & 15;
        float v = 0.f;
        if (e < 12) v = a.post ? a.post[(size_t)(b0 + bb) * 12 + e] : ((e == 0 || e == 5 || e == 10) ? 1.f : 0.f);
        else if (e < 15) v = a.transl ? a.transl[(size_t)(b0 + bb) * 3 + (e - 12)] : 0.f;
        Ps[bb][e] = v;
    }
    __syncthreads();
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= m.V) return;
    int idx[KB > 0 ? KB : 1];
    float w[KB > 0 ? KB : 1];
    if constexpr (KB > 0) {
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            idx[k] = m.skin_idx[(size_t)v * KB + k];
            w[k] = m.skin_w[(size_t)v * KB + k];
        }
    }
    // running pointers (no 64-bit multiply per body) and the next body's point requested before this one is skinned:
    // the counters showed 186 VALU instructions per vertex-body and waves parked on memory 61 % of their cycles
    const float* vp = a.vposed + (size_t)b0 * m.ldv + 3 * (size_t)v;
    float* dst = a.vertices + ((size_t)b0 * m.V + v) * 3;
    const size_t dstep = (size_t)m.V * 3;
    float nx = vp[0], ny = vp[1], nz = vp[2];
    for (int bb = 0; bb < nb; ++bb) {
        const float x = nx, y = ny, z = nz;
        vp += m.ldv;
        if (bb + 1 < nb) { nx = vp[0]; ny = vp[1]; nz = vp[2]; }
        float o[3];
        if constexpr (KB > 0) skin_point<KB>(As[bb], idx, w, x, y, z, o);
        else skin_point_dyn(As[bb], m.skin_idx + (size_t)v * m.K, m.skin_w + (size_t)v * m.K, m.K, x, y, z, o);
        o[0] += Ps[bb][12]; o[1] += Ps[bb][13]; o[2] += Ps[bb][14];        // + transl (upstream SMPLX.forward)
        if (a.post) apply_post(Ps[bb], o);                                 // transform_smpl
        dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
        dst += dstep;
    }
}

// one block per body: 55 chain joints, 21 vertex picks, 51 barycentric landmarks, projection
__global__ void __launch_bounds__(128) smplx_joints_kernel(const SmplxModelDev m, const SmplxFwdArgs a) {
    __shared__ float As[SKIN_MAXJ * 12];
    __shared__ float Ps[16];
    const int b = blockIdx.x, t = threadIdx.x;
    {
        constexpr int AIT = (SKIN_MAXJ * 12 + 127) / 128;    // all of a thread's loads before its first LDS write
        const int n = m.J * 12;
        float av[AIT];
#pragma unroll
        for (int k = 0; k < AIT; ++k) av[k] = a.A[(size_t)b * n + min(t + k * 128, n - 1)];
#pragma unroll
        for (int k = 0; k < AIT; ++k) if (t + k * 128 < n) As[t + k * 128] = av[k];
    }
    if (t < 16) {
        float v = 0.f;
        if (t < 12) v = a.post ? a.post[(size_t)b * 12 + t] : ((t == 0 || t == 5 || t == 10) ? 1.f : 0.f);
        else if (t < 15) v = a.transl ? a.transl[(size_t)b * 3 + (t - 12)] : 0.f;
        Ps[t] = v;
    }
    __syncthreads();
    const int nj = m.J + m.n_extra + m.n_lmk;
    if (t >= nj) return;
    const float* vp = a.vposed + (size_t)b * m.ldv;
    float o[3];
    auto skin = [&](int v, float* out) {
        skin_point_dyn(As, m.skin_idx + (size_t)v * m.K, m.skin_w + (size_t)v * m.K, m.K, vp[3 * v], vp[3 * v + 1],
                       vp[3 * v + 2], out);
    };
    if (t < m.J) {
        for (int c = 0; c < 3; ++c) o[c] = a.jposed[((size_t)b * m.J + t) * 3 + c];
    } else if (t < m.J + m.n_extra) {
        skin(m.extra_verts[t - m.J], o);
    } else {
        const int l = t - m.J - m.n_extra;
        o[0] = o[1] = o[2] = 0.f;
        for (int f = 0; f < 3; ++f) {
            float p[3];
            skin(m.lmk_tri[l * 3 + f], p);
            const float bw = m.lmk_bary[l * 3 + f];
            o[0] = fmaf(p[0], bw, o[0]); o[1] = fmaf(p[1], bw, o[1]); o[2] = fmaf(p[2], bw, o[2]);
        }
    }
    o[0] += Ps[12]; o[1] += Ps[13]; o[2] += Ps[14];
    if (a.post) apply_post(Ps, o);
    float* dst = a.joints + ((size_t)b * nj + t) * 3;
    dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
    if (a.joints2d && a.cam_center) {
        // perspective_projection with R = I, t = 0 (geometry.py:63-91 as called at copenet_twoview.py:307-311)
        const float px = o[0] / o[2], py = o[1] / o[2];
        a.joints2d[((size_t)b * nj + t) * 2 + 0] = a.fx * px + a.cam_center[(size_t)b * 2 + 0];
        a.joints2d[((size_t)b * nj + t) * 2 + 1] = a.fy * py + a.cam_center[(size_t)b * 2 + 1];
    }
}

// ------------------------------------------------------------------------------------------------
__global__ void rot6d_kernel(const float* __restrict__ x6, int n, float* __restrict__ R) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r[9];
    rot6d_dev(x6 + (size_t)i * 6, r);
    for (int e = 0; e < 9; ++e) R[(size_t)i * 9 + e] = r[e];
}

// axis-angle -> rotation matrix, [n][3] -> [n][3][3], in the two forms the reference uses:
//   variant 0: smplx lbs.batch_rodrigues (the fork's `lbs` export, copenet/dsets/aerialpeople.py:177): angle = |r + 1e-8|,
//              K = skew(r / angle), R = I + sin(angle) K + (1 - cos(angle)) K K
//   variant 1: copenet/utils/geometry.py:9-45 batch_rodrigues: the same angle and axis through a re-normalised unit quaternion
__global__ void batch_rodrigues_kernel(const float* __restrict__ aa, int n, int variant, float* __restrict__ R) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float rx = aa[(size_t)i * 3], ry = aa[(size_t)i * 3 + 1], rz = aa[(size_t)i * 3 + 2];
    const float ex = rx + 1e-8f, ey = ry + 1e-8f, ez = rz + 1e-8f;
    const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
    const float dx = rx / angle, dy = ry / angle, dz = rz / angle;
    float* o = R + (size_t)i * 9;
    if (variant == 0) {
        const float s = sinf(angle), c1 = 1.f - cosf(angle);
        // K = [[0,-dz,dy],[dz,0,-dx],[-dy,dx,0]];  K K = d d^T - |d|^2 I
        const float dd = dx * dx + dy * dy + dz * dz;
        o[0] = 1.f + c1 * (dx * dx - dd);      o[1] = -s * dz + c1 * dx * dy;        o[2] = s * dy + c1 * dx * dz;
        o[3] = s * dz + c1 * dx * dy;          o[4] = 1.f + c1 * (dy * dy - dd);     o[5] = -s * dx + c1 * dy * dz;
        o[6] = -s * dy + c1 * dx * dz;         o[7] = s * dx + c1 * dy * dz;         o[8] = 1.f + c1 * (dz * dz - dd);
    } else {
        const float h = angle * 0.5f, sh = sinf(h);
        float w = cosf(h), x = sh * dx, y = sh * dy, z = sh * dz;
        const float qn = sqrtf(w * w + x * x + y * y + z * z);
        w /= qn; x /= qn; y /= qn; z /= qn;
        o[0] = w * w + x * x - y * y - z * z;  o[1] = 2 * x * y - 2 * w * z;         o[2] = 2 * w * y + 2 * x * z;
        o[3] = 2 * w * z + 2 * x * y;          o[4] = w * w - x * x + y * y - z * z; o[5] = 2 * y * z - 2 * w * x;
        o[6] = 2 * x * z - 2 * w * y;          o[7] = 2 * w * x + 2 * y * z;         o[8] = w * w - x * x - y * y + z * z;
    }
}

// rotation_matrix_to_angle_axis of torchgeometry 0.1.2 (rotation_matrix_to_quaternion on the TRANSPOSED matrix with its
// four trace branches, eps = 1e-6, then quaternion_to_angle_axis), as called at copenet_twoview.py:323-324
__global__ void rotmat_to_angle_axis_kernel(const float* __restrict__ R, int n, int ld, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = R + (size_t)i * ld * 3;                 // row stride ld (3 or 4)
    // rt[a][b] = r[b][a]
    const float t00 = r[0], t10 = r[1], t20 = r[2], t01 = r[ld], t11 = r[ld + 1], t21 = r[ld + 2],
                t02 = r[2 * ld], t12 = r[2 * ld + 1], t22 = r[2 * ld + 2];
    float q[4], t;
    if (t22 < 1e-6f) {
        if (t00 > t11) { t = 1 + t00 - t11 - t22; q[0] = t12 - t21; q[1] = t; q[2] = t01 + t10; q[3] = t20 + t02; }
        else           { t = 1 - t00 + t11 - t22; q[0] = t20 - t02; q[1] = t01 + t10; q[2] = t; q[3] = t12 + t21; }
    } else {
        if (t00 < -t11) { t = 1 - t00 - t11 + t22; q[0] = t01 - t10; q[1] = t20 + t02; q[2] = t12 + t21; q[3] = t; }
        else            { t = 1 + t00 + t11 + t22; q[0] = t; q[1] = t12 - t21; q[2] = t20 - t02; q[3] = t01 - t10; }
    }
    const float s = 0.5f / sqrtf(t);
    const float w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
    const float ss = x * x + y * y + z * z, sn = sqrtf(ss);
    const float two_theta = 2.0f * (w < 0.f ? atan2f(-sn, -w) : atan2f(sn, w));
    const float k = ss > 0.f ? two_theta / sn : 2.0f;
    out[(size_t)i * 3 + 0] = x * k;
    out[(size_t)i * 3 + 1] = y * k;
    out[(size_t)i * 3 + 2] = z * k;
}

__global__ void transform_points_kernel(const float* __restrict__ rt, const float* __restrict__ pts, int B, int P,
                                        float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * P) return;
    const float* M = rt + (i / P) * 12;
    float v[3] = {pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]};
    apply_post(M, v);
    out[i * 3] = v[0]; out[i * 3 + 1] = v[1]; out[i * 3 + 2] = v[2];
}

__global__ void projection_kernel(const float* __restrict__ pts, int B, int P, const float* __restrict__ R,
                                  const float* __restrict__ t, float fx, float fy, const float* __restrict__ center,
                                  float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * P) return;
    const size_t b = i / P;
    float x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    if (R) {
        const float* r = R + b * 9;
        const float nx = r[0] * x + r[1] * y + r[2] * z, ny = r[3] * x + r[4] * y + r[5] * z,
                    nz = r[6] * x + r[7] * y + r[8] * z;
        x = nx; y = ny; z = nz;
    }
    if (t) { x += t[b * 3]; y += t[b * 3 + 1]; z += t[b * 3 + 2]; }
    out[i * 2] = fx * (x / z) + center[b * 2];
    out[i * 2 + 1] = fy * (y / z) + center[b * 2 + 1];
}

}  // namespace

hipError_t ap_launch_smplx_prep(const SmplxModelDev& m, const SmplxFwdArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(smplx_prep_kernel, dim3(a.n), dim3(64), 0, st, m, a);
    return hipGetLastError();
}

hipError_t ap_launch_smplx_skin(const SmplxModelDev& m, const SmplxFwdArgs& a, hipStream_t st) {
    dim3 grid((m.V + 255) / 256, (a.n + SKIN_BPB - 1) / SKIN_BPB);
    if (m.K == 4) hipLaunchKernelGGL(smplx_skin_kernel<4>, grid, dim3(256), 0, st, m, a);
    else if (m.K == 8) hipLaunchKernelGGL(smplx_skin_kernel<8>, grid, dim3(256), 0, st, m, a);
    else hipLaunchKernelGGL(smplx_skin_kernel<0>, grid, dim3(256), 0, st, m, a);
    return hipGetLastError();
}

hipError_t ap_launch_smplx_joints(const SmplxModelDev& m, const SmplxFwdArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(smplx_joints_kernel, dim3(a.n_main > 0 ? a.n_main : a.n), dim3(128), 0, st, m, a);
    return hipGetLastError();
}

hipError_t ap_launch_batch_rodrigues(const float* aa, int n, int variant, float* R, hipStream_t st) {
    hipLaunchKernelGGL(batch_rodrigues_kernel, dim3((n + 255) / 256), dim3(256), 0, st, aa, n, variant, R);
    return hipGetLastError();
}

hipError_t ap_launch_rotmat_to_angle_axis(const float* R, int n, int ld, float* out, hipStream_t st) {
    hipLaunchKernelGGL(rotmat_to_angle_axis_kernel, dim3((n + 255) / 256), dim3(256), 0, st, R, n, ld, out);
    return hipGetLastError();
}

hipError_t ap_launch_rot6d(const float* x6, int n, float* R, hipStream_t st) {
    hipLaunchKernelGGL(rot6d_kernel, dim3((n + 255) / 256), dim3(256), 0, st, x6, n, R);
    return hipGetLastError();
}

hipError_t ap_launch_transform_points(const float* rt, const float* pts, int B, int P, float* out, hipStream_t st) {
    const size_t tot = (size_t)B * P;
    hipLaunchKernelGGL(transform_points_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, rt, pts, B, P, out);
    return hipGetLastError();
}

hipError_t ap_launch_projection(const float* pts, int B, int P, const float* R, const float* t, float fx, float fy,
                                const float* center, float* out, hipStream_t st) {
    const size_t tot = (size_t)B * P;
    hipLaunchKernelGGL(projection_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, pts, B, P, R, t, fx,
                       fy, center, out);
    return hipGetLastError();
}
