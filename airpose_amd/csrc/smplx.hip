// SMPL-X forward kernels for gfx950: pose prep + kinematic chain, sparse 4-bone skinning with the
// root transform fused, joint/landmark gather + pin-hole projection; plus the stand-alone geometry
// helpers.  The dense blend-shape contraction runs on the fp32 MFMA GEMM (conv_igemm.hip).
//
// Semantics restated from upstream smplx 0.1.28 (lbs.lbs / batch_rigid_transform /
// vertices2landmarks, SMPLX.forward, VertexJointSelector) as called by the reference at
// copenet/src/copenet/copenet_twoview.py:237-246 (SMPLX.forward + transform_smpl) and :307-311
// (perspective_projection); rot6d_to_rotmat: copenet/src/copenet/utils/geometry.py:47-61;
// transform_smpl: copenet/src/copenet/utils/utils.py:237-256.
#include <type_traits>

#include "ap_common.h"
#include "kernels.h"

namespace {

template <int I, int N, typename F> __device__ __forceinline__ void lf_sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        lf_sfor<I + 1, N>(f);
    }
}

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
    __shared__ float Psh[12];                                // the body's post transform (root lane), for A22
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
    // ---- every global load of the wave is requested here, before the first result is used (as written top-down the kernel
    // was six dependent memory round trips: coefficients -> pose -> translation -> parents -> joint shape directions -> ...).
    // Pointers are selected, loads unconditional (a load inside a branch is waited for at the branch's end); lanes without a
    // value of their own read betas[sb][0..9], which every call supplies.
    const float* safe = a.betas + (size_t)sb * 10;
    const int jc = j < m.J ? j : m.J - 1;
    const float c_in = *(j < 10 ? safe + j : (a.expression && j < 20) ? a.expression + (size_t)sb * 10 + (j - 10) : safe);
    float x9[9];                                             // pose6d: six numbers of joint j (lanes 0..21); else: its 3x3 rotation
    const float* src9 = safe;
    bool has9 = false;
    if (a.pose6d) {
        if (j < 22) src9 = a.pose6d + (size_t)sb * a.pose6d_ld + 6 * j;
    } else {
        if (j == 0) { if (a.global_orient) { src9 = a.global_orient + (size_t)b * 9; has9 = true; } }
        else if (j < 22) { src9 = a.body_pose + ((size_t)b * 21 + (j - 1)) * 9; has9 = true; }
        else if (j < m.J && a.extra_pose) { src9 = a.extra_pose + ((size_t)b * (m.J - 22) + (j - 22)) * 9; has9 = true; }
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) x9[e] = (e < 6 || !a.pose6d) ? src9[e] : 0.f;
    // root lane: translation of the post transform, camera centre (copenet_twoview.py:237-243, :311,317)
    const float* tsrc = !a.pose6d ? nullptr : inmesh ? a.in_trans + (size_t)sb * 3 : a.post_t ? a.post_t + (size_t)b * a.post_t_ld : nullptr;
    const float* tp = tsrc ? tsrc : safe;
    const float l0 = tp[0], l1 = tp[1], l2 = tp[2];
    const bool cc = a.pose6d && a.cc_ws && !inmesh;
    const int half = a.n_main / 2;
    const float* Kc = !cc ? safe : (b < half ? a.intr0 + (size_t)b * 9 : a.intr1 + (size_t)(b - half) * 9);
    const float ccx = Kc[2], ccy = Kc[5];
    float prt[12];                                           // caller-supplied post transform (SMPLX.forward path)
#pragma unroll
    for (int e = 0; e < 12; ++e) prt[e] = (!a.pose6d && a.post && a.post_rt) ? a.post_rt[(size_t)b * 12 + e] : ((e == 0 || e == 5 || e == 10) ? 1.f : 0.f);
    const int par_in = m.parents[jc], dep_in = m.depth[jc];
    float jt_in[3], sd_in[3][20];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        jt_in[c] = m.j_template[jc * 3 + c];
#pragma unroll
        for (int l = 0; l < 20; ++l) sd_in[c][l] = m.j_shapedirs[((size_t)jc * 3 + c) * 20 + l];
    }

    if (j < 20) {
        const float c = (inmesh || (j >= 10 && !a.expression)) ? 0.f : c_in;
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
            rot6d_dev(x9, Rj);
            if (j == 0) {
                // root 6D is the transform_smpl rotation; the chain root stays identity (copenet_twoview.py:237-243)
                float* P = a.post + (size_t)b * 12;
                float t3[3] = {0.f, 0.f, 0.f};
                if (tsrc) { t3[0] = l0; t3[1] = l1; t3[2] = l2; }
                const bool rw = !inmesh && a.post_t && a.pose_rw;
                if (rw) {                                   // pred_smpltrans /= trans_scale, in place on pred_pose (:214-218)
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr) t3[rr] = t3[rr] / a.trans_scale;
                }
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
                    if (rw) a.pose_rw[(size_t)b * a.post_t_ld + rr] = t3[rr];
                    P[rr * 4 + 0] = inmesh ? (rr == 0 ? 1.f : 0.f) : Rj[rr * 3 + 0];
                    P[rr * 4 + 1] = inmesh ? (rr == 1 ? 1.f : 0.f) : Rj[rr * 3 + 1];
                    P[rr * 4 + 2] = inmesh ? (rr == 2 ? 1.f : 0.f) : Rj[rr * 3 + 2];
                    P[rr * 4 + 3] = t3[rr];
                    Psh[rr * 4 + 0] = P[rr * 4 + 0]; Psh[rr * 4 + 1] = P[rr * 4 + 1]; Psh[rr * 4 + 2] = P[rr * 4 + 2]; Psh[rr * 4 + 3] = t3[rr];
                }
                if (cc) {
                    a.cc_ws[(size_t)b * 2 + 0] = ccx;
                    a.cc_ws[(size_t)b * 2 + 1] = ccy;
                }
            } else {
                for (int e = 0; e < 9; ++e) R[e] = Rj[e];
            }
            if (a.rotmat_out && !inmesh)
                for (int e = 0; e < 9; ++e) a.rotmat_out[((size_t)b * 22 + j) * 9 + e] = Rj[e];
        }
    } else {
        if (has9)
            for (int e = 0; e < 9; ++e) R[e] = x9[e];
        if (j == 0 && a.post) {
            float* P = a.post + (size_t)b * 12;
#pragma unroll
            for (int e = 0; e < 12; ++e) { P[e] = prt[e]; Psh[e] = prt[e]; }
        }
    }
    if (j >= 1 && j < m.J) {
        for (int e = 0; e < 9; ++e) put(20 + (j - 1) * 9 + e, R[e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f));
    }
    __syncthreads();
    if (j < m.J) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int l = 0; l < 20; ++l) acc = fmaf(sd_in[c][l], cf[l], acc);
            Jr[j][c] = jt_in[c] + acc;
        }
    }
    __syncthreads();
    const int par = (j < m.J && j > 0) ? par_in : 0;
    const int dep = j < m.J ? dep_in : -1;
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
        float Av[12];
        for (int rr = 0; rr < 3; ++rr) {
            const float g0 = G[j][rr * 4 + 0], g1 = G[j][rr * 4 + 1], g2 = G[j][rr * 4 + 2], g3 = G[j][rr * 4 + 3];
            Av[rr * 4 + 0] = g0; Av[rr * 4 + 1] = g1; Av[rr * 4 + 2] = g2;
            Av[rr * 4 + 3] = g3 - (g0 * Jr[j][0] + g1 * Jr[j][1] + g2 * Jr[j][2]);
            Aj[rr * 4 + 0] = Av[rr * 4 + 0]; Aj[rr * 4 + 1] = Av[rr * 4 + 1]; Aj[rr * 4 + 2] = Av[rr * 4 + 2]; Aj[rr * 4 + 3] = Av[rr * 4 + 3];
            jp[rr] = g3;
        }
        // the posed body transforms with the caller's post transform composed in (P o A_j; skinning weights sum to one, so
        // sum_j w_j (P o A_j) = P o sum_j w_j A_j): what the vertex-stationary kernel skins with -- no per-vertex post transform
        if (a.A22 && j < 22) {
            float* Bj = a.A22 + ((size_t)b * 22 + j) * 12;
            for (int rr = 0; rr < 3; ++rr) {
                const float p0 = Psh[rr * 4 + 0], p1 = Psh[rr * 4 + 1], p2 = Psh[rr * 4 + 2], p3 = Psh[rr * 4 + 3];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    Bj[rr * 4 + c] = p0 * Av[c] + p1 * Av[4 + c] + p2 * Av[8 + c] + (c == 3 ? p3 : 0.f);
            }
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

// ------------------------------------------------------------------------------------------------
// Blend-shape contraction AND skinning in one kernel: v_posed never leaves the chip (the two-kernel path writes 63 MB of it
// at 512 bodies and reads them back: 3.6x the algorithmic traffic of the tail).  upstream lbs.lbs:
//   v_posed = v_template + [betas | expr | pose_feature] . dirs;  T_v = sum_j w[v][j] A_j;  verts = T_v [v_posed; 1]
// (+ transl, then transform_smpl of the caller, copenet_twoview.py:237-246).
// (The round-3 first cut -- rows = vertices, eight waves with a hand-counted register ring -- was retired in round 5: git history.)
// The kernel (round 4).  What the first cut's profile said
// (profiles/r03_lbs_fused_ablation.txt, PMC): every unit at a quarter of its rate, WRITE_SIZE twice the vertex bytes, and an
// intermittent wrong vertex in one build of it (hand-counted asm loads with loop-carried destinations).  Changes:
//   * ORIENTATION: rows = bodies, columns = vertices (A = coefficient rows from LDS, B = direction fragments from L2 -- the same
//     dirs_frag buffer: A and B fragments of v_mfma_f32_16x16x32 share their (16 x 8-k) lane layout).  A lane then holds (x, y, z)
//     of ONE vertex (lr) for 8 bodies, so (i) its skinning operands (bone ids, weights, template) are loaded once per group and
//     reused for the 8 bodies, and (ii) a store instruction writes, per body, the 16 consecutive vertices of the group: 192
//     contiguous bytes per 16-lane group instead of 12-byte pieces to 32 different bodies -- lines complete within one or two
//     back-to-back instructions instead of being evicted half-written;
//   * K = 224 (7 steps): the eighth step (jaw / eye features: identically zero without face poses) is not multiplied;
//   * SIXTEEN waves per workgroup (128 registers each) instead of eight: latency is covered by occupancy, not by a hand-counted
//     register ring -- every load is a plain load the compiler counts (no inline-asm destinations, nothing loop-carried in
//     flight), the next K step's fragments are requested before the current step's MFMAs.
// Same arithmetic per product as the first cut and the two-kernel path (hi.hi + lo.hi + hi.lo on the bf16 pipe, fp32 skinning).
constexpr int T_KS = 7, T_NW = 16, T_CROW = T_KS * 128 + 16, T_MAXJ = 55;
// Timing-only builds (results WRONG, times valid): -DT_ABLATE=<bits>: 1 no vertex stores | 2 no bone gathers / blend | 4 direction
// fragments loaded once per group instead of once per K step | 8 no MFMAs | 16 no prologue copies
#ifndef T_ABLATE
#define T_ABLATE 0
#endif

// NBJ = bone transforms per body held in LDS: T_MAXJ (all joints, skin_idx8 / skin_w4) or 22 (body-only calls: the merged table
// skin_idx8b / skin_w4b over root + 21 body joints -- every hand / face joint skins like its posed ancestor; 34 instead of 84 KB
// of bone tables, and a vertex gathers only the transforms that differ: zero weights are skipped)
// BB = bodies per workgroup: 32, or 64 = two halves of 32 handled by wave pairs (w, w + 8) that walk the SAME vertex groups: the
// direction fragments -- the kernel's L2 stream, 42 KB per group of 16 vertices -- are then requested by both waves of a pair at
// about the same time, and half as many body groups pull the 27.6 MB of fragments through L2
template <int NBJ, int BB>
__global__ void __launch_bounds__(64 * T_NW) smplx_lbs_tail_kernel(const SmplxModelDev m, const SmplxFwdArgs a, int n_vr, int groups_per_vr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lf_smem[];
    constexpr bool MERGED = NBJ < T_MAXJ;
    float* bones = (float*)lf_smem;                                               // [BB][NBJ][12]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, g4 = lane >> 4;
    constexpr int GW = T_NW / (BB / 32);                                          // waves that share the groups of a body half
    const int hb = wave / GW;                                                     // this wave's half of the workgroup's bodies
    const int vr = blockIdx.x % n_vr, bg = blockIdx.x / n_vr;
    const int b0 = bg * BB, J12 = m.J * 12;
    // bone tables at the compile-time stride [BB][T_MAXJ][12]: a body's table is then an immediate offset from ONE per-lane base
    // (with the runtime stride J * 12 hipcc kept eight per-body addresses in registers across the group loop)
    constexpr int J12C = NBJ * 12;
    unsigned char* coefs = lf_smem + BB * J12C * 4;                             // [BB][T_CROW]: 7 K steps x (4 x [8 hi | 8 lo])
    float* Ps = (float*)(coefs + BB * T_CROW);                                  // [BB][16]: post transform [12] | translation [3]
    const int ngroups = (m.V + 15) >> 4;
    const int gend = min(ngroups, (vr + 1) * groups_per_vr);
    const unsigned char* dbase = (const unsigned char*)m.dirs_frag + (size_t)lane * 16;
    // fragment block of (group, K step, component c, plane): ((g * 8 + ks) * 3 + c) * 2 + plane, 1 KiB each (K = 256 layout)
    auto frag = [&](int g, int ks, int i) { return *(const u32x4*)(dbase + ((size_t)g * 8 + ks) * 6144 + i * 1024); };
    // The direction fragments of a K step are requested ONE STEP AHEAD into the other of two register sets (the step after a
    // group's last one is the next group's first: its loads fly under the skinning), all six of a step back to back: with one set
    // hipcc serialised load -> s_waitcnt vmcnt(0) -> MFMA pairs inside a step (three to four dependent L2 round trips per K step).
    // Plain loads, counted by the compiler; the address of a prefetch past the last group is clamped to the group itself.
    int g = vr * groups_per_vr + wave % GW;
    u32x4 fa[6], fb[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) fa[i] = frag(min(g, ngroups - 1), 0, i);           // (issued ahead of the prologue's own loads)
    {   // bone transforms, coefficient rows, post transforms of the 32 bodies (rows past the last body: clamped duplicates, never
        // stored); every load of a thread is issued before its first LDS write
        constexpr int NT = 64 * T_NW, BIT = (BB * NBJ * 3 + NT - 1) / NT, CCH = T_KS * 8, CIT = (BB * CCH + NT - 1) / NT;
        const int jl = MERGED ? NBJ : m.J;                   // joints of a body that go to LDS (the first jl of its J)
        const int nb = min(BB, a.n - b0), n4 = nb * jl * 3, tot = BB * jl * 3;
        // MERGED with a.A22: the 22 posed transforms with the post transform composed in (smplx_prep_kernel): one contiguous run
        const float4* src = (MERGED && a.A22) ? (const float4*)(a.A22 + (size_t)b0 * (NBJ * 12)) : (const float4*)(a.A + (size_t)b0 * J12);
        float4 tb[BIT];
        u32x4 tc[CIT];
#pragma unroll
        for (int k = 0; k < BIT; ++k) {
            const int i = tid + k * NT, ic = i < n4 ? i : i % n4;
            if constexpr (MERGED) { const int bb = ic / (NBJ * 3); tb[k] = a.A22 ? src[ic] : src[bb * (m.J * 3) + (ic - bb * (NBJ * 3))]; }
            else tb[k] = src[ic];
        }
#pragma unroll
        for (int k = 0; k < CIT; ++k) {
            const int i = min(tid + k * NT, BB * CCH - 1), bb = i / CCH, c16 = i - bb * CCH, bsrc = min(b0 + bb, a.n - 1);
            tc[k] = *(const u32x4*)((const unsigned char*)(a.coef + (size_t)bsrc * m.ncoef) + c16 * 16);
        }
        float pv = 0.f;
        if (tid < BB * 16) {
            const int bb = tid >> 4, e = tid & 15, bsrc = min(b0 + bb, a.n - 1);
            if (e < 12) pv = a.post ? a.post[(size_t)bsrc * 12 + e] : ((e == 0 || e == 5 || e == 10) ? 1.f : 0.f);
            else if (e < 15) pv = a.transl ? a.transl[(size_t)bsrc * 3 + (e - 12)] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < BIT; ++k) {
            const int i = tid + k * NT, J3 = jl * 3, bb = i / J3;
            if (i < tot) ((float4*)bones)[bb * (NBJ * 3) + (i - bb * J3)] = tb[k];
        }
#pragma unroll
        for (int k = 0; k < CIT; ++k) {
            const int i = tid + k * NT;
            if (i < BB * CCH) { const int bb = i / CCH, c16 = i - bb * CCH; *(u32x4*)(coefs + bb * T_CROW + c16 * 16) = tc[k]; }
        }
        if (tid < BB * 16) Ps[tid] = pv;
    }
    __syncthreads();
    const unsigned char* ca = coefs + (hb * 32 + lr) * T_CROW + g4 * 32;           // this lane's A rows: bodies lr and 16 + lr (of its half)
    const float* const bones_l = bones + (hb * 32 + g4 * 4) * J12C;                // this lane's skinning bodies: 4 g4 + r (+ 16)
    const float* const Ps_l = Ps + (hb * 32 + g4 * 4) * 16;
#ifdef AP_TRACE   // cycle stamps of wave 0 of workgroups 0 and 100, their SECOND vertex group (24 slots each): tools/probes/lbs_trace.py
    int grp_no = 0;
#define LSTAMP(i) do { if (a.dbg && grp_no == 1 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == 100)) { \
        const unsigned long long t_ = __builtin_readcyclecounter(); \
        if (lane == 0) a.dbg[(blockIdx.x ? 24 : 0) + (i)] = t_; } } while (0)
#else
#define LSTAMP(i) do { } while (0)
#endif
    for (; g < gend; g += GW) {
        LSTAMP(0);
        // ------------------------------------------------ contraction: acc[c][s][r] = v_posed component c of vertex 16 g + lr
        // for body s * 16 + 4 g4 + r
        f32x4 acc[3][2];
#pragma unroll
        for (int c = 0; c < 3; ++c) { acc[c][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[c][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        // this lane's vertex: bone ids (6 bits each, + joint-vertex slot), weights, template (requested under the contraction)
        const int v = g * 16 + lr;
        const uint32_t id = (MERGED ? m.skin_idx8b : m.skin_idx8)[v];
        const float4 w4 = *(const float4*)((MERGED ? m.skin_w4b : m.skin_w4) + (size_t)v * 4);
        const float tx = m.v_template[(size_t)v * 3], ty = m.v_template[(size_t)v * 3 + 1], tz = m.v_template[(size_t)v * 3 + 2];
        const int gn = g + GW < gend ? g + GW : g;
        lf_sfor<0, T_KS>([&](auto KS) {
            constexpr int ks = decltype(KS)::value;
            u32x4 (&cur)[6] = (ks & 1) ? fb : fa;
            u32x4 (&nxt)[6] = (ks & 1) ? fa : fb;
            if (!(T_ABLATE & 4)) {
#pragma unroll
                for (int i = 0; i < 6; ++i) nxt[i] = ks + 1 < T_KS ? frag(g, ks + 1, i) : frag(gn, 0, i);
            }
            __builtin_amdgcn_sched_barrier(0);               // requests first: hipcc otherwise sinks them behind the step's MFMAs
            const bf16x8 ah0 = __builtin_bit_cast(bf16x8, *(const u32x4*)(ca + ks * 128)), al0 = __builtin_bit_cast(bf16x8, *(const u32x4*)(ca + ks * 128 + 16));
            const bf16x8 ah1 = __builtin_bit_cast(bf16x8, *(const u32x4*)(ca + 16 * T_CROW + ks * 128));
            const bf16x8 al1 = __builtin_bit_cast(bf16x8, *(const u32x4*)(ca + 16 * T_CROW + ks * 128 + 16));
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const bf16x8 dh = __builtin_bit_cast(bf16x8, cur[2 * c]), dl = __builtin_bit_cast(bf16x8, cur[2 * c + 1]);
                if (T_ABLATE & 8) { asm volatile("" : "+v"(acc[c][0]), "+v"(acc[c][1]) : "v"(dh), "v"(dl), "v"(ah0), "v"(al0), "v"(ah1), "v"(al1)); continue; }
                acc[c][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, dh, acc[c][0], 0, 0, 0);
                acc[c][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, dh, acc[c][1], 0, 0, 0);
                acc[c][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al0, dh, acc[c][0], 0, 0, 0);
                acc[c][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al1, dh, acc[c][1], 0, 0, 0);
                acc[c][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, dl, acc[c][0], 0, 0, 0);
                acc[c][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, dl, acc[c][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);               // keep the K steps in order: nothing of step ks + 2 before step ks is done
            LSTAMP(1 + ks);
        });
        // ------------------------------------------------ skinning of the lane's vertex for its 8 bodies
        const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
        const bool vok = v < m.V;
        // store addresses: one uniform base per workgroup + a 32-bit lane offset that walks the lane's 8 bodies (the eight
        // 64-bit per-body pointers hipcc otherwise keeps across the whole group loop were spilled once the second fragment
        // set took their registers, and a scratch reload sits behind every earlier store of the wave)
        char* const vbase = (char*)(a.vertices + (size_t)b0 * m.V * 3);
        const uint32_t vstep = (uint32_t)m.V * 12u;
        uint32_t voff = ((uint32_t)(hb * 32 + g4 * 4) * (uint32_t)m.V + (uint32_t)v) * 12u;
        char* const sbase = (char*)(a.vp_side + (size_t)b0 * m.n_jv * 3);          // joint-vertex side buffer, same scheme
        const uint32_t sstep = (uint32_t)m.n_jv * 12u;
        uint32_t soff = ((uint32_t)(hb * 32 + g4 * 4) * (uint32_t)m.n_jv + ((id >> 24) - 1u)) * 12u;   // (only used when id >> 24 != 0)
        asm volatile("" : "+v"(voff), "+v"(soff));           // opaque: hipcc otherwise hoists eight per-body offsets out of the group loop (and spills them)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int bl = hb * 32 + s * 16 + g4 * 4 + r;
                const bool bok = b0 + bl < a.n;
                const float x = acc[0][s][r] + tx, y = acc[1][s][r] + ty, z = acc[2][s][r] + tz;
                if (!a.grp_cnt && a.vp_side && bok && (id >> 24)) {   // joint vertex: slot + 1 in the top byte (0 = none, padding rows too)
                    float* q = (float*)(sbase + soff);
                    q[0] = x; q[1] = y; q[2] = z;
                }
                const float* Ab = bones_l + (s * 16 + r) * J12C;
                float T[12];
#pragma unroll
                for (int e = 0; e < 12; ++e) T[e] = (T_ABLATE & 2) ? wv[e & 3] : 0.f;
#pragma unroll
                for (int k = 0; k < ((T_ABLATE & 2) ? 0 : 4); ++k) {
                    if (MERGED && k > 0 && wv[k] == 0.f) continue;               // (heaviest first: the zero weights are the last ones)
                    const float4* Ak = (const float4*)(Ab + ((id >> (6 * k)) & 0x3fu) * 12);
                    const float4 r0 = Ak[0], r1 = Ak[1], r2 = Ak[2];
                    T[0] = fmaf(wv[k], r0.x, T[0]); T[1] = fmaf(wv[k], r0.y, T[1]); T[2] = fmaf(wv[k], r0.z, T[2]); T[3] = fmaf(wv[k], r0.w, T[3]);
                    T[4] = fmaf(wv[k], r1.x, T[4]); T[5] = fmaf(wv[k], r1.y, T[5]); T[6] = fmaf(wv[k], r1.z, T[6]); T[7] = fmaf(wv[k], r1.w, T[7]);
                    T[8] = fmaf(wv[k], r2.x, T[8]); T[9] = fmaf(wv[k], r2.y, T[9]); T[10] = fmaf(wv[k], r2.z, T[10]); T[11] = fmaf(wv[k], r2.w, T[11]);
                }
                const float* Pb = Ps_l + (s * 16 + r) * 16;
                float q[3];
                q[0] = T[0] * x + T[1] * y + T[2] * z + T[3];
                q[1] = T[4] * x + T[5] * y + T[6] * z + T[7];
                q[2] = T[8] * x + T[9] * y + T[10] * z + T[11];
                if (a.grp_cnt && bok && (id >> 24)) {        // the SKINNED joint vertex (before translation / post transform: what
                    float* sq = (float*)(sbase + soff);       // skin_point hands the joints stage), for the group's last workgroup:
                    // write-through (sc1) stores, so that publishing needs no L2 write-back of the 250 KB of vertices beside them
                    __hip_atomic_store(sq + 0, q[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(sq + 1, q[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(sq + 2, q[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                soff += sstep;
                if (!(MERGED && a.A22)) {                    // (A22: the bones carry the post transform; such calls have no translation)
                    q[0] += Pb[12]; q[1] += Pb[13]; q[2] += Pb[14];
                    if (a.post) apply_post(Pb, q);
                }
                if (T_ABLATE & 1) asm volatile("" ::"v"(q[0]), "v"(q[1]), "v"(q[2]));
                else if (bok && vok) {
                    float* dst = (float*)(vbase + voff);
                    dst[0] = q[0]; dst[1] = q[1]; dst[2] = q[2];
                }
                voff += vstep;
                __builtin_amdgcn_sched_barrier(0);           // one body at a time: the 8 bodies' bone rows all in flight spill
                LSTAMP(8 + s * 4 + r);
            }
            voff += 12u * vstep;                             // bodies 16 + 4 g4 ...
            soff += 12u * sstep;
        }
        static_assert(T_KS & 1, "an odd number of K steps leaves the next group's first fragments in the second set");
#pragma unroll
        for (int i = 0; i < 6; ++i) fa[i] = fb[i];           // the next group's first step (requested before the skinning)
        LSTAMP(16);
#ifdef AP_TRACE
        ++grp_no;
#endif
    }
    // ---------------------------------------------------- joints, landmarks and projection of the 32 bodies, by the LAST of the
    // body group's n_vr workgroups (smplx_joints_kernel's arithmetic on the skinned joint vertices the workgroups left in the
    // side buffer): one launch less per forward.  Hand-off: write-through (sc1) payload stores -> every wave drains -> barrier ->
    // one lane takes a ticket (relaxed, agent scope); the last ticket acquires (one L1 invalidate) and reads with plain loads.
    // (A release fence per workgroup instead -- buffer_wbl2 over its 250 KB of freshly written vertices -- cost 13 us per launch.)
    if (a.grp_cnt) {
        __shared__ int s_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(a.grp_cnt + bg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = old == n_vr - 1;
        }
        __syncthreads();
        if (!s_last) return;
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(a.grp_cnt + bg, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
        __syncthreads();
        const int nj = m.J + m.n_extra + m.n_lmk;
        const int nbj = a.n_main > 0 ? a.n_main : a.n;       // bodies that have joints (the test-mode input meshes do not)
        for (int item = tid; item < BB * nj; item += 64 * T_NW) {
            const int bl = item / nj, t = item - bl * nj, b = b0 + bl;
            if (b >= nbj) continue;
            const float* vs = a.vp_side + (size_t)b * m.n_jv * 3;
            const float* Pb = Ps + bl * 16;
            float o[3];
            if (t < m.J) {
                for (int c = 0; c < 3; ++c) o[c] = a.jposed[((size_t)b * m.J + t) * 3 + c];
            } else if (t < m.J + m.n_extra) {
                const float* q = vs + 3 * m.jv_slot[m.extra_verts[t - m.J]];
                o[0] = q[0]; o[1] = q[1]; o[2] = q[2];
            } else {
                const int l = t - m.J - m.n_extra;
                int vid[3];
                float bw[3];
#pragma unroll
                for (int f = 0; f < 3; ++f) { vid[f] = m.lmk_tri[l * 3 + f]; bw[f] = m.lmk_bary[l * 3 + f]; }
                int slot[3];
#pragma unroll
                for (int f = 0; f < 3; ++f) slot[f] = m.jv_slot[vid[f]];
                o[0] = o[1] = o[2] = 0.f;
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    const float* q = vs + 3 * slot[f];
                    o[0] = fmaf(q[0], bw[f], o[0]); o[1] = fmaf(q[1], bw[f], o[1]); o[2] = fmaf(q[2], bw[f], o[2]);
                }
            }
            o[0] += Pb[12]; o[1] += Pb[13]; o[2] += Pb[14];
            if (a.post) apply_post(Pb, o);
            float* dst = a.joints + ((size_t)b * nj + t) * 3;
            dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
            if (a.joints2d && a.cam_center) {
                const float px = o[0] / o[2], py = o[1] / o[2];
                a.joints2d[((size_t)b * nj + t) * 2 + 0] = a.fx * px + a.cam_center[(size_t)b * 2 + 0];
                a.joints2d[((size_t)b * nj + t) * 2 + 1] = a.fy * py + a.cam_center[(size_t)b * 2 + 1];
            }
        }
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
    const float* vs = a.vp_side ? a.vp_side + (size_t)b * m.n_jv * 3 : nullptr;    // fused path: [slot][3], slot = m.jv_slot[v]
    float o[3];
    if (vs && m.K == 4 && m.jt_pack && t >= m.J) {
        // fused path, four bones per vertex: the thread's three corner vertices (a vertex pick = the same vertex three times with
        // barycentric weights 1, 0, 0) from ONE packed record per output joint (slots, packed bone ids, weights, barycentric
        // weights: built at ap_smplx_create) -- two dependent load levels (record | v_posed of the corners) instead of four
        // (vertex ids -> slot / ids / weights -> v_posed); same blend order and arithmetic as skin_point_dyn with K = 4
        const float4* rec = m.jt_pack + (size_t)(t - m.J) * 6;
        const float4 r0 = rec[0], r1 = rec[1], wa = rec[2], wb = rec[3], wc = rec[4], r5 = rec[5];
        const int slot[3] = {__builtin_bit_cast(int, r0.x), __builtin_bit_cast(int, r0.y), __builtin_bit_cast(int, r0.z)};
        const uint32_t id8[3] = {__builtin_bit_cast(uint32_t, r1.x), __builtin_bit_cast(uint32_t, r1.y), __builtin_bit_cast(uint32_t, r1.z)};
        const float4 w4[3] = {wa, wb, wc};
        const float bw[3] = {r5.x, r5.y, r5.z};
        float q[3][3];
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            q[f][0] = vs[3 * slot[f]]; q[f][1] = vs[3 * slot[f] + 1]; q[f][2] = vs[3 * slot[f] + 2];
        }
        float pf[3][3];
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            const int idx[4] = {(int)(id8[f] & 0x3fu), (int)((id8[f] >> 6) & 0x3fu), (int)((id8[f] >> 12) & 0x3fu), (int)((id8[f] >> 18) & 0x3fu)};
            const float w[4] = {w4[f].x, w4[f].y, w4[f].z, w4[f].w};
            skin_point<4>(As, idx, w, q[f][0], q[f][1], q[f][2], pf[f]);
        }
        if (t >= m.J + m.n_extra) {
            o[0] = o[1] = o[2] = 0.f;
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                o[0] = fmaf(pf[f][0], bw[f], o[0]); o[1] = fmaf(pf[f][1], bw[f], o[1]); o[2] = fmaf(pf[f][2], bw[f], o[2]);
            }
        } else {
            o[0] = pf[0][0]; o[1] = pf[0][1]; o[2] = pf[0][2];
        }
    } else {
    auto skin = [&](int v, float* out) {
        const float* q = vs ? vs + 3 * m.jv_slot[v] : vp + 3 * v;
        skin_point_dyn(As, m.skin_idx + (size_t)v * m.K, m.skin_w + (size_t)v * m.K, m.K, q[0], q[1], q[2], out);
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

// T_KS_PACK: K steps of 32 in the packed direction fragments (K = 256: 20 shape / expression coefficients + 189 body-pose features +
// the jaw / eye features, which are exactly zero on this path; the kernel multiplies the first T_KS = 7 steps)
constexpr int T_KS_PACK = 8;
bool ap_smplx_lbs_fused_supported(const SmplxModelDev& m) {
    return m.K == 4 && m.J <= T_MAXJ && m.dirs_frag != nullptr && m.skin_idx8 != nullptr && m.coef_split && m.ncoef * 4 >= T_KS_PACK * 128;
}

size_t ap_smplx_dirs_frag_bytes(int V) { return (size_t)((V + 15) / 16) * T_KS_PACK * 6 * 1024; }

hipError_t ap_launch_smplx_lbs_fused(const SmplxModelDev& m, const SmplxFwdArgs& a, int n_cu, int merged, hipStream_t st) {
    static bool attr_set[AP_MAX_DEVICES] = {};
    if (!ap_smplx_lbs_fused_supported(m)) return hipErrorInvalidValue;
    constexpr int NBM = 22;
    auto lds_of = [](int nbj, int bb) { return bb * nbj * 12 * 4 + bb * T_CROW + bb * 64; };
    // merged: the 22 posed transforms (body-only skin table); 2 = ... with 64 bodies per workgroup (A/B: measured slower, profiles/r06_lbs_ab.txt)
    const bool mg = merged && m.nb == NBM && m.skin_idx8b && m.skin_w4b;
    const bool wide = mg && merged == 2 && a.n >= 256;
    int dev = 0;
    hipError_t e = ap_current_device(&dev);
    if (e != hipSuccess) return e;
    if (!attr_set[dev]) {
        e = hipFuncSetAttribute((const void*)smplx_lbs_tail_kernel<T_MAXJ, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_of(T_MAXJ, 32));
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)smplx_lbs_tail_kernel<NBM, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_of(NBM, 32));
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)smplx_lbs_tail_kernel<NBM, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_of(NBM, 64));
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    // body groups x vertex ranges ~ one workgroup per CU; 8 | n_vr keeps the blocks of one range on one XCD (block b runs on XCD
    // b % 8), so its direction rows are fetched from HBM once and served to the other body groups from that XCD's L2
    const int bb = wide ? 64 : 32, gw = T_NW / (bb / 32);
    const int bgs = (a.n + bb - 1) / bb, ngroups = (m.V + 15) / 16;
    int n_vr = (n_cu + bgs - 1) / bgs;
    n_vr = n_vr >= 8 ? (n_vr / 8) * 8 : n_vr;
    n_vr = n_vr < 1 ? 1 : (n_vr > (ngroups + gw - 1) / gw ? (ngroups + gw - 1) / gw : n_vr);
    const int gpv = (ngroups + n_vr - 1) / n_vr;
    n_vr = (ngroups + gpv - 1) / gpv;
    if (wide) hipLaunchKernelGGL((smplx_lbs_tail_kernel<NBM, 64>), dim3(bgs * n_vr), dim3(64 * T_NW), lds_of(NBM, 64), st, m, a, n_vr, gpv);
    else if (mg) hipLaunchKernelGGL((smplx_lbs_tail_kernel<NBM, 32>), dim3(bgs * n_vr), dim3(64 * T_NW), lds_of(NBM, 32), st, m, a, n_vr, gpv);
    else hipLaunchKernelGGL((smplx_lbs_tail_kernel<T_MAXJ, 32>), dim3(bgs * n_vr), dim3(64 * T_NW), lds_of(T_MAXJ, 32), st, m, a, n_vr, gpv);
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
