// conv3 of an identity bottleneck and conv1 of the NEXT bottleneck as ONE pixel-local kernel (bf16, gfx950).
//
//   block k :  out = relu(bn3(conv3(t2)) + x)            Bottleneck.forward, model_copenet.py:38-45
//   block k+1: t1' = relu(bn1(conv1(out)))               model_copenet.py:29-31
//
// Both convolutions are 1x1: a pixel's t1' depends on that pixel only, so `out` is written to HBM once (it is the next
// block's identity) and never read back for conv1 -- the 4P-channel tensor makes one HBM trip less per block boundary
// (layer2: 411 MB of 1.44 GB per pair at 512 images, layer3: 206 MB of 719 MB).
//
// Structure (MI355X-first, not a two-GEMM translation): a workgroup is FOUR waves that share nothing but the weight
// stream; each wave owns 16 pixels and keeps everything of them in registers:
//   * the t2 rows of its pixels as MFMA B fragments (P/32 x 4 VGPRs), loaded once per tile;
//   * conv3 in chunks of 128 output channels: acc3[8] (16 channels x 16 pixels each, v_mfma_f32_16x16x32_bf16);
//   * the chunk's epilogue (BN, + identity, ReLU, bf16) in registers; the packed result IS the B fragment of the second
//     GEMM: an MFMA D fragment holds 4 channels per lane, two neighbouring fragments 8 -- the 8 consecutive K values a
//     B fragment wants -- because the weight ROWS of a tile are permuted at pack time (row f*16 + i of a tile is channel
//     (f>>1)*32 + (i>>2)*8 + (f&1)*4 + (i&3) of its 128): a lane then owns 8 CONSECUTIVE channels per fragment pair, so the
//     identity load, the `out` store and the conv1 operand are all one aligned 16-byte piece per lane (16 pixel rows x 64 B
//     per wave instruction), with no LDS round trip and no cross-lane traffic;
//   * acc1[N1/16] for conv1, accumulated chunk by chunk in the K order of the stand-alone kernels (results are
//     bit-identical to conv3 followed by conv1 through conv_pipe.hip: same K-step order, same k-slot assignment, same
//     epilogue expression).
// Weights: ONE linear stream of 16-KiB tiles (128 rows x 64 K, swizzle baked in by pair_pack_kernel) in consumption order,
// fetched by LDS-DMA into a 4-slot ring, 4 pieces per wave and tile issued between the MFMA groups; one raw s_barrier per
// tile, every wait counted by hand (vmcnt retires in order; loads, LDS-DMA pieces and stores of a wave share the counter:
// a wait for "at most N younger operations" is exact when only DMA pieces follow and conservative when epilogue loads /
// stores sit among them).  -DPR_SAFE=1 turns every counted vmcnt into vmcnt(0) for cross-checking.
// Two workgroups (8 waves, <= 256 VGPRs) per CU: the waves of a SIMD belong to different workgroups and drift apart, so one
// wave's epilogue (VALU + HBM latency) runs under the other's MFMA groups.
#include <type_traits>

#include "ap_common.h"
#include "kernels.h"

#ifndef PR_SAFE
#define PR_SAFE 0
#endif

namespace {

constexpr int PR_BM = 64, PR_TILE = 16384, PR_S = 4, PR_NT = 256, PR_RING = PR_S * PR_TILE;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PR_SAFE ? 0 : N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int N> __device__ __forceinline__ void wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);                       // register-only MFMAs ignore "memory": keep them below the wait
}
template <int OFF> __device__ __forceinline__ u32x4 lds_read_b128(uint32_t addr) {
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
// (fp32 table rows are read as f32x4 at once: __builtin_bit_cast applied to ONE ELEMENT of an integer vector mis-compiles
// with this hipcc -- every element but the first comes out wrong; conv_pipe.hip's epilogue carries the same note)
template <int OFF> __device__ __forceinline__ f32x4 lds_read_f32x4(uint32_t addr) {
    f32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
// register load the compiler does not count (a load it counts is answered with vmcnt(0) beside LDS-DMA and would drain the
// weight ring): the destination is valid only behind the hand-placed wait that names it
template <int OFF> __device__ __forceinline__ u32x4 gload_b128(const unsigned char* p) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r) : "v"(p), "n"(OFF) : "memory");
    return r;
}
template <int I, int N, typename F> __device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}
__device__ __forceinline__ f32x4 mfma16(const u32x4& w, const u32x4& x, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
}

// channel (within its 128) of row rho of a weight tile: lane group g4 of fragment pair q = f >> 1 owns channels
// q*32 + g4*8 .. + 7 (fragment f = 2q + e holds e*4 .. e*4 + 3 of them)
__host__ __device__ __forceinline__ int pr_row_channel(int rho) {
    const int f = rho >> 4, i = rho & 15;
    return (f >> 1) * 32 + (i >> 2) * 8 + (f & 1) * 4 + (i & 3);
}

// weight stream: for every 128-channel chunk nb of conv3: P/64 tiles of conv3 (K steps of 64), then for each 64-deep half
// kh of the chunk and each 128-row half hn of conv1: one tile of conv1.  Tile = 128 rows x 128 B, 16-byte chunk c of row
// rho at position c ^ (rho & 7); thread = one 16-byte chunk.
__global__ void __launch_bounds__(256) pair_pack_kernel(const bf16_t* __restrict__ w3, const bf16_t* __restrict__ w1,
                                                        unsigned char* __restrict__ dst, int P, int N1) {
    const int KP = P / 64, HN = N1 / 128, SPC = KP + 2 * HN, T = (4 * P / 128) * SPC;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * 1024) return;
    const int t = idx >> 10, rho = (idx >> 3) & 127, c = (idx & 7) ^ (rho & 7);
    const int nb = t / SPC, j = t - nb * SPC, chl = pr_row_channel(rho);
    const bf16_t* src;
    if (j < KP) {
        src = w3 + (size_t)(nb * 128 + chl) * P + j * 64 + c * 8;
    } else {
        const int g = j - KP, kh = g / HN, hn = g - kh * HN;
        src = w1 + (size_t)(hn * 128 + chl) * (4 * P) + nb * 128 + kh * 64 + c * 8;
    }
    *(u32x4*)(dst + (size_t)idx * 16) = *(const u32x4*)src;
}

template <int P, int N1>
__global__ void __launch_bounds__(PR_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) conv_pair_kernel(const PairArgs p) {
    constexpr int C3 = 4 * P, KP = P / 64, HN = N1 / 128, KG = 2 * HN, NB = C3 / 128, SPC = KP + KG, S = PR_S;
    constexpr int NXF = P / 32;
    constexpr int TAB3 = PR_RING, TAB1 = TAB3 + 2 * C3 * 4;
    static_assert(SPC >= S, "the tail waits assume at least S tiles per chunk");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, g4 = lane >> 4;
    const int m = blockIdx.x * PR_BM + wave * 16 + lr;
    const bool mok = m < p.M;
    const size_t mc = mok ? (size_t)m : (size_t)(p.M - 1);      // ragged tail: loads clamped, stores masked

    // BatchNorm tables into LDS (read in the epilogues by inline-asm ds_read: a load the compiler counts would be fenced
    // against the LDS-DMA writes of the ring with vmcnt(0))
    {
        float* t3 = (float*)(smem + TAB3);
        float* t1 = (float*)(smem + TAB1);
        for (int i = tid; i < C3; i += PR_NT) { t3[i] = p.s3[i]; t3[C3 + i] = p.h3[i]; }
        for (int i = tid; i < N1; i += PR_NT) { t1[i] = p.s1[i]; t1[N1 + i] = p.h1[i]; }
    }
    __syncthreads();

    const unsigned char* wnext = (const unsigned char*)p.wstream + (size_t)(wave * 4) * 1024 + lane * 16;   // next tile to issue
    const unsigned char* t2p = (const unsigned char*)p.t2 + (mc * P + g4 * 8) * 2;
    const unsigned char* resp = (const unsigned char*)p.res + (mc * C3 + g4 * 8) * 2;
    unsigned char* outp = (unsigned char*)p.out + ((size_t)m * C3 + g4 * 8) * 2;
    unsigned char* t1p = (unsigned char*)p.t1n + ((size_t)m * N1 + g4 * 8) * 2;
    // every kernel-argument load completes here: a scalar load the compiler believes pending inside the loop costs an
    // s_waitcnt lgkmcnt(0) in front of each DMA instruction, which also drains the fragment reads in flight
    asm volatile("" ::"s"(p.wstream), "s"(p.t2), "s"(p.res), "s"(p.out), "s"(p.t1n), "s"(p.M));

    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // A fragment of tile row f*16 + lr, K half s: chunk s*4 + g4 at position (s*4 + g4) ^ (lr & 7)
    const uint32_t fb0 = lds0 + lr * 128 + ((g4 ^ (lr & 7)) << 4), fb1 = fb0 ^ 64u;
    const uint32_t tb3 = lds0 + TAB3 + g4 * 32, tb1 = lds0 + TAB1 + g4 * 32;

    int so = 0, si = (S - 1) * PR_TILE;                      // ring byte offsets: tile of this step / slot of the tile issued in it
    auto piece = [&](int i, bool issue) {                   // one 1-KiB piece of the tile S-1 steps ahead
        if (issue)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wnext + i * 1024),
                                             (__attribute__((address_space(3))) void*)(smem + si + (wave * 4 + i) * 1024), 16, 0, 0);
    };
    // one weight tile: 16 MFMAs (8 row fragments x 2 K halves) on the 16 pixels of this wave
    auto step = [&](f32x4* acc, const u32x4& b0, const u32x4& b1, bool issue) {
        const uint32_t a0 = fb0 + so, a1 = fb1 + so;
        u32x4 wa[4], wb[4];
        wa[0] = lds_read_b128<0>(a0); wa[1] = lds_read_b128<2048>(a0); wa[2] = lds_read_b128<4096>(a0); wa[3] = lds_read_b128<6144>(a0);
        wb[0] = lds_read_b128<8192>(a0); wb[1] = lds_read_b128<10240>(a0); wb[2] = lds_read_b128<12288>(a0); wb[3] = lds_read_b128<14336>(a0);
        wait_lgkmcnt<4>();
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[f] = mfma16(wa[f], b0, acc[f]);
        __builtin_amdgcn_sched_barrier(0);
        piece(0, issue);
        wa[0] = lds_read_b128<0>(a1); wa[1] = lds_read_b128<2048>(a1); wa[2] = lds_read_b128<4096>(a1); wa[3] = lds_read_b128<6144>(a1);
        wait_lgkmcnt<4>();
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[4 + f] = mfma16(wb[f], b0, acc[4 + f]);
        __builtin_amdgcn_sched_barrier(0);
        piece(1, issue);
        wb[0] = lds_read_b128<8192>(a1); wb[1] = lds_read_b128<10240>(a1); wb[2] = lds_read_b128<12288>(a1); wb[3] = lds_read_b128<14336>(a1);
        wait_lgkmcnt<4>();
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[f] = mfma16(wa[f], b1, acc[f]);
        __builtin_amdgcn_sched_barrier(0);
        piece(2, issue);
        wait_lgkmcnt<0>();
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[4 + f] = mfma16(wb[f], b1, acc[4 + f]);
        __builtin_amdgcn_sched_barrier(0);
        piece(3, issue);
        if (issue) wnext += PR_TILE;
        so = (so + PR_TILE) & (PR_RING - 1);
        si = (si + PR_TILE) & (PR_RING - 1);
    };
    // BN + (identity) + ReLU + bf16 of the 8 consecutive channels a lane holds in fragments (2q, 2q+1); sc / sh: their tables
    auto bn8 = [&](const f32x4& lo, const f32x4& hi, const f32x4& s0, const f32x4& s1, const f32x4& h0, const f32x4& h1,
                   const u32x4* res) -> u32x4 {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = lo[e] * s0[e] + h0[e];
            v[4 + e] = hi[e] * s1[e] + h1[e];
        }
        if (res) {
            const uint32_t r0 = (*res).x, r1 = (*res).y, r2 = (*res).z, r3 = (*res).w;
            float a, b;
            unpack_bf16x2(r0, a, b); v[0] += a; v[1] += b;
            unpack_bf16x2(r1, a, b); v[2] += a; v[3] += b;
            unpack_bf16x2(r2, a, b); v[4] += a; v[5] += b;
            unpack_bf16x2(r3, a, b); v[6] += a; v[7] += b;
        }
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(fmaxf(v[2 * e], 0.f), fmaxf(v[2 * e + 1], 0.f));
        return o;
    };

    // ---------------------------------------------------------------- prologue: t2 fragments, tiles 0 .. S-2
    u32x4 xf[NXF];
    sfor<0, NXF>([&](auto I) { xf[I] = gload_b128<I * 64>(t2p); });
#pragma unroll
    for (int t = 0; t < S - 1; ++t) {
        si = t * PR_TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) piece(i, true);
        wnext += PR_TILE;
    }
    si = (S - 1) * PR_TILE;
    f32x4 acc1[HN * 8];
#pragma unroll
    for (int i = 0; i < HN * 8; ++i) acc1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 ry[4];                                             // identity pieces of the chunk, then its packed result (= conv1 operand)

    for (int nb = 0; nb < NB; ++nb) {
        const bool lastc = nb == NB - 1;
        f32x4 acc3[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc3[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        sfor<0, SPC>([&](auto J) {
            constexpr int j = J;
            // own pieces of this step's tile have landed: S-2 younger tiles may stay in flight (fewer at the very end)
            if (lastc && j == SPC - 1) wait_vmcnt<0>();
            else if (lastc && j == SPC - 2) wait_vmcnt<4>();
            else wait_vmcnt<4 * (S - 2)>();
            if constexpr (j == 0) {                          // (first chunk: the t2 fragments are older than tile 0's pieces)
#pragma unroll
                for (int i = 0; i < NXF; ++i) asm volatile("" : "+v"(xf[i]));
            }
            __builtin_amdgcn_s_barrier();                    // everybody's pieces landed; the slot read last step is free
            const bool issue = !(lastc && j + S - 1 >= SPC);
            if constexpr (j < KP) {
                step(acc3, xf[2 * j], xf[2 * j + 1], issue);
            } else {
                constexpr int g = j - KP, kh = g / HN, hn = g % HN;
                step(&acc1[hn * 8], ry[2 * kh], ry[2 * kh + 1], issue);
            }
            if constexpr (j == 0) {                          // identity of this chunk: 4 x 16 B per lane, behind the step's DMA pieces
                const unsigned char* rp = resp + nb * 256;
                ry[0] = gload_b128<0>(rp); ry[1] = gload_b128<64>(rp); ry[2] = gload_b128<128>(rp); ry[3] = gload_b128<192>(rp);
            }
            if constexpr (j == KP - 1) {
                // ---------------------------------------------------- conv3 epilogue of chunk nb, in registers
                // younger than the identity loads: the DMA pieces of steps 1 .. KP-1 of this chunk (those that were issued)
                constexpr int YS = 4 * (KP - 1);
                constexpr int nl = (KP - 1 < SPC - S ? KP - 1 : (SPC - S > 0 ? SPC - S : 0));
                if (lastc) wait_vmcnt<4 * nl>(); else wait_vmcnt<YS>();
                asm volatile("" : "+v"(ry[0]), "+v"(ry[1]), "+v"(ry[2]), "+v"(ry[3]));
                const uint32_t ta = tb3 + nb * 512;
                f32x4 s0 = lds_read_f32x4<0>(ta), s1 = lds_read_f32x4<16>(ta), h0 = lds_read_f32x4<C3 * 4>(ta), h1 = lds_read_f32x4<C3 * 4 + 16>(ta);
                sfor<0, 4>([&](auto Q) {
                    constexpr int q = Q;
                    f32x4 ns0, ns1, nh0, nh1;
                    if constexpr (q < 3) {
                        ns0 = lds_read_f32x4<(q + 1) * 128>(ta); ns1 = lds_read_f32x4<(q + 1) * 128 + 16>(ta);
                        nh0 = lds_read_f32x4<C3 * 4 + (q + 1) * 128>(ta); nh1 = lds_read_f32x4<C3 * 4 + (q + 1) * 128 + 16>(ta);
                        wait_lgkmcnt<4>();
                    } else {
                        wait_lgkmcnt<0>();
                    }
                    ry[q] = bn8(acc3[2 * q], acc3[2 * q + 1], s0, s1, h0, h1, &ry[q]);
                    if (mok) *(u32x4*)(outp + nb * 256 + q * 64) = ry[q];
                    if constexpr (q < 3) { s0 = ns0; s1 = ns1; h0 = nh0; h1 = nh1; }
                });
            }
        });
    }
    // ---------------------------------------------------------------- conv1 epilogue: BN + ReLU + bf16, 16-byte stores
    sfor<0, HN * 4>([&](auto I) {
        constexpr int hn = I / 4, q = I % 4;
        const uint32_t ta = tb1 + (hn * 128 + q * 32) * 4;
        const f32x4 s0 = lds_read_f32x4<0>(ta), s1 = lds_read_f32x4<16>(ta), h0 = lds_read_f32x4<N1 * 4>(ta), h1 = lds_read_f32x4<N1 * 4 + 16>(ta);
        wait_lgkmcnt<0>();
        const u32x4 o = bn8(acc1[hn * 8 + 2 * q], acc1[hn * 8 + 2 * q + 1], s0, s1, h0, h1, nullptr);
        if (mok) *(u32x4*)(t1p + (hn * 128 + q * 32) * 2) = o;
    });
}

template <int P, int N1>
hipError_t launch_pair(const PairArgs& a, hipStream_t st) {
    static bool attr_set[AP_MAX_DEVICES] = {};
    auto kern = conv_pair_kernel<P, N1>;
    constexpr int lds = PR_RING + (2 * 4 * P + 2 * N1) * 4;
    static_assert(lds <= 81920, "two workgroups per CU");
    int dev = 0;
    hipError_t e = ap_current_device(&dev);
    if (e != hipSuccess) return e;
    if (!attr_set[dev]) {
        e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3((a.M + PR_BM - 1) / PR_BM), dim3(PR_NT), lds, st, a);
    return hipGetLastError();
}

}  // namespace

bool ap_conv_pair_supported(int P, int N1) { return (P == 128 && (N1 == 128 || N1 == 256)) || (P == 256 && N1 == 256); }

size_t ap_conv_pair_stream_bytes(int P, int N1) {
    return (size_t)(4 * P / 128) * (P / 64 + 2 * (N1 / 128)) * PR_TILE;
}

hipError_t ap_launch_pair_pack(const void* w3, const void* w1, void* dst, int P, int N1, hipStream_t st) {
    if (!ap_conv_pair_supported(P, N1)) return hipErrorInvalidValue;
    const size_t chunks = ap_conv_pair_stream_bytes(P, N1) / 16;
    hipLaunchKernelGGL(pair_pack_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, (const bf16_t*)w3,
                       (const bf16_t*)w1, (unsigned char*)dst, P, N1);
    return hipGetLastError();
}

hipError_t ap_launch_conv_pair(const PairArgs& a, int P, int N1, hipStream_t st) {
    if (a.M <= 0 || !a.t2 || !a.res || !a.wstream || !a.out || !a.t1n) return hipErrorInvalidValue;
    if (P == 128 && N1 == 128) return launch_pair<128, 128>(a, st);
    if (P == 128 && N1 == 256) return launch_pair<128, 256>(a, st);
    if (P == 256 && N1 == 256) return launch_pair<256, 256>(a, st);
    return hipErrorInvalidValue;
}
