// gdyn_kernels.hip -- hand-written HIP kernels (gfx950, wave64) of the Brownian-dynamics path.
//
//   k_step<MODE,PERIODIC,TILED,PK,S16>
//                          one thread per bead: neighbour-list pair forces (AB-mixed soft cores), bonded / bending /
//                          point-source / ellipsoid- and inner-sphere-wall forces, Euler-Maruyama update with
//                          Philox4x32-10 noise, Verlet-skin check, per-block wall-reaction partials, running displacement
//                          maximum.  TILED: the block's beads and their neighbour cells are staged into LDS by DMA, list
//                          entries address that tile, lists come in a near and a far class (the far class is skipped
//                          while it cannot matter).  The per-step "callback" state (time, bead/bond scale, wall semiaxes
//                          ODE; reference 5-sim-genome/src/simulation_interphase/simulation_driver_interphase.cc:12-80) is
//                          advanced by wave 0 of every block in the prologue of the NEXT launch: no host round trip per step.
//   k_ctx                  one wave per replica: the pending callback at the end of a chunk, the reaction fold of a force
//                          evaluation.
//   k_softwell             droplet attraction among the few hundred target beads (after k_step; linear in the force).
//   k_bbox .. k_fill       neighbour search (micromd md::neighbor_searcher; call sites e.g.
//                          simulation_interphase/contact_map.cc:64-66): bounding box, cell binning, counting sort into
//                          slot order, tile descriptors, list fill (27-cell sweep from the LDS tile, or from global
//                          memory on the generic path).
//   k_pairs                pair search (contact map, glue candidates) filtered from the resident list.
//   k_ct_*                 time-integrated contact maps in HBM (per-replica hash tables: insert, grow, dump).
//   GD_STAMP / GD_FSTAMP   in-kernel section stamps of the developer timing builds (gdyn_stamps.h; empty in the product).
//
// MFMA is not used: the path is an irregular short-range N-body sum.
#include <hip/hip_fp16.h>

#include <algorithm>

#include "gdyn_types.h"

#include "gdyn_stamps.h"

// Replay builds of k_step (developer timing builds, never shipped: make -C csrc abl N=40 / N=41; tools/replay.sh): the kernel's
// memory pattern with the arithmetic stripped (N = 40, RP_MEM: records, tile DMA, list and adjacency chunks, LDS gathers, the store;
// every loaded value is consumed by an empty asm), and its arithmetic with the operands resident (N = 41, RP_ALU: trip counts
// from the real records, noise, pair / bond / wall arithmetic on register values the compiler cannot fold; no tile DMA, no
// chunk loads, no LDS gathers).  Same grid, same LDS allocation.  Every hook sits behind the preprocessor: the product source
// is token for token what it is without them.
#if GD_ABL == 40 || GD_ABL == 41 || GD_ABL == 43
#define GD_REPLAY 1
constexpr bool RP_MEM = (GD_ABL == 40), RP_ALU = (GD_ABL == 41 || GD_ABL == 43);
#define GD_CONSUME4(v) asm volatile("" :: "v"((v).x), "v"((v).y), "v"((v).z), "v"((v).w))
#define GD_LAUNDER4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))
#endif

#define TERM_PAIR 1u
#define TERM_BOND 2u
#define TERM_BEND 4u
#define TERM_POINT 8u
#define TERM_WALL 16u
#define TERM_DYNAMIC 32u
#define RUN_UPDATE_SCALES 1
#define RUN_WALL_DYNAMICS 2
#define NOISE_PHILOX 0
#define NOISE_ZERO 1
#define NOISE_HOST 2
#define POT_HARMONIC 0
#define POT_SPRING 1
#define POT_SEMISPRING 2
#define POT_SOFTCORE 3

// ------------------------------------------------------------------ helpers

// XCD-aware block mapping: the dispatcher deals workgroups round-robin over the 8 XCDs, so
// physical block b runs on XCD b%8.  Every XCD gets one contiguous chunk (cpb blocks = a slab
// of cell-sorted slots) of every replica, so the halo re-reads of neighbouring blocks hit that
// XCD's L2.  Speed only: any placement gives the same results.
// cpb == 0 (replica count a multiple of 8): whole replicas per XCD instead -- XCD x runs replicas x, x+8, ... block by
// block, so no tile halo is shared between XCDs at all.
__device__ __forceinline__ bool block_map(unsigned b, unsigned nblk, unsigned cpb, unsigned &r, unsigned &blk)
{
    const unsigned xcd = b % GD_XCDS, q = b / GD_XCDS;
    if (cpb == 0) { r = (q / nblk) * GD_XCDS + xcd; blk = q % nblk; return true; }
    r = q / cpb;
    blk = xcd * cpb + q % cpb;
    return blk < nblk;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 nt_load(const uint4 *ptr)      // streaming 16-byte load (non-temporal hint)
{
    const u32x4 v = __builtin_nontemporal_load((const u32x4 *)ptr);
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ float2 unpack_ab(float w)
{
    const unsigned u = __float_as_uint(w);
    return make_float2(__half2float(__ushort_as_half((unsigned short)(u & 0xffffu))),
                       __half2float(__ushort_as_half((unsigned short)(u >> 16))));
}
__device__ __forceinline__ float pack_ab(float2 ab)
{
    const unsigned lo = __half_as_ushort(__float2half_rn(ab.x)), hi = __half_as_ushort(__float2half_rn(ab.y));
    return __uint_as_float(lo | (hi << 16));
}

// softcore<2,3> + softcore<8,3> (the pair force of all four reference models), branch-free:
// the clamp makes each term vanish beyond its own diameter.  ca = 6 eps_a / sa^2, cb = 24 eps_b / sb^2.
// waca = wa*ca and wbcb = wb*cb come in pre-multiplied (one fma each from the neighbour's a/b).
__device__ __forceinline__ float softcore_2383(float r2, float inv_sa2, float inv_sb2, float waca, float wbcb)
{
    // both arguments are <= 1, so max(., 0) is the [0,1] clamp -- which is free as the output modifier of the fma
    const float ga = __builtin_amdgcn_fmed3f(fmaf(-r2, inv_sa2, 1.0f), 0.0f, 1.0f);
    const float u2 = r2 * inv_sb2, u4 = u2 * u2;
    const float gb = __builtin_amdgcn_fmed3f(fmaf(-u4, u4, 1.0f), 0.0f, 1.0f);
    return fmaf(wbcb * (gb * gb), u4 * u2, waca * (ga * ga));
}
__device__ __forceinline__ float softcore_2383_energy(float r2, float inv_sa2, float inv_sb2, float ea, float eb, float wa, float wb)
{
    const float ga = fmaxf(1.0f - r2 * inv_sa2, 0.0f);
    const float u2 = r2 * inv_sb2, u4 = u2 * u2;
    const float gb = fmaxf(1.0f - u4 * u4, 0.0f);
    return wa * ea * ga * ga * ga + wb * eb * gb * gb * gb;
}

// Wave reductions over all 64 lanes by DPP (row shifts inside the rows of 16, then the row broadcasts): six fused
// add / max instructions and one v_readlane instead of six ds_bpermute round trips through the LDS crossbar.  All lanes
// must be active; the result is wave-uniform.  Lanes without a source keep the identity 0 (sums; maxima of non-negative values).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp0(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_f(float v)
{
    v += dpp0<0x111, 0xf>(v);      // row_shr:1
    v += dpp0<0x112, 0xf>(v);      // row_shr:2
    v += dpp0<0x114, 0xf>(v);      // row_shr:4
    v += dpp0<0x118, 0xf>(v);      // row_shr:8
    v += dpp0<0x142, 0xa>(v);      // row_bcast:15 into rows 1, 3
    v += dpp0<0x143, 0xc>(v);      // row_bcast:31 into rows 2, 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp0_d(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_sum_d(double v)
{
    v += dpp0_d<0x111, 0xf>(v);
    v += dpp0_d<0x112, 0xf>(v);
    v += dpp0_d<0x114, 0xf>(v);
    v += dpp0_d<0x118, 0xf>(v);
    v += dpp0_d<0x142, 0xa>(v);
    v += dpp0_d<0x143, 0xc>(v);
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// (v >= 0, not NaN: the maximum is taken on the bit patterns as unsigned integers, which order like the floats.  fmaxf on the DPP
// operand compiled to a zero move, the DPP move, a canonicalising self-maximum and the maximum -- four instructions per step on a
// kernel that is bound by VALU issue; the integer form fuses into one v_max_u32 with the DPP modifier)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp0_u(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ float wave_max_f(float vf)      // vf >= 0
{
    unsigned v = __float_as_uint(vf);
    v = max(v, dpp0_u<0x111, 0xf>(v));
    v = max(v, dpp0_u<0x112, 0xf>(v));
    v = max(v, dpp0_u<0x114, 0xf>(v));
    v = max(v, dpp0_u<0x118, 0xf>(v));
    v = max(v, dpp0_u<0x142, 0xa>(v));
    v = max(v, dpp0_u<0x143, 0xc>(v));
    return __int_as_float(__builtin_amdgcn_readlane((int)v, 63));
}
// max without the canonicalising self-maximum the compiler puts in front of fmaxf on a value loaded from memory (neither operand is
// a signalling NaN here: a bond record's lower clamp, an elongation)
__device__ __forceinline__ float fmax_plain(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// u2^(n/2) for even n in {0,2,4,6,10}
__device__ __forceinline__ float pow_half(float u2, int n)
{
    const float u4 = u2 * u2;
    switch (n) {
    case 0: return 1.0f;
    case 2: return u2;
    case 4: return u4;
    case 6: return u4 * u2;
    default: return u4 * u4 * u2;   // 10
    }
}
__device__ __forceinline__ float pow_q(float g, int q)   // g^q, q in 0..3
{
    switch (q) {
    case 0: return 1.0f;
    case 1: return g;
    case 2: return g * g;
    default: return g * g * g;
    }
}

// md::softcore_potential<P,Q>: U = eps (1-(r/sigma)^P)^Q for r<sigma.
// Returns fr with F_on_i = fr * (x_i - x_j); energy in e.
__device__ __forceinline__ void softcore(float eps, float inv_s2, int P, int Q, float r2, float &e, float &fr)
{
    const float u2 = r2 * inv_s2;
    e = 0.0f; fr = 0.0f;
    if (u2 < 1.0f && eps != 0.0f) {
        const float upm2 = pow_half(u2, P - 2);
        const float g = 1.0f - upm2 * u2;
        const float gq1 = pow_q(g, Q - 1);
        e = eps * gq1 * g;
        fr = eps * (float)(P * Q) * inv_s2 * gq1 * upm2;
    }
}

__device__ __forceinline__ void bond_pot(int kind, float K, float l, int P, int Q, float r2, float &e, float &fr)
{
    e = 0.0f; fr = 0.0f;
    if (kind == POT_HARMONIC) {
        e = 0.5f * K * r2; fr = -K;
    } else if (kind == POT_SOFTCORE) {
        softcore(K, 1.0f / (l * l), P, Q, r2, e, fr);
    } else {
        const float inv_d = r2 > 0.0f ? rsqrtf(r2) : 0.0f;     // d = r2 * inv_d, no divide
        const float x = r2 * inv_d - l;
        if (kind == POT_SPRING || x > 0.0f) { e = 0.5f * K * x * x; fr = -K * x * inv_d; }
    }
}

__device__ __forceinline__ float3 min_image(float3 d, const float *box, const float *inv_box)
{
    d.x -= box[0] * rintf(d.x * inv_box[0]);
    d.y -= box[1] * rintf(d.y * inv_box[1]);
    d.z -= box[2] * rintf(d.z * inv_box[2]);
    return d;
}

// Philox4x32-10 (Salmon et al., SC'11); counter = (bead, step_lo, step_hi, replica), key = seed.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned &o0, unsigned &o1, unsigned &o2, unsigned &o3)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // one 32 x 32 -> 64 multiply (v_mad_u64_u32) per product instead of separate high and low halves
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned h0 = (unsigned)(p0 >> 32), l0 = (unsigned)p0, h1 = (unsigned)(p1 >> 32), l1 = (unsigned)p1;
        c0 = h1 ^ c1 ^ k0; c1 = l1; c2 = h0 ^ c3 ^ k1; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}
// ((x >> 9) + 0.5) 2^-23 as one fma: k 2^-23 + 2^-24 is a 24-bit value, so the single rounding of the fma is exact
__device__ __forceinline__ float u01(unsigned x) { return fmaf((float)(x >> 9), 1.0f / 8388608.0f, 1.0f / 16777216.0f); }

__device__ __forceinline__ float3 philox_normal3(unsigned long long seed, unsigned bead, long long step, unsigned replica)
{
    unsigned o0, o1, o2, o3;
    philox4x32_10(bead, (unsigned)((unsigned long long)step & 0xffffffffull), (unsigned)((unsigned long long)step >> 32), replica,
                  (unsigned)(seed & 0xffffffffull), (unsigned)(seed >> 32), o0, o1, o2, o3);
    // Box-Muller; v_sin/v_cos take their argument in revolutions.  r = sqrt(-2 ln u) on the hardware log2 and square root (1 ulp
    // each, u in [2^-24, 1 - 2^-24]: no denormal, no range reduction): -2 ln u = (-2 ln 2) log2 u.  The library forms -- ln through an
    // extended-precision product with ln 2 and denormal scaling, the IEEE-exact square root with its two correction steps -- are 27
    // instructions per radius against 3, on a kernel that is bound by VALU issue (profiles/r04_replay.json); the normals stay within
    // 1e-7 relative of the oracle's (tests/test_parity_gpu.py::test_device_philox_normals_kat: <= 3e-6 absolute)
    const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(o0)));
    const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(o2)));
    const float t0 = u01(o1), t1 = u01(o3);
    return make_float3(r0 * __builtin_amdgcn_cosf(t0), r0 * __builtin_amdgcn_sinf(t0), r1 * __builtin_amdgcn_cosf(t1));
}

// cosine bending: U = e (1 - cos theta), d1 = x_j - x_i, d2 = x_k - x_j. Forces on the end beads.
__device__ __forceinline__ bool bend_forces(float3 d1, float3 d2, float e, float3 &fi, float3 &fk, float &cs)
{
    const float n1 = d1.x * d1.x + d1.y * d1.y + d1.z * d1.z, n2 = d2.x * d2.x + d2.y * d2.y + d2.z * d2.z;
    if (n1 == 0.0f || n2 == 0.0f) return false;
    const float il1 = rsqrtf(n1), il2 = rsqrtf(n2);
    cs = (d1.x * d2.x + d1.y * d2.y + d1.z * d2.z) * il1 * il2;
    const float a1 = e * il1, a2 = e * il2;
    fi = make_float3(-a1 * (d2.x * il2 - cs * d1.x * il1), -a1 * (d2.y * il2 - cs * d1.y * il1), -a1 * (d2.z * il2 - cs * d1.z * il1));
    fk = make_float3(a2 * (d1.x * il1 - cs * d2.x * il2), a2 * (d1.y * il1 - cs * d2.y * il2), a2 * (d1.z * il1 - cs * d2.z * il2));
    return true;
}

// The per-step callback state of the drivers, advanced on the device
// (simulation_driver_interphase.cc:14,42-43,59-80). Wave-cooperative: all 64 lanes call it.
__device__ __forceinline__ void apply_callback(DevCtx &c, const StepParams &p, unsigned r, int lane)
{
    double rx = 0, ry = 0, rz = 0;
    if (p.wall.enabled) {
        for (unsigned b = lane; b < p.nblk; b += 64) {
            const float4 v = p.react_in[(size_t)r * p.nblk + b];
            rx += v.x; ry += v.y; rz += v.z;
        }
        rx = wave_sum_d(rx); ry = wave_sum_d(ry); rz = wave_sum_d(rz);
    }
    c.step += 1;
    c.time = (double)c.step * p.dt_d;
    if ((p.run_flags & RUN_UPDATE_SCALES) && p.scaling.enabled) {
        if (p.scaling.from_host) { c.bead_scale = p.scaling.bead_next; c.bond_scale = p.scaling.bond_next; }      // (pure functions of the step index)
        else {
            c.bead_scale = 1.0 - (1.0 - p.scaling.bead_init) * exp(-c.time / p.scaling.bead_tau);
            c.bond_scale = 1.0 - (1.0 - p.scaling.bond_init) * exp(-c.time / p.scaling.bond_tau);
        }
    }
    c.react[0] = rx; c.react[1] = ry; c.react[2] = rz;
    if ((p.run_flags & RUN_WALL_DYNAMICS) && p.wall.enabled) {
#pragma unroll
        for (int k = 0; k < 3; k++) c.semi[k] += p.dt_d * p.wall.mobility * (c.react[k] - p.wall.spring[k] * c.semi[k]);
    }
    c.pending = 0;
}

// Float copy of a replica's context plus the block-uniform wall constants derived from it (CtxF, gdyn_types.h):
// filled by wave 0 of every k_step block (lane 0).
__device__ __forceinline__ void fill_ctxf(CtxF &o, const DevCtx &c, const StepParams &p)
{
    o.step = c.step;
    o.bead_scale = (float)c.bead_scale; o.bond_scale = (float)c.bond_scale;
    o.semi[0] = (float)c.semi[0]; o.semi[1] = (float)c.semi[1]; o.semi[2] = (float)c.semi[2];
    // (hardware reciprocals and square root, 1 ulp: this runs on ONE wave of every block while the other seven wait for it at the
    // barrier -- the IEEE-exact division is a ten-instruction sequence, a dozen of them were a quarter of the wait)
    o.inv_bond_scale2 = __builtin_amdgcn_rcpf((float)c.bond_scale * (float)c.bond_scale);
    o.near2 = 0.f; o.w_inv_sa2 = 0.f; o.w_inv_sb2 = 0.f; o.w_ca = 0.f; o.w_cb = 0.f;
    {   // block-uniform constants of the pair term and of the integrator: divisions and the square root once per block
        const float sc = p.pair.scaled ? (float)c.bead_scale : 1.0f;
        const float sa = p.pair.sigma_a * sc, sb = p.pair.sigma_b * sc;
        o.p_inv_sa2 = sa > 0.f ? __builtin_amdgcn_rcpf(sa * sa) : 0.f; o.p_inv_sb2 = sb > 0.f ? __builtin_amdgcn_rcpf(sb * sb) : 0.f;
        o.p_cut = p.pair.cutoff * sc;
        o.sg_uniform = p.mob_uniform >= 0.f ? __builtin_amdgcn_sqrtf(2.0f * p.kT * p.mob_uniform * p.dt) : -1.0f;
    }
    for (int k = 0; k < 3; k++) { o.inv_semi[k] = 0.f; o.inv_semi2[k] = 0.f; }
    if (p.wall.enabled) {
        float smin = 3.4e38f;
        for (int k = 0; k < 3; k++) {
            const float a = (float)c.semi[k];
            o.inv_semi[k] = __builtin_amdgcn_rcpf(a); o.inv_semi2[k] = o.inv_semi[k] * o.inv_semi[k]; smin = fminf(smin, a);
        }
        {
            const double a2 = c.semi[0] * c.semi[0], b2 = c.semi[1] * c.semi[1], c2 = c.semi[2] * c.semi[2];
            o.w_q[0] = b2 * c2; o.w_q[1] = a2 * c2; o.w_q[2] = a2 * b2; o.w_q[3] = a2 * o.w_q[0];
            o.w_inv_q3 = __builtin_amdgcn_rcpf((float)o.w_q[3]);
        }
        const float wsc = p.wall.scaled ? (float)c.bead_scale : 1.0f;
        const float sa = 0.5f * p.wall.sigma_a * wsc, sb = 0.5f * p.wall.sigma_b * wsc;
        // A bead on the level set s E (s = sqrt(C+1) < 1) is at least (1-s) min(a,b,c) away from the surface
        // (E contains s E + (1-s) min(a,b,c) B), so beyond the larger half diameter the wall force is exactly zero
        const float sthr = 1.0f - fmaxf(sa, sb) * __builtin_amdgcn_rcpf(smin);
        o.near2 = sthr > 0.f ? sthr * sthr * 0.999f : 0.f;
        o.w_inv_sa2 = sa > 0.f ? __builtin_amdgcn_rcpf(sa * sa) : 0.f; o.w_inv_sb2 = sb > 0.f ? __builtin_amdgcn_rcpf(sb * sb) : 0.f;
        o.w_ca = 6.0f * p.wall.eps_a * o.w_inv_sa2; o.w_cb = 24.0f * p.wall.eps_b * o.w_inv_sb2;
    }
}

// ------------------------------------------------------------------- k_step
// TILED: the block first stages its LDS tile (its own 256 slots + all slots of the adjacent
// cells, 9 contiguous slot ranges, TileDesc) with coalesced loads; pair-list entries are 16-bit
// indices into that tile, so the neighbour gather is an LDS read.  PK=1/2: softcore<2,3>+<8,3>
// with / without AB mixing (compile-time), PK=0: runtime powers.
// S16 (tiled lists only): entries are BYTE offsets into the tile (tile capacity < 4096 entries), one decode
// instruction per entry instead of mask + shift-add
template <int MODE, bool PERIODIC, bool TILED, int PK, bool S16, bool SPLIT = false>
#ifndef GD_STEP_WAVES
#define GD_STEP_WAVES 8      // waves per SIMD the tiled step kernel with byte-offset entries is compiled for: 8 = at most 64 VGPRs.  LDS
                             // admits three blocks per CU (6 waves per SIMD, 80 VGPRs would do), but the 64-register form -- four
                             // neighbour reads in flight instead of eight, a scheduling barrier between the half batches -- is
                             // 4.5 % faster at that occupancy (measured; 6 restores the 80-register form)
#endif
__global__ __launch_bounds__(GD_BLOCK, (TILED && MODE == GD_MODE_STEP) ? (S16 ? GD_STEP_WAVES : 4) : 1) void k_step(const StepParams p)      // tiled stepping: three
                                                                                     // blocks per CU = 6 waves per SIMD = at most 80 VGPRs
{
    extern __shared__ __attribute__((aligned(16))) float4 s_tile[];
    __shared__ BondType s_bt[GD_MAX_BOND_TYPES];
    __shared__ CtxF s_ctx;
    __shared__ float s_red[GD_BLOCK / 64][4];
    __shared__ double s_e[GD_BLOCK / 64];

    unsigned r, blk;
    if (!block_map(blockIdx.x, p.nblk, p.cpb, r, blk)) return;
    if (TILED && SPLIT) {      // a step split by tile size (launch_step_mode): this launch takes the blocks of one LDS class
                               // (an instantiation of its own: the unsplit kernel carries no trace of it)
        const unsigned held = p.tiles[(size_t)r * p.nblk + blk].total;
        if (held <= p.tile_lo || held > p.tile_hi) return;
    }
    const unsigned tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const size_t rbase = (size_t)r * p.Np;
    const float4 *__restrict__ rpos = p.pos_in + rbase;

    // every independent per-bead load is issued up front, ahead of the tile staging and the barrier,
    // so their latencies overlap (the kernel is latency-bound, not ALU-bound)
    GD_STAMP_BEGIN();
    // replica context: the two scalars every wave needs (noise counter) are fetched here, before any DMA, so that they
    // stay scalar loads; wave 0 also starts its full-context and reaction-partial loads first
    const long long ctx_step0 = p.ctx_in[r].step;
    const int ctx_pending0 = p.ctx_in[r].pending;
    // largest squared displacement of any bead of the replica since the list build, as of the positions this step reads
    // (kept by the previous steps): with the bead's own displacement it decides whether the far class of its list can matter in this step
    const float dmax0 = (MODE == GD_MODE_STEP && TILED) ? __uint_as_float(p.dmax[r * GD_DMAX_STRIDE]) : 0.f;
    // ---- prologue (tiled path), ordered for the IN-ORDER vmcnt counter:
    //   1. the thread's (meta, bead id) record -- the one per-bead value the noise needs -- and the tile descriptor
    //      (scalar loads: after a DMA the compiler would turn them into per-lane vector loads); both waited for here;
    //   2. bond table + tile DMAs (straight into LDS, nothing returns to registers);
    //   3. the other per-bead loads (build position record, first adjacency and list chunks);
    //   4. the Brownian noise: no wait in front of it, it runs while the DMAs are in flight.
    // The bead's own position is read from the tile after the barrier (tile index own_base + block-local slot), so no
    // load depends on another one.
    // (tiled path: the thread's records sit at rbase + blk * GD_BLOCK + tid -- the balanced thread order of the build)
    // (its wave index is wave-uniform: chunk addresses are a scalar base + lane, no per-lane 64-bit multiplies)
    const size_t gw6 = (rbase + blk * GD_BLOCK) / 64 + (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)wid);
    // (tiled lists: ragged rows -- first KiB and chunks per lane of this wave's rows, one scalar load)
    const uint2 wrow = TILED ? p.wtab[gw6] : make_uint2(0u, 0u);
    const unsigned NCL = TILED ? wrow.y : p.W / 4, NCB = p.WB / 4;
    uint2 mo = make_uint2(0u, 0u);
    float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
    uint4 adj0 = make_uint4(0, 0, 0, 0), qa = adj0, qb = adj0;
    unsigned own_base = 0, tile_ok = 0;     // tile_ok == 0: the tile did not fit (flagged by the build; the host rolls the chunk back)
    if (TILED) {
        // (per-thread records, chunks and tile pieces are addressed as a block- / wave-uniform base plus a 32-bit lane offset:
        // scalar address arithmetic, no 64-bit per-lane adds)
        const size_t tbase = rbase + (size_t)blk * GD_BLOCK;      // first thread position of the block
        mo = *(const uint2 *)((const char *)(p.rec_mo + tbase) + tid * 8u);
        const TileDesc *td = p.tiles + (size_t)r * p.nblk + blk;
        unsigned tlen[GD_TILE_RANGES], tst[GD_TILE_RANGES];      // (the LDS base of a range is the sum of the lengths before it: not loaded)
#pragma unroll
        for (int k = 0; k < GD_TILE_RANGES; k++) { tlen[k] = td->len[k]; tst[k] = td->start[k]; }
        own_base = td->own_base; tile_ok = td->nranges;
        // the record is consumed HERE (an empty asm the compiler has to wait in front of), while it is still the only
        // vector load in flight: placed behind the DMAs, its wait would be a wait for the whole tile.  Its latency
        // overlaps the scalar descriptor loads above.
        asm volatile("" : "+v"(mo.x), "+v"(mo.y));
        GD_STAMP(9);      // record + descriptor arrived
        {
            // tile staging by LDS-DMA (global_load_lds_dwordx4): each wave copies 64 consecutive slots =
            // 1 KiB straight into LDS (destination = wave-uniform base + lane*16), no register hop; the
            // __syncthreads() below waits for the outstanding DMAs (vmcnt) before the tile is read.
            // Stepping: wave 0 stages the bond table only and leaves the tile to the other seven waves (pieces of 64 slots, stride
            // 448) -- it carries the block's context work in front of the barrier, and without its share of the DMA issue (and of
            // the DMAs in front of its context load) it no longer is the wave the others wait for.
            constexpr bool W0_FREE = MODE == GD_MODE_STEP;
            const unsigned wq = (unsigned)__builtin_amdgcn_readfirstlane((int)((W0_FREE ? (wid == 0 ? 0x100000u : wid - 1u) : wid) * 64u));
            constexpr unsigned DMA_STRIDE = W0_FREE ? GD_BLOCK - 64u : GD_BLOCK;
            if (wid == 0 && lane < 2 * GD_MAX_BOND_TYPES)       // the bond-type table: 32 B per type, 16 B per lane (the buffer always holds the full table)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const uint4 *)p.btab + lane),
                                                 (__attribute__((address_space(3))) void *)s_bt, 16, 0, 0);
            unsigned base = 0;
#pragma unroll
#ifdef GD_REPLAY
            for (int k = 0; k < (RP_ALU ? 0 : GD_TILE_RANGES); base += tlen[k], k++) {
#else
            for (int k = 0; k < GD_TILE_RANGES; base += tlen[k], k++) {
#endif
                const unsigned len = tlen[k], st = tst[k];
                for (unsigned q0 = wq; q0 < len; q0 += DMA_STRIDE) {      // (wave-uniform loop: scalar control, one compare per lane)
                    if (q0 + lane < len)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)(rpos + st + q0) + lane * 16u),
                                                         (__attribute__((address_space(3))) void *)(s_tile + base + q0), 16, 0, 0);
                }
            }
        }
        GD_STAMP(11);     // record + descriptor + DMA issue
        // three more loads per thread, unconditionally (every array is allocated for all threads of the block); nothing
        // before the barrier waits for them
        rec = *(const float4 *)((const char *)(p.rec_x0 + tbase) + tid * 16u);
#ifdef GD_REPLAY
        if (!RP_ALU)
#endif
        {
        adj0 = ((const uint4 *)p.badj)[gw6 * NCB * 64 + lane];
        qa = nt_load((const uint4 *)p.nbr16 + (size_t)wrow.x * 64 + lane);
        }
        GD_STAMP(8);      // per-bead loads issued
    } else {
        for (unsigned t = tid; t < (unsigned)p.nbt; t += GD_BLOCK) s_bt[t] = p.btab[t];
    }
    // generic path: one thread per slot, every independent per-bead load issued up front
    unsigned slot = blk * GD_BLOCK + tid;
    bool valid = slot < p.N;
    size_t g = rbase + slot;
    const uint4 *__restrict__ lst = TILED ? (const uint4 *)p.nbr16 + (size_t)wrow.x * 64 + lane : (const uint4 *)p.nbr + (size_t)(g >> 6) * NCL * 64 + (g & 63);
    const uint4 *__restrict__ adj = (const uint4 *)p.badj + (TILED ? gw6 * NCB * 64 + lane : (size_t)(g >> 6) * NCB * 64 + (g & 63));
    float4 xi4 = make_float4(0.f, 0.f, 0.f, 0.f), x0 = xi4;
    unsigned meta = 0, oid = 0;
    float mu = p.mob_uniform;
    if (!TILED && valid) {
        xi4 = p.pos_in[g];
        meta = p.meta[g];
        if (MODE != GD_MODE_ENERGY) oid = p.orig[g];
        if (MODE == GD_MODE_STEP) x0 = p.xb[g];
        if (MODE == GD_MODE_STEP && p.mob_uniform < 0.f) mu = p.mob[g];
        if (p.has_bonds) adj0 = adj[0];
        if (p.pair.enabled) { qa = lst[0]; qb = lst[64]; }
    }
    unsigned local = 0, nA = 0, nAq = 0, nB = 0;      // tiled: block-local slot of the thread's bead, chunks of the near and far class (nAq: near entries in fours)
    if (TILED) {
        local = (mo.x >> 12) & 0x1ffu; nAq = mo.x >> 21; nA = (nAq + 1u) >> 1; nB = mo.y >> 26;
        meta = mo.x & 0xfffu;                    // degree | point-source mask << 8 (the generic layout without the length)
        oid = mo.y & GD_REC_ID_MASK;
    }
    // Brownian noise needs only (seed, bead, step, replica): it is generated here, while the tile DMAs
    // and the per-bead loads are in flight (the step index comes from a uniform scalar load).  Tiled path: whether the
    // thread owns a bead is not known yet (that is in the build-position record); threads without one have bead id 0.
    float3 z = make_float3(0.f, 0.f, 0.f);
    auto draw_noise = [&]() {
#ifdef GD_REPLAY
        if (RP_MEM) return;
#endif
#if GD_ABL == 53
        if (MODE == GD_MODE_STEP) return;      // (census build: no noise)
#endif
        if (MODE == GD_MODE_STEP && (TILED ? oid != GD_REC_ID_MASK : valid) && p.kT > 0.f) {
            if (p.noise_mode == NOISE_PHILOX) {
                const long long step_now = ctx_step0 + (ctx_pending0 ? 1 : 0);
                z = p.seeds ? philox_normal3(p.seeds[r], oid, step_now + 1, 0u) : philox_normal3(p.seed, oid, step_now + 1, r);
            } else if (p.noise_mode == NOISE_HOST) {
                const float *h = p.host_noise + ((size_t)r * p.N + oid) * 3;
                z = make_float3(h[0], h[1], h[2]);
            }
        }
    };
    // Wave 0 of a tiled stepping block carries the block's context work (pending callback, float constants) in front of the barrier,
    // on top of what every wave does there; the other seven wait for it (section stamps: 12 % of a wave's life).  Its noise is needed
    // by the update at the very end only: wave 0 draws it behind the barrier, so that what it does in front of the barrier is about
    // what the others do (record, DMA issue, noise | record, DMA issue, context).
    const bool noise_late = TILED && MODE == GD_MODE_STEP && wid == 0;
    if (!noise_late) draw_noise();
    GD_STAMP_USE3(z.x, z.y, z.z);
    GD_STAMP(10);     // noise
    // wave 0: pending callback + float copy of the context for the block.  After the noise: its vector loads sit behind the
    // DMAs in the vmcnt order, so this is where wave 0 waits for its share of the tile
    if (wid == 0) {
        DevCtx c = p.ctx_in[r];
        if (MODE == GD_MODE_STEP) {
            if (c.pending) apply_callback(c, p, r, lane);
            if (lane == 0 && blk == 0) { DevCtx o = c; o.pending = 1; p.ctx_out[r] = o; }
        }
        if (lane == 0) fill_ctxf(s_ctx, c, p);
    }
    GD_STAMP(0);      // prologue: loads issued, tile DMA issued, noise
    __syncthreads();
    GD_STAMP(1);      // barrier (tile arrival)
    if (noise_late) draw_noise();
    if (TILED) {
        valid = oid != GD_REC_ID_MASK;
        slot = valid ? blk * GD_BLOCK + local : p.N;
        g = rbase + slot;
        if (valid) {
            // (a truncated tile holds other beads at these indices: take the bead's position from memory, so that the
            // rolled-back chunk at least leaves plausible positions to the builds that follow in it)
#if GD_ABL == 41 || GD_ABL == 43
            xi4 = p.pos_in[g];
#else
            xi4 = tile_ok ? s_tile[own_base + local] : p.pos_in[g];
#endif
            x0 = rec;
            if (MODE == GD_MODE_STEP && p.mob_uniform < 0.f) mu = p.mob[g];
        }
    }

    // (census builds, make abl N=51 ... 54: the stepping kernel without its pair / bond / wall section, or without the noise -- timing and
    // counter builds only, wrong physics: the section's dynamic instruction count is the product's minus the build's, tools/census_pmc.sh)
#if GD_ABL == 51
    constexpr unsigned STEP_TERMS = 63u & ~TERM_PAIR;
#elif GD_ABL == 52
    constexpr unsigned STEP_TERMS = 63u & ~(TERM_BOND | TERM_DYNAMIC);
#elif GD_ABL == 54
    constexpr unsigned STEP_TERMS = 63u & ~TERM_WALL;
#else
    constexpr unsigned STEP_TERMS = 63u;
#endif
    const unsigned mask = (MODE == GD_MODE_STEP) ? STEP_TERMS : p.term_mask;

    float3 F = make_float3(0.f, 0.f, 0.f);
    float3 react = make_float3(0.f, 0.f, 0.f);
    float E = 0.f;
    float disp2 = 0.f, dnew2 = 0.f;

    if (valid) {
        const float3 xi = make_float3(xi4.x, xi4.y, xi4.z);
        const float2 *__restrict__ rab = p.ab + rbase;
        const float2 abi = p.packed_ab ? unpack_ab(xi4.w) : p.ab[g];

        // ---- non-bonded pairs over the Verlet list (a3, a5)
        if (p.pair.enabled && (mask & TERM_PAIR)) {
            const float inv_sa2 = s_ctx.p_inv_sa2, inv_sb2 = s_ctx.p_inv_sb2;      // (block-uniform: computed once, fill_ctxf)
            const float cut = s_ctx.p_cut, cut2 = cut * cut;
            const float ca = 6.0f * p.pair.eps_a * inv_sa2, cb = 24.0f * p.pair.eps_b * inv_sb2;
            const float hca = 0.5f * ca, hcb = 0.5f * cb, Ai = hca * abi.x, Bi = hcb * abi.y;
            if (MODE == GD_MODE_STEP) {
                // Verlet-skin check: the list is complete for this force evaluation iff every
                // bead moved less than (rv - cutoff)/2 since the build.
                const float dx = xi.x - x0.x, dy = xi.y - x0.y, dz = xi.z - x0.z;
                disp2 = dx * dx + dy * dy + dz * dz;
                const float lim = 0.5f * (p.rv - cut);
                if (!(lim > 0.f) || disp2 > lim * lim) p.flags[r * GD_NFLAGS + GD_FLAG_VIOLATION] = 1u;
            }
            const unsigned cnt = meta >> 16;             // generic path: exact length (tiled: chunk count nA)
            // Pair lists are stored in chunks of 16 bytes per bead, wave-interleaved:
            // chunk c of bead g is uint4 #((g/64)*NC + c)*64 + g%64  (one coalesced 1 KiB read per wave).
            // Tiled: 8 x u16 tile indices per chunk; generic: 4 x u32 slots per chunk.
            const bool mix = (PK == 1) || (PK == 0 && p.pair.mix != 0);
            // The list is padded to a multiple of GD_UNROLL with the bead's own index (zero
            // displacement => zero force), so every batch issues its index loads and its
            // neighbour reads together and the loop body has no bounds test.
            // Tiled lists: near class (d0 < rn at the build) in chunks 0 .. nA-1, far class in chunks NCL-1 .. NCL-nB.  A pair
            // exerts no force while d0 >= T = cutoff + D_i + D_j (displacements since the build; D_j <= the replica's running
            // maximum, D_i is this bead's own): the far chunks are walked once T exceeds rn -- the last steps of an interval,
            // and only by the waves that hold a bead that has moved that far.  Skipped entries have exactly zero force (the
            // soft cores clamp), so the result does not depend on when the far class joins.  Force / energy evaluations
            // take both classes.
            const unsigned nchA = nA;
            unsigned nchB = nB;
            if (MODE == GD_MODE_STEP && TILED) {
                // T = cutoff + D_i + D < rn  <=>  D_i < S with the block-uniform S = rn - cutoff - D (compared in squares: one square
                // root per block instead of two per bead)
                const float S = p.rn * (1.0f / 1.00005f) - cut - __builtin_amdgcn_sqrtf(dmax0);
                if (S > 0.f && disp2 < S * S) nchB = 0u;
            }
            const unsigned nch = nchA + nchB;
            const unsigned cntp = TILED ? nch * GD_UNROLL : (cnt + GD_UNROLL - 1u) & ~(GD_UNROLL - 1u);
            const unsigned self_e = TILED ? (S16 ? (own_base + local) << 4 : own_base + local) : 0u;     // the padding entries (energy mode skips them)
            // software pipeline: the chunk(s) of the next batch are in flight while one is processed
            // (a second batch of look-ahead bought nothing and costs the registers of one occupancy step)
            // (tiled: chunk c of the thread = wave base + c KiB + lane x 16 bytes -- a scalar base and a 32-bit offset, no per-lane
            // 64-bit pointer lives across the loop)
            const char *lst_w = (const char *)((const uint4 *)p.nbr16 + (size_t)wrow.x * 64);
#ifdef GD_REPLAY
            float4 rp_x = make_float4(xi.x + 0.11f, xi.y - 0.07f, xi.z + 0.05f, xi4.w);      // (RP_ALU: the neighbour every entry stands for)
            auto chunk = [&](unsigned c) -> uint4 {
                if (RP_ALU) { const unsigned e2 = (((own_base + local) << 4) & 0xffffu) * 0x10001u; return make_uint4(e2, e2, e2, e2 + c * 0u); }
                return nt_load((const uint4 *)(lst_w + (c * 1024u + lane * 16u)));
            };
            if (RP_ALU) qa = chunk(0u);
#else
            auto chunk = [&](unsigned c) -> uint4 { return nt_load((const uint4 *)(lst_w + (c * 1024u + lane * 16u))); };
#endif
            if (TILED && nchA == 0u && nch != 0u) qa = chunk(NCL - 1u);      // (no near chunk: the first one is a far chunk)
            for (unsigned k0 = 0; k0 < cntp; k0 += GD_UNROLL) {
                unsigned jj[GD_UNROLL];
                float4 xjv[GD_UNROLL];
                const uint4 q = qa, q0 = qa, q1 = qb;
                if (k0 + GD_UNROLL < cntp) {
                    if (TILED) { const unsigned c1 = k0 / 8 + 1; qa = chunk(c1 < nchA ? c1 : NCL - 1u - (c1 - nchA)); }
                    else { qa = lst[(size_t)(k0 / 4 + 2) * 64]; qb = lst[(size_t)(k0 / 4 + 3) * 64]; }
                }
                if (TILED) {
                    jj[0] = q.x & 0xffffu; jj[1] = q.x >> 16; jj[2] = q.y & 0xffffu; jj[3] = q.y >> 16;
                    jj[4] = q.z & 0xffffu; jj[5] = q.z >> 16; jj[6] = q.w & 0xffffu; jj[7] = q.w >> 16;
                } else {
                    jj[0] = q0.x; jj[1] = q0.y; jj[2] = q0.z; jj[3] = q0.w; jj[4] = q1.x; jj[5] = q1.y; jj[6] = q1.z; jj[7] = q1.w;
                }
              // the batch of eight in two halves on the 64-register form: the reads of the second half are not hoisted over the
              // arithmetic of the first
              constexpr int GD_HALF = (TILED && S16 && MODE == GD_MODE_STEP && GD_STEP_WAVES == 8) ? 4 : (int)GD_UNROLL;
              // the near class is counted in fours: the upper half of its last chunk is padding for a bead with an odd count, and
              // the threads of a block are ordered by that count -- most waves skip the half batch altogether
              const bool upper = !(TILED && k0 / GD_UNROLL + 1u == nchA && (nAq & 1u));
#pragma unroll
              for (int uh = 0; uh < (int)GD_UNROLL; uh += GD_HALF) {
                if (GD_HALF != (int)GD_UNROLL && uh != 0 && __builtin_amdgcn_ballot_w64(upper) == 0ull) break;
                if (GD_HALF != (int)GD_UNROLL) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = uh; u < uh + GD_HALF; u++) {
#ifdef GD_REPLAY
                    if (RP_ALU) { GD_LAUNDER4(rp_x); xjv[u] = rp_x; continue; }
#endif
                    xjv[u] = !TILED ? rpos[jj[u]] : S16 ? *(const float4 *)((const char *)s_tile + jj[u]) : s_tile[jj[u]];
                }
#pragma unroll
                for (int u = uh; u < uh + GD_HALF; u++) {
                    const float4 xj = xjv[u];
                    const unsigned j = jj[u];
#ifdef GD_REPLAY
                    if (RP_MEM) { GD_CONSUME4(xj); continue; }
#endif
                    float3 d = make_float3(xi.x - xj.x, xi.y - xj.y, xi.z - xj.z);
                    if (PERIODIC) d = min_image(d, p.box, p.inv_box);
                    const float r2 = d.x * d.x + d.y * d.y + d.z * d.z;
                    float wa = 1.0f, wb = 1.0f;
                    if (PK != 0) {
                        float waca = ca, wbcb = cb;
                        if (PK == 1) {
                            const float2 abj = (TILED || p.packed_ab) ? unpack_ab(xj.w) : rab[j];
                            waca = fmaf(abj.x, hca, Ai); wbcb = fmaf(abj.y, hcb, Bi);      // (a_i+a_j)/2 * ca, (b_i+b_j)/2 * cb
                            if (MODE == GD_MODE_ENERGY) { wa = 0.5f * (abi.x + abj.x); wb = 0.5f * (abi.y + abj.y); }
                        }
                        const float f = softcore_2383(r2, inv_sa2, inv_sb2, waca, wbcb);
                        F.x = fmaf(f, d.x, F.x); F.y = fmaf(f, d.y, F.y); F.z = fmaf(f, d.z, F.z);
                        if (MODE == GD_MODE_ENERGY && (TILED ? j != self_e : k0 + u < cnt))
                            E += 0.5f * softcore_2383_energy(r2, inv_sa2, inv_sb2, p.pair.eps_a, p.pair.eps_b, wa, wb);
                    } else if (r2 < cut2) {
                        if (mix) {
                            const float2 abj = (TILED || p.packed_ab) ? unpack_ab(xj.w) : rab[j];
                            wa = 0.5f * (abi.x + abj.x); wb = 0.5f * (abi.y + abj.y);
                        }
                        float ea, fa, eb, fb;
                        softcore(p.pair.eps_a, inv_sa2, p.pair.p_a, p.pair.q_a, r2, ea, fa);
                        softcore(p.pair.eps_b, inv_sb2, p.pair.p_b, p.pair.q_b, r2, eb, fb);
                        const float f = wa * fa + wb * fb;
                        F.x += f * d.x; F.y += f * d.y; F.z += f * d.z;
                        if (MODE == GD_MODE_ENERGY && (TILED ? j != self_e : k0 + u < cnt)) E += 0.5f * (wa * ea + wb * eb);
                    }
                }
              }
            }
        }

        GD_STAMP(2);  // pair loop
        // ---- bonded pairs (a6, a12): per-bead adjacency, each bond evaluated from both ends
        if (p.has_bonds && (mask & (TERM_BOND | TERM_DYNAMIC))) {
            const unsigned deg = meta & 0xffu;
            const float inv_bs2 = s_ctx.inv_bond_scale2;
            if (TILED) {
                // Tiled path, branch-free: two adjacency entries per batch, their partner positions and bond-type records all
                // requested from LDS before the first is used (one round of latency per batch, no divergent branches).  An entry
                // beyond the bead's degree stands for the bead itself (zero separation: zero force from every bond form); partners
                // outside the tile are a rare, wave-uniform detour through global memory.
                const unsigned own_idx = own_base + local;
                for (unsigned k0 = 0; k0 < deg; k0 += 4) {
                    // (chunks beyond the first -- a bead with more than four bonds -- by scalar base + 32-bit lane offset, like the lists)
#ifdef GD_REPLAY
                    uint4 aq = (k0 == 0 || RP_ALU) ? adj0 : *(const uint4 *)((const char *)((const uint4 *)p.badj + gw6 * NCB * 64) + ((k0 >> 2) * 1024u + lane * 16u));
                    if (RP_ALU) { const unsigned el = GD_ADJ_LOCAL | own_idx; aq = make_uint4(el, el, el, el); }
#else
                    const uint4 aq = k0 == 0 ? adj0 : *(const uint4 *)((const char *)((const uint4 *)p.badj + gw6 * NCB * 64) + ((k0 >> 2) * 1024u + lane * 16u));
#endif
                    const unsigned ents4[4] = {aq.x, aq.y, aq.z, aq.w};
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        if (h == 1 && __builtin_amdgcn_ballot_w64(k0 + 2u < deg) == 0ull) break;
                        float4 xjs[2], ta[2];
                        float2 tb[2];
                        unsigned ty[2];
                        bool on[2], away = false;
#pragma unroll
                        for (int u = 0; u < 2; u++) {
                            on[u] = k0 + 2u * h + u < deg;
                            const unsigned e = on[u] ? ents4[2 * h + u] : GD_ADJ_LOCAL;
                            const bool loc = (e & GD_ADJ_LOCAL) != 0u;
#ifdef GD_REPLAY
                            if (RP_ALU) { float4 t = make_float4(xi.x + 0.13f, xi.y + 0.09f, xi.z - 0.11f, xi4.w); GD_LAUNDER4(t); xjs[u] = t; } else
#endif
                            xjs[u] = s_tile[(on[u] && loc) ? (e & GD_ADJ_MASK) : own_idx];
                            ty[u] = (e >> GD_ADJ_SHIFT) & (GD_MAX_BOND_TYPES - 1);
                            ta[u] = *(const float4 *)&s_bt[ty[u]];
                            tb[u] = *(const float2 *)&s_bt[ty[u]].flags;
                            away |= !loc;
                        }
                        if (__builtin_amdgcn_ballot_w64(away) != 0ull) {
#pragma unroll
                            for (int u = 0; u < 2; u++)
                                if (on[u] && !(ents4[2 * h + u] & GD_ADJ_LOCAL)) xjs[u] = rpos[ents4[2 * h + u] & GD_ADJ_MASK];
                        }
#pragma unroll
                        for (int u = 0; u < 2; u++) {
                            const float4 xj = xjs[u];
#ifdef GD_REPLAY
                            if (RP_MEM) { GD_CONSUME4(xj); GD_CONSUME4(ta[u]); asm volatile("" :: "v"(tb[u].x), "v"(tb[u].y)); continue; }
#endif
                            const unsigned flags = __float_as_uint(tb[u].x);
                            float3 d = make_float3(xi.x - xj.x, xi.y - xj.y, xi.z - xj.z);
                            if (PERIODIC && (flags & 4u)) d = min_image(d, p.box, p.inv_box);
                            const float r2 = d.x * d.x + d.y * d.y + d.z * d.z;
                            const bool scaled = (flags & 2u) != 0u;
                            float K = ta[u].x, l = ta[u].z;
                            if (!p.bonds_premixed) {      // (uniform: the host could not resolve the AB mixing per bond)
                                const float2 abj = unpack_ab(xj.w);
                                const bool mixed = (flags & 1u) != 0u;
                                const float a = mixed ? 0.5f * (abi.x + abj.x) : 1.0f, b = mixed ? 0.5f * (abi.y + abj.y) : 0.0f;
                                K = a * ta[u].x + b * ta[u].y; l = a * ta[u].z + b * ta[u].w;
                            }
                            if (p.bonds_all_scaled) { K *= inv_bs2; l *= s_ctx.bond_scale; }      // (uniform: every bond set of the model scales with bond_scale)
                            else { K *= scaled ? inv_bs2 : 1.0f; l *= scaled ? s_ctx.bond_scale : 1.0f; }
                            // harmonic / spring / semispring in one form: elongation x = r - l clamped from below
                            // (r2 = 0 -- a padding entry, coincident beads -- gives x = -l and a finite fr on d = 0: zero force, like the guarded form)
                            const float inv_d = __builtin_amdgcn_rsqf(fmaxf(r2, 1e-30f));     // hardware rsq, 1 ulp
                            const float x = fmax_plain(fmaf(r2, inv_d, -l), tb[u].y);
                            float fr = -K * x * inv_d, e = 0.f;
                            if (MODE == GD_MODE_ENERGY) e = 0.5f * K * x * x;
                            if (p.has_softcore_bonds && s_bt[ty[u]].kind == POT_SOFTCORE) {      // (rare bond form; uniform test first)
                                const int pq = s_bt[ty[u]].pq;
                                softcore(K, 1.0f / (l * l), pq & 0xff, pq >> 8, r2, e, fr);
                            }
                            if (MODE != GD_MODE_STEP && !(mask & (flags >> 8))) { fr = 0.f; e = 0.f; }
                            F.x = fmaf(fr, d.x, F.x); F.y = fmaf(fr, d.y, F.y); F.z = fmaf(fr, d.z, F.z);
                            if (MODE == GD_MODE_ENERGY && on[u]) E += 0.5f * e;
                        }
                    }
                }
            } else
            // four adjacency entries (one 16-byte chunk) per round: the partner positions are fetched together, so
            // their LDS / global latencies overlap instead of adding up bond by bond
            for (unsigned k0 = 0; k0 < deg; k0 += 4) {
                const uint4 aq = k0 == 0 ? adj0 : adj[(size_t)(k0 >> 2) * 64];
                const unsigned ents[4] = {aq.x, aq.y, aq.z, aq.w};
                float4 xjs[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    xjs[u] = xi4;
                    if (k0 + u < deg) {
                        const unsigned j = ents[u] & GD_ADJ_MASK;
                        if (TILED && (ents[u] & GD_ADJ_LOCAL)) xjs[u] = s_tile[j]; else xjs[u] = rpos[j];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (k0 + u >= deg) break;
                    const unsigned ent = ents[u];
                    const unsigned j = ent & GD_ADJ_MASK;
                    const BondType bt = s_bt[(ent >> GD_ADJ_SHIFT) & (GD_MAX_BOND_TYPES - 1)];
                    if (!(mask & ((unsigned)bt.flags >> 8))) continue;
                    const float4 xj = xjs[u];
                    float3 d = make_float3(xi.x - xj.x, xi.y - xj.y, xi.z - xj.z);
                    if (PERIODIC && (bt.flags & 4)) d = min_image(d, p.box, p.inv_box);
                    const float r2 = d.x * d.x + d.y * d.y + d.z * d.z;
                    float K = bt.ka, l = bt.la;
                    if (bt.flags & 1) {
                        const float2 abj = (TILED || p.packed_ab) ? unpack_ab(xj.w) : rab[j];
                        const float a = 0.5f * (abi.x + abj.x), b = 0.5f * (abi.y + abj.y);
                        K = a * bt.ka + b * bt.kb; l = a * bt.la + b * bt.lb;
                    }
                    if (bt.flags & 2) { K = K * inv_bs2; l = l * s_ctx.bond_scale; }
                    float e = 0.f, fr;
                    if (p.has_softcore_bonds && bt.kind == POT_SOFTCORE) {
                        softcore(K, 1.0f / (l * l), bt.pq & 0xff, bt.pq >> 8, r2, e, fr);
                    } else {
                        // harmonic / spring / semispring in one branch-free form: elongation x = r - l clamped from below
                        const float inv_d = r2 > 0.0f ? __builtin_amdgcn_rsqf(r2) : 0.0f;     // hardware rsq, 1 ulp
                        const float x = fmaxf(fmaf(r2, inv_d, -l), bt.xmin);
                        fr = -K * x * inv_d;
                        if (MODE == GD_MODE_ENERGY) e = 0.5f * K * x * x;
                    }
                    F.x += fr * d.x; F.y += fr * d.y; F.z += fr * d.z;
                    if (MODE == GD_MODE_ENERGY) E += 0.5f * e;
                }
            }
        }

        GD_STAMP(3);  // bonds
        // ---- cosine bending over the (up to) three triplets this bead belongs to (a7)
        if (p.has_bend && (mask & TERM_BEND)) {
            const float4 be = p.bendE[g];
            if (be.x != 0.f || be.y != 0.f || be.z != 0.f) {
                const int4 c4 = p.chain[g];
                float3 xm2 = xi, xm1 = xi, xp1 = xi, xp2 = xi;
                auto fetch = [&](int c) -> float3 {
                    float4 t;
                    if (TILED && (c & GD_CHAIN_LOCAL)) t = s_tile[c & 0xffff]; else t = rpos[c];
                    return make_float3(t.x, t.y, t.z);
                };
                if (c4.x >= 0) xm2 = fetch(c4.x);
                if (c4.y >= 0) xm1 = fetch(c4.y);
                if (c4.z >= 0) xp1 = fetch(c4.z);
                if (c4.w >= 0) xp2 = fetch(c4.w);
                float3 fi, fk; float cs;
                if (be.x != 0.f) {   // (i-2, i-1, i): this bead is the last one
                    const float3 d1 = make_float3(xm1.x - xm2.x, xm1.y - xm2.y, xm1.z - xm2.z);
                    const float3 d2 = make_float3(xi.x - xm1.x, xi.y - xm1.y, xi.z - xm1.z);
                    if (bend_forces(d1, d2, be.x, fi, fk, cs)) { F.x += fk.x; F.y += fk.y; F.z += fk.z; }
                }
                if (be.y != 0.f) {   // (i-1, i, i+1): middle
                    const float3 d1 = make_float3(xi.x - xm1.x, xi.y - xm1.y, xi.z - xm1.z);
                    const float3 d2 = make_float3(xp1.x - xi.x, xp1.y - xi.y, xp1.z - xi.z);
                    if (bend_forces(d1, d2, be.y, fi, fk, cs)) {
                        F.x -= fi.x + fk.x; F.y -= fi.y + fk.y; F.z -= fi.z + fk.z;
                        if (MODE == GD_MODE_ENERGY) E += be.y * (1.0f - cs);
                    }
                }
                if (be.z != 0.f) {   // (i, i+1, i+2): first
                    const float3 d1 = make_float3(xp1.x - xi.x, xp1.y - xi.y, xp1.z - xi.z);
                    const float3 d2 = make_float3(xp2.x - xp1.x, xp2.y - xp1.y, xp2.z - xp1.z);
                    if (bend_forces(d1, d2, be.z, fi, fk, cs)) { F.x += fi.x; F.y += fi.y; F.z += fi.z; }
                }
            }
        }

        // ---- point sources (a8)
        if (p.nps > 0 && (mask & TERM_POINT)) {
            const unsigned pm = (meta >> 8) & 0xffu;
            for (int s = 0; s < p.nps; s++) {
                if (!((pm >> s) & 1u)) continue;
                const float3 d = make_float3(xi.x - p.ps[s].p[0], xi.y - p.ps[s].p[1], xi.z - p.ps[s].p[2]);
                const float r2 = d.x * d.x + d.y * d.y + d.z * d.z;
                float e, fr;
                bond_pot(p.ps[s].kind, p.ps[s].k, p.ps[s].b, 2, 1, r2, e, fr);
                F.x += fr * d.x; F.y += fr * d.y; F.z += fr * d.z;
                if (MODE == GD_MODE_ENERGY) E += e;
            }
        }

        GD_STAMP(4);  // bending + point sources
        // ---- ellipsoid wall (a9): second-order nearest-surface construction
        // (5-sim-genome/src/analyze_lamina/geometry.py:13-28), conjugate form u = C/(B+sqrt(B^2-AC)).
#ifdef GD_REPLAY
        if (!RP_MEM)
#endif
        if (p.wall.enabled && (mask & TERM_WALL)) {
            const float ia = s_ctx.inv_semi2[0], ib = s_ctx.inv_semi2[1], ic = s_ctx.inv_semi2[2];
            const float3 s1 = make_float3(xi.x * ia, xi.y * ib, xi.z * ic);
            const float C1 = xi.x * s1.x + xi.y * s1.y + xi.z * s1.z;
            // waves of interior beads (the slots are cell-sorted) skip the nearest-point construction altogether
            if (__builtin_amdgcn_ballot_w64(C1 >= s_ctx.near2) != 0ull) {
            // C = C1 - 1 is a difference of two numbers near 1 for every bead the wall acts on (|C| < 0.05): its numerator in fp64
            // (full-rate on this chip) from the fp64 semiaxes; everything downstream is a product, fp32 is enough there
            const double xd = (double)xi.x, yd = (double)xi.y, zd = (double)xi.z;
            const float C = (float)fma(xd * xd, s_ctx.w_q[0], fma(yd * yd, s_ctx.w_q[1], fma(zd * zd, s_ctx.w_q[2], -s_ctx.w_q[3]))) * s_ctx.w_inv_q3;
            const float B = s1.x * s1.x + s1.y * s1.y + s1.z * s1.z;
            const float A = s1.x * s1.x * ia + s1.y * s1.y * ib + s1.z * s1.z * ic;
            // (hardware sqrt / rcp, 1 ulp: the IEEE-exact sequences are ~10 instructions each)
            const float den = B + __builtin_amdgcn_sqrtf(fmaxf(B * B - A * C, 0.f));
            if (den > 0.f && C != 0.f) {
                const float u = C * __builtin_amdgcn_rcpf(den);
                const float3 dl = make_float3(u * s1.x, u * s1.y, u * s1.z);
                const float r2 = dl.x * dl.x + dl.y * dl.y + dl.z * dl.z;
                float e = 0.f, fr = 0.f;
                if (C < 0.f) {
                    const float wa = 0.5f * (abi.x + p.wall.wall_a), wb = 0.5f * (abi.y + p.wall.wall_b);
                    if (p.wall.fast2383 && MODE != GD_MODE_ENERGY) {
                        // the wall's soft cores are the pair family (<2,3> + <8,3>, half diameters): branch-free form
                        fr = softcore_2383(r2, s_ctx.w_inv_sa2, s_ctx.w_inv_sb2, wa * s_ctx.w_ca, wb * s_ctx.w_cb);
                    } else {
                        float ea, fa, eb, fb;
                        softcore(p.wall.eps_a, s_ctx.w_inv_sa2, p.wall.p_a, p.wall.q_a, r2, ea, fa);
                        softcore(p.wall.eps_b, s_ctx.w_inv_sb2, p.wall.p_b, p.wall.q_b, r2, eb, fb);
                        e = wa * ea + wb * eb; fr = wa * fa + wb * fb;
                    }
                } else {
                    e = 0.5f * p.wall.packing_spring * r2; fr = -p.wall.packing_spring;
                }
                if (fr != 0.f) {
                    const float3 fw = make_float3(fr * dl.x, fr * dl.y, fr * dl.z);
                    F.x += fw.x; F.y += fw.y; F.z += fw.z;
                    // axial_reaction_k = -F_k q_k / a_k, q = contact point on the surface
                    react.x = -fw.x * (xi.x - dl.x) * s_ctx.inv_semi[0];
                    react.y = -fw.y * (xi.y - dl.y) * s_ctx.inv_semi[1];
                    react.z = -fw.z * (xi.z - dl.z) * s_ctx.inv_semi[2];
                }
                if (MODE == GD_MODE_ENERGY) E += e;
            }
            }
        }

        // ---- inner spherical wall (excluded core of the 4-sim-ab sphere model): soft repulsion outside, harmonic inside
        if (p.wall.inner_enabled && (mask & TERM_WALL)) {
            const float rr2 = xi.x * xi.x + xi.y * xi.y + xi.z * xi.z;
            const float reach = p.wall.in_radius + 0.5f * fmaxf(p.wall.in_sigma_a, p.wall.in_sigma_b);
            if (__builtin_amdgcn_ballot_w64(rr2 < reach * reach) != 0ull && rr2 > 0.f) {
                const float inv_r = rsqrtf(rr2), rad = rr2 * inv_r, gap = rad - p.wall.in_radius;
                const float r2 = gap * gap, gs = gap * inv_r;          // delta = gs * x
                float e = 0.f, fr = 0.f;
                if (gap > 0.f) {
                    const float sa = 0.5f * p.wall.in_sigma_a, sb = 0.5f * p.wall.in_sigma_b;
                    const float wa = 0.5f * (abi.x + p.wall.in_wall_a), wb = 0.5f * (abi.y + p.wall.in_wall_b);
                    float ea, fa, eb, fb;
                    softcore(p.wall.in_eps_a, sa > 0.f ? 1.0f / (sa * sa) : 0.f, p.wall.in_p_a, p.wall.in_q_a, r2, ea, fa);
                    softcore(p.wall.in_eps_b, sb > 0.f ? 1.0f / (sb * sb) : 0.f, p.wall.in_p_b, p.wall.in_q_b, r2, eb, fb);
                    e = wa * ea + wb * eb; fr = wa * fa + wb * fb;
                } else if (gap < 0.f) {
                    e = 0.5f * p.wall.in_spring * r2; fr = -p.wall.in_spring;
                }
                F.x += fr * gs * xi.x; F.y += fr * gs * xi.y; F.z += fr * gs * xi.z;
                if (MODE == GD_MODE_ENERGY) E += e;
            }
        }

        if (MODE == GD_MODE_STEP) {
            GD_STAMP(5);  // wall
            // ---- overdamped Langevin / Euler-Maruyama (a1): x += mu F dt + sqrt(2 mu kT dt) xi
            const float mu_dt = mu * p.dt;
            const float sg = s_ctx.sg_uniform >= 0.f ? s_ctx.sg_uniform : sqrtf(2.0f * p.kT * mu_dt);
            const float ex = mu_dt * F.x + sg * z.x, ey = mu_dt * F.y + sg * z.y, ez = mu_dt * F.z + sg * z.z;
            float nx = xi.x + ex, ny = xi.y + ey, nz = xi.z + ez;
            if (p.comp) {
                // Compensated update (uniform branch; gd_run selects it when the increment of a step is within a few ulp of an
                // fp32 coordinate: the reference's deterministic fine-sampling run, T = 0 and dt = 1e-7,
                // simulation_fine_sampling/simulation_driver.cc:30-34, moves a bead by 1e-7 ... 2e-6 per step at |x| of 3 ... 8,
                // i.e. by 0.1 ... 4 ulp).  The bead's true position is x + lo; the increment is added to the residual first and
                // the pair is re-normalised by a two-sum (Knuth; exact in round-to-nearest whatever the magnitudes), so that no
                // part of mu F dt is lost.  Forces are evaluated on x alone: |lo| <= ulp(x) / 2, the rounding every fp32
                // position carries anyway.  The residual lives by BEAD index, outside the cell sort.
                float4 *lp = p.lo + ((size_t)r * p.N + oid);
                const float4 l = *lp;
                const float tx = l.x + ex, ty = l.y + ey, tz = l.z + ez;
                nx = xi.x + tx; ny = xi.y + ty; nz = xi.z + tz;
                const float bx = nx - xi.x, by = ny - xi.y, bz = nz - xi.z;
                *lp = make_float4((xi.x - (nx - bx)) + (tx - bx), (xi.y - (ny - by)) + (ty - by), (xi.z - (nz - bz)) + (tz - bz), 0.f);
            }
            if (TILED) *(float4 *)((char *)(p.pos_out + rbase + (size_t)blk * GD_BLOCK) + local * 16u) = make_float4(nx, ny, nz, xi4.w);
            else p.pos_out[g] = make_float4(nx, ny, nz, xi4.w);
            // displacement of the NEW position since the build, bounded by the triangle inequality (the build position
            // need not stay in registers): |x + dx - x0| <= |x - x0| + |dx|
            // ((d + e)^2 = d^2 + e^2 + 2 sqrt(d^2 e^2): one square root)
            if (TILED) { const float e2 = ex * ex + ey * ey + ez * ez; dnew2 = (disp2 + e2 + 2.0f * __builtin_amdgcn_sqrtf(disp2 * e2)) * 1.000002f; }
        } else if (MODE == GD_MODE_FORCE) {
            p.fout[(size_t)r * p.N + oid] = make_float4(F.x, F.y, F.z, 0.f);
        }
    }

    GD_STAMP(6);      // integrate + store
    GD_STAMP_END((unsigned long long *)p.fout);
    // ---- block reductions: wall reaction partial (deterministic), energy, max displacement
    // running maximum of the displacement since the build, for the far-class test of the NEXT step: one atomic per block
    // (folded into the reaction reduction) or per wave, only from those that raise the value this launch started from, each
    // replica's word on a cache line of its own (right after a build every wave raises it: 60 000 atomics on four lines
    // took longer than the step itself)
    const bool track = MODE == GD_MODE_STEP && TILED, record = MODE == GD_MODE_STEP && p.record_disp;
    const float dwave = track ? wave_max_f(dnew2) : 0.f;
    // the interval adaptation's measurement (last step of an interval only): exact displacement of the positions read.  It
    // takes the block reduction's fourth column on that step; the running maximum then goes by wave (late in an interval few
    // waves still raise it)
    const float rwave = record ? wave_max_f(disp2) : 0.f;
    if (p.wall.enabled && MODE != GD_MODE_ENERGY) {
        float sx = 0.f, sy = 0.f, sz = 0.f;      // (waves of interior beads have nothing to add)
        if (__builtin_amdgcn_ballot_w64(react.x != 0.f || react.y != 0.f || react.z != 0.f) != 0ull) { sx = wave_sum_f(react.x); sy = wave_sum_f(react.y); sz = wave_sum_f(react.z); }
        if (lane == 0) { s_red[wid][0] = sx; s_red[wid][1] = sy; s_red[wid][2] = sz; s_red[wid][3] = record ? rwave : dwave; }
        __syncthreads();
        if (tid == 0) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            float m = 0.f;
            for (int w = 0; w < GD_BLOCK / 64; w++) { v.x += s_red[w][0]; v.y += s_red[w][1]; v.z += s_red[w][2]; m = fmaxf(m, s_red[w][3]); }
            p.react_out[(size_t)r * p.nblk + blk] = v;
            if (record) atomicMax(&p.flags[r * GD_NFLAGS + GD_FLAG_MAXDISP2], __float_as_uint(m));
            else if (track && m > dmax0) atomicMax(&p.dmax[r * GD_DMAX_STRIDE], __float_as_uint(m));
        }
        if (record && track && lane == 0 && dwave > dmax0) atomicMax(&p.dmax[r * GD_DMAX_STRIDE], __float_as_uint(dwave));
    } else if (lane == 0) {
        if (track && dwave > dmax0) atomicMax(&p.dmax[r * GD_DMAX_STRIDE], __float_as_uint(dwave));
        if (record) atomicMax(&p.flags[r * GD_NFLAGS + GD_FLAG_MAXDISP2], __float_as_uint(rwave));
    }
    if (MODE == GD_MODE_ENERGY) {
        const double se = wave_sum_d((double)E);
        if (lane == 0) s_e[wid] = se;
        __syncthreads();
        if (tid == 0) {
            double v = 0;
            for (int w = 0; w < GD_BLOCK / 64; w++) v += s_e[w];
            p.epart[(size_t)r * p.nblk + blk] = v;
        }
    }
}

// One wave per replica.  mode 0: apply the pending callback of the last step (end of a gd_run chunk); mode 1: fold the
// reaction partials of a force evaluation into the context.
__global__ __launch_bounds__(64) void k_ctx(const StepParams p, int mode)
{
    const unsigned r = blockIdx.x, lane = threadIdx.x;
    DevCtx c = p.ctx_in[r];
    if (mode == 1) {
        double rx = 0, ry = 0, rz = 0;
        for (unsigned b = lane; b < p.nblk; b += 64) {
            const float4 v = p.react_in[(size_t)r * p.nblk + b];
            rx += v.x; ry += v.y; rz += v.z;
        }
        c.react[0] = wave_sum_d(rx); c.react[1] = wave_sum_d(ry); c.react[2] = wave_sum_d(rz);
    } else if (c.pending) {
        apply_callback(c, p, r, lane);
    }
    if (lane == 0) p.ctx_out[r] = c;
}

template <int MODE>
static void launch_step_mode(const StepParams &p, hipStream_t st)
{
    const dim3 grid(p.cpb ? GD_XCDS * p.cpb * p.R : p.R * p.nblk), block(GD_BLOCK);
    const size_t lds = p.tiled ? (size_t)p.tile_cap * sizeof(float4) : 0;
    // (more than 64 KB of dynamic LDS: opted in per device at gd_create, gd_kernels_init_device below)
    const bool s16 = p.tiled && p.tile_cap < 4096u;      // must match k_fill's choice (gd_launch_build)
    // The LDS class of a launch is set by the LARGEST tile of any replica, and one tile a few beads over the three-block class
    // (3 312) costs every block a third of its occupancy (S-genome-62k x 64: a handful of tiles in the nucleus' centre).  While the
    // list entries stay 16-bit byte offsets (tile_cap < 4 096: the same kernel variant), the step is launched twice instead:
    // the blocks whose tile fits 3 312 with that much LDS, then the rest; each block returns at once from the launch it is not in.
    // The same one class up: tiles beyond 5 072 entries (dense states: one block per CU) next to tiles that fit two blocks per CU.
    // (Stepping only: force / energy evaluations are rare and run in the larger class.)
    // (Plain-index lists -- tiles of 4 096 entries and more, the periodic 1 kb model -- split the same way at 3 312: their kernel fits
    // 80 registers, so the blocks with small tiles run three to a CU next to the few large ones.)
    const unsigned split_at = !p.tiled ? 0u : (s16 && p.tile_cap > 3312u) ? 3312u : (!s16 && p.tile_cap > 5072u) ? 5072u : (!s16 && p.tile_cap > 3312u) ? 3312u : 0u;
    if (MODE == GD_MODE_STEP && split_at) {
        StepParams q = p;
        for (int half = 0; half < 2; half++) {
            q.tile_cap = half ? p.tile_cap : split_at; q.tile_lo = half ? split_at : 0u; q.tile_hi = half ? 0xffffffffu : split_at;
            const size_t lds_q = (size_t)q.tile_cap * sizeof(float4);
#define LS(PER, PK, S) hipLaunchKernelGGL((k_step<GD_MODE_STEP, PER, true, PK, S, true>), grid, block, lds_q, st, q)
            if (s16) {
                if (p.periodic) { if (p.pk == 1) LS(true, 1, true); else if (p.pk == 2) LS(true, 2, true); else LS(true, 0, true); }
                else { if (p.pk == 1) LS(false, 1, true); else if (p.pk == 2) LS(false, 2, true); else LS(false, 0, true); }
            } else {
                if (p.periodic) { if (p.pk == 1) LS(true, 1, false); else if (p.pk == 2) LS(true, 2, false); else LS(true, 0, false); }
                else { if (p.pk == 1) LS(false, 1, false); else if (p.pk == 2) LS(false, 2, false); else LS(false, 0, false); }
            }
#undef LS
        }
        return;
    }
#define L(PER, TIL, PK, S) hipLaunchKernelGGL((k_step<MODE, PER, TIL, PK, S>), grid, block, lds, st, p)
    if (p.periodic && p.tiled && s16) { if (p.pk == 1) L(true, true, 1, true); else if (p.pk == 2) L(true, true, 2, true); else L(true, true, 0, true); }
    else if (p.periodic && p.tiled) { if (p.pk == 1) L(true, true, 1, false); else if (p.pk == 2) L(true, true, 2, false); else L(true, true, 0, false); }
    else if (p.periodic) { if (p.pk == 1) L(true, false, 1, false); else if (p.pk == 2) L(true, false, 2, false); else L(true, false, 0, false); }
    else if (p.tiled && s16) { if (p.pk == 1) L(false, true, 1, true); else if (p.pk == 2) L(false, true, 2, true); else L(false, true, 0, true); }
    else if (p.tiled) { if (p.pk == 1) L(false, true, 1, false); else if (p.pk == 2) L(false, true, 2, false); else L(false, true, 0, false); }
    else { if (p.pk == 1) L(false, false, 1, false); else if (p.pk == 2) L(false, false, 2, false); else L(false, false, 0, false); }
#undef L
}

void gd_launch_step(const StepParams &p, int mode, hipStream_t st)
{
    if (mode == GD_MODE_STEP) launch_step_mode<GD_MODE_STEP>(p, st);
    else if (mode == GD_MODE_FORCE) launch_step_mode<GD_MODE_FORCE>(p, st);
    else launch_step_mode<GD_MODE_ENERGY>(p, st);
}

void gd_launch_finalize(const StepParams &p, int mode, hipStream_t st)
{
    hipLaunchKernelGGL(k_ctx, dim3(p.R), dim3(64), 0, st, p, mode);
}

// --------------------------------------------------------- neighbour search

// bounding box partials (open box): bbox[(r*nblk + blk)*6 + {lo xyz, hi xyz}].  One block covers FOUR chunks of 512 slots (four
// loads in flight per thread: the pass is latency-bound) and writes its partial into the first of their four entries, the neutral
// element into the others.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dppi(float v, float ident)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <bool MAX>
__device__ __forceinline__ float wave_minmax(float v)      // all 64 lanes active; result in every lane
{
    const float id = MAX ? -INFINITY : INFINITY;
#define GD_MM(a, b) (MAX ? fmaxf(a, b) : fminf(a, b))
    v = GD_MM(v, (dppi<0x111, 0xf>(v, id))); v = GD_MM(v, (dppi<0x112, 0xf>(v, id))); v = GD_MM(v, (dppi<0x114, 0xf>(v, id)));
    v = GD_MM(v, (dppi<0x118, 0xf>(v, id))); v = GD_MM(v, (dppi<0x142, 0xa>(v, id))); v = GD_MM(v, (dppi<0x143, 0xc>(v, id)));
#undef GD_MM
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__global__ __launch_bounds__(GD_BLOCK) void k_bbox(const BuildParams p)
{
    __shared__ float s_lo[GD_BLOCK / 64][3], s_hi[GD_BLOCK / 64][3];
    const unsigned nb4 = (p.nblk + 3u) / 4u;
    const unsigned r = blockIdx.x / nb4, blk0 = (blockIdx.x % nb4) * 4u;
    const unsigned lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    float4 x[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const unsigned slot = (blk0 + i) * GD_BLOCK + threadIdx.x;
        x[i] = slot < p.N ? p.pos_in[(size_t)r * p.Np + slot] : make_float4(INFINITY, INFINITY, INFINITY, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const bool on = (blk0 + i) * GD_BLOCK + threadIdx.x < p.N;
        lo[0] = fminf(lo[0], x[i].x); lo[1] = fminf(lo[1], x[i].y); lo[2] = fminf(lo[2], x[i].z);
        if (on) { hi[0] = fmaxf(hi[0], x[i].x); hi[1] = fmaxf(hi[1], x[i].y); hi[2] = fmaxf(hi[2], x[i].z); }
    }
    for (int k = 0; k < 3; k++) {
        const float l = wave_minmax<false>(lo[k]), h = wave_minmax<true>(hi[k]);
        if (lane == 0) { s_lo[wid][k] = l; s_hi[wid][k] = h; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        float l = s_lo[0][k], h = s_hi[0][k];
        for (int w = 1; w < GD_BLOCK / 64; w++) { l = fminf(l, s_lo[w][k]); h = fmaxf(h, s_hi[w][k]); }
        for (unsigned i = 0; i < 4u && blk0 + i < p.nblk; i++) {
            p.bbox[((size_t)r * p.nblk + blk0 + i) * 6 + k] = i == 0 ? l : INFINITY;
            p.bbox[((size_t)r * p.nblk + blk0 + i) * 6 + 3 + k] = i == 0 ? h : -INFINITY;
        }
    }
}

// one block per replica: its first wave reduces the bounding box and chooses the cell grid (every wave computes the same grid: the
// inputs are uniform), all GD_GRIDP_THREADS threads clear the cell counters (148 877 cells per replica on the 1 kb model: one wave
// took 38 us over them)
// the cell grid of one replica on the box (lo, lo + e): cells >= rv (1/kx of that in x, open boxes); the cell size grows until the cell
// count fits the allocation
__device__ __forceinline__ GridP grid_on(const BuildParams &p, float lo0, float lo1, float lo2, float e0, float e1, float e2)
{
    const int extra = p.periodic ? 0 : 1;
    if (!p.periodic) {
        if (!(e0 > 0.f)) e0 = p.rv;
        if (!(e1 > 0.f)) e1 = p.rv;
        if (!(e2 > 0.f)) e2 = p.rv;
    }
    float cs = p.rv;
    const int kx = p.periodic ? 1 : max(p.kx, 1);
    int n0 = 1, n1 = 1, n2 = 1;
    for (int it = 0; it < 64; it++) {
        n0 = kx * max((int)floorf(e0 / cs) + extra, 1);
        n1 = max((int)floorf(e1 / cs) + extra, 1);
        n2 = max((int)floorf(e2 / cs) + extra, 1);
        if ((float)n0 * (float)n1 * (float)n2 <= (float)p.ncell_cap) break;
        cs *= 1.26f;
    }
    GridP g;
    g.org[0] = lo0; g.org[1] = lo1; g.org[2] = lo2;
    if (p.periodic) { g.inv[0] = (float)n0 / e0; g.inv[1] = (float)n1 / e1; g.inv[2] = (float)n2 / e2; }
    else { g.inv[0] = (float)kx / cs; g.inv[1] = 1.0f / cs; g.inv[2] = 1.0f / cs; }
    g.nc[0] = n0; g.nc[1] = n1; g.nc[2] = n2;
    g.ncell = n0 * n1 * n2;
    return g;
}
// warm builds: the grid from the box (periodic) or from the bounding box the build before recorded -- every block of k_bin / k_scatter
// computes the same grid from the same six words
__device__ __forceinline__ GridP grid_warm(const BuildParams &p, unsigned r)
{
    if (p.periodic) return grid_on(p, 0.f, 0.f, 0.f, p.box[0], p.box[1], p.box[2]);
    const float *b = p.bbox_cur + r * 6;
    return grid_on(p, b[0], b[1], b[2], b[3] - b[0], b[4] - b[1], b[5] - b[2]);
}

#define GD_GRIDP_THREADS 256
__global__ __launch_bounds__(GD_GRIDP_THREADS) void k_gridp(const BuildParams p)      // cold builds (open boxes): after k_bbox
{
    const unsigned r = blockIdx.x, lane = threadIdx.x & 63u;
    float lo0 = 0.f, lo1 = 0.f, lo2 = 0.f, e0 = p.box[0], e1 = p.box[1], e2 = p.box[2];
    if (!p.periodic) {
        float l0 = INFINITY, l1 = INFINITY, l2 = INFINITY, h0 = -INFINITY, h1 = -INFINITY, h2 = -INFINITY;
        for (unsigned b = lane; b < p.nblk; b += 64) {
            const float *bb = p.bbox + ((size_t)r * p.nblk + b) * 6;
            l0 = fminf(l0, bb[0]); l1 = fminf(l1, bb[1]); l2 = fminf(l2, bb[2]);
            h0 = fmaxf(h0, bb[3]); h1 = fmaxf(h1, bb[4]); h2 = fmaxf(h2, bb[5]);
        }
        for (int o = 32; o > 0; o >>= 1) {
            l0 = fminf(l0, __shfl_xor(l0, o, 64)); l1 = fminf(l1, __shfl_xor(l1, o, 64)); l2 = fminf(l2, __shfl_xor(l2, o, 64));
            h0 = fmaxf(h0, __shfl_xor(h0, o, 64)); h1 = fmaxf(h1, __shfl_xor(h1, o, 64)); h2 = fmaxf(h2, __shfl_xor(h2, o, 64));
        }
        lo0 = l0; lo1 = l1; lo2 = l2;
        e0 = h0 - l0; e1 = h1 - l1; e2 = h2 - l2;
    }
    const GridP g = grid_on(p, lo0, lo1, lo2, e0, e1, e2);
    if (threadIdx.x == 0) {
        p.grid[r] = g;
        p.flags[r * GD_NFLAGS + GD_FLAG_NCELL] = (unsigned)g.ncell;
    }
}

template <bool PERIODIC>
__device__ __forceinline__ void cell_coords(const GridP &g, const float4 x, const float *inv_box, int &cx, int &cy, int &cz)
{
    if (PERIODIC) {
        float fx = x.x * inv_box[0], fy = x.y * inv_box[1], fz = x.z * inv_box[2];
        fx -= floorf(fx); fy -= floorf(fy); fz -= floorf(fz);
        cx = min((int)(fx * g.nc[0]), g.nc[0] - 1);
        cy = min((int)(fy * g.nc[1]), g.nc[1] - 1);
        cz = min((int)(fz * g.nc[2]), g.nc[2] - 1);
    } else {
        cx = min(max((int)floorf((x.x - g.org[0]) * g.inv[0]), 0), g.nc[0] - 1);
        cy = min(max((int)floorf((x.y - g.org[1]) * g.inv[1]), 0), g.nc[1] - 1);
        cz = min(max((int)floorf((x.z - g.org[2]) * g.inv[2]), 0), g.nc[2] - 1);
    }
}

// WARM: every block lays the replica's grid itself (grid_warm) and the replica's first block records it for the kernels that follow;
// cold builds read the grid k_gridp wrote.
template <bool PERIODIC, bool WARM>
__global__ __launch_bounds__(GD_BLOCK) void k_bin(const BuildParams p)
{
    const unsigned r = blockIdx.x / p.nblk, blk = blockIdx.x % p.nblk;
    const unsigned slot = blk * GD_BLOCK + threadIdx.x, lane = threadIdx.x & 63;
    const bool valid = slot < p.N;
    const size_t g = (size_t)r * p.Np + slot;
    unsigned c = 0xffffffffu;
    const GridP gp = WARM ? grid_warm(p, r) : p.grid[r];
    if (WARM && blk == 0 && threadIdx.x == 0) { p.grid[r] = gp; p.flags[r * GD_NFLAGS + GD_FLAG_NCELL] = (unsigned)gp.ncell; }
    if (valid) {
        int cx, cy, cz;
        cell_coords<PERIODIC>(gp, p.pos_in[g], p.inv_box, cx, cy, cz);
        c = (unsigned)((cz * gp.nc[1] + cy) * gp.nc[0] + cx);
    }
    // Slots are still nearly cell-sorted from the previous build, so equal cells sit in runs of
    // consecutive lanes: one atomic per run (its head lane) instead of one per bead.
    const unsigned prev = __shfl_up(c, 1, 64);
    const bool head = lane == 0 || prev != c;
    const unsigned long long hm = __ballot(head);
    const unsigned long long below = hm & ((2ull << lane) - 1ull);          // heads at or below this lane
    const unsigned head_lane = 63u - (unsigned)__clzll(below);
    const unsigned long long above = lane == 63 ? 0ull : (hm >> (lane + 1)) << (lane + 1);
    const unsigned next_head = above ? (unsigned)__ffsll((long long)above) - 1u : 64u;
    unsigned base = 0;
    if (head && valid) base = atomicAdd(&p.cell_cnt[(size_t)r * (p.ncell_cap + 1) + c], next_head - lane);
    base = __shfl(base, head_lane, 64);
    if (valid) p.rank[g] = base + (lane - head_lane);      // (k_scatter finds the cell again from the position: 4 bytes per bead less each way)
}

// exclusive scan of the cell counts of one replica; the counts go through LDS in coalesced tiles so that each thread can scan a
// contiguous run.  One 1024-thread block per (replica, segment of GD_SCAN_TILE cells): a block first sums the counts of the segments
// in front of its own (coalesced reads, L2-resident), then scans its segment from that base -- the grid of the 1 kb model has
// 148 877 cells per replica, which ONE block per replica walked tile after tile in 92 us while 240 CUs idled.
#define GD_SCAN_TILE 8192u
__global__ __launch_bounds__(1024) void k_scan(const BuildParams p)
{
    __shared__ unsigned s_val[GD_SCAN_TILE];
    __shared__ unsigned s_sum[16];
    const unsigned r = blockIdx.x, tid = threadIdx.x;
    const unsigned n = (unsigned)p.grid[r].ncell;
    unsigned t0 = blockIdx.y * GD_SCAN_TILE;
    if (blockIdx.y == 0 && tid == 1023) {      // the per-build words of the replica start over (the last wave: off the path of the scan)
        p.lcount[r] = 0ull; p.lcount[p.R + r] = 0ull;
        if (r == 0 && p.pool) { p.pool[1] = max(p.pool[1], p.pool[0]); p.pool[0] = 0u; p.pool[3] = 0u; }      // the row pool's cursor starts over; [1] keeps the largest use since the host looked (BuildParams)
        p.dmax[r * GD_DMAX_STRIDE] = 0u;        // largest squared displacement since this build (k_step keeps it current)
        if (p.flags[r * GD_NFLAGS + GD_FLAG_OVERFLOW] | p.flags[r * GD_NFLAGS + GD_FLAG_TILE_OVERFLOW]) p.flags[r * GD_NFLAGS + GD_FLAG_TAINT] = 1u;
    }
    if (t0 >= n) return;
    const bool last_block = blockIdx.y + 1 == gridDim.y;      // walks every tile that is left (the launch is sized from the previous build)
    const unsigned *cnt = p.cell_cnt + (size_t)r * (p.ncell_cap + 1);
    unsigned *start = p.cell_start + (size_t)r * (p.ncell_cap + 1);
    unsigned carry = 0;
    // (the counts of the block's own first tile are requested before those in front of it: one memory round trip for both)
    unsigned pre[GD_SCAN_TILE / 1024];
    {
        const unsigned m0 = min(GD_SCAN_TILE, n - t0);
#pragma unroll
        for (unsigned j = 0; j < GD_SCAN_TILE / 1024; j++) pre[j] = tid + j * 1024 < m0 ? cnt[t0 + tid + j * 1024] : 0u;
    }
    bool first_tile = true;
    if (t0 > 0) {      // sum of everything in front of this segment
        unsigned a = 0;
        for (unsigned i = tid; i < t0; i += 1024) a += cnt[i];
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if ((tid & 63u) == 0u) s_sum[tid >> 6] = a;
        __syncthreads();
        for (unsigned w = 0; w < 16; w++) carry += s_sum[w];
        __syncthreads();
    }
    for (;; t0 += GD_SCAN_TILE) {
        const unsigned m = min(GD_SCAN_TILE, n - t0);
        if (first_tile) {
#pragma unroll
            for (unsigned j = 0; j < GD_SCAN_TILE / 1024; j++) if (tid + j * 1024 < m) s_val[tid + j * 1024] = pre[j];
            first_tile = false;
        } else for (unsigned i = tid; i < m; i += 1024) s_val[i] = cnt[t0 + i];
        __syncthreads();
        const unsigned per = GD_SCAN_TILE / 1024, b = tid * per;
        unsigned s = 0;
        for (unsigned i = 0; i < per; i++) if (b + i < m) s += s_val[b + i];
        // inclusive scan of the 1024 per-thread sums: inside each wave by shuffles, across the 16 waves through LDS (two barriers
        // instead of the twenty of a 1024-wide Hillis-Steele scan)
        unsigned incl = s;
        for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_up(incl, o, 64); if ((int)(tid & 63u) >= o) incl += v; }
        if ((tid & 63u) == 63u) s_sum[tid >> 6] = incl;
        __syncthreads();
        unsigned wbase = 0;
        for (unsigned w = 0; w < (tid >> 6); w++) wbase += s_sum[w];
        unsigned total = 0;
        for (unsigned w = 0; w < 16; w++) total += s_sum[w];
        incl += wbase;
        unsigned run = carry + incl - s;
        for (unsigned i = 0; i < per; i++) if (b + i < m) { const unsigned v = s_val[b + i]; s_val[b + i] = run; run += v; }
        __syncthreads();
        for (unsigned i = tid; i < m; i += 1024) start[t0 + i] = s_val[i];
        if (tid == 0 && t0 + m >= n) start[n] = carry + total;      // (the last segment closes the table)
        if (!last_block || t0 + m >= n) break;
        carry += total;
        __syncthreads();
    }
}

// The (up to 9) slot ranges of the LDS tile of one block of slots. A block covers the contiguous
// cell range [c0,c1] (cells are numbered x-fastest); the cells adjacent to any of them lie, for each
// (dz,dy), in the linear range [c0+off-kx, c1+off+kx] with off = (dz*ny + dy)*nx (a superset; kx cells in x are one list radius).
// (One thread per block of slots.  Run by one thread of the k_scatter block (r, blk) instead -- its index arrays live in scratch memory,
// and a kernel that needs scratch runs few waves at a time -- k_scatter took 914 us instead of 60.)
// PERIODIC = false: straight-line code over the nine (dz,dy) ranges -- every array index is a compile-time constant, so the ranges
// live in registers and the 27 reads of the cell table are in flight together (the looped form kept its index arrays in scratch
// memory and read the table range by range: 16 us per build, 10 us of a single replica's).
template <bool PERIODIC>
__global__ __launch_bounds__(64) void k_tiles(const BuildParams p)
{
    const unsigned nb_tiles = (p.R * p.nblk + 63u) / 64u;
    if (blockIdx.x >= nb_tiles) {      // open boxes, one more wave per replica: the bounding box of the sorted positions from k_scatter's partials
        const unsigned r = blockIdx.x - nb_tiles, nw = p.nblk * (GD_BLOCK / 64);
        float l0 = INFINITY, l1 = INFINITY, l2 = INFINITY, h0 = -INFINITY, h1 = -INFINITY, h2 = -INFINITY;
        for (unsigned i = threadIdx.x; i < nw; i += 64) {
            const float *b = p.bbox_w + ((size_t)r * nw + i) * 6;
            l0 = fminf(l0, b[0]); l1 = fminf(l1, b[1]); l2 = fminf(l2, b[2]); h0 = fmaxf(h0, b[3]); h1 = fmaxf(h1, b[4]); h2 = fmaxf(h2, b[5]);
        }
        l0 = wave_minmax<false>(l0); l1 = wave_minmax<false>(l1); l2 = wave_minmax<false>(l2);
        h0 = wave_minmax<true>(h0); h1 = wave_minmax<true>(h1); h2 = wave_minmax<true>(h2);
        if (threadIdx.x < 6) p.bbox_next[r * 6 + threadIdx.x] = threadIdx.x == 0 ? l0 : threadIdx.x == 1 ? l1 : threadIdx.x == 2 ? l2 : threadIdx.x == 3 ? h0 : threadIdx.x == 4 ? h1 : h2;
        return;
    }
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.R * p.nblk) return;
    const unsigned r = t / p.nblk, blk = t % p.nblk;
    const bool taint = p.flags[r * GD_NFLAGS + GD_FLAG_TAINT] != 0u;      // (positions not trustworthy: report no needs)
    const size_t rbase = (size_t)r * p.Np;
    const GridP gp = p.grid[r];
    const unsigned *__restrict__ cs = p.cell_start + (size_t)r * (p.ncell_cap + 1);
    const unsigned first = blk * GD_BLOCK, last = min(first + GD_BLOCK, p.N) - 1;
    // cells of the block's first and last slot: found again from the sorted positions (the sort keeps no cell index per slot)
    int c0, c1;
    {
        const float4 xa = p.pos_out[rbase + first], xz = p.pos_out[rbase + last];
        int ax, ay, az, zx, zy, zz;
        cell_coords<PERIODIC>(gp, xa, p.inv_box, ax, ay, az); cell_coords<PERIODIC>(gp, xz, p.inv_box, zx, zy, zz);
        c0 = (az * gp.nc[1] + ay) * gp.nc[0] + ax; c1 = (zz * gp.nc[1] + zy) * gp.nc[0] + zx;
    }
    TileDesc td;
    if (PERIODIC) {
        // Periodic boxes: the tile is made of WHOLE rows of cells (a row = fixed (z,y), all x; the x-neighbours of the
        // first and last cell of a row are in the same row).  Rows needed = the 3 x 3 (dz,dy) neighbours, wrapped, of
        // every row the block touches; consecutive rows are contiguous in slot order and merge into one range.
        const int nx = gp.nc[0], ny = gp.nc[1], nz = gp.nc[2];
        const int r0 = c0 / nx, r1 = c1 / nx;
        // Runs of consecutive needed rows (a run = one contiguous slot range).  The touched rows r0..r1 are consecutive; inside
        // one z-plane they are the y-interval [ya, yb], whose neighbour rows in each of the planes z-1, z, z+1 (wrapped) are
        // the y-interval [ya-1, yb+1] (wrapped: one or two runs, or the whole plane).  The runs of all touched planes are sorted
        // by their first row and merged.  Any number of touched rows (sparse regions, small boxes); a block that touches more than
        // GD_TILE_PLANES planes goes to the generic path.
        constexpr int GD_TILE_PLANES = 10;
        const int z0 = r0 / ny, z1 = r1 / ny;
        const bool ok = nx >= 3 && ny >= 3 && nz >= 3 && z1 - z0 + 1 <= GD_TILE_PLANES;
        unsigned total = 0, full = 0; int nm = 0; bool truncated = !ok;
        for (int k = 0; k < GD_TILE_RANGES; k++) { td.start[k] = 0; td.len[k] = 0; td.base[k] = 0; td.kstart[k] = 0xffffffffu; td.kbase[k] = 0; }
        if (!ok) {      // grid too small (aliasing neighbours) or the block spans too many planes: generic path
            if (!taint) p.flags[r * GD_NFLAGS + GD_FLAG_TILE_OVERFLOW] = 1u;
            if (!taint) atomicMax(&p.flags[r * GD_NFLAGS + GD_FLAG_NEED_TILE], 1u << 20);
        }
        int ra[6 * GD_TILE_PLANES], rb[6 * GD_TILE_PLANES], nrun = 0;
        for (int z = z0; ok && z <= z1; z++) {
            const int ya = z == z0 ? r0 % ny : 0, yb = z == z1 ? r1 % ny : ny - 1;
            for (int dz = -1; dz <= 1; dz++) {
                const int base = ((z + dz + nz) % nz) * ny;
                if (yb - ya + 3 >= ny) { ra[nrun] = base; rb[nrun++] = base + ny - 1; continue; }
                const int lo = ya - 1, hi = yb + 1;
                if (lo < 0) { ra[nrun] = base; rb[nrun++] = base + hi; ra[nrun] = base + ny - 1; rb[nrun++] = base + ny - 1; }
                else if (hi >= ny) { ra[nrun] = base; rb[nrun++] = base; ra[nrun] = base + lo; rb[nrun++] = base + ny - 1; }
                else { ra[nrun] = base + lo; rb[nrun++] = base + hi; }
            }
        }
        for (int i = 1; i < nrun; i++) {                    // insertion sort by first row (at most 60 runs, typically 3 to 6)
            const int va = ra[i], vb = rb[i];
            int j = i - 1;
            while (j >= 0 && ra[j] > va) { ra[j + 1] = ra[j]; rb[j + 1] = rb[j]; j--; }
            ra[j + 1] = va; rb[j + 1] = vb;
        }
        for (int i = 0; i < nrun; ) {
            const int q = ra[i];
            int e = rb[i];
            i++;
            while (i < nrun && ra[i] <= e + 1) { e = max(e, rb[i]); i++; }      // overlapping and adjacent runs merge
            const unsigned st = cs[q * nx], len = cs[(e + 1) * nx] - st;
            full += len;
            if (truncated) continue;      // (the tile does not fit: the rest is only counted -- the host sizes the next attempt from the whole)
            if (nm >= GD_TILE_RANGES || total + len > p.tile_cap) {
                if (!taint) p.flags[r * GD_NFLAGS + GD_FLAG_TILE_OVERFLOW] = 1u;
                // too many ranges cannot be cured by a larger tile: report a need beyond every capacity
                if (!taint && nm >= GD_TILE_RANGES) atomicMax(&p.flags[r * GD_NFLAGS + GD_FLAG_NEED_TILE], 1u << 20);
                truncated = true;
                continue;
            }
            td.start[nm] = st; td.len[nm] = len; td.base[nm] = total;
            total += len; nm++;
        }
        if (truncated && !taint) atomicMax(&p.flags[r * GD_NFLAGS + GD_FLAG_NEED_TILE], full);
        td.nranges = truncated ? 0u : (unsigned)nm;
        td.total = total;
        td.own_base = 0;
        for (int k = 0; k < nm && !truncated; k++)
            if (first - td.start[k] < td.len[k]) td.own_base = td.base[k] + (first - td.start[k]);
        if (!taint) atomicMax(&p.flags[r * GD_NFLAGS + GD_FLAG_NEED_TILE], total);
        p.tiles[t] = td;
        return;
    }
    // the 9 (dz,dy) cell ranges come out ordered by their first cell; overlapping and adjacent ones merge.  (hi grows with k, so the
    // largest hi so far is the hi of the merged range in progress.)
    constexpr int NR = GD_TILE_RANGES;
    int lo[NR], hi[NR], mid[NR];
    bool val[NR], head[NR];
    int nm = 0, hprev = 0;
    bool any = false;
#pragma unroll
    for (int k = 0; k < NR; k++) {
        const int dz = k / 3 - 1, dy = k % 3 - 1;
        const int off = (dz * gp.nc[1] + dy) * gp.nc[0];
        lo[k] = max(c0 + off - p.kx, 0); hi[k] = min(c1 + off + p.kx, gp.ncell - 1);      // (kx cells in x are one list radius)
        val[k] = lo[k] <= hi[k];
        head[k] = val[k] && (!any || lo[k] > hprev + 1);
        if (head[k]) nm++;
        mid[k] = nm - 1;
        if (val[k]) { hprev = any ? max(hprev, hi[k]) : hi[k]; any = true; }
    }
    int mlo[NR], mhi[NR];
#pragma unroll
    for (int m = 0; m < NR; m++) { mlo[m] = 0; mhi[m] = 0; }
#pragma unroll
    for (int k = 0; k < NR; k++)
#pragma unroll
        for (int m = 0; m <= k; m++)
            if (val[k] && mid[k] == m) { if (head[k]) mlo[m] = lo[k]; mhi[m] = hi[k]; }
    // every read of the cell table, unconditionally (an index of a range that does not exist is 0)
    unsigned st[NR], en[NR], ks[NR];
#pragma unroll
    for (int m = 0; m < NR; m++) { st[m] = cs[m < nm ? mlo[m] : 0]; en[m] = cs[m < nm ? mhi[m] + 1 : 0]; }
#pragma unroll
    for (int k = 0; k < NR; k++) ks[k] = cs[val[k] ? lo[k] : 0];
    unsigned total = 0, full = 0;
    bool truncated = false;
#pragma unroll
    for (int m = 0; m < NR; m++) {
        const unsigned s0 = m < nm ? st[m] : 0u;
        unsigned len = m < nm ? en[m] - st[m] : 0u;
        full += len;
        if (total + len > p.tile_cap) {   // does not fit the LDS budget: flag, host rolls back and re-plans (from the WHOLE tile's size, below)
            if (!taint) p.flags[r * GD_NFLAGS + GD_FLAG_TILE_OVERFLOW] = 1u;
            len = 0; truncated = true;
        }
        td.start[m] = s0; td.len[m] = len; td.base[m] = total;
        total += len;
    }
    unsigned own_base = 0;
#pragma unroll
    for (int k = 0; k < NR; k++) {
        unsigned kstart = 0xffffffffu, kbase = 0;
        if (val[k] && !truncated) {
            unsigned b0 = 0, s0 = 0;
#pragma unroll
            for (int m = 0; m <= k; m++) if (mid[k] == m) { b0 = td.base[m]; s0 = td.start[m]; }
            kstart = ks[k]; kbase = b0 + (ks[k] - s0);
        }
        td.kstart[k] = kstart; td.kbase[k] = kbase;
        if (k < nm && !truncated && first - td.start[k] < td.len[k]) own_base = td.base[k] + (first - td.start[k]);
    }
    td.nranges = truncated ? 0u : (unsigned)nm;
    td.total = total;
    td.own_base = own_base;
    td.pad_[0] = 0; td.pad_[1] = 0;

    if (!taint) atomicMax(&p.flags[r * GD_NFLAGS + GD_FLAG_NEED_TILE], truncated ? full : total);
    p.tiles[t] = td;
}

// The ranks k_bin hands out are arrival orders of its atomics: they differ from run to run, and with them the order of the beads
// inside a cell, the order of every list, the fp32 summation order of every force -- the trajectory (the reference is one thread
// in fp64: one seed, one trajectory; 5-sim-genome/scripts/run_simulation:8-25, simulation_fine_sampling/simulation_driver.cc:44-54
// restarts from a stored frame and relies on it).  k_members lists the bead ids of every cell (at the arrival rank, any order);
// k_scatter then places a bead behind the members of its cell with a smaller id: slot order = (cell, bead id), a function of the
// positions alone.
template <bool PERIODIC>
__global__ __launch_bounds__(GD_BLOCK) void k_members(const BuildParams p)
{
    const unsigned r = blockIdx.x / p.nblk, blk = blockIdx.x % p.nblk;
    const unsigned slot = blk * GD_BLOCK + threadIdx.x;
    if (slot >= p.N) return;
    const size_t rbase = (size_t)r * p.Np, g = rbase + slot;
    const GridP gp = p.grid[r];
    int cx, cy, cz;
    cell_coords<PERIODIC>(gp, p.pos_in[g], p.inv_box, cx, cy, cz);       // as k_bin found it
    const unsigned c = (unsigned)((cz * gp.nc[1] + cy) * gp.nc[0] + cx);
    p.members[rbase + p.cell_start[(size_t)r * (p.ncell_cap + 1) + c] + p.rank[g]] = p.orig_in[g];
}

// Counting-sort scatter into the new slot order (+ the static per-slot data); open boxes: every wave also records the bounding box of
// the positions it moves, for the next build's grid (the kernel is memory-bound: the reductions are free).
template <bool PERIODIC>
__global__ __launch_bounds__(GD_BLOCK) void k_scatter(const BuildParams p)
{
    const unsigned r = blockIdx.x / p.nblk, blk = blockIdx.x % p.nblk;
    const unsigned slot = blk * GD_BLOCK + threadIdx.x, lane = threadIdx.x & 63u;
    const bool valid = slot < p.N;
    const size_t rbase = (size_t)r * p.Np, g = rbase + slot;
    const GridP gp = p.grid[r];
    float4 x = valid ? p.pos_in[g] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (!PERIODIC) {
        const float l0 = wave_minmax<false>(valid ? x.x : INFINITY), l1 = wave_minmax<false>(valid ? x.y : INFINITY), l2 = wave_minmax<false>(valid ? x.z : INFINITY);
        const float h0 = wave_minmax<true>(valid ? x.x : -INFINITY), h1 = wave_minmax<true>(valid ? x.y : -INFINITY), h2 = wave_minmax<true>(valid ? x.z : -INFINITY);
        // (one partial per wave, plain stores: atomics on the replica's six words -- 470 waves each -- made this kernel 5 x slower)
        if (lane < 6) p.bbox_w[((size_t)r * p.nblk * (GD_BLOCK / 64) + blk * (GD_BLOCK / 64) + (threadIdx.x >> 6)) * 6 + lane] =
            lane == 0 ? l0 : lane == 1 ? l1 : lane == 2 ? l2 : lane == 3 ? h0 : lane == 4 ? h1 : h2;
    }
    if (!valid) return;
    int cx, cy, cz;
    cell_coords<PERIODIC>(gp, x, p.inv_box, cx, cy, cz);       // as k_bin found it
    const unsigned c = (unsigned)((cz * gp.nc[1] + cy) * gp.nc[0] + cx);
    const unsigned o = p.orig_in[g];
    // rank inside the cell by bead id (the members of a cell share a cache line or two; the lanes of a wave sit in a handful of cells)
    const unsigned c_lo = p.cell_start[(size_t)r * (p.ncell_cap + 1) + c], c_hi = p.cell_start[(size_t)r * (p.ncell_cap + 1) + c + 1];
    unsigned ns = c_lo;
    {
        const unsigned *__restrict__ mem = p.members + rbase;
        for (unsigned q = c_lo; q < c_hi; q++) ns += mem[q] < o ? 1u : 0u;
    }
    // (a,b) ride in pos.w and every kernel carries w along: only positions that came from the host (w = 0) need the per-bead
    // gather again
    float2 ab = make_float2(0.f, 0.f);
    if (!p.packed_ab || !p.w_valid) ab = p.ab_o[o];
    if (p.packed_ab && !p.w_valid) x.w = pack_ab(ab);
    const size_t gn = rbase + ns;
    p.pos_out[gn] = x;
    if (!p.tiled) p.xb[gn] = x;      // build positions by slot: the generic path's skin check (the tiled path keeps them per thread, rec_x0)
    p.orig_out[gn] = o;
    p.slot_of[(size_t)r * p.N + o] = ns;
    if (!p.packed_ab) p.ab[gn] = ab;
    if (!p.mob_is_uniform) p.mob[gn] = p.mob_o[o];
    if (p.has_bend) p.bendE[gn] = p.bendE_o[o];
}

// Per new slot: re-map the bonded topology to slots and fill the Verlet list (27-cell sweep).
// TILED: candidates are read from the block's LDS tile and list entries are tile indices.
// (register budget of the tiled variants: 6 waves per SIMD = three blocks per CU = at most 80 VGPRs for open boxes, as the LDS admits
// with the 3 312-entry tile class; the periodic sweep holds 18 window bounds: 4 waves per SIMD = two blocks = 128)
// REPAIR (tiled lists; launched behind the build's k_fill with blocks of ONE wave): the rows of a k_step wave in which some list outgrew
// the predicted width are written again at the width they turned out to need -- the block's tile staged again, the 64 beads of
// that wave listed by the 64 lanes, fresh rows from the pool; the queue of such waves is filled by the last wave of every k_fill
// block (below) and is empty in almost every build.
template <bool PERIODIC, bool TILED, bool S16, bool REPAIR = false>
__global__ __launch_bounds__(REPAIR ? 64 : GD_BLOCK, REPAIR ? 1 : (TILED ? (PERIODIC ? 4 : 6) : 1)) void k_fill(const BuildParams p)
{
    static_assert(TILED || !REPAIR, "only the ragged rows of the tiled lists are repaired");
    constexpr unsigned NTHR = REPAIR ? 64u : (unsigned)GD_BLOCK;
    GD_FSTAMP_BEGIN();
    extern __shared__ __attribute__((aligned(16))) float4 s_tile[];
    // block totals (list entries, longest list): accumulated by LDS atomics as the waves finish; the last one to finish hands them on --
    // no barrier at the end of the kernel, a wave that is done leaves.  (Static LDS is budgeted: with the tile class of 3 312 entries
    // three blocks fit a CU only up to 704 bytes of it -- LDS is granted in 1 280-byte granules; tools/kregs.py shows the figure.)
    __shared__ unsigned long long s_acc_cnt;
    __shared__ unsigned s_acc_max, s_acc_done;
    // ragged rows: chunks per lane and first KiB of the rows of each of the block's eight k_step waves
    __shared__ unsigned s_wn[GD_BLOCK / 64], s_woff[GD_BLOCK / 64], s_wneed[GD_BLOCK / 64];      // (s_wneed: what the lists turned out to need)
    unsigned r, blk, rep_w = 0, rep_need = 0;
    if (REPAIR) {
        if (blockIdx.x >= min(p.pool[3], p.rq_cap)) return;      // (queue items; one block = one wave each; the launch has rq_grid blocks)
        const uint2 item = p.rqueue[blockIdx.x];
        r = (item.x >> 3) / p.nblk; blk = (item.x >> 3) % p.nblk; rep_w = item.x & 7u; rep_need = item.y;
    } else if (!block_map(blockIdx.x, p.nblk, p.cpb, r, blk)) return;
    const unsigned lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_acc_cnt = 0ull; s_acc_max = 0u; s_acc_done = 0u; }
    if (threadIdx.x < GD_BLOCK / 64) { s_wn[threadIdx.x] = 0u; s_woff[threadIdx.x] = 0xffffffffu; s_wneed[threadIdx.x] = 0u; }
    if (!TILED) __syncthreads();      // (the tiled path has its barriers below)
    const size_t rbase = (size_t)r * p.Np;
    unsigned slot = blk * GD_BLOCK + threadIdx.x;      // (REPAIR: set below, from the record of the k_step thread this lane stands for)
    size_t gt = rbase + slot;
    const size_t g = rbase + slot;
    const float4 *__restrict__ rpos = p.pos_out + rbase;
    // block-uniform descriptor, read through a uniform pointer (scalar loads; a local copy indexed in
    // loops would be demoted to scratch memory)
    const TileDesc *__restrict__ tdp = p.tiles + (size_t)r * p.nblk + blk;
    // The descriptor is read many times, also after this thread has stored list chunks: from global memory those
    // reads become per-lane vector loads behind a full vmcnt(0) wait (the compiler cannot keep them scalar after a
    // store), so it is copied once into LDS (188 bytes) and read from there.
    __shared__ TileDesc s_tdesc;
#define s_td s_tdesc
    const unsigned o_pre = (TILED && !REPAIR && slot < p.N) ? p.orig_out[g] : 0u;      // (issued ahead of the DMAs: the balancing key below depends on it)
    if (TILED) {
        // LDS-DMA staging as in k_step: descriptor by scalar loads first, then all pieces back to back
        unsigned tlen[GD_TILE_RANGES], tst[GD_TILE_RANGES], tbase[GD_TILE_RANGES];
        const unsigned wq = (unsigned)__builtin_amdgcn_readfirstlane((int)(wid * 64u));
#pragma unroll
        for (int k = 0; k < GD_TILE_RANGES; k++) { tlen[k] = tdp->len[k]; tst[k] = tdp->start[k]; tbase[k] = tdp->base[k]; }
#pragma unroll
        for (int k = 0; k < GD_TILE_RANGES; k++) {
            const unsigned len = tlen[k], st = tst[k], base = tbase[k];
            for (unsigned q0 = wq; q0 < len; q0 += NTHR) {
                if (q0 + lane < len)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rpos + st + q0 + lane),
                                                     (__attribute__((address_space(3))) void *)(s_tile + base + q0), 16, 0, 0);
            }
        }
    }
    // Tiled path: k_step's THREADS are ordered by the list length their bead had at the previous build (a very good
    // predictor of the new one), longest first, so that the 64 lanes of a k_step wave run the same number of list
    // batches.  This kernel still works slot by slot (neighbouring slots share their row windows: broadcast LDS reads, equal
    // trip counts), but writes each bead's list, adjacency chunks and record at the position gt of the k_step thread
    // that will own it.
    if (TILED && REPAIR) {
        if (threadIdx.x < sizeof(TileDesc) / 4) ((unsigned *)&s_tdesc)[threadIdx.x] = ((const unsigned *)tdp)[threadIdx.x];
        __syncthreads();      // (also waits for the tile DMAs)
        if (PERIODIC) {       // the wrapped tile, as below
            const unsigned total = s_tdesc.total;
            for (unsigned t = threadIdx.x; t < total; t += NTHR) {
                float4 x = s_tile[t];
                x.x -= p.box[0] * floorf(x.x * p.inv_box[0]); x.y -= p.box[1] * floorf(x.y * p.inv_box[1]); x.z -= p.box[2] * floorf(x.z * p.inv_box[2]);
                s_tile[t] = x;
            }
            __syncthreads();
        }
        gt = rbase + (size_t)blk * GD_BLOCK + rep_w * 64u + lane;
    }
    if (TILED && !REPAIR) {
        // (two barriers: histogram cleared + descriptor copied | histogram complete; every wave then scans the 64 bins itself.
        // The first barrier also waits for the tile DMAs issued above.)
        // (the order is STABLE -- by bin, then by slot: ranks handed out by an LDS atomic would be arrival orders, and the thread a bead
        // lands on decides the order in which the wall-reaction partials of a block are summed)
        // (one byte per (bin, wave) -- a wave holds at most 64 threads of a bin; lane b reads the eight counts of bin b as one 64-bit
        // word.  GD_KBINS bins: the lists beyond 4 x (GD_KBINS - 2) near entries share the first one.)
        constexpr unsigned GD_KBINS = 48u;
        static_assert(GD_BLOCK / 64 == 8, "one byte per wave in a 64-bit word");
        __shared__ unsigned long long s_hist8[GD_KBINS];
        if (threadIdx.x < GD_KBINS) s_hist8[threadIdx.x] = 0ull;
        if (threadIdx.x >= 64 && threadIdx.x - 64 < sizeof(TileDesc) / 4) ((unsigned *)&s_tdesc)[threadIdx.x - 64] = ((const unsigned *)tdp)[threadIdx.x - 64];
        unsigned bin = GD_KBINS - 1u;                              // slots past N: last
        if (slot < p.N) bin = (GD_KBINS - 2u) - min((unsigned)p.len_prev[(size_t)r * p.N + o_pre], GD_KBINS - 2u);
        // lanes of the wave in the same bin (six ballots), the thread's rank among them, their number
        unsigned long long same = ~0ull;
#pragma unroll
        for (int bit = 0; bit < 6; bit++) {
            const bool on = (bin >> bit) & 1u;
            const unsigned long long b = __builtin_amdgcn_ballot_w64(on);
            same &= on ? b : ~b;
        }
        const unsigned rank_w = (unsigned)__popcll(same & ((1ull << lane) - 1ull));
        __syncthreads();
        if (PERIODIC) {
            // Periodic boxes: wrap the staged tile into [0, L) in place (coordinates are kept unwrapped in memory: a chain that has
            // diffused around the box brings every periodic image into one cell).  With every candidate of a (row, x-interval) in
            // the same image relative to the bead, the minimum image of the sweep below is ONE shift of the bead per interval
            // instead of three round-to-nearest per candidate (15 -> 9 instructions per test; the wrap costs ~6 entries per thread).
            const unsigned total = s_tdesc.total;
            for (unsigned t = threadIdx.x; t < total; t += GD_BLOCK) {
                float4 x = s_tile[t];
                x.x -= p.box[0] * floorf(x.x * p.inv_box[0]); x.y -= p.box[1] * floorf(x.y * p.inv_box[1]); x.z -= p.box[2] * floorf(x.z * p.inv_box[2]);
                s_tile[t] = x;
            }
        }
        if (rank_w == 0u) ((unsigned char *)s_hist8)[bin * 8u + wid] = (unsigned char)__popcll(same);
        __syncthreads();
        // lane b stands for bin b: threads of the block in that bin, and those of them in the waves in front of this one
        unsigned incl = 0, before = 0;
        {
            const unsigned long long v = lane < GD_KBINS ? s_hist8[lane] : 0ull;
            const unsigned long long vb = v & ((1ull << (8u * wid)) - 1ull);      // the waves in front (wid < 8)
            incl = __builtin_amdgcn_sad_u8((unsigned)v, 0u, __builtin_amdgcn_sad_u8((unsigned)(v >> 32), 0u, 0u));
            before = __builtin_amdgcn_sad_u8((unsigned)vb, 0u, __builtin_amdgcn_sad_u8((unsigned)(vb >> 32), 0u, 0u));
        }
        const unsigned own = incl;
        for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += v; }
        gt = rbase + blk * GD_BLOCK + (unsigned)__shfl((int)(incl - own + before), (int)bin, 64) + rank_w;
    }
    // Ragged rows: the k_step wave a bead goes to (wk) gets rows as wide as the longest PREDICTED list among its 64 beads -- what the
    // bead needed at the build before plus an eighth and a chunk per class (no history: the caller's guess, p.W entries); at least one
    // chunk, so that k_step's unconditional first chunk load stays inside the pool.  A list that outgrows its row is REPAIRED at the
    // end of the kernel (no rollback): see the repair pass below.
    const unsigned wk = TILED ? ((unsigned)(gt - rbase) - blk * GD_BLOCK) >> 6 : 0u;
    unsigned row_nc = 0u, row_off = 0u;
    if (TILED && REPAIR) {      // fresh rows of the width the wave's lists need
        unsigned off = 0u;
        if (lane == 0) {
            off = atomicAdd(&p.pool[0], rep_need);
            if (!(off <= p.pool_cap && rep_need <= p.pool_cap - off)) {
                if (!p.flags[r * GD_NFLAGS + GD_FLAG_TAINT]) atomicOr(&p.flags[r * GD_NFLAGS + GD_FLAG_OVERFLOW], 4u);
                off = 0xffffffffu;
            } else p.wtab[(rbase + (size_t)blk * GD_BLOCK) / 64 + rep_w] = make_uint2(off, rep_need);
        }
        off = (unsigned)__builtin_amdgcn_readfirstlane((int)off);
        if (off == 0xffffffffu) return;      // (the pool is full: the rows stay as they are, flagged)
        row_off = off; row_nc = rep_need;
    }
    if (TILED && !REPAIR) {
        unsigned want = 1u;
        if (slot < p.N) {
            if (p.predict) {
                const unsigned q = p.need_prev[(size_t)r * p.N + o_pre];
                unsigned pn = q & 1023u, pf = q >> 10;
                pn += max(1u, pn >> 3); pf += max(1u, pf >> 3);
                want = max(min(pn, GD_TILED_MAX_NEAR / 8u) + min(pf, GD_TILED_MAX_FAR / 8u), 1u);
            } else want = max(p.W / 8u, 1u);
        }
        atomicMax(&s_wn[wk], want);
        // No barrier for this: every wave counts itself in once its lanes' widths are in (the counter the end of the kernel uses for "last
        // wave done": it simply starts from eight there), wave 0 alone waits for the eight, allocates, and publishes (s_wn, then s_woff);
        // the other waves go on into the re-map and pick their rows up where they first need them -- a barrier behind the re-map made
        // all eight waves start their sweeps in lock step and cost the block the latency of the cursor's atomic.
        if (lane == 0) { __threadfence_block(); atomicAdd(&s_acc_done, 1u); }
        if (threadIdx.x < GD_BLOCK / 64) {      // eight lanes of wave 0: prefix of the widths, ONE atomic on the pool's cursor per block
            while (atomicAdd(&s_acc_done, 0u) < GD_BLOCK / 64) __builtin_amdgcn_s_sleep(1);
            __threadfence_block();
            const unsigned nc = atomicMax(&s_wn[threadIdx.x], 0u);
            unsigned incl = nc;
            for (int o = 1; o < GD_BLOCK / 64; o <<= 1) { const unsigned v = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += v; }
            const unsigned total = __shfl(incl, GD_BLOCK / 64 - 1, 64);
            unsigned base = 0;
            if (threadIdx.x == GD_BLOCK / 64 - 1) base = atomicAdd(&p.pool[0], total);
            base = __shfl(base, GD_BLOCK / 64 - 1, 64);
            // (a pool that is full: flagged like a row overflow, the rows of this block get no chunks -- every bead of it then counts
            // as overflowed -- and k_step's first chunk load reads the pool's first KiB)
            const bool fits = base <= p.pool_cap && total <= p.pool_cap - base;
            if (!fits && threadIdx.x == 0 && !p.flags[r * GD_NFLAGS + GD_FLAG_TAINT]) atomicOr(&p.flags[r * GD_NFLAGS + GD_FLAG_OVERFLOW], 4u);
            const unsigned off = fits ? base + incl - nc : 0u;
            p.wtab[(rbase + (size_t)blk * GD_BLOCK) / 64 + threadIdx.x] = make_uint2(off, fits ? nc : 0u);
            s_wn[threadIdx.x] = fits ? nc : 0u;
            __threadfence_block();
            *(volatile unsigned *)&s_woff[threadIdx.x] = off;      // (published: 0xffffffff until here)
        }
    }
    GD_FSTAMP(0);     // staging + barrier
    unsigned cnt = 0, near4 = 0;
    unsigned o = 0, deg = 0;
    const unsigned nr_tile = TILED ? (unsigned)__builtin_amdgcn_readfirstlane((int)s_td.nranges) : 0u;      // (merged ranges in use: three, typically)
    auto to_local = [&](unsigned ps, unsigned &idx) -> bool {   // slot -> tile index
        for (unsigned k = 0; k < nr_tile; k++) {
            const unsigned d = ps - s_td.start[k];
            if (d < s_td.len[k]) { idx = s_td.base[k] + d; return true; }
        }
        return false;
    };
    const size_t gw = TILED ? gt : g;          // where this thread's chunks and records go
    bool on = slot < p.N;
    if (REPAIR) {
        const uint2 mo = p.rec_mo[gt];
        on = mo.y != GD_REC_NOBEAD;
        slot = blk * GD_BLOCK + ((mo.x >> 12) & 0x1ffu); o = mo.y & GD_REC_ID_MASK; deg = mo.x & 0xffu;
    } else if (on) {
        o = p.orig_out[g];
        deg = p.bdeg_o[o];
    }
    if (!REPAIR && on) {
        const unsigned *so = p.slot_of + (size_t)r * p.N;
        uint4 *__restrict__ adjw = (uint4 *)p.badj + (size_t)(gw >> 6) * (p.WB / 4) * 64 + (gw & 63);
        // one 16-byte adjacency chunk per round: its four gathers (entry by bead, then slot by partner) are in flight
        // together and the chunk is written with one store
        for (unsigned k0 = 0; k0 < deg; k0 += 4) {
            unsigned ent[4], ps[4], out[4];
#pragma unroll
            for (int u = 0; u < 4; u++) ent[u] = k0 + u < deg ? p.badj_o[(size_t)(k0 + u) * p.N + o] : 0u;
#pragma unroll
            for (int u = 0; u < 4; u++) ps[u] = k0 + u < deg ? so[ent[u] & GD_ADJ_MASK] : 0u;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                unsigned idx;
                out[u] = ps[u] | (ent[u] & ~GD_ADJ_MASK);
                if (TILED && to_local(ps[u], idx)) out[u] = idx | (ent[u] & ~GD_ADJ_MASK) | GD_ADJ_LOCAL;
                if (k0 + u >= deg) out[u] = 0u;
            }
            adjw[(size_t)(k0 >> 2) * 64] = make_uint4(out[0], out[1], out[2], out[3]);
        }
        if (p.chain_o) {
            const int4 c = p.chain_o[o];
            auto conv = [&](int cb) -> int {
                if (cb < 0) return -1;
                const unsigned ps = so[cb];
                unsigned idx;
                if (TILED && to_local(ps, idx)) return (int)(idx | GD_CHAIN_LOCAL);
                return (int)ps;
            };
            p.chain[g] = make_int4(conv(c.x), conv(c.y), conv(c.z), conv(c.w));
        }
    }
    GD_FSTAMP(1);     // bond / chain re-map
    if (TILED && !REPAIR) {      // the rows of the block's waves have their place once wave 0 has published it (long ago, as a rule)
        while ((row_off = *(volatile unsigned *)&s_woff[wk]) == 0xffffffffu) __builtin_amdgcn_s_sleep(1);
        __threadfence_block();
        row_nc = *(volatile unsigned *)&s_wn[wk];
    }
    if (on) {
        unsigned listlen = 0, nAq = 0, nB = 0;
        if (!(p.nbr || p.nbr16)) {      // (a sort without lists: the counter of the bead's cell still goes back to zero, see below)
            const GridP gp = p.grid[r];
            int cx, cy, cz;
            cell_coords<PERIODIC>(gp, rpos[slot], p.inv_box, cx, cy, cz);
            p.cell_cnt[(size_t)r * (p.ncell_cap + 1) + (unsigned)((cz * gp.nc[1] + cy) * gp.nc[0] + cx)] = 0u;
        }
        if (p.nbr || p.nbr16) {
            const GridP gp = p.grid[r];
            const float4 xi = rpos[slot];
            const unsigned *__restrict__ cs = p.cell_start + (size_t)r * (p.ncell_cap + 1);
            int cx, cy, cz;
            cell_coords<PERIODIC>(gp, xi, p.inv_box, cx, cy, cz);
            // the counters k_bin counted into go back to zero for the next build: every bead clears its own cell's (the lanes of a
            // wave sit in a handful of neighbouring cells: a line or two per store instruction); nothing reads them after k_scan
            p.cell_cnt[(size_t)r * (p.ncell_cap + 1) + (unsigned)((cz * gp.nc[1] + cy) * gp.nc[0] + cx)] = 0u;
            const float rv2 = p.rv * p.rv;
            // Tiled lists are kept in TWO classes by the distance at the build: "near" entries (closer than rn) in the chunks
            // from the front of the bead's row, "far" entries (rn <= d < rv) in the chunks from its back.  A pair that is not
            // in the near class was at least rn apart at the build, so while cutoff + 2 x (largest displacement since the
            // build) <= rn it exerts no force and k_step leaves the far chunks alone -- most steps of an interval.
            const float dnear = rv2 - p.rn * p.rn;           // (r2 - rv2) + dnear = r2 - rn2
            // list writer: entries go straight into the wave-interleaved 16-byte chunk layout k_step
            // reads (8 x u16 tiled, 4 x u32 generic); a bead's 8 (4) consecutive entries share one chunk.
            constexpr unsigned PER = TILED ? 8u : 4u;
            // (tiled: the rows of the thread's k_step wave -- NC chunks per lane from KiB s_woff[wk] of the pool; generic: uniform rows)
            // (tiled: the width is a per-lane value now, and the sweep below sits exactly at its register budget -- 80 VGPRs for three
            // blocks per CU: the width, the k_step wave and the bond degree share one register through the sweep and are unpacked
            // where they are needed, a few times per bead; the empty asm keeps the unpacking from being hoisted back out)
            unsigned pk = deg | (wk << 8) | (row_nc << 11);
            auto NCf = [&]() -> unsigned { if (!TILED) return p.W / PER; unsigned t = pk; asm volatile("" : "+v"(t)); return t >> 11; };
#define NC NCf()
            uint4 *__restrict__ lst = TILED ? (uint4 *)p.nbr16 + ((size_t)row_off * 64 + (gw & 63)) : (uint4 *)p.nbr + (size_t)(gw >> 6) * (p.W / PER) * 64 + (gw & 63);
            // the 16-byte chunk under construction lives in four registers (an LDS staging slot per thread would cost
            // the 8 KB that separate two from three resident blocks per CU); every PER-th entry the finished chunk
            // goes out as one 16-byte global store (2-byte scattered global stores were 25% of the build)
            unsigned w0 = 0, w1 = 0, w2 = 0, w3 = 0;
            // A finished chunk is parked in p0..p3 and stored at the next flush point (the end of a row window), where
            // the lanes of the wave store together: one store instruction per row instead of one per append iteration.
            unsigned p0 = 0, p1 = 0, p2 = 0, p3 = 0, pend = 0;      // pend: chunk index + 1 of the parked chunk, 0 = none
            auto flush = [&]() {
                if (pend) { lst[(size_t)(pend - 1) * 64] = make_uint4(p0, p1, p2, p3); pend = 0; }      // (non-temporal stores here cost 15%: the partial lines no longer merge in L2)
            };
            unsigned cntB = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;      // far class (tiled): its own shift register, chunks stored from the back
            auto push = [&](unsigned j) {
                if (TILED) {        // 128-bit shift register: eight 16-bit entries, the first one ends up lowest
                    w0 = __builtin_amdgcn_alignbit(w1, w0, 16); w1 = __builtin_amdgcn_alignbit(w2, w1, 16);
                    w2 = __builtin_amdgcn_alignbit(w3, w2, 16); w3 = __builtin_amdgcn_alignbit(j, w3, 16);
                } else {
                    w0 = w1; w1 = w2; w2 = w3; w3 = j;
                }
                cnt++;
                if (cnt % PER == 0) {
                    if (cnt / PER <= NC) {      // (inside the row; a list that outgrows its row is flagged below, its chunk rolled back)
                        flush();                               // (only if a second chunk fills before the next flush point)
                        p0 = w0; p1 = w1; p2 = w2; p3 = w3; pend = cnt / PER;
                    }
                }
            };
            auto push_far = [&](unsigned j) {                  // (tiled only; far chunks are few: stored as they fill)
                b0 = __builtin_amdgcn_alignbit(b1, b0, 16); b1 = __builtin_amdgcn_alignbit(b2, b1, 16);
                b2 = __builtin_amdgcn_alignbit(b3, b2, 16); b3 = __builtin_amdgcn_alignbit(j, b3, 16);
                cntB++;
                if (cntB % 8u == 0) { const unsigned nc = NC; if (cntB / 8u <= nc) lst[(size_t)(nc - cntB / 8u) * 64] = make_uint4(b0, b1, b2, b3); }
            };
            if (TILED && PERIODIC) {
                // periodic tile = whole rows: for each of the 9 wrapped (dz,dy) rows the x-window cx-1..cx+1 is one slot interval, or
                // two when it wraps around the row end.  The tile holds wrapped coordinates (above), so the candidates of one
                // (row, interval) are all in the same periodic image relative to the bead: the bead, wrapped the same way, is shifted
                // by that image once and the tests are the open-box ones.  (Exact while cells are at least the list radius wide and
                // the grid has three cells per axis: k_tiles sends smaller grids to the generic path.)
                const int nx = gp.nc[0], ny = gp.nc[1], nz = gp.nc[2];
                const float3 xw = make_float3(xi.x - p.box[0] * floorf(xi.x * p.inv_box[0]), xi.y - p.box[1] * floorf(xi.y * p.inv_box[1]),
                                              xi.z - p.box[2] * floorf(xi.z * p.inv_box[2]));
                unsigned pb[2 * GD_TILE_RANGES], pe[2 * GD_TILE_RANGES];
#pragma unroll
                for (int k = 0; k < GD_TILE_RANGES; k++) {
                    const int zz = (cz + k / 3 - 1 + nz) % nz, yy = (cy + k % 3 - 1 + ny) % ny;
                    const unsigned row = (unsigned)((zz * ny + yy) * nx);
                    if (cx == 0) { pb[2 * k] = cs[row]; pe[2 * k] = cs[row + 2]; pb[2 * k + 1] = cs[row + nx - 1]; pe[2 * k + 1] = cs[row + nx]; }
                    else if (cx == nx - 1) { pb[2 * k] = cs[row]; pe[2 * k] = cs[row + 1]; pb[2 * k + 1] = cs[row + nx - 2]; pe[2 * k + 1] = cs[row + nx]; }
                    else { pb[2 * k] = cs[row + cx - 1]; pe[2 * k] = cs[row + cx + 2]; pb[2 * k + 1] = 0; pe[2 * k + 1] = 0; }
                }
#pragma unroll
                for (int k2 = 0; k2 < 2 * GD_TILE_RANGES; k2++) {
                    const unsigned b = pb[k2], e = pe[k2];
                    if (e > b) {
                        unsigned lb = 0;
                        if (!to_local(b, lb)) { p.flags[r * GD_NFLAGS + GD_FLAG_TILE_OVERFLOW] = 1u; continue; }   // (cannot happen: the row is staged)
                        const unsigned le = lb + (e - b);
                        const unsigned self_l = lb + (slot - b);
                        // image of this interval: rows below / above the box in y and z; in x the first interval of a bead in the last
                        // cell is cell 0 (one period up), the second interval of a bead in cell 0 is the last cell (one period down)
                        const int y0 = cy + (k2 >> 1) % 3 - 1, z0 = cz + (k2 >> 1) / 3 - 1;
                        const float ax = xw.x + ((k2 & 1) ? (cx == 0 ? p.box[0] : 0.f) : (cx == nx - 1 ? -p.box[0] : 0.f));
                        const float ay = xw.y + (y0 < 0 ? p.box[1] : y0 >= ny ? -p.box[1] : 0.f);
                        const float az = xw.z + (z0 < 0 ? p.box[2] : z0 >= nz ? -p.box[2] : 0.f);
                        for (unsigned j0 = lb; j0 < le; j0 += 32) {
                            const unsigned n = min(32u, le - j0);
                            unsigned m = 0, mn = 0;
                            for (unsigned u0 = 0; u0 < n; u0 += 4) {
                                const float4 *cj = s_tile + j0 + u0;
#pragma unroll
                                for (int u = 0; u < 4; u++) {
                                    const float4 xj = cj[u];
                                    const float dx = ax - xj.x, dy = ay - xj.y, dz = az - xj.z;
                                    const float t = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, -rv2)));
                                    m = __builtin_amdgcn_alignbit(m, __float_as_uint(t), 31);
                                    mn = __builtin_amdgcn_alignbit(mn, __float_as_uint(t + dnear), 31);
                                }
                            }
                            m >>= ((n + 3u) & ~3u) - n; mn >>= ((n + 3u) & ~3u) - n;
                            const unsigned sd = self_l - j0;
                            if (sd < n) m &= ~(1u << (n - 1u - sd));
                            mn &= m; m ^= mn;                                 // near class, far class
                            // (lowest set bit first: find-first-bit, then clear it with m & (m - 1) -- three instructions less per entry than
                            // isolating the highest bit; the order of a bead's entries is immaterial)
                            while (mn) {
                                const unsigned bit = (unsigned)__builtin_ctz(mn);
                                mn &= mn - 1u;
                                push(S16 ? (j0 + (n - 1u - bit)) << 4 : j0 + (n - 1u - bit));
                            }
                            while (m) {
                                const unsigned bit = (unsigned)__builtin_ctz(m);
                                m &= m - 1u;
                                push_far(S16 ? (j0 + (n - 1u - bit)) << 4 : j0 + (n - 1u - bit));
                            }
                        }
                    }
                    if (k2 & 1) flush();
                }
            } else if (TILED) {
                const unsigned x_lo = (unsigned)max(cx - p.kx, 0), x_hi = (unsigned)min(cx + p.kx, gp.nc[0] - 1);
                // all 18 row-bound loads are issued before the first sweep (memory-level parallelism)
                unsigned rb[GD_TILE_RANGES], re[GD_TILE_RANGES];
#pragma unroll
                for (int k = 0; k < GD_TILE_RANGES; k++) {
                    const int zz = cz + k / 3 - 1, yy = cy + k % 3 - 1;
                    rb[k] = 0; re[k] = 0;
                    if (zz < 0 || zz >= gp.nc[2] || yy < 0 || yy >= gp.nc[1] || s_td.kstart[k] == 0xffffffffu) continue;
                    const unsigned row = (unsigned)((zz * gp.nc[1] + yy) * gp.nc[0]);
                    rb[k] = cs[row + x_lo]; re[k] = cs[row + x_hi + 1];
                }
                GD_FSTAMP(2);     // row bounds
#pragma unroll
                for (int k = 0; k < GD_TILE_RANGES; k++) {
                    const unsigned b = rb[k], e = re[k];
                    if (e > b) {
                        const unsigned lb = s_td.kbase[k] + (b - s_td.kstart[k]), le = lb + (e - b);
                        const unsigned self_l = lb + (slot - b);       // == own tile index when the bead is in this row segment
                        // branch-free distance tests into a per-lane bit mask, then one append per set bit:
                        // the (divergent) append code runs max-popcount times per segment, not once per candidate
                        for (unsigned j0 = lb; j0 < le; j0 += 32) {
                            const unsigned n = min(32u, le - j0);
                            unsigned m = 0, mn = 0;
                            // nine instructions per candidate: r2 - rv2 by three subtractions and three fmas, its sign bit
                            // shifted into the mask by one v_alignbit (candidate i of the n4 tested ends up at bit
                            // n4-1-i), one add and one more v_alignbit for the near-class mask.  Reads may run up to 3
                            // slots past the window (slack is allocated behind the tile); those bits are shifted out
                            // below, the bead itself is masked once.
                            for (unsigned u0 = 0; u0 < n; u0 += 4) {
                                GD_FCOUNT(8);         // wave-level test groups (lane 0 runs while any lane does)
                                const float4 *cj = s_tile + j0 + u0;
#pragma unroll
                                for (int u = 0; u < 4; u++) {
                                    const float4 xj = cj[u];
                                    const float dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
                                    const float t = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, -rv2)));      // < 0 inside the list radius
                                    m = __builtin_amdgcn_alignbit(m, __float_as_uint(t), 31);              // m = m << 1 | sign(t)
                                    mn = __builtin_amdgcn_alignbit(mn, __float_as_uint(t + dnear), 31);    // < 0 inside the near radius
                                }
                            }
                            m >>= ((n + 3u) & ~3u) - n; mn >>= ((n + 3u) & ~3u) - n;          // candidate i now at bit n-1-i
                            const unsigned sd = self_l - j0;
                            if (sd < n) m &= ~(1u << (n - 1u - sd));
                            mn &= m; m ^= mn;                                // near class, far class
                            GD_FSTAMP(3);     // distance tests
                            // (lowest set bit first: find-first-bit, then clear it with m & (m - 1) -- three instructions less per entry than
                            // isolating the highest bit; the order of a bead's entries is immaterial)
                            while (mn) {
                                GD_FCOUNT(9);
                                const unsigned bit = (unsigned)__builtin_ctz(mn);
                                mn &= mn - 1u;
                                push(S16 ? (j0 + (n - 1u - bit)) << 4 : j0 + (n - 1u - bit));
                            }
                            while (m) {
                                const unsigned bit = (unsigned)__builtin_ctz(m);
                                m &= m - 1u;
                                push_far(S16 ? (j0 + (n - 1u - bit)) << 4 : j0 + (n - 1u - bit));
                            }
                            GD_FSTAMP(4);     // appends
                        }
                    }
                    flush();
                }
            } else {
                // distinct neighbour cells per dimension (small periodic grids alias)
                const int nz = PERIODIC ? min(3, gp.nc[2]) : 3, ny = PERIODIC ? min(3, gp.nc[1]) : 3, nx = PERIODIC ? min(3, gp.nc[0]) : 2 * p.kx + 1;
                const int z0 = (PERIODIC && gp.nc[2] < 3) ? 0 : cz - 1, y0 = (PERIODIC && gp.nc[1] < 3) ? 0 : cy - 1,
                          x0 = PERIODIC ? (gp.nc[0] < 3 ? 0 : cx - 1) : cx - p.kx;
                for (int iz = 0; iz < nz; iz++) {
                    int zz = z0 + iz;
                    if (PERIODIC) zz = (zz + gp.nc[2]) % gp.nc[2]; else if (zz < 0 || zz >= gp.nc[2]) continue;
                    for (int iy = 0; iy < ny; iy++) {
                        int yy = y0 + iy;
                        if (PERIODIC) yy = (yy + gp.nc[1]) % gp.nc[1]; else if (yy < 0 || yy >= gp.nc[1]) continue;
                        for (int ix = 0; ix < nx; ix++) {
                            int xx = x0 + ix;
                            if (PERIODIC) xx = (xx + gp.nc[0]) % gp.nc[0]; else if (xx < 0 || xx >= gp.nc[0]) continue;
                            const unsigned c = (unsigned)((zz * gp.nc[1] + yy) * gp.nc[0] + xx);
                            const unsigned b = cs[c], e = cs[c + 1];
                            for (unsigned j = b; j < e; j++) {
                                if (j == slot) continue;
                                const float4 xj = rpos[j];
                                float3 d = make_float3(xi.x - xj.x, xi.y - xj.y, xi.z - xj.z);
                                if (PERIODIC) d = min_image(d, p.box, p.inv_box);
                                if (d.x * d.x + d.y * d.y + d.z * d.z < rv2) push(j);
                            }
                        }
                    }
                }
            }
            const unsigned found = cnt + cntB;
            // pad both classes to whole chunks with the bead itself: zero displacement, zero force
            unsigned self = slot;
            if (TILED) { unsigned idx = 0; if (to_local(slot, idx)) self = S16 ? idx << 4 : idx; }
            const unsigned needA = (cnt + GD_UNROLL - 1u) & ~(GD_UNROLL - 1u), needB = (cntB + GD_UNROLL - 1u) & ~(GD_UNROLL - 1u);
            const unsigned needw = needA + needB;
            // the tiled record counts the near entries in fours (11 bits: 8 184 entries) and the far chunks in 6 bits (504 entries): a
            // class beyond that does not fit it even when the row is wide enough for the sum -- flagged like a row overflow, bit 1 on
            // top (the host then builds single-class lists, or generic ones beyond 8 184 entries, until the dense transient has passed)
            const bool class_over = TILED && (needA > GD_TILED_MAX_NEAR || needB > GD_TILED_MAX_FAR);
            // (tiled lists: a row that is too narrow is repaired below, only a class beyond its field is flagged; generic lists: the
            // host widens the uniform rows and builds again)
            const unsigned nc_row = NC, Wrow = nc_row * PER;
            if (((!TILED && needw > Wrow) || class_over) && !p.flags[r * GD_NFLAGS + GD_FLAG_TAINT]) {
                atomicOr(&p.flags[r * GD_NFLAGS + GD_FLAG_OVERFLOW], class_over ? 3u : 1u);
                atomicMax(&p.flags[r * GD_NFLAGS + GD_FLAG_NEED_W], needw);
            }
            if (TILED) {
                // what this bead needed: for the rows of the next build, and for the repair pass -- the k_step wave it goes to needs
                // the sum for its longest list
                const unsigned na8 = min(needA / 8u, GD_TILED_MAX_NEAR / 8u), nb8 = min(needB / 8u, GD_TILED_MAX_FAR / 8u);
                p.need_prev[(size_t)r * p.N + o] = (unsigned short)(na8 | (nb8 << 10));
                if (!REPAIR) atomicMax(&s_wneed[(pk >> 8) & 7u], na8 + nb8);
            }
            listlen = min(found, Wrow);
            nAq = min((cnt + 3u) / 4u, GD_TILED_MAX_NEAR / 4u);            // near entries in fours (the record's count; chunks are still written whole)
            // pad the chunk under construction with the bead itself: the shift register moves down by the missing entries in three
            // branch-free stages (four, two, one entry -- a loop of single pushes ran seven times in nearly every wave, 16 instructions
            // each) and goes out with one store
            flush();
            auto pad = [&](unsigned &a0, unsigned &a1, unsigned &a2, unsigned &a3, unsigned sh) {      // sh entries (< PER) of `self` in from the top
                const unsigned S = TILED ? self | (self << 16) : self;
                if (TILED) {
                    const bool s4 = (sh & 4u) != 0u, s2 = (sh & 2u) != 0u, s1 = (sh & 1u) != 0u;
                    a0 = s4 ? a2 : a0; a1 = s4 ? a3 : a1; a2 = s4 ? S : a2; a3 = s4 ? S : a3;
                    a0 = s2 ? a1 : a0; a1 = s2 ? a2 : a1; a2 = s2 ? a3 : a2; a3 = s2 ? S : a3;
                    const unsigned c0 = __builtin_amdgcn_alignbit(a1, a0, 16), c1 = __builtin_amdgcn_alignbit(a2, a1, 16),
                                   c2 = __builtin_amdgcn_alignbit(a3, a2, 16), c3 = __builtin_amdgcn_alignbit(S, a3, 16);
                    a0 = s1 ? c0 : a0; a1 = s1 ? c1 : a1; a2 = s1 ? c2 : a2; a3 = s1 ? c3 : a3;
                } else {
                    const bool s2 = (sh & 2u) != 0u, s1 = (sh & 1u) != 0u;
                    a0 = s2 ? a2 : a0; a1 = s2 ? a3 : a1; a2 = s2 ? S : a2; a3 = s2 ? S : a3;
                    a0 = s1 ? a1 : a0; a1 = s1 ? a2 : a1; a2 = s1 ? a3 : a2; a3 = s1 ? S : a3;
                }
            };
            if (cnt % PER) {
                pad(w0, w1, w2, w3, PER - cnt % PER);
                cnt = (cnt + PER - 1u) & ~(PER - 1u);
                if (cnt / PER <= nc_row) lst[(size_t)(cnt / PER - 1u) * 64] = make_uint4(w0, w1, w2, w3);
            }
            if (!TILED && cnt % GD_UNROLL) {      // (generic lists are walked in batches of two chunks: a whole chunk of the bead itself behind an odd one)
                cnt += PER;
                if (cnt / PER <= nc_row) lst[(size_t)(cnt / PER - 1u) * 64] = make_uint4(self, self, self, self);
            }
            if (TILED && cntB % 8u) {
                pad(b0, b1, b2, b3, 8u - cntB % 8u);
                cntB = (cntB + 7u) & ~7u;
                if (cntB / 8u <= nc_row) lst[(size_t)(nc_row - cntB / 8u) * 64] = make_uint4(b0, b1, b2, b3);
            }
            // (an overflowed list: the chunk counts have to stay inside the row until it is repaired)
            nAq = min(nAq, 2u * nc_row); nB = min(min(cntB / GD_UNROLL, GD_TILED_MAX_FAR / GD_UNROLL), nc_row - (nAq + 1u) / 2u);
            cnt = found; near4 = 4u * nAq;
            deg = pk & 0xffu;
#undef NC
        }
        const unsigned meta = deg | ((unsigned)p.psmask_o[o] << 8) | (listlen << 16);
        if (TILED) {
            const float4 xb = rpos[slot];
            p.rec_x0[gt] = make_float4(xb.x, xb.y, xb.z, __uint_as_float(slot - blk * GD_BLOCK));
            // tiled record: bond degree | point-source mask << 8 | block-local slot << 12 | near entries / 4 << 21, bead id | far chunks << 26
            // (y all ones: no bead)
            p.rec_mo[gt] = make_uint2(deg | (((unsigned)p.psmask_o[o] & 0xfu) << 8) | ((slot - blk * GD_BLOCK) << 12) | (nAq << 21), o | (nB << 26));
            p.len_prev[(size_t)r * p.N + o] = (unsigned char)min(nAq, 255u);       // (the near class is what most steps run over)
        } else p.meta[g] = meta;
    }
    if (REPAIR) return;
    if (TILED && !on) { p.rec_x0[gt] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0xffffu)); p.rec_mo[gt] = make_uint2(0u, GD_REC_NOBEAD); }
    GD_FSTAMP(5);     // padding, meta
    // (per block both counts fit 32 bits: the entries in the low word, the near entries -- in the fours k_step walks -- in the high one)
    unsigned long long c64 = (unsigned long long)cnt | ((unsigned long long)near4 << 32);
    unsigned cmax = cnt;
    for (int o2 = 32; o2 > 0; o2 >>= 1) { c64 += __shfl_xor(c64, o2, 64); cmax = max(cmax, (unsigned)__shfl_xor((int)cmax, o2, 64)); }
    if (lane == 0) {
        atomicAdd(&s_acc_cnt, c64); atomicMax(&s_acc_max, cmax);
        __threadfence_block();       // (this wave's needs are in LDS before it counts as done)
        if (atomicAdd(&s_acc_done, 1u) == (TILED ? 2u : 1u) * (GD_BLOCK / 64) - 1u) {      // the last wave of the block to finish (tiled: the counter starts from the eight arrivals above)
            __threadfence_block();
            const unsigned long long t = atomicAdd(&s_acc_cnt, 0ull);
            const unsigned m = atomicMax(&s_acc_max, 0u);
            if (t) { atomicAdd(&p.lcount[r], t & 0xffffffffull); if (TILED) atomicAdd(&p.lcount[p.R + r], t >> 32); }
            if (m && !p.flags[r * GD_NFLAGS + GD_FLAG_TAINT]) atomicMax(&p.flags[r * GD_NFLAGS + GD_FLAG_NEED_W], m);      // longest list, always
            // k_step waves of this block whose lists outgrew their rows: queued for the repair kernel (a block without rows -- the
            // pool was full -- is not repaired: flagged above, the host enlarges the pool)
            if (TILED) {
                for (unsigned w = 0; w < GD_BLOCK / 64; w++) {
                    const unsigned need = atomicMax(&s_wneed[w], 0u), have = s_wn[w];
                    if (have != 0u && need > have) {
                        const unsigned at = atomicAdd(&p.pool[3], 1u);
                        if (at < p.rq_cap) p.rqueue[at] = make_uint2(((r * p.nblk + blk) << 3) | w, need);      // (the queue holds every wave of the handle)
                        // more waves than the repair launch has blocks (a state that changes faster than a build predicts): flagged, the
                        // chunk is rolled back and the host launches the repair kernel with a block for every wave until that has passed
                        if (at >= p.rq_grid && !p.flags[r * GD_NFLAGS + GD_FLAG_TAINT]) atomicOr(&p.flags[r * GD_NFLAGS + GD_FLAG_OVERFLOW], 4u);
                        atomicAdd(&p.pool[2], 1u);      // (repaired waves: diagnostics)
                    }
                }
            }
        }
    }
    GD_FSTAMP(6);     // count
    GD_FSTAMP_END(p.dbg);
}

#undef s_td

void gd_launch_build(const BuildParams &p, hipStream_t st)
{
    const dim3 grid(p.R * p.nblk), block(GD_BLOCK), gridx(p.cpb ? GD_XCDS * p.cpb * p.R : p.R * p.nblk);
    // six launches: count + rank (the grid laid by every block itself) | scan | cell members | scatter (+ the box of the next build's
    // grid) | tile descriptors | fill (+ the cell counters back to zero).  An open box without a bounding box from the build before
    // (the first build of a handle, positions set by the caller): k_bbox and k_gridp in front.
    if (p.periodic) hipLaunchKernelGGL((k_bin<true, true>), grid, block, 0, st, p);
    else if (p.warm) hipLaunchKernelGGL((k_bin<false, true>), grid, block, 0, st, p);
    else {
        hipLaunchKernelGGL(k_bbox, dim3(p.R * ((p.nblk + 3u) / 4u)), block, 0, st, p);
        hipLaunchKernelGGL(k_gridp, dim3(p.R), dim3(GD_GRIDP_THREADS), 0, st, p);
        hipLaunchKernelGGL((k_bin<false, false>), grid, block, 0, st, p);
    }
    hipLaunchKernelGGL(k_scan, dim3(p.R, std::max(1u, p.scan_segments)), dim3(1024), 0, st, p);
    if (p.periodic) { hipLaunchKernelGGL(k_members<true>, grid, block, 0, st, p); hipLaunchKernelGGL(k_scatter<true>, grid, block, 0, st, p); }
    else { hipLaunchKernelGGL(k_members<false>, grid, block, 0, st, p); hipLaunchKernelGGL(k_scatter<false>, grid, block, 0, st, p); }
    if (p.tiled && p.periodic) hipLaunchKernelGGL(k_tiles<true>, dim3((p.R * p.nblk + 63) / 64), dim3(64), 0, st, p);
    else if (p.tiled) hipLaunchKernelGGL(k_tiles<false>, dim3((p.R * p.nblk + 63) / 64 + p.R), dim3(64), 0, st, p);
    if (p.tiled) {
        const size_t lds = (size_t)(p.tile_cap + 4) * sizeof(float4);   // +4: read slack
        // (behind it the repair kernel: one wave per queued k_step wave; its blocks leave at once while the queue is empty)
        const dim3 gridr(p.rq_grid), blockr(64);
        if (p.periodic) {
            if (p.tile_cap < 4096u) { hipLaunchKernelGGL((k_fill<true, true, true>), gridx, block, lds, st, p); hipLaunchKernelGGL((k_fill<true, true, true, true>), gridr, blockr, lds, st, p); }
            else { hipLaunchKernelGGL((k_fill<true, true, false>), gridx, block, lds, st, p); hipLaunchKernelGGL((k_fill<true, true, false, true>), gridr, blockr, lds, st, p); }
        } else if (p.tile_cap < 4096u) {      // byte-offset entries, as k_step expects
            hipLaunchKernelGGL((k_fill<false, true, true>), gridx, block, lds, st, p); hipLaunchKernelGGL((k_fill<false, true, true, true>), gridr, blockr, lds, st, p);
        } else { hipLaunchKernelGGL((k_fill<false, true, false>), gridx, block, lds, st, p); hipLaunchKernelGGL((k_fill<false, true, false, true>), gridr, blockr, lds, st, p); }
    } else if (p.periodic) hipLaunchKernelGGL((k_fill<true, false, false>), gridx, block, 0, st, p);
    else hipLaunchKernelGGL((k_fill<false, false, false>), gridx, block, 0, st, p);
}

// ------------------------------------------------------------- droplet attraction
// U = -eps / (1 + (r/decay)^6), r < cutoff, between all pairs of the (few hundred) target beads of a replica
// (simulation_driver_forcefield.cc:153-178; potential form: documented choice, include/gdyn.h).  A separate small
// kernel after k_step: the update is linear in the force, so x_out += mu dt F_droplet on top of k_step's result is
// the same step; the forces are evaluated on the same (old) positions.
template <int MODE>
__global__ __launch_bounds__(256) void k_softwell(const SoftwellP p)
{
    extern __shared__ __attribute__((aligned(16))) float4 s_x[];
    const unsigned r = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    const unsigned *__restrict__ so = p.slot_of + (size_t)r * p.N;
    const float4 *__restrict__ rpos = p.pos_in + (size_t)r * p.Np;
    for (unsigned q = threadIdx.x; q < p.M; q += 256) s_x[q] = rpos[so[p.targets[q]]];
    __syncthreads();
    float3 F = make_float3(0.f, 0.f, 0.f);
    float E = 0.f;
    unsigned bead = 0;
    if (t < p.M) {
        bead = p.targets[t];
        const float4 xi = s_x[t];
        for (unsigned q = 0; q < p.M; q++) {
            const float4 xj = s_x[q];
            float3 d = make_float3(xi.x - xj.x, xi.y - xj.y, xi.z - xj.z);
            if (p.periodic) d = min_image(d, p.box, p.inv_box);
            const float r2 = d.x * d.x + d.y * d.y + d.z * d.z;
            if (q != t && p.targets[q] != bead && r2 < p.rc2) {
                const float u2 = r2 * p.inv_d2, u6 = u2 * u2 * u2, inv_den = 1.0f / (1.0f + u6);
                const float fr = -6.0f * p.eps * u2 * u2 * p.inv_d2 * inv_den * inv_den;
                F.x += fr * d.x; F.y += fr * d.y; F.z += fr * d.z;
                E += -0.5f * p.eps * inv_den;        // every pair is visited from both ends
            }
        }
    }
    if (MODE == 0) {
        if (t < p.M) {
            const float mu_dt = (p.mob_o ? p.mob_o[bead] : p.mob_uniform) * p.dt;
            float4 *o = p.pos_out + (size_t)r * p.Np + so[bead];
            float4 x = *o;
            const float ex = mu_dt * F.x, ey = mu_dt * F.y, ez = mu_dt * F.z;
            if (p.comp) {
                // compensated runs (T = 0, dt = 1e-7: the droplet's share of mu F dt is around or below an ulp of the coordinate):
                // the same two-sum over (x, lo) as k_step's update, on the pair k_step has just written
                float4 *lp = p.lo + ((size_t)r * p.N + bead);
                const float4 l = *lp;
                const float tx = l.x + ex, ty = l.y + ey, tz = l.z + ez;
                const float nx = x.x + tx, ny = x.y + ty, nz = x.z + tz;
                const float bx = nx - x.x, by = ny - x.y, bz = nz - x.z;
                *lp = make_float4((x.x - (nx - bx)) + (tx - bx), (x.y - (ny - by)) + (ty - by), (x.z - (nz - bz)) + (tz - bz), 0.f);
                x.x = nx; x.y = ny; x.z = nz;
            } else { x.x += ex; x.y += ey; x.z += ez; }
            *o = x;
        }
    } else if (MODE == 1) {
        if (t < p.M) {
            float4 *o = p.fout + (size_t)r * p.N + bead;
            float4 f = *o;
            f.x += F.x; f.y += F.y; f.z += F.z;
            *o = f;
        }
    } else {
        double e = wave_sum_d((double)E);
        if ((threadIdx.x & 63) == 0 && e != 0.0) atomicAdd(&p.esum[r], e);
    }
}

void gd_launch_softwell(const SoftwellP &p, int mode, hipStream_t st)
{
    const dim3 grid((p.M + 255) / 256, p.R), block(256);
    const size_t lds = (size_t)p.M * sizeof(float4);
    if (mode == 0) hipLaunchKernelGGL(k_softwell<0>, grid, block, lds, st, p);
    else if (mode == 1) hipLaunchKernelGGL(k_softwell<1>, grid, block, lds, st, p);
    else hipLaunchKernelGGL(k_softwell<2>, grid, block, lds, st, p);
}

// ------------------------------------------------------------- pair search
// md::neighbor_searcher<Box>{box, dcut}.search(out) (simulation_interphase/contact_map.cc:64-66, glues/glue_simulator.cpp:41,67-77)
// served from the RESIDENT Verlet list: every pair closer than dcut now was closer than dcut + 2 D at the build (D = largest
// displacement since), so for dcut + 2 D <= list radius the list holds them all -- no rebuild, no list download.  One thread per
// list owner: entries within dcut whose partner has the larger bead id are appended to the output (wave-aggregated atomic).
template <bool TILED>
__global__ __launch_bounds__(GD_BLOCK) void k_pairs(const PairsP p)
{
    __shared__ TileDesc s_td;
    const unsigned blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, ry = blockIdx.y, rr = p.r + ry;
    const size_t rbase = (size_t)rr * p.Np, gt = rbase + blk * GD_BLOCK + tid;
    unsigned long long *__restrict__ count = p.count + 2u * ry;
    uint2 *__restrict__ out = p.out + (size_t)ry * p.cap;
    if (p.dmax && __uint_as_float(p.dmax[(size_t)rr * GD_DMAX_STRIDE]) > p.lim2) {      // (block-uniform: the whole replica leaves at once)
        if (tid == 0) count[1] = 1ull;
        return;
    }
    if (TILED) {
        const unsigned *src = (const unsigned *)(p.tiles + (size_t)rr * p.nblk + blk);
        if (tid < sizeof(TileDesc) / 4) ((unsigned *)&s_td)[tid] = src[tid];
        __syncthreads();
    }
    unsigned slot = blk * GD_BLOCK + tid, cnt = 0, nA = 0;      // tiled: cnt = entries of the row, padding included
    bool valid = slot < p.N;
    if (TILED) {
        const uint2 mo = p.rec_mo[gt];
        const unsigned local = (mo.x >> 12) & 0x1ffu;
        nA = ((mo.x >> 21) + 1u) >> 1;
        valid = mo.y != GD_REC_NOBEAD; slot = blk * GD_BLOCK + (valid ? local : 0u); cnt = (nA + (mo.y >> 26)) * 8u;
    } else if (valid) cnt = p.meta[rbase + slot] >> 16;
    const float4 *__restrict__ rpos = p.pos + rbase;
    float4 xi = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned oi = 0;
    if (valid) {
        xi = rpos[slot]; oi = p.orig[rbase + slot];
        const float4 x0 = TILED ? p.x0[gt] : p.x0[rbase + slot];
        const float dx = xi.x - x0.x, dy = xi.y - x0.y, dz = xi.z - x0.z;
        if (dx * dx + dy * dy + dz * dz > p.lim2) count[1] = 1ull;
    } else cnt = 0;
    const size_t gl = TILED ? gt : rbase + slot;
    const uint2 wrow = TILED ? p.wtab[gl >> 6] : make_uint2(0u, 0u);      // (tiled lists: the ragged rows of the thread's wave)
    const unsigned PER = TILED ? 8u : 4u, NC = TILED ? wrow.y : p.W / PER;
    const uint4 *__restrict__ lst = TILED ? (const uint4 *)p.nbr16 + ((size_t)wrow.x * 64 + (gl & 63)) : (const uint4 *)p.nbr + (size_t)(gl >> 6) * NC * 64 + (gl & 63);
    auto partner = [&](unsigned k) -> unsigned {       // slot of list entry k (tiled: near chunks from the front, far chunks from the back)
        const unsigned c = k / PER;
        const uint4 q = lst[(size_t)(TILED && c >= nA ? NC - 1u - (c - nA) : c) * 64];
        const unsigned w = k % PER;
        if (!TILED) return w == 0 ? q.x : w == 1 ? q.y : w == 2 ? q.z : q.w;
        const unsigned word = (w >> 1) == 0 ? q.x : (w >> 1) == 1 ? q.y : (w >> 1) == 2 ? q.z : q.w;
        unsigned idx = (w & 1u) ? word >> 16 : word & 0xffffu;
        if (p.s16) idx >>= 4;
        for (int k2 = 0; k2 < GD_TILE_RANGES; k2++) { const unsigned d = idx - s_td.base[k2]; if (d < s_td.len[k2]) return s_td.start[k2] + d; }
        return slot;      // (cannot happen for a complete tile; the bead itself is no pair)
    };
    auto close_pair = [&](unsigned js, unsigned &oj) -> bool {
        if (js == slot) return false;
        const float4 xj = rpos[js];
        float3 d = make_float3(xi.x - xj.x, xi.y - xj.y, xi.z - xj.z);
        if (p.periodic) d = min_image(d, p.box, p.inv_box);
        if (!(d.x * d.x + d.y * d.y + d.z * d.z < p.dcut2)) return false;
        oj = p.orig[rbase + js];
        return oi < oj;
    };
    // first pass: count, and remember the hits among the first 64 entries (most lists) so that the second pass only decodes those
    unsigned n = 0, oj;
    unsigned long long hits = 0;
    for (unsigned k = 0; k < cnt; k++) {
        const bool c = close_pair(partner(k), oj);
        n += c ? 1u : 0u;
        if (c && k < 64u) hits |= 1ull << k;
    }
    // wave-aggregated append: exclusive scan of the lane counts, one atomic per wave
    unsigned incl = n;
    for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += v; }
    const unsigned total = __shfl(incl, 63, 64);
    unsigned long long base = 0;
    if (lane == 63 && total) base = atomicAdd(count, (unsigned long long)total);
    base = __shfl(base, 63, 64);
    unsigned long long at = base + (incl - n);
    if (n && at + n <= p.cap) {
        while (hits) {
            const unsigned k = (unsigned)__ffsll((long long)hits) - 1u;
            hits &= hits - 1ull;
            out[at++] = make_uint2(oi, p.orig[rbase + partner(k)]);
        }
        for (unsigned k = 64u; k < cnt; k++)
            if (close_pair(partner(k), oj)) out[at++] = make_uint2(oi, oj);
    }
}

void gd_launch_pairs(const PairsP &p, hipStream_t st)
{
    const dim3 grid(p.nblk, p.nrep ? p.nrep : 1u);
    if (p.tiled) hipLaunchKernelGGL(k_pairs<true>, grid, dim3(GD_BLOCK), 0, st, p);
    else hipLaunchKernelGGL(k_pairs<false>, grid, dim3(GD_BLOCK), 0, st, p);
}

// ------------------------------------------------------------- contact maps
// contact_map::update (simulation_interphase/contact_map.cc:31-74) adds a 0/1 matrix of the pairs in contact to a sparse count
// matrix; here the count matrix of a replica is an open-addressing table in HBM (linear probing, at most half full: the host
// grows it ahead of an update), so that neither the pairs nor the counts leave the device between two dumps.
__device__ __forceinline__ unsigned long long ct_hash(unsigned long long k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}

__device__ __forceinline__ void ct_add(unsigned long long *words, unsigned *distinct, unsigned long long mask, unsigned cbits,
                                       unsigned long long key, unsigned long long add)
{
    unsigned long long h = ct_hash(key) & mask;
    for (unsigned long long probe = 0; probe <= mask; probe++, h = (h + 1ull) & mask) {      // (terminates: the table is never full)
        unsigned long long w = words[h];
        if (w == GD_CT_EMPTY) {
            w = atomicCAS(&words[h], GD_CT_EMPTY, (key << cbits) | add);
            if (w == GD_CT_EMPTY) { atomicAdd(distinct, 1u); return; }
        }
        if ((w >> cbits) == key) { atomicAdd(&words[h], add); return; }
    }
}

__global__ __launch_bounds__(256) void k_ct_insert(const ContactTab t, const uint2 *__restrict__ pairs, unsigned long long cap_pairs,
                                                   const unsigned long long *__restrict__ count)
{
    const unsigned r = blockIdx.y;
    const unsigned long long n = min(count[2u * r], cap_pairs);
    unsigned long long *words = t.words + (size_t)r * t.cap;
    for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (unsigned long long)gridDim.x * blockDim.x) {
        const uint2 q = pairs[(size_t)r * cap_pairs + k];
        ct_add(words, t.distinct + r, t.cap - 1ull, 64u - 2u * t.jbits, ((unsigned long long)q.x << t.jbits) | q.y, 1ull);
    }
}

__global__ __launch_bounds__(256) void k_ct_rehash(const ContactTab from, const ContactTab to)
{
    const unsigned r = blockIdx.y, cbits = 64u - 2u * from.jbits;
    for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < from.cap; k += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long w = from.words[(size_t)r * from.cap + k];
        if (w != GD_CT_EMPTY) ct_add(to.words + (size_t)r * to.cap, to.distinct + r, to.cap - 1ull, cbits, w >> cbits, w & ((1ull << cbits) - 1ull));
    }
}

__global__ __launch_bounds__(256) void k_ct_compact(const ContactTab t, unsigned r, unsigned long long *__restrict__ keys_out,
                                                    unsigned *__restrict__ vals_out, unsigned *n_out)
{
    const unsigned lane = threadIdx.x & 63, cbits = 64u - 2u * t.jbits;
    const unsigned long long nround = (t.cap + 255ull) & ~255ull;      // whole waves take part in the ballot
    for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < nround; k += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long w = k < t.cap ? t.words[(size_t)r * t.cap + k] : GD_CT_EMPTY;
        const bool used = w != GD_CT_EMPTY;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(used);
        unsigned base = 0;
        if (lane == 0 && m) base = atomicAdd(n_out, (unsigned)__popcll(m));
        base = (unsigned)__shfl((int)base, 0, 64);
        if (used) {
            const unsigned at = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            keys_out[at] = w >> cbits; vals_out[at] = (unsigned)(w & ((1ull << cbits) - 1ull));
        }
    }
}

void gd_launch_contacts_insert(const ContactTab &t, const uint2 *pairs, unsigned long long cap_pairs, const unsigned long long *count,
                               unsigned long long max_count, unsigned R, hipStream_t st)
{
    if (!max_count) return;
    const unsigned nb = (unsigned)std::min<unsigned long long>((max_count + 255ull) / 256ull, 4096ull);
    hipLaunchKernelGGL(k_ct_insert, dim3(nb, R), dim3(256), 0, st, t, pairs, cap_pairs, count);
}

void gd_launch_contacts_rehash(const ContactTab &from, const ContactTab &to, unsigned R, hipStream_t st)
{
    const unsigned nb = (unsigned)std::min<unsigned long long>((from.cap + 255ull) / 256ull, 4096ull);
    hipLaunchKernelGGL(k_ct_rehash, dim3(nb, R), dim3(256), 0, st, from, to);
}

void gd_launch_contacts_compact(const ContactTab &t, unsigned r, unsigned long long *keys_out, unsigned *vals_out, unsigned *n_out, hipStream_t st)
{
    const unsigned nb = (unsigned)std::min<unsigned long long>((t.cap + 255ull) / 256ull, 8192ull);
    hipLaunchKernelGGL(k_ct_compact, dim3(nb), dim3(256), 0, st, t, r, keys_out, vals_out, n_out);
}

// ------------------------------------------------------- per-device set-up
// Every kernel that takes more than 64 KB of dynamic LDS (gfx950: 160 KB per CU) opts in HERE, once per device, with checked
// return codes -- called by gd_create under the per-device guard of gdyn_once.hpp, so that no launch on a device can precede it and
// two threads creating handles at the same time do not race (until round 5 each launcher did this behind an unsynchronised
// `static bool once`, per process: a second device never got the attribute).  The caller has made `device` current.
hipError_t gd_kernels_init_device(void)
{
    hipError_t first = hipSuccess;
    auto set = [&](const void *f, int bytes) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess && first == hipSuccess) first = e;
    };
#define GD_AS(MODE, PER, PK, S, SP) set(reinterpret_cast<const void *>(&k_step<MODE, PER, true, PK, S, SP>), 128 * 1024)
#define GD_AS_PK(MODE, PER, S, SP) GD_AS(MODE, PER, 0, S, SP); GD_AS(MODE, PER, 1, S, SP); GD_AS(MODE, PER, 2, S, SP)
#define GD_AS_ALL(MODE, SP) GD_AS_PK(MODE, false, false, SP); GD_AS_PK(MODE, false, true, SP); GD_AS_PK(MODE, true, false, SP); GD_AS_PK(MODE, true, true, SP)
    GD_AS_ALL(GD_MODE_STEP, false); GD_AS_ALL(GD_MODE_STEP, true);
    GD_AS_ALL(GD_MODE_FORCE, false); GD_AS_ALL(GD_MODE_ENERGY, false);
#undef GD_AS_ALL
#undef GD_AS_PK
#undef GD_AS
    set(reinterpret_cast<const void *>(&k_fill<false, true, false>), 128 * 1024);
    set(reinterpret_cast<const void *>(&k_fill<false, true, true>), 128 * 1024);
    set(reinterpret_cast<const void *>(&k_fill<true, true, false>), 128 * 1024);
    set(reinterpret_cast<const void *>(&k_fill<true, true, true>), 128 * 1024);
    set(reinterpret_cast<const void *>(&k_fill<false, true, false, true>), 128 * 1024);
    set(reinterpret_cast<const void *>(&k_fill<false, true, true, true>), 128 * 1024);
    set(reinterpret_cast<const void *>(&k_fill<true, true, false, true>), 128 * 1024);
    set(reinterpret_cast<const void *>(&k_fill<true, true, true, true>), 128 * 1024);
    set(reinterpret_cast<const void *>(&k_softwell<0>), 64 * 1024);
    set(reinterpret_cast<const void *>(&k_softwell<1>), 64 * 1024);
    set(reinterpret_cast<const void *>(&k_softwell<2>), 64 * 1024);
    return first;
}

// ------------------------------------------------------------------- misc

__global__ void k_gather_positions(const float4 *pos, const unsigned *slot_of, float4 *out, unsigned N, unsigned Np, int quantize)
{
    const unsigned r = blockIdx.y, o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= N) return;
    float4 x = pos[(size_t)r * Np + slot_of[(size_t)r * N + o]];
    if (quantize) {
        // simulation_common/simulation_store.cc:403-407: 16 fractional bits
        x.x = rintf(x.x * 65536.0f) * (1.0f / 65536.0f);
        x.y = rintf(x.y * 65536.0f) * (1.0f / 65536.0f);
        x.z = rintf(x.z * 65536.0f) * (1.0f / 65536.0f);
    }
    out[(size_t)r * N + o] = x;
}

// Snapshot download: bead order, xyz packed (12 bytes per bead cross PCIe), optional 2^-16 rounding on the device
__global__ void k_gather_xyz(const float4 *pos, const unsigned *slot_of, float *out, unsigned N, unsigned Np, int quantize)
{
    const unsigned r = blockIdx.y, o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= N) return;
    float4 x = pos[(size_t)r * Np + slot_of[(size_t)r * N + o]];
    // (simulation_store.cc:403-407 rounds float(val): with compensated positions val = pos + lo and float(pos + lo) == pos, the pair
    // being normalised -- the residual never enters a snapshot)
    if (quantize) {
        x.x = rintf(x.x * 65536.0f) * (1.0f / 65536.0f);
        x.y = rintf(x.y * 65536.0f) * (1.0f / 65536.0f);
        x.z = rintf(x.z * 65536.0f) * (1.0f / 65536.0f);
    }
    float *q = out + ((size_t)r * N + o) * 3;
    q[0] = x.x; q[1] = x.y; q[2] = x.z;
}

void gd_launch_gather_xyz(const float4 *pos, const unsigned *slot_of, float *out, unsigned N, unsigned Np, unsigned R, int quantize,
                          hipStream_t st)
{
    hipLaunchKernelGGL(k_gather_xyz, dim3((N + 255) / 256, R), dim3(256), 0, st, pos, slot_of, out, N, Np, quantize);
}

void gd_launch_gather_positions(const float4 *pos, const unsigned *slot_of, float4 *out, unsigned N, unsigned Np, unsigned R,
                                int quantize, hipStream_t st)
{
    hipLaunchKernelGGL(k_gather_positions, dim3((N + 255) / 256, R), dim3(256), 0, st, pos, slot_of, out, N, Np, quantize);
}

__global__ void k_identity(unsigned *orig, unsigned *slot_of, unsigned N, unsigned Np)
{
    const unsigned r = blockIdx.y, s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= Np) return;
    orig[(size_t)r * Np + s] = s < N ? s : 0xffffffffu;
    if (s < N) slot_of[(size_t)r * N + s] = s;
}

void gd_launch_identity(unsigned *orig, unsigned *slot_of, unsigned N, unsigned Np, unsigned R, hipStream_t st)
{
    hipLaunchKernelGGL(k_identity, dim3((Np + 255) / 256, R), dim3(256), 0, st, orig, slot_of, N, Np);
}
