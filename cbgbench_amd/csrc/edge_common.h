// libcbgx -- device helpers shared by the fused edge kernels (edge_mfma.hip: forward; train_bwd_x2h.hip: backward):
// cross-lane reductions, scalar-base / 32-bit-offset addressing, packed-fp32 helpers, 1-ulp hardware approximations, and
// the split-f16 pieces of the rbf pre-activation (rbf tuples, weight tuples).  Lane l: c = l & 15, q = l >> 4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "layout.h"

namespace cbgx {

typedef float floatx4 __attribute__((ext_vector_type(4)));


#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// split-f16 rbf pre-activation (protein destinations): v_mfma_f32_16x16x16_f16, four f16 per lane and operand
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
#define MFMAH(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16f16((a), (b), (c), 0, 0, 0)

// ---- cross-lane reductions without LDS traffic -----------------------------------------------------
// within a 16-lane row: DPP quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror (every lane gets the sum)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}
// across the 4 rows (q): v_permlane16_swap / v_permlane32_swap (gfx950); every lane gets the result
__device__ __forceinline__ float xrow_sum(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xrow_max(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float wave_sum(float v) { return xrow_sum(row16_sum(v)); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// Wave-uniform row base (pinned in scalar registers) + 32-bit per-lane BYTE offset: selects the
// `global_load v, v_off, s[base:base+1]` form -- one address VGPR and no 64-bit vector arithmetic per gathered row.  The
// empty asm makes the base opaque: without it LICM re-associates `base + lane part` into hoisted 64-bit per-lane pointers
// (two VGPRs per address, all live across the whole node loop).  Every offset used this way is < 2^32 (checked by the launcher).
typedef const __attribute__((address_space(1))) char* gptr;
typedef __attribute__((address_space(1))) char* gwptr;
__device__ __forceinline__ gptr sbase(const void* p) {
    uint64_t v = reinterpret_cast<uint64_t>(p);
    asm volatile("" : "+s"(v));
    return (gptr)v;
}
__device__ __forceinline__ gwptr sbase_w(void* p) {
    uint64_t v = reinterpret_cast<uint64_t>(p);
    asm volatile("" : "+s"(v));
    return (gwptr)v;
}
// loop-invariant lane offsets are laundered once per use site: the zero-extension then sits in the block of the load
// (instruction selection is per block) instead of being hoisted out of the node loop as a 64-bit register pair
__device__ __forceinline__ unsigned vop(unsigned off) {
    asm volatile("" : "+v"(off));
    return off;
}
typedef float nfloat2 __attribute__((ext_vector_type(2)));
typedef int nint4 __attribute__((ext_vector_type(4)));
#define CBGX_GLOBAL_AS(T) const __attribute__((address_space(1))) T*
__device__ __forceinline__ float4 ldo4(gptr base, unsigned byte_off) {
    const floatx4 v = *reinterpret_cast<CBGX_GLOBAL_AS(floatx4)>(base + byte_off);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float ldo1(gptr base, unsigned byte_off) {
    return *reinterpret_cast<CBGX_GLOBAL_AS(float)>(base + byte_off);
}
__device__ __forceinline__ float2 ldo2(gptr base, unsigned byte_off) {
    const nfloat2 v = *reinterpret_cast<CBGX_GLOBAL_AS(nfloat2)>(base + byte_off);
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ int ldoi(gptr base, unsigned byte_off) {
    return *reinterpret_cast<CBGX_GLOBAL_AS(int32_t)>(base + byte_off);
}
__device__ __forceinline__ int4 ldoi4(gptr base, unsigned byte_off) {
    const nint4 v = *reinterpret_cast<CBGX_GLOBAL_AS(nint4)>(base + byte_off);
    return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int ldob(gptr base, unsigned byte_off) {
    return *reinterpret_cast<CBGX_GLOBAL_AS(uint8_t)>(base + byte_off);
}
__device__ __forceinline__ void sto2(gwptr base, unsigned byte_off, float2 v) {
    *reinterpret_cast<__attribute__((address_space(1))) nfloat2*>(base + byte_off) = nfloat2{v.x, v.y};
}
// no-return fp32 atomic add in the same scalar-base + 32-bit-offset form (global_atomic_add_f32 v_off, v_data, s[base])
__device__ __forceinline__ void atomo(gwptr base, unsigned byte_off, float v) {
    __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(1))) float*>(base + byte_off), v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}
// packed fp32: v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 do two lanes' worth of work per issue slot, so the elementwise
// parts of the kernel (sum of squares, LayerNorm affine) are written on register pairs
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2v lo2(floatx4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ float2v hi2(floatx4 v) { return __builtin_shufflevector(v, v, 2, 3); }
__device__ __forceinline__ float2v splat2(float a) { float2v r = {a, a}; return r; }
// 1-ulp hardware approximations (v_rsq_f32, v_rcp_f32, v_sqrt_f32, v_exp_f32): the IEEE-exact library forms cost ~10
// instructions each and the parity tolerance (1e-4 rel) is three orders of magnitude away
__device__ __forceinline__ float fast_rsqrt(float a) { return __builtin_amdgcn_rsqf(a); }
__device__ __forceinline__ float fast_rcp(float a) { return __builtin_amdgcn_rcpf(a); }
__device__ __forceinline__ float fast_sqrt(float a) { return __builtin_amdgcn_sqrtf(a); }
__device__ __forceinline__ float fast_exp(float a) { return __builtin_amdgcn_exp2f(a * 1.44269504088896340736f); }
__device__ __forceinline__ floatx4 f4(float4 a) { floatx4 r = {a.x, a.y, a.z, a.w}; return r; }

// |x_i - x_j| with one fixed evaluation order (explicit FMAs, no re-association): the prologue and the pipelined loop body
// must give the same bits for the same edge, or a node's result would depend on its position in the work list
__device__ __forceinline__ float edge_len(float xi, float yi, float zi, float xj, float yj, float zj) {
    const float rx = xi - xj, ry = yi - yj, rz = zi - zj;
    return fast_sqrt(__builtin_fmaf(rx, rx, __builtin_fmaf(ry, ry, rz * rz)));
}
// The five rbf values of a lane (g = 4s + q, already masked to this pass's source class) as the four f16 tuples that pair
// with the weight tuples T1, T1, T2, T3 (load_wtuples below):  [rh0..3]  [rl0..3]  [rh4 rh0 rh1 rh2]  [rh3 rh4 rl4 0]
// (rh = f16(r) round to nearest, rl = f16(r - rh); the dropped rl * wl term is 2^-22 relative)
__device__ __forceinline__ void rbf_tuples(const float (&R)[5], half4 (&B)[4]) {
    _Float16 h[5], l[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        h[s] = (_Float16)R[s];
        l[s] = (_Float16)(R[s] - (float)h[s]);
    }
    B[0] = half4{h[0], h[1], h[2], h[3]};
    B[1] = half4{l[0], l[1], l[2], l[3]};
    B[2] = half4{h[4], h[0], h[1], h[2]};
    B[3] = half4{h[3], h[4], l[4], (_Float16)0.f};
}
// The weight pieces of (type, tile) for this lane: T1 = (d0, d1) = [h0 h1 h2 h3], T2 = (d2, d3) = [h4 l0 l1 l2],
// T3 = (d4, d2) = [l3 l4 h4 l0] -- d2 is used twice, so the table needs no duplicate and is exactly as large as the five
// fp32 fragments it replaces.  Layout of one type's 8 tiles (8 FRAG_BLK floats), chosen so that EVERY read is
// `table base + 16 lane + immediate`: two groups of four tiles, each group = four 1 KB planes [lane][d0 d1 d2 d3] followed by one
// 1 KB plane [lane][d4 of the group's four tiles].  One address register per table and source class; the tile offsets sit in the
// instructions' immediate fields.  (The former [d0 d1 | d2 d3 | d4] planes per tile needed a VALU add per read: 16-lane and
// 4-lane strides cannot share a base register, and ds_read2st64_b64 counts its offsets in 512-byte units.)
struct WTuples { half4 t1, t2, t3; };
constexpr int FRAG_GROUP = 4 * (int)FRAG_BLK;   // floats per group of four tiles
__device__ __forceinline__ int frag_d0_index(int t, int lane) { return (t >> 2) * FRAG_GROUP + (t & 3) * 256 + 4 * lane; }
__device__ __forceinline__ int frag_d4_index(int t, int lane) { return (t >> 2) * FRAG_GROUP + 1024 + 4 * lane + (t & 3); }
__device__ __forceinline__ WTuples load_wtuples(const float* type_base, int t, int lane) {
    typedef unsigned uint2v __attribute__((ext_vector_type(2)));
    typedef unsigned uint4v __attribute__((ext_vector_type(4)));
    WTuples w;
    const uint4v d = *reinterpret_cast<const uint4v*>(type_base + frag_d0_index(t, lane));
    const unsigned d4 = *reinterpret_cast<const unsigned*>(type_base + frag_d4_index(t, lane));
    const uint2v p1 = {d.x, d.y}, p2 = {d.z, d.w}, p3 = {d4, d.z};
    w.t1 = __builtin_bit_cast(half4, p1);
    w.t2 = __builtin_bit_cast(half4, p2);
    w.t3 = __builtin_bit_cast(half4, p3);
    return w;
}
// Range-safe split-f16 of the rbf pre-activation (layout.h A_RBF_SC): the weight tuples hold Wr 2^kw, the kernels produce the rbf
// values times 2^RBF_EXP (the 0 / 1 validity factor becomes 0 / 2^RBF_EXP) and carry the whole pre-activation tile scaled by
// S = 2^(kw + RBF_EXP): PD + PS and the type column are multiplied by S when they enter the accumulator, and LayerNorm divides
// it out again -- rstd = rsqrt(sum(acc^2) c1 + eps) with c1 = 1 / (H S^2), normalised value = acc (rstd c2) with c2 = 1 / S.
// Exact powers of two: no rounding is added anywhere, only the f16 exponent range is used where it has full precision.
struct RbfScale { float S, c1, c2; };
__device__ __forceinline__ RbfScale load_rbf_scale(const float* att, int kv) {
    RbfScale r;
    r.S = att[A_RBF_SC + 4 * kv]; r.c1 = att[A_RBF_SC + 4 * kv + 1]; r.c2 = att[A_RBF_SC + 4 * kv + 2];
    return r;
}
constexpr float RBF_UP = (float)(1 << RBF_EXP);

// XOR mask of the second v Linear's LDS rows (x2h epilogue): heads {0-3, 12-15} -> 0..7, heads {4-11} -> 8..15, so that the two
// half-rows of lanes a ds_read_b128 lane group combines never meet on a 16-byte slot (edge_mfma.hip, epilogue)
__host__ __device__ __forceinline__ int wbv_swizzle(int head) { return head < 4 ? head : (head < 12 ? head + 4 : head - 8); }

// edge type, unitransformer.py:92-97: (src lig, dst lig)->0, (lig, prot)->1, (prot, lig)->2, (prot, prot)->3
__device__ __forceinline__ int etype(bool src_lig, int lig_i) { return src_lig ? (lig_i ? 0 : 1) : (lig_i ? 2 : 3); }

}  // namespace cbgx
